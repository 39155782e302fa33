"""graft_pkg.py — imports the package directory `llama.cpp-gfx906_amd/` (not a valid identifier)
under the alias `llama_cpp_gfx906_amd`."""
import importlib.util
import sys
from pathlib import Path

_ALIAS = "llama_cpp_gfx906_amd"
_DIR = Path(__file__).resolve().parent / "llama.cpp-gfx906_amd"


def load():
    if _ALIAS in sys.modules:
        return sys.modules[_ALIAS]
    spec = importlib.util.spec_from_file_location(_ALIAS, _DIR / "__init__.py", submodule_search_locations=[str(_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
