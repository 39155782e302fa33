"""llama.cpp-gfx906_amd — MI355X-native ggml backend for llama.cpp's quantized mat-mul hot path.

Holds only what the path needs: csrc/ (HIP kernels + the ggml backend C-ABI),
build.py (hipcc, gfx950) and ggml_ctypes.py (host-side mirror of the ggml API the reference's
tests drive a backend with). The directory name is not a Python identifier; import it through
`graft_pkg.load()` at the repo root (alias `llama_cpp_gfx906_amd`).
"""
from . import ggml_ctypes as ggml  # noqa: F401
from . import llama_synth  # noqa: F401
from . import layer_split  # noqa: F401
from . import build as _build


def build_native(verbose=False):
    return _build.build(verbose=verbose)
