"""i8_one.py — one Q4_K mat-mul shape through the int8 prefill kernel, a few launches (for rocprofv3 runs)."""
import os, sys
import numpy as np
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(_root, _d))
import oracle as orc
from gpu_util import QTYPES, backend, gg
L = gg.base(); be = backend()
m, k, n = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (14336, 4096, 512)))
rng = np.random.default_rng(0)
with gg.Context() as ctx:
    w = ctx.new_tensor(QTYPES["q4_K"], [k, m]); b = ctx.new_tensor(gg.F32, [k, n])
    out = L.ggml_mul_mat(ctx.ctx, w, b)
    ctx.alloc(be)
    gg.tensor_set(w, orc.random_blocks(rng, QTYPES["q4_K"], (m,), k)); gg.tensor_set(b, rng.uniform(-1, 1, size=(n, k)).astype(np.float32))
    g = gg.graph_of(ctx, out)
    for _ in range(6):
        be.compute(g)
    be.synchronize()
