"""i8_check.py — the int8 matrix-core prefill kernel (csrc/mmq_i8.hip, GGML_MI355X_MMQ_I8=1) against the oracle's CPU-style and exact products."""
import os
import sys

import numpy as np

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(_root, _d))
os.environ.setdefault("GGML_MI355X_MMQ_I8", "1")
import oracle as orc
from gpu_util import QTYPES, backend, gg, run_mul_mat

be = backend()
rng = np.random.default_rng(3)
worst = 0.0
for (m, k, n) in [(256, 1024, 64), (192, 512, 100), (128, 768, 33), (64, 256, 9), (1024, 4096, 512), (320, 2048, 257)]:
    w = orc.random_blocks(rng, QTYPES["q4_K"], (m,), k, scale=1.0/np.sqrt(k))
    x = rng.standard_normal((n, k)).astype(np.float32)
    be.reset_counters()
    got = run_mul_mat(QTYPES["q4_K"], w, x, m, k)
    ec = orc.mul_mat_2d(w, QTYPES["q4_K"], x, "cpu"); ee = orc.mul_mat_2d(w, QTYPES["q4_K"], x, "exact")
    a, b = orc.nmse(ec, got), orc.nmse(ee, got)
    worst = max(worst, a)
    print(f"m={m} k={k} n={n}: nmse vs cpu-style {a:.3e}, vs exact {b:.3e} (cpu-style vs exact {orc.nmse(ee, ec):.3e})", flush=True)
assert worst <= 1e-6, worst
print("I8 OK")
