"""mmq_probe.py — prefill mat-mul rate (MUL_MAT, n = 512 tokens) per weight type and Llama-3-8B shape; wall clock over repeated graphs."""
import os
import sys
import time

import numpy as np

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(_root, _d))
import oracle as orc
from gpu_util import QTYPES, backend, gg

L = gg.base(); be = backend()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
types = sys.argv[2].split(",") if len(sys.argv) > 2 else ["q4_K", "q6_K", "q8_0", "q4_0"]
for name in types:
    for (m, k) in [(4096, 4096), (14336, 4096), (4096, 14336)]:
        rng = np.random.default_rng(0)
        with gg.Context() as ctx:
            w = ctx.new_tensor(QTYPES[name], [k, m])
            b = ctx.new_tensor(gg.F32, [k, n])
            out = L.ggml_mul_mat(ctx.ctx, w, b)
            ctx.alloc(be)
            gg.tensor_set(w, orc.random_blocks(rng, QTYPES[name], (m,), k))
            gg.tensor_set(b, rng.uniform(-1, 1, size=(n, k)).astype(np.float32))
            g = gg.graph_of(ctx, out)
            for _ in range(3):
                be.compute(g)
            be.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                be.compute_async(g)
            be.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            print(f"{name} m={m} k={k} n={n}: {us:8.1f} us  {2.0*m*k*n/us/1e6:7.1f} TFLOP/s", flush=True)
