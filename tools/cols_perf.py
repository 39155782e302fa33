"""cols_perf.py — MUL_MAT of a K-quant weight matrix with 2..8 activation columns (the reference's perf shape m = 4096, k = 14336, and a
lm_head-like m = 32000, k = 4096): wall clock per one-node graph. Run under rocprofv3 --kernel-trace --stats for the kernel's own time."""
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle as orc
from gpu_util import QTYPES, backend, gg

L = gg.base(); be = backend()
types = sys.argv[1].split(",") if len(sys.argv) > 1 else ["q4_K", "q5_K", "q6_K"]
for name in types:
    for m, k in ((4096, 14336), (32000, 4096))[: int(sys.argv[2]) if len(sys.argv) > 2 else 2]:
        rng = np.random.default_rng(1234)
        wb = orc.random_blocks(rng, QTYPES[name], (m,), k)
        for n in (2, 4, 8):
            with gg.Context() as ctx:
                w = ctx.new_tensor(QTYPES[name], [k, m]); b = ctx.new_tensor(gg.F32, [k, n])
                o = L.ggml_mul_mat(ctx.ctx, w, b)
                ctx.alloc(be); gg.tensor_set(w, wb); gg.tensor_set(b, rng.uniform(-1, 1, size=(n, k)).astype(np.float32))
                g = gg.graph_of(ctx, o)
                for _ in range(3):
                    be.compute_async(g)
                be.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    be.compute_async(g)
                be.synchronize()
                us = (time.perf_counter() - t0)/50*1e6
            print(f"{name} m={m} k={k} n={n}: {us:.2f} us wall, {wb.nbytes/us/1e3:.0f} GB/s of weight bytes", flush=True)
