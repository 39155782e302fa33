"""ad-hoc perf look (not a test): op-level mat-vec rates at the reference's perf shape (tests/test-backend-ops.cpp:6190-6196)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle as orc
from gpu_util import QTYPES, backend, gg, proc

L = gg.base(); be = backend()
hbm = proc("ggml_backend_mi355x_test_hbm_read_gbps", C.c_double, [C.c_void_p, C.c_size_t, C.c_int])
print("hbm read probe GB/s:", [round(hbm(be.be, 1 << 30, 10), 1) for _ in range(3)], flush=True)
for name, (m, k) in [("q4_K", (4096, 14336)), ("q6_K", (4096, 14336)), ("q4_K", (14336, 4096)), ("q4_K", (4096, 4096)), ("q6_K", (128256, 4096)),
                     ("q8_0", (4096, 14336)), ("q4_0", (4096, 14336)), ("q5_K", (4096, 14336)), ("mxfp4", (2880, 2880))]:
    rng = np.random.default_rng(0)
    NW = 12 if m < 100000 else 2
    with gg.Context() as ctx:
        ws = [ctx.new_tensor(QTYPES[name], [k, m]) for _ in range(NW)]
        b = ctx.new_tensor(gg.F32, [k, 1])
        outs = [L.ggml_mul_mat(ctx.ctx, w, b) for w in ws]
        ctx.alloc(be)
        wb = orc.random_blocks(rng, QTYPES[name], (m,), k)
        for w in ws:
            gg.tensor_set(w, wb)
        gg.tensor_set(b, rng.uniform(-1, 1, size=(1, k)).astype(np.float32))
        g = gg.graph_of(ctx, *outs)
        be.set_option("profile", 1)
        for _ in range(5):
            be.compute(g)
        pr = be.profile()[0]
        be.set_option("profile", 0)
        us = pr["total_ms"] / pr["launches"] * 1e3
        print(f"{name} m={m} k={k}: {us:.1f} us/launch (events)  {pr['bytes_per_launch']/us/1e3:.0f} GB/s", flush=True)
