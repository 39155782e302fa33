"""mmvq_stream_probe.py — streaming rate of one big mat-vec (fixed costs negligible) per type and prefetch depth (GGML_MI355X_MMVQ_DEPTH)."""
import sys

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle as orc
from gpu_util import QTYPES, backend, gg

L = gg.base(); be = backend()
for name, (m, k) in [("q4_K", (131072, 4096)), ("q6_K", (131072, 4096)), ("q5_K", (131072, 4096)), ("q8_0", (131072, 4096)), ("q4_0", (131072, 4096)),
                     ("q4_K", (32768, 14336)), ("q6_K", (32768, 14336))]:
    rng = np.random.default_rng(0)
    with gg.Context() as ctx:
        w = ctx.new_tensor(QTYPES[name], [k, m])
        b = ctx.new_tensor(gg.F32, [k, 1])
        out = L.ggml_mul_mat(ctx.ctx, w, b)
        ctx.alloc(be)
        wb = orc.random_blocks(rng, QTYPES[name], (1024,), k)
        gg.tensor_set(w, np.tile(wb, (m // 1024, 1)))
        gg.tensor_set(b, rng.uniform(-1, 1, size=(1, k)).astype(np.float32))
        g = gg.graph_of(ctx, out)
        be.set_option("profile", 1)
        for _ in range(6):
            be.compute(g)
        pr = be.profile()[0]
        be.set_option("profile", 0)
        us = pr["total_ms"] / pr["launches"] * 1e3
        print(f"{name} m={m} k={k}: {us:7.1f} us/launch  {pr['bytes_per_launch']/us/1e3:6.0f} GB/s", flush=True)
