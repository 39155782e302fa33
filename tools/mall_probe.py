"""mall_probe.py — does a mat-vec whose weights were just read (Infinity-Cache resident) run faster than one streaming from HBM?
NW distinct weight tensors are cycled; NW x bytes <= 256 MB keeps the set inside the Infinity Cache."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle as orc
from gpu_util import QTYPES, backend, gg

L = gg.base(); be = backend()
for name, (m, k) in [("q4_K", (14336, 4096)), ("q4_K", (4096, 14336)), ("q4_K", (4096, 4096)), ("q6_K", (4096, 14336))]:
    for NW in (1, 2, 4, 8, 16, 32):
        rng = np.random.default_rng(0)
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
            for _ in range(max(2, 32 // NW)):
                be.compute(g)
            pr = be.profile()[0]
            be.set_option("profile", 0)
            us = pr["total_ms"] / pr["launches"] * 1e3
            print(f"{name} m={m} k={k} NW={NW:2d} set={NW*pr['bytes_per_launch']/1e6:7.1f} MB: {us:6.1f} us/launch  {pr['bytes_per_launch']/us/1e3:6.0f} GB/s", flush=True)
