#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own Python
definition of the block formats (gguf-py/gguf/quants.py), which its test
gguf-py/tests/test_quants.py:116-141 declares bit-exact to ggml's C code.

Run ONLY in the build container (the reference tree does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The committed .npz files are DATA (inputs + expected outputs). Nothing here or in
tests/ copies reference source text.

Recipes follow the reference's tests:
  * random float16 bit patterns viewed as block bytes — gguf-py/tests/test_quants.py:221-225
  * a zero row in the quantizer input                 — gguf-py/tests/test_quants.py:150-151
  * synthetic 0.1 + 2*cos(i + offset) rows            — tests/test-quantize-fns.cpp:31-35
  * MUL_MAT shapes m=16, n=1..9(16), k in {256, 1024}   — tests/test-backend-ops.cpp:5709-5761
  * MUL_MAT_ID ids = shuffled permutation per row     — tests/test-backend-ops.cpp:3247-3266
"""
import os
import sys
from pathlib import Path

import numpy as np

REF = Path(os.environ.get("GGUF_PY", "/root/reference/gguf-py"))
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))
import gguf  # noqa: E402
from gguf.constants import GGMLQuantizationType as T  # noqa: E402

OUT = Path(__file__).resolve().parent
TYPES = {"q4_0": T.Q4_0, "q8_0": T.Q8_0, "q4_K": T.Q4_K, "q5_K": T.Q5_K, "q6_K": T.Q6_K, "mxfp4": T.MXFP4}
HAS_QUANT = ("q4_0", "q8_0", "mxfp4")


def f16_field_offsets(name):
    # byte offsets of the f16 super-scales inside a block (gguf-py/gguf/quants.py dequantize_blocks hsplit order)
    return {"q4_0": [0], "q8_0": [0], "q4_K": [0, 2], "q5_K": [0, 2], "q6_K": [208], "mxfp4": []}[name]


def sanitize(blocks, name, rng):
    """make the f16 scale fields finite so the expected output holds no NaN/inf"""
    for off in f16_field_offsets(name):
        d = rng.uniform(-2.0, 2.0, size=blocks.shape[0]).astype(np.float16)
        blocks[:, off:off + 2] = d.view(np.uint8).reshape(-1, 2)
    if name == "mxfp4":
        blocks[:, 0] = rng.integers(100, 140, size=blocks.shape[0], dtype=np.uint8)
    return blocks


def edge_blocks(name, ts):
    e = []
    z = np.zeros(ts, dtype=np.uint8); e.append(z)                        # all-zero block
    f = np.full(ts, 0xFF, dtype=np.uint8); e.append(f.copy())            # all bits set (scales NaN for f16 -> sanitized below)
    a = np.full(ts, 0xAA, dtype=np.uint8); e.append(a.copy())
    s = np.full(ts, 0x55, dtype=np.uint8); e.append(s.copy())
    blocks = np.stack(e)
    one = np.array([1.0], dtype=np.float16).view(np.uint8)
    sub = np.array([6e-8], dtype=np.float16).view(np.uint8)              # f16 subnormal super-scale
    big = np.array([1024.0], dtype=np.float16).view(np.uint8)
    for off in f16_field_offsets(name):
        blocks[1, off:off + 2] = one
        blocks[2, off:off + 2] = sub
        blocks[3, off:off + 2] = big
    if name == "mxfp4":
        blocks[0, 0] = 0; blocks[1, 0] = 1; blocks[2, 0] = 127; blocks[3, 0] = 250   # e8m0 edge exponents (quants.py:663-665); 254 would overflow f32
    return blocks


def main():
    rng = np.random.default_rng(20250815)
    for name, qt in TYPES.items():
        bs, ts = gguf.GGML_QUANT_SIZES[qt]
        # --- dequant goldens ------------------------------------------------
        n_f16 = 24
        nb_bytes = n_f16 * ts + (n_f16 * ts) % 2
        rq = rng.random(nb_bytes // 2).astype(np.float16).view(np.uint8)[: n_f16 * ts].reshape(n_f16, ts).copy()
        if name == "mxfp4":
            rq[:, 0] = np.minimum(rq[:, 0], 200)                       # keep 12*2^(e-128) finite
        rb = sanitize(rng.integers(0, 256, size=(24, ts), dtype=np.uint8), name, rng)
        blocks = np.concatenate([rq, rb, edge_blocks(name, ts)], axis=0)
        expected = gguf.quants.dequantize(blocks, qt).astype(np.float32)
        assert expected.shape == (blocks.shape[0], bs) and np.isfinite(expected).all(), name
        np.savez_compressed(OUT / f"dequant_{name}.npz", blocks=blocks, expected=expected)

        # --- quantizer goldens ------------------------------------------------
        if name in HAS_QUANT:
            k = 256
            x = rng.standard_normal((12, k)).astype(np.float32)
            x[0, :] = 0                                                # zero blocks
            x[1, :] = (0.1 + 2 * np.cos(np.arange(k, dtype=np.float32) + 0.0)).astype(np.float32)
            x[2, :] = (0.1 + 2 * np.cos(np.arange(k, dtype=np.float32) + 1.0)).astype(np.float32)
            x[3, :] *= 1e-6
            x[4, :] *= 1e4
            x[5, :32] = np.linspace(-1, 1, 32, dtype=np.float32)       # ties / symmetric extrema
            x[5, 32:64] = -x[5, :32]
            q = gguf.quants.quantize(x, qt)
            np.savez_compressed(OUT / f"quant_{name}.npz", x=x, expected=q)

        # --- MUL_MAT goldens: dequant(W) @ X^T in float64 -----------------------
        for k in (256, 1024):
            m, n = 16, 16
            if name in HAS_QUANT:
                wf = rng.uniform(-1, 1, size=(m, k)).astype(np.float32)
                w = gguf.quants.quantize(wf, qt)
            else:
                w = sanitize(rng.integers(0, 256, size=(m * (k // bs), ts), dtype=np.uint8), name, rng)
                # keep magnitudes tame: super-scales in [2^-8, 2^-4]
                for off in f16_field_offsets(name):
                    d = (2.0 ** rng.uniform(-8, -4, size=w.shape[0])).astype(np.float16)
                    w[:, off:off + 2] = d.view(np.uint8).reshape(-1, 2)
                w = w.reshape(m, -1)
            x = rng.uniform(-1, 1, size=(n, k)).astype(np.float32)
            wd = gguf.quants.dequantize(w, qt).astype(np.float64)
            y = x.astype(np.float64) @ wd.T                              # [n, m]
            np.savez_compressed(OUT / f"mulmat_{name}_k{k}.npz", w=w, x=x, expected=y)

        # --- MUL_MAT_ID golden --------------------------------------------------
        n_mats, n_used, m, n, k = 8, 2, 32, 5, 256
        if name in HAS_QUANT:
            w = gguf.quants.quantize(rng.uniform(-1, 1, size=(n_mats, m, k)).astype(np.float32), qt)
        else:
            w = sanitize(rng.integers(0, 256, size=(n_mats * m * (k // bs), ts), dtype=np.uint8), name, rng)
            for off in f16_field_offsets(name):
                d = (2.0 ** rng.uniform(-8, -4, size=w.shape[0])).astype(np.float16)
                w[:, off:off + 2] = d.view(np.uint8).reshape(-1, 2)
            w = w.reshape(n_mats, m, -1)
        ids = np.stack([rng.permutation(n_mats)[:n_used] for _ in range(n)]).astype(np.int32)
        b = rng.uniform(-1, 1, size=(n, n_used, k)).astype(np.float32)
        wd = gguf.quants.dequantize(w, qt).astype(np.float64)            # [n_mats, m, k]
        y = np.einsum("tumk,tuk->tum", wd[ids], b.astype(np.float64))
        np.savez_compressed(OUT / f"mulmatid_{name}.npz", w=w, b=b, ids=ids, expected=y)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
