"""ops_ref.py — numpy (float64) restatement of the element ops on the decode path.
TEST INFRASTRUCTURE ONLY (same rule as ggml_oracle.c).

The op definitions live in the reference's missing ggml.c, so the formulas are
[UPSTREAM-KNOWLEDGE]; what is citeable is how the reference calls and tests each op
 — cited per function. Arrays use numpy order [ne3, ne2, ne1, ne0]
(the reverse of ggml's ne[]).
"""
from __future__ import annotations

import math

import numpy as np


def rms_norm(x: np.ndarray, eps: float) -> np.ndarray:
    """src/llama-graph.cpp:605; tests/test-backend-ops.cpp:2773 — y = x / sqrt(mean(x^2) + eps) per row of ne0"""
    x = x.astype(np.float64)
    return x / np.sqrt(np.mean(x * x, axis=-1, keepdims=True) + eps)


def bcast(b: np.ndarray, shape) -> np.ndarray:
    """ggml repeat-broadcast of b to `shape` (tests/test-backend-ops.cpp:2469)"""
    reps = [s // bs for s, bs in zip(shape, b.shape)]
    return np.tile(b, reps)


def silu(x):
    x = x.astype(np.float64)
    return x / (1.0 + np.exp(-x))


def swiglu(a, b):
    """src/llama-graph.cpp:691; tests/test-backend-ops.cpp:1832-1888 — silu(a) * b"""
    return silu(a) * b.astype(np.float64)


def swiglu_oai(a, b, alpha=1.702, limit=7.0):
    """src/llama-graph.cpp:961-968; tests/test-backend-ops.cpp:1890-1949"""
    x = np.minimum(a.astype(np.float64), limit)
    y = np.clip(b.astype(np.float64), -limit, limit)
    return (x / (1.0 + np.exp(-x * alpha))) * (y + 1.0)


def gelu(x):
    x = x.astype(np.float64)
    return 0.5 * x * (1.0 + np.tanh(0.79788456080286535587989211986876 * x * (1.0 + 0.044715 * x * x)))


def soft_max(x, mask=None, scale=1.0, max_bias=0.0, sinks=None):
    """src/llama-graph.cpp:1312-1313; tests/test-backend-ops.cpp:3569-3626.
    x [ne3, ne2(heads), ne1, ne0]; mask [ne3', ne2', >=ne1, ne0] broadcast over heads; sinks [ne2]."""
    x = x.astype(np.float64) * scale
    n3, n_head, n1, n0 = x.shape
    if mask is not None:
        m = mask.astype(np.float64)[:, :, :n1, :]
        m = np.tile(m, (n3 // m.shape[0], n_head // m.shape[1], 1, 1))
        if max_bias > 0.0:
            n_head_log2 = 1 << int(math.floor(math.log2(n_head)))
            m0 = 2.0 ** (-max_bias / n_head_log2)
            m1 = 2.0 ** (-(max_bias / 2.0) / n_head_log2)
            slope = np.array([m0 ** (h + 1) if h < n_head_log2 else m1 ** (2 * (h - n_head_log2) + 1) for h in range(n_head)])
            m = m * slope[None, :, None, None]
        x = x + m
    mx = x.max(axis=-1, keepdims=True)
    if sinks is not None:
        s = sinks.astype(np.float64)[None, :, None, None]
        mx = np.maximum(mx, s)
    e = np.exp(x - mx)
    den = e.sum(axis=-1, keepdims=True)
    if sinks is not None:
        den = den + np.exp(s - mx)
    return e / den


def _rope_corr_dims(n_dims, n_ctx_orig, freq_base, beta_fast, beta_slow):
    def corr_dim(n_rot):
        return n_dims * math.log(n_ctx_orig / (n_rot * 2 * math.pi)) / (2 * math.log(freq_base))
    start = math.floor(corr_dim(beta_fast))
    end = math.ceil(corr_dim(beta_slow))
    return max(0, start), min(n_dims - 1, end)


def rope(x, pos, n_dims, mode, n_ctx_orig=0, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0,
         beta_fast=32.0, beta_slow=1.0, freq_factors=None):
    """src/llama-model.cpp:6030-6040; tests/test-backend-ops.cpp:3660-3782.
    x [ne3, ne2(tokens), ne1(heads), ne0(head dim)]; pos [ne2]; NORM rotates (2i, 2i+1), NEOX (i, i + n_dims/2)."""
    x = x.astype(np.float64)
    out = x.copy()
    neox = (mode & 2) != 0
    theta_scale = freq_base ** (-2.0 / n_dims)
    lo, hi = _rope_corr_dims(n_dims, n_ctx_orig, freq_base, beta_fast, beta_slow) if ext_factor != 0.0 else (0, 0)
    half = n_dims // 2
    ip = np.arange(half)
    for t in range(x.shape[1]):
        theta_extrap = float(pos[t]) * theta_scale ** ip
        if freq_factors is not None:
            theta_extrap = theta_extrap / freq_factors[:half].astype(np.float64)
        theta_interp = freq_scale * theta_extrap
        theta = theta_interp
        mscale = attn_factor
        if ext_factor != 0.0:
            y = (ip - lo) / max(0.001, hi - lo)
            ramp_mix = (1.0 - np.clip(y, 0.0, 1.0)) * ext_factor
            theta = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix
            mscale = attn_factor * (1.0 + 0.1 * math.log(1.0 / freq_scale))
        c, s = np.cos(theta) * mscale, np.sin(theta) * mscale
        if neox:
            x0, x1 = x[:, t, :, :half], x[:, t, :, half:n_dims]
            out[:, t, :, :half] = x0 * c - x1 * s
            out[:, t, :, half:n_dims] = x0 * s + x1 * c
        else:
            x0, x1 = x[:, t, :, 0:n_dims:2], x[:, t, :, 1:n_dims:2]
            out[:, t, :, 0:n_dims:2] = x0 * c - x1 * s
            out[:, t, :, 1:n_dims:2] = x0 * s + x1 * c
    return out


def set_rows(dst, src, idx):
    """src/llama-kv-cache-unified.cpp:1123; tests/test-backend-ops.cpp:2060-2127.
    dst [ne3, ne2, nrows, ne0]; src [ne3, ne2, ne1, ne0]; idx i64 [ne12, ne11, ne1] broadcast over (ne2 % ne11, ne3 % ne12)"""
    out = dst.copy()
    n3, n2, n1, _ = src.shape
    for i3 in range(n3):
        for i2 in range(n2):
            for i1 in range(n1):
                r = int(idx[i3 % idx.shape[0], i2 % idx.shape[1], i1])
                out[i3, i2, r, :] = src[i3, i2, i1, :].astype(out.dtype)
    return out


def get_rows(src, idx):
    """tests/test-backend-ops.cpp:1951 — src [ne3?, ne2, nrows, ne0], idx i32 [ne12, ne11, ne10] -> [ne12, ne11, ne10, ne0]"""
    n12, n11, n10 = idx.shape
    out = np.empty((n12, n11, n10, src.shape[-1]), dtype=np.float64)
    for i12 in range(n12):
        for i11 in range(n11):
            for i10 in range(n10):
                out[i12, i11, i10] = src[i12 if src.shape[0] > 1 else 0, i11, int(idx[i12, i11, i10])]
    return out


def add_id(a, bias, ids):
    """src/llama-graph.cpp:927; tests/test-backend-ops.cpp:2548 — a [n_tok, n_used, ne0], bias [n_expert, ne0], ids [n_tok, n_used]"""
    return a.astype(np.float64) + bias.astype(np.float64)[ids]


def argsort_desc(x):
    """tests/test-backend-ops.cpp:4120 — indices sorting each row of ne0 in descending order"""
    return np.argsort(-x, axis=-1, kind="stable").astype(np.int32)


def mul_mat_dense(a, b):
    """tests/test-backend-ops.cpp:3127-3191 — a [ne03, ne02, m, k], b [ne13, ne12, n, k] -> [ne13, ne12, n, m], broadcast r2/r3"""
    a = a.astype(np.float64); b = b.astype(np.float64)
    r3, r2 = b.shape[0] // a.shape[0], b.shape[1] // a.shape[1]
    out = np.empty((b.shape[0], b.shape[1], b.shape[2], a.shape[2]))
    for i3 in range(b.shape[0]):
        for i2 in range(b.shape[1]):
            out[i3, i2] = b[i3, i2] @ a[i3 // r3, i2 // r2].T
    return out
