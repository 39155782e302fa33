"""oracle.py — numpy-facing wrapper around oracle/libggml_oracle.so.

TEST INFRASTRUCTURE ONLY (see the header of ggml_oracle.c): importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.

What each function restates is cited in ggml_oracle.c. The broadcast rules of
MUL_MAT / MUL_MAT_ID implemented here follow the reference's op tests:
tests/test-backend-ops.cpp:3127-3191 (MUL_MAT, bs/nr broadcast) and
tests/test-backend-ops.cpp:3226-3266 (MUL_MAT_ID).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

# ggml_type ids — gguf-py/gguf/constants.py:2698-2730
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, BF16, MXFP4 = 0, 1, 2, 8, 12, 13, 14, 15, 30, 39
I32, I64 = 26, 27
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K",
              Q6_K: "q6_K", Q8_K: "q8_K", MXFP4: "mxfp4"}
QUANT_TYPES = (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, MXFP4)
# (block size, type size) — gguf-py/gguf/constants.py:2839-2872
QUANT_SIZES = {F32: (1, 4), F16: (1, 2), BF16: (1, 2), Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176),
               Q6_K: (256, 210), Q8_K: (256, 292), MXFP4: (32, 17)}


def build(native: bool = False, out_dir: Path | None = None) -> Path:
    """Compile the C restatement (gcc). native=True adds -march=native (cpu_baseline leg)."""
    out_dir = Path(out_dir) if out_dir else _HERE
    out = out_dir / ("libggml_oracle_native.so" if native else "libggml_oracle.so")
    src = _HERE / "ggml_oracle.c"
    if out.exists() and out.stat().st_mtime >= src.stat().st_mtime:
        return out
    cmd = ["gcc", "-O2", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-std=c11",
           "-march=native" if native else "-march=x86-64-v3", "-shared", "-o", str(out), str(src), "-lm"]
    subprocess.run(cmd, check=True)
    return out


_lib = None


def cpu_budget(cap: int = 16) -> int:
    """Threads worth starting: the affinity mask, the cgroup CPU quota when one is set, and `cap` (a GPU box hands each job about 16
    CPUs of a 256-thread host; OpenMP's default of one spinning thread per visible CPU made small oracle calls ~100x slower there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, min(n, cap))


def lib(path: Path | None = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    # before libgomp initialises: a bounded team that sleeps between the many small parallel regions of a test run
    os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    p = Path(path) if path else build()
    L = ctypes.CDLL(str(p))
    c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    L.orc_blck_size.restype = c_i64; L.orc_blck_size.argtypes = [c_int]
    L.orc_type_size.restype = c_i64; L.orc_type_size.argtypes = [c_int]
    L.orc_row_size.restype = c_i64; L.orc_row_size.argtypes = [c_int, c_i64]
    L.orc_dequantize_row.restype = c_int; L.orc_dequantize_row.argtypes = [c_int, c_vp, c_vp, c_i64]
    L.orc_quantize_row.restype = c_int; L.orc_quantize_row.argtypes = [c_int, c_vp, c_vp, c_i64]
    L.orc_vec_dot_type.restype = c_int; L.orc_vec_dot_type.argtypes = [c_int]
    L.orc_vec_dot.restype = ctypes.c_float; L.orc_vec_dot.argtypes = [c_int, c_i64, c_vp, c_vp]
    L.orc_mul_mat.restype = c_int; L.orc_mul_mat.argtypes = [c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int]
    L.orc_mul_mat_q.restype = c_int; L.orc_mul_mat_q.argtypes = [c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64]
    L.orc_set_simd.restype = None; L.orc_set_simd.argtypes = [c_int]
    if path is None:
        _lib = L
    return L


def _ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def row_size(qtype: int, k: int) -> int:
    bs, ts = QUANT_SIZES[qtype]
    assert k % bs == 0, f"k={k} not a multiple of block size {bs}"
    return k // bs * ts


def dequantize(data: np.ndarray, qtype: int) -> np.ndarray:
    """bytes [..., row_bytes] -> f32 [..., k]  (to_float of the type traits)"""
    data = np.ascontiguousarray(data).view(np.uint8)
    bs, ts = QUANT_SIZES[qtype]
    assert data.shape[-1] % ts == 0
    k = data.shape[-1] // ts * bs
    out = np.empty(data.shape[:-1] + (k,), dtype=np.float32)
    rc = lib().orc_dequantize_row(qtype, _ptr(data), _ptr(out), out.size)
    assert rc == 0, f"dequantize: unsupported type {qtype}"
    return out


def quantize(x: np.ndarray, qtype: int) -> np.ndarray:
    """f32 [..., k] -> bytes [..., row_bytes]  (from_float_ref; Q4_0/Q8_0/MXFP4/Q8_K/F16 only)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    k = x.shape[-1]
    out = np.empty(x.shape[:-1] + (row_size(qtype, k),), dtype=np.uint8)
    rc = lib().orc_quantize_row(qtype, _ptr(x), _ptr(out), x.size)
    assert rc == 0, f"quantize: no reference quantizer for type {qtype}"
    return out


def set_simd(on: bool):
    """the AVX2 forms of the Q4_K / Q6_K dots on (default where the CPU has AVX2) or off; both give bit-identical results"""
    lib().orc_set_simd(int(on))


def vec_dot_type(qtype: int) -> int:
    return lib().orc_vec_dot_type(qtype)


def vec_dot(qtype: int, xrow: np.ndarray, yrow_q: np.ndarray, k: int) -> float:
    return float(lib().orc_vec_dot(qtype, k, _ptr(np.ascontiguousarray(xrow)), _ptr(np.ascontiguousarray(yrow_q))))


def mul_mat_2d(a: np.ndarray, qtype: int, b: np.ndarray, mode: str = "exact") -> np.ndarray:
    """a: bytes [m, row_bytes] (or f32/f16 [m, k]); b: f32 [n, k] -> f32 [n, m]."""
    b = np.ascontiguousarray(b, dtype=np.float32)
    n, k = b.shape
    a = np.ascontiguousarray(a)
    m = a.shape[0]
    dst = np.empty((n, m), dtype=np.float32)
    rc = lib().orc_mul_mat(qtype, _ptr(a), _ptr(b), _ptr(dst), m, n, k, 0 if mode == "exact" else 1)
    assert rc == 0
    return dst


def mul_mat(a: np.ndarray, qtype: int, b: np.ndarray, mode: str = "exact") -> np.ndarray:
    """ggml MUL_MAT with broadcast (tests/test-backend-ops.cpp:3140-3141).

    a: [ne03, ne02, m, row_bytes|k]  b: f32 [ne13, ne12, n, k]  ->  f32 [ne13, ne12, n, m]
    (numpy order = reversed ggml ne order); ne12 % ne02 == 0, ne13 % ne03 == 0.
    """
    assert a.ndim == 4 and b.ndim == 4
    ne03, ne02 = a.shape[0], a.shape[1]
    ne13, ne12, n, _ = b.shape
    assert ne12 % ne02 == 0 and ne13 % ne03 == 0
    r2, r3 = ne12 // ne02, ne13 // ne03
    m = a.shape[2]
    out = np.empty((ne13, ne12, n, m), dtype=np.float32)
    for i3 in range(ne13):
        for i2 in range(ne12):
            out[i3, i2] = mul_mat_2d(a[i3 // r3, i2 // r2], qtype, b[i3, i2], mode)
    return out


def mul_mat_id(as_: np.ndarray, qtype: int, b: np.ndarray, ids: np.ndarray, mode: str = "exact") -> np.ndarray:
    """ggml MUL_MAT_ID (tests/test-backend-ops.cpp:3226-3245; src/llama-graph.cpp:569-595).

    as_: [n_expert, m, row_bytes]; b: f32 [n_tokens, n_b, k] with n_b in {1, n_used};
    ids: i32 [n_tokens, n_used]  ->  f32 [n_tokens, n_used, m]
    dst[t, u, :] = as_[ids[t, u]] @ b[t, u % n_b, :]
    """
    n_tokens, n_used = ids.shape
    m = as_.shape[1]
    n_b = b.shape[1]
    out = np.empty((n_tokens, n_used, m), dtype=np.float32)
    for t in range(n_tokens):
        for u in range(n_used):
            e = int(ids[t, u])
            assert 0 <= e < as_.shape[0]
            out[t, u] = mul_mat_2d(as_[e], qtype, b[t, u % n_b][None, :], mode)[0]
    return out


# ---------------------------------------------------------------------------
# synthetic weights: random *valid* blocks (SURVEY.md §8d "Concrete synthetic inputs")
# ---------------------------------------------------------------------------
def random_blocks(rng: np.random.Generator, qtype: int, shape_rows: tuple, k: int, scale: float = 1.0) -> np.ndarray:
    """Random valid block bytes for a [*shape_rows, k] tensor of qtype.

    Quantized types without an off-ggml quantizer (K-quants) get random
    quants/scales with f16 super-scales sized so dequantized values are O(scale);
    Q4_0/Q8_0/MXFP4 are produced by the pinned reference quantizers from U(-1,1).
    """
    nrows = int(np.prod(shape_rows)) if shape_rows else 1
    bs, ts = QUANT_SIZES[qtype]
    nb = k // bs
    if qtype in (Q4_0, Q8_0, MXFP4):
        x = rng.uniform(-scale, scale, size=(nrows, k)).astype(np.float32)
        return quantize(x, qtype).reshape(tuple(shape_rows) + (nb * ts,))
    blocks = rng.integers(0, 256, size=(nrows * nb, ts), dtype=np.uint8)
    if qtype in (Q4_K, Q5_K):
        # x = d*sc*q - dmin*m, sc,m in [0,63], q in [0,15|31]
        qmax = 15 if qtype == Q4_K else 31
        d = rng.uniform(0.5, 1.0, size=nrows * nb) * scale / (63 * qmax) * 2.0
        dmin = rng.uniform(0.5, 1.0, size=nrows * nb) * scale / 63
        blocks[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
        blocks[:, 2:4] = dmin.astype(np.float16).view(np.uint8).reshape(-1, 2)
    elif qtype == Q6_K:
        # x = d*sc*(q-32), sc int8, q in [0,63]
        d = rng.uniform(0.5, 1.0, size=nrows * nb) * scale / (127 * 32)
        blocks[:, 208:210] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    else:
        raise ValueError(qtype)
    return blocks.reshape(tuple(shape_rows) + (nb * ts,))


def nmse(a: np.ndarray, b: np.ndarray) -> float:
    """normalized mean squared error = mse(a, b) / mse(a, 0) — tests/test-backend-ops.cpp:180-193"""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    den = float(np.sum(a * a))
    num = float(np.sum((a - b) ** 2))
    return num / den if den > 0 else (0.0 if num == 0 else float("inf"))
