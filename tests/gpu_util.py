"""helpers shared by the -m gpu tests: everything goes through the ggml C-ABI of libggml-mi355x.so."""
import ctypes as C

import numpy as np

import graft_pkg
import oracle as orc

pkg = graft_pkg.load()
gg = pkg.ggml

QTYPES = {"q4_0": gg.Q4_0, "q8_0": gg.Q8_0, "q4_K": gg.Q4_K, "q5_K": gg.Q5_K, "q6_K": gg.Q6_K, "mxfp4": gg.MXFP4}

_backend = None


def backend():
    global _backend
    if _backend is None:
        _backend = gg.Backend(0)
    return _backend


def proc(name, restype, argtypes):
    L = gg.base()
    p = L.ggml_backend_reg_get_proc_address(backend().reg, name.encode())
    assert p, f"proc {name} not exported"
    return C.CFUNCTYPE(restype, *argtypes)(p)


SENTINEL_ELEMS = 1024      # tests/test-backend-ops.cpp:1005-1018: sentinel tensors next to every tensor catch out-of-bounds writes


def _sentinels(ctx, n=2):
    return [ctx.new_tensor(gg.F32, [SENTINEL_ELEMS], f"sent{i}") for i in range(n)]


def run_mul_mat(qtype, w_bytes, x, m, k):
    """w_bytes [m, row_bytes] u8, x f32 [n, k] -> f32 [n, m] via GGML_OP_MUL_MAT on the device.
    Sentinel tensors sit before and after the result in the buffer (the kernels do wave-wide clamped loads and per-row stores near the
    ends of rows): they must come back untouched (tests/test-backend-ops.cpp:1005-1018,1173-1185)."""
    L = gg.base()
    be = backend()
    n = x.shape[0]
    rng = np.random.default_rng(99)
    with gg.Context() as ctx:
        a = ctx.new_tensor(qtype, [k, m], "a")
        b = ctx.new_tensor(gg.F32, [k, n], "b")
        s_before = _sentinels(ctx, 1)
        out = L.ggml_mul_mat(ctx.ctx, a, b)
        s_after = _sentinels(ctx, 1)
        assert be.supports_op(out)
        assert ctx.alloc(be)
        pat = [rng.standard_normal((1, SENTINEL_ELEMS)).astype(np.float32) for _ in s_before + s_after]
        for t, pv in zip(s_before + s_after, pat):
            gg.tensor_set(t, pv)
        gg.tensor_set(a, w_bytes)
        gg.tensor_set(b, x)
        be.compute(gg.graph_of(ctx, out))
        res = gg.tensor_get(out)[0, 0].copy()
        for t, pv in zip(s_before + s_after, pat):
            assert np.array_equal(gg.tensor_get(t)[0, 0, 0], pv[0]), "a sentinel next to the MUL_MAT result was overwritten"
        return res


def run_mul_mat_bcast(qtype, w_bytes, x, m, k, bs, nr):
    """the batched / broadcast form (tests/test-backend-ops.cpp:3127-3191): a [k, m, bs0, bs1] quantized, b [k, n, bs0*nr0, bs1*nr1] f32
    -> [m, n, bs0*nr0, bs1*nr1]; w_bytes [bs1, bs0, m, row_bytes], x [bs1*nr1, bs0*nr0, n, k]"""
    L = gg.base()
    be = backend()
    n = x.shape[2]
    with gg.Context() as ctx:
        a = ctx.new_tensor(qtype, [k, m, bs[0], bs[1]], "a")
        b = ctx.new_tensor(gg.F32, [k, n, bs[0]*nr[0], bs[1]*nr[1]], "b")
        s_before = _sentinels(ctx, 1)
        out = L.ggml_mul_mat(ctx.ctx, a, b)
        s_after = _sentinels(ctx, 1)
        assert be.supports_op(out)
        assert ctx.alloc(be)
        pat = np.full((1, SENTINEL_ELEMS), 3.25, np.float32)
        for t in s_before + s_after:
            gg.tensor_set(t, pat)
        gg.tensor_set(a, w_bytes)
        gg.tensor_set(b, x)
        be.compute(gg.graph_of(ctx, out))
        res = gg.tensor_get(out).copy()
        for t in s_before + s_after:
            assert np.array_equal(gg.tensor_get(t)[0, 0, 0], pat[0]), "a sentinel next to the MUL_MAT result was overwritten"
        return res
