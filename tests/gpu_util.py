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


def run_mul_mat(qtype, w_bytes, x, m, k):
    """w_bytes [m, row_bytes] u8, x f32 [n, k] -> f32 [n, m] via GGML_OP_MUL_MAT on the device"""
    L = gg.base()
    be = backend()
    n = x.shape[0]
    with gg.Context() as ctx:
        a = ctx.new_tensor(qtype, [k, m], "a")
        b = ctx.new_tensor(gg.F32, [k, n], "b")
        out = L.ggml_mul_mat(ctx.ctx, a, b)
        assert be.supports_op(out)
        assert ctx.alloc(be)
        gg.tensor_set(a, w_bytes)
        gg.tensor_set(b, x)
        be.compute(gg.graph_of(ctx, out))
        return gg.tensor_get(out)[0, 0].copy()
