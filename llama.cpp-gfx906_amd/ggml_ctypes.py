"""ggml_ctypes.py — ctypes binding of the ggml public API surface the reference's tests use to drive
a backend (tests/test-backend-ops.cpp:1082-1240): context + tensor construction, op constructors,
buffer allocation, tensor_set/get, graph_compute. The functions live in lib/libggml-base-compat.so
(harness stand-in for libggml-base); the backend under test is loaded the way ggml's registry loads a
dynamic backend: dlopen + `ggml_backend_init` (docs/build.md:613).

Host plumbing only — no arithmetic happens in Python. If the native libraries are missing this module
raises: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIBDIR = HERE / ("lib-" + os.environ["MI355X_BUILD_VARIANT"] if os.environ.get("MI355X_BUILD_VARIANT") else "lib")   # build.py: variant builds

GGML_MAX_DIMS, GGML_MAX_SRC, GGML_MAX_NAME, GGML_MAX_OP_PARAMS = 4, 10, 64, 64

# enum ggml_type — gguf-py/gguf/constants.py:2698-2730
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, I32, I64, BF16, MXFP4 = 0, 1, 2, 8, 12, 13, 14, 15, 26, 27, 30, 39
TYPE_NP = {F32: np.float32, F16: np.float16, I32: np.int32, I64: np.int64}
# (block size, type size) — gguf-py/gguf/constants.py:2839-2872
QUANT_SIZES = {F32: (1, 4), F16: (1, 2), BF16: (1, 2), I32: (1, 4), I64: (1, 8), Q4_0: (32, 18), Q8_0: (32, 34),
               Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210), MXFP4: (32, 17)}

GGML_STATUS_SUCCESS = 0
GGML_ROPE_TYPE_NEOX = 2
GGML_SORT_ORDER_ASC, GGML_SORT_ORDER_DESC = 0, 1
(GLU_REGLU, GLU_GEGLU, GLU_SWIGLU, GLU_SWIGLU_OAI, GLU_GEGLU_ERF, GLU_GEGLU_QUICK) = range(6)
(UNARY_ABS, UNARY_SGN, UNARY_NEG, UNARY_STEP, UNARY_TANH, UNARY_ELU, UNARY_RELU, UNARY_SIGMOID, UNARY_GELU,
 UNARY_GELU_QUICK, UNARY_SILU, UNARY_HARDSWISH, UNARY_HARDSIGMOID, UNARY_EXP, UNARY_GELU_ERF) = range(15)


class ggml_tensor(C.Structure):
    pass


ggml_tensor._fields_ = [
    ("type", C.c_int),
    ("buffer", C.c_void_p),
    ("ne", C.c_int64 * GGML_MAX_DIMS),
    ("nb", C.c_size_t * GGML_MAX_DIMS),
    ("op", C.c_int),
    ("op_params", C.c_int32 * (GGML_MAX_OP_PARAMS // 4)),
    ("flags", C.c_int32),
    ("src", C.POINTER(ggml_tensor) * GGML_MAX_SRC),
    ("view_src", C.POINTER(ggml_tensor)),
    ("view_offs", C.c_size_t),
    ("data", C.c_void_p),
    ("name", C.c_char * GGML_MAX_NAME),
    ("extra", C.c_void_p),
    ("padding", C.c_char * 8),
]
tensor_p = C.POINTER(ggml_tensor)


class ggml_init_params(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


class mi355x_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "graphs_computed", "nodes_computed", "kernels_launched", "graph_replays", "graph_captures", "mmvq_launches",
        "mmq_launches", "weight_bytes", "act_quant_launches", "act_quant_reused", "split_mul_mats")]


class mi355x_prof_entry(C.Structure):
    _fields_ = [("type", C.c_int32), ("n", C.c_int32), ("m", C.c_int64), ("k", C.c_int64), ("launches", C.c_uint64),
                ("total_ms", C.c_double), ("bytes_per_launch", C.c_uint64), ("kernel", C.c_char * 96)]


class dev_caps(C.Structure):
    _fields_ = [("async_", C.c_bool), ("host_buffer", C.c_bool), ("buffer_from_host_ptr", C.c_bool), ("events", C.c_bool)]


class dev_props(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("memory_free", C.c_size_t), ("memory_total", C.c_size_t),
                ("type", C.c_int), ("caps", dev_caps)]


_base = None
_backend_lib = None

# every symbol include/ggml-mi355x.h declares (checked by tests/test_abi.py without a GPU)
MI355X_EXPORTS = [
    "ggml_backend_init", "ggml_backend_score", "ggml_backend_mi355x_reg", "ggml_backend_mi355x_init", "ggml_backend_is_mi355x",
    "ggml_backend_mi355x_get_device_count", "ggml_backend_mi355x_get_device_description", "ggml_backend_mi355x_get_device_memory",
    "ggml_backend_mi355x_buffer_type", "ggml_backend_mi355x_host_buffer_type", "ggml_backend_mi355x_get_stream",
    "ggml_backend_mi355x_get_counters", "ggml_backend_mi355x_reset_counters", "ggml_backend_mi355x_set_option",
    "ggml_backend_mi355x_get_profile", "ggml_backend_mi355x_split_buffer_type",
    "ggml_backend_mi355x_tensor_set_from_device_async", "ggml_backend_mi355x_tensor_get_to_device_async",
]


def _sig(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


def base():
    """libggml-base-compat.so, loaded RTLD_GLOBAL so that the backend .so resolves ggml_* against it."""
    global _base
    if _base is not None:
        return _base
    path = LIBDIR / "libggml-base-compat.so"
    if not path.exists():
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no fallback exists)")
    L = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    vp, i64, sz, ci, cf = C.c_void_p, C.c_int64, C.c_size_t, C.c_int, C.c_float
    T = tensor_p
    _sig(L, "ggml_init", vp, [ggml_init_params]); _sig(L, "ggml_free", None, [vp])
    _sig(L, "ggml_new_tensor_4d", T, [vp, ci, i64, i64, i64, i64])
    _sig(L, "ggml_nbytes", sz, [T]); _sig(L, "ggml_nelements", i64, [T])
    _sig(L, "ggml_set_name", T, [T, C.c_char_p])
    for n in ("ggml_add", "ggml_mul", "ggml_div", "ggml_mul_mat", "ggml_get_rows", "ggml_cpy", "ggml_swiglu_split"):
        _sig(L, n, T, [vp, T, T])
    for n in ("ggml_add_id", "ggml_mul_mat_id", "ggml_set_rows"):
        _sig(L, n, T, [vp, T, T, T])
    for n in ("ggml_sum_rows", "ggml_cont", "ggml_transpose", "ggml_soft_max", "ggml_silu", "ggml_sigmoid"):
        _sig(L, n, T, [vp, T])
    _sig(L, "ggml_scale", T, [vp, T, cf]); _sig(L, "ggml_scale_bias", T, [vp, T, cf, cf])
    _sig(L, "ggml_rms_norm", T, [vp, T, cf])
    _sig(L, "ggml_mul_mat_set_prec", None, [T, ci])
    _sig(L, "ggml_cont_2d", T, [vp, T, i64, i64])
    _sig(L, "ggml_reshape_2d", T, [vp, T, i64, i64]); _sig(L, "ggml_reshape_3d", T, [vp, T, i64, i64, i64])
    _sig(L, "ggml_reshape_4d", T, [vp, T, i64, i64, i64, i64])
    _sig(L, "ggml_view_1d", T, [vp, T, i64, sz]); _sig(L, "ggml_view_2d", T, [vp, T, i64, i64, sz, sz])
    _sig(L, "ggml_view_3d", T, [vp, T, i64, i64, i64, sz, sz, sz])
    _sig(L, "ggml_view_4d", T, [vp, T, i64, i64, i64, i64, sz, sz, sz, sz])
    _sig(L, "ggml_permute", T, [vp, T, ci, ci, ci, ci])
    _sig(L, "ggml_soft_max_ext", T, [vp, T, T, cf, cf]); _sig(L, "ggml_soft_max_add_sinks", None, [T, T])
    _sig(L, "ggml_flash_attn_ext", T, [vp, T, T, T, T, cf, cf, cf]); _sig(L, "ggml_flash_attn_ext_add_sinks", None, [T, T])
    _sig(L, "ggml_flash_attn_ext_set_prec", None, [T, C.c_int]); _sig(L, "ggml_cast", T, [vp, T, C.c_int])
    _sig(L, "ggml_rope_ext", T, [vp, T, T, T, ci, ci, ci, cf, cf, cf, cf, cf, cf])
    _sig(L, "ggml_argsort", T, [vp, T, ci]); _sig(L, "ggml_top_k", T, [vp, T, ci])
    _sig(L, "ggml_swiglu_oai", T, [vp, T, T, cf, cf]); _sig(L, "ggml_glu_split", T, [vp, T, T, ci])
    _sig(L, "ggml_unary", T, [vp, T, ci])
    _sig(L, "ggml_new_graph_custom", vp, [vp, sz, C.c_bool]); _sig(L, "ggml_build_forward_expand", None, [vp, T])
    _sig(L, "ggml_graph_n_nodes", ci, [vp]); _sig(L, "ggml_graph_node", T, [vp, ci])
    # backend API
    _sig(L, "ggml_backend_load", vp, [C.c_char_p])
    _sig(L, "ggml_backend_reg_name", C.c_char_p, [vp]); _sig(L, "ggml_backend_reg_dev_count", sz, [vp])
    _sig(L, "ggml_backend_reg_dev_get", vp, [vp, sz]); _sig(L, "ggml_backend_reg_get_proc_address", vp, [vp, C.c_char_p])
    _sig(L, "ggml_backend_dev_name", C.c_char_p, [vp]); _sig(L, "ggml_backend_dev_description", C.c_char_p, [vp])
    _sig(L, "ggml_backend_dev_memory", None, [vp, C.POINTER(sz), C.POINTER(sz)])
    _sig(L, "ggml_backend_dev_type", ci, [vp]); _sig(L, "ggml_backend_dev_get_props", None, [vp, C.POINTER(dev_props)])
    _sig(L, "ggml_backend_dev_init", vp, [vp, C.c_char_p]); _sig(L, "ggml_backend_dev_buffer_type", vp, [vp])
    _sig(L, "ggml_backend_dev_host_buffer_type", vp, [vp])
    _sig(L, "ggml_backend_dev_supports_op", C.c_bool, [vp, T]); _sig(L, "ggml_backend_dev_supports_buft", C.c_bool, [vp, vp])
    _sig(L, "ggml_backend_supports_op", C.c_bool, [vp, T])
    _sig(L, "ggml_backend_name", C.c_char_p, [vp]); _sig(L, "ggml_backend_free", None, [vp])
    _sig(L, "ggml_backend_synchronize", None, [vp])
    _sig(L, "ggml_backend_graph_compute", ci, [vp, vp]); _sig(L, "ggml_backend_graph_compute_async", ci, [vp, vp])
    _sig(L, "ggml_backend_alloc_ctx_tensors", vp, [vp, vp]); _sig(L, "ggml_backend_alloc_ctx_tensors_from_buft", vp, [vp, vp])
    _sig(L, "ggml_backend_buft_name", C.c_char_p, [vp]); _sig(L, "ggml_backend_buft_alloc_buffer", vp, [vp, sz])
    _sig(L, "ggml_backend_buft_get_alignment", sz, [vp]); _sig(L, "ggml_backend_buft_get_alloc_size", sz, [vp, T])
    _sig(L, "ggml_backend_buft_is_host", C.c_bool, [vp])
    _sig(L, "ggml_backend_buffer_free", None, [vp]); _sig(L, "ggml_backend_buffer_get_size", sz, [vp])
    _sig(L, "ggml_backend_buffer_get_base", vp, [vp]); _sig(L, "ggml_backend_buffer_clear", None, [vp, C.c_uint8])
    _sig(L, "ggml_backend_buffer_set_usage", None, [vp, ci]); _sig(L, "ggml_backend_buffer_is_host", C.c_bool, [vp])
    _sig(L, "ggml_backend_tensor_set", None, [T, vp, sz, sz]); _sig(L, "ggml_backend_tensor_get", None, [T, vp, sz, sz])
    _sig(L, "ggml_backend_tensor_memset", None, [T, C.c_uint8, sz, sz])
    _sig(L, "ggml_backend_tensor_set_async", None, [vp, T, vp, sz, sz]); _sig(L, "ggml_backend_tensor_get_async", None, [vp, T, vp, sz, sz])
    _sig(L, "ggml_backend_tensor_copy", None, [T, T]); _sig(L, "ggml_backend_tensor_copy_async", None, [vp, vp, T, T])
    _sig(L, "ggml_backend_event_new", vp, [vp]); _sig(L, "ggml_backend_event_free", None, [vp])
    _sig(L, "ggml_backend_event_record", None, [vp, vp]); _sig(L, "ggml_backend_event_synchronize", None, [vp])
    _sig(L, "ggml_backend_event_wait", None, [vp, vp])
    _sig(L, "ggml_fp16_to_fp32", cf, [C.c_uint16]); _sig(L, "ggml_fp32_to_fp16", C.c_uint16, [cf])
    _base = L
    return L


def backend_lib_path() -> Path:
    return LIBDIR / "libggml-mi355x.so"


def backend_cdll():
    """dlopen the product .so (no GPU needed: used by the ABI export test)."""
    global _backend_lib
    if _backend_lib is None:
        base()
        p = backend_lib_path()
        if not p.exists():
            raise RuntimeError(f"{p} is missing: the HIP backend is not built (no fallback exists)")
        _backend_lib = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
    return _backend_lib


_reg = None


def load_backend():
    """ggml_backend_load(): dlopen + ggml_backend_score + ggml_backend_init. Raises without a gfx950 device."""
    global _reg
    if _reg is None:
        L = base()
        backend_cdll()
        reg = L.ggml_backend_load(str(backend_lib_path()).encode())
        if not reg:
            raise RuntimeError("ggml_backend_load(libggml-mi355x.so) failed: no gfx950 device visible or ABI mismatch")
        _reg = reg
    return _reg


# ---------------------------------------------------------------------------------------------
# thin object layer used by tests / bench
# ---------------------------------------------------------------------------------------------
class Context:
    """ggml_init(no_alloc=true) context; tensors get their memory from a backend buffer (test-backend-ops.cpp:1134)."""

    def __init__(self):
        self.L = base()
        self.ctx = self.L.ggml_init(ggml_init_params(0, None, True))
        self.buffers = []

    def new_tensor(self, type_, ne, name=None):
        ne = list(ne) + [1] * (4 - len(ne))
        t = self.L.ggml_new_tensor_4d(self.ctx, type_, *ne)
        if name:
            self.L.ggml_set_name(t, name.encode())
        return t

    def alloc(self, backend):
        buf = self.L.ggml_backend_alloc_ctx_tensors(self.ctx, backend.be)
        if buf:
            self.buffers.append(buf)
        return buf

    def new_graph(self, size=8192):
        return self.L.ggml_new_graph_custom(self.ctx, size, False)

    def free(self):
        for b in self.buffers:
            self.L.ggml_backend_buffer_free(b)
        self.buffers = []
        if self.ctx:
            self.L.ggml_free(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()


class Backend:
    """one ggml_backend_t (= one HIP stream) on device `index` of the MI355X registry."""

    def __init__(self, index=0):
        self.L = base()
        self.reg = load_backend()
        n = self.L.ggml_backend_reg_dev_count(self.reg)
        if index >= n:
            raise RuntimeError(f"MI355X device {index} requested, registry has {n}")
        self.dev = self.L.ggml_backend_reg_dev_get(self.reg, index)
        self.be = self.L.ggml_backend_dev_init(self.dev, None)
        if not self.be:
            raise RuntimeError("ggml_backend_dev_init failed")
        self._lib = backend_cdll()
        self._lib.ggml_backend_mi355x_get_stream.restype = C.c_void_p
        self._lib.ggml_backend_mi355x_get_stream.argtypes = [C.c_void_p]
        self._lib.ggml_backend_mi355x_get_counters.argtypes = [C.c_void_p, C.POINTER(mi355x_counters)]
        self._lib.ggml_backend_mi355x_reset_counters.argtypes = [C.c_void_p]
        self._lib.ggml_backend_mi355x_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._lib.ggml_backend_mi355x_get_profile.restype = C.c_int
        self._lib.ggml_backend_mi355x_get_profile.argtypes = [C.c_void_p, C.POINTER(mi355x_prof_entry), C.c_int]

    def name(self):
        return self.L.ggml_backend_name(self.be).decode()

    def stream(self):
        return self._lib.ggml_backend_mi355x_get_stream(self.be)

    def counters(self):
        c = mi355x_counters()
        self._lib.ggml_backend_mi355x_get_counters(self.be, C.byref(c))
        return {n: getattr(c, n) for n, _ in mi355x_counters._fields_}

    def reset_counters(self):
        self._lib.ggml_backend_mi355x_reset_counters(self.be)

    def set_option(self, key, value):
        return self._lib.ggml_backend_mi355x_set_option(self.be, key.encode(), int(value))

    def profile(self, cap=64):
        """aggregated (type, m, k, n) -> launches / total_ms of the mat-mul launches recorded under option 'profile'"""
        arr = (mi355x_prof_entry * cap)()
        n = self._lib.ggml_backend_mi355x_get_profile(self.be, arr, cap)
        return [dict(type=e.type, n=e.n, m=e.m, k=e.k, launches=e.launches, total_ms=e.total_ms, bytes_per_launch=e.bytes_per_launch,
                     kernel=e.kernel.decode())
                for e in arr[:n]]

    def supports_op(self, t):
        return bool(self.L.ggml_backend_supports_op(self.be, t))

    def compute(self, graph):
        st = self.L.ggml_backend_graph_compute(self.be, graph)
        if st != GGML_STATUS_SUCCESS:
            raise RuntimeError(f"graph_compute returned ggml_status {st}")

    def compute_async(self, graph):
        return self.L.ggml_backend_graph_compute_async(self.be, graph)

    def synchronize(self):
        self.L.ggml_backend_synchronize(self.be)

    def free(self):
        if self.be:
            self.L.ggml_backend_free(self.be)
            self.be = None


def tensor_set(t, arr: np.ndarray):
    arr = np.ascontiguousarray(arr)
    L = base()
    nbytes = L.ggml_nbytes(t)
    assert arr.nbytes == nbytes, f"tensor_set: array has {arr.nbytes} bytes, tensor needs {nbytes}"
    L.ggml_backend_tensor_set(t, arr.ctypes.data_as(C.c_void_p), 0, nbytes)


def tensor_get(t) -> np.ndarray:
    """contiguous tensors only: returns an array shaped [ne3, ne2, ne1, ne0] (or raw bytes rows for quantized types)"""
    L = base()
    tt = t.contents
    nbytes = L.ggml_nbytes(t)
    raw = np.empty(nbytes, dtype=np.uint8)
    L.ggml_backend_tensor_get(t, raw.ctypes.data_as(C.c_void_p), 0, nbytes)
    shape = (tt.ne[3], tt.ne[2], tt.ne[1])
    if tt.type in TYPE_NP:
        return raw.view(TYPE_NP[tt.type]).reshape(shape + (tt.ne[0],))
    bs, ts = QUANT_SIZES[tt.type]
    return raw.reshape(shape + (tt.ne[0] // bs * ts,))


def graph_of(ctx: Context, *outs, size=8192):
    g = ctx.new_graph(size)
    for o in outs:
        ctx.L.ggml_build_forward_expand(g, o)
    return g
