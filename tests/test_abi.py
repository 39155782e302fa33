"""No-GPU checks of the drop-in boundary: the product library loads, exports every symbol include/ggml-mi355x.h
declares, reports "not for this system" without a gfx950 device, and the restated ggml struct layouts have the sizes
and offsets the ABI depends on. No compute call is made."""
import ctypes as C
import re
from pathlib import Path

import pytest

import graft_pkg

ROOT = Path(__file__).resolve().parent.parent
pkg = graft_pkg.load()
gg = pkg.ggml


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "ggml-mi355x.h").read_text()
    declared = set(re.findall(r"GGML_BACKEND_API\s+[\w\s\*]+?\b(ggml_backend_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(gg.MI355X_EXPORTS), declared ^ set(gg.MI355X_EXPORTS)
    lib = gg.backend_cdll()
    for s in declared:
        assert getattr(lib, s) is not None


def test_score_is_zero_without_gfx950_and_load_refuses():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = gg.backend_cdll()
    lib.ggml_backend_score.restype = C.c_int
    assert lib.ggml_backend_score() == 0
    with pytest.raises(RuntimeError):
        gg.load_backend()          # the product fails loudly; there is no CPU fallback
    gg._reg = None


def test_tensor_struct_layout():
    T = gg.ggml_tensor
    assert C.sizeof(T) == 336          # 4+4(pad)+8+32+32+4+64+4+80+8+8+8+64+8+8 (ggml_tensor_overhead = 336 + 32-byte object header)
    assert T.buffer.offset == 8 and T.ne.offset == 16 and T.nb.offset == 48 and T.op.offset == 80
    assert T.op_params.offset == 84 and T.flags.offset == 148 and T.src.offset == 152
    assert T.view_src.offset == 232 and T.view_offs.offset == 240 and T.data.offset == 248 and T.name.offset == 256


def test_type_traits_match_gguf_constants():
    L = gg.base()
    L.ggml_blck_size.restype = C.c_int64; L.ggml_blck_size.argtypes = [C.c_int]
    L.ggml_type_size.restype = C.c_size_t; L.ggml_type_size.argtypes = [C.c_int]
    for t, (bs, ts) in gg.QUANT_SIZES.items():      # gguf-py/gguf/constants.py:2839-2872
        assert L.ggml_blck_size(t) == bs and L.ggml_type_size(t) == ts


def test_graph_construction_and_shapes_on_host():
    """op constructors follow the shape rules the reference's tests rely on (no backend involved)"""
    L = gg.base()
    with gg.Context() as ctx:
        a = ctx.new_tensor(gg.Q4_K, (256, 16)); b = ctx.new_tensor(gg.F32, (256, 3))
        o = L.ggml_mul_mat(ctx.ctx, a, b)                       # tests/test-backend-ops.cpp:3128: (k,m) x (k,n) -> (m,n)
        assert list(o.contents.ne) == [16, 3, 1, 1] and o.contents.type == gg.F32
        as_ = ctx.new_tensor(gg.Q4_K, (256, 32, 8)); ids = ctx.new_tensor(gg.I32, (2, 5)); bb = ctx.new_tensor(gg.F32, (256, 2, 5))
        o2 = L.ggml_mul_mat_id(ctx.ctx, as_, bb, ids)           # :3226-3245 -> (m, n_used, n_tokens)
        assert list(o2.contents.ne) == [32, 2, 5, 1]
        q = ctx.new_tensor(gg.F32, (128, 32, 7))
        p = L.ggml_permute(ctx.ctx, q, 0, 2, 1, 3)
        assert list(p.contents.ne) == [128, 7, 32, 1] and list(p.contents.nb)[:3] == [4, 128 * 32 * 4, 128 * 4]
        g = gg.graph_of(ctx, o, o2)
        assert L.ggml_graph_n_nodes(g) == 2


def test_fp16_conversion_round_trip():
    import numpy as np
    L = gg.base()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * (10.0 ** rng.integers(-8, 5, 2000)).astype(np.float32),
                           np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5, -6.0e-5], np.float32)])
    with np.errstate(over="ignore"):
        for v in vals:
            h = L.ggml_fp32_to_fp16(float(v))
            assert h == int(np.float32(v).astype(np.float16).view(np.uint16)), v
            back = L.ggml_fp16_to_fp32(h)
            assert back == float(np.uint16(h).view(np.float16).astype(np.float32))
