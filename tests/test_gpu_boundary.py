"""Boundary behaviours a real host relies on, driven through the ggml C-ABI of the backend:

  * events + tensor_set_async from the pinned host buffer type in a 4 x 1 MiB ring  — llama_model_loader::load_all_data,
    src/llama-model-loader.cpp:930-1010,1079-1090 (a second backend on the same device does the uploads, :1002);
  * the weight_buft_supported probe: supports_op on a dummy MUL_MAT / MUL_MAT_ID whose weight sits in a 0-BYTE buffer of our
    buffer type and has no data                                                     — src/llama-model.cpp:152-286 (:178-189, :278-283);
  * offload_op                                                                      — the scheduler's op_offload path (src/llama-context.cpp:281);
  * cpy_tensor_async between two backends (the layer-split hand-off)                — src/llama-context.cpp:255-279;
  * graph_compute returns GGML_STATUS_FAILED instead of taking the process down     — src/llama-context.cpp:1078-1107;
  * tensors a fusion would leave unwritten exist when somebody else can read them   — the scheduler's eval callback
    (tools/imatrix/imatrix.cpp:223-247 reads a MUL_MAT's src1) and result_norm (src/llama-context.cpp:1137-1151).
"""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
import ops_ref
from gpu_util import QTYPES, backend, gg

pytestmark = pytest.mark.gpu


def _sig_extra(L):
    L.ggml_backend_offload_op.restype = C.c_bool; L.ggml_backend_offload_op.argtypes = [C.c_void_p, gg.tensor_p]
    L.ggml_backend_get_device.restype = C.c_void_p; L.ggml_backend_get_device.argtypes = [C.c_void_p]
    return L


def test_event_ordered_async_upload_through_pinned_ring():
    """the loader's upload loop: 4 pinned staging buffers of 1 MiB, one event per buffer, a dedicated upload backend"""
    L = gg.base(); be = backend()
    up = gg.Backend(0)                                      # ggml_backend_dev_init(dev) for uploads only (:1002)
    props = gg.dev_props(); L.ggml_backend_dev_get_props(be.dev, C.byref(props))
    assert props.caps.async_ and props.caps.host_buffer and props.caps.events      # the condition at :967
    host_buft = L.ggml_backend_dev_host_buffer_type(be.dev)
    assert host_buft and L.ggml_backend_buft_is_host(host_buft)
    n_buf, buf_size = 4, 1 << 20
    bufs = [L.ggml_backend_buft_alloc_buffer(host_buft, buf_size) for _ in range(n_buf)]
    ptrs = [L.ggml_backend_buffer_get_base(b) for b in bufs]
    events = [L.ggml_backend_event_new(be.dev) for _ in range(n_buf)]
    assert all(bufs) and all(ptrs) and all(events)
    rng = np.random.default_rng(5)
    m, k = 2300, 4096                                      # 5.3 MB of Q4_K rows: six ring slots, the last one partial
    w = orc.random_blocks(rng, gg.Q4_K, (m,), k)
    raw = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
    with gg.Context() as ctx:
        a = ctx.new_tensor(gg.Q4_K, [k, m], "w")
        assert ctx.alloc(be)
        off, idx = 0, 0
        while off < raw.size:
            L.ggml_backend_event_synchronize(events[idx])   # (a fresh event is signalled: first round passes)
            n = min(buf_size, raw.size - off)
            C.memmove(ptrs[idx], raw[off:off + n].ctypes.data, n)
            L.ggml_backend_tensor_set_async(up.be, a, C.c_void_p(ptrs[idx]), off, n)
            L.ggml_backend_event_record(events[idx], up.be)
            off += n; idx = (idx + 1) % n_buf
        for e in events:
            L.ggml_backend_event_synchronize(e)
        # the compute backend waits for the upload backend's last event before it uses the weights (event_wait is the scheduler's hand-off)
        L.ggml_backend_event_record(events[0], up.be)
        L.ggml_backend_event_wait(be.be, events[0])
        got = gg.tensor_get(a).reshape(-1)
        assert np.array_equal(got, raw)
        # ... and a mat-vec on them gives the oracle's result
        x = rng.uniform(-1, 1, size=(1, k)).astype(np.float32)
        b = ctx.new_tensor(gg.F32, [k, 1]); out = L.ggml_mul_mat(ctx.ctx, a, b)
        buf2 = L.ggml_backend_alloc_ctx_tensors(ctx.ctx, be.be); assert buf2; ctx.buffers.append(buf2)
        gg.tensor_set(b, x)
        be.compute(gg.graph_of(ctx, out))
        res = gg.tensor_get(out)[0, 0]
        cpu = orc.mul_mat_2d(w, gg.Q4_K, x, "cpu")
        assert float(np.abs(res - cpu).max()) <= 2e-5*float(np.abs(cpu).max())
    for e in events:
        L.ggml_backend_event_free(e)
    for b in bufs:
        L.ggml_backend_buffer_free(b)
    up.free()


@pytest.mark.parametrize("tname", ["q4_0", "q8_0", "q4_K", "q5_K", "q6_K", "mxfp4"])
def test_weight_buft_supported_probe_on_zero_byte_buffer(tname):
    """supports_op must answer from shapes and types alone: the probe's weight lives in a 0-byte buffer and has no data"""
    L = gg.base(); be = backend()
    buft = L.ggml_backend_dev_buffer_type(be.dev)
    buf = L.ggml_backend_buft_alloc_buffer(buft, 0)         # src/llama-model.cpp:280
    assert buf, "a 0-byte buffer must be allocatable"
    k, m, n_expert, n_used = 4096, 1024, 8, 2
    with gg.Context() as ctx:
        w = ctx.new_tensor(QTYPES[tname], [k, m]); w.contents.buffer = buf
        b = ctx.new_tensor(gg.F32, [k, 512])
        op = L.ggml_mul_mat(ctx.ctx, w, b)                  # :178-182
        assert w.contents.data is None and op.contents.data is None
        assert L.ggml_backend_dev_supports_op(be.dev, op)
        w3 = ctx.new_tensor(QTYPES[tname], [k, m, n_expert]); w3.contents.buffer = buf
        b3 = ctx.new_tensor(gg.F32, [k, n_used, 512]); ids = ctx.new_tensor(gg.I32, [n_used, 512])
        op3 = L.ggml_mul_mat_id(ctx.ctx, w3, b3, ids)       # :183-189
        assert L.ggml_backend_dev_supports_op(be.dev, op3)
        # a type this backend has no kernel for is refused, not asserted on
        bad = ctx.new_tensor(10, [k, m]); bad.contents.buffer = buf      # GGML_TYPE_Q2_K
        assert not L.ggml_backend_dev_supports_op(be.dev, L.ggml_mul_mat(ctx.ctx, bad, b))
        assert L.ggml_backend_dev_supports_buft(be.dev, buft)
        assert not L.ggml_backend_dev_supports_buft(be.dev, L.ggml_backend_dev_host_buffer_type(be.dev))
    L.ggml_backend_buffer_free(buf)


def test_offload_op_threshold():
    """host-resident weights are worth uploading for batches >= 32 only; GET_ROWS never (the rule the reference's GPU backends use)"""
    L = _sig_extra(gg.base()); be = backend()
    with gg.Context() as ctx:
        w = ctx.new_tensor(gg.Q4_K, [4096, 4096])
        small = L.ggml_mul_mat(ctx.ctx, w, ctx.new_tensor(gg.F32, [4096, 8]))
        big = L.ggml_mul_mat(ctx.ctx, w, ctx.new_tensor(gg.F32, [4096, 512]))
        rows = L.ggml_get_rows(ctx.ctx, ctx.new_tensor(gg.F32, [4096, 100]), ctx.new_tensor(gg.I32, [64]))
        w3 = ctx.new_tensor(gg.Q4_K, [4096, 1024, 8])
        id_small = L.ggml_mul_mat_id(ctx.ctx, w3, ctx.new_tensor(gg.F32, [4096, 2, 4]), ctx.new_tensor(gg.I32, [2, 4]))
        id_big = L.ggml_mul_mat_id(ctx.ctx, w3, ctx.new_tensor(gg.F32, [4096, 2, 64]), ctx.new_tensor(gg.I32, [2, 64]))
        assert not L.ggml_backend_offload_op(be.be, small) and L.ggml_backend_offload_op(be.be, big)
        assert not L.ggml_backend_offload_op(be.be, rows)
        assert not L.ggml_backend_offload_op(be.be, id_small) and L.ggml_backend_offload_op(be.be, id_big)


def _copy_between(be_src, be_dst, n=4096, reps=3):
    L = gg.base()
    rng = np.random.default_rng(11)
    with gg.Context() as cs, gg.Context() as cd:
        src = cs.new_tensor(gg.F32, [n, 4]); assert cs.alloc(be_src)
        dst = cd.new_tensor(gg.F32, [n, 4]); out = L.ggml_scale(cd.ctx, dst, 2.0); assert cd.alloc(be_dst)
        g = gg.graph_of(cd, out)
        for r in range(reps):
            x = rng.standard_normal((4, n)).astype(np.float32)
            gg.tensor_set(src, x)
            L.ggml_backend_tensor_copy_async(be_src.be, be_dst.be, src, dst)     # the destination stream waits for the copy: no host sync here
            assert be_dst.compute_async(g) == gg.GGML_STATUS_SUCCESS
            be_dst.synchronize()
            assert np.array_equal(gg.tensor_get(out)[0, 0], 2.0*x)


def test_cpy_tensor_async_between_two_backends_on_one_device():
    a, b = backend(), gg.Backend(0)
    _copy_between(a, b); _copy_between(b, a)
    b.free()


def test_cpy_tensor_async_between_two_devices():
    L = gg.base()
    if L.ggml_backend_reg_dev_count(backend().reg) < 2:
        pytest.skip("one MI355X visible here: tests/test_gpu_inproc_layer_split.py runs this copy between two devices of a GGML_MI355X_VIRTUAL_DEVICES=2 registry")
    b = gg.Backend(1)
    _copy_between(backend(), b); _copy_between(b, backend())
    b.free()


def test_graph_compute_reports_failure_and_backend_stays_usable():
    """FLASH_ATTN_EXT on a q view that is not 16-byte aligned: supports_op cannot see data pointers, the kernel needs the alignment ->
    GGML_STATUS_FAILED (llama_context::decode maps it and rolls back, src/llama-context.cpp:1101-1106), never an abort"""
    L = gg.base(); be = backend()
    hd, n_head, n_kv = 128, 4, 256
    with gg.Context() as ctx:
        qbig = ctx.new_tensor(gg.F32, [hd*n_head + 4])
        q = L.ggml_view_3d(ctx.ctx, qbig, hd, 1, n_head, 4*hd*n_head, 4*hd, 4)            # starts 4 bytes into the buffer
        k = ctx.new_tensor(gg.F16, [hd, n_kv, 1]); v = ctx.new_tensor(gg.F16, [hd, n_kv, 1])
        mask = ctx.new_tensor(gg.F16, [n_kv, 32])
        out = L.ggml_flash_attn_ext(ctx.ctx, q, k, v, mask, 0.088, 0.0, 0.0)
        assert ctx.alloc(be)
        for t in (qbig, k, v, mask):
            L.ggml_backend_tensor_memset(t, 0, 0, L.ggml_nbytes(t))
        assert be.supports_op(out)
        st = L.ggml_backend_graph_compute(be.be, gg.graph_of(ctx, out))
        assert st != gg.GGML_STATUS_SUCCESS
    # the stream and the backend are intact
    rng = np.random.default_rng(3)
    w = orc.random_blocks(rng, gg.Q4_K, (64,), 512); x = rng.uniform(-1, 1, size=(1, 512)).astype(np.float32)
    from gpu_util import run_mul_mat
    got = run_mul_mat(gg.Q4_K, w, x, 64, 512)
    cpu = orc.mul_mat_2d(w, gg.Q4_K, x, "cpu")
    assert float(np.abs(got - cpu).max()) <= 2e-5*float(np.abs(cpu).max())


def _norm_mm_graph(ctx, L, k, m, name, tail):
    x = ctx.new_tensor(gg.F32, [k, 1], "x"); nw = ctx.new_tensor(gg.F32, [k], "nw"); w = ctx.new_tensor(gg.Q4_K, [k, m], "w")
    nrm = L.ggml_rms_norm(ctx.ctx, x, 1e-5)
    prod = L.ggml_mul(ctx.ctx, nrm, nw); L.ggml_set_name(prod, name.encode())
    mm = L.ggml_mul_mat(ctx.ctx, w, prod)
    out = L.ggml_scale(ctx.ctx, mm, 1.0) if tail else mm
    return x, nw, w, prod, mm, out


@pytest.mark.parametrize("name,tail,must_exist", [("attn_norm-0", False, True),      # the view ends at the mat-mul: the eval-callback case
                                                    ("result_norm", True, True),      # the embeddings tensor
                                                    ("attn_norm-0", True, False)])    # an ordinary interior tensor may stay unwritten
def test_norm_product_exists_when_someone_else_can_read_it(name, tail, must_exist):
    L = gg.base(); be = backend()
    be.set_option("fusion", 1)
    k, m = 4096, 512
    rng = np.random.default_rng(17)
    xv = rng.standard_normal((1, k)).astype(np.float32); nwv = rng.uniform(0.5, 1.5, size=(1, k)).astype(np.float32)
    wv = orc.random_blocks(rng, gg.Q4_K, (m,), k)
    with gg.Context() as ctx:
        x, nw, w, prod, mm, out = _norm_mm_graph(ctx, L, k, m, name, tail)
        assert ctx.alloc(be)
        gg.tensor_set(x, xv); gg.tensor_set(nw, nwv); gg.tensor_set(w, wv)
        L.ggml_backend_tensor_memset(prod, 0xFF, 0, L.ggml_nbytes(prod))       # NaN pattern: "never written" is visible
        be.compute(gg.graph_of(ctx, out))
        ref = ops_ref.rms_norm(xv, 1e-5)*nwv
        got_mm = gg.tensor_get(mm)[0, 0]
        cpu = orc.mul_mat_2d(wv, gg.Q4_K, ref.astype(np.float32), "cpu")
        assert orc.nmse(cpu, got_mm) <= 5e-4
        got = gg.tensor_get(prod)[0, 0]
        if must_exist:
            assert np.isfinite(got).all() and float(np.abs(got - ref).max()) <= 1e-5*float(np.abs(ref).max())
