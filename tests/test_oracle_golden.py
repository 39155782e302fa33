"""Pin the oracle (oracle/ggml_oracle.c) against the golden vectors generated from the
reference's Python block-format definition (tests/golden/make_golden.py).

Bar: bit-exact for (de)quantization (the reference's own test demands np.array_equal,
gguf-py/tests/test_quants.py:116-119); MUL_MAT "exact" mode equals the float64 product
of the reference dequantization to f32 rounding; the CPU-style integer path stays inside
the reference's op gate NMSE <= 5e-4 (tests/test-backend-ops.cpp:3106-3108).
"""
import numpy as np
import pytest

import oracle as orc

NAMES = {"q4_0": orc.Q4_0, "q8_0": orc.Q8_0, "q4_K": orc.Q4_K, "q5_K": orc.Q5_K, "q6_K": orc.Q6_K, "mxfp4": orc.MXFP4}


@pytest.mark.parametrize("name", list(NAMES))
def test_dequantize_bit_exact(name, golden_dir):
    g = np.load(golden_dir / f"dequant_{name}.npz")
    got = orc.dequantize(g["blocks"], NAMES[name])
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), g["expected"].view(np.uint32)), name


@pytest.mark.parametrize("name", ["q4_0", "q8_0", "mxfp4"])
def test_quantize_bit_exact(name, golden_dir):
    g = np.load(golden_dir / f"quant_{name}.npz")
    got = orc.quantize(g["x"], NAMES[name])
    assert np.array_equal(got, g["expected"]), name


@pytest.mark.parametrize("k", [256, 1024])
@pytest.mark.parametrize("name", list(NAMES))
def test_mul_mat_exact_matches_reference_dequant(name, k, golden_dir):
    g = np.load(golden_dir / f"mulmat_{name}_k{k}.npz")
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16):
        got = orc.mul_mat_2d(g["w"], NAMES[name], g["x"][:n], "exact")
        exp = g["expected"][:n]
        np.testing.assert_allclose(got, exp.astype(np.float32), rtol=0, atol=float(np.abs(exp).max()) * 2e-7)


@pytest.mark.parametrize("k", [256, 1024])
@pytest.mark.parametrize("name", list(NAMES))
def test_mul_mat_cpu_style_within_reference_gate(name, k, golden_dir):
    g = np.load(golden_dir / f"mulmat_{name}_k{k}.npz")
    got = orc.mul_mat_2d(g["w"], NAMES[name], g["x"], "cpu")
    assert orc.nmse(g["expected"], got) < 5e-4


@pytest.mark.parametrize("name", list(NAMES))
def test_mul_mat_id_matches_reference(name, golden_dir):
    g = np.load(golden_dir / f"mulmatid_{name}.npz")
    got = orc.mul_mat_id(g["w"], NAMES[name], g["b"], g["ids"], "exact")
    exp = g["expected"]
    np.testing.assert_allclose(got, exp.astype(np.float32), rtol=0, atol=float(np.abs(exp).max()) * 2e-7)
    got_cpu = orc.mul_mat_id(g["w"], NAMES[name], g["b"], g["ids"], "cpu")
    assert orc.nmse(exp, got_cpu) < 5e-4


@pytest.mark.parametrize("name", ["q4_0", "q8_0", "mxfp4"])
def test_vec_dot_error_gate(name):
    """tests/test-quantize-fns.cpp:82-99: |vec_dot(from_float(a), q8(b)) - a.b| / n <= 0.02 on 0.1+2cos(i+off)."""
    n = 32 * 128
    i = np.arange(n, dtype=np.float32)
    a = (0.1 + 2 * np.cos(i + np.float32(0.0))).astype(np.float32)
    b = (0.1 + 2 * np.cos(i + np.float32(1.0))).astype(np.float32)
    qa = orc.quantize(a[None, :], NAMES[name])[0]
    qb = orc.quantize(b[None, :], orc.vec_dot_type(NAMES[name]))[0]
    got = orc.vec_dot(NAMES[name], qa, qb, n)
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    limit = 0.02
    assert abs(got - ref) / n <= limit


def test_q8_K_round_trip_and_bsums():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 512)).astype(np.float32)
    x[0, :256] = 0
    q = orc.quantize(x, orc.Q8_K).reshape(4, 2, 292)
    d = q[:, :, 0:4].copy().view(np.float32).reshape(4, 2)
    qs = q[:, :, 4:260].view(np.int8).astype(np.int32)
    bs = q[:, :, 260:292].copy().view(np.int16).reshape(4, 2, 16).astype(np.int32)
    assert d[0, 0] == 0 and not qs[0, 0].any()
    assert np.array_equal(bs, qs.reshape(4, 2, 16, 16).sum(-1))
    rec = (qs * d[:, :, None]).reshape(4, 512)
    amax = np.abs(x.reshape(4, 2, 256)).max(-1, keepdims=True)
    assert np.all(np.abs(rec.reshape(4, 2, 256) - x.reshape(4, 2, 256)) <= amax / 127 * 0.5001 + 1e-12)


@pytest.mark.parametrize("tname", ["q4_K", "q6_K"])
def test_avx2_dot_is_bit_identical_to_the_scalar_restatement(tname):
    """oracle/ggml_oracle.c carries AVX2 forms of the two dots bench.py's cpu_baseline spends its time in; they keep the scalar code's eight integer and
    eight float lanes, so every result must be the same bits (random valid blocks, edge blocks with all-ones / all-zero bytes, several row lengths)"""
    import numpy as np
    import oracle as orc
    qt = {"q4_K": orc.Q4_K, "q6_K": orc.Q6_K}[tname]
    rng = np.random.default_rng(11)
    for k in (256, 1024, 4096):
        w = orc.random_blocks(rng, qt, (64,), k, scale=1.0/np.sqrt(k))
        w[0, :] = 0xFF; w[1, :] = 0x00; w[2, :] = 0xAA
        w[0, -2:] = np.array([1.0], np.float16).view(np.uint8); w[2, -2:] = np.array([0.5], np.float16).view(np.uint8)      # (Q6_K: d is last; keep it finite)
        if tname == "q4_K":
            blocks = w.reshape(64, -1, 144); blocks[0:3, :, 0:4] = np.array([1.0, 0.5], np.float16).view(np.uint8)
        x = (rng.standard_normal((5, k))*rng.uniform(0.01, 30.0)).astype(np.float32)
        x[4, :] = 0
        try:
            orc.set_simd(False); ref = orc.mul_mat_2d(w, qt, x, "cpu")
            orc.set_simd(True);  got = orc.mul_mat_2d(w, qt, x, "cpu")
        finally:
            orc.set_simd(True)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (tname, k)


def test_streams_oracle_equals_single_stream_oracle():
    """oracle/ref_llama.py: RefLlamaStreams (S independent streams in lockstep, used by the perplexity statistics) is the SAME arithmetic as S
    instances of RefLlama — bit for bit, in every mode — on a small random model."""
    import ref_llama
    rng = np.random.default_rng(5)
    cfg = dict(n_embd=256, n_ff=512, n_layer=2, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=96, rope_freq_base=10000.0, n_ctx_orig=256)
    ne, nff, hd, nh, nkv, nv = cfg["n_embd"], cfg["n_ff"], cfg["n_embd_head"], cfg["n_head"], cfg["n_head_kv"], cfg["n_vocab"]
    rb = lambda qt, m, k: orc.random_blocks(rng, qt, (m,), k, scale=1.0/np.sqrt(k))
    W = {}
    for il in range(cfg["n_layer"]):
        W[(il, "attn_norm")] = (0, rng.uniform(0.5, 1.5, ne).astype(np.float32)); W[(il, "ffn_norm")] = (0, rng.uniform(0.5, 1.5, ne).astype(np.float32))
        W[(il, "attn_q")] = (orc.Q4_K, rb(orc.Q4_K, nh*hd, ne)); W[(il, "attn_k")] = (orc.Q4_K, rb(orc.Q4_K, nkv*hd, ne)); W[(il, "attn_v")] = (orc.Q6_K, rb(orc.Q6_K, nkv*hd, ne))
        W[(il, "attn_output")] = (orc.Q4_K, rb(orc.Q4_K, ne, nh*hd)); W[(il, "ffn_gate")] = (orc.Q4_K, rb(orc.Q4_K, nff, ne))
        W[(il, "ffn_up")] = (orc.Q4_K, rb(orc.Q4_K, nff, ne)); W[(il, "ffn_down")] = (orc.Q6_K, rb(orc.Q6_K, ne, nff))
    W["output_norm"] = (0, np.ones(ne, np.float32)); W["output"] = (orc.Q6_K, rb(orc.Q6_K, nv, ne))
    S, T = 3, 6
    for mode in ("cpu16", "cpu", "exact"):
        st = ref_llama.RefLlamaStreams(cfg, W, S, 16, mode)
        singles = [ref_llama.RefLlama(cfg, W, 16, mode) for _ in range(S)]
        for _ in range(T):
            emb = rng.standard_normal((S, ne)).astype(np.float32)
            a = st.decode(emb)
            b = np.stack([singles[i].decode(emb[i:i + 1]) for i in range(S)])
            assert np.array_equal(a, b), mode


def test_flash_attention_cache_oracle_is_consistent():
    """oracle/ref_llama.py, the -fa 1 branch with typed caches (used by the GPU tests of FLASH_ATTN_EXT with -ctk / -ctv): with F16 caches it is the plain f16-cache
    oracle bit for bit; a Q8_0 / Q4_0 / BF16 cache changes the logits by no more than its row quantization can (and does change them)."""
    import ref_llama
    rng = np.random.default_rng(11)
    cfg = dict(n_embd=256, n_ff=512, n_layer=2, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=64, rope_freq_base=10000.0, n_ctx_orig=256)
    ne, nff, hd, nh, nkv, nv = cfg["n_embd"], cfg["n_ff"], cfg["n_embd_head"], cfg["n_head"], cfg["n_head_kv"], cfg["n_vocab"]
    rb = lambda qt, m, k: orc.random_blocks(rng, qt, (m,), k, scale=1.0/np.sqrt(k))
    W = {}
    for il in range(cfg["n_layer"]):
        W[(il, "attn_norm")] = (0, rng.uniform(0.5, 1.5, ne).astype(np.float32)); W[(il, "ffn_norm")] = (0, rng.uniform(0.5, 1.5, ne).astype(np.float32))
        W[(il, "attn_q")] = (orc.Q4_K, rb(orc.Q4_K, nh*hd, ne)); W[(il, "attn_k")] = (orc.Q4_K, rb(orc.Q4_K, nkv*hd, ne)); W[(il, "attn_v")] = (orc.Q6_K, rb(orc.Q6_K, nkv*hd, ne))
        W[(il, "attn_output")] = (orc.Q4_K, rb(orc.Q4_K, ne, nh*hd)); W[(il, "ffn_gate")] = (orc.Q4_K, rb(orc.Q4_K, nff, ne))
        W[(il, "ffn_up")] = (orc.Q4_K, rb(orc.Q4_K, nff, ne)); W[(il, "ffn_down")] = (orc.Q6_K, rb(orc.Q6_K, ne, nff))
    W["output_norm"] = (0, np.ones(ne, np.float32)); W["output"] = (orc.Q6_K, rb(orc.Q6_K, nv, ne))
    embs = [rng.standard_normal((n, ne)).astype(np.float32) for n in (5, 1, 1, 3, 1)]
    def run(**kw):
        m = ref_llama.RefLlama(dict(cfg, **kw), W, 32, "cpu16")
        return [m.decode(e) for e in embs]
    base = run()
    fa16 = run(flash_attn=1, type_k=0, type_v=0)
    for a, b in zip(base, fa16):
        assert np.array_equal(a, b)
    for tk, tv, lim in ((orc.Q8_0, orc.Q8_0, 2e-4), (orc.Q4_0, orc.Q4_0, 2e-2), (orc.BF16, orc.BF16, 2e-4), (orc.Q8_0, 0, 2e-4)):
        got = run(flash_attn=1, type_k=tk, type_v=tv)
        errs = [orc.nmse(a, b) for a, b in zip(base, got)]
        assert 0.0 < max(errs) <= lim, (tk, tv, errs)
