"""A model loaded from a GGUF file (the reference: llama_model_loader, src/llama-model-loader.cpp:919-1150): the file is the golden
one the REFERENCE's writer produced (tests/golden/make_gguf_fixture.py). Checked: the hyper-parameters read from the metadata, that every
device tensor holds exactly the file's bytes (through the pinned upload ring), the input layer's rows, the logits against the oracle
evaluated on weights taken from the FILE by the oracle's own reader, the loader's error cases, tied embeddings, and a 2-way layer split."""
import numpy as np
import pytest

import gguf_ref
import oracle as orc
from gpu_util import backend, gg, pkg
from ref_llama import RefLlama

pytestmark = pytest.mark.gpu
ls = pkg.llama_synth
NAME = "tiny_llama_q4_k_m.gguf"


def run_against_oracle(m, W, kv=64):
    rc = RefLlama(m.cfg, W, kv, "cpu16"); re_ = RefLlama(m.cfg, W, kv, "exact")
    emb_rows = orc.dequantize(W["token_embd"][1], W["token_embd"][0])
    for toks in [[5, 9, 200, 17, 3, 44, 101], [7], [8], [255], [0], [11], list(range(20, 32))]:
        for t in toks:
            assert np.array_equal(m.embedding(t), emb_rows[t]), t             # the host-side GET_ROWS of the input layer
        emb = np.stack([emb_rows[t] for t in toks]).astype(np.float32)
        got = m.decode(toks)
        assert np.isfinite(got).all()
        if len(toks) <= 8:
            assert orc.nmse(rc.decode(emb), got) <= 1e-3, toks
        else:
            rc.decode(emb)                                                     # the prefill kernels are held to the exact oracle only
        assert orc.nmse(re_.decode(emb), got) <= 2e-3, toks


def test_golden_file_loads_and_matches_the_oracle(golden_dir):
    be = backend()
    be.set_option("graphs", 1); be.set_option("fusion", 1)
    g = gguf_ref.read(golden_dir / NAME)
    m = ls.GgufLlama(be, golden_dir / NAME, n_ctx=64)
    try:
        hp = m.hp
        assert (hp.n_embd, hp.n_ff, hp.n_layer, hp.n_head, hp.n_head_kv, hp.n_embd_head, hp.n_vocab) == (256, 256, 2, 4, 2, 64, 256)
        assert (hp.ftype, hp.arch, hp.rope_type, hp.n_ctx_orig, hp.has_rope_freqs, hp.n_expert) == (15, 0, 0, 256, 0, 0)
        assert hp.rope_freq_base == 10000.0 and hp.rope_freq_scale == 1.0 and hp.f_norm_rms_eps == np.float32(1e-5)
        for t in g["tensors"]:
            if t["name"] == "token_embd.weight":
                continue                                                        # stays on the host
            dt = m.tensor(t["name"])
            assert dt.contents.type == t["type"], t["name"]
            assert np.array_equal(gg.tensor_get(dt).reshape(-1).view(np.uint8), np.asarray(t["data"])), t["name"]
        run_against_oracle(m, gguf_ref.llama_weights(g, 2))
    finally:
        m.free()


def test_tied_embeddings_and_other_types(golden_dir, tmp_path):
    """no output.weight: the output projection is token_embd (TENSOR_DUPLICATED); and a file with Q8_0 / Q5_K / Q4_0 tensors and a Q8_0 input layer"""
    be = backend()
    g = gguf_ref.read(golden_dir / NAME)
    rng = np.random.default_rng(5)
    by = {t["name"]: t for t in g["tensors"]}
    for name, qt in (("token_embd.weight", 8), ("blk.0.attn_q.weight", 8), ("blk.0.ffn_up.weight", 13), ("blk.1.attn_output.weight", 2), ("blk.1.ffn_down.weight", 13)):
        t = by[name]
        t["type"] = qt
        t["data"] = orc.random_blocks(rng, qt, (t["ne"][1],), t["ne"][0], scale=1.0/np.sqrt(t["ne"][0])).reshape(-1)
    g["tensors"] = [t for t in g["tensors"] if t["name"] != "output.weight"]
    gguf_ref.write(tmp_path / "tied.gguf", g)
    g2 = gguf_ref.read(tmp_path / "tied.gguf")
    m = ls.GgufLlama(be, tmp_path / "tied.gguf", n_ctx=64)
    try:
        assert m.tensor("output.weight").contents.type == 8
        run_against_oracle(m, gguf_ref.llama_weights(g2, 2))
    finally:
        m.free()


def test_loader_errors(golden_dir, tmp_path):
    be = backend()
    g = gguf_ref.read(golden_dir / NAME)

    def variant(fn):
        v = {"version": 3, "alignment": 32, "kv": [dict(e) for e in g["kv"]], "tensors": [dict(t) for t in g["tensors"]]}
        fn(v)
        gguf_ref.write(tmp_path / "v.gguf", v)
        return tmp_path / "v.gguf"

    def drop(name):
        def f(v): v["tensors"] = [t for t in v["tensors"] if t["name"] != name]
        return f

    def reshape(v):
        t = next(t for t in v["tensors"] if t["name"] == "blk.1.ffn_gate.weight"); t["ne"] = [256, 128]; t["data"] = t["data"][:len(t["data"])//2]

    def set_kv(key, val):
        def f(v): next(e for e in v["kv"] if e["key"] == key)["value"] = val
        return f

    def q2k(v):
        t = next(t for t in v["tensors"] if t["name"] == "blk.0.attn_k.weight"); t["type"] = 10; t["data"] = np.zeros(128*84, np.uint8)

    cases = [(drop("blk.1.ffn_down.weight"), "missing tensor 'blk.1.ffn_down.weight'"), (drop("token_embd.weight"), "missing tensor 'token_embd.weight'"),
             (reshape, "has wrong shape"), (set_kv("general.architecture", "mamba"), "unknown model architecture"),
             (drop("no such"), None), (set_kv("llama.attention.head_count", 3), "invalid head"), (q2k, "does not run"),
             (lambda v: v["kv"].__delitem__(next(i for i, e in enumerate(v["kv"]) if e["key"] == "llama.block_count")), "key not found: llama.block_count")]
    for fn, msg in cases:
        p = variant(fn)
        if msg is None:
            ls.GgufLlama(be, p, n_ctx=32).free()
            continue
        with pytest.raises(RuntimeError, match="error loading model: .*" + msg):
            ls.GgufLlama(be, p, n_ctx=32)
    with pytest.raises(RuntimeError, match="cannot open"):
        ls.GgufLlama(be, tmp_path / "absent.gguf", n_ctx=32)
    m = ls.GgufLlama(be, golden_dir / NAME, n_ctx=32)
    try:
        with pytest.raises(RuntimeError):
            m.decode([256])                                                    # a token id outside the vocabulary
        assert np.isfinite(m.decode([255])).all()
    finally:
        m.free()


def test_layer_split_of_a_file_model(golden_dir):
    """two instances, each loading its own layer range of the same file (-sm layer: src/llama-model.cpp:1949-1972), chained through device hand-off buffers"""
    import torch
    be = backend()
    whole = ls.GgufLlama(be, golden_dir / NAME, n_ctx=32)
    a = ls.GgufLlama(be, golden_dir / NAME, n_ctx=32, layer_begin=0, layer_end=1)
    b = ls.GgufLlama(be, golden_dir / NAME, n_ctx=32, layer_begin=1, layer_end=2)
    try:
        assert not a.has_output and b.has_output
        for toks in [[3, 1, 4, 1, 5], [9], [2]]:
            hand = torch.empty(len(toks)*256, dtype=torch.float32, device="cuda")
            a.decode(toks, dev_result_out=hand.data_ptr(), want_host=False)
            got = b.decode(None, dev_act_in=hand.data_ptr(), n_tokens=len(toks))
            assert np.array_equal(got, whole.decode(toks))
    finally:
        whole.free(); a.free(); b.free()


def test_bench_runs_on_a_gguf_file(golden_dir):
    """bench.py --gguf FILE: the llama-bench protocol on a model read from a file; one JSON line that names the file as its data"""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gguf", str(golden_dir / NAME), "--steps", "16", "--warmup", "4", "--pp", "32", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["data"] == f"gguf file {NAME}" and d["value"] > 0 and d["steps"] == 16 and d["roofline"]["frac"] > 0
