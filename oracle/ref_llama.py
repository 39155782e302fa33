"""ref_llama.py — TEST INFRASTRUCTURE (oracle): a numpy restatement of the graphs the reference builds for a Llama-family model, evaluated with
the oracle's mat-mul (oracle.mul_mat_2d: "cpu" = the CPU backend's arithmetic — Q8 activations, integer dots; "exact" = dequantized weights x f32)
and ops_ref's element ops. Used by tests/ and by bench.py's checker leg (perplexity delta); never by the product path.
read_weights(m, gg) pulls the synthetic model's tensors back from the device through the backend's tensor_get."""
import numpy as np

import oracle as orc
import ops_ref as ref


def read_weights(m, gg):
    W = {}
    moe = m.cfg.get("n_expert", 0) > 0
    oai = m.cfg.get("arch", 0) == 1
    names = ["attn_norm.weight", "attn_q.weight", "attn_k.weight", "attn_v.weight", "attn_output.weight", "ffn_norm.weight"]
    names += ["ffn_gate_inp.weight", "ffn_gate_exps.weight", "ffn_up_exps.weight", "ffn_down_exps.weight"] if moe else \
             ["ffn_gate.weight", "ffn_up.weight", "ffn_down.weight"]
    if oai:
        names += ["attn_q.bias", "attn_k.bias", "attn_v.bias", "attn_output.bias", "attn_sinks.weight",
                  "ffn_gate_inp.bias", "ffn_gate_exps.bias", "ffn_up_exps.bias", "ffn_down_exps.bias"]
    for il in range(m.cfg["n_layer"]):
        for nm in names:
            t = m.tensor(f"blk.{il}.{nm}")
            a = gg.tensor_get(t)[0]
            key = nm[:-7] if nm.endswith(".weight") else nm
            W[(il, key)] = (t.contents.type, (a if t.contents.ne[2] > 1 else a[0]).copy())
    for nm in ("output_norm", "output"):
        t = m.tensor(f"{nm}.weight")
        W[nm] = (t.contents.type, gg.tensor_get(t)[0, 0].copy())
    return W


def mm(W, key, x, mode):
    qt, data = W[key]
    return orc.mul_mat_2d(data, qt, x.astype(np.float32), mode).astype(np.float32)


class RefLlama:
    """numpy restatement of the graph llm_build_llama emits (src/llama-model.cpp:5969-6123), of its MoE branch (build_moe_ffn,
    src/llama-graph.cpp:811-1023) and of llm_build_openai_moe_iswa (src/llama-model.cpp:17610-17738, incl. the sliding window of its even layers)"""

    def __init__(self, cfg, W, kv_size, mode):
        self.f16_attn = mode == "cpu16"
        mode = "cpu" if mode == "cpu16" else mode
        self.c, self.W, self.mode = cfg, W, mode
        hd, hkv = cfg["n_embd_head"], cfg["n_head_kv"]
        self.k = np.zeros((cfg["n_layer"], kv_size, hkv, hd), np.float16)
        self.v = np.zeros((cfg["n_layer"], kv_size, hkv, hd), np.float16)
        # a quantized K cache (cfg["type_k"], -ctk q8_0 / q4_0): rows are quantized on the way in by the reference row quantizer (SET_ROWS) and K.q
        # becomes a quantized mat-mul (src0 = the cache), evaluated per head over the head's blocks of the row
        self.tk = cfg.get("type_k", 0)
        if self.tk and not cfg.get("flash_attn", 0):
            self.kq = np.zeros((cfg["n_layer"], kv_size, orc.row_size(self.tk, hkv*hd)), np.uint8)
        # -fa 1 with a quantized / bf16 cache (cfg["flash_attn"], type_k / type_v): FLASH_ATTN_EXT dequantizes the rows it reads (ggml/src/ggml-cpu/ops.cpp
        # ggml_compute_forward_flash_attn_ext_f16: V through v_to_float; K through a dot with the Q8_0-quantized q — restated here as the exact product
        # with the dequantized K row, the dot's own quantization of q being below the 5e-4 the reference's test allows)
        self.fa = bool(cfg.get("flash_attn", 0))
        self.tv = cfg.get("type_v", 0)
        if self.tv or (self.fa and self.tk):
            self.kd = np.zeros((cfg["n_layer"], kv_size, hkv, hd), np.float32); self.vd = np.zeros((cfg["n_layer"], kv_size, hkv, hd), np.float32)
        self.n_past = 0
        self.selected = []      # expert choices, so that a test can tell a routing flip from an arithmetic error

    def moe_ffn(self, il, h):
        c, W, mode = self.c, self.W, self.mode
        n_used, oai = c["n_expert_used"], c.get("arch", 0) == 1
        logits = (h.astype(np.float64) @ W[(il, "ffn_gate_inp")][1].astype(np.float64).T).astype(np.float32)
        if oai:
            logits = logits + W[(il, "ffn_gate_inp.bias")][1]
        probs = logits if oai else ref.soft_max(logits[None, None])[0, 0]
        sel = ref.argsort_desc(probs[None, None])[0, 0][:, :n_used]
        self.selected.append(sel.copy())
        w = np.take_along_axis(probs, sel, axis=1).astype(np.float32)
        if oai:
            w = ref.soft_max(w[None, None])[0, 0]
        else:
            w = (w / w.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
        qt_u, up_w = W[(il, "ffn_up_exps")]; qt_g, gate_w = W[(il, "ffn_gate_exps")]; qt_d, down_w = W[(il, "ffn_down_exps")]
        x3 = h[:, None, :].astype(np.float32)
        up = orc.mul_mat_id(up_w, qt_u, x3, sel, mode).astype(np.float32)
        gate = orc.mul_mat_id(gate_w, qt_g, x3, sel, mode).astype(np.float32)
        if oai:
            up = up + W[(il, "ffn_up_exps.bias")][1][sel]; gate = gate + W[(il, "ffn_gate_exps.bias")][1][sel]
            act = ref.swiglu_oai(gate, up).astype(np.float32)
        else:
            act = ref.swiglu(gate, up).astype(np.float32)
        ex = orc.mul_mat_id(down_w, qt_d, act, sel, mode).astype(np.float32)
        if oai:
            ex = ex + W[(il, "ffn_down_exps.bias")][1][sel]
        ex = (ex * w[:, :, None]).astype(np.float32)
        out = ex[:, 0]
        for i in range(1, n_used):
            out = out + ex[:, i]
        return out

    def decode(self, emb):
        c, W = self.c, self.W
        n_tok = emb.shape[0]
        hd, nh, hkv = c["n_embd_head"], c["n_head"], c["n_head_kv"]
        oai, moe, rmode = c.get("arch", 0) == 1, c.get("n_expert", 0) > 0, c.get("rope_type", 0)
        pos = np.arange(self.n_past, self.n_past + n_tok).astype(np.int32)
        x = emb.astype(np.float32)
        for il in range(c["n_layer"]):
            last = il == c["n_layer"] - 1
            h = (ref.rms_norm(x, 1e-5) * W[(il, "attn_norm")][1]).astype(np.float32)
            q = mm(W, (il, "attn_q"), h, self.mode); k = mm(W, (il, "attn_k"), h, self.mode); v = mm(W, (il, "attn_v"), h, self.mode)
            if oai:
                q = q + W[(il, "attn_q.bias")][1]; k = k + W[(il, "attn_k.bias")][1]; v = v + W[(il, "attn_v.bias")][1]
            q = q.reshape(1, n_tok, nh, hd); k = k.reshape(1, n_tok, hkv, hd); v = v.reshape(n_tok, hkv, hd)
            q = ref.rope(q, pos, hd, rmode, c["n_ctx_orig"], c["rope_freq_base"]).astype(np.float32)[0]
            k = ref.rope(k, pos, hd, rmode, c["n_ctx_orig"], c["rope_freq_base"]).astype(np.float32)[0]
            self.k[il, pos] = k.astype(np.float16)
            if self.tk and not self.fa:
                self.kq[il, pos] = orc.quantize(k.reshape(n_tok, hkv*hd), self.tk)
            self.v[il, pos] = v.astype(np.float16)
            n_kv = self.n_past + n_tok
            K = self.k[il, :n_kv].astype(np.float32); V = self.v[il, :n_kv].astype(np.float32)
            fa_deq = self.tv or (self.fa and self.tk)
            if fa_deq:
                def through(x, t):      # the row as the cache holds it, read back
                    if t in (0, orc.F16):
                        return x.astype(np.float16).astype(np.float32)
                    if t == orc.BF16:
                        u = np.ascontiguousarray(x, np.float32).view(np.uint32)
                        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
                    return orc.dequantize(orc.quantize(x.reshape(n_tok, hkv*hd), t), t).reshape(x.shape)
                self.kd[il, pos] = through(k, self.tk); self.vd[il, pos] = through(v, self.tv)
                K = self.kd[il, :n_kv]; V = self.vd[il, :n_kv]
            out = np.zeros((n_tok, nh, hd), np.float32)
            for hh in range(nh):
                kvh = hh // (nh // hkv)
                # the CPU backend's mat-mul with an F16 src0 converts src1 to its vec_dot_type, F16 (ggml_compute_forward_mul_mat): q, and below
                # the probabilities, are rounded to f16 before the dot; sums in f32. Mode "cpu16" restates that too (used where the CPU path as
                # a whole is the yardstick: the perplexity delta); "cpu" and "exact" keep q and p in f32, as this backend's decode kernel does.
                qh = q[:, hh, :].astype(np.float16).astype(np.float64) if self.f16_attn else q[:, hh, :].astype(np.float64)
                if self.tk and not fa_deq:
                    hb = orc.row_size(self.tk, hd)
                    s = orc.mul_mat_2d(np.ascontiguousarray(self.kq[il, :n_kv, kvh*hb:(kvh + 1)*hb]), self.tk, q[:, hh, :], self.mode).astype(np.float32)
                else:
                    s = (qh @ K[:, kvh, :].T.astype(np.float64)).astype(np.float32)   # kq (f32 result)
                s = s.astype(np.float64) / np.sqrt(hd)
                causal = np.arange(n_kv)[None, :] <= pos[:, None]
                n_swa, pat = c.get("n_swa", 0), c.get("swa_pattern", 0)
                if n_swa > 0 and (pat <= 0 or il % pat < pat - 1):      # sliding-window layer (llama_hparams::set_swa_pattern; is_masked_swa: p1 - p0 >= n_swa)
                    causal = causal & (pos[:, None] - np.arange(n_kv)[None, :] < n_swa)
                s = np.where(causal, s, -np.inf)
                mx = s.max(-1, keepdims=True)
                if oai:     # attention sink: one more logit per head in the max and the denominator (src/llama-graph.cpp:1313)
                    sink = float(W[(il, "attn_sinks")][1].reshape(-1)[hh])
                    mx = np.maximum(mx, sink)
                p = np.exp(s - mx); den = p.sum(-1, keepdims=True)
                if oai:
                    den = den + np.exp(sink - mx)
                p = (p / den).astype(np.float32)
                if self.f16_attn:
                    p = p.astype(np.float16).astype(np.float32)
                out[:, hh, :] = (p.astype(np.float64) @ V[:, kvh, :].astype(np.float64)).astype(np.float32)
            a = mm(W, (il, "attn_output"), out.reshape(n_tok, nh * hd), self.mode)
            if oai:
                a = a + W[(il, "attn_output.bias")][1]
            if last:
                a = a[-1:]; x = x[-1:]
            ffn_inp = a + x
            h = (ref.rms_norm(ffn_inp, 1e-5) * W[(il, "ffn_norm")][1]).astype(np.float32)
            if moe:
                x = self.moe_ffn(il, h) + ffn_inp
            else:
                up = mm(W, (il, "ffn_up"), h, self.mode); gate = mm(W, (il, "ffn_gate"), h, self.mode)
                act = ref.swiglu(gate, up).astype(np.float32)
                x = mm(W, (il, "ffn_down"), act, self.mode) + ffn_inp
        h = (ref.rms_norm(x, 1e-5) * W["output_norm"][1]).astype(np.float32)
        self.n_past += n_tok
        return mm(W, "output", h, self.mode)[0]


def nll(logits, tok):
    """next-token negative log-likelihood from one row of logits"""
    z = logits.astype(np.float64); z = z - z.max()
    return float(np.log(np.exp(z).sum()) - z[tok])


def _mean_se(a):
    a = np.asarray(a, np.float64)
    return float(a.mean()), float(a.std(ddof=1)/np.sqrt(len(a))) if len(a) > 1 else 0.0


def logit_parity(m, gg, n_seq=8, seq_len=128, modes=("cpu", "cpu16", "exact"), seed=77, prefill=True, min_prefill=9, W=None):
    """The statistics llama-perplexity --kl-divergence reports (tools/perplexity/perplexity.cpp:1743-2005: mean KL divergence, RMS of the
    probability / logit differences, top-1 agreement, ln(PPL(Q)/PPL(base)), each with its standard error), between THIS backend's logits and
    the oracle's on the same weights and tokens, over n_seq independent token streams of seq_len positions (the cache is cleared between
    them, as between perplexity chunks). The backend's logits at a position are taken twice: token by token (decode kernels) and, for
    prefixes of >= min_prefill tokens, from one prompt pass over the prefix (prefill kernels). Oracle modes: "cpu" = the CPU backend's mat-mul
    arithmetic (int8 activation blocks, integer dots) with q and the attention probabilities kept in f32 (as this backend's decode kernel
    keeps them); "cpu16" = the same with q and p rounded to f16 first, which is what the CPU backend's F16 mat-mul does — the yardstick
    north_star names; "exact" = dequantized weights x f32. The oracles are also compared with each other: those rows are the yardsticks
    (what the CPU backend's own f16 rounding, and what the weight format itself, do to the same statistics)."""
    W = W if W is not None else read_weights(m, gg)
    nv = m.cfg["n_vocab"]
    rng = np.random.default_rng(seed)
    rows = {}

    def stats(lg, lc, nxt):
        lg = lg.astype(np.float64); lc = lc.astype(np.float64)
        zg = lg - lg.max(); zc = lc - lc.max()
        lpg = zg - np.log(np.exp(zg).sum()); lpc = zc - np.log(np.exp(zc).sum())
        pc = np.exp(lpc)
        return (float((pc*(lpc - lpg)).sum()),                       # KL(P_base || P_other)
                float(np.sqrt(np.mean((lg - lc)**2))),               # RMS delta logit at this position
                float(np.std(lc)),                                   # scale of the logits themselves
                float(-lpg[nxt]), float(-lpc[nxt]),                  # next-token NLL under each
                float(np.exp(lpg[nxt]) - np.exp(lpc[nxt])),          # delta p(next token)
                int(np.argmax(lg) == np.argmax(lc)))

    for s in range(n_seq):
        toks = rng.integers(0, nv, size=seq_len + 1).astype(np.int32)
        refs = {k: RefLlama(m.cfg, W, seq_len + 8, k) for k in modes}
        m.kv_clear()
        ref_logits = {k: [] for k in modes}
        for t in range(seq_len):
            emb = np.stack([m.embedding(int(toks[t]))])
            lg = m.decode([int(toks[t])])
            for k in modes:
                lc = refs[k].decode(emb)
                ref_logits[k].append(lc)
                rows.setdefault(("decode_path", k), []).append(stats(lg, lc, int(toks[t + 1])))
            for a, b in (("cpu16", "cpu"), ("cpu", "exact")):
                if a in modes and b in modes:
                    rows.setdefault(("oracle", f"{a}_vs_{b}"), []).append(stats(ref_logits[a][t], ref_logits[b][t], int(toks[t + 1])))
        if prefill:
            for t in range(min_prefill - 1, seq_len):
                m.kv_clear()
                lg = m.decode([int(x) for x in toks[: t + 1]])
                for k in modes:
                    rows.setdefault(("prefill_path", k), []).append(stats(lg, ref_logits[k][t], int(toks[t + 1])))
    m.kv_clear()
    out = {"positions": n_seq*seq_len, "sequences": n_seq, "seq_len": seq_len}
    for (path, k), r in rows.items():
        a = np.array(r, np.float64)
        kl, se_kl = _mean_se(a[:, 0])
        dn, se_dn = _mean_se(a[:, 3] - a[:, 4])                      # paired: ln PPL(other) - ln PPL(base)
        out.setdefault(path, {})[k] = {
            "positions": int(len(a)), "kl_mean": kl, "kl_se": se_kl, "kl_max": float(a[:, 0].max()),
            "rms_dlogit_mean": float(a[:, 1].mean()), "rms_dlogit_max": float(a[:, 1].max()),
            "rms_dlogit_over_logit_std": float((a[:, 1]/a[:, 2]).mean()),
            "delta_ln_ppl": dn, "delta_ln_ppl_se": se_dn, "ln_ppl_base": float(a[:, 4].mean()),
            "rms_dp_next": float(np.sqrt(np.mean(a[:, 5]**2))), "top1_agree": float(a[:, 6].mean())}
    return out


class RefLlamaStreams:
    """RefLlama for S INDEPENDENT token streams advanced in lockstep (one token per stream per call, every stream at the same position): the
    same arithmetic, step for step (tests/test_oracle_golden.py holds it equal to S separate RefLlama instances), but the mat-muls take all S
    columns in one call and the attention is one einsum per layer instead of S * n_head small products — the perplexity statistics evaluate
    tens of thousands of positions (logit_parity_peaked), which S separate numpy models cannot do in the time a test has. Plain llama
    architecture with an f16 cache only (no experts, sinks, sliding windows, quantized cache)."""

    def __init__(self, cfg, W, n_streams, kv_size, mode):
        assert cfg.get("arch", 0) == 0 and cfg.get("n_expert", 0) == 0 and not cfg.get("type_k", 0)
        self.f16_attn = mode == "cpu16"
        self.mode = "cpu" if mode == "cpu16" else mode
        self.c, self.W, self.S = cfg, W, n_streams
        hd, hkv = cfg["n_embd_head"], cfg["n_head_kv"]
        self.k = np.zeros((cfg["n_layer"], n_streams, kv_size, hkv, hd), np.float16)
        self.v = np.zeros((cfg["n_layer"], n_streams, kv_size, hkv, hd), np.float16)
        self.n_past = 0

    def decode(self, emb):
        """emb [S, n_embd]: the next token of every stream; returns logits [S, n_vocab]"""
        c, W, S = self.c, self.W, self.S
        hd, nh, hkv = c["n_embd_head"], c["n_head"], c["n_head_kv"]
        g = nh // hkv
        t = self.n_past
        pos = np.full(S, t, np.int32)
        x = emb.astype(np.float32)
        for il in range(c["n_layer"]):
            h = (ref.rms_norm(x, 1e-5) * W[(il, "attn_norm")][1]).astype(np.float32)
            q = mm(W, (il, "attn_q"), h, self.mode); k = mm(W, (il, "attn_k"), h, self.mode); v = mm(W, (il, "attn_v"), h, self.mode)
            q = ref.rope(q.reshape(1, S, nh, hd), pos, hd, c.get("rope_type", 0), c["n_ctx_orig"], c["rope_freq_base"]).astype(np.float32)[0]
            k = ref.rope(k.reshape(1, S, hkv, hd), pos, hd, c.get("rope_type", 0), c["n_ctx_orig"], c["rope_freq_base"]).astype(np.float32)[0]
            self.k[il, :, t] = k.astype(np.float16); self.v[il, :, t] = v.reshape(S, hkv, hd).astype(np.float16)
            K = self.k[il, :, :t + 1].astype(np.float64); V = self.v[il, :, :t + 1].astype(np.float64)       # [S, T, hkv, hd]
            qh = (q.astype(np.float16) if self.f16_attn else q).astype(np.float64).reshape(S, hkv, g, hd)
            sc = np.einsum("skgd,stkd->skgt", qh, K).astype(np.float32).astype(np.float64) / np.sqrt(hd)     # kq as an f32 result, then the scale
            mx = sc.max(-1, keepdims=True)
            p = np.exp(sc - mx); p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
            if self.f16_attn:
                p = p.astype(np.float16).astype(np.float32)
            out = np.einsum("skgt,stkd->skgd", p.astype(np.float64), V).astype(np.float32).reshape(S, nh*hd)
            ffn_inp = mm(W, (il, "attn_output"), out, self.mode) + x
            h = (ref.rms_norm(ffn_inp, 1e-5) * W[(il, "ffn_norm")][1]).astype(np.float32)
            up = mm(W, (il, "ffn_up"), h, self.mode); gate = mm(W, (il, "ffn_gate"), h, self.mode)
            act = ref.swiglu(gate, up).astype(np.float32)
            x = mm(W, (il, "ffn_down"), act, self.mode) + ffn_inp
        h = (ref.rms_norm(x, 1e-5) * W["output_norm"][1]).astype(np.float32)
        self.n_past += 1
        return mm(W, "output", h, self.mode)


def logit_parity_peaked(m, gg, n_seq=64, seq_len=128, target_ppl=8.0, base="cpu16", others=("cpu",), seed=78, W=None, progress=None):
    """The same statistics on a model WITH STRUCTURE (a random-init model scores every token at chance level, PPL ~ n_vocab, the least
    sensitive probe there is). Two changes make the evaluation behave like a perplexity run of a trained model on real text
    (tools/perplexity/perplexity.cpp:541-642):
      * the lm_head is scaled by a scalar s (applied to the logits of every evaluator alike, which is the same thing), chosen once so that the
        reference model is confident: its perplexity on its own text is ~target_ppl instead of ~n_vocab;
      * the text is what the REFERENCE model (oracle mode `base` = the CPU backend's arithmetic) generates itself: token t + 1 is sampled from
        its softmax(s * logits_t). A model evaluated on its own samples has ln PPL = its entropy, so the reference is well calibrated on the
        text, and a backend whose logits differ pays for it in the next-token likelihood of tokens that matter (the likely ones).
    Decode path only (one token per step, the kernels tg128 runs on). Returns, per evaluator, the paired statistics against `base`."""
    W = W if W is not None else read_weights(m, gg)
    nv = m.cfg["n_vocab"]
    rng = np.random.default_rng(seed)

    def lsm(z):
        z = z.astype(np.float64); z = z - z.max()
        return z - np.log(np.exp(z).sum())

    # the scale: on a short random-token stream, the s at which the reference's mean entropy is ln(target_ppl)
    cal = RefLlama(m.cfg, W, 72, base)
    cl = [cal.decode(np.stack([m.embedding(int(t))])) for t in rng.integers(0, nv, size=64)]
    def mean_entropy(sc):
        return float(np.mean([-(np.exp(lsm(l*sc))*lsm(l*sc)).sum() for l in cl]))
    lo, hi = 0.05, 200.0
    for _ in range(40):
        mid = np.sqrt(lo*hi)
        if mean_entropy(mid) > np.log(target_ppl): lo = mid
        else: hi = mid
    scale = float(np.sqrt(lo*hi))

    rows = {k: [] for k in ("backend",) + tuple(others)}
    # the streams advance in lockstep, `batch` of them at a time: one oracle step serves a whole batch (RefLlamaStreams), the backend decodes
    # each stream's token in its own sequence of the model's cache when it has that many (n_seq_max), else stream after stream
    n_slot = getattr(m, "n_seq_max", 1)
    batch = min(n_seq, 64)
    for b0 in range(0, n_seq, batch):
        nb_ = min(batch, n_seq - b0)
        refs = {k: RefLlamaStreams(m.cfg, W, nb_, seq_len + 8, k) for k in (base,) + tuple(others)}
        toks = rng.integers(0, nv, size=nb_).astype(np.int64)
        hist = [[int(toks[i])] for i in range(nb_)]
        blg = np.zeros((nb_, seq_len, nv), np.float32); nxts = np.zeros((nb_, seq_len), np.int64)
        lcs_all = {k: np.zeros((nb_, seq_len, nv), np.float32) for k in refs}
        for t in range(seq_len):
            emb = np.stack([m.embedding(int(tk)) for tk in toks])
            lcs = {k: refs[k].decode(emb) for k in refs}
            for k in refs:
                lcs_all[k][:, t] = lcs[k]
            nxt = np.zeros(nb_, np.int64)
            for i in range(nb_):
                lpb = lsm(lcs[base][i]*scale)
                nxt[i] = int(rng.choice(nv, p=np.exp(lpb)))              # the reference model writes the text
            nxts[:, t] = nxt
            toks = nxt
            for i in range(nb_):
                hist[i].append(int(nxt[i]))
        # the backend: every stream token by token through the decode kernels (its own cache, cleared per stream)
        for i in range(nb_):
            m.kv_clear()
            for t in range(seq_len):
                blg[i, t] = m.decode([hist[i][t]])
        for i in range(nb_):
            for t in range(seq_len):
                lpb = lsm(lcs_all[base][i, t]*scale); pb = np.exp(lpb); nxt = int(nxts[i, t])
                for name, l in (("backend", blg[i, t]),) + tuple((k, lcs_all[k][i, t]) for k in others):
                    lpo = lsm(l*scale)
                    rows[name].append((float((pb*(lpb - lpo)).sum()), float(-lpo[nxt]), float(-lpb[nxt]), int(np.argmax(l) == np.argmax(lcs_all[base][i, t])),
                                       float(np.sqrt(np.mean((l.astype(np.float64) - lcs_all[base][i, t].astype(np.float64))**2))*scale)))
        if progress:
            progress(min(b0 + batch, n_seq), n_seq)
    m.kv_clear()
    out = {"positions": n_seq*seq_len, "sequences": n_seq, "seq_len": seq_len, "logit_scale": scale, "base": base, "target_ppl": target_ppl}
    for name, r in rows.items():
        a = np.array(r, np.float64)
        kl, se_kl = _mean_se(a[:, 0]); dn, se_dn = _mean_se(a[:, 1] - a[:, 2])
        out[name] = {"positions": int(len(a)), "kl_mean": kl, "kl_se": se_kl, "delta_ln_ppl": dn, "delta_ln_ppl_se": se_dn,
                     "ln_ppl_base": float(a[:, 2].mean()), "ppl_base": float(np.exp(a[:, 2].mean())), "top1_agree": float(a[:, 3].mean()),
                     "rms_dlogit_scaled_mean": float(a[:, 4].mean())}
    return out
