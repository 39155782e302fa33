"""llama_synth.py — Python handle on the C++ synthetic-model harness (csrc/harness/llama_harness.cpp).

The harness restates the reference's CALLER rows (llm_build_llama, build_attn, the unified KV cache's
set_rows writes, llama_context::decode — SURVEY.md §8 a4-a11) through the ggml API; this module only
fills in hyper-parameters and forwards calls. Model shapes: SURVEY.md §8 header.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import ggml_ctypes as gg

# llama_ftype ids (include/llama.h) used by BASELINE.json's configs
FTYPE = {"Q4_0": 2, "Q8_0": 7, "Q4_K_M": 15, "Q5_K_M": 17, "Q6_K": 18, "MXFP4_MOE": 38}


class hparams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_embd", "n_ff", "n_layer", "n_head", "n_head_kv", "n_embd_head", "n_vocab", "n_ctx",
                                         "ftype", "rope_type", "n_ctx_orig", "has_rope_freqs", "is_70b")] + \
               [(n, C.c_float) for n in ("rope_freq_base", "rope_freq_scale", "f_norm_rms_eps")] + \
               [(n, C.c_int32) for n in ("layer_begin", "layer_end", "has_output", "n_seq_max", "n_expert", "n_expert_used", "arch", "flash_attn",
                                         "n_swa", "swa_pattern", "n_ubatch", "type_k", "row_split", "type_v")]


# model shapes used by the configs
MODELS = {
    # Llama-3-8B: n_embd=4096, n_ff=14336, n_layer=32, n_head=32, n_head_kv=8, head=128, n_vocab=128256
    "llama3-8b": dict(n_embd=4096, n_ff=14336, n_layer=32, n_head=32, n_head_kv=8, n_embd_head=128, n_vocab=128256,
                      rope_freq_base=500000.0, n_ctx_orig=8192, is_70b=0),
    # Llama-3-70B: 8192/28672/80/64/8/128/128256
    "llama3-70b": dict(n_embd=8192, n_ff=28672, n_layer=80, n_head=64, n_head_kv=8, n_embd_head=128, n_vocab=128256,
                       rope_freq_base=500000.0, n_ctx_orig=8192, is_70b=1),
    # stories15M (tinyllama): 288/768/6/6/6/48/32000 — BASELINE.json configs[0] shape
    "stories15m": dict(n_embd=288, n_ff=768, n_layer=6, n_head=6, n_head_kv=6, n_embd_head=48, n_vocab=32000,
                       rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0),
    # Mixtral-8x7B: 4096/14336/32/32/8/128/32000, 8 experts top-2 (llm_build_llama's MoE branch)
    "mixtral-8x7b": dict(n_embd=4096, n_ff=14336, n_layer=32, n_head=32, n_head_kv=8, n_embd_head=128, n_vocab=32000,
                         rope_freq_base=1000000.0, n_ctx_orig=32768, is_70b=0, n_expert=8, n_expert_used=2),
    # gpt-oss-20b: n_embd=2880, n_ff_exp=2880, 24 layers, 64/8 heads x 64, 32 experts top-4, vocab 201088 (llm_build_openai_moe_iswa:
    # even layers attend through a 128-token sliding window with their own cache of min(n_ctx, PAD(128 + n_ubatch)) cells)
    "gpt-oss-20b": dict(n_embd=2880, n_ff=2880, n_layer=24, n_head=64, n_head_kv=8, n_embd_head=64, n_vocab=201088,
                        rope_freq_base=150000.0, n_ctx_orig=4096, is_70b=0, n_expert=32, n_expert_used=4, arch=1, rope_type=2,
                        n_swa=128, swa_pattern=2),
    # small MoE models for graph-level parity tests
    "tiny-moe": dict(n_embd=256, n_ff=512, n_layer=2, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=512,
                     rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0, n_expert=8, n_expert_used=2),
    # 32 experts top-4: from 16 experts on the one-token router runs on several workgroups (elem.hip k_moe_route_wide)
    "tiny-moe32": dict(n_embd=256, n_ff=256, n_layer=2, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=512,
                       rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0, n_expert=32, n_expert_used=4),
    # gpt-oss-shaped: head size 64 (the fused attention kernels' other instantiation), sinks, biases, and the iswa cache pair: even layers
    # attend through a 16-token window held in a 32-cell ring (PAD(n_swa + n_ubatch, 32)), odd layers through the full cache
    "tiny-oai": dict(n_embd=128, n_ff=128, n_layer=2, n_head=8, n_head_kv=2, n_embd_head=64, n_vocab=512,
                     rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0, n_expert=8, n_expert_used=4, arch=1, rope_type=2,
                     n_swa=16, swa_pattern=2, n_ubatch=8),
    # head size 128 with GQA 4:1 at test size (the fused attention kernels are instantiated for 64 and 128)
    "tiny-hd128": dict(n_embd=512, n_ff=512, n_layer=2, n_head=4, n_head_kv=1, n_embd_head=128, n_vocab=512,
                       rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0),
    # ONE Llama-3-8B-shaped layer with a small vocabulary: the prefill fusions (grouped QKV with ROPE and KV store in its combine pass, 16-wave
    # tiles, gate/up/SwiGLU, split-k combine + norm) only engage at these sizes with >= 256 tokens
    "llama3-8b-1l": dict(n_embd=4096, n_ff=14336, n_layer=1, n_head=32, n_head_kv=8, n_embd_head=128, n_vocab=512,
                         rope_freq_base=500000.0, n_ctx_orig=8192, is_70b=0),
    # TWO Llama-3-70B-shaped layers (BASELINE.json configs[3] shapes: k = 8192 / 28672, 64 heads over 8 KV heads; layer 0 takes the LLM_TYPE_70B
    # Q5_K bump of attn_v, layer 1 the use_more_bits Q6_K) with a small vocabulary
    "llama3-70b-2l": dict(n_embd=8192, n_ff=28672, n_layer=2, n_head=64, n_head_kv=8, n_embd_head=128, n_vocab=512,
                          rope_freq_base=500000.0, n_ctx_orig=8192, is_70b=1),
    # the perplexity-statistics model: wide enough that ONE flipped int8 rounding in an activation block is a small perturbation (the 256-wide
    # models amplify such flips into 1e-4 of KL divergence between two implementations of the same arithmetic), small enough that the oracle
    # evaluates 8192 positions twice in about a minute
    "mid-1k": dict(n_embd=1024, n_ff=2816, n_layer=2, n_head=8, n_head_kv=2, n_embd_head=128, n_vocab=512,
                   rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0),
    # small model for graph-level parity tests (oracle finishes in seconds)
    "tiny": dict(n_embd=256, n_ff=512, n_layer=2, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=512,
                 rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0),
    # the same with four layers: the smallest model both ranks of a 2-way layer split own layers of (tests/test_gpu_layer_split.py)
    "tiny4": dict(n_embd=256, n_ff=512, n_layer=4, n_head=4, n_head_kv=2, n_embd_head=64, n_vocab=512,
                  rope_freq_base=10000.0, n_ctx_orig=256, is_70b=0),
}

_hz = None


def harness():
    global _hz
    if _hz is None:
        gg.base()
        p = gg.LIBDIR / "libmi355x-harness.so"
        if not p.exists():
            raise RuntimeError(f"{p} is missing: build the native code first (no fallback exists)")
        L = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
        L.mi_llama_create.restype = C.c_void_p; L.mi_llama_create.argtypes = [C.c_void_p, C.POINTER(hparams), C.c_uint64]
        L.mi_llama_free.argtypes = [C.c_void_p]
        L.mi_llama_weight_bytes.restype = C.c_uint64; L.mi_llama_weight_bytes.argtypes = [C.c_void_p]
        L.mi_llama_n_past.restype = C.c_int; L.mi_llama_n_past.argtypes = [C.c_void_p, C.c_int]
        L.mi_llama_kv_clear.argtypes = [C.c_void_p]
        L.mi_llama_n_result.restype = C.c_int; L.mi_llama_n_result.argtypes = [C.c_void_p]
        L.mi_llama_get_tensor.restype = gg.tensor_p; L.mi_llama_get_tensor.argtypes = [C.c_void_p, C.c_char_p]
        L.mi_llama_last_logits.restype = C.c_void_p; L.mi_llama_last_logits.argtypes = [C.c_void_p]
        L.mi_llama_decode.restype = C.c_int
        L.mi_llama_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mi_llama_synth_embedding.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.mi_llama_graph_nodes.restype = C.c_int; L.mi_llama_graph_nodes.argtypes = [C.c_void_p, C.c_int]
        L.mi_llama_create_from_gguf.restype = C.c_void_p
        L.mi_llama_create_from_gguf.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(hparams), C.c_char_p, C.c_int]
        L.mi_gguf_describe.restype = C.c_longlong; L.mi_gguf_describe.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong]
        _hz = L
    return _hz


class SynthLlama:
    def __init__(self, backend: gg.Backend, model="llama3-8b", ftype="Q4_K_M", n_ctx=128, seed=1, layer_begin=0, layer_end=None,
                 has_output=None, rope_type=0, has_rope_freqs=False, n_seq_max=1, flash_attn=False, row_split=0, type_k=0, type_v=0, **over):
        cfg = dict(MODELS[model]); cfg.update(over)
        self.cfg = cfg
        n_layer = cfg["n_layer"]
        layer_end = n_layer if layer_end is None else layer_end
        has_output = (layer_end == n_layer) if has_output is None else has_output
        hp = hparams()
        for k in ("n_embd", "n_ff", "n_layer", "n_head", "n_head_kv", "n_embd_head", "n_vocab", "n_ctx_orig", "is_70b"):
            setattr(hp, k, cfg[k])
        pad = 256 if flash_attn else 32       # the KV cache's padding (src/llama-kv-cache-unified.cpp:2407-2410)
        hp.n_ctx = (n_ctx + pad - 1) // pad * pad
        hp.flash_attn = int(flash_attn)
        hp.ftype = FTYPE[ftype]
        hp.rope_type = cfg.get("rope_type", rope_type)
        hp.n_expert, hp.n_expert_used, hp.arch = cfg.get("n_expert", 0), cfg.get("n_expert_used", 0), cfg.get("arch", 0)
        hp.has_rope_freqs = int(has_rope_freqs)
        hp.rope_freq_base = cfg["rope_freq_base"]; hp.rope_freq_scale = 1.0; hp.f_norm_rms_eps = 1e-5
        hp.layer_begin, hp.layer_end, hp.has_output = layer_begin, layer_end, int(has_output)
        hp.n_seq_max = n_seq_max
        hp.n_swa, hp.swa_pattern, hp.n_ubatch = cfg.get("n_swa", 0), cfg.get("swa_pattern", 0), cfg.get("n_ubatch", 512)
        hp.type_k = type_k             # K cache type (ggml type id; 0 = F16): llama-bench -ctk q8_0 = 8
        cfg["type_k"] = type_k
        hp.type_v = type_v             # V cache type (llama-bench -ctv; anything but F16 needs flash_attn)
        cfg["type_v"] = type_v; cfg["flash_attn"] = int(flash_attn)
        hp.row_split = row_split       # -sm row: the weight matrices' rows spread over this many devices (csrc/backend.cpp: the split buffer type)
        self.hp = hp
        self.backend = backend
        self.L = harness()
        self.m = self.L.mi_llama_create(backend.be, C.byref(hp), seed)
        if not self.m:
            raise RuntimeError("mi_llama_create failed")
        self.has_output = bool(has_output)
        self.n_result = self.L.mi_llama_n_result(self.m)

    @property
    def weight_bytes(self):
        """bytes of every MUL_MAT weight tensor held by this instance = algorithmic bytes read per decoded token"""
        return int(self.L.mi_llama_weight_bytes(self.m))

    @property
    def n_past(self):
        return self.L.mi_llama_n_past(self.m, 0)

    def kv_clear(self):
        self.L.mi_llama_kv_clear(self.m)

    def tensor(self, name):
        t = self.L.mi_llama_get_tensor(self.m, name.encode())
        if not t:
            raise KeyError(name)
        return t

    def decode(self, tokens, dev_act_in=None, dev_result_out=None, want_host=True, sync=True, n_tokens=None, seq=0, view=False):
        """one llama_decode of len(tokens) tokens; returns the host result (logits of the last token) or None. The result lands in the model's pinned
        output buffer (llama_context's buf_output, src/llama-context.cpp:1260-1330): view=True returns a numpy view of it (valid until the next decode,
        like llama_get_logits), otherwise a copy"""
        if tokens is not None:
            tok = np.ascontiguousarray(tokens, dtype=np.int32)
            n = tok.size; tp = tok.ctypes.data_as(C.c_void_p)
        else:
            n = n_tokens; tp = None
        rc = self.L.mi_llama_decode(self.m, seq, tp, n, C.c_void_p(dev_act_in) if dev_act_in else None, None,
                                    C.c_void_p(dev_result_out) if dev_result_out else None, int(bool(sync)) | (0 if want_host else 2))
        if rc != 0:
            raise RuntimeError(f"mi_llama_decode returned {rc}")
        if not want_host or dev_result_out:
            return None
        n_res = self.L.mi_llama_n_result(self.m)
        out = np.ctypeslib.as_array(C.cast(self.L.mi_llama_last_logits(self.m), C.POINTER(C.c_float)), shape=(n_res,))
        if not sync:
            return out if view else None        # (not complete before the caller synchronizes)
        return out if view else out.copy()

    def embedding(self, token):
        out = np.empty(self.cfg["n_embd"], dtype=np.float32)
        self.L.mi_llama_synth_embedding(self.m, int(token), out.ctypes.data_as(C.c_void_p))
        return out

    def graph_nodes(self, n_tokens=1):
        return self.L.mi_llama_graph_nodes(self.m, n_tokens)

    def free(self):
        if self.m:
            self.L.mi_llama_free(self.m)
            self.m = None


class GgufLlama(SynthLlama):
    """a model read from a GGUF file (csrc/harness: gguf_file.h + mi_llama_create_from_gguf): hyper-parameters, tensor types and weights come
    from the file, the embedding rows from its token_embd; the rest of the handle (decode, tensor, embedding ...) is SynthLlama's"""

    def __init__(self, backend: gg.Backend, path, n_ctx=128, n_seq_max=1, flash_attn=False, n_ubatch=512, layer_begin=0, layer_end=-1):
        self.backend = backend
        self.L = harness()
        hp = hparams()
        err = C.create_string_buffer(512)
        self.m = self.L.mi_llama_create_from_gguf(backend.be, str(path).encode(), n_ctx, n_seq_max, int(flash_attn), n_ubatch, layer_begin, layer_end,
                                                  C.byref(hp), err, len(err))
        if not self.m:
            raise RuntimeError(err.value.decode() or "mi_llama_create_from_gguf failed")
        self.hp = hp
        self.cfg = {k: getattr(hp, k) for k, _ in hparams._fields_}
        self.has_output = bool(hp.has_output)
        self.n_result = self.L.mi_llama_n_result(self.m)


def gguf_describe(path):
    """what the C++ reader sees in a GGUF file, as a dict (metadata, tensor placement, a hash of each tensor's bytes); no GPU needed"""
    import json
    L = harness()
    cap = 1 << 20
    while True:
        buf = C.create_string_buffer(cap)
        n = L.mi_gguf_describe(str(path).encode(), buf, cap)
        if n < 0:
            raise RuntimeError(buf.value.decode())
        if n < cap:
            return json.loads(buf.value.decode())
        cap = n + 1
