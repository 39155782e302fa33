#!/usr/bin/env python3
"""Generate tests/golden/tiny_llama_q4_k_m.gguf and tiny_llama_q4_k_m.gguf.json with the REFERENCE's own GGUF writer and reader
(gguf-py/gguf/gguf_writer.py, gguf_reader.py): the file is what a converter would hand the model loader, the JSON is what the
reference's reader says the file holds (metadata, tensor placement, a hash of each tensor's bytes). The C++ reader
(csrc/harness/gguf_file.h) and the oracle's reader (oracle/gguf_ref.py) are pinned against that JSON.

Run ONLY in the build container (the reference tree does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_gguf_fixture.py

The committed files are DATA. The model is a two-layer Llama with Q4_K_M's type mix (token_embd Q4_K, output Q6_K, attn_v / ffn_down
Q6_K where use_more_bits says so: src/llama-quant.cpp:185-187,302-364) over random valid blocks, plus tokenizer-style arrays so that
every value type of the format (gguf-py/gguf/constants.py:2791-2804) occurs at least once.
"""
import json
import os
import sys
from pathlib import Path

import numpy as np

REF = Path(os.environ.get("GGUF_PY", "/root/reference/gguf-py"))
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))
import gguf  # noqa: E402
from gguf.constants import GGMLQuantizationType as T, GGUFValueType as V  # noqa: E402

OUT = Path(__file__).resolve().parent
NAME = "tiny_llama_q4_k_m.gguf"
CFG = dict(n_embd=256, n_ff=256, n_layer=2, n_head=4, n_head_kv=2, n_vocab=256)
BLOCK = {T.Q4_K: (256, 144), T.Q6_K: (256, 210), T.Q8_0: (32, 34), T.Q4_0: (32, 18), T.Q5_K: (256, 176)}


def random_blocks(rng, qt, rows, k):
    """rows x k elements of random valid blocks whose dequantized values have std ~ 1/sqrt(k)"""
    blck, ts = BLOCK[qt]
    nb = rows * k // blck
    b = rng.integers(0, 256, size=(nb, ts), dtype=np.uint8)
    sigma = 1.0 / np.sqrt(k)
    u = rng.uniform(0.75, 1.25, size=nb)

    def put(off, val):
        b[:, off:off + 2] = val.astype(np.float16).view(np.uint8).reshape(-1, 2)
    if qt == T.Q4_K:
        put(0, u * sigma / 258.0); put(2, u * sigma / 258.0 * 7.5)
    elif qt == T.Q5_K:
        put(0, u * sigma / 530.0); put(2, u * sigma / 530.0 * 15.5)
    elif qt == T.Q6_K:
        put(208, u * sigma / 1367.0)
    elif qt == T.Q8_0:
        put(0, u * sigma / 73.9)
    elif qt == T.Q4_0:
        put(0, u * sigma / 4.61)
    return b.reshape(rows, k // blck * ts)


def use_more_bits(il, n):
    return il < n // 8 or il >= 7 * n // 8 or (il - n // 8) % 3 == 2


def main():
    rng = np.random.default_rng(20261004)
    c = CFG
    hd = c["n_embd"] // c["n_head"]
    w = gguf.GGUFWriter(OUT / NAME, "llama")
    w.add_name("tiny synthetic llama")
    w.add_context_length(256)
    w.add_embedding_length(c["n_embd"])
    w.add_block_count(c["n_layer"])
    w.add_feed_forward_length(c["n_ff"])
    w.add_head_count(c["n_head"])
    w.add_head_count_kv(c["n_head_kv"])
    w.add_rope_dimension_count(hd)
    w.add_rope_freq_base(10000.0)
    w.add_layer_norm_rms_eps(1e-5)
    w.add_file_type(15)                                         # LLAMA_FTYPE_MOSTLY_Q4_K_M
    w.add_vocab_size(c["n_vocab"])
    # tokenizer-style arrays and one key of every remaining scalar type
    w.add_tokenizer_model("llama")
    w.add_token_list([f"<tok{i}>" if i != 7 else 'quote"back\\slash' for i in range(c["n_vocab"])])
    w.add_token_scores([float(np.float32(-0.25 * i)) for i in range(c["n_vocab"])])
    w.add_token_types([1 + (i % 6 == 0) for i in range(c["n_vocab"])])
    w.add_bos_token_id(1)
    w.add_eos_token_id(2)
    w.add_add_bos_token(True)
    w.add_key_value("test.u8", 200, V.UINT8)
    w.add_key_value("test.i8", -100, V.INT8)
    w.add_key_value("test.u16", 60000, V.UINT16)
    w.add_key_value("test.i16", -30000, V.INT16)
    w.add_key_value("test.i32", -2000000000, V.INT32)
    w.add_key_value("test.u64", 2**63 + 5, V.UINT64)
    w.add_key_value("test.i64", -(2**62) - 3, V.INT64)
    w.add_key_value("test.f64", 1.0 / 3.0, V.FLOAT64)
    w.add_key_value("test.u8s", [1, 2, 250], V.ARRAY, sub_type=V.UINT8)

    def quant(name, qt, rows, k):
        w.add_tensor(name, random_blocks(rng, qt, rows, k), raw_dtype=qt)

    def f32(name, n, lo, hi):
        w.add_tensor(name, rng.uniform(lo, hi, size=n).astype(np.float32))

    quant("token_embd.weight", T.Q4_K, c["n_vocab"], c["n_embd"])
    for il in range(c["n_layer"]):
        more = use_more_bits(il, c["n_layer"])
        p = f"blk.{il}."
        f32(p + "attn_norm.weight", c["n_embd"], 0.5, 1.5)
        quant(p + "attn_q.weight", T.Q4_K, hd * c["n_head"], c["n_embd"])
        quant(p + "attn_k.weight", T.Q4_K, hd * c["n_head_kv"], c["n_embd"])
        quant(p + "attn_v.weight", T.Q6_K if more else T.Q4_K, hd * c["n_head_kv"], c["n_embd"])
        quant(p + "attn_output.weight", T.Q4_K, c["n_embd"], hd * c["n_head"])
        f32(p + "ffn_norm.weight", c["n_embd"], 0.5, 1.5)
        quant(p + "ffn_gate.weight", T.Q4_K, c["n_ff"], c["n_embd"])
        quant(p + "ffn_up.weight", T.Q4_K, c["n_ff"], c["n_embd"])
        quant(p + "ffn_down.weight", T.Q6_K if more else T.Q4_K, c["n_embd"], c["n_ff"])
    f32("output_norm.weight", c["n_embd"], 0.5, 1.5)
    quant("output.weight", T.Q6_K, c["n_vocab"], c["n_embd"])
    w.write_header_to_file()
    w.write_kv_data_to_file()
    w.write_tensors_to_file()
    w.close()

    # ---- what the reference's reader sees
    r = gguf.GGUFReader(OUT / NAME)
    kv = []
    for key, f in r.fields.items():
        if key.startswith("GGUF."):
            continue
        t = f.types[0]
        e = {"key": key, "type": int(t)}
        val = f.contents()
        if t == V.ARRAY:
            it = f.types[1]
            e["item_type"] = int(it)
            e["count"] = len(val)
            e["value"] = list(val[:16])
        else:
            e["value"] = val
        kv.append(e)
    version = int(r.fields["GGUF.version"].parts[0][0])
    tensors = []
    for t in r.tensors:
        raw = np.asarray(t.data).view(np.uint8).reshape(-1)
        h = 1469598103934665603
        for byte in raw.tobytes():
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        tensors.append({"name": t.name, "type": int(t.tensor_type), "ne": [int(x) for x in t.shape], "offset": int(t.data_offset - r.data_offset),
                        "nbytes": int(t.n_bytes), "fnv1a": f"{h:016x}"})
    desc = {"version": version, "alignment": int(r.alignment), "data_offset": int(r.data_offset), "kv": kv, "tensors": tensors}
    (OUT / (NAME + ".json")).write_text(json.dumps(desc, indent=1) + "\n")
    print("wrote", OUT / NAME, (OUT / NAME).stat().st_size, "bytes;", len(kv), "keys,", len(tensors), "tensors")


if __name__ == "__main__":
    main()
