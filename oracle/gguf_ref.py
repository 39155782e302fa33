"""gguf_ref.py — TEST INFRASTRUCTURE (oracle): a numpy restatement of the GGUF container as the reference reads and writes it
(gguf-py/gguf/gguf_reader.py:132-190,230-345 and gguf_writer.py:210-270,1040-1090; constants.py:10-12,2791-2804). Used by tests/ to take a
model's bytes straight from the file (so that a loader which uploaded the wrong bytes cannot pass by agreeing with itself) and to write
variants of the golden file (a tensor removed or reshaped, tied embeddings). Never used by the product path.

Pinned by tests/test_gguf.py: read() of tests/golden/tiny_llama_q4_k_m.gguf equals the description the reference's own reader gave
(tests/golden/tiny_llama_q4_k_m.gguf.json), and write() of what read() returned reproduces the reference writer's file byte for byte.
"""
import struct

import numpy as np

MAGIC = 0x46554747
(U8, I8, U16, I16, U32, I32, F32, BOOL, STR, ARR, U64, I64, F64) = range(13)
_FMT = {U8: "<B", I8: "<b", U16: "<H", I16: "<h", U32: "<I", I32: "<i", F32: "<f", BOOL: "<?", U64: "<Q", I64: "<q", F64: "<d"}
# elements per block, bytes per block (constants.py:2839-2872) for the types the path runs
TYPE_SIZE = {0: (1, 4), 1: (1, 2), 30: (1, 2), 2: (32, 18), 8: (32, 34), 12: (256, 144), 13: (256, 176), 14: (256, 210), 39: (32, 17)}


def nbytes(qt, ne):
    blck, ts = TYPE_SIZE[qt]
    n = 1
    for d in ne:
        n *= d
    return n // blck * ts


def fnv1a(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def read(path):
    """-> dict(version, alignment, data_offset, kv=[{key,type,value | item_type,count,value}], tensors=[{name,type,ne,offset,nbytes,data}])"""
    buf = np.fromfile(path, dtype=np.uint8)
    raw = buf.tobytes()
    pos = 0

    def rd(t):
        nonlocal pos
        v = struct.unpack_from(_FMT[t], raw, pos)[0]
        pos += struct.calcsize(_FMT[t])
        return v

    def rd_str():
        nonlocal pos
        n = rd(U64)
        s = raw[pos:pos + n].decode("utf-8")
        pos += n
        return s

    if rd(U32) != MAGIC:
        raise ValueError("bad magic")
    version = rd(U32)
    n_tensors, n_kv = rd(U64), rd(U64)
    kv = []
    for _ in range(n_kv):
        key = rd_str()
        t = rd(U32)
        e = {"key": key, "type": t}
        if t == STR:
            e["value"] = rd_str()
        elif t == ARR:
            it = rd(U32)
            cnt = rd(U64)
            e["item_type"], e["count"] = it, cnt
            e["value"] = [rd_str() if it == STR else rd(it) for _ in range(cnt)]
        else:
            e["value"] = rd(t)
        kv.append(e)
    alignment = next((e["value"] for e in kv if e["key"] == "general.alignment"), 32)
    tensors = []
    for _ in range(n_tensors):
        name = rd_str()
        nd = rd(U32)
        ne = [rd(U64) for _ in range(nd)]
        qt = rd(U32)
        off = rd(U64)
        tensors.append({"name": name, "type": qt, "ne": ne, "offset": off})
    data_offset = (pos + alignment - 1) // alignment * alignment
    for t in tensors:
        t["nbytes"] = nbytes(t["type"], t["ne"])
        a = data_offset + t["offset"]
        t["data"] = buf[a:a + t["nbytes"]]
    return {"version": version, "alignment": alignment, "data_offset": data_offset, "kv": kv, "tensors": tensors}


def describe(g, head=16):
    """the JSON-able view tests/golden/make_gguf_fixture.py stores (arrays cut to their first `head` items, tensors hashed)"""
    kv = []
    for e in g["kv"]:
        e = dict(e)
        if e["type"] == ARR:
            e["value"] = e["value"][:head]
        kv.append(e)
    ts = [{"name": t["name"], "type": t["type"], "ne": list(t["ne"]), "offset": t["offset"], "nbytes": t["nbytes"], "fnv1a": fnv1a(t["data"])}
          for t in g["tensors"]]
    return {"version": g["version"], "alignment": g["alignment"], "data_offset": g["data_offset"], "kv": kv, "tensors": ts}


def write(path, g):
    """write kv + tensors (each: name, type, ne, data) in the writer's layout; offsets are recomputed (each tensor padded to the alignment)"""
    out = bytearray()

    def wr(t, v):
        out.extend(struct.pack(_FMT[t], v))

    def wr_str(s):
        b = s.encode("utf-8")
        wr(U64, len(b))
        out.extend(b)

    alignment = g.get("alignment", 32)
    wr(U32, MAGIC); wr(U32, g.get("version", 3)); wr(U64, len(g["tensors"])); wr(U64, len(g["kv"]))
    for e in g["kv"]:
        wr_str(e["key"]); wr(U32, e["type"])
        if e["type"] == STR:
            wr_str(e["value"])
        elif e["type"] == ARR:
            wr(U32, e["item_type"]); wr(U64, len(e["value"]))
            for v in e["value"]:
                wr_str(v) if e["item_type"] == STR else wr(e["item_type"], v)
        else:
            wr(e["type"], e["value"])
    off = 0
    for t in g["tensors"]:
        wr_str(t["name"]); wr(U32, len(t["ne"]))
        for d in t["ne"]:
            wr(U64, d)
        wr(U32, t["type"]); wr(U64, off)
        off += (len(t["data"]) + alignment - 1) // alignment * alignment
    out.extend(b"\0" * (-len(out) % alignment))
    for t in g["tensors"]:
        d = bytes(np.asarray(t["data"], dtype=np.uint8))
        out.extend(d)
        out.extend(b"\0" * (-len(d) % alignment))
    with open(path, "wb") as f:
        f.write(out)


def kv_value(g, key, default=None):
    return next((e["value"] for e in g["kv"] if e["key"] == key), default)


def llama_weights(g, n_layer):
    """the W dict oracle/ref_llama.py evaluates (as read_weights builds it from the device), from the FILE's bytes"""
    by = {t["name"]: t for t in g["tensors"]}

    def mat(name):
        t = by[name]
        rows = t["ne"][1] if len(t["ne"]) > 1 else 1
        d = np.asarray(t["data"])
        return (t["type"], d.view(np.float32).reshape(rows, -1).copy() if t["type"] == 0 else d.reshape(rows, -1).copy())

    W = {}
    for il in range(n_layer):
        for nm in ("attn_norm", "attn_q", "attn_k", "attn_v", "attn_output", "ffn_norm", "ffn_gate", "ffn_up", "ffn_down"):
            W[(il, nm)] = mat(f"blk.{il}.{nm}.weight")
    W["output_norm"] = mat("output_norm.weight")
    W["output"] = mat("output.weight" if "output.weight" in by else "token_embd.weight")
    W["token_embd"] = mat("token_embd.weight")
    return W
