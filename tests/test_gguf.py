"""The GGUF container: the C++ reader the harness loads models with (csrc/harness/gguf_file.h) and the oracle's numpy
reader / writer (oracle/gguf_ref.py), both against a file written by the REFERENCE's writer and described by the REFERENCE's reader
(tests/golden/make_gguf_fixture.py -> tiny_llama_q4_k_m.gguf + .json). No GPU: the reader is host code."""
import json

import numpy as np
import pytest

import graft_pkg
import gguf_ref

pkg = graft_pkg.load()
ls = pkg.llama_synth
NAME = "tiny_llama_q4_k_m.gguf"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.loads((golden_dir / (NAME + ".json")).read_text())


def test_cpp_reader_matches_reference_reader(golden_dir, gold):
    assert ls.gguf_describe(golden_dir / NAME) == gold


def test_oracle_reader_matches_reference_reader(golden_dir, gold):
    assert gguf_ref.describe(gguf_ref.read(golden_dir / NAME)) == gold


def test_oracle_writer_reproduces_reference_writer(golden_dir, tmp_path):
    g = gguf_ref.read(golden_dir / NAME)
    gguf_ref.write(tmp_path / "copy.gguf", g)
    assert (tmp_path / "copy.gguf").read_bytes() == (golden_dir / NAME).read_bytes()


def test_every_value_type_occurs(gold):
    seen = {e["type"] for e in gold["kv"]} | {e.get("item_type") for e in gold["kv"] if e["type"] == gguf_ref.ARR}
    assert set(range(13)) <= seen


@pytest.mark.parametrize("cut", [0, 3, 4, 8, 23, 24, 40, 100, 1000, 5000, 8383])
def test_truncated_file_is_an_error_not_a_crash(golden_dir, tmp_path, cut):
    raw = (golden_dir / NAME).read_bytes()
    p = tmp_path / "cut.gguf"
    p.write_bytes(raw[:cut])
    with pytest.raises(RuntimeError):
        ls.gguf_describe(p)


def test_data_section_cut_short_is_an_error(golden_dir, tmp_path):
    raw = (golden_dir / NAME).read_bytes()
    p = tmp_path / "cut.gguf"
    p.write_bytes(raw[:len(raw) - 1000])
    with pytest.raises(RuntimeError, match="outside the file"):
        ls.gguf_describe(p)


def test_corrupt_headers_are_errors(golden_dir, tmp_path):
    raw = bytearray((golden_dir / NAME).read_bytes())
    p = tmp_path / "bad.gguf"
    for off, val, what in ((0, b"GGML", "magic"), (4, (9).to_bytes(4, "little"), "version"), (8, (2**60).to_bytes(8, "little"), "counts"),
                           (24, (2**62).to_bytes(8, "little"), "string length")):
        b = bytearray(raw); b[off:off + len(val)] = val
        p.write_bytes(b)
        with pytest.raises(RuntimeError, match=what):
            ls.gguf_describe(p)


def test_alignment_key_and_v2_are_honoured(golden_dir, tmp_path):
    g = gguf_ref.read(golden_dir / NAME)
    g["kv"].append({"key": "general.alignment", "type": gguf_ref.U32, "value": 256})
    g["alignment"] = 256; g["version"] = 2
    gguf_ref.write(tmp_path / "a256.gguf", g)
    d = ls.gguf_describe(tmp_path / "a256.gguf")
    assert d["alignment"] == 256 and d["version"] == 2 and d["data_offset"] % 256 == 0
    assert all(t["offset"] % 256 == 0 for t in d["tensors"])
    assert d == gguf_ref.describe(gguf_ref.read(tmp_path / "a256.gguf"))
    assert [t["fnv1a"] for t in d["tensors"]] == [gguf_ref.fnv1a(t["data"]) for t in g["tensors"]]
