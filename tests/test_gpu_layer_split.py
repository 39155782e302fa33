"""The layer split (`-sm layer`) with a REAL model across two processes: `bench.py --gpus 2` launched the way the driver launches
it (torch.distributed.run, one process per rank), hand-offs over gloo through host memory (BENCH_TRANSPORT=gloo: both ranks share the one GPU
of this box), a 4-layer synthetic model split by the reference's cumulative-fraction rule (src/llama-model.cpp:1917-1972). The logits the last
rank produces for every (sequence, position) must equal a single-process run of the same token streams on the whole model."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle as orc
from gpu_util import backend, pkg

pytestmark = pytest.mark.gpu
ls, lsp = pkg.llama_synth, pkg.layer_split
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ftype", ["Q4_K_M"])
def test_two_rank_layer_split_equals_single_process(ftype, tmp_path):
    K, W, G = 12, 4, 2
    dump = tmp_path / "logits.npz"
    env = dict(os.environ, BENCH_TRANSPORT="gloo", BENCH_DUMP_LOGITS=str(dump), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={G}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(G), "--steps", str(K), "--warmup", str(W),
           "--model", "tiny4", "--ftype", ftype, "--no-cpu-baseline", "--no-profile", "--pp", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == G and line["steps"] == K and line["value"] > 0 and line["scaling"] == "weak"
    ranges = lsp.layer_ranges(ls.MODELS["tiny4"]["n_layer"], G)
    assert line["layers"] == [list(x) for x in ranges] and ranges[0][1] > 0 and ranges[1][1] > ranges[1][0], "both ranks must own layers"
    # rank 0 reports every stage's clock: compute + waits account for the step time, and every link has a measured hand-off latency
    st = line["stages"]
    assert [e["rank"] for e in st] == list(range(G)) and [e["layers"] for e in st] == [list(x[:2]) for x in ranges]
    for e in st:
        assert e["compute_ms_per_step"] > 0 and e["recv_wait_ms_per_step"] >= 0 and e["send_buffer_wait_ms_per_step"] >= 0
        assert e["compute_ms_per_step"] + e["recv_wait_ms_per_step"] + e["send_buffer_wait_ms_per_step"] <= line["ms_per_step"]*1.05
        assert (e["handoff_to_next_one_way_us"] is None) == (e["rank"] == G - 1) and (e["rank"] == G - 1 or e["handoff_to_next_one_way_us"] > 0)
    assert st[0]["recv_wait_ms_per_step"] == 0
    d = np.load(dump)
    assert len(d["seq"]) == K, "every rank runs K local steps (local step j of stage r is pipeline tick j + r)"
    # single process, whole model, the same streams: sequence s is fed tokens[j] at the ticks j = s, s + G, ...
    be = backend(); be.set_option("graphs", 1); be.set_option("fusion", 1)
    m = ls.SynthLlama(be, "tiny4", ftype, n_ctx=64, seed=1)
    try:
        want = {}
        for s in range(G):
            m.kv_clear()
            for p, j in enumerate(range(s, K, G)):
                want[(s, p)] = m.decode([int(d["tokens"][j % len(d["tokens"])])])
    finally:
        m.free()
    for s, p, got in zip(d["seq"], d["pos"], d["logits"]):
        e = orc.nmse(want[(int(s), int(p))], got)
        assert e <= 1e-9, (int(s), int(p), e)
