"""The N>1 path on CPU: the reference's layer->device rule and the pipelined hand-off schedule, exercised with
world_size 2 (and 3) gloo process groups. Each "stage" is a deterministic affine map, so the final outputs can be
compared with a single-process evaluation of the whole chain."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import graft_pkg  # noqa: E402

lsp = graft_pkg.load().layer_split


def test_layer_ranges_follow_reference_rule():
    # src/llama-model.cpp:1933-1972: 33 units (32 layers + output) over 8 equal devices
    r = lsp.layer_ranges(32, 8)
    assert r[0] == (0, 5, False) and r[1] == (5, 9, False) and r[7] == (29, 32, True)
    assert sum(e - b for b, e, _ in r) == 32 and [o for _, _, o in r].count(True) == 1
    for n_layer, n_dev in ((32, 1), (32, 2), (32, 4), (80, 8), (6, 4)):
        rr = lsp.layer_ranges(n_layer, n_dev)
        cover = [il for b, e, _ in rr for il in range(b, e)]
        assert cover == list(range(n_layer)) and rr[-1][2]
    # -ts style uneven shares
    rr = lsp.layer_ranges(32, 2, [3.0, 1.0])
    assert rr[0][1] - rr[0][0] > rr[1][1] - rr[1][0]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _stage_params(rank):
    rng = np.random.default_rng(100 + rank)
    return rng.uniform(0.5, 1.5, 8).astype(np.float32), rng.uniform(-1, 1, 8).astype(np.float32)


def _worker(rank, world, port, n_steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lsp_ = graft_pkg.load().layer_split
    a, b = _stage_params(rank)
    recv = [torch.zeros(8) for _ in range(lsp_.N_BUF)]; send = [torch.zeros(8) for _ in range(lsp_.N_BUF)]
    rw = [None] * lsp_.N_BUF; sw = [None] * lsp_.N_BUF
    outs = []
    n_seq = world
    state = [0] * n_seq                                   # per-sequence step counter (stands in for the KV position)

    def post_recv(j): rw[j % lsp_.N_BUF] = dist.irecv(recv[j % lsp_.N_BUF], src=rank - 1)
    def wait_recv(j): rw[j % lsp_.N_BUF].wait()
    def send_(j): sw[j % lsp_.N_BUF] = dist.isend(send[j % lsp_.N_BUF], dst=rank + 1)
    def flush():
        for w in sw:
            if w is not None: w.wait()

    def stage(seq, j, has_input):
        bi = j % lsp_.N_BUF
        if sw[bi] is not None:
            sw[bi].wait(); sw[bi] = None
        x = recv[bi].numpy().copy() if has_input else np.full(8, float(seq * 1000 + state[seq]), np.float32)
        y = a * x + b + np.float32(state[seq])            # depends on the sequence's own history length, like a KV cache
        state[seq] += 1
        if rank < world - 1:
            send[bi].copy_(torch.from_numpy(y))
        else:
            outs.append((seq, state[seq] - 1, y.copy()))
    done = lsp_.run_steps(lsp_.Transport(rank, world, post_recv, wait_recv, send_, flush), n_steps, stage, 0, n_seq)
    dist.barrier()
    q.put((rank, done, outs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_handoff_matches_single_process_chain(world):
    n_steps = 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    res = {r: (d, o) for r, d, o in res}
    assert all(res[r][0] == n_steps for r in range(world))           # every rank ran the same number of steps
    outs = res[world - 1][1]
    assert len(outs) == n_steps
    # single-process reference: sequence j % world, its own step counter, through every stage in order
    state = [[0] * world for _ in range(world)]
    for j, (seq, step, y) in enumerate(outs):
        assert seq == j % world and step == j // world
        x = np.full(8, float(seq * 1000 + state[0][seq]), np.float32)
        for r in range(world):
            a, b = _stage_params(r)
            x = a * x + b + np.float32(state[r][seq])
            state[r][seq] += 1
        assert np.array_equal(x, y), (j, seq)


def _chain_worker(rank, world, port, n_steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lsp_ = graft_pkg.load().layer_split
    a, b = _stage_params(rank)
    recv = [torch.zeros(8) for _ in range(lsp_.N_BUF)]; send = [torch.zeros(8) for _ in range(lsp_.N_BUF)]
    rw = [None] * lsp_.N_BUF; sw = [None] * lsp_.N_BUF
    outs, order = [], []
    state = [0]
    ack = torch.zeros(1)

    def post_recv(j): rw[j % lsp_.N_BUF] = dist.irecv(recv[j % lsp_.N_BUF], src=rank - 1)
    def wait_recv(j): rw[j % lsp_.N_BUF].wait()
    def send_(j): sw[j % lsp_.N_BUF] = dist.isend(send[j % lsp_.N_BUF], dst=rank + 1)
    def flush():
        for w in sw:
            if w is not None: w.wait()

    def stage(seq, j, has_input):
        bi = j % lsp_.N_BUF
        if sw[bi] is not None:
            sw[bi].wait(); sw[bi] = None
        x = recv[bi].numpy().copy() if has_input else np.full(8, float(state[0]), np.float32)
        y = a * x + b + np.float32(state[0])
        state[0] += 1
        order.append(("stage", j))
        if rank < world - 1:
            send[bi].copy_(torch.from_numpy(y))
        else:
            outs.append(y.copy())

    def token_done(j):
        if rank == world - 1:
            ack[0] = float(j); dist.send(ack, dst=0)
        elif rank == 0:
            dist.recv(ack, src=world - 1)
            assert int(ack[0]) == j                       # token j is through the last stage before rank 0 starts token j + 1
            order.append(("done", j))
    done = lsp_.run_chain_steps(lsp_.Transport(rank, world, post_recv, wait_recv, send_, flush), n_steps, stage, token_done)
    dist.barrier()
    q.put((rank, done, outs, order))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_single_sequence_chain_is_token_by_token(world):
    """llama-bench's own -sm layer protocol (tools/llama-bench/llama-bench.cpp:1791-1810; bench.py `single_sequence_chain_tok_s`): one sequence,
    rank 0 does not start token j + 1 before the last stage has finished token j; the results equal the single-process chain."""
    n_steps = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, world, port, n_steps, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    res = {r: (d, o, od) for r, d, o, od in res}
    assert all(res[r][0] == n_steps for r in range(world))
    assert res[0][2] == [e for j in range(n_steps) for e in (("stage", j), ("done", j))]      # strictly alternating on rank 0
    outs = res[world - 1][1]
    assert len(outs) == n_steps
    for j, y in enumerate(outs):
        x = np.full(8, float(j), np.float32)
        for r in range(world):
            a, b = _stage_params(r)
            x = a * x + b + np.float32(j)
        assert np.array_equal(x, y), j
