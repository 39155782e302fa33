"""layer_split.py — the reference's `-sm layer` split across the GPUs of one node, one process per GPU.

What is restated:
  * which layer lives on which device: the cumulative-fraction rule of llama_model::load_tensors
    (src/llama-model.cpp:1917-1972): n_layer + 1 units (the output layer is the last unit and sits on the
    last device, :1972), splits[] = cumulative normalised shares, unit il -> upper_bound(splits, il/(n_layer+1)).
  * the only exchange on the path: a point-to-point hand-off of the [n_embd, n_tokens] F32 activation at each
    of the G-1 boundaries (in the reference: scheduler -> cpy_tensor_async between backends in ONE process).
    Here each GPU has its own process, so the hand-off is an RCCL send/recv over one xGMI link
    (torch.distributed, backend "nccl" = RCCL); there is no collective on the path.

Decode is a sequential chain through the G stages, so one sequence cannot go faster with more GPUs
(SURVEY.md §8e "Honest expectation"). To keep every stage busy the driver runs G independent sequences
round-robin (pipeline ticks): at global tick t stage r runs its local step j = t - r on sequence j mod G. Per-GPU work per tick is
constant in G -> weak scaling.

The transport is abstracted so that the schedule can be exercised on CPU with gloo (tests/test_layer_split.py).
"""
from __future__ import annotations

import bisect
from dataclasses import dataclass
from typing import Callable, List, Optional


def layer_ranges(n_layer: int, n_dev: int, shares: Optional[List[float]] = None):
    """[(begin, end, has_output)] per device, following src/llama-model.cpp:1933-1972 with equal (or given) shares."""
    shares = shares or [1.0] * n_dev
    tot = float(sum(shares))
    splits, acc = [], 0.0
    for s in shares:
        acc += s
        splits.append(acc / tot)
    splits[-1] = 1.0
    n_units = n_layer + 1                      # act_gpu_layers = min(n_gpu_layers, n_layer + 1) with -ngl 99
    owner = []
    for il in range(n_units):
        dev = bisect.bisect_right(splits, il / n_units)   # std::upper_bound
        owner.append(min(dev, n_dev - 1))
    owner[n_layer] = n_dev - 1 if owner[n_layer] != n_dev - 1 else owner[n_layer]   # output layer: last device (:1972)
    out = []
    for d in range(n_dev):
        ls = [il for il in range(n_layer) if owner[il] == d]
        b, e = (ls[0], ls[-1] + 1) if ls else (0, 0)
        out.append((b, e, owner[n_layer] == d))
    return out


N_BUF = 2   # hand-off buffers per direction: step j uses buffer j % N_BUF


@dataclass
class Transport:
    """point-to-point activation hand-off between neighbouring stages, on preallocated double buffers.
    All three callables take the local step index j; buffer = j % N_BUF."""
    rank: int
    world: int
    post_recv: Callable[[int], None]     # start receiving step j's input from rank-1 (non-blocking)
    wait_recv: Callable[[int], None]     # block until step j's input has landed
    send: Callable[[int], None]          # hand step j's output to rank+1 (must first make sure buffer j % N_BUF is free again)
    flush: Callable[[], None] = lambda: None   # block until every send issued so far has been delivered


def run_steps(transport: Transport, n_steps: int, stage_fn: Callable[[int, int, bool], None], first_step: int = 0, n_seq: Optional[int] = None):
    """Every rank runs the SAME number of local steps; local step j of stage r is global pipeline tick j + r and
    works on sequence j % n_seq. Matching is by step index (send j of rank r <-> recv j of rank r+1), so the
    sequence of sends and receives is identical on both sides of every link and nothing can be left unmatched.

    stage_fn(seq, j, has_input): compute this stage for sequence `seq`, reading recv buffer j % N_BUF when
    has_input and writing send buffer j % N_BUF. The next step's receive is posted before computing, so the
    hand-off of step j+1 overlaps the compute of step j. Returns the number of stage evaluations.
    """
    r, G = transport.rank, transport.world
    n_seq = n_seq or G
    last = first_step + n_steps
    if r > 0 and n_steps > 0:
        transport.post_recv(first_step)
    for j in range(first_step, last):
        if r > 0:
            transport.wait_recv(j)
            if j + 1 < last:
                transport.post_recv(j + 1)
        stage_fn(j % n_seq, j, r > 0)
        if r < G - 1:
            transport.send(j)
    transport.flush()
    return n_steps


def run_chain_steps(transport: Transport, n_steps: int, stage_fn: Callable[[int, int, bool], None], token_done: Callable[[int], None]):
    """llama-bench's own `-sm layer` protocol (tools/llama-bench/llama-bench.cpp:1791-1810): ONE sequence, one llama_decode per token followed by
    llama_synchronize — token j + 1 does not start before token j has passed the last stage. Every rank runs n_steps local steps on sequence 0;
    token_done(j) is the per-token synchronisation: the last rank reports (a 1-element message to rank 0), rank 0 blocks on it, the ranks in
    between do nothing (they block on their next receive anyway). Tokens/s of this loop is the single-sequence chain rate: a sequential chain
    through the G stages, so it does not grow with G."""
    r, G = transport.rank, transport.world
    for j in range(n_steps):
        if r > 0:
            transport.post_recv(j)
            transport.wait_recv(j)
        stage_fn(0, j, r > 0)
        if r < G - 1:
            transport.send(j)
        token_done(j)
    transport.flush()
    return n_steps
