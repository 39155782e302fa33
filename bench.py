#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on the MI355X backend: llama-bench tg128 (+ pp512) for
Llama-3-8B Q4_K_M (BASELINE.json configs[1]) on synthetic weights of that architecture.

    python bench.py --gpus N --steps K --warmup W

A "step" is one decoded token through the whole hot path (225 quantized mat-vecs + the element ops of
llm_build_llama), driven with llama-bench's protocol (tools/llama-bench/llama-bench.cpp:1791-1810: one
llama_decode per token, synchronize after every token, tokens are random ids). Inputs (weights, KV cache)
are resident in HBM when the timed region starts.

N = 1 : the single-GPU configuration the metric is quoted on.
N > 1 : `-sm layer` split, one process per GPU, activations handed over by RCCL send/recv (layer_split.py);
        N independent sequences are kept in flight so that every stage is busy (weak scaling).

Rank 0 prints ONE JSON line: the contract fields + `roofline` (dominant kernel: the quantized mat-vec, HBM-bound,
timed live with HIP events on the backend's own stream) + `cpu_baseline` (the oracle's CPU restatement of the
same mat-vec work on the host cores, a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak (spec)
TYPE_NAMES = {2: "q4_0", 8: "q8_0", 12: "q4_K", 13: "q5_K", 14: "q6_K", 39: "mxfp4"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def perplexity_delta(be, ls, gg, ftype, gguf=None, positions=8192, seq_len=128):
    """checker leg (north_star: perplexity delta vs the CPU reference, <= 1e-3): what llama-perplexity --kl-divergence reports
    (tools/perplexity/perplexity.cpp:541-642,1743-2005) — ln PPL ratio and mean KL divergence with their standard errors, top-1 agreement — between
    this backend's logits (decode kernels, token by token) and the oracle's CPU-backend arithmetic ("cpu16": int8 activation blocks, integer
    dots, q / p rounded to f16) on the same weights. The model is made CONFIDENT (oracle/ref_llama.py: logit_parity_peaked — the lm_head scaled so
    that the reference's perplexity on text it generates itself is ~8): a random-init model scored on random tokens sits at PPL ~ n_vocab and
    barely notices logit errors. `gguf`: the same on a model read from a file (its hyper-parameters and weights; the oracle is numpy + C: sized
    for small models — a real 8B file needs hours per thousand positions). Same procedure with gates in
    tests/test_gpu_llama_graph.py::test_perplexity_delta_on_a_confident_model_32768_positions."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import ref_llama
    n_seq = max(1, positions // seq_len)
    if gguf:
        m = ls.GgufLlama(be, gguf, n_ctx=seq_len + 32); label = f"gguf file {os.path.basename(gguf)}"
    else:
        ft = ftype if ftype in ("Q4_K_M", "Q4_0", "Q8_0", "Q6_K") else "Q4_K_M"
        m = ls.SynthLlama(be, "tiny", ft, n_ctx=seq_len + 32, seed=21); label = f"tiny {ft} (synthetic weights)"
    try:
        r = ref_llama.logit_parity_peaked(m, gg, n_seq=n_seq, seq_len=seq_len)
    finally:
        m.free()
    keep = ("positions", "kl_mean", "kl_se", "delta_ln_ppl", "delta_ln_ppl_se", "top1_agree")
    def cut(e):
        return {k: (round(e[k], 8) if isinstance(e[k], float) else e[k]) for k in keep}
    b = r["backend"]
    return {"model": label, "positions": r["positions"], "logit_scale": round(r["logit_scale"], 4), "ppl_cpu_reference_on_its_own_text": round(b["ppl_base"], 3),
            "target_abs_delta_ln_ppl": 1e-3, "abs_delta_plus_2se": round(abs(b["delta_ln_ppl"]) + 2*b["delta_ln_ppl_se"], 8),
            "backend_vs_cpu_reference": cut(b), "yardstick_cpu_arithmetic_with_f32_q_p_vs_cpu_reference": cut(r["cpu"])}


def cpu_baseline(model_cfg, ftype, budget_s=20.0):
    """Time the oracle (CPU restatement, OpenMP) on the same mat-vec work: a bounded sample of layers + the lm_head."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    native = orc.build(native=True, out_dir=Path(os.environ.get("TMPDIR", "/tmp")))
    cores = orc.cpu_budget(cap=64)          # affinity mask / cgroup quota, at most 64 threads (a GPU box shares a 256-thread host)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    L = orc.lib(native)
    rng = np.random.default_rng(0)
    c = model_cfg
    assert ftype == "Q4_K_M"
    shapes = [("attn_q", c["n_embd"], c["n_embd"], orc.Q4_K), ("attn_k", c["n_embd_head"] * c["n_head_kv"], c["n_embd"], orc.Q4_K),
              ("attn_v", c["n_embd_head"] * c["n_head_kv"], c["n_embd"], orc.Q6_K), ("attn_output", c["n_embd"], c["n_embd"], orc.Q4_K),
              ("ffn_gate", c["n_ff"], c["n_embd"], orc.Q4_K), ("ffn_up", c["n_ff"], c["n_embd"], orc.Q4_K),
              ("ffn_down", c["n_embd"], c["n_ff"], orc.Q6_K)]
    n_sample_layers = 3
    layers = []
    for _ in range(n_sample_layers):
        lw = []
        for _, m, k, qt in shapes:
            bs, ts = orc.QUANT_SIZES[qt]
            w = rng.integers(0, 256, size=(m, k // bs * ts), dtype=np.uint8)     # timing only: any bytes cost the same
            lw.append((w, m, k, qt))
        layers.append(lw)
    x = {k: rng.uniform(-1, 1, size=(1, k)).astype(np.float32) for k in (c["n_embd"], c["n_ff"])}
    import ctypes
    def matvec(w, m, k, qt):
        dst = np.empty((1, m), np.float32)
        L.orc_mul_mat(qt, ctypes.c_void_p(w.ctypes.data), ctypes.c_void_p(x[k].ctypes.data), ctypes.c_void_p(dst.ctypes.data), m, 1, k, 1)
    for lw in layers[:1]:
        for w, m, k, qt in lw:
            matvec(w, m, k, qt)
    t0 = time.perf_counter(); reps = 0
    while True:
        for lw in layers:
            for w, m, k, qt in lw:
                matvec(w, m, k, qt)
        reps += 1
        if time.perf_counter() - t0 > budget_s * 0.6 or reps >= 40:
            break
    t_layer = (time.perf_counter() - t0) / (reps * n_sample_layers)
    # lm_head: 128256 x 4096 Q6_K, once
    bs, ts = orc.QUANT_SIZES[orc.Q6_K]
    w = rng.integers(0, 256, size=(c["n_vocab"], c["n_embd"] // bs * ts), dtype=np.uint8)
    matvec(w[:1024], 1024, c["n_embd"], orc.Q6_K)
    t1 = time.perf_counter(); matvec(w, c["n_vocab"], c["n_embd"], orc.Q6_K); t_head = time.perf_counter() - t1
    t_tok = t_layer * c["n_layer"] + t_head
    return {"value": round(1.0 / t_tok, 3), "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ggml_oracle.c (Q8_K activation quantize + integer vec_dot — AVX2 forms of the Q4_K / Q6_K dots, bit-identical to the "
                      f"scalar restatement — OpenMP): {n_sample_layers} distinct "
                      f"Llama-3-8B Q4_K_M layers x {reps} reps + one full lm_head mat-vec, extrapolated to {c['n_layer']} layers; "
                      f"mat-vec work only (attention/norm/rope excluded)"}



def roofline_from_profile(prof):
    """the dominant kernel's achieved HBM rate from the backend's per-dispatch timings (option "profile": each grouped mat-vec dispatch carries its own
    start/stop event pair = the interval rocprofv3's kernel trace reports)"""
    prof.sort(key=lambda e: -e["total_ms"])
    top = prof[0]
    avg_s = top["total_ms"] / top["launches"] * 1e-3
    ach = top["bytes_per_launch"] / avg_s / 1e9
    roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": None, "traffic_source": None,
            "kernel": top.get("kernel") or f"k_mmvq<{TYPE_NAMES.get(top['type'], top['type'])}> m={top['m']} k={top['k']} n={top['n']}",
            "timing": "eager launches, each dispatch's own start/stop events (hipExtLaunchKernelGGL) on the backend stream = the interval "
                      "rocprofv3 --kernel-trace reports for that kernel name",
            "bytes_per_launch": top["bytes_per_launch"], "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": top["launches"],
            "all": [{"type": TYPE_NAMES.get(e["type"], e["type"]), "m": e["m"], "k": e["k"], "n": e["n"], "launches": e["launches"],
                     "kernel": e.get("kernel", ""), "avg_us": round(e["total_ms"] / e["launches"] * 1e3, 2),
                     "GBps": round(e["bytes_per_launch"] / (e["total_ms"] / e["launches"] * 1e-3) / 1e9, 1)} for e in prof]}
    return roof, top

def main():
    # the one JSON line is the ONLY thing on stdout: libraries that write to file descriptor 1 (gloo's "[Gloo] Rank 0 is connected ..." during
    # the rendezvous, the HIP runtime) are sent to stderr, and the result is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--ftype", default="Q4_K_M")
    ap.add_argument("--ctk", default="f16", choices=["f16", "q8_0", "q4_0", "bf16"], help="llama-bench -ctk: K cache type")
    ap.add_argument("--ctv", default="f16", choices=["f16", "q8_0", "q4_0", "bf16"], help="llama-bench -ctv: V cache type (anything but f16 needs --fa 1, as in the reference)")
    ap.add_argument("--row-split", type=int, default=0, help="-sm row inside ONE process: spread the weight matrices' rows over this many devices of the "
                    "registry (csrc/backend.cpp: the split buffer type); on a one-GPU box set GGML_MI355X_VIRTUAL_DEVICES to list the GPU several times")
    ap.add_argument("--gguf", default=None, help="run the same protocol on a model read from this GGUF file (llama / gpt-oss architectures) instead of the "
                    "synthetic weights of --model / --ftype")
    ap.add_argument("--pp", type=int, default=512, help="prompt length for the extra pp measurement (0 = skip)")
    ap.add_argument("--fa", type=int, default=0, help="1 = llama-bench -fa 1: FLASH_ATTN_EXT, V cache not transposed, n_kv padded to 256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--parity", action="store_true", help="also run the checker leg: perplexity statistics against the oracle on the small synthetic model, or on "
                                                           "the --gguf model (its launches use the same kernel templates: keep it out of a run whose rocprofv3 summary is read per kernel)")
    ap.add_argument("--parity-positions", type=int, default=8192)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    import torch   # first: its HIP runtime is the one in the process
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    transport = os.environ.get("BENCH_TRANSPORT", "nccl")     # "gloo": debug path — ranks may share a GPU, hand-off staged through host memory
    dev_index = local_rank if transport == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    pg = None      # the group the activation hand-offs go through (None = the default group)
    if world > 1:
        # rendezvous, barriers and the final max-reduction over gloo (host); the hand-offs over RCCL when it comes up on every rank,
        # else over gloo through host memory — a bench line from the slower transport beats none
        dist.init_process_group("gloo")
        if transport == "nccl":
            ok = 1
            try:
                pg = dist.new_group(backend="nccl")
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe, group=pg); torch.cuda.synchronize()
                if int(probe.item()) != world:
                    ok = 0
            except Exception as e:   # noqa: BLE001
                log(f"[rank {rank}] RCCL group unavailable ({type(e).__name__}: {e}); falling back to gloo hand-offs")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                # a number measured over host memory must not appear under an N-GPU label by accident: fail, unless the caller asked for
                # the fallback (BENCH_ALLOW_GLOO_FALLBACK=1), in which case the metric string itself says so
                if os.environ.get("BENCH_ALLOW_GLOO_FALLBACK") != "1":
                    raise SystemExit("bench.py: the RCCL group did not come up on every rank (set BENCH_TRANSPORT=gloo or BENCH_ALLOW_GLOO_FALLBACK=1 to run over host memory)")
                transport = "gloo"; pg = None

    import __graft_entry__ as ge
    pkg = ge.load_package()
    gg, ls, lsp = pkg.ggml, pkg.llama_synth, pkg.layer_split

    be = gg.Backend(dev_index)
    if args.gguf and (args.ctk != "f16" or args.ctv != "f16" or args.row_split):
        raise SystemExit("bench.py: --ctk / --ctv / --row-split apply to the synthetic models only (a --gguf model runs with an f16 cache, unsplit)")
    if args.ctv != "f16" and not args.fa:
        raise SystemExit("bench.py: --ctv other than f16 requires --fa 1 (V cache quantization requires flash_attn)")
    if args.gguf:     # hyper-parameters from the file's metadata (csrc/harness/gguf_file.h); the model label follows the file
        d = ls.gguf_describe(args.gguf)
        kvs = {e["key"]: e["value"] for e in d["kv"]}
        arch = kvs["general.architecture"]
        emb = next(t for t in d["tensors"] if t["name"] == "token_embd.weight")
        cfg = dict(n_layer=int(kvs[arch + ".block_count"]), n_vocab=int(emb["ne"][1]))
        args.model, args.ftype = os.path.basename(args.gguf), f"ftype{kvs.get('general.file_type', '?')}"

        def new_model(n_ctx, **kw):
            kw.pop("has_output", None)
            return ls.GgufLlama(be, args.gguf, n_ctx=n_ctx, n_seq_max=kw.get("n_seq_max", 1), flash_attn=kw.get("flash_attn", False),
                                layer_begin=kw.get("layer_begin", 0), layer_end=kw.get("layer_end", -1))
    else:
        cfg = ls.MODELS[args.model]

        def new_model(n_ctx, **kw):
            return ls.SynthLlama(be, args.model, args.ftype, n_ctx=n_ctx, seed=1, row_split=args.row_split, type_k={"f16": 0, "q8_0": 8, "q4_0": 2, "bf16": 30}[args.ctk],
                                 type_v={"f16": 0, "q8_0": 8, "q4_0": 2, "bf16": 30}[args.ctv], **kw)
    K, W = args.steps, args.warmup
    ranges = lsp.layer_ranges(cfg["n_layer"], world)
    lb, le, has_out = ranges[rank]
    n_seq = world
    steps_per_seq = (max(K, W) + n_seq - 1) // n_seq + 1
    n_ctx = max(32, (steps_per_seq + 31) // 32 * 32) if world > 1 else max(128, (K + 31) // 32 * 32)
    t0 = time.time()
    m = new_model(n_ctx, layer_begin=lb, layer_end=le, has_output=has_out, n_seq_max=n_seq, flash_attn=bool(args.fa))
    cfg = m.cfg
    log(f"[rank {rank}] layers [{lb},{le}) output={has_out} weights {m.weight_bytes/1e9:.3f} GB, model ready in {time.time()-t0:.1f}s")
    rng = np.random.default_rng(1)   # llama-bench: std::rand() % n_vocab, default seed (tools/llama-bench/llama-bench.cpp:1798)
    tokens = rng.integers(0, cfg["n_vocab"], size=max(K, W) + 8).astype(np.int32)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = {}
    if world == 1:
        def run_tokens(n):
            for i in range(n):
                m.decode(tokens[i:i + 1], want_host=True, sync=True, view=True)     # llama_decode + llama_synchronize per token; the logits land in the pinned output buffer (llama_get_logits' view)
        # warm-up: builds the graphs for every n_kv bucket and lets the backend capture them
        done = 0
        while done < W:
            m.kv_clear(); n = min(W - done, n_ctx); run_tokens(n); done += n
        m.kv_clear()
        be.reset_counters()
        # (the interpreter's cyclic garbage collector once took 52 ms out of ONE token of a 128-token run — a full collection over torch's objects: it is a property of
        # this Python harness, not of the path measured, and stays out of the timed region; tools/tg_reps_probe.py)
        import gc
        gc.collect(); gc.disable()
        sync_all(); t0 = time.perf_counter()
        run_tokens(K)
        sync_all(); dt = time.perf_counter() - t0
        gc.enable()
        cnt = be.counters()
        tok_s = K / dt
        result.update(value=tok_s, ms_per_step=dt / K * 1e3)
        extra = {"graph_replays": cnt["graph_replays"], "kernels_per_token": None, "graph_nodes": m.graph_nodes(1),
                 "weight_GB_per_token": m.weight_bytes / 1e9,
                 "hbm_frac_whole_token": (m.weight_bytes * tok_s / 1e9) / HBM_PEAK_GBPS}
        # kernel count per token from one eager token
        be.set_option("graphs", 0); m.kv_clear(); be.reset_counters(); run_tokens(1)
        extra["kernels_per_token"] = be.counters()["kernels_launched"]
        be.set_option("graphs", 1)

        roof = None
        if not args.no_profile and args.row_split <= 1:      # (the per-device launches of a row split are not bracketed: that mode reports throughput only)
            # dominant kernel, timed live with HIP events on the backend stream. Option "profile" = 1: eager launches, each grouped mat-vec
            # dispatch carrying its own start/stop event pair (hipExtLaunchKernelGGL), i.e. kernel start -> kernel end by the packet's
            # timestamps, the same interval rocprofv3's kernel trace reports (round 1 recorded an event either side of the launch call,
            # which put the host launch gap inside the pair: 20.7 us where the trace says 14.8 us). = 2 (BENCH_PROFILE_MODE=2) captures
            # record-event nodes into the hipGraphs, but on ROCm 7.2 those cannot be read back (hipEventElapsedTime fails): falls back to 1.
            pmode = int(os.environ.get("BENCH_PROFILE_MODE", "1"))
            be.set_option("profile", pmode); m.kv_clear(); run_tokens(min(K, 32)); prof = be.profile()
            if not prof and pmode == 2:
                be.set_option("profile", 1); m.kv_clear(); run_tokens(min(K, 32)); prof = be.profile()
            be.set_option("profile", 0)
            roof, top = roofline_from_profile(prof)
            # HBM bytes per launch of that kernel from the PMC pass (a separate rocprofv3 --pmc FETCH_SIZE run, doubled as the guide's
            # gfx950 correction prescribes; tools/pmc_traffic.py) — it cannot be collected inside this run, so the committed summary
            # of the same workload is quoted (the newest round's), and only when its kernel is the dominant launch found live
            try:
                pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
                for fn in ("r04_pmc_fetch_size_summary.json", "r03_pmc_fetch_size_summary.json", "r02_pmc_fetch_size_summary.json", "r01_l_pmc_fetch_size_summary.json"):
                    pmc_file = os.path.join(pdir, fn)
                    if roof["traffic"] is not None or not (args.model == "llama3-8b" and args.ftype == "Q4_K_M" and os.path.exists(pmc_file)):
                        continue
                    for e in json.load(open(pmc_file)):
                        t = e.get("hbm_read_bytes_per_launch_corrected")
                        same = (top.get("kernel") or "k_mmvq_fused") in e["kernel"]
                        if same and t and abs(t - top["bytes_per_launch"]) <= 0.05 * top["bytes_per_launch"]:
                            roof["traffic"] = int(t)
                            roof["traffic_source"] = f"profiles/{fn} ({e['kernel']}): a separate rocprofv3 --pmc FETCH_SIZE pass of this workload, x2 per the gfx950 correction; not collected in this run"
                            break
            except Exception:
                pass
        # achievable streaming-read rate on this box, same load instruction as the kernels
        import ctypes as C
        p = gg.base().ggml_backend_reg_get_proc_address(be.reg, b"ggml_backend_mi355x_test_hbm_read_gbps")
        hbm = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_size_t, C.c_int)(p)
        extra["hbm_read_probe_GBps"] = round(max(hbm(be.be, 2 << 30, 5) for _ in range(2)), 1)
        m.free()

        if args.pp > 0:
            mp = new_model(args.pp, flash_attn=bool(args.fa))
            ptoks = rng.integers(0, cfg["n_vocab"], size=args.pp).astype(np.int32)
            mp.decode(ptoks); mp.kv_clear()                       # warm-up prompt pass (llama-bench.cpp:1949-1971)
            reps = []
            for _ in range(3):
                mp.kv_clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); mp.decode(ptoks); torch.cuda.synchronize()
                reps.append(args.pp / (time.perf_counter() - t0))
            extra[f"pp{args.pp}_tok_s"] = round(float(np.mean(reps)), 1)
            # the prompt pass against the matrix cores: 2 FLOP per token and weight element of the layers' mat-muls (MoE: the used experts), the attention
            # products on top; dense bf16 MFMA peak from MI355X_MICROARCH.md (2.5 PFLOP/s)
            hd, nh, nkv, ne, nff = cfg["n_embd_head"], cfg["n_head"], cfg["n_head_kv"], cfg["n_embd"], cfg["n_ff"]
            w_layer = ne*(nh*hd + 2*nkv*hd) + nh*hd*ne + 3*ne*nff*max(1, cfg.get("n_expert_used", 0))
            flop = 2.0*args.pp*w_layer*cfg["n_layer"] + 2.0*2.0*nh*hd*cfg["n_layer"]*args.pp*(args.pp + 1)/2
            tfs = flop*float(np.mean(reps))/args.pp/1e12
            extra[f"pp{args.pp}_mfma"] = {"TFLOP_per_pass": round(flop/1e12, 3), "achieved_TFLOPs": round(tfs, 1), "peak_TFLOPs": 2500.0, "frac": round(tfs/2500.0, 4)}
            mp.free()
        result["extra"] = extra
        result["roofline"] = roof
        if args.parity:
            try:
                extra["perplexity"] = perplexity_delta(be, ls, gg, args.ftype, gguf=args.gguf, positions=args.parity_positions)
            except Exception as e:
                extra["perplexity"] = {"failed": str(e)}
        else:
            extra["parity_skipped"] = True      # (--parity runs the checker leg; the GPU test suite holds its gates)
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, args.ftype)
            except Exception as e:   # the baseline is reporting only; never let it take the bench line down
                result["cpu_baseline"] = {"value": None, "unit": "tok/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    else:
        n_embd = cfg["n_embd"]
        buf_dev = "cuda" if transport == "nccl" else "cpu"
        recv_buf = [torch.empty(n_embd, dtype=torch.float32, device=buf_dev) for _ in range(lsp.N_BUF)]
        send_buf = [torch.empty(n_embd, dtype=torch.float32, device=buf_dev) for _ in range(lsp.N_BUF)]
        recv_work = [None] * lsp.N_BUF; send_work = [None] * lsp.N_BUF
        tcur = torch.cuda.current_stream()
        if transport == "gloo":                        # device-side staging buffers the model reads / writes
            recv_dev = [torch.empty(n_embd, dtype=torch.float32, device="cuda") for _ in range(lsp.N_BUF)]
            send_dev = [torch.empty(n_embd, dtype=torch.float32, device="cuda") for _ in range(lsp.N_BUF)]
        # BENCH_DUMP_LOGITS=<file.npz>: the last rank keeps the logits of every timed step (tests/test_gpu_layer_split.py compares them
        # with a single-process run of the same token streams)
        dump = {"on": False, "rows": []} if os.environ.get("BENCH_DUMP_LOGITS") else None

        # per-stage clock (rank 0 prints every rank's numbers): seconds this rank spent computing its layers, blocked on the hand-off from the stage before,
        # and blocked on a send buffer still in flight — summed over the timed steps only
        clock = {"on": False, "compute": 0.0, "recv_wait": 0.0, "send_wait": 0.0}

        def post_recv(j):
            recv_work[j % lsp.N_BUF] = dist.irecv(recv_buf[j % lsp.N_BUF], src=rank - 1, group=pg)
        def wait_recv(j):
            t = time.perf_counter()
            recv_work[j % lsp.N_BUF].wait(); recv_work[j % lsp.N_BUF] = None; tcur.synchronize()
            if clock["on"]: clock["recv_wait"] += time.perf_counter() - t
        def send(j):
            b = j % lsp.N_BUF
            send_work[b] = dist.isend(send_buf[b], dst=rank + 1, group=pg)
        def flush():
            for i, w in enumerate(send_work):
                if w is not None:
                    w.wait(); send_work[i] = None      # a Work is waited exactly once (gloo blocks on a second wait)
            tcur.synchronize()
        tr = lsp.Transport(rank, world, post_recv, wait_recv, send, flush)
        pos = [0] * n_seq

        def stage(seq, j, has_input):
            b = j % lsp.N_BUF
            t_in = time.perf_counter()
            if send_work[b] is not None:          # buffer reuse: the send issued N_BUF steps ago must be done
                send_work[b].wait(); tcur.synchronize(); send_work[b] = None
            t_c = time.perf_counter()
            if has_input and transport == "gloo":      # debug transport: the hand-off was received into host memory
                recv_dev[b].copy_(recv_buf[b]); tcur.synchronize()
            out = m.decode(tokens[j % len(tokens):j % len(tokens) + 1] if not has_input else None, n_tokens=1, seq=seq,
                           dev_act_in=(recv_dev[b] if transport == "gloo" else recv_buf[b]).data_ptr() if has_input else None,
                           dev_result_out=(send_dev[b] if transport == "gloo" else send_buf[b]).data_ptr() if rank < world - 1 else None,
                           want_host=has_out, sync=True)
            if rank < world - 1 and transport == "gloo":
                send_buf[b].copy_(send_dev[b]); tcur.synchronize()
            if clock["on"]:
                clock["send_wait"] += t_c - t_in; clock["compute"] += time.perf_counter() - t_c
            if dump is not None and has_out and dump["on"]:
                dump["rows"].append((seq, pos[seq], out.copy()))
            pos[seq] += 1
        log(f"[rank {rank}] warm-up: {W} pipeline steps")
        lsp.run_steps(tr, W, stage, 0, n_seq)
        m.kv_clear()
        log(f"[rank {rank}] warm-up done")
        if os.environ.get("BENCH_DEBUG"):
            import faulthandler; faulthandler.dump_traceback_later(int(os.environ["BENCH_DEBUG"]), exit=True)
        for i in range(n_seq):
            pos[i] = 0
        if dump is not None:
            dump["on"] = True
        sync_all(); t0 = time.perf_counter()
        log(f"[rank {rank}] timed region: {K} pipeline steps")
        clock["on"] = True
        lsp.run_steps(tr, K, stage, 0, n_seq)
        clock["on"] = False
        log(f"[rank {rank}] timed steps done")
        sync_all(); dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64)          # default group = gloo
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        result.update(value=K / dt, ms_per_step=dt / K * 1e3)
        result["extra"] = {"layers": [list(r) for r in ranges], "sequences_in_flight": n_seq, "handoff_bytes": n_embd * 4}
        # hand-off latency, measured after the timed region: 32 ping-pongs of one hand-off message over every link in turn (rank r <-> r + 1), one way = half
        # the round trip, through the same transport and buffers the pipeline uses
        lat = [0.0] * world
        for link in range(world - 1):
            dist.barrier()
            if rank in (link, link + 1):
                peer = link + 1 if rank == link else link
                for it in range(40):
                    if it == 8: tcur.synchronize(); tl = time.perf_counter()
                    if rank == link:
                        dist.send(send_buf[0], dst=peer, group=pg); dist.recv(recv_buf[0], src=peer, group=pg)
                    else:
                        dist.recv(recv_buf[0], src=peer, group=pg); dist.send(send_buf[0], dst=peer, group=pg)
                    tcur.synchronize()
                if rank == link: lat[link] = (time.perf_counter() - tl) / 32 / 2 * 1e6
        per_rank = torch.tensor([[clock["compute"], clock["recv_wait"], clock["send_wait"]] + lat], dtype=torch.float64)
        gathered = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)                     # default group = gloo
        result["extra"]["stages"] = [{"rank": r, "layers": list(ranges[r][:2]), "output_layer": bool(ranges[r][2]),
                                      "compute_ms_per_step": round(float(gathered[r][0, 0]) / K * 1e3, 4),
                                      "recv_wait_ms_per_step": round(float(gathered[r][0, 1]) / K * 1e3, 4),
                                      "send_buffer_wait_ms_per_step": round(float(gathered[r][0, 2]) / K * 1e3, 4),
                                      "handoff_to_next_one_way_us": round(float(gathered[r][0, 3 + r]), 1) if r < world - 1 else None} for r in range(world)]
        result["extra"]["stages_note"] = ("host clocks around the synchronised stage call and around the blocking waits, timed steps only; a step's wall time = compute + "
                                          "recv_wait + send_buffer_wait + loop overhead; the hand-off latency is a ping-pong of one n_embd x f32 message after the timed region")
        # llama-bench's own -sm layer number: ONE sequence, a token does not start before the previous one is through the last stage
        # (tools/llama-bench/llama-bench.cpp:1791-1810). Untimed by the contract's `value` (whole-job aggregate); reported beside it.
        if dump is not None:
            dump["on"] = False
        ack = torch.zeros(1, dtype=torch.float32, device=buf_dev)
        def token_done(j):
            if rank == world - 1:
                dist.send(ack, dst=0, group=pg)
            elif rank == 0:
                dist.recv(ack, src=world - 1, group=pg)
            tcur.synchronize()
        m.kv_clear()
        for i in range(n_seq):
            pos[i] = 0
        kc = min(K, n_ctx - 1)
        lsp.run_chain_steps(tr, min(8, kc), stage, token_done)      # warm-up of the one-sequence graphs
        m.kv_clear(); pos[0] = 0
        sync_all(); t0 = time.perf_counter()
        lsp.run_chain_steps(tr, kc, stage, token_done)
        sync_all(); dtc = time.perf_counter() - t0
        tmax = torch.tensor([dtc], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        result["extra"]["single_sequence_chain_tok_s"] = round(kc / float(tmax.item()), 2)
        result["extra"]["single_sequence_chain_note"] = "llama-bench's -sm layer protocol: one sequence, per-token synchronisation through all stages; does not scale with the GPU count by construction"
        # the dominant kernel's roofline, from rank 0's own stage (the same kernel on every rank: each streams its layers' weights from its own HBM)
        result["roofline"] = None
        if rank == 0 and not args.no_profile:
            try:
                be.set_option("profile", 1); m.kv_clear(); pos[0] = 0
                for j in range(min(kc, 16)):
                    stage(0, j, False)
                flush()
                prof = be.profile(); be.set_option("profile", 0)
                if prof:
                    roof, _ = roofline_from_profile(prof)
                    roof["note"] = "rank 0's stage only, timed after the throughput measurement"
                    result["roofline"] = roof
            except Exception as e:   # noqa: BLE001 — reporting only
                log(f"[rank 0] roofline leg failed: {e}")
        dist.barrier()
        if dump is not None and has_out:
            np.savez(os.environ["BENCH_DUMP_LOGITS"], seq=np.array([r[0] for r in dump["rows"]]), pos=np.array([r[1] for r in dump["rows"]]),
                     logits=np.stack([r[2] for r in dump["rows"]]), tokens=tokens, n_seq=n_seq)
        m.free()

    if rank == 0:
        out = {
            "metric": (f"llama-bench tg{K} tok/s, Llama-3-8B Q4_K_M" if (args.model, args.ftype) == ("llama3-8b", "Q4_K_M") else f"llama-bench tg{K} tok/s, {args.model} {args.ftype}")
                      + (" [hand-offs over gloo / host memory, NOT RCCL]" if world > 1 and transport != "nccl" else ""),
            "value": round(result["value"], 2), "unit": "tok/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(result["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 dot (4-6 bit weights x Q8 activations), f32 accumulate", "data": f"gguf file {args.model}" if args.gguf else "synthetic",
            "config": {"workload": f"{args.model} {args.ftype}, llama-bench tg{K} protocol (BASELINE.json configs[1]): 1 token/step, sync per token, "
                                   f"{'f16 KV cache' if args.ctk == 'f16' and args.ctv == 'f16' else args.ctk + ' K / ' + args.ctv + ' V cache'}, {'flash-attn' if args.fa else 'no flash-attn'}, n_ctx={n_ctx}",
                       "parallelism": (f"rows of the weight matrices split over {args.row_split} devices in one process" if args.row_split > 1 else "single GPU") if world == 1 else f"layer split over {world} GPUs, {world} sequences in flight, {'RCCL' if transport == 'nccl' else 'gloo (host memory)'} p2p hand-off"},
            "roofline": result.get("roofline"), "cpu_baseline": result.get("cpu_baseline"),
        }
        out.update(result.get("extra", {}))
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    be.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
