"""op_perf.py — the reference's own MUL_MAT perf cases (tests/test-backend-ops.cpp:6190-6196: type_a x F32, m = 4096, k = 14336,
n in {1, 2, 3, 4, 5, 8, 512} — a Llama-3-8B ffn_down) plus the MoE cases (:6226-6229 shapes: MUL_MAT_ID 2880 x 2880, 32 experts top-4,
and Mixtral's 8 experts top-2) through the backend's C-ABI. n <= 8 reports GB/s of weight bytes (memory-bound), n = 512 TFLOP/s.
Wall clock over back-to-back graph computes (each case is its own one-node graph, as in test-backend-ops' perf mode)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle as orc
from gpu_util import QTYPES, backend, gg

L = gg.base(); be = backend()
out = []


def timed(g, reps):
    for _ in range(3):
        be.compute_async(g)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        be.compute_async(g)
    be.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for name in QTYPES:
    m, k = 4096, 14336
    rng = np.random.default_rng(1234)
    wb = orc.random_blocks(rng, QTYPES[name], (m,), k)
    for n in (1, 2, 3, 4, 5, 8, 512):
        with gg.Context() as ctx:
            w = ctx.new_tensor(QTYPES[name], [k, m]); b = ctx.new_tensor(gg.F32, [k, n])
            o = L.ggml_mul_mat(ctx.ctx, w, b)
            ctx.alloc(be); gg.tensor_set(w, wb); gg.tensor_set(b, rng.uniform(-1, 1, size=(n, k)).astype(np.float32))
            us = timed(gg.graph_of(ctx, o), 50 if n <= 8 else 20)
        e = {"op": "MUL_MAT", "type_a": name, "m": m, "k": k, "n": n, "us": round(us, 2)}
        if n <= 8:
            e["GBps_weights"] = round(wb.nbytes / us / 1e3, 1)
        else:
            e["TFLOPs"] = round(2.0 * m * k * n / us / 1e6, 1)
        out.append(e); print(e, flush=True)

for name, (k, m, n_exp, n_used) in [("mxfp4", (2880, 2880, 32, 4)), ("q4_K", (4096, 14336, 8, 2)), ("q4_K", (14336, 4096, 8, 2))]:
    rng = np.random.default_rng(99)
    wb = orc.random_blocks(rng, QTYPES[name], (n_exp, m), k)
    for n in (1, 512):
        ids_full = np.stack([rng.permutation(n_exp) for _ in range(n)]).astype(np.int32)
        with gg.Context() as ctx:
            as_ = ctx.new_tensor(QTYPES[name], (k, m, n_exp)); ids = ctx.new_tensor(gg.I32, (n_exp, n)); b = ctx.new_tensor(gg.F32, (k, 1, n))
            idv = L.ggml_view_2d(ctx.ctx, ids, n_used, n, n_exp * 4, 0)
            o = L.ggml_mul_mat_id(ctx.ctx, as_, b, idv)
            ctx.alloc(be); gg.tensor_set(as_, wb); gg.tensor_set(ids, ids_full.reshape(1, 1, n, n_exp))
            gg.tensor_set(b, rng.uniform(-1, 1, size=(1, n, 1, k)).astype(np.float32))
            us = timed(gg.graph_of(ctx, o), 50 if n == 1 else 10)
        e = {"op": "MUL_MAT_ID", "type_a": name, "m": m, "k": k, "n_expert": n_exp, "n_used": n_used, "n": n, "us": round(us, 2)}
        used_bytes = wb.nbytes / n_exp * (n_used if n == 1 else n_exp)
        if n == 1:
            e["GBps_weights"] = round(used_bytes / us / 1e3, 1)
        else:
            e["TFLOPs"] = round(2.0 * m * k * n * n_used / us / 1e6, 1)
        out.append(e); print(e, flush=True)
# FLASH_ATTN_EXT perf cases (tests/test-backend-ops.cpp:6210-6216): head size 64 / 128, 8 KV heads, nr = 1 / 4 query heads per KV head, kv = 4096 / 8192 / 16384
# cached cells, ONE token, f16 K / V, f16 mask; GB/s = K + V bytes read once per launch
for kv in (4096, 8192, 16384):
    for hs in (64, 128):
        for nr in (1, 4):
            rng = np.random.default_rng(5)
            n_head_kv, n_head, T = 8, 8*nr, 1
            kc = rng.uniform(-1, 1, size=(1, 1, kv, hs*n_head_kv)).astype(np.float16); vc = rng.uniform(-1, 1, size=(1, 1, kv, hs*n_head_kv)).astype(np.float16)
            with gg.Context() as ctx:
                k_l = ctx.new_tensor(gg.F16, (hs*n_head_kv, kv)); v_l = ctx.new_tensor(gg.F16, (hs*n_head_kv, kv))
                q_cur = ctx.new_tensor(gg.F32, (hs, n_head, T)); m_ = ctx.new_tensor(gg.F16, (kv, 64))
                k = L.ggml_view_3d(ctx.ctx, k_l, hs, n_head_kv, kv, hs*2, hs*n_head_kv*2, 0); v = L.ggml_view_3d(ctx.ctx, v_l, hs, n_head_kv, kv, hs*2, hs*n_head_kv*2, 0)
                q = L.ggml_permute(ctx.ctx, q_cur, 0, 2, 1, 3); k = L.ggml_permute(ctx.ctx, k, 0, 2, 1, 3); v = L.ggml_permute(ctx.ctx, v, 0, 2, 1, 3)
                fa = L.ggml_flash_attn_ext(ctx.ctx, q, k, v, m_, float(1.0/np.sqrt(hs)), 0.0, 0.0)
                L.ggml_flash_attn_ext_set_prec(fa, 10)
                assert be.supports_op(fa); ctx.alloc(be)
                gg.tensor_set(k_l, kc); gg.tensor_set(v_l, vc); gg.tensor_set(q_cur, rng.uniform(-1, 1, size=(1, T, n_head, hs)).astype(np.float32))
                gg.tensor_set(m_, np.zeros((1, 1, 64, kv), np.float16))
                us = timed(gg.graph_of(ctx, fa), 50)
            e = {"op": "FLASH_ATTN_EXT", "hs": hs, "n_head_kv": n_head_kv, "nr": nr, "kv": kv, "n_tokens": 1, "type_kv": "f16", "us": round(us, 2),
                 "GBps_kv": round((kc.nbytes + vc.nbytes)/us/1e3, 1)}
            out.append(e); print(e, flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/op_perf.json", "w"), indent=1)
