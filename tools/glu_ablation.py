"""glu_ablation.py — the prefill gate/up/SwiGLU kernel (mmq.hip k_mmq<.., 256, DUAL>) on Llama-3-8B's shape (14336 x 4096, 512 tokens) with parts switched off
(GGML_MI355X_MMQ_DBG, read once per process: run once per setting): 1 = no weight decode, 2 = no MFMAs, 4 = no global loads in the loop, 8 = no LDS commits.
Needs the diagnostic build (the switches are runtime branches that cost the kernel 10 %):  MI355X_BUILD_VARIANT=mmqdbg MI_EXTRA_HIPFLAGS=-DMI_MMQ_DBG python llama.cpp-gfx906_amd/build.py
and MI355X_BUILD_VARIANT=mmqdbg when running. Round 3, Q4_K (µs per op incl. the activation copy pass): all on 203.6; no decode 165.4; no MFMAs 141.2; no loads 155.3; no commits 153.7;
no decode + no MFMAs 122.2; no MFMAs + no loads 99.4; everything off 50.7 — the parts ADD UP (62 + 48 + 50 + 38 over a 51 us floor) instead of overlapping."""
import os, sys, time
import numpy as np
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(_root, _d))
import oracle as orc
from gpu_util import QTYPES, backend, gg
L = gg.base(); be = backend(); be.set_option("fusion", 1)
k, ff, n = 4096, 14336, 512
rng = np.random.default_rng(0)
with gg.Context() as ctx:
    g_ = ctx.new_tensor(QTYPES["q4_K"], (k, ff)); u_ = ctx.new_tensor(QTYPES["q4_K"], (k, ff)); x = ctx.new_tensor(gg.F32, (k, n))
    act = L.ggml_swiglu_split(ctx.ctx, L.ggml_mul_mat(ctx.ctx, g_, x), L.ggml_mul_mat(ctx.ctx, u_, x))
    ctx.alloc(be)
    gg.tensor_set(g_, orc.random_blocks(rng, QTYPES["q4_K"], (ff,), k)); gg.tensor_set(u_, orc.random_blocks(rng, QTYPES["q4_K"], (ff,), k))
    if os.environ.get('ZERO_DATA'):      # the same instruction stream on all-zero operands: what the chip's clock does under less switching activity (DVFS check)
        gg.tensor_set(g_, np.zeros_like(orc.random_blocks(rng, QTYPES['q4_K'], (ff,), k))); gg.tensor_set(u_, np.zeros_like(orc.random_blocks(rng, QTYPES['q4_K'], (ff,), k)))
    gg.tensor_set(x, rng.uniform(-1, 1, size=(n, k)).astype(np.float32) if not os.environ.get('ZERO_DATA') else np.zeros((n, k), np.float32))
    gr = gg.graph_of(ctx, act)
    for _ in range(3): be.compute(gr)
    be.synchronize(); t0 = time.perf_counter()
    for _ in range(20): be.compute_async(gr)
    be.synchronize()
    us = (time.perf_counter() - t0)/20*1e6
print(f"MMQ_DBG={os.environ.get('GGML_MI355X_MMQ_DBG', '0')}: {us:7.1f} us per gate/up/SwiGLU (incl. the bf16 activation copy pass)  {2*2.0*ff*k*n/us/1e6:6.1f} TFLOP/s")
