"""mega_check.py — the persistent decode kernel against the launch-per-phase path on the same model and tokens (logits bit for bit / max diff),
and tokens per second of both. Usage: python tools/mega_check.py [model] [ftype] [n_tokens]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg

pkg = graft_pkg.load()
gg, ls = pkg.ggml, pkg.llama_synth
model = sys.argv[1] if len(sys.argv) > 1 else "tiny-hd128"
ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
n_tok = int(sys.argv[3]) if len(sys.argv) > 3 else 12
be = gg.Backend(0)
m = ls.SynthLlama(be, model, ftype, n_ctx=max(32, (n_tok + 31)//32*32), seed=1)
toks = np.random.default_rng(3).integers(0, m.cfg["n_vocab"], size=n_tok).astype(np.int32)
out = {}
for mode in (0, 1):
    be.set_option("mega", mode)
    m.kv_clear()
    res = []
    for rep in range(2):        # second pass: captured graphs
        m.kv_clear()
        t0 = time.perf_counter()
        res = [m.decode(toks[i:i + 1]).copy() for i in range(n_tok)]
        dt = time.perf_counter() - t0
    out[mode] = np.stack(res)
    print(f"mega={mode}: {n_tok/dt:.1f} tok/s, kernels/token {be.counters()['kernels_launched'] if hasattr(be, 'counters') else '?'}", flush=True)
d = np.abs(out[0] - out[1])
bad = d > 1e-3*np.abs(out[0]).max()
print("tokens with differences:", [int(i) for i in np.nonzero(bad.any(axis=1))[0]], "fraction of logits off per token:", [round(float(b.mean()), 3) for b in bad])
print("finite:", np.isfinite(out[1]).all(), "max|diff|", float(d.max()), "max|ref|", float(np.abs(out[0]).max()), "bitwise equal:", bool((out[0] == out[1]).all()))
m.free(); be.free()
