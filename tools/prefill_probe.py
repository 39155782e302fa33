"""prefill_probe.py — run the pp<N> prompt pass a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/prefill_probe.py`)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg

pkg = graft_pkg.load()
gg, ls = pkg.ggml, pkg.llama_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = sys.argv[2] if len(sys.argv) > 2 else "llama3-8b"
ftype = sys.argv[3] if len(sys.argv) > 3 else "Q4_K_M"
be = gg.Backend(0)
m = ls.SynthLlama(be, model, ftype, n_ctx=n, seed=1)
toks = np.random.default_rng(0).integers(0, ls.MODELS[model]["n_vocab"], size=n).astype(np.int32)
m.decode(toks); m.kv_clear()
for _ in range(4):
    m.kv_clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); m.decode(toks); torch.cuda.synchronize()
    print(f"pp{n}: {n/(time.perf_counter()-t0):.0f} tok/s", flush=True)
m.free()
