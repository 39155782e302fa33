"""tg_reps_probe.py — the same 128-token decode loop (llama-bench protocol) repeated in one process: per-repetition tok/s, to tell a cold GPU / first-pass effect from the steady rate."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg
pkg = graft_pkg.load(); gg, ls = pkg.ggml, pkg.llama_synth
be = gg.Backend(0)
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = ls.SynthLlama(be, "llama3-8b", "Q4_K_M", n_ctx=n_ctx, seed=1)
rng = np.random.default_rng(1)
tokens = rng.integers(0, 128256, size=256).astype(np.int32)
import gc
if os.environ.get('NOGC'): gc.disable()
for rep in range(6):
    m.kv_clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = []; slow = []
    for i in range(128):
        ta = time.perf_counter(); m.decode(tokens[i:i + 1], want_host=True, sync=True, view=True); tb = time.perf_counter()
        if tb - ta > 2.5e-3: slow.append((i, round((tb - ta)*1e3, 2)))
        if i % 32 == 31: marks.append(time.perf_counter())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    seg = [32/(marks[0] - t0)] + [32/(marks[j] - marks[j - 1]) for j in range(1, 4)]
    print(f"rep {rep}: {128/dt:7.1f} tok/s; per 32-token bucket: " + " ".join(f"{s:7.1f}" for s in seg), "slow tokens (index, ms):", slow, flush=True)
