import sys, time, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import graft_pkg
pkg = graft_pkg.load(); gg, ls = pkg.ggml, pkg.llama_synth
be = gg.Backend(0)
m = ls.SynthLlama(be, "llama3-8b", "Q4_K_M", n_ctx=512, seed=1)
rng = np.random.default_rng(0)
toks = rng.integers(0, 128256, size=512).astype(np.int32)
for _ in range(2):
    m.kv_clear(); m.decode(toks)
def t(f, n=3):
    r = []
    for _ in range(n):
        m.kv_clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); r.append((time.perf_counter() - t0)*1e3)
    return min(r)
print("decode(tokens) ms:", t(lambda: m.decode(toks)))
x = torch.zeros(512*4096, dtype=torch.float32, device="cuda")
print("decode(dev_act_in) ms:", t(lambda: m.decode(None, n_tokens=512, dev_act_in=x.data_ptr(), want_host=True, sync=True)))
e = np.zeros(4096, np.float32)
t0 = time.perf_counter()
for tk in toks[:64]: m.embedding(int(tk))
print("embedding() per 512 tokens ms (python loop, upper bound):", (time.perf_counter() - t0)*1e3*8)
