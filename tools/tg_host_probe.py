"""tg_host_probe.py — what a decoded token costs OUTSIDE its kernels: the same 128-token loop (sync per token, llama-bench's protocol) with and without the
logits read-back, and with the token's activation already on the device (no host-side embedding row / upload)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg
pkg = graft_pkg.load(); gg, ls = pkg.ggml, pkg.llama_synth
be = gg.Backend(0)
m = ls.SynthLlama(be, "llama3-8b", "Q4_K_M", n_ctx=160, seed=1)
tok = np.array([1], dtype=np.int32)
x = torch.zeros(4096, dtype=torch.float32, device="cuda")
def loop(n, **kw):
    m.kv_clear()
    for _ in range(16): m.decode(tok, **kw) if "dev_act_in" not in kw else m.decode(None, n_tokens=1, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.decode(tok, **kw) if "dev_act_in" not in kw else m.decode(None, n_tokens=1, **kw)
    torch.cuda.synchronize()
    return n/(time.perf_counter() - t0)
for rep in range(2):
    print("logits to host, sync per token      :", round(loop(128), 1), "tok/s")
    print("no logits read-back, sync per token :", round(loop(128, want_host=False, sync=True), 1), "tok/s")
    print("device activation in, logits to host:", round(loop(128, dev_act_in=x.data_ptr(), want_host=True, sync=True), 1), "tok/s")
    print("device activation in, no read-back  :", round(loop(128, dev_act_in=x.data_ptr(), want_host=False, sync=True), 1), "tok/s")
    print("no read-back, NO per-token sync     :", round(loop(128, want_host=False, sync=False), 1), "tok/s")
