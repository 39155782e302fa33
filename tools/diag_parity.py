import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
from gpu_util import backend, gg, pkg
import ref_llama, oracle as orc
ls = pkg.llama_synth
for ftype in ("Q8_0", "Q4_K_M"):
    for fusion in (0, 1):
        be = backend(); be.set_option("graphs", 0); be.set_option("fusion", fusion)
        m = ls.SynthLlama(be, "tiny", ftype, n_ctx=64, seed=21)
        W = ref_llama.read_weights(m, gg)
        refs = {k: ref_llama.RefLlama(m.cfg, W, 64, k) for k in ("cpu", "cpu16", "exact")}
        toks = [5, 9, 200, 17, 3, 44]
        for i, t in enumerate(toks):
            emb = np.stack([m.embedding(t)])
            got = m.decode([t])
            ex = {k: r.decode(emb) for k, r in refs.items()}
            print(ftype, "fusion", fusion, "pos", i, " ".join(f"nmse_vs_{k} {orc.nmse(v, got):.2e}" for k, v in ex.items()),
                  f"| cpu_vs_exact {orc.nmse(ex['exact'], ex['cpu']):.2e} cpu16_vs_cpu {orc.nmse(ex['cpu'], ex['cpu16']):.2e}")
        m.free()
