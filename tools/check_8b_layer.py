"""check_8b_layer.py — two Llama-3-8B-shaped layers (n_ff = 14336: the shape whose gate/up launch takes the strided fin path) against the oracle."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
from gpu_util import backend, gg, pkg
import ref_llama, oracle as orc
ls = pkg.llama_synth
be = backend(); be.set_option("graphs", 1); be.set_option("fusion", 1)
m = ls.SynthLlama(be, "llama3-8b", "Q4_K_M", n_ctx=32, seed=5, n_layer=2, n_vocab=512)
W = ref_llama.read_weights(m, gg)
rc = ref_llama.RefLlama(m.cfg, W, 32, "cpu16")
for t in (3, 7, 9, 11):
    got = m.decode([t]); exp = rc.decode(np.stack([m.embedding(t)]))
    print("token", t, "nmse vs cpu-style oracle", orc.nmse(exp, got))
m.free()
