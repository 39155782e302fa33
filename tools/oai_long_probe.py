"""oai_long_probe.py — a 300-token prompt pass through two full-width gpt-oss-20b layers with every fusion on and node by node: the grouped QKV launch with
bias rows in its epilogue needs >= 256 tokens, more than the oracle-checked prompts of the test suite hold."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, oracle as orc
from gpu_util import backend, pkg
ls = pkg.llama_synth
be = backend(); be.set_option("graphs", 1)
m = ls.SynthLlama(be, "gpt-oss-20b", "MXFP4_MOE", n_ctx=512, seed=10, n_layer=2, n_vocab=512)
toks = np.random.default_rng(0).integers(0, m.cfg["n_vocab"], size=300).astype(np.int32)
outs = []
for f in (1, 0):
    be.set_option("fusion", f); m.kv_clear(); be.reset_counters()
    outs.append(m.decode(toks).copy()); print("fusion", f, be.counters()["kernels_launched"])
print("nmse fused vs node-by-node", orc.nmse(outs[1], outs[0]))
