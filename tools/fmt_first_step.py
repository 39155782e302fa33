"""fmt_first_step.py — first decode step of a two-layer full-width Llama-3-8B in one weight format against the oracle, under the arrangements that take a
kernel or a fusion out of the path (each in its own process: the options are read once). Locates which launch a first-step difference comes from.

    python tools/fmt_first_step.py Q8_0"""
import os
import subprocess
import sys

if len(sys.argv) > 2:
    sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
    import numpy as np
    import oracle as orc
    from gpu_util import backend, pkg
    from test_gpu_llama_graph import RefLlama, read_weights
    ls = pkg.llama_synth
    ftype = sys.argv[1]
    be = backend(); be.set_option("graphs", 1); be.set_option("fusion", int(os.environ.get("FUSION", "1")))
    m = ls.SynthLlama(be, "llama3-8b", ftype, n_ctx=32, seed=6, n_layer=int(os.environ.get("NL", "2")), n_vocab=512)
    W = read_weights(m)
    rc = RefLlama(m.cfg, W, 32, "cpu16"); re_ = RefLlama(m.cfg, W, 32, "exact")
    emb = np.stack([m.embedding(3)])
    got = m.decode([3]); c = rc.decode(emb); e = re_.decode(emb)
    if sys.argv[2].startswith("tap "):
        # MI_HARNESS_TAP (set by the parent): the device result is an intermediate tensor; the oracle's counterpart is an input or an output of one of its mat-muls
        import ref_llama
        point = sys.argv[2][4:]
        which = {"attn_norm": ("attn_q", 0), "v": ("attn_v", 1), "attn": ("attn_output", 0), "wo": ("attn_output", 1), "ffn_norm": ("ffn_up", 0), "glu": ("ffn_down", 0),
                 "down": ("ffn_down", 1)}
        calls = {}
        real_mm = ref_llama.mm
        def spy(W_, key, x, mode):
            y = real_mm(W_, key, x, mode); calls[key] = (x.astype(np.float32).copy(), y.copy()); return y
        ref_llama.mm = spy
        rc2 = RefLlama(m.cfg, W, 32, "cpu16"); rc2.decode(emb); ref_llama.mm = real_mm
        key = "output" if point == "result_norm" else (0, which[point][0])
        exp = calls[key][0 if point == "result_norm" else which[point][1]].reshape(-1)
        got = got.reshape(-1)
        bad = np.abs(got - exp) > 1e-5*np.abs(exp).max()
        print(f"tap {point:12s} n {got.size:6d}  dev vs cpu-style {orc.nmse(exp, got):.3e}  elements off by > 1e-5 of max: {int(bad.sum())}  first {np.nonzero(bad)[0][:8]}", flush=True)
        sys.exit(0)
    if sys.argv[2] == "ops":
        # every mat-mul of the oracle's own first step, replayed as a one-node graph on the device with the SAME weights and the SAME input vector
        import ref_llama
        from gpu_util import run_mul_mat
        calls = []
        real_mm = ref_llama.mm
        def spy(W_, key, x, mode):
            y = real_mm(W_, key, x, mode); calls.append((key, x.astype(np.float32).copy(), y.copy())); return y
        ref_llama.mm = spy
        rc2 = RefLlama(m.cfg, W, 32, "cpu16"); rc2.decode(emb); ref_llama.mm = real_mm
        for key, x, y in calls:
            qt, data = W[key]
            x2 = x.reshape(-1, x.shape[-1]); mrows = data.shape[0]
            dev = run_mul_mat(qt, data, x2, mrows, x2.shape[1])
            ex = orc.mul_mat_2d(data, qt, x2, "exact")
            print(f"{str(key):24s} type {qt:2d} m {mrows:6d} k {x2.shape[1]:6d}  dev vs cpu-style {orc.nmse(y.reshape(dev.shape), dev):.3e}  cpu-style vs exact {orc.nmse(ex, y.reshape(ex.shape)):.3e}"
                  f"  |x| max {np.abs(x2).max():.3e} rms {np.sqrt((x2**2).mean()):.3e}", flush=True)
        sys.exit(0)
    print(f"{sys.argv[2]:40s} vs cpu-style {orc.nmse(c, got):.3e}  vs exact {orc.nmse(e, got):.3e}  cpu-style vs exact {orc.nmse(e, c):.3e}", flush=True)
    sys.exit(0)
if os.environ.get("TAPS"):
    for point in ("attn_norm", "v", "attn", "wo", "ffn_norm", "glu", "down", "result_norm"):
        subprocess.run([sys.executable, __file__, sys.argv[1], "tap " + point], env={**os.environ, "NL": "1", "FUSION": os.environ.get("FUSION", "0"), "MI_HARNESS_TAP": "0:" + point}, check=False)
    sys.exit(0)
if os.environ.get("OPS"):
    subprocess.run([sys.executable, __file__, sys.argv[1], "ops"], env={**os.environ, "NL": "1"}, check=False); sys.exit(0)
for label, env in (("default", {}), ("STREAM=0", {"GGML_MI355X_STREAM": "0"}), ("fusion off", {"FUSION": "0"}), ("FIN=0", {"GGML_MI355X_FIN": "0"}),
                   ("one layer", {"NL": "1"}), ("one layer, fusion off", {"NL": "1", "FUSION": "0"})):
    subprocess.run([sys.executable, __file__, sys.argv[1], label], env={**os.environ, **env}, check=False)
