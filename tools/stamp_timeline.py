"""stamp_timeline.py — per-launch timeline of one decoded token from in-kernel wall-clock stamps.

Needs the debug build:  touch llama.cpp-gfx906_amd/csrc/decode_fused.hip && MI_EXTRA_HIPFLAGS=-DMI_STAMPS python llama.cpp-gfx906_amd/build.py
Each grouped mat-vec workgroup stamps (100 MHz clock, common to all CUs): 0 entry, 1 activation image ready (after the prologue
barrier), 2 first row pair of wave 0 done, 3 exit. The slots of the last captured graph are read after a few replays."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg

pkg = graft_pkg.load()
gg, ls = pkg.ggml, pkg.llama_synth
model = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/stamps.json"
be = gg.Backend(0)
lib = C.CDLL(str(gg.LIBDIR / "libggml-mi355x.so"))
lib.mi355x_stamps_enable.argtypes = [C.c_int]
lib.mi355x_stamps_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
assert lib.mi355x_stamps_enable(4096) == 0
m = ls.SynthLlama(be, model, ftype, n_ctx=128, seed=1)
tok = np.array([1], dtype=np.int32)
for i in range(12):
    m.decode(tok)
torch.cuda.synchronize()
used = lib.mi355x_stamps_used()
per_token = m.graph_nodes() if hasattr(m, "graph_nodes") else 0
# launches per token: eager token 0 used the first slots; the captured graph holds the last group of the same length
n_first = None
rows = []
SN = 16
buf = np.zeros(1024 * SN, dtype=np.uint64); meta = np.zeros(8, dtype=np.int32); nbytes = C.c_longlong(0)
metas = []
for s in range(used):
    lib.mi355x_stamps_read(s, buf.ctypes.data, meta.ctypes.data, C.byref(nbytes))
    metas.append((tuple(int(v) for v in meta[:7]), int(nbytes.value), buf[: int(meta[0]) * SN].reshape(-1, SN).copy()))
# find the period: the launch sequence repeats (eager pass, then captured pass)
sig = [mm[0] for mm in metas]
period = next(p for p in range(1, used + 1) if used % p == 0 and all(sig[i] == sig[i % p] for i in range(used)))
last = metas[used - period:]
t_base = min(int(st[:, 0].min()) for _, _, st in last)
prev_end = None
tl = []
for (blocks, k, rws, ta, tb, mode, glu), nb, st in last:
    st = (st.astype(np.int64) - t_base) / 100.0   # us
    e = {"blocks": blocks, "k": k, "rows": rws, "types": [ta, tb], "mode": mode, "glu": glu, "MB": round(nb / 1e6, 2),
         "t0_min": float(st[:, 0].min()), "t0_max": float(st[:, 0].max()),
         "t1_med": float(np.median(st[:, 1])), "t4_med": float(np.median(st[:, 4])), "t5_med": float(np.median(st[:, 5])), "t6_med": float(np.median(st[:, 6])), "t7_med": float(np.median(st[:, 7])), "t2_med": float(np.median(st[:, 2])),
         "t3_min": float(st[:, 3].min()), "t3_med": float(np.median(st[:, 3])), "t3_max": float(st[:, 3].max())}
    e["sclk_ghz"] = float(np.median((st[:, 9] - st[:, 8]) * 100.0 / np.maximum(st[:, 3] - st[:, 0], 1e-3))) / 1e3   # shader cycles per wall us / 1000
    e["gap_before"] = None if prev_end is None else round(e["t0_min"] - prev_end, 2)
    e["dur"] = round(e["t3_max"] - e["t0_min"], 2)
    prev_end = e["t3_max"]
    tl.append(e)
json.dump({"model": model, "ftype": ftype, "launches": tl}, open(out, "w"))
np.savez_compressed(out.replace(".json", "_raw.npz"), **{f"l{i}": (st.astype(np.int64) - t_base) for i, (_, _, st) in enumerate(last[:20])})
print(f"period {period} launches/token; token span {tl[-1]['t3_max'] - tl[0]['t0_min']:.1f} us")
print(" idx  MB     blocks gap   t0spread  ss(t4) scale(t7) chunk0(t5) quant(t6) pro(t1-t0) first(t2-t0) t3min-t0  t3med-t0  dur   GB/s(dur)")
for i, e in enumerate(tl[: 5 * 3 + 2]):
    print(f"{i:3d} {e['MB']:7.2f} {e['blocks']:5d} {str(e['gap_before']):>6s} {e['t0_max']-e['t0_min']:7.2f} {e['t4_med']-e['t0_min']:7.2f} {e['t7_med']-e['t0_min']:7.2f} {e['t5_med']-e['t0_min']:7.2f} {e['t6_med']-e['t0_min']:7.2f} {e['t1_med']-e['t0_min']:9.2f} "
          f"{e['t2_med']-e['t0_min']:10.2f} {e['t3_min']-e['t0_min']:9.2f} {e['t3_med']-e['t0_min']:9.2f} {e['dur']:7.2f} {e['MB']/e['dur']*1e3/1e3:8.2f} TB/s  sclk {e['sclk_ghz']:.2f} GHz")
tot_dur = sum(e["dur"] for e in tl); tot_gap = sum(e["gap_before"] or 0 for e in tl)
print(f"sum of launch durations {tot_dur:.1f} us, sum of gaps between them {tot_gap:.1f} us")
m.free()
