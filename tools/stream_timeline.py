"""stream_timeline.py — per-launch timeline of one decoded token from the streamed mat-vec's in-kernel stamps (csrc/mmvq_stream.h, ST_STAMP).

Needs the diagnostic build:  MI355X_BUILD_VARIANT=stamps MI_EXTRA_HIPFLAGS=-DMI_STAMPS python llama.cpp-gfx906_amd/build.py
and MI355X_BUILD_VARIANT=stamps when running. Every wave stamps (100 MHz clock common to all CUs): 0 entry; consumers: 1 activation loads
queued + first barrier passed, 2 image ready, 3 first slot landed, 4 last slot computed, 5 all consumers done, 6 exit; loader: 1 everything landed, 2 the next
launch's prefetch issued and returned (= the loader wave's exit)."""
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
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/stream_stamps.json"
be = gg.Backend(0)
lib = C.CDLL(str(gg.LIBDIR / "libggml-mi355x.so"))
lib.mi355x_stream_stamps_enable.argtypes = [C.c_int]
lib.mi355x_stream_stamps_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
assert lib.mi355x_stream_stamps_enable(2048) == 0
m = ls.SynthLlama(be, model, ftype, n_ctx=128, seed=1)
tok = np.array([1], dtype=np.int32)
for i in range(8):
    m.decode(tok)
torch.cuda.synchronize()
used = lib.mi355x_stream_stamps_used()
NW = 9
NS = 16
buf = np.zeros(256 * NW * NS, dtype=np.uint64); meta = np.zeros(8, dtype=np.int32); nbytes = C.c_longlong(0)
metas = []
for s in range(used):
    lib.mi355x_stream_stamps_read(s, buf.ctypes.data, meta.ctypes.data, C.byref(nbytes))
    metas.append((tuple(int(v) for v in meta[:7]), int(nbytes.value), buf.reshape(256, NW, NS)[: int(meta[0])].copy()))
sig = [mm[0] for mm in metas]
period = next(p for p in range(1, used + 1) if used % p == 0 and all(sig[i] == sig[i % p] for i in range(used)))
last = metas[used - period:]
t_base = min(int(st[:, :, 0][st[:, :, 0] > 0].min()) for _, _, st in last)
names = ["entry", "loads queued", "image ready", "first slot", "last slot done", "consumers met", "exit"]
print(f"period {period} stream launches/token")
print(" idx   MB   wgs  mode glu | gap | entry med/max | x loads back med/max | block max known med/max | quants packed med/max | wave's image part med/max | image med/max | 1st slot med/max | loader starts med/max | loader landed med/max | prefetch done med/max | last slot med/max | exit med/max | dur (incl. loader exit) | TB/s")
prev_end = None; tl = []
for i, ((blocks, k, rws, ta, tb, mode, glu), nb, st) in enumerate(last):
    t = (st.astype(np.int64) - t_base) / 100.0
    t[st == 0] = np.nan
    cons = t[:, :8, :]; load = t[:, 8, :]
    t0 = np.nanmin(t[:, :, 0]); tend = max(np.nanmax(cons[:, :, 6]), np.nanmax(load[:, 2]) if np.isfinite(load[:, 2]).any() else 0.0)
    def mm(a): return (float(np.nanmedian(a) - t0), float(np.nanmax(a) - t0))
    e = dict(idx=i, MB=nb / 1e6, wgs=blocks, mode=mode, glu=glu, types=[ta, tb], gap=None if prev_end is None else t0 - prev_end,
             entry=mm(t[:, :, 0]), queued=mm(cons[:, :, 1]), xback=mm(cons[:, :, 7]), q_max=mm(cons[:, :4, 8]), q_quant=mm(cons[:, :4, 9]), image=mm(cons[:, :, 2]), first=mm(cons[:, :, 3]), landed=mm(load[:, 1]), lstart=mm(load[:, 3]), pf=mm(load[:, 2]) if np.isfinite(load[:, 2]).any() else (float('nan'), float('nan')), last=mm(cons[:, :, 4]), exit=mm(cons[:, :, 6]), dur=tend - t0)
    prev_end = tend; tl.append(e)
    if i < 14 or i >= len(last) - 2:
        print(f"{i:3d} {e['MB']:6.2f} {blocks:4d} {mode:4d} {glu:3d} | {('%.2f' % e['gap']) if e['gap'] is not None else '   -':>5s} | {e['entry'][0]:5.2f} {e['entry'][1]:5.2f} | {e['xback'][0]:5.2f} {e['xback'][1]:5.2f} | {e['q_max'][0]:5.2f} {e['q_max'][1]:5.2f} | {e['q_quant'][0]:5.2f} {e['q_quant'][1]:5.2f} | {e['queued'][0]:5.2f} {e['queued'][1]:5.2f} | {e['image'][0]:5.2f} {e['image'][1]:5.2f} | "
              f"{e['first'][0]:5.2f} {e['first'][1]:5.2f} | {e['lstart'][0]:5.2f} {e['lstart'][1]:5.2f} | {e['landed'][0]:5.2f} {e['landed'][1]:5.2f} | {e['pf'][0]:5.2f} {e['pf'][1]:5.2f} | {e['last'][0]:5.2f} {e['last'][1]:5.2f} | {e['exit'][0]:5.2f} {e['exit'][1]:5.2f} | {e['dur']:5.2f} | {e['MB'] / e['dur'] / 1e0 / 1e3 * 1e3 / 1e3:5.2f}")
print(f"sum of stream-launch durations {sum(e['dur'] for e in tl):.1f} us; gaps between consecutive stream launches (attention etc. inside) {sum(e['gap'] or 0 for e in tl):.1f} us; span {tl[-1]['dur'] + sum((e['gap'] or 0) + e['dur'] for e in tl[:-1]):.1f} us")
json.dump({"model": model, "ftype": ftype, "launches": tl}, open(out, "w"))
m.free()
