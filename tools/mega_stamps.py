"""mega_stamps.py — per-phase timeline of the persistent decode kernel from in-kernel wall-clock stamps (debug build:
MI355X_BUILD_VARIANT=stamps MI_EXTRA_HIPFLAGS=-DMI_STAMPS python llama.cpp-gfx906_amd/build.py).
Per (phase, workgroup): 0 entry, 1 weight ring issued, 2 input signalled, 3 image in LDS, 4 rows done, 5 phase end (signalled / finalised)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graft_pkg

pkg = graft_pkg.load()
gg, ls = pkg.ggml, pkg.llama_synth
model = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
ftype = sys.argv[2] if len(sys.argv) > 2 else "Q4_K_M"
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/mega_stamps.npz"
be = gg.Backend(0)
lib = C.CDLL(str(gg.LIBDIR / "libggml-mi355x.so"))
lib.mi355x_mega_stamps_enable.argtypes = [C.c_int, C.c_int]
lib.mi355x_mega_stamps_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
NWG = 256
assert lib.mi355x_mega_stamps_enable(1024, NWG) == 0
m = ls.SynthLlama(be, model, ftype, n_ctx=128, seed=1)
tok = np.array([1], dtype=np.int32)
for i in range(10):
    m.decode(tok)
buf = np.zeros(1024 * NWG * 8, dtype=np.uint64); nph = C.c_int(0); nwg = C.c_int(0)
assert lib.mi355x_mega_stamps_read(buf.ctypes.data, C.byref(nph), C.byref(nwg)) == 0
P, W = nph.value, nwg.value
st = buf[: P * W * 8].reshape(P, W, 8).astype(np.int64)
np.savez_compressed(out, st=st)
t0 = st[st > 0].min()
us = np.where(st > 0, (st - t0) / 100.0, np.nan)
print(f"{P} phases x {W} workgroups; span {np.nanmax(us):.1f} us")
print("phase  act  entry(med)  ring    signalled  image   rows(min/med/max)          end(med/max)   dur(max end - prev max end)")
prev = 0.0
for p in range(min(P, 14)):
    a = ~np.isnan(us[p, :, 0])
    if not a.any():
        print(p, "  (no stamps)"); continue
    e = us[p, a]
    med = lambda k: np.nanmedian(e[:, k])
    print(f"{p:4d} {a.sum():4d} {med(0):9.2f} {med(1)-med(0):7.2f} {med(2)-med(0):9.2f} {med(3)-med(0):7.2f}   {np.nanmin(e[:,4])-med(0):6.2f}/{med(4)-med(0):6.2f}/{np.nanmax(e[:,4])-med(0):6.2f}   "
          f"{med(5)-med(0):7.2f}/{np.nanmax(e[:,5])-med(0):7.2f}   {np.nanmax(e[:,5]) - prev:7.2f}")
    prev = np.nanmax(e[:, 5])
# the chunk owners of the down phases (stamps 6: previous phase hinted complete, 7: piece published), relative to the phase's median entry
for p in range(min(P, 14)):
    a = us[p, :, 6]
    if (~np.isnan(a)).any():
        e0 = np.nanmedian(us[p, :, 0])
        print(f"phase {p}: owners: prev-phase-complete seen {np.nanmin(a)-e0:.2f}/{np.nanmedian(a)-e0:.2f}/{np.nanmax(a)-e0:.2f}  piece published {np.nanmin(us[p,:,7])-e0:.2f}/{np.nanmedian(us[p,:,7])-e0:.2f}/{np.nanmax(us[p,:,7])-e0:.2f}"
              f"  consumers hinted {np.nanmin(us[p,:,2])-e0:.2f}/{np.nanmedian(us[p,:,2])-e0:.2f}/{np.nanmax(us[p,:,2])-e0:.2f}; prev phase rows done max {np.nanmax(us[p-1,:,4])-e0:.2f} end max {np.nanmax(us[p-1,:,5])-e0:.2f}")
# per-layer period
ends = np.array([np.nanmax(us[p, :, 5]) if (~np.isnan(us[p, :, 5])).any() else np.nan for p in range(P)])
print("layer period (us):", [round(float(ends[1 + 5*(l+1)] - ends[1 + 5*l]), 2) for l in range(0, min(6, (P - 2)//5 - 1))])
m.free(); be.free()
