#!/usr/bin/env python3
"""build.py — compile the MI355X backend in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Outputs (git-ignored, but they travel to the GPU box with the gpurun snapshot):
  lib/libggml-base-compat.so   harness stand-in for libggml-base (csrc/compat/ggml-compat.cpp)
  lib/libggml-mi355x.so        THE PRODUCT: the ggml backend (csrc/backend.cpp + csrc/*.hip)
  lib/libmi355x-harness.so     synthetic llama graph driver used by bench.py/tests (csrc/harness/*.cpp)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
CSRC = HERE / "csrc"
# MI355X_BUILD_VARIANT=<tag>: a second build beside the product (lib-<tag>/, build-<tag>/), e.g. the -DMI_STAMPS diagnostic build
VARIANT = os.environ.get("MI355X_BUILD_VARIANT", "")
LIB = HERE / ("lib-" + VARIANT if VARIANT else "lib")
OBJ = HERE / ("build-" + VARIANT if VARIANT else "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

INC = ["-I", str(ROOT / "include"), "-I", str(ROOT / "include" / "ggml-compat"), "-I", str(CSRC)]
COMMON = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-function",
          "-Wno-missing-field-initializers", "-Wno-array-bounds"]
HIPFLAGS = [f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"] + os.environ.get("MI_EXTRA_HIPFLAGS", "").split()

KERNEL_SRCS = ["quantize_act.hip", "mmvq.hip", "mmvq_cols_mfma.hip", "mmq.hip", "mm_dense.hip", "elem.hip", "decode_fused.hip", "attn_prefill.hip", "mmvq_fused.hip", "mmvq_stream.hip", "mmvq_stream_cols.hip", 
               # the persistent grouped mat-vec, one translation unit per weight format (they compile in parallel)
               "mmvq_fused_q4_K.hip", "mmvq_fused_q5_K.hip", "mmvq_fused_q6_K.hip", "mmvq_fused_q8_0.hip", "mmvq_fused_q4_0.hip", "mmvq_fused_mxfp4.hip",
               "mmvq_fused_q4_K_q5_K.hip", "mmvq_fused_q4_K_q6_K.hip", "mmvq_fused_q5_K_q6_K.hip", "mmvq_fused_q8_0_q4_K.hip"]


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise SystemExit(f"build failed: {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def newer(src_files, out):
    out = Path(out)
    if not out.exists():
        return True
    t = out.stat().st_mtime
    hdrs = list(CSRC.glob("*.h")) + list((ROOT / "include").rglob("*.h")) + [Path(__file__)]
    return any(Path(s).stat().st_mtime > t for s in list(src_files) + hdrs)


def compile_obj(src, extra):
    out = OBJ / (Path(src).name + ".o")
    if newer([src], out):
        run([HIPCC, *COMMON, *extra, *INC, "-c", str(src), "-o", str(out)])
    return out


def build(verbose=True):
    LIB.mkdir(exist_ok=True)
    OBJ.mkdir(exist_ok=True)
    jobs = []
    with ThreadPoolExecutor(max_workers=8) as ex:
        jobs.append(ex.submit(compile_obj, CSRC / "compat" / "ggml-compat.cpp", ["-x", "c++"]))
        for s in KERNEL_SRCS:
            jobs.append(ex.submit(compile_obj, CSRC / s, HIPFLAGS))
        jobs.append(ex.submit(compile_obj, CSRC / "backend.cpp", ["-x", "hip", *HIPFLAGS]))
        harness_srcs = sorted((CSRC / "harness").glob("*.cpp")) if (CSRC / "harness").exists() else []
        for s in harness_srcs:
            jobs.append(ex.submit(compile_obj, s, ["-x", "c++"]))
        objs = [j.result() for j in jobs]
    compat_o, kern_o, backend_o, harness_o = objs[0], objs[1:1 + len(KERNEL_SRCS)], objs[1 + len(KERNEL_SRCS)], objs[2 + len(KERNEL_SRCS):]

    base = LIB / "libggml-base-compat.so"
    if newer([compat_o], base):
        run([HIPCC, "-shared", "-o", str(base), str(compat_o), "-ldl"])
    be = LIB / "libggml-mi355x.so"
    if newer([*kern_o, backend_o, base], be):
        run([HIPCC, "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", str(be), *map(str, kern_o), str(backend_o),
             "-L", str(LIB), "-lggml-base-compat", "-Wl,-rpath,$ORIGIN"])
    if harness_o:
        hz = LIB / "libmi355x-harness.so"
        if newer([*harness_o, base], hz):
            run([HIPCC, "-shared", "-o", str(hz), *map(str, harness_o), "-L", str(LIB), "-lggml-base-compat", "-ldl",
                 "-Wl,-rpath,$ORIGIN"])
    if verbose:
        print("built:", ", ".join(p.name for p in sorted(LIB.glob("*.so"))))
    return LIB


if __name__ == "__main__":
    build()
