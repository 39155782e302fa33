"""mall_size_probe.py — streaming-read rate of the same buffer read over and over, by size: does the 256 MB Infinity Cache keep read data?"""
import ctypes as C
import sys

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gpu_util import backend, proc

be = backend()
hbm = proc("ggml_backend_mi355x_test_hbm_read_gbps", C.c_double, [C.c_void_p, C.c_size_t, C.c_int])
for mb in (8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048):
    r = [hbm(be.be, mb << 20, max(4, 2048 // mb)) for _ in range(3)]
    print(f"{mb:5d} MB: {max(r):8.1f} GB/s  (runs {[round(x) for x in r]})", flush=True)
