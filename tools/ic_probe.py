"""ic_probe.py — streaming-read rate of the backend's probe kernel over working sets from 16 MB to 2 GB, read repeatedly: what the memory-side cache
(256 MB Infinity Cache) returns when the same range is read again, against HBM for ranges that do not fit."""
import ctypes as C
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _d in ("tests", "oracle", ""):
    sys.path.insert(0, os.path.join(_root, _d))
from gpu_util import backend, gg

be = backend()
p = gg.base().ggml_backend_reg_get_proc_address(be.reg, b"ggml_backend_mi355x_test_hbm_read_gbps")
hbm = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_size_t, C.c_int)(p)
for mb in (16, 32, 64, 128, 192, 256, 512, 2048):
    r = max(hbm(be.be, mb << 20, 20) for _ in range(3))
    print(f"{mb:5d} MB read 20 times back to back: {r:8.1f} GB/s", flush=True)
