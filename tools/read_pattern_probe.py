import ctypes as C, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gpu_util import backend, proc
be = backend()
hbm = proc("ggml_backend_mi355x_test_hbm_read_gbps", C.c_double, [C.c_void_p, C.c_size_t, C.c_int])
for mb in (64, 512, 2048):
    print(mb, "MB:", [round(hbm(be.be, mb << 20, 5)) for _ in range(3)], flush=True)
