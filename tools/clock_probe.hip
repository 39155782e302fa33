// clock_probe.hip — what shader clock does the chip hold (a) inside short dependent kernels, (b) inside a long streaming kernel?
// in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_spin(unsigned long long * out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a*1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long) a; }
}
__global__ void k_small(float * p, int n) { int i = blockIdx.x*256 + threadIdx.x; if (i < n) p[i] = p[i]*1.0001f + 1.0f; }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long * d; CK(hipMalloc(&d, 64)); float * p; CK(hipMalloc(&p, 1 << 20));
    unsigned long long h[3];
    for (int rep = 0; rep < 3; rep++) {
        for (int iters : { 2000, 200000 }) {
            // preceded by 3000 tiny dependent kernels (decode-like duty)
            for (int i = 0; i < 3000; i++) hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, s, p, 4096);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, iters);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
            printf("after tiny kernels: spin iters %6d: %llu shader cycles / %llu ref ticks -> %.0f MHz\n", iters, h[0], h[1], (double) h[0]/(double) h[1]*100.0);
        }
    }
    return 0;
}
