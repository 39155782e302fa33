// launch_shape_probe.hip — what does the SHAPE of a launch cost at a dependent kernel boundary? A chain of N identical, (almost) empty kernels replayed
// from a hipGraph: time per kernel for workgroups of 64 .. 1024 threads, 0 .. 150 KB of dynamic LDS, a 64-byte or a 928-byte argument struct, 256 or 32
// workgroups. The streamed mat-vec launches 256 workgroups x 576 threads with ~150 KB of LDS and 928 bytes of arguments.
//   hipcc --offload-arch=gfx950 -O3 -o launch_shape_probe launch_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct small_args { float * p; long pad[7]; };
struct big_args { float * p; long pad[115]; };
template <typename A> __global__ void k_shape(const A a) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) { lds[0] = (char) blockIdx.x; a.p[blockIdx.x] = (float) a.pad[3] + lds[0]; }
}
template <typename F> double time_it(hipStream_t s, int reps, F f) {
    f(); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < reps; i++) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count()/reps;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float * p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void *) k_shape<small_args>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void *) k_shape<big_args>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int N = 300;
    small_args sa = {}; sa.p = p; big_args ba = {}; ba.p = p;
    printf("%6s %8s %8s %6s | us per kernel in a dependent chain (graph replay)\n", "grid", "threads", "LDS", "args");
    const int grids[] = { 256, 32 }; const int threads[] = { 64, 256, 320, 576, 1024 }; const int ldss[] = { 0, 65536, 153600 };
    for (int grid : grids) for (int th : threads) for (int lds : ldss) for (int big = 0; big < 2; big++) {
        if (grid == 32 && (lds == 65536 || big)) continue;
        auto launch = [&](hipStream_t st) {
            if (big) hipLaunchKernelGGL(k_shape<big_args>, dim3(grid), dim3(th), lds, st, ba);
            else     hipLaunchKernelGGL(k_shape<small_args>, dim3(grid), dim3(th), lds, st, sa);
        };
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) launch(s);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double t = time_it(s, 10, [&] { hipGraphLaunch(ge, s); })/N;
        printf("%6d %8d %8d %6d | %.2f\n", grid, th, lds, big ? 928 : 64, t);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
