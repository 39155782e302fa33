// launch_floor.hip — calibration: cost of a dependent kernel boundary on this box (eager vs hipGraph replay),
// for an empty kernel, a tiny 16 KiB element kernel, and a kernel with a large by-value argument struct.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct big { long a[40]; };
__global__ void k_empty() {}
__global__ void k_small(float * p, int n) { int i = blockIdx.x*256 + threadIdx.x; if (i < n) p[i] = p[i]*1.0001f + 1.0f; }
__global__ void k_big(big b, float * p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += (float) b.a[3]; }
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
    const int N = 400; big b = {};
    for (int variant = 0; variant < 3; variant++) {
        auto launch = [&](hipStream_t st) {
            if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
            else if (variant == 1) hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, st, p, 4096);
            else hipLaunchKernelGGL(k_big, dim3(32), dim3(256), 0, st, b, p);
        };
        double eager = time_it(s, 20, [&] { for (int i = 0; i < N; i++) launch(s); })/N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) launch(s);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double graph = time_it(s, 20, [&] { hipGraphLaunch(ge, s); })/N;
        printf("variant %d: eager %.2f us/kernel, graph replay %.2f us/kernel\n", variant, eager, graph);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
