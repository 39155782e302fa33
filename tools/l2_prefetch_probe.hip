// l2_prefetch_probe.hip — does it pay to pull the NEXT launch's first weight bytes into the XCD's L2 from the tail of the current launch?
//
// Kernel B is a cut-down loader of the streamed mat-vec (csrc/mmvq_stream.h): one wave per workgroup, one workgroup per CU, copies the workgroup's
// contiguous S bytes HBM -> LDS with global_load_lds_dwordx4 nt, 36 KiB in flight, no consumers. Stamps (100 MHz): entry, first 9 KiB landed, all landed.
// Kernel A runs right before it on the same stream and (variant 1) reads the first P bytes of "its" workgroup's range with plain loads (results
// discarded), (variant 2) the range of workgroup b + 1 (another XCD), (variant 0) nothing. Between pairs a 1 GiB read (default cache policy: a nontemporal one leaves the caches as they are) evicts L2 and the Infinity Cache.
// Also printed: how often workgroup b of A and of B ran on the same XCD (HW_REG_XCC_ID).
//
//   hipcc --offload-arch=gfx950 -O3 -o l2_prefetch_probe l2_prefetch_probe.hip && ./l2_prefetch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s at line %d\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int int4v __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ void __launch_bounds__(256) k_evict(const int4v * p, size_t n16, unsigned * sink) {
    int4v acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t) blockIdx.x*256 + threadIdx.x; i < n16; i += (size_t) gridDim.x*256) { const int4v v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) *sink = 1;
}

// kernel A: `busy_us` of spinning (stands for the tail of the launch before), then the prefetch
__global__ void __launch_bounds__(64) k_a(const char * W, size_t S, int P, int shift, int busy_ticks, unsigned * xcc, unsigned * sink) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) xcc[b] = xcc_id();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(4);
    if (P <= 0) return;
    const char * src = W + (size_t)((b + shift) % gridDim.x)*S;
    int acc = 0;
    for (int off = 0; off < P; off += 8192) {       // 8 x 1 KiB per trip, all issued before any is waited for
        int4v v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = *(const int4v *) (src + off + j*1024 + lane*16);
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x;
    }
    if (acc == 0x12345678) *sink = 1;
}

template <int N>
static __device__ __forceinline__ void dma(const char * gbase, unsigned voff, unsigned lds_dst) {
#define MI_DMA(OFF_) "global_load_lds_dwordx4 %0, %2 offset:" #OFF_ " nt\n\t"
    if (N == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" MI_DMA(0) MI_DMA(1024) MI_DMA(2048) :: "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
#undef MI_DMA
}

// kernel B: the loader. stamps[b][0..3] = entry, first slot (9 KiB) landed, all landed, xcc
__global__ void __launch_bounds__(64) k_b(const char * W, size_t S, unsigned long long * stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const char * src = W + (size_t) b*S;
    const unsigned ring = (unsigned)(size_t)(const char __attribute__((address_space(3))) *) lds;
    const int nslots = (int)(S/9216);      // 9 KiB slots, 12 of them = 108 KiB ring
    unsigned long long t1 = 0;
    for (int i = 0; i < nslots; i++) {
        const char * g = src + (size_t) i*9216;
        const unsigned dst = ring + (unsigned)(i % 12)*9216;
        dma<3>(g, lane*16, dst); dma<3>(g + 3072, lane*16, dst + 3072); dma<3>(g + 6144, lane*16, dst + 6144);
        if (i == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memrealtime(); }
        else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { stamps[b*4 + 0] = t0; stamps[b*4 + 1] = t1; stamps[b*4 + 2] = t2; stamps[b*4 + 3] = xcc_id(); }
}

// a small kernel between A and B (the rotation-table launch, the attention launch): does it move the workgroup -> XCD assignment of B?
__global__ void __launch_bounds__(64) k_mid(unsigned * sink) { if (threadIdx.x == 9999) *sink = 1; }

int main(int argc, char ** argv) {
    const bool use_graph = argc > 1 && strcmp(argv[1], "graph") == 0;
    const int mid = argc > 2 ? atoi(argv[2]) : 0;      // workgroups of the kernel between A and B (0: none)
    printf("%s launches, %d workgroups between A and B\n", use_graph ? "graph" : "eager", mid);
    const int NWG = 256;
    const size_t S_list[] = { 36864, 258048 };          // wo-like (37 KB per CU), QKV-like, GLU-like
    const size_t EV = (size_t) 1 << 30;
    char * W; char * E; unsigned * xcc; unsigned * sink; unsigned long long * stamps;
    CK(hipMalloc(&W, (size_t) NWG*258048 + 65536)); CK(hipMalloc(&E, EV)); CK(hipMalloc(&xcc, NWG*4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&stamps, NWG*4*8));
    CK(hipMemset(W, 1, (size_t) NWG*258048 + 65536)); CK(hipMemset(E, 2, EV));
    CK(hipFuncSetAttribute((const void *) k_b, hipFuncAttributeMaxDynamicSharedMemorySize, 12*9216));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> hs(NWG*4); std::vector<unsigned> hx(NWG);
    printf("%8s %8s %6s | %8s | first slot med  max | all landed med  max | B event us | same-XCD\n", "S/CU", "prefetch", "shift", "A busy us");
    for (size_t S : S_list) {
        for (int busy : { 3 }) {
            for (int variant : { 0, 1, 4 }) {
                const int P = variant == 0 ? 0 : variant == 3 ? 16384 : variant == 4 ? 65536 : variant == 5 ? 131072 : 32768, shift = variant == 2 ? 1 : 0;
                const int Peff = (int) std::min<size_t>((size_t) P, S) & ~8191;
                double f_med = 0, f_max = 0, a_med = 0, a_max = 0, ev = 0; int same = 0; const int reps = 6;
                // argv[1] == "graph": A and B replayed as one captured hipGraph (what the backend does) instead of two eager launches
                hipGraphExec_t gx = nullptr;
                if (use_graph) {
                    hipGraph_t gr; CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    hipLaunchKernelGGL(k_a, dim3(NWG), dim3(64), 0, st, W, S, Peff, shift, busy*100, xcc, sink);
                    if (mid) hipLaunchKernelGGL(k_mid, dim3(mid), dim3(64), 0, st, sink);
                    hipLaunchKernelGGL(k_b, dim3(NWG), dim3(64), 12*9216, st, W, S, stamps);
                    CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&gx, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
                }
                for (int r = 0; r < reps; r++) {
                    hipLaunchKernelGGL(k_evict, dim3(2048), dim3(256), 0, st, (const int4v *) E, EV/16, sink);
                    CK(hipEventRecord(e0, st));
                    if (use_graph) CK(hipGraphLaunch(gx, st));
                    else {
                        hipLaunchKernelGGL(k_a, dim3(NWG), dim3(64), 0, st, W, S, Peff, shift, busy*100, xcc, sink);
                        if (mid) hipLaunchKernelGGL(k_mid, dim3(mid), dim3(64), 0, st, sink);
                        hipLaunchKernelGGL(k_b, dim3(NWG), dim3(64), 12*9216, st, W, S, stamps);
                    }
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    CK(hipMemcpy(hs.data(), stamps, NWG*4*8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, NWG*4, hipMemcpyDeviceToHost));
                    unsigned long long tmin = ~0ull; for (int b = 0; b < NWG; b++) tmin = std::min(tmin, hs[b*4]);
                    std::vector<double> f(NWG), a(NWG);
                    for (int b = 0; b < NWG; b++) { f[b] = (hs[b*4 + 1] - tmin)/100.0; a[b] = (hs[b*4 + 2] - tmin)/100.0; if (r == reps - 1 && hx[b] == (unsigned) hs[b*4 + 3]) same++; }
                    std::sort(f.begin(), f.end()); std::sort(a.begin(), a.end());
                    if (r > 0) { f_med += f[NWG/2]; f_max += f[NWG - 1]; a_med += a[NWG/2]; a_max += a[NWG - 1]; ev += ms*1e3; }
                }
                if (gx) CK(hipGraphExecDestroy(gx));
                const int n = reps - 1;
                printf("%8zu %8d %6d | %8d | %13.2f %5.2f | %13.2f %5.2f | %10.2f | %d/%d\n", S, Peff, shift, busy, f_med/n, f_max/n, a_med/n, a_max/n, ev/n, same, NWG);
            }
        }
    }
    return 0;
}
