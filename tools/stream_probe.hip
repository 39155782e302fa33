// stream_probe.hip — timing + stamps for the streamed mat-vec structure of csrc/mmvq_stream.h (loader wave + slot ring + one unit per lane)
// on the Llama-3-8B decode shapes, Q4_K, with a pre-quantized activation (the product kernel adds the prologues / epilogues).
// Build: hipcc -O3 --offload-arch=gfx950 -I llama.cpp-gfx906_amd/csrc -I include -I include/ggml-compat -o tools/stream_probe tools/stream_probe.hip
#include "mmvq_stream.h"
#include <vector>
#include <algorithm>
#include <math.h>
using namespace mi355x;
// (the probe predates two renames in mmvq_stream.h: the sync area's size and the number of 16-byte chunks of a unit)
static constexpr int ST_SYNC_BYTES = 2*ST_SYNC_WORDS*4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct probe_args {
    const char * W; const char * W2; int m, k;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;
    float * dst; long long w_bytes; int nslot_ring; unsigned long long * stamps;
};
#define STAMP(i_) do { if (p.stamps && lane == 0) p.stamps[((size_t) blockIdx.x*(ST_NC + 2) + wave)*8 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)

template <bool GLU, bool NT, int MODE, int NL, int DEPTH>
__global__ void __launch_bounds__((ST_NC + NL)*64, 3) k_stream_probe(const probe_args p) {
    typedef st_unit<T_Q4_K> U;
    constexpr int PPS = (64*U::UB + 1023)/1024, SLOT = PPS*1024;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int G = gridDim.x, b = blockIdx.x, nb = p.k >> 8;
    const int r0 = (int)((long long) b*p.m/G), r1 = (int)((long long)(b + 1)*p.m/G), R = r1 - r0;
    const int n1 = R*nb, ns1 = (n1 + 63) >> 6, nslots = GLU ? 2*ns1 : ns1;      // units / slots of one stream; slots of the workgroup
    const int S = p.nslot_ring;
    uint32_t * sync = (uint32_t *) lds;        // [0..1] slots landed per loader, [2] image parts ready, [3] consumers finished, [16 + s] done[s]
    char * act = lds + ST_SYNC_BYTES;
    float * dd = (float *) (act + (size_t) nb*ST_ACT_STRIDE);
    float * part = dd + ((nb + 3) & ~3);
    const int n_part = (nb == 16 ? R : n1)*(GLU ? 2 : 1);
    char * ring = (char *) (((size_t)(part + n_part) + 15) & ~(size_t) 15);
    STAMP(0);
    if (threadIdx.x < 64) sync[threadIdx.x] = 0;
    // consumers: request the activation before any weight is requested (a CU returns loads in request order)
    int4v areg[2] = { {0,0,0,0}, {0,0,0,0} }, breg[2] = { {0,0,0,0}, {0,0,0,0} }; float dreg = 0.0f;
    if (wave < ST_NC) {
        const int nq = p.k >> 4;
#pragma unroll
        for (int i = 0; i < 2; i++) { const int q = min((int) threadIdx.x + i*ST_NC*64, nq - 1); areg[i] = *(const int4v *) (p.a_qs + (size_t) q*16); }
        const int ibl = min((int) threadIdx.x, nb - 1);
        breg[0] = *(const int4v *) (p.a_bs + (size_t) ibl*16); breg[1] = *(const int4v *) (p.a_bs + (size_t) ibl*16 + 8);
        dreg = p.a_d[ibl];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (wave >= ST_NC) {
        // ================= the loaders: loader l takes slots l, l + NL, ... =================
        const int ld = wave - ST_NC;
        const uint32_t ring_a = st_lds_addr(ring);
        const uint32_t voff = lane*16;
        int landed = 0, n = 0;                                  // of this loader's slots
        for (int i = ld; i < nslots; i += NL, n++) {
            const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
            const char * gb = (si ? p.W2 : p.W) + (long long) r0*nb*U::UB + (long long) il*64*U::UB;
            if (i >= S) {
                // the slot must have been consumed; publish what is in flight first so that nobody waits for us meanwhile
                if (st_poll_ld(&sync[16 + i % S]) < (uint32_t)(i - S + 1)) {
                    if (landed < n) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed = n; if (lane == 0) st_flag_st(&sync[ld], (uint32_t) landed); }
                    st_wait_ge(&sync[16 + i % S], (uint32_t)(i - S + 1));
                }
            }
            st_dma_slot<NT, PPS>(gb, voff, ring_a + (uint32_t)(i % S)*SLOT);
            if (n >= DEPTH - 1) {      // all but the youngest DEPTH - 1 slots of this loader have landed
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1)*PPS) : "memory");
                if (landed < n - (DEPTH - 2)) { landed = n - (DEPTH - 2); if (lane == 0) st_flag_st(&sync[ld], (uint32_t) landed); }
            }
        }
#pragma unroll
        for (int d = DEPTH - 2; d >= 0; d--) {
            if (d == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3*PPS) : "memory");
            if (d == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2*PPS) : "memory");
            if (d == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1*PPS) : "memory");
            if (d == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (d <= 3 && n - d > landed) { landed = n - d; if (lane == 0) st_flag_st(&sync[ld], (uint32_t) landed); }
        }
        STAMP(1);
        return;
    }

    // ================= consumers =================
    {   // the activation image
        STAMP(1);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = threadIdx.x + i*ST_NC*64;
            if (q < (p.k >> 4)) { const int ib = q >> 4, c = q & 15; *(int4v *) (act + (size_t) ib*ST_ACT_STRIDE + c*16) = areg[i]; }
        }
        if ((int) threadIdx.x < nb) {
            const int ib = threadIdx.x;
            uint32_t hw[2] = { 0, 0 }, lw[2] = { 0, 0 };
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t wsum = (uint32_t)(j < 4 ? breg[0][j] : breg[1][j - 4]);
                const int s = (int)(int16_t)(wsum & 0xFFFF) + (int)(int16_t)(wsum >> 16);
                const int h = (s + 64) >> 7, l = s - (h << 7);
                hw[j >> 2] |= (uint32_t)(h & 0xFF) << (8*(j & 3));
                lw[j >> 2] |= (uint32_t)(l & 0xFF) << (8*(j & 3));
            }
            *(int4v *) (act + (size_t) ib*ST_ACT_STRIDE + 256) = int4v{ (int) hw[0], (int) hw[1], (int) lw[0], (int) lw[1] };
            dd[ib] = dreg;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) st_flag_add(&sync[2], 1u);
        st_wait_ge(&sync[2], ST_NC);
        STAMP(2);
    }
    const uint32_t magic = (uint32_t)((0x100000000ull + nb - 1)/nb);      // u / nb for u < 2^16
    bool first = true;
    for (int i = wave; i < nslots; i += ST_NC) {
        const int si = GLU ? (i >= ns1) : 0, il = i - si*ns1;
        const int u = il*64 + lane;                              // unit inside the stream
        const bool live = u < n1;
        const int uc = live ? u : n1 - 1;
        const int ib = uc - (int) __umulhi((uint32_t) uc, magic)*nb;
        // the block's activation first (it does not depend on the loader)
        int4v A[16];
        const char * ap = act + (size_t) ib*ST_ACT_STRIDE;
#pragma unroll
        for (int j = 0; j < 16; j++) A[j] = *(const int4v *) (ap + 16*j);
        const int4v HL = *(const int4v *) (ap + 256);
        const float d8 = dd[ib];
        if (MODE != 2) st_wait_ge(&sync[i % NL], (uint32_t)(i/NL + 1));
        if (first) { STAMP(3); first = false; }
        int4v c[(U::UB/16)];
        const char * sp = ring + (size_t)(i % S)*SLOT + (size_t)(live ? lane : 0)*U::UB;
#pragma unroll
        for (int j = 0; j < (U::UB/16); j++) c[j] = *(const int4v *) (sp + 16*j);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (nslots > S && lane == 0) st_flag_st(&sync[16 + i % S], (uint32_t)(i + 1));
        float res;
        if (MODE == 1) { int x = 0;
#pragma unroll
            for (int j = 0; j < (U::UB/16); j++) x ^= c[j].x ^ c[j].y ^ c[j].z ^ c[j].w;
#pragma unroll
            for (int j = 0; j < 16; j++) x ^= A[j].x ^ A[j].y ^ A[j].z ^ A[j].w;
            res = (float)(x ^ HL.x) + d8;
        } else { typename U::wfrag wf; for (int j = 0; j < (U::UB/16); j++) wf.c[j] = c[j]; res = U::dot(wf, ap, d8); }      // (the kernel's own unit arithmetic: mmvq_stream.h)
        if (!live) res = 0.0f;
        if (nb == 16) {        // 16 lanes = one row
            res = row16_sum(res);
            if ((lane & 15) == 0 && live) part[si*R + (u >> 4)] = res;
        } else if (live) part[si*n1 + u] = res;
    }
    STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) st_flag_add(&sync[3], 1u);
    st_wait_ge(&sync[3], ST_NC);
    STAMP(5);
    for (int rr = threadIdx.x; rr < R; rr += ST_NC*64) {
        float g = 0.0f, u = 0.0f;
        if (nb == 16) { g = part[rr]; if (GLU) u = part[R + rr]; }
        else {
            for (int i = 0; i < nb; i++) g += part[rr*nb + i];
            if (GLU) for (int i = 0; i < nb; i++) u += part[n1 + rr*nb + i];
        }
        if (GLU) g = (g/(1.0f + expf(-g)))*u;
        p.dst[r0 + rr] = g;
    }
    STAMP(6);
}

// ---- host ----
static float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
static uint16_t f32_to_f16(float f) { return __half_as_ushort(__float2half(f)); }
static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state*1664525u + 1013904223u; return rng_state >> 8; }
static void ref_row(const uint8_t * row, int nb, const int8_t * qs, const float * ad, double & out) {
    double acc = 0;
    for (int ib = 0; ib < nb; ib++) {
        const uint8_t * bl = row + (size_t) ib*144;
        const float d = f16_to_f32(*(const uint16_t *) bl), dmin = f16_to_f32(*(const uint16_t *) (bl + 2));
        const uint8_t * sc = bl + 4; const uint8_t * q = bl + 16;
        int isum = 0, msum = 0;
        for (int j = 0; j < 8; j++) {
            int s, m;
            if (j < 4) { s = sc[j] & 63; m = sc[j + 4] & 63; }
            else { s = (sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4); m = (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4); }
            const int g = j >> 1; int dot = 0, bsum = 0;
            for (int e = 0; e < 32; e++) { const int w = (j & 1) ? (q[32*g + e] >> 4) : (q[32*g + e] & 0xF); const int a = qs[ib*256 + 32*j + e]; dot += w*a; bsum += a; }
            isum += s*dot; msum += m*bsum;
        }
        acc += (double) d*ad[ib]*isum - (double) dmin*ad[ib]*msum;
    }
    out = acc;
}

int main(int argc, char ** argv) {
    const int stamps_on = argc > 1 ? atoi(argv[1]) : 1;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct shape { const char * name; int m, k; bool glu; } shapes[] = {
        { "gate/up/SwiGLU 2x14336x4096", 14336, 4096, true }, { "ffn_down 4096x14336", 4096, 14336, false }, { "wo 4096x4096", 4096, 4096, false },
        { "wq+wk+wv-like 6144x4096", 6144, 4096, false }, { "lm_head-like 128256x4096", 128256, 4096, false } };
    for (const shape & sh : shapes) {
        const int m = sh.m, k = sh.k, nb = k/256;
        const size_t wbytes = (size_t) m*nb*144, tb = wbytes*(sh.glu ? 2 : 1);
        std::vector<uint8_t> hw(tb);
        for (size_t i = 0; i < hw.size(); i += 4) *(uint32_t *) &hw[i] = rnd() ^ (rnd() << 12);
        for (size_t bidx = 0; bidx < hw.size()/144; bidx++) {
            *(uint16_t *) &hw[bidx*144] = f32_to_f16(0.001f + (rnd() % 1000)*1e-5f);
            *(uint16_t *) &hw[bidx*144 + 2] = f32_to_f16(0.001f + (rnd() % 1000)*1e-5f);
        }
        std::vector<int8_t> hq(k); std::vector<float> hd(nb); std::vector<int16_t> hbs(k/16);
        for (int i = 0; i < k; i++) hq[i] = (int8_t)((int)(rnd() % 255) - 127);
        for (int i = 0; i < nb; i++) hd[i] = 0.01f + (rnd() % 100)*1e-4f;
        for (int i = 0; i < k/16; i++) { int s = 0; for (int e = 0; e < 16; e++) s += hq[i*16 + e]; hbs[i] = (int16_t) s; }
        int nc = (int)((size_t) 640*1024*1024/tb) + 1; if (nc > 24) nc = 24;
        char * dW; int8_t * dq; float * dd; int16_t * dbs; float * dst; unsigned long long * dstamps;
        CK(hipMalloc(&dW, tb*nc + 4096)); CK(hipMalloc(&dq, k)); CK(hipMalloc(&dd, nb*4)); CK(hipMalloc(&dbs, k/16*2)); CK(hipMalloc(&dst, (size_t) m*4));
        CK(hipMalloc(&dstamps, 256*(ST_NC + 2)*8*8));
        for (int c = 0; c < nc; c++) CK(hipMemcpy(dW + (size_t) c*tb, hw.data(), tb, hipMemcpyHostToDevice));
        CK(hipMemcpy(dq, hq.data(), k, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), nb*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbs, hbs.data(), k/16*2, hipMemcpyHostToDevice));
        probe_args a; a.m = m; a.k = k; a.a_qs = dq; a.a_d = dd; a.a_bs = dbs; a.dst = dst; a.w_bytes = (long long) wbytes; a.stamps = nullptr;
        const int G = 256, Rmax = (m + G - 1)/G;
        const size_t fixed = ST_SYNC_BYTES + (size_t) nb*ST_ACT_STRIDE + ((nb + 3) & ~3)*4 + (size_t)(nb == 16 ? Rmax : Rmax*nb)*(sh.glu ? 2 : 1)*4 + 16;
        const int SLOT = 9*1024;
        int S = (int)((163840 - fixed)/SLOT); const int ns_max = ((Rmax*nb + 63)/64)*(sh.glu ? 2 : 1); if (S > ns_max) S = ns_max; if (S > 48) S = 48;
        a.nslot_ring = S;
        const size_t ldsb = fixed + (size_t) S*SLOT;
        for (int variant = 0; variant < 6; variant++) {      // loaders x slots in flight each: 0: 2x3  1: 2x3 nt  2: 2x5 nt  3: 1x7 nt  4: 2x3 nt no compute  5: 1x5 nt
            const void * kf; int nl = 2;
#define KF(G_, NT_, M_, NL_, D_) (const void *) k_stream_probe<G_, NT_, M_, NL_, D_>
            if (sh.glu) { const void * t[6] = { KF(true, false, 0, 2, 3), KF(true, true, 0, 2, 3), KF(true, true, 0, 2, 5), KF(true, true, 0, 1, 7), KF(true, true, 1, 2, 3), KF(true, true, 0, 1, 5) }; kf = t[variant]; }
            else        { const void * t[6] = { KF(false, false, 0, 2, 3), KF(false, true, 0, 2, 3), KF(false, true, 0, 2, 5), KF(false, true, 0, 1, 7), KF(false, true, 1, 2, 3), KF(false, true, 0, 1, 5) }; kf = t[variant]; }
            if (variant == 3 || variant == 5) nl = 1;
            CK(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));
            int li = 0;
            auto launch = [&]() { a.W = dW + (size_t)(li % nc)*tb; a.W2 = sh.glu ? a.W + wbytes : nullptr; li++; void * kargs[] = { (void *) &a }; CK(hipLaunchKernel(kf, dim3(G), dim3((ST_NC + nl)*64), kargs, ldsb, st)); };
            launch(); CK(hipStreamSynchronize(st));
            std::vector<float> out(m); CK(hipMemcpy(out.data(), dst, (size_t) m*4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0;
            for (int r = 0; r < m; r += (m > 20000 ? 997 : 61)) {
                double g, u = 0; ref_row(&hw[(size_t) r*nb*144], nb, hq.data(), hd.data(), g);
                if (sh.glu) { ref_row(&hw[wbytes + (size_t) r*nb*144], nb, hq.data(), hd.data(), u); g = g/(1.0 + exp(-g))*u; }
                maxerr = fmax(maxerr, fabs(g - out[r])); maxref = fmax(maxref, fabs(g));
            }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 20; float best = 1e9f;
            for (int t = 0; t < 5; t++) {
                CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; i++) launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms);
            }
            const double us = best*1000.0/reps;
            printf("%-30s v%d %7.2f us/launch %6.0f GB/s  ring %2d slots, LDS %6zu  err %.3g / %.3g\n", sh.name, variant, us, (double) tb/us/1e3, S, ldsb, maxerr, maxref);
            if (stamps_on && (variant == 1 || variant == 4)) {
                a.stamps = dstamps; CK(hipMemset(dstamps, 0, 256*(ST_NC + 2)*8*8)); launch(); CK(hipStreamSynchronize(st)); a.stamps = nullptr;
                std::vector<unsigned long long> hs(256*(ST_NC + 2)*8); CK(hipMemcpy(hs.data(), dstamps, hs.size()*8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull; for (int i = 0; i < 256*(ST_NC + 2); i++) if (hs[i*8]) t0 = hs[i*8] < t0 ? hs[i*8] : t0;
                const char * names[8] = { "entry", "(loader: all landed) image loads back", "image ready", "first slot landed", "last slot computed", "all consumers done", "exit", "" };
                for (int j = 0; j < 7; j++) {
                    std::vector<double> v, vl;
                    for (int i = 0; i < 256*(ST_NC + 2); i++) if (hs[i*8 + j]) { if (j == 1 && i % (ST_NC + 2) >= ST_NC) vl.push_back((hs[i*8 + j] - t0)*0.01); else v.push_back((hs[i*8 + j] - t0)*0.01); }
                    std::sort(v.begin(), v.end()); std::sort(vl.begin(), vl.end());
                    if (!v.empty()) printf("    %-40s min %6.2f  med %6.2f  max %6.2f us\n", names[j], v[0], v[v.size()/2], v.back());
                    if (!vl.empty()) printf("    %-40s min %6.2f  med %6.2f  max %6.2f us\n", "loader: everything landed", vl[0], vl[vl.size()/2], vl.back());
                }
            }
        }
        CK(hipFree(dstamps)); CK(hipFree(dW)); CK(hipFree(dq)); CK(hipFree(dd)); CK(hipFree(dbs)); CK(hipFree(dst));
    }
    return 0;
}
