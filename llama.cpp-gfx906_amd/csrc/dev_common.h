// dev_common.h — device-side helpers for gfx950 (CDNA4): wave64 DPP reductions,
// f16 conversion, integer dot, loads of arbitrarily aligned block bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#define MI_WAVE 64

#define MI_HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: a process that drives several GPUs (a layer split or a row split held in
// one process) must raise a kernel's dynamic-LDS limit once per device, not once per process. Returns false (and clears the error) on failure.
#define MI_LDS_LIMIT(bytes, ...) ([&]() -> bool { static std::atomic<uint32_t> done_{ 0 }; int d_ = 0; (void) hipGetDevice(&d_); \
    if (done_.load(std::memory_order_acquire) >> (d_ & 31) & 1) return true; \
    if (hipFuncSetAttribute((const void *) (__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) { (void) hipGetLastError(); return false; } \
    done_.fetch_or(1u << (d_ & 31), std::memory_order_release); return true; }())
#define MI_LDS_LIMIT_OR_DIE(bytes, ...) do { if (!MI_LDS_LIMIT(bytes, __VA_ARGS__)) { fprintf(stderr, "hipFuncSetAttribute failed at %s:%d\n", __FILE__, __LINE__); abort(); } } while (0)

namespace mi355x {

typedef int   int4v   __attribute__((ext_vector_type(4)));
typedef int   int2v   __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// ---- wave64 reductions: DPP inside a 16-lane row, readlane across the 4 rows ---------------
template <int CTRL>
static __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
static __device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// sum over the 16 lanes of each DPP row; every lane of the row ends with the row total
static __device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}
static __device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
static __device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// full-wave sum; the result is wave-uniform (held in SGPR-fed adds)
static __device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
static __device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// sum over aligned groups of 8 lanes (half a DPP row)
static __device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v;
}
static __device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    return v;
}
static __device__ __forceinline__ int group8_sum_i(int v) {
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    return v;
}

// ---- numerics ----------------------------------------------------------------------------------
static __device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
static __device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    return __half_as_ushort(__float2half_rn(f));
}
// gguf-py/gguf/quants.py:663-665
static __device__ __forceinline__ float e8m0_to_f32_half(uint32_t x) {
    const uint32_t bits = x < 2 ? (0x00200000u << x) : ((x - 1) << 23);
    return __builtin_bit_cast(float, bits);
}

// 4 x int8 dot with int32 accumulate: v_dot4_i32_i8
static __device__ __forceinline__ int dot4(int a, int b, int c) {
    return __builtin_amdgcn_sdot4(a, b, c, false);
}

// ---- loads --------------------------------------------------------------------------------------
// Block bytes in raw GGUF layout are only 1- or 2-byte aligned for Q4_0/Q8_0/Q6_K/MXFP4
// (18/34/210/17-byte blocks). gfx950 global memory is accessed in unaligned mode, so the packed
// vector types below lower to single global_load_dwordx{1,2,4} instructions at any byte address.
struct __attribute__((packed, aligned(1))) u32_u { uint32_t v; };
struct __attribute__((packed, aligned(1))) u32x2_u { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) u32x4_u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) u16_u { uint16_t v; };

static __device__ __forceinline__ uint32_t ld_u32(const void * p)  { return ((const u32_u *) p)->v; }
static __device__ __forceinline__ uint16_t ld_u16(const void * p)  { return ((const u16_u *) p)->v; }
static __device__ __forceinline__ int4v ld_b128(const void * p) {
    const u32x4_u t = *(const u32x4_u *) p;
    return int4v{ (int) t.x, (int) t.y, (int) t.z, (int) t.w };
}
static __device__ __forceinline__ int2v ld_b64(const void * p) {
    const u32x2_u t = *(const u32x2_u *) p;
    return int2v{ (int) t.x, (int) t.y };
}
// 8 bytes at a 2-byte-aligned address as ONE dword-aligned 12-byte load + a per-lane byte funnel shift (v_alignbyte_b32): a
// global_load_dwordx2 at 2 or 6 bytes past a dword boundary is split by the address unit into several accesses per lane (Q6_K's
// 210-byte blocks put three quarters of their fragments there: tools/stamp_timeline.py showed the workgroups of a Q6_K group issue
// their first weight step 2 us slower than their Q4_K neighbours)
struct __attribute__((packed, aligned(4))) u32x3_a { uint32_t x, y, z; };
static __device__ __forceinline__ int2v ld_b64_a2(const void * p) {
    const uintptr_t a = (uintptr_t) p;
    const u32x3_a t = *(const u32x3_a *) (a & ~(uintptr_t) 3);
    const uint32_t sh = (uint32_t)(a & 3);
    return int2v{ (int) __builtin_amdgcn_alignbyte(t.y, t.x, sh), (int) __builtin_amdgcn_alignbyte(t.z, t.y, sh) };
}
// 16 bytes at a 2-byte-aligned address from dword-aligned loads: five dwords (b128 + b32) and four v_alignbyte
struct __attribute__((packed, aligned(4))) u32x4_a { uint32_t x, y, z, w; };
static __device__ __forceinline__ int4v ld_b128_a2(const void * p) {
    const uintptr_t a = (uintptr_t) p;
    const uintptr_t base = a & ~(uintptr_t) 3;
    const u32x4_a t = *(const u32x4_a *) base;
    const uint32_t t4 = *(const uint32_t *) (base + 16);
    const uint32_t sh = (uint32_t)(a & 3);
    return int4v{ (int) __builtin_amdgcn_alignbyte(t.y, t.x, sh), (int) __builtin_amdgcn_alignbyte(t.z, t.y, sh),
                  (int) __builtin_amdgcn_alignbyte(t.w, t.z, sh), (int) __builtin_amdgcn_alignbyte(t4, t.w, sh) };
}
// 16-byte aligned weight bytes. Measured on the tg128 bench (profiles/r01_*): plain loads 500 tok/s vs nontemporal 482,
// so plain is the default; -DMI_NT_WEIGHTS switches the streamed-once hint back on for experiments.
static __device__ __forceinline__ int4v ld_b128_nt(const void * p) {
#ifdef MI_NT_WEIGHTS
    return __builtin_nontemporal_load((const int4v *) p);
#else
    return *(const int4v *) p;
#endif
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float  f32x2_t  __attribute__((ext_vector_type(2)));
// two floats -> two bf16 in one word: one v_cvt_pk_bf16_f32 (round to nearest even), not the ~8 integer operations of the bit formula
static __device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{ a, b }, bf16x2_t));
}

} // namespace mi355x
