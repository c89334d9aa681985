// Shared device helpers for the gfx950 GAST-Net kernels (wave = 64 lanes; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gast_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;  // raw bits of the 16-bit STORAGE type of this build (GAST_BF16): bfloat16 -- or IEEE binary16, see below

// hardware conversions (gfx950): v_cvt_pk_bf16_f32, round to nearest even.  pack_bf16x2 is ALWAYS bfloat16: the hi/lo operand pairs of
// the GAST_F32X3 arithmetic are built with it whatever the storage flavour of the build.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return *(uint32_t*)&v;
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {      // v_cvt_pk_f16_f32, round to nearest even, subnormals kept
    f32x2_t v = {lo, hi};
    f16x2_t h = __builtin_convertvector(v, f16x2_t);
    return *(uint32_t*)&h;
}

// ---------------------------------------------------------------- the 16-bit storage flavour of the build
// The sources are compiled twice (build.sh): libgast_hip.so stores GAST_BF16 tensors as bfloat16 (8 significand bits, fp32's range),
// libgast_hip_f16.so (-DGAST_H16_F16) as IEEE binary16 (11 significand bits, |x| <= 65504: GAST_HIP_DTYPE=f16, the 16-bit mode that
// meets the north star's 1e-2 bound; gradients travel multiplied by a power-of-two loss scale).  Same ABI, same kernels: only the
// four conversions below and the matrix instruction of the 16-bit operands (mfma_h16) differ.
//   h16_to_f / f_to_h16: one value;  h16x2_unpack / pack_h16x2: the two halves of a 32-bit word (low half = lower address)
#ifdef GAST_H16_F16
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)*(const _Float16*)&h; }
__device__ __forceinline__ bf16_t f2bf(float f) { const _Float16 h = (_Float16)f; return *(const bf16_t*)&h; }
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) { return pack_f16x2(lo, hi); }
__device__ __forceinline__ void h16x2_unpack(uint32_t w, float& lo, float& hi) {
    const f16x2_t h = *(const f16x2_t*)&w;
    lo = (float)h.x;
    hi = (float)h.y;
}
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *(bf16_t*)&b;
}
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ void h16x2_unpack(uint32_t w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
}
#endif

// fp16 pairs (GAST_F32X3H): pack_f16x2 above; subnormal results are kept (the f16 MFMA honours them:
// scripts/toolchain_smoke/f16_denorm_probe.hip)
// x = hi + lo, both halves as packed pairs: PAIR = 1 bf16 (GAST_F32X3), 2 fp16 (GAST_F32X3H).  x - hi is exact in fp32.
template <int PAIR>
__device__ __forceinline__ void split_pair4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
    if (PAIR == 2) {
        hi.x = pack_f16x2(x0, x1);
        hi.y = pack_f16x2(x2, x3);
        const f16x2_t a = *(const f16x2_t*)&hi.x, b = *(const f16x2_t*)&hi.y;
        lo.x = pack_f16x2(x0 - (float)a.x, x1 - (float)a.y);
        lo.y = pack_f16x2(x2 - (float)b.x, x3 - (float)b.y);
    } else {
        hi.x = pack_bf16x2(x0, x1);
        hi.y = pack_bf16x2(x2, x3);
        lo.x = pack_bf16x2(x0 - __uint_as_float(hi.x << 16), x1 - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack_bf16x2(x2 - __uint_as_float(hi.y << 16), x3 - __uint_as_float(hi.y & 0xffff0000u));
    }
}
// one 32x32x16 MFMA step on packed pairs held as 8 x 16 bit
template <int PAIR>
__device__ __forceinline__ f32x16 mfma_pair(const uint4& a, const uint4& b, const f32x16& c) {
    if (PAIR == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)&a, *(const f16x8*)&b, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const s16x8*)&a, *(const s16x8*)&b, c, 0, 0, 0);
}
// one 32x32x16 MFMA step on 8 values of the 16-bit STORAGE type per lane and operand
__device__ __forceinline__ f32x16 mfma_h16(const uint4& a, const uint4& b, const f32x16& c) {
#ifdef GAST_H16_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)&a, *(const f16x8*)&b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const s16x8*)&a, *(const s16x8*)&b, c, 0, 0, 0);
#endif
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// 4 consecutive elements <-> float4 (fp32: one 16-byte access, bf16: one 8-byte access)
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 u = *(const uint2*)p;
    float4 v;
    h16x2_unpack(u.x, v.x, v.y);
    h16x2_unpack(u.y, v.z, v.w);
    return v;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = pack_h16x2(v.x, v.y);
    u.y = pack_h16x2(v.z, v.w);
    *(uint2*)p = u;
}
__device__ __forceinline__ float4 rnd4(float4 v, const float*) { return v; }
__device__ __forceinline__ float4 rnd4(float4 v, const bf16_t*) {
    return make_float4(bf2f(f2bf(v.x)), bf2f(f2bf(v.y)), bf2f(f2bf(v.z)), bf2f(f2bf(v.w)));
}

// ---------------------------------------------------------------- dropout stream (see gast_dropout in gast_hip.h)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_key(const gast_dropout& d, uint32_t salt) {
    uint32_t seed = d.seed ? *d.seed : 0u;
    return seed * 0x9E3779B9u + salt * 0x85EBCA6Bu;
}
// multiplier (0 or inv_keep) for the element at linear element offset e of its tensor
__device__ __forceinline__ float drop_mul(uint32_t key, uint32_t thresh, float inv_keep, uint32_t e) {
    uint32_t h = hash32((e >> 1) ^ key);
    uint32_t bits = (e & 1u) ? (h >> 16) : (h & 0xffffu);
    return bits >= thresh ? inv_keep : 0.f;
}

// ---------------------------------------------------------------- row maps
// m in [0, B*Tn*J) -> (b, t, j);  mapped row or -1
__device__ __forceinline__ long map_row(const gast_rowmap& mp, int b, int t, int j, int J) {
    int ts = t * mp.t_stride + mp.t_off;
    if (ts < 0 || ts >= mp.T_total) return -1;
    return ((long)b * mp.T_total + ts) * J + j;
}

// ---------------------------------------------------------------- XCD-aware block remap (bijective; 8 XCDs)
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b%8).  Give each XCD a contiguous chunk of the
// logical tile order so that tiles sharing an operand panel hit the same L2 (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// ---------------------------------------------------------------- asynchronous 16-byte global loads
// Inline asm so that (a) the destination is an early-clobber tuple the allocator cannot alias with the address, (b) no
// exec-masked branch / compiler-inserted s_waitcnt separates consecutive loads.  The compiler does not count these loads:
// gload_wait_n<N>() (s_waitcnt vmcnt(N)) must precede the first use of the destination registers.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void gload_wait_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The wait does not NAME the destination registers, so by itself it does not stop the compiler from moving register-only uses of them
// above it (round 5: hipcc did exactly that in a gemm_big build whose conversion had no other anchor -- scripts/asm_load_hazard.py
// --strict finds such code).  gload_pin(r) directly behind the wait makes every later use of r depend on a statement that volatile-asm
// ordering keeps behind the wait; it emits no instruction.
__device__ __forceinline__ void gload_pin(u32x4& r) { asm volatile("" : "+v"(r)); }

// ---------------------------------------------------------------- GAST_DETERMINISTIC=1 (host side, read once per process)
// Run-to-run bit-reproducible results: every reduction whose summation order depends on block scheduling is replaced by one with a
// fixed order -- no split-K (gemm.hip: the finish pass adds its column statistics with atomics), no split-M in the weight gradients
// (wgrad*.hip: partial tiles meet in dW through atomics), the expand-conv backward's parameter epilogue in one block per input
// feature (norm_ops.hip).  Slower (the M = B*J stage and the weight gradients lose their parallel slack): a test / debugging mode,
// see DESIGN.md section 5.  Covers the fp32-storage plans (fp32, bf16x3); the 16-bit flavours keep LDS float atomics in the ELL
// aggregation backward.
#include <stdlib.h>
static inline bool gast_deterministic() {
    static const int v = getenv("GAST_DETERMINISTIC") ? atoi(getenv("GAST_DETERMINISTIC")) : 0;
    return v != 0;
}

#define GAST_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
