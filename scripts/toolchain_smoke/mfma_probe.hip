// Toolchain + MFMA fragment-layout probe (one wave).  D[32][32] = A[32][K] * Bt[32][K]^T.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ void probe_f32(const float* A, const float* Bt, float* D, int K) {
    int l = threadIdx.x, li = l & 31, lh = l >> 5;
    f32x16 acc = {0};
    for (int kc = 0; kc < K / 8; ++kc) {
        float4 a = *(const float4*)(A + li * K + kc * 8 + lh * 4);
        float4 b = *(const float4*)(Bt + li * K + kc * 8 + lh * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        D[row * 32 + li] = acc[r];
    }
}

__global__ void probe_bf16(const uint16_t* A, const uint16_t* Bt, float* D, int K) {
    int l = threadIdx.x, li = l & 31, lh = l >> 5;
    f32x16 acc = {0};
    for (int kc = 0; kc < K / 16; ++kc) {
        s16x8 a = *(const s16x8*)(A + li * K + kc * 16 + lh * 8);
        s16x8 b = *(const s16x8*)(Bt + li * K + kc * 16 + lh * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        D[row * 32 + li] = acc[r];
    }
}

extern "C" int probe_launch(int which, const void* A, const void* Bt, float* D, int K, hipStream_t s) {
    if (which == 0) hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, s, (const float*)A, (const float*)Bt, D, K);
    else hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, s, (const uint16_t*)A, (const uint16_t*)Bt, D, K);
    return (int)hipGetLastError();
}
extern "C" int probe_memset(void* p, size_t n, hipStream_t s) { return (int)hipMemsetAsync(p, 0, n, s); }
