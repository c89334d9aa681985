// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs?  (question behind an fp16 hi/lo split mode: the lo part of a value
// below ~0.1 is an fp16 subnormal.)  A = 2^-20 (subnormal), B = 2^10: every product is 2^-10, K = 16 -> D = 2^-6 unless flushed.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(float* D, float a_val, float b_val) {
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { D[0] = acc[0]; D[1] = (float)a[0]; }
}
int main() {
    float* D; hipMalloc(&D, 8);
    const float cases[3][2] = {{9.5367431640625e-07f, 1024.f}, {6.103515625e-05f, 1024.f}, {5.9604644775390625e-08f, 32768.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, D, c[0], c[1]);
        float h[2]; hipMemcpy(h, D, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as f16: %g) b=%g  D=%g  expected %g\n", c[0], h[1], c[1], h[0], 16.0 * c[0] * c[1]);
    }
    return 0;
}
