// Probe (gfx950): does global_load_lds_dwordx4 reach LDS addresses at and beyond 64 KB (M0 = byte offset)?
// Usage: ./lds_dma_probe  -> prints, per destination offset, where the 1 KB piece landed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(64) probe(const uint32_t* src, uint32_t dst_off, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s32 = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) s32[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t m = __builtin_amdgcn_readfirstlane(lds0 + dst_off);
    const uint32_t voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(m) : "memory", "m0");
    __syncthreads();
    // find where word 0 of the source (0x1000) landed
    int found = -1, count = 0;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64)
        if (s32[i] != 0xdeadbeefu) { atomicAdd((int*)&out[1], 1); if (s32[i] == 0x1000u) out[0] = i * 4; }
    (void)found; (void)count;
}
int main() {
    uint32_t h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    uint32_t *d, *o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const uint32_t offs[] = {0, 1024, 33280, 65536 - 1024, 65536, 66048, 98304, 131072, 160 * 1024 - 1024};
    for (uint32_t off : offs) {
        uint32_t z[2] = {0xffffffffu, 0};
        hipMemcpy(o, z, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, off, o);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(z, o, 8, hipMemcpyDeviceToHost);
        printf("dst %6u: %s  landed at %d (%u words changed)\n", off, hipGetErrorString(e), (int)z[0], z[1]);
    }
    return 0;
}
