// PROTOTYPE (not part of the product, not built by __graft_entry__.build()): the loop structure DESIGN.md section 9 plans for the
// channel GEMMs, as a plain bf16 GEMM  C[M,N] = A[M,K] . W[N,K]^T  (fp32 accumulate, bf16 out) so that it can be timed against the
// production kernel on the fat shapes of the B=128 step:
//   256x256 tile, 8 waves (2x4, 128x64 each), BK = 64, TWO LDS stages, operands loaded straight to LDS with
//   global_load_lds_dwordx4 (16 bytes per lane, 1 KiB contiguous per wave-instruction = 8 tile rows), XOR swizzle applied to
//   the SOURCE address (slot = chunk ^ ((row >> 1) & 7)) and undone by the fragment reads, tile t+1 in flight while tile t
//   multiplies, one barrier per K tile, LDS-staged 16-byte-store epilogue.
// Build:  hipcc --offload-arch=gfx950 -O3 gemm256.hip -o gemm256        Run: ./gemm256   (self-check against a naive kernel, then timings)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;

static inline bf16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static inline float bf2f_host(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return *(bf16_t*)&b; }

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = BK * 2;                  // 128 bytes per tile row
constexpr int TILE_BYTES = 256 * ROWB;        // 32 KiB per operand and stage
constexpr int LDS_BYTES = 4 * TILE_BYTES;     // 2 stages x (A | W) = 128 KiB

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// 16 bytes per lane straight into LDS: lane l lands at m0 + 16*l.  Inline asm on purpose: with the builtin
// (__builtin_amdgcn_global_load_lds) hipcc treats the DMA as an LDS write that every later ds_read may alias and puts an
// s_waitcnt vmcnt(0) in front of the fragment reads of the CURRENT tile -- the prefetch of the next tile would never overlap them.
__device__ __forceinline__ void glds16(const bf16_t* g, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "memory", "m0");
}

__global__ void __launch_bounds__(512, 2) gemm256_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                         bf16_t* __restrict__ C, int ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;             // LDS byte offset of the dynamic segment (low half of the generic address)
    const int tilesN = N / BN, tilesM = (M + BM - 1) / BM;
    const int lb = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int mt = lb / tilesN, nt = lb - mt * tilesN;
    const int m0 = mt * BM, n0 = nt * BN;

    // staging: wave w, instruction i covers tile rows (w*4 + i)*8 .. +7; lane = (row r8 = lane/8, LDS slot s = lane%8)
    const int r8 = lane >> 3, s8 = lane & 7;
    const bf16_t* gA[4];
    const bf16_t* gW[4];
    int ldsoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + r8;
        const int chunk = s8 ^ ((row >> 1) & 7);                    // swizzle on the source: slot s holds chunk s ^ key(row)
        const int am = m0 + row < M ? m0 + row : M - 1;              // rows past M: clamped (never stored)
        gA[i] = A + (long)am * lda + chunk * 8;
        gW[i] = W + (long)(n0 + row) * ldw + chunk * 8;
        ldsoff[i] = (w * 4 + i) * 8 * ROWB;                          // wave-uniform base of the 1 KiB piece
    }
    auto issue = [&](int t, int stage) {
        const uint32_t sA = lds0 + stage * 2 * TILE_BYTES, sW = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(gA[i] + t * BK, __builtin_amdgcn_readfirstlane(sA + ldsoff[i]));
            glds16(gW[i] + t * BK, __builtin_amdgcn_readfirstlane(sW + ldsoff[i]));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addresses (byte offsets inside an operand tile), swizzle undone: slot = chunk ^ key(row)
    int offA[4], offB[2], keyA[4], keyB[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { const int row = wr * 128 + mi * 32 + li; offA[mi] = row * ROWB; keyA[mi] = (row >> 1) & 7; }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { const int row = wc * 64 + ni * 32 + li; offB[ni] = row * ROWB; keyB[ni] = (row >> 1) & 7; }

    const int ntile = K / BK;
    issue(0, 0);
    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile t have landed
        __syncthreads();                                      // everybody's have, and everybody is done reading stage (t+1)&1
        if (t + 1 < ntile) issue(t + 1, (t + 1) & 1);
        const unsigned char* sA = smem + (t & 1) * 2 * TILE_BYTES;
        const unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            union { uint4 u; s16x8 s; } fa[4], fb[2];
            const int chunk = kc * 2 + lh;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi].u = *(const uint4*)(sA + offA[mi] + ((chunk ^ keyA[mi]) << 4));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[ni].u = *(const uint4*)(sW + offB[ni] + ((chunk ^ keyB[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi].s, fb[ni].s, acc[mi][ni], 0, 0, 0);
        }
    }
    // epilogue: acc -> bf16 tile in LDS ([256][256], 512-byte rows) -> 16-byte coalesced stores
    __syncthreads();
    bf16_t* sC = (bf16_t*)smem;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wc * 64 + ni * 32 + li;
                sC[row * BN + col] = f2bf(acc[mi][ni][r]);
            }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int idx = i * 512 + tid;
        const int row = idx >> 5, chunk = idx & 31;
        if (m0 + row < M) *(uint4*)(C + (long)(m0 + row) * ldc + n0 + chunk * 8) = *(const uint4*)(sC + row * BN + chunk * 8);
    }
}

__global__ void naive_kernel(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += bf2f(A[(long)m * lda + k]) * bf2f(W[(long)n * ldw + k]);
    C[(long)m * ldc + n] = acc;
}

static void fill(std::vector<bf16_t>& v, unsigned seed) {
    for (size_t i = 0; i < v.size(); ++i) { seed = seed * 1664525u + 1013904223u; v[i] = f2bf_host(((seed >> 9) & 0xffff) / 65536.f - 0.5f); }
}

static int run(int M, int N, int K, bool check) {
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    fill(hA, 1); fill(hW, 2);
    bf16_t *dA, *dW, *dC; float* dR = nullptr;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, (size_t)M * N * 2);
    hipFuncSetAttribute((const void*)gemm256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipLaunchKernelGGL(gemm256_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, dA, K, dW, K, dC, N, M, N, K);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    if (check) {
        hipMalloc(&dR, (size_t)M * N * 4);
        hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, dA, K, dW, K, dR, N, M, N, K);
        std::vector<float> hR((size_t)M * N); std::vector<bf16_t> hC((size_t)M * N);
        hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0, ref = 0; size_t bad = 0;
        for (size_t i = 0; i < hR.size(); ++i) {
            double d = fabs((double)bf2f_host(hC[i]) - hR[i]);
            if (d > worst) worst = d;
            if (fabs(hR[i]) > ref) ref = fabs(hR[i]);
            if (d > 0.02 * (fabs(hR[i]) + 1.0)) ++bad;
        }
        printf("check M=%d N=%d K=%d: max abs err %.4f (max |ref| %.2f), %zu elements out of tolerance\n", M, N, K, worst, ref, bad);
        hipFree(dR);
        if (bad) return 2;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm256_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, dA, K, dW, K, dC, N, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("gemm256 M=%6d N=%5d K=%5d  blocks %5d  %8.1f us  %7.1f TF/s\n", M, N, K, grid, us, 2.0 * M * N * K / us / 1e6);
    hipFree(dA); hipFree(dW); hipFree(dC);
    return 0;
}

int main() {
    if (run(700, 512, 256, true)) return 1;          // M tail + two N tiles + 4 K tiles
    if (run(2176, 1024, 1536, true)) return 1;
    run(41344, 512, 768, false);                     // G4 s1 (production kernel: 88-102 us, vendor GEMM 38 us)
    run(41344, 1280, 256, false);                    // ~G1 s1 (N = 1288 in the model)
    run(41344, 256, 512, false);                     // dG4 s1
    run(54400, 256, 384, false);                     // G4 s0
    run(4096, 4096, 4096, false);
    run(8192, 8192, 8192, false);
    return 0;
}
