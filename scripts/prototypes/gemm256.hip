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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// Variant with the production kernel's A-side prologue: y = relu(a * scale[k] + shift[k]) applied on load.  W still goes straight
// to LDS by DMA; A goes global -> registers (ONE set, tile t+2 in flight while tile t multiplies) -> prologue -> swizzled
// ds_write_b128 into the stage that tile t+1 will be read from, right after the barrier that ends tile t-1's reads.
__global__ void __launch_bounds__(512, 2) gemm256_pro_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             bf16_t* __restrict__ C, int ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int tilesN = N / BN, tilesM = (M + BM - 1) / BM;
    const int lb = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int mt = lb / tilesN, nt = lb - mt * tilesN;
    const int m0 = mt * BM, n0 = nt * BN;
    const int r8 = lane >> 3, s8 = lane & 7;
    const bf16_t* gA[4];
    const bf16_t* gW[4];
    int ldsoff[4], achunk[4], arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + r8;
        const int chunk = s8 ^ ((row >> 1) & 7);
        const int am = m0 + row < M ? m0 + row : M - 1;
        gA[i] = A + (long)am * lda + chunk * 8;
        gW[i] = W + (long)(n0 + row) * ldw + chunk * 8;
        ldsoff[i] = (w * 4 + i) * 8 * ROWB;
        achunk[i] = chunk;
        arow[i] = row;
    }
    // scale / shift of all K channels in LDS (behind the two stages): the per-tile prologue must not issue global loads
    float* sSc = (float*)(smem + LDS_BYTES);
    float* sSh = sSc + K;
    for (int k = tid; k < K; k += 512) { sSc[k] = scale[k]; sSh[k] = shift[k]; }
    __syncthreads();
    // the A register set.  Loaded with inline asm: a compiler-tracked load gets an s_waitcnt vmcnt(2) right after the issue
    // (hipcc copies two of the destination registers to reuse them as temporaries in the MFMA block), and that wait also drains
    // the W DMA issued just before.  The explicit s_waitcnt vmcnt(0) at the top of the loop covers these loads.
    u32x4 ra[4];
    auto load_a = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra[i]) : "v"(gA[i] + t * BK) : "memory");
    };
    auto write_a = [&](int t, int stage) {              // prologue + swizzled store (slot s8 of row arow[i] holds chunk achunk[i])
        unsigned char* sA = smem + stage * 2 * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = t * BK + achunk[i] * 8;
            const float4 s0 = *(const float4*)(sSc + k), s1 = *(const float4*)(sSc + k + 4);        // LDS table: a global load here
            const float4 h0 = *(const float4*)(sSh + k), h1 = *(const float4*)(sSh + k + 4);        // is waited for on the spot
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            uint32_t wv[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float lo = __uint_as_float(wv[p] << 16), hi = __uint_as_float(wv[p] & 0xffff0000u);
                lo = fmaxf(fmaf(lo, sc[2 * p], sh[2 * p]), 0.f);
                hi = fmaxf(fmaf(hi, sc[2 * p + 1], sh[2 * p + 1]), 0.f);
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                bf16x2_t v = {(__bf16)lo, (__bf16)hi};
                wv[p] = *(uint32_t*)&v;
            }
            *(uint4*)(sA + arow[i] * ROWB + s8 * 16) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    };
    auto issue_w = [&](int t, int stage) {
        const uint32_t sW = lds0 + stage * 2 * TILE_BYTES + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(gW[i] + t * BK, __builtin_amdgcn_readfirstlane(sW + ldsoff[i]));
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    int offA[4], offB[2], keyA[4], keyB[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { const int row = wr * 128 + mi * 32 + li; offA[mi] = row * ROWB; keyA[mi] = (row >> 1) & 7; }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { const int row = wc * 64 + ni * 32 + li; offB[ni] = row * ROWB; keyB[ni] = (row >> 1) & 7; }

    const int ntile = K / BK;
    // prologue of the pipeline: tile 0 complete in stage 0, A registers hold tile 1
    issue_w(0, 0);
    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    write_a(0, 0);
    if (ntile > 1) load_a(1);
    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // W(t) landed (and the A registers of tile t+1)
        __syncthreads();                                      // stage t&1 complete and visible; stage (t+1)&1 free
        if (t + 1 < ntile) {
            // order matters: the compiler guards the first use of `ra` with its own s_waitcnt vmcnt(0) -- harmless here (nothing is
            // outstanding right after the wait above), but AFTER the DMA issue it would drain W(t+1) on the spot
            write_a(t + 1, (t + 1) & 1);
            issue_w(t + 1, (t + 1) & 1);
            if (t + 2 < ntile) load_a(t + 2);
        }
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

__global__ void naive_kernel(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, int M, int N, int K,
                             const float* scale, const float* shift) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        float a = bf2f(A[(long)m * lda + k]);
        if (scale) a = bf2f(f2bf(fmaxf(fmaf(a, scale[k], shift[k]), 0.f)));
        acc += a * bf2f(W[(long)n * ldw + k]);
    }
    C[(long)m * ldc + n] = acc;
}

static void fill(std::vector<bf16_t>& v, unsigned seed) {
    for (size_t i = 0; i < v.size(); ++i) { seed = seed * 1664525u + 1013904223u; v[i] = f2bf_host(((seed >> 9) & 0xffff) / 65536.f - 0.5f); }
}

static int run(int M, int N, int K, bool check, bool pro = false) {
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    fill(hA, 1); fill(hW, 2);
    bf16_t *dA, *dW, *dC; float* dR = nullptr;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    const int grid_ = ((M + BM - 1) / BM) * (N / BN);
    hipMemset(dC, 0, (size_t)M * N * 2);
    float *dS = nullptr, *dH = nullptr;
    if (pro) {
        std::vector<float> hs(K), hh(K);
        for (int k = 0; k < K; ++k) { hs[k] = 0.5f + (k % 7) * 0.25f; hh[k] = ((k % 5) - 2) * 0.05f; }
        hipMalloc(&dS, K * 4); hipMalloc(&dH, K * 4);
        hipMemcpy(dS, hs.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(dH, hh.data(), K * 4, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)gemm256_pro_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 2 * K * 4);
    }
    auto launch = [&]() {
        if (pro) hipLaunchKernelGGL(gemm256_pro_kernel, dim3(grid_), dim3(512), LDS_BYTES + 2 * K * 4, 0, dA, K, dW, K, dS, dH, dC, N, M, N, K);
        else hipLaunchKernelGGL(gemm256_kernel, dim3(grid_), dim3(512), LDS_BYTES, 0, dA, K, dW, K, dC, N, M, N, K);
    };
    hipFuncSetAttribute((const void*)gemm256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    if (check) {
        hipMalloc(&dR, (size_t)M * N * 4);
        hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, dA, K, dW, K, dR, N, M, N, K, dS, dH);
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
        printf("check%s M=%d N=%d K=%d: max abs err %.4f (max |ref| %.2f), %zu elements out of tolerance\n", pro ? " (prologue)" : "", M, N, K, worst, ref, bad);
        hipFree(dR);
        if (bad) return 2;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("gemm256%s M=%6d N=%5d K=%5d  blocks %5d  %8.1f us  %7.1f TF/s\n", pro ? "+prologue" : "", M, N, K, grid, us, 2.0 * M * N * K / us / 1e6);
    hipFree(dA); hipFree(dW); hipFree(dC);
    return 0;
}

int main() {
    if (run(700, 512, 256, true)) return 1;          // M tail + two N tiles + 4 K tiles
    if (run(2176, 1024, 1536, true)) return 1;
    if (run(700, 512, 256, true, true)) return 1;
    if (run(2176, 1024, 1536, true, true)) return 1;
    run(41344, 512, 768, false, true);
    run(41344, 256, 512, false, true);
    run(41344, 512, 768, false);                     // G4 s1 (production kernel: 88-102 us, vendor GEMM 38 us)
    run(41344, 1280, 256, false);                    // ~G1 s1 (N = 1288 in the model)
    run(41344, 256, 512, false);                     // dG4 s1
    run(54400, 256, 384, false);                     // G4 s0
    run(4096, 4096, 4096, false);
    run(8192, 8192, 8192, false);
    return 0;
}
