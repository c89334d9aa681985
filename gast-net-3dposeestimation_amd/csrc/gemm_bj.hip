// gast_gemm, the M = B*J stage (few rows, long K, fat N) for GAST_F32X3 / GAST_F32X3H (fp32 storage, split 16-bit products), gfx950.
//
// Same contract as gemm.hip / gemm_big.hip (K segments with row maps = channel concat / temporal taps of reference gast_net.py:28-32,
// 145-149,173-176; BN+ReLU load prologue; STATS / BNRELU_BWD epilogues).  The last stage of the network has M = B*J rows (2 176 at
// B = 128: 17 row tiles of 128) and K up to 3 592: too few output tiles to fill 256 CUs, and a K loop that is one long dependent chain
// per block.  Rounds 1-5 ran it on gemm.hip's 128 x 128 kernel with a CROSS-block split-K (fp32 partial tiles out to a workspace and
// back through a finish launch: 4.3 x the algorithmic bytes, two launches per GEMM, 0.08 of the HBM roof).  Here the split is INSIDE
// the block:
//   * block tile 64 x 64 (NJ = 1) or 64 x 128 (NJ = 2), 512 threads = 8 waves = 2 k-groups x (2 x 2) waves of 32 x 32 NJ; a K step
//     covers 32 values, k-group g multiplies the 16-deep half g of every step -- each wave's dependent MFMA chain is half as long, 34
//     row tiles x N / 64 blocks fill the chip without a workspace (272 blocks for N = 512), two blocks = 4 waves per SIMD cover each
//     other's barrier / LDS / VALU phases;
//   * operands as in gemm_big.hip: weights stream global -> LDS by DMA (global_load_lds_dwordx4) from the pre-split k-group-major
//     image (gast_x3_image_multi) into a ring of three stages, activations pass through registers (BN+ReLU prologue, hi/lo split) into
//     two LDS stages, prefetch distance two, ONE counted s_waitcnt and one barrier per K step; 64-byte row images [16 hi | 16 lo]
//     with the same XOR swizzle;
//   * the two k-groups' accumulators meet in LDS after the loop (each group keeps the rows it then finishes: half of the epilogue
//     per wave), in a fixed order: no atomics on the output, results are run-to-run reproducible;
//   * branch-free buffer-addressed epilogue straight from the accumulators; the column statistics of a 64-row block are ADDED
//     into the 128-row statistics block of the shared layout partials[ceil(M/128)][N][2] (two blocks per row, a + b = b + a:
//     still reproducible) -- `partials` arrives zero-filled, as on the split-K path this kernel replaces.
#include "common.h"
#include "gemm_big.h"
#include <stdlib.h>
#include <stdio.h>
#include <atomic>
#include <type_traits>

namespace {

constexpr int ROWB = 64;                         // LDS row image of one 16-deep k-group: 16 x 16-bit hi | 16 x 16-bit lo
constexpr int TM = 64, NT = 512, KG = 2;
constexpr int OFF_BAD = 2 * TM * 4;              // crow[TM] | addrow[TM] in front, then the per-row zero-row flags
constexpr int OFF_A = OFF_BAD + TM * 4;
constexpr int A_BYTES = KG * TM * ROWB;          // 8 KB per stage
constexpr int OFF_W = OFF_A + 2 * A_BYTES;
constexpr int WS = 4;                            // weight stages
constexpr int tn_of(int nj) { return 64 * nj; }
constexpr int w_bytes(int nj) { return KG * tn_of(nj) * ROWB; }          // 8 KB per stage at the 64-column tile
constexpr int off_tab(int nj) { return OFF_W + WS * w_bytes(nj); }
constexpr int LDS_BLOCK = 64 * 1024;             // two blocks per CU with room to spare
constexpr int max_tab(int nj) { return (LDS_BLOCK - off_tab(nj)) / 8; }
// the lean loop's pipeline depth: weight stages = activation register sets = unroll period (a multiple of 4)
#ifndef GAST_BJ_DEPTH
#define GAST_BJ_DEPTH 4
#endif
constexpr int FD = GAST_BJ_DEPTH;
constexpr int off_tab_fast() { return OFF_W + FD * w_bytes(1); }
constexpr int LDS_BLOCK_FAST = FD == 4 ? LDS_BLOCK : 100 * 1024;
constexpr int max_tab_fast() { return (LDS_BLOCK_FAST - off_tab_fast()) / 8; }

// The scalar base of these loads may reach the statement through v_readfirstlane (uni() below: a pointer the compiler kept in vector
// registers).  gfx9 needs 5 wait states between a VALU write of an SGPR and a VMEM instruction that reads it; hipcc inserts them for the
// code it generates but cannot see inside an inline-asm statement -- without the s_nop the load uses the OLD register content, i.e. a wild
// address (the memory access faults that came and went with unrelated code changes in rounds 5 and 6: 204 such pairs in the build of
// this file that faulted, none in the next one -- scripts/asm_load_hazard.py --sgpr).  The wait states are part of
// the statement: the order no longer depends on what the scheduler puts in front.
template <int OFF>
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(OFF) : "memory", "m0");
}
__device__ __forceinline__ void gload16s(u32x4& dst, uint32_t voff, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// a wave-uniform pointer the compiler's divergence analysis may have lost track of (loop-carried through a lambda): scalar registers
__device__ __forceinline__ const char* uni(const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

struct Frag { uint4 u; };

// EPI: 0 PLAIN, 1 STATS, 2 BNRELU_BWD, 3 BNRELU_BWD with the dropout mask of the forward re-derived
// H16 (round 6, PAIR = 3): C / C2 / X / addend are tensors of the 16-bit storage type -- two-byte buffer accesses, values rounded to the
// storage type BEFORE they enter the column sums (the statistics are those of the tensor that is stored), as in gemm.hip / gemm_big.hip
template <int EPI, int NJ, bool H16 = false>
__device__ __forceinline__ void bj_epilogue(const gast_gemm_args& a, const BjPlan& pl, unsigned char* smem, const f32x16 (&acc)[NJ],
                                            int m0, int n0, int mt) {
    constexpr int TN = tn_of(NJ);
    constexpr bool bwd = EPI >= 2, xdrop = EPI == 3;
    constexpr uint32_t OOB = 0x80000000u, RSRC3 = 0x00020000u;
    constexpr uint32_t ESO = H16 ? 2u : 4u, QSTEP = 32u * ESO;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kg = w >> 2, wr = (w >> 1) & 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int M = pl.M, N = a.N;
    const int* const sCrow = (const int*)smem;
    const int* const sAdd = sCrow + TM;
    const uint32_t thresh = a.drop.thresh;
    const float inv_keep = a.drop.inv_keep;
    const uint32_t xkey = xdrop ? drop_key(a.drop, a.xsalt) : 0u;
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    const bool add = a.addend != nullptr, has2 = bwd && a.C2 != nullptr;
    auto ldv = [&](const __amdgpu_buffer_rsrc_t& r, uint32_t off) -> float {
        if constexpr (H16) return bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    };
    auto stv = [&](float v, const __amdgpu_buffer_rsrc_t& r, uint32_t off) {
        if constexpr (H16) __builtin_amdgcn_raw_buffer_store_b16(f2bf(v), r, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, 0);
    };
    // (a tensor that is absent gets a zero-sized descriptor: its loads return 0, its stores are dropped)
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((rowsC - 1) * a.ldc + N) * ESO), RSRC3);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd ? a.X : a.C), 0, bwd ? (int)(((rowsC - 1) * a.ldx + N) * ESO) : 0, RSRC3);
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(has2 ? a.C2 : a.C, 0, has2 ? (int)(((rowsC - 1) * a.ldc2 + N) * ESO) : 0, RSRC3);
    const long rowsAdd = add ? (long)a.B * a.addmap.T_total * a.J : 1;
    const __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc((void*)(add ? a.addend : a.C), 0, add ? (int)(((rowsAdd - 1) * a.ldadd + N) * ESO) : 0, RSRC3);
    const int col0 = n0 + wc * (TN / 2) + li;          // the lane's first column; the others are + 32 q
    bool nin[NJ];
    float bias[NJ], xs[NJ], xh[NJ], s1[NJ], s2[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int n = col0 + 32 * q;
        nin[q] = n < N;
        const int ncl = nin[q] ? n : N - 1;
        bias[q] = a.bias ? (a.bias_neg ? -a.bias[ncl] : a.bias[ncl]) : 0.f;
        xs[q] = bwd ? a.xscale[ncl] : 0.f;
        xh[q] = bwd ? a.xshift[ncl] : 0.f;
        s1[q] = 0.f; s2[q] = 0.f;
    }
    // this k-group finishes the accumulator registers 8 kg .. 8 kg + 7: units u = 0, 1 of 4 rows each,
    // rows wr*32 + 8 (2 kg + u) + 4 lh + {0..3}
    int crow[2][4], arow[2][4];
    float xv[2][NJ][4], av[2][NJ][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int base = wr * 32 + 8 * (2 * kg + u) + 4 * lh;
        const int4 c4 = *(const int4*)(sCrow + base);
        crow[u][0] = c4.x; crow[u][1] = c4.y; crow[u][2] = c4.z; crow[u][3] = c4.w;
        const int4 a4 = *(const int4*)(sAdd + base);
        arow[u][0] = a4.x; arow[u][1] = a4.y; arow[u][2] = a4.z; arow[u][3] = a4.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t offx = crow[u][r] >= 0 ? (uint32_t)(crow[u][r] * a.ldx + col0) * ESO : OOB;
            const uint32_t offa = arow[u][r] >= 0 ? (uint32_t)(arow[u][r] * a.ldadd + col0) * ESO : OOB;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                xv[u][q][r] = bwd ? ldv(rX, offx + QSTEP * q) : 0.f;
                av[u][q][r] = ldv(rAdd, offa + QSTEP * q);            // (no addend: zero-sized descriptor, reads 0)
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cr = crow[u][r];
            const uint32_t coff = (uint32_t)(cr * a.ldc + col0) * ESO;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const bool ok = cr >= 0 && nin[q];
                float v = acc[q][8 * kg + 4 * u + r] + bias[q] + av[u][q][r];
                if (bwd) {
                    stv(v, rC2, ok ? (uint32_t)(cr * a.ldc2 + col0) * ESO + QSTEP * q : OOB);
                    const float x = xv[u][q][r];
                    v = fmaf(x, xs[q], xh[q]) > 0.f ? v : 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)(cr * a.ldx + col0 + 32 * q));
                    if constexpr (H16) v = bf2f(f2bf(v));
                    s1[q] += ok ? v : 0.f;
                    s2[q] += ok ? v * x : 0.f;
                } else if (EPI == 1) {
                    if constexpr (H16) v = bf2f(f2bf(v));
                    s1[q] += ok ? v : 0.f;
                    s2[q] += ok ? v * v : 0.f;
                }
                stv(v, rC, ok ? coff + QSTEP * q : OOB);
            }
        }
    }
    if (EPI != 0) {
        // column sums of the block's 64 rows: the two lane halves by shuffle, the four waves (kg, wr) of a column half through LDS
        // (the K loop's stages are free by now), then ONE add per column into the 128-row statistics block
        float* const sRed = (float*)(smem + OFF_W + 2 * w_bytes(NJ));
        static_assert(OFF_W + 2 * w_bytes(NJ) >= OFF_A + 2 * NJ * 8 * 256 * 4, "sRed overlaps the exchange buffer");      // [4][TN][2]: the third weight stage (the accumulator exchange uses the space in front)
        (void)M;
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            s1[q] += __shfl_xor(s1[q], 32);
            s2[q] += __shfl_xor(s2[q], 32);
            if (lh == 0) {
                const int cl = wc * (TN / 2) + q * 32 + li;
                sRed[((kg * 2 + wr) * TN + cl) * 2] = s1[q];
                sRed[((kg * 2 + wr) * TN + cl) * 2 + 1] = s2[q];
            }
        }
        __syncthreads();
        const int n = n0 + tid;
        if (tid < TN && n < N) {
            float* pp = a.partials + ((long)(mt >> 1) * N + n) * 2;
            const float t1 = (sRed[tid * 2] + sRed[(TN + tid) * 2]) + (sRed[(2 * TN + tid) * 2] + sRed[(3 * TN + tid) * 2]);
            const float t2 = (sRed[tid * 2 + 1] + sRed[(TN + tid) * 2 + 1]) + (sRed[(2 * TN + tid) * 2 + 1] + sRed[(3 * TN + tid) * 2 + 1]);
            if (!(pl.ablate & 1)) {
                atomicAdd(pp, t1);
                atomicAdd(pp + 1, t2);
            }
        }
    }
}

// ---- the two k-groups' accumulators meet: group g keeps registers 8 g .. 8 g + 7 (rows 16 g .. 16 g + 15 of the wave tile) and hands the
// other half over through LDS ([8 NJ][256] floats per direction, conflict-free); then the epilogue.  Called behind a barrier that
// every wave reaches after its last fragment read.
template <int NJ, bool H16 = false>
__device__ __forceinline__ void bj_finish(const gast_gemm_args& a, const BjPlan& pl, unsigned char* smem, f32x16 (&acc)[NJ], int m0, int n0, int mt) {
    const int tid = threadIdx.x, kg = tid >> 8;
    {
        float* const xch = (float*)(smem + OFF_A);
        const int t256 = tid & 255;
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r) xch[((kg * NJ + q) * 8 + r) * 256 + t256] = acc[q][8 * (1 - kg) + r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float o = xch[(((1 - kg) * NJ + q) * 8 + r) * 256 + t256];
                // (always group 0's sum + group 1's sum, whichever group finishes the row)
                acc[q][8 * kg + r] = kg == 0 ? acc[q][8 * kg + r] + o : o + acc[q][8 * kg + r];
            }
    }
    if (pl.ablate & 2) { if (acc[0][0] == 12345.678f) ((float*)a.C)[0] = acc[0][5]; return; }
    const int v = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    if (v == 0) bj_epilogue<0, NJ, H16>(a, pl, smem, acc, m0, n0, mt);
    else if (v == 1) bj_epilogue<1, NJ, H16>(a, pl, smem, acc, m0, n0, mt);
    else if (v == 2) bj_epilogue<2, NJ, H16>(a, pl, smem, acc, m0, n0, mt);
    else bj_epilogue<3, NJ, H16>(a, pl, smem, acc, m0, n0, mt);
}

// PAIR: 1 = bf16 hi/lo pairs (GAST_F32X3), 2 = fp16 pairs (GAST_F32X3H: forward epilogues, images of the f16 kind)
//
// The K loop.  Three operand streams run ahead of the MFMAs, each with its own tile generator in scalar registers (a queue of full
// tile descriptors cost 140 spilled SGPRs):
//   * weights: DMA into a ring of WS = 4 stages, DW = 3 steps ahead;
//   * activations: NA = 4 register sets, loaded 5 steps ahead, converted (prologue + hi/lo split) one step ahead into 2 LDS stages;
//   * conversion facts (prologue? zero rows / K tail? scale / shift of the thread's 4 values): one step ahead of the conversion.
// The loop is unrolled by 4 so that every stage / register-set index is a compile-time constant (LDS offsets are immediates); zero
// rows / K tails and the prologue are block-uniform branches off the common path.
template <int NJ, int PAIR>
__device__ __forceinline__ void bj_body(const gast_gemm_args& a, const BjPlan& pl, int blk, unsigned char* smem) {
    static_assert(NJ == 1, "the stage budget below is the 64-column tile's");
    constexpr int TN = tn_of(NJ), W_BYTES = w_bytes(NJ), OFF_TAB = off_tab(NJ);
    constexpr int NW = NJ;                           // 1 KB DMA pieces per wave and K step
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kg = w >> 2, wr = (w >> 1) & 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M;
    // tile order: a contiguous chunk of the logical order runs on one XCD (xcd_remap), i.e. shares one L2.  Every activation row is
    // read by all N / 64 column tiles and every weight row by all M / 64 row tiles; the operand that is re-fetched per XCD should be the
    // SMALLER one: M >= N (most GEMMs of the stage: N = 512 / 1024 against 2 176 rows) -> the column tiles of one row tile are
    // consecutive (activations cross the fabric once, the weight panel once per XCD), else (G1: N = 5C + 8) the other way round.
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    int mt, nt;
    if (pl.M >= a.N) { mt = lb / pl.tilesN; nt = lb - mt * pl.tilesN; }
    else { nt = lb / pl.tilesM; mt = lb - nt * pl.tilesM; }
    const int m0 = mt * TM, n0 = nt * TN;

    int* const sCrow = (int*)smem;
    int* const sAdd = sCrow + TM;
    int* const sBad = (int*)(smem + OFF_BAD);        // [TM]: bit s = segment s maps this tile row to a row that must read as zero
    float* const sSc = (float*)(smem + OFF_TAB);
    float* const sSh = sSc + pl.ntab;
    const int nseg = a.nseg;

    // ---- staging duties.  Activations: thread = (row ra_row of the tile, 16-byte chunk c8 of the 32-value step: k-group c8 >> 2,
    // chunk c8 & 3 of its 16 values)
    const int ra_row = tid >> 3, c8 = tid & 7, kgA = c8 >> 2, cA = c8 & 3;
    const int TJ = a.Tn * a.J;
    int pb = -1, pt = 0, pj = 0;
    {
        const int m = m0 + ra_row;
        if (m < M) { pb = m / TJ; const int rem = m - pb * TJ; pt = rem / a.J; pj = rem - pt * a.J; }
    }
    if (c8 == 0) {          // one thread per tile row: output / addend rows, and which segments read this row as zero
        int crow = -1, arow = -1;
        if (pb >= 0) {
            crow = (int)map_row(a.cmap, pb, pt, pj, a.J);
            if (a.addend) arow = (int)map_row(a.addmap, pb, pt, pj, a.J);
        }
        sCrow[ra_row] = crow;
        sAdd[ra_row] = arow;
        int bad = 0;
        for (int s = 0; s < nseg; ++s) {
            const int ts = pt * a.seg[s].map.t_stride + a.seg[s].map.t_off;
            if (pb < 0 || ts < 0 || ts >= a.seg[s].map.T_total) bad |= 1 << s;
        }
        sBad[ra_row] = bad;
    }

    // ---- the activation-load stream (tile t + 5 at step t)
    struct Gen { int seg, k0, K; };
    auto advance = [&](Gen& g) -> bool {             // next tile; past the last tile: stay.  true: entered a new segment
        g.k0 += 32;
        if (g.k0 < g.K) return false;
        if (g.seg + 1 < nseg) { ++g.seg; g.k0 = 0; g.K = a.seg[g.seg].K; return true; }
        g.k0 -= 32;
        return false;
    };
    Gen ga = {0, 0, a.seg[0].K};
    const char* A_l = (const char*)a.seg[0].A;
    uint32_t offA = 0;
    bool zrow = true;
    auto enter_a = [&]() {
        const gast_gemm_seg& sg = a.seg[ga.seg];
        A_l = (const char*)sg.A;
        const int ts = pt * sg.map.t_stride + sg.map.t_off;
        const bool ok = pb >= 0 && ts >= 0 && ts < sg.map.T_total;
        const uint32_t srow = ok ? (uint32_t)((pb * sg.map.T_total + ts) * a.J + pj) : 0u;
        zrow = !ok;                                      // out-of-range tap (or a row past M): reads as zero
        offA = (srow * (uint32_t)sg.lda + c8 * 4) * 4u;
    };
    enter_a();
    u32x4 ra[4];
    bool rz[4];
    auto load_a = [&](u32x4& r, bool& z) {
        // (K tail: chunks past K re-read the step's first chunk and are zeroed by the conversion; branch-free -- three VALU
        //  instructions -- so that the load is ONE instruction behind the address arithmetic on every path)
        const bool kin = ga.k0 + c8 * 4 < ga.K;
        gload16s(r, offA - (kin ? 0u : (uint32_t)c8 * 16u), uni(A_l + ga.k0 * 4));
        z = zrow || !kin;
        if (advance(ga)) enter_a();
    };
    // ---- the weight stream (tile t + 3 at step t).  DMA: wave w fills the 1 KB pieces (w & 3) * NW + i of k-group w >> 2: 16 rows x
    // 64 B each, contiguous in the k-group-major image; lane = (row r16, slot s4), slot s4 receives source chunk s4 ^ key(row)
    const int r16 = lane >> 2, s4 = lane & 3;
    const int kgW = __builtin_amdgcn_readfirstlane(w >> 2), pieceW = __builtin_amdgcn_readfirstlane((w & 3) * NW);
    const uint32_t offW = (uint32_t)(n0 + pieceW * 16 + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);
    const uint32_t sW0 = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + kgW * (TN * ROWB) + pieceW * 1024);
    Gen gw = {0, 0, a.seg[0].K};
    const char* W_l = (const char*)a.seg[0].Wx;
    long ldg_l = (long)a.seg[0].ldwx * 2;                     // bytes per k-group of the weight image
    auto dma_w = [&](int stage) {
        // this wave's k-group of the step; a step whose second half lies past the segment's last 16-value group re-reads the first
        // one (the activations of that half are written as zeros)
        const int g = (gw.k0 >> 4) + kgW;
        const char* wbase = uni(W_l + (long)(g * 16 < gw.K ? g : (gw.k0 >> 4)) * ldg_l);
        const uint32_t sW = sW0 + stage * W_BYTES;
        glds16<0>(offW, wbase, sW);
        if (NW == 2) glds16<1024>(offW, wbase, sW);
        if (advance(gw)) { W_l = (const char*)a.seg[gw.seg].Wx; ldg_l = (long)a.seg[gw.seg].ldwx * 2; }
    };
    // ---- the conversion stream: facts of the tile that the NEXT step converts
    Gen gc = {0, 0, a.seg[0].K};
    int toff_l = pl.taboff[0];
    float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
    bool c_pro = false, c_fix = false, bad_l = false;
    int rowmask = 0;                                      // lane l: the zero-row bits of tile row l (loaded behind the first barrier)
    auto conv_facts = [&]() {                             // facts of the generator's tile, then advance
        c_pro = toff_l >= 0;
        c_fix = bad_l || gc.k0 + 32 > gc.K;
        if (c_pro) {
            const int k = toff_l + min(gc.k0 + c8 * 4, gc.K - 4);
            tsc = *(const float4*)(sSc + k);
            tsh = *(const float4*)(sSh + k);
        }
        if (advance(gc)) { toff_l = pl.taboff[gc.seg]; bad_l = __ballot((rowmask >> gc.seg) & 1) != 0; }
    };
    const int wa_key = (ra_row >> 2) & 3;
    const int wa_base = OFF_A + kgA * (TM * ROWB) + ra_row * ROWB + (cA & 1) * 8;
    const int wa_hi = wa_base + (((cA >> 1) ^ wa_key) << 4), wa_lo = wa_base + (((2 + (cA >> 1)) ^ wa_key) << 4);
    auto write_a = [&](int stage, const u32x4& r, bool z) {
        float x0 = __uint_as_float(r.x), x1 = __uint_as_float(r.y), x2 = __uint_as_float(r.z), x3 = __uint_as_float(r.w);
        if (c_pro) {         // BN + ReLU prologue
            x0 = fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f);
            x1 = fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f);
            x2 = fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f);
            x3 = fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f);
        }
        if (c_fix) {         // zero rows / the K tail must read as zero (relu(shift) must not leak in)
            x0 = z ? 0.f : x0; x1 = z ? 0.f : x1; x2 = z ? 0.f : x2; x3 = z ? 0.f : x3;
        }
        uint2 h, l;
        split_pair4<PAIR>(x0, x1, x2, x3, h, l);
        *(uint2*)(smem + stage * A_BYTES + wa_hi) = h;
        *(uint2*)(smem + stage * A_BYTES + wa_lo) = l;
    };

    int ntile = 0;
    for (int s = 0; s < nseg; ++s) ntile += (a.seg[s].K + 31) / 32;

    // two accumulators per output tile: the large products and the two correction products -- two independent MFMA chains per step
    f32x16 acc[NJ], acl[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[q][r] = 0.f; acl[q][r] = 0.f; }
    const int fkey = (li >> 2) & 3;
    const int ohi = li * ROWB + ((lh ^ fkey) << 4), olo = li * ROWB + (((2 + lh) ^ fkey) << 4);
    const unsigned char* const fA = smem + OFF_A + kg * (TM * ROWB) + wr * 32 * ROWB;
    const unsigned char* const fW = smem + OFF_W + kg * (TN * ROWB) + wc * (TN / 2) * ROWB;

    // ---- pipeline fill: W(0), A(0), A(1) first (the wait below needs exactly these), then W(1), A(2), W(2), A(3)
    dma_w(0);
    load_a(ra[0], rz[0]);
    load_a(ra[1], rz[1]);
    dma_w(1);
    load_a(ra[2], rz[2]);
    dma_w(2);
    load_a(ra[3], rz[3]);
    for (int s = 0; s < nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
        if (pl.taboff[s] >= 0) {
            const float* sc = a.seg[s].scale;
            const float* sh = a.seg[s].shift;
            for (int k = tid; k < a.seg[s].K; k += NT) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
        }
    }
    gload_wait_n<2 + 2 * NW>();
    __syncthreads();                                   // tables, row maps and zero-row flags complete
    gload_pin(ra[0]);
    gload_pin(ra[1]);
    rowmask = sBad[lane];
    bad_l = __ballot(rowmask & 1) != 0;
    conv_facts();                                      // tile 0
    write_a(0, ra[0], rz[0]);
    load_a(ra[0], rz[0]);                              // tile 4
    conv_facts();                                      // tile 1 (converted by step 0)

    // One K step (U = t % 4).  At the top, after the counted wait + barrier: LDS holds tile t (activations in stage t & 1, weights in
    // stage t % 4); register set (t + 1) % 4 holds tile t + 1's activations; in flight: the weights of tiles t+1, t+2, the activations
    // of tiles t+2 .. t+4 -- 3 + 2 NW transfers, everything older is complete.
    auto step = [&](auto Uc, int t) {
        constexpr int U = decltype(Uc)::value;
        constexpr int SA = U & 1, SW = U, SET = (U + 1) & 3;
        gload_wait_n<3 + 2 * NW>();
        __syncthreads();
        gload_pin(ra[SET]);
        Frag ah, al, bh[NJ], bl[NJ];
        ah.u = *(const uint4*)(fA + SA * A_BYTES + ohi);
        al.u = *(const uint4*)(fA + SA * A_BYTES + olo);
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            bh[q].u = *(const uint4*)(fW + SW * W_BYTES + q * 32 * ROWB + ohi);
            bl[q].u = *(const uint4*)(fW + SW * W_BYTES + q * 32 * ROWB + olo);
        }
        if (t + 1 < ntile) write_a(SA ^ 1, ra[SET], rz[SET]);    // tile t+1: registers -> LDS (the set is then free for tile t+5)
        if (t < ntile) {
#pragma unroll
            for (int q = 0; q < NJ; ++q) acl[q] = mfma_pair<PAIR>(al.u, bh[q].u, acl[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acc[q] = mfma_pair<PAIR>(ah.u, bh[q].u, acc[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acl[q] = mfma_pair<PAIR>(ah.u, bl[q].u, acl[q]);
        }
        dma_w((U + 3) & 3);                                      // tile t + 3 -> stage (t + 3) % 4 (tile t - 1's: every wave is past its reads)
        load_a(ra[SET], rz[SET]);                                // tile t + 5
        conv_facts();                                            // tile t + 2
    };
    for (int t = 0; t < ((pl.ablate & 4) ? 0 : ntile); t += 4) {
        step(std::integral_constant<int, 0>{}, t);
        step(std::integral_constant<int, 1>{}, t + 1);
        step(std::integral_constant<int, 2>{}, t + 2);
        step(std::integral_constant<int, 3>{}, t + 3);
    }
    gload_wait_n<0>();                 // (the re-requested tiles past the end: nothing may land in LDS or in registers after this point)
    asm volatile("" ::"v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]));
#pragma unroll
    for (int q = 0; q < NJ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] += acl[q][r];
    __syncthreads();

    bj_finish<NJ>(a, pl, smem, acc, m0, n0, mt);
}

// The same kernel for the shapes the stage is made of -- every segment's K a multiple of 128, full row maps (pl.fast) -- with a K
// loop of a THIRD of the instructions.  SQ counters of the general loop above (profiles/r06_pmc_gemm_bj.txt): ~120 instructions per
// wave and 32-value step for 3 MFMAs, a wave active (issuing) a third of its cycles, L2 read latency 340 cycles at a 74 % hit rate --
// the loop is bound by its own instruction stream (a wave issues in order, ~4.5 cycles per instruction), not by memory.  Here the
// three operand streams advance in GROUPS of four steps: inside a group every address is (group base + compile-time offset), a
// stream looks at its segment once per four steps (in the one sub-step where its look-ahead crosses a group boundary), and there is
// no tail / zero-row / past-the-end handling at all (K steps come in fours).
template <int OFF>
__device__ __forceinline__ void gload16s_o(u32x4& dst, uint32_t voff, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// CHAINS = 2: the correction products accumulate in a register set of their own (two independent MFMA chains per step: what a lone
// block per CU wants); CHAINS = 1: one accumulator, 80 registers -- three blocks per CU, for the grids that do not fit two per CU
// (measured: N = 1024 launches 55 -> 50 and 47 -> 43 us; the 34-block output layer 22 -> 27 us with it, so it keeps two chains)
// PAIR = 3 (round 6): 16-BIT STORAGE (GAST_BF16 tensors: bfloat16, or binary16 in the -DGAST_H16_F16 build), ONE product.  A step covers
// 64 values: k-group g multiplies the 32-value half g, whose row image is [32 x 16 bit] = the same 64 bytes -- the byte geometry of the
// activation loads (16 bytes per thread, 128 bytes per row and step, 512 per group of four), of the weight DMA (layout image of
// gast_x3_image_multi kind 2) and of the fragment reads is the split kernel's; the chunk a lane reads for the first 16-deep MFMA is the
// one that holds the hi halves there (chunk lh), for the second the lo one (chunk 2 + lh).  The conversion is a pass-through, or BN +
// ReLU on the 8 unpacked values.  A group of four steps is 256 values: every K of the launch must be a multiple of 256.
template <int PAIR, int CHAINS>
__device__ __forceinline__ void bj_body_fast(const gast_gemm_args& a, const BjPlan& pl, int blk, unsigned char* smem) {
    constexpr int NJ = 1, TN = tn_of(NJ), W_BYTES = w_bytes(NJ), OFF_TAB = off_tab_fast(), NW = 1;
    constexpr bool H16 = PAIR == 3;
    constexpr int ESZ = H16 ? 2 : 4, CV = 16 / ESZ;       // bytes per stored element, values per 16-byte chunk
    constexpr int GSH = H16 ? 8 : 7;                      // log2(values per group of four steps)
    constexpr int TSTEP = 8 * CV * 4;                     // bytes of the scale / shift tables per step: 128 / 256
    static_assert(FD % 4 == 0 && FD >= 4, "pipeline depth");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kg = w >> 2, wr = (w >> 1) & 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    int mt, nt;
    if (pl.M >= a.N) { mt = lb / pl.tilesN; nt = lb - mt * pl.tilesN; }
    else { nt = lb / pl.tilesM; mt = lb - nt * pl.tilesM; }
    const int m0 = mt * TM, n0 = nt * TN;
    int* const sCrow = (int*)smem;
    int* const sAdd = sCrow + TM;
    const int nseg = a.nseg;
    const int ra_row = tid >> 3, c8 = tid & 7, kgA = c8 >> 2, cA = c8 & 3;
    const int TJ = a.Tn * a.J;
    int pb, pt, pj;
    const bool rin = m0 + ra_row < pl.M;             // (the last row tile may be ragged: its rows past M load row M - 1 and store nothing)
    {
        const int m = rin ? m0 + ra_row : pl.M - 1;
        pb = m / TJ; const int rem = m - pb * TJ; pt = rem / a.J; pj = rem - pt * a.J;
    }
    if (c8 == 0) {
        sCrow[ra_row] = rin ? (int)map_row(a.cmap, pb, pt, pj, a.J) : -1;
        sAdd[ra_row] = (rin && a.addend) ? (int)map_row(a.addmap, pb, pt, pj, a.J) : -1;
    }
    // ---- activation loads: group base (scalar) + the thread's byte offset; groups of 4 steps = 128 values = 512 bytes
    int a_seg = 0, a_grp = a.seg[0].K >> GSH;
    const char* a_ptr = (const char*)a.seg[0].A;
    uint32_t offA;
    auto a_row = [&](const gast_gemm_seg& sg) {      // (pl.fast: every frame of the domain maps into the segment's tensor)
        const uint32_t srow = (uint32_t)((pb * sg.map.T_total + pt * sg.map.t_stride + sg.map.t_off) * a.J + pj);
        offA = (srow * (uint32_t)sg.lda + c8 * CV) * (uint32_t)ESZ;
    };
    a_row(a.seg[0]);
    auto a_next_group = [&]() {
        a_ptr += 512;
        if (--a_grp == 0) {
            if (a_seg + 1 < nseg) {
                ++a_seg;
                const gast_gemm_seg& sg = a.seg[a_seg];
                a_ptr = (const char*)sg.A;
                a_grp = sg.K >> GSH;
                a_row(sg);
            } else { a_ptr -= 512; a_grp = 1; }      // past the last group: the last group again
        }
    };
    u32x4 ra[FD];
    // ---- weights: running pointer of the next step's 16-value group of this wave's k-group; a step is 2 groups further
    const int r16 = lane >> 2, s4 = lane & 3;
    const int kgW = __builtin_amdgcn_readfirstlane(w >> 2), pieceW = __builtin_amdgcn_readfirstlane((w & 3) * NW);
    const uint32_t offW = (uint32_t)(n0 + pieceW * 16 + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);
    const uint32_t sW0 = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + kgW * (TN * ROWB) + pieceW * 1024);
    int w_seg = 0, w_grp = a.seg[0].K >> GSH;
    long w_step = (long)a.seg[0].ldwx * 4;           // bytes per step: two k-groups (16 values each; PAIR = 3: 32) of ldwx 16-bit elements
    const char* w_cur = (const char*)a.seg[0].Wx + (w_step >> 1) * kgW;
    auto dma_w = [&](int stage) {
        glds16<0>(offW, uni(w_cur), sW0 + stage * W_BYTES);
        w_cur += w_step;
    };
    auto w_next_group = [&]() {                      // (called when the next request is the first step of a new group)
        if (--w_grp == 0) {
            if (w_seg + 1 < nseg) {
                ++w_seg;
                w_grp = a.seg[w_seg].K >> GSH;
                w_step = (long)a.seg[w_seg].ldwx * 4;
                w_cur = (const char*)a.seg[w_seg].Wx + (w_step >> 1) * kgW;
            } else { w_cur -= 4 * w_step; w_grp = 1; }
        }
    };
    // ---- conversion facts: prologue of the segment? + the LDS address of the thread's scale / shift values of the group
    int c_seg = 0, c_grp = a.seg[0].K >> GSH;
    bool c_pro = pl.taboff[0] >= 0;
    const int tab_sh = pl.ntab * 4;
    int tabv = OFF_TAB + (c_pro ? pl.taboff[0] : 0) * 4 + c8 * (CV * 4);
    auto c_next_group = [&]() {
        tabv += 4 * TSTEP;
        if (--c_grp == 0) {
            if (c_seg + 1 < nseg) {
                ++c_seg;
                c_grp = a.seg[c_seg].K >> GSH;
                c_pro = pl.taboff[c_seg] >= 0;
                tabv = OFF_TAB + (c_pro ? pl.taboff[c_seg] : 0) * 4 + c8 * (CV * 4);
            } else { tabv -= 4 * TSTEP; c_grp = 1; }
        }
    };
    float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tsc2 = tsc, tsh2 = tsh;                   // (PAIR = 3: values 4 .. 7 of the thread's chunk)
    bool t_pro = false;                              // prologue flag of the tile whose scale / shift are held
    auto fetch_tab = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        t_pro = c_pro;
        if (c_pro) {
            tsc = *(const float4*)(smem + tabv + J * TSTEP);
            tsh = *(const float4*)(smem + tabv + tab_sh + J * TSTEP);
            if constexpr (H16) {
                tsc2 = *(const float4*)(smem + tabv + J * TSTEP + 16);
                tsh2 = *(const float4*)(smem + tabv + tab_sh + J * TSTEP + 16);
            }
        }
    };
    const int wa_key = (ra_row >> 2) & 3;
    const int wa_base = OFF_A + kgA * (TM * ROWB) + ra_row * ROWB + (cA & 1) * 8;
    const int wa_hi = wa_base + (((cA >> 1) ^ wa_key) << 4), wa_lo = wa_base + (((2 + (cA >> 1)) ^ wa_key) << 4);
    const int wa_h16 = OFF_A + kgA * (TM * ROWB) + ra_row * ROWB + ((cA ^ wa_key) << 4);      // PAIR = 3: chunk cA of the row image, whole
    auto write_a = [&](int stage, const u32x4& r) {
        if constexpr (H16) {
            uint32_t w4[4] = {r.x, r.y, r.z, r.w};
            if (t_pro) {     // BN + ReLU on the 8 unpacked values, rounded back to the storage type
                const float sc8[8] = {tsc.x, tsc.y, tsc.z, tsc.w, tsc2.x, tsc2.y, tsc2.z, tsc2.w};
                const float sh8[8] = {tsh.x, tsh.y, tsh.z, tsh.w, tsh2.x, tsh2.y, tsh2.z, tsh2.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float lo, hi;
                    h16x2_unpack(w4[p], lo, hi);
                    w4[p] = pack_h16x2(fmaxf(fmaf(lo, sc8[2 * p], sh8[2 * p]), 0.f), fmaxf(fmaf(hi, sc8[2 * p + 1], sh8[2 * p + 1]), 0.f));
                }
            }
            *(uint4*)(smem + stage * A_BYTES + wa_h16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            return;
        }
        float x0 = __uint_as_float(r.x), x1 = __uint_as_float(r.y), x2 = __uint_as_float(r.z), x3 = __uint_as_float(r.w);
        if (t_pro) {         // BN + ReLU prologue
            x0 = fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f);
            x1 = fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f);
            x2 = fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f);
            x3 = fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f);
        }
        uint2 h, l;
        split_pair4<H16 ? 1 : PAIR>(x0, x1, x2, x3, h, l);
        *(uint2*)(smem + stage * A_BYTES + wa_hi) = h;
        *(uint2*)(smem + stage * A_BYTES + wa_lo) = l;
    };
    const int ntile = pl.ntile32;                    // K steps of the launch (a multiple of 4)
    f32x16 acc[NJ], acl[CHAINS == 2 ? NJ : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; if (CHAINS == 2) acl[0][r] = 0.f; }
    const int fkey = (li >> 2) & 3;
    const int ohi = li * ROWB + ((lh ^ fkey) << 4), olo = li * ROWB + (((2 + lh) ^ fkey) << 4);
    const unsigned char* const fA = smem + OFF_A + kg * (TM * ROWB) + wr * 32 * ROWB;
    const unsigned char* const fW = smem + OFF_W + kg * (TN * ROWB) + wc * (TN / 2) * ROWB;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---- pipeline fill: W(0), A(0), A(1) first (the first wait needs exactly these), then W(j), A(j + 1) for j = 1 .. FD - 2;
    // tile j is step j & 3 of group j >> 2
    auto load_tile = [&](auto Jc, u32x4& r) {
        constexpr int J = decltype(Jc)::value;
        if (J != 0 && (J & 3) == 0) a_next_group();
        gload16s_o<(J & 3) * 128>(r, offA, uni(a_ptr));
    };
    auto dma_tile = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        if (J != 0 && (J & 3) == 0) w_next_group();
        dma_w(J % FD);
    };
    dma_tile(I0{});
    load_tile(I0{}, ra[0]);
    load_tile(I1{}, ra[1]);
    dma_tile(I1{}); load_tile(I2{}, ra[2]);
    dma_tile(I2{}); load_tile(I3{}, ra[3]);
    if constexpr (FD == 8) {
        dma_tile(std::integral_constant<int, 3>{}); load_tile(std::integral_constant<int, 4>{}, ra[4]);
        dma_tile(std::integral_constant<int, 4>{}); load_tile(std::integral_constant<int, 5>{}, ra[5]);
        dma_tile(std::integral_constant<int, 5>{}); load_tile(std::integral_constant<int, 6>{}, ra[6]);
        dma_tile(std::integral_constant<int, 6>{}); load_tile(std::integral_constant<int, 7>{}, ra[7]);
    }
    {
        float* const sSc = (float*)(smem + OFF_TAB);
        float* const sSh = sSc + pl.ntab;
        for (int s = 0; s < nseg; ++s) {
            if (pl.taboff[s] >= 0) {
                const float* sc = a.seg[s].scale;
                const float* sh = a.seg[s].shift;
                for (int k = tid; k < a.seg[s].K; k += NT) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
            }
        }
    }
    gload_wait_n<(FD - 2) * (NW + 1)>();               // everything but W(0), A(0), A(1)
    __syncthreads();
    gload_pin(ra[0]);
    gload_pin(ra[1]);
    fetch_tab(I0{});                                   // tile 0
    write_a(0, ra[0]);
    load_tile(std::integral_constant<int, FD>{}, ra[0]);      // tile FD = the first step of a new group
    fetch_tab(I1{});                                   // tile 1 (converted by step 0)

    // step t = FD g + U.  Stage / register-set indices are compile-time; so are the streams' positions inside their groups of four:
    // activations tile t + 1 + FD = step (U + 1) % 4 of its group, weights tile t + FD - 1 = step (U + 3) % 4, facts tile t + 2.
    // In flight at the top: the weights of tiles t+1 .. t+FD-2 and the activations of tiles t+2 .. t+FD.
    auto step = [&](auto Uc, bool wr_next, bool do_mma) {
        constexpr int U = decltype(Uc)::value;
        constexpr int SA = U & 1, SW = U % FD, SET = (U + 1) % FD;
        gload_wait_n<(FD - 2) * NW + (FD - 1)>();
        __syncthreads();
        gload_pin(ra[SET]);
        Frag ah, al, bh, bl;
        ah.u = *(const uint4*)(fA + SA * A_BYTES + ohi);
        al.u = *(const uint4*)(fA + SA * A_BYTES + olo);
        bh.u = *(const uint4*)(fW + SW * W_BYTES + ohi);
        bl.u = *(const uint4*)(fW + SW * W_BYTES + olo);
        if (wr_next) write_a(SA ^ 1, ra[SET]);
        if (do_mma) {
            if constexpr (H16) {      // k 0..15 of the k-group's 32 values (chunks lh), then k 16..31 (chunks 2 + lh)
                if constexpr (CHAINS == 2) {
                    acc[0] = mfma_h16(ah.u, bh.u, acc[0]);
                    acl[0] = mfma_h16(al.u, bl.u, acl[0]);
                } else {
                    acc[0] = mfma_h16(ah.u, bh.u, acc[0]);
                    acc[0] = mfma_h16(al.u, bl.u, acc[0]);
                }
            } else if constexpr (CHAINS == 2) {
                acl[0] = mfma_pair<H16 ? 1 : PAIR>(al.u, bh.u, acl[0]);
                acc[0] = mfma_pair<H16 ? 1 : PAIR>(ah.u, bh.u, acc[0]);
                acl[0] = mfma_pair<H16 ? 1 : PAIR>(ah.u, bl.u, acl[0]);
            } else {       // small terms first
                acc[0] = mfma_pair<H16 ? 1 : PAIR>(al.u, bh.u, acc[0]);
                acc[0] = mfma_pair<H16 ? 1 : PAIR>(ah.u, bl.u, acc[0]);
                acc[0] = mfma_pair<H16 ? 1 : PAIR>(ah.u, bh.u, acc[0]);
            }
        }
        if ((U & 3) == 1) w_next_group();
        dma_w((U + FD - 1) % FD);
        if ((U & 3) == 3) a_next_group();
        gload16s_o<((U + 1) & 3) * 128>(ra[SET], offA, uni(a_ptr));
        if ((U & 3) == 2) c_next_group();
        fetch_tab(std::integral_constant<int, (U + 2) & 3>{});
    };
    for (int t = 0; t < ((pl.ablate & 4) ? 0 : ntile); t += FD) {
        if constexpr (FD == 4) {
            step(I0{}, true, true);
            step(I1{}, true, true);
            step(I2{}, true, true);
            step(I3{}, t + 4 < ntile, true);
        } else {
            const bool more = t + 4 < ntile;           // (K steps come in fours: the second half of the trip may lie past the end)
            step(I0{}, true, true);
            step(I1{}, true, true);
            step(I2{}, true, true);
            step(I3{}, more, true);
            step(std::integral_constant<int, 4>{}, more, more);
            step(std::integral_constant<int, 5>{}, more, more);
            step(std::integral_constant<int, 6>{}, more, more);
            step(std::integral_constant<int, 7>{}, more && t + 8 < ntile, more);
        }
    }
    gload_wait_n<0>();
#pragma unroll
    for (int i = 0; i < FD; ++i) asm volatile("" ::"v"(ra[i]));
    if constexpr (CHAINS == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acl[0][r];
    }
    __syncthreads();
    bj_finish<NJ, H16>(a, pl, smem, acc, m0, n0, mt);
}

struct BjBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    BjPlan pl[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int n;
};
static_assert(sizeof(BjBatch) <= 3840, "BjBatch travels as a kernel argument (4 KB limit)");

template <int NJ, int PAIR, int FAST>      // FAST: 0 general loop, 1 lean loop (two chains, two blocks per CU), 2 lean loop (one chain, three blocks per CU)
__global__ void __launch_bounds__(NT, (FAST && FD > 4) ? 2 : (FAST == 2 ? 6 : 4)) gemm_bj_kernel(const gast_gemm_args a, const BjPlan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (FAST != 0) bj_body_fast<PAIR, FAST == 2 ? 1 : 2>(a, pl, blockIdx.x, smem);
    else bj_body<NJ, PAIR>(a, pl, blockIdx.x, smem);
}
template <int NJ, int PAIR, int FAST>
__global__ void __launch_bounds__(NT, (FAST && FD > 4) ? 2 : (FAST == 2 ? 6 : 4)) gemm_bj_multi_kernel(const BjBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    if constexpr (FAST != 0) bj_body_fast<PAIR, FAST == 2 ? 1 : 2>(b.a[d], b.pl[d], blockIdx.x - b.first[d], smem);
    else bj_body<NJ, PAIR>(b.a[d], b.pl[d], blockIdx.x - b.first[d], smem);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
std::atomic<bool> bj_setup_done[64];

typedef void (*bj_kernel_t)(const gast_gemm_args, const BjPlan);
typedef void (*bj_multi_kernel_t)(const BjBatch);
// (pair 3 -- 16-bit storage -- exists for the two-chain lean loop only: the plan refuses what it cannot run, and the 80-register variant
//  would spill three registers inside the K loop -- compiler-issued scratch accesses break the counted waits)
bj_kernel_t bj_kernel(int fast, int pair) {
    if (fast == 2 && pair != 3) return pair == 2 ? gemm_bj_kernel<1, 2, 2> : gemm_bj_kernel<1, 1, 2>;
    if (fast) return pair == 3 ? gemm_bj_kernel<1, 3, 1> : pair == 2 ? gemm_bj_kernel<1, 2, 1> : gemm_bj_kernel<1, 1, 1>;
    return pair == 3 ? nullptr : pair == 2 ? gemm_bj_kernel<1, 2, 0> : gemm_bj_kernel<1, 1, 0>;
}
bj_multi_kernel_t bj_multi_kernel(int fast, int pair) {
    if (fast == 2 && pair != 3) return pair == 2 ? gemm_bj_multi_kernel<1, 2, 2> : gemm_bj_multi_kernel<1, 1, 2>;
    if (fast) return pair == 3 ? gemm_bj_multi_kernel<1, 3, 1> : pair == 2 ? gemm_bj_multi_kernel<1, 2, 1> : gemm_bj_multi_kernel<1, 1, 1>;
    return pair == 3 ? nullptr : pair == 2 ? gemm_bj_multi_kernel<1, 2, 0> : gemm_bj_multi_kernel<1, 1, 0>;
}
int bj_lds_bytes(int ntab, int fast) {
    static const int pad = getenv("GAST_GEMM_BJ_LDS_PAD") ? atoi(getenv("GAST_GEMM_BJ_LDS_PAD")) : 0;      // (occupancy experiments: fewer blocks per CU)
    const int n = (fast ? off_tab_fast() : off_tab(1)) + 2 * ntab * 4 + pad;
    return n > LDS_BLOCK ? LDS_BLOCK : n;
}

void bj_setup() {
    int dev = 0;
    hipGetDevice(&dev);
    dev &= 63;
    if (bj_setup_done[dev].load(std::memory_order_acquire)) return;
    for (int fast = 0; fast <= 2; ++fast)
        for (int pair = 1; pair <= 3; ++pair) {
            if (!bj_kernel(fast, pair)) continue;
            const hipError_t e1 = hipFuncSetAttribute((const void*)bj_kernel(fast, pair), hipFuncAttributeMaxDynamicSharedMemorySize, fast ? LDS_BLOCK_FAST : LDS_BLOCK);
            const hipError_t e2 = hipFuncSetAttribute((const void*)bj_multi_kernel(fast, pair), hipFuncAttributeMaxDynamicSharedMemorySize, fast ? LDS_BLOCK_FAST : LDS_BLOCK);
            if (e1 != hipSuccess || e2 != hipSuccess) {
                fprintf(stderr, "gast_hip: gemm_bj set-up failed for fast %d pair %d: %s\n", fast, pair, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                (void)hipGetLastError();
            }
        }
    bj_setup_done[dev].store(true, std::memory_order_release);
    if (getenv("GAST_GEMM_BJ_DEBUG")) {
        for (int fast = 0; fast <= 2; ++fast) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)bj_kernel(fast, 1), NT, bj_lds_bytes(0, fast));
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void*)bj_kernel(fast, 1));
            fprintf(stderr, "gemm_bj fast %d: %d blocks/CU at %d B LDS, %d regs, %zu B scratch\n", fast, nb, bj_lds_bytes(0, fast), fa.numRegs, (size_t)fa.localSizeBytes);
        }
    }
}

}  // namespace

// Can this GEMM run on the M = B*J kernel?  Fills the plan when it can (nj = 0: the caller picks the tile width for the launch).
int gast_gemm_bj_plan(const gast_gemm_args& a, BjPlan& pl) {
    static const int enabled = getenv("GAST_GEMM_BJ") ? atoi(getenv("GAST_GEMM_BJ")) : 1;
    static const int max_rows = getenv("GAST_GEMM_BJ_MAX_M") ? atoi(getenv("GAST_GEMM_BJ_MAX_M")) : 8191;
    // Round 6: 16-bit storage (GAST_BF16 tensors: bfloat16, or binary16 in the f16 build; layout images of the weights) on the lean loop,
    // every epilogue, 16-bit output only.  GAST_GEMM_BJ_H16=0: those GEMMs stay on gemm.hip's split-K path.
    static const int h16_enabled = getenv("GAST_GEMM_BJ_H16") ? atoi(getenv("GAST_GEMM_BJ_H16")) : 1;
    const bool h16 = a.dtype == GAST_BF16;
    if (!enabled || (a.dtype != GAST_F32X3 && a.dtype != GAST_F32X3H && !(h16 && h16_enabled))) return 0;
    if (a.dtype == GAST_F32X3H && a.epi == GAST_EPI_BNRELU_BWD) return 0;     // (a gradient operand does not fit fp16's range)
    if (a.out_f32 || a.f8_scale) return 0;
    pl.pair = h16 ? 3 : a.dtype == GAST_F32X3H ? 2 : 1;
    const int esz = h16 ? 2 : 4, cv = 16 / esz;
    const long Ml = (long)a.B * a.Tn * a.J;
    if (Ml < 1 || Ml > max_rows || a.N < 1) return 0;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG || !a.C) return 0;
    int ntab = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.Wx || !aligned16(g.Wx) || g.ldwx % 8 || !g.A || !aligned16(g.A) || g.lda % cv || g.K % cv || g.K < cv) return 0;
        if ((long)a.B * g.map.T_total * a.J * g.lda * esz >= 0xffffffffL) return 0;      // 32-bit byte offsets into the activation tensor
        if (g.pro == GAST_PRO_BNRELU_DROP) return 0;
        pl.taboff[s] = -1;
        if (g.pro == GAST_PRO_BNRELU) {
            if (!g.scale || !g.shift) return 0;
            for (int q = 0; q < s; ++q)
                if (pl.taboff[q] >= 0 && a.seg[q].scale == g.scale && a.seg[q].shift == g.shift && a.seg[q].K == g.K) pl.taboff[s] = pl.taboff[q];
            if (pl.taboff[s] < 0) { pl.taboff[s] = ntab; ntab += (g.K + 3) / 4 * 4; }
        }
    }
    if (ntab > max_tab(1) || ntab > max_tab_fast()) return 0;
    if (a.epi < 0 || a.epi > GAST_EPI_BNRELU_BWD) return 0;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return 0;
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    if (rowsC * a.ldc * esz >= 0x7fffffffL || (a.epi == GAST_EPI_BNRELU_BWD && rowsC * a.ldx * esz >= 0x7fffffffL)) return 0;
    if (a.C2 && (a.epi != GAST_EPI_BNRELU_BWD || rowsC * a.ldc2 * esz >= 0x7fffffffL)) return 0;
    if (a.addend && (long)a.B * a.addmap.T_total * a.J * a.ldadd * esz >= 0x7fffffffL) return 0;
    pl.M = (int)Ml;
    // the lean K loop (bj_body_fast): K steps in groups of four for every segment, no zero rows anywhere
    static const int fast_ok = getenv("GAST_GEMM_BJ_FAST") ? atoi(getenv("GAST_GEMM_BJ_FAST")) : 1;
    pl.fast = fast_ok != 0;                   // (any row count: the rows of a ragged last tile past M are clamped on load and never stored)
    pl.ntile32 = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_rowmap& mp = a.seg[s].map;
        const long lo = mp.t_stride >= 0 ? mp.t_off : (long)(a.Tn - 1) * mp.t_stride + mp.t_off;
        const long hi = mp.t_stride >= 0 ? (long)(a.Tn - 1) * mp.t_stride + mp.t_off : mp.t_off;
        if (a.seg[s].K % (h16 ? 256 : 128) || lo < 0 || hi >= mp.T_total) pl.fast = 0;
        pl.ntile32 += h16 ? (a.seg[s].K + 63) / 64 : (a.seg[s].K + 31) / 32;
    }
    if (h16 && !pl.fast) return 0;          // (16-bit storage: the lean loop is the only one)
    // Where this kernel is used (measured on MI355X at M = 2 176, scripts/gemm_table.py bf16x3: one launch per graph replay, this kernel
    // against gemm.hip's split-K pair): with the lean loop it wins or ties on every shape of the stage whose K steps come in fours and
    // whose grid is at most ~2 blocks per CU -- 24 / 39 / 47 / 41 / 25 / 42 us against 34 / 52 / 54 / 52 / 34 / 61 (K <= 1024), 40 / 54 /
    // 21 against 40 / 53 / 22 (K = 1536, the N = 3 output layer) -- and saves the finish launch and the workspace round trip either way.
    // It loses on N = 5C + 8 (1 394 tiles: 56 vs 44 without any split) and, through the general loop, on K = 8 (27 vs 18) and K = 3 592
    // (82 vs 62): those stay on gemm.hip.  GAST_GEMM_BJ_ALL=1: every eligible shape (kernel tests).
    static const int all_shapes = getenv("GAST_GEMM_BJ_ALL") ? atoi(getenv("GAST_GEMM_BJ_ALL")) : 0;
    static const int max_tiles = getenv("GAST_GEMM_BJ_MAX_TILES") ? atoi(getenv("GAST_GEMM_BJ_MAX_TILES")) : 600;
    if (!all_shapes && (!pl.fast || ((Ml + TM - 1) / TM) * ((a.N + 63) / 64) > max_tiles)) return 0;
    pl.tilesM = (pl.M + TM - 1) / TM;
    pl.ntab = ntab;
    pl.nj = 0;
    pl.tilesN = 0;
    static const int ablate = getenv("GAST_GEMM_BJ_ABLATE") ? atoi(getenv("GAST_GEMM_BJ_ABLATE")) : 0;
    pl.ablate = ablate;
    return 1;
}

static int bj_pick_nj(const gast_gemm_args*, const BjPlan*, int) { return 1; }      // (one tile width: 64 columns)

int gast_gemm_bj_launch_multi(const gast_gemm_args* args, BjPlan* pls, int n, hipStream_t st) {
    bj_setup();
    const int nj = bj_pick_nj(args, pls, n), pair = pls[0].pair;
    BjBatch b;
    b.n = n;
    b.first[0] = 0;
    int ntab = 0, fast = 1;
    long blocks = 0;
    for (int d = 0; d < n; ++d) { fast &= pls[d].fast; blocks += (long)pls[d].tilesM * ((args[d].N + 63) / 64); }      // (one kernel per launch: the lean loop when every job qualifies)
    static const int occ3_blocks = getenv("GAST_GEMM_BJ_OCC3_BLOCKS") ? atoi(getenv("GAST_GEMM_BJ_OCC3_BLOCKS")) : 512;
    if (fast && FD == 4 && blocks > occ3_blocks && pair != 3) fast = 2;          // more blocks than two per CU: the 80-register variant, three per CU
    for (int d = 0; d < n; ++d) {
        if (pls[d].pair != pair) return GAST_EINVAL;
        pls[d].nj = nj;
        pls[d].fast = fast != 0;
        pls[d].tilesN = (args[d].N + tn_of(nj) - 1) / tn_of(nj);
        b.a[d] = args[d];
        b.pl[d] = pls[d];
        b.first[d + 1] = b.first[d] + pls[d].tilesM * pls[d].tilesN;
        if (pls[d].ntab > ntab) ntab = pls[d].ntab;
    }
    if (n == 1) hipLaunchKernelGGL(bj_kernel(fast, pair), dim3(b.first[1]), dim3(NT), bj_lds_bytes(ntab, fast), st, b.a[0], b.pl[0]);
    else hipLaunchKernelGGL(bj_multi_kernel(fast, pair), dim3(b.first[n]), dim3(NT), bj_lds_bytes(ntab, fast), st, b);
    GAST_CHECK_LAUNCH();
    return 0;
}
