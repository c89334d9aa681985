// gast_gemm, the M = B*J stage (few rows, long K, fat N) for GAST_F32X3 / GAST_F32X3H (fp32 storage, split 16-bit products), gfx950.
//
// Same contract as gemm.hip / gemm_big.hip (K segments with row maps = channel concat / temporal taps of reference gast_net.py:28-32,
// 145-149,173-176; BN+ReLU load prologue; STATS / BNRELU_BWD epilogues).  The last stage of the network has M = B*J rows (2 176 at
// B = 128: 17 row tiles of 128) and K up to 3 592: too few output tiles to fill 256 CUs, and a K loop that is one long dependent chain
// per block.  Rounds 1-5 ran it on gemm.hip's 128 x 128 kernel with a CROSS-block split-K (fp32 partial tiles out to a workspace and
// back through a finish launch: 4.3 x the algorithmic bytes, two launches per GEMM, 0.08 of the HBM roof).  Here the split is INSIDE
// the block:
//   * block tile 64 x 64 (NJ = 1) or 64 x 128 (NJ = 2), 512 threads = 8 waves = 2 k-groups x (2 x 2) waves of 32 x 32 NJ, ONE block
//     per CU; a K step covers 64 values (four 16-deep sub-tiles), k-group g multiplies sub-tiles 2g, 2g + 1 -- each wave's dependent
//     chain is half as long, 34 row tiles x N / 64 blocks fill the chip without a workspace (272 blocks for N = 512);
//   * a wave issues one instruction per four cycles, and a K step is one chain through a barrier: what bounds this loop is the
//     INSTRUCTION COUNT per step, whatever the unit (measured on the first version of this file: 190 instructions for 3 MFMAs per
//     32-value step = 0.55 us per step, the same with a third of the VALU work removed).  Hence 64 values per step, operand streams
//     advanced by pointer increments in scalar registers, every LDS stage / register-set index a compile-time constant (the loop is
//     unrolled over the stage periods), zero rows / K tails / the prologue as block-uniform branches off the common path:
//     ~80 instructions for 6 MFMAs (NJ = 1);
//   * operands as in gemm_big.hip: weights stream global -> LDS by DMA (global_load_lds_dwordx4) from the pre-split k-group-major
//     image (gast_x3_image_multi) into a ring of 4 (NJ = 1) / 3 (NJ = 2) stages, activations pass through 4 / 3 register sets
//     (BN+ReLU prologue, hi/lo split) into two LDS stages, ONE counted s_waitcnt and one barrier per K step; 64-byte row images
//     [16 hi | 16 lo] with the same XOR swizzle;
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

constexpr int ROWB = 64;                         // LDS row image of one 16-deep sub-tile: 16 x 16-bit hi | 16 x 16-bit lo
constexpr int TM = 64, NT = 1024, KG = 4, KS = 64, NSUB = KS / 16;      // 16 waves: k-group g = sub-tile g of every step
constexpr int OFF_A = 2 * TM * 4;                // crow[TM] | addrow[TM] in front
constexpr int A_BYTES = NSUB * TM * ROWB;        // 16 KB per stage
constexpr int OFF_W = OFF_A + 2 * A_BYTES;
constexpr int tn_of(int nj) { return 64 * nj; }
constexpr int ws_of(int nj) { return nj == 1 ? 4 : 3; }                  // weight stages
constexpr int na_of(int nj) { return nj == 1 ? 4 : 3; }                  // activation register sets
constexpr int w_bytes(int nj) { return NSUB * tn_of(nj) * ROWB; }        // 16 KB / 32 KB per stage
constexpr int off_tab(int nj) { return OFF_W + ws_of(nj) * w_bytes(nj); }
constexpr int LDS_BLOCK = 160 * 1024;            // one block per CU
constexpr int max_tab(int nj) { return (LDS_BLOCK - off_tab(nj)) / 8; }
constexpr int OFF_RED(int nj) { return OFF_A + KG * nj * 16 * 256 * 4; }      // column-sum scratch behind the accumulator exchange
constexpr int RED_BYTES = 8 * 128 * 2 * 4;

template <int OFF>
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(OFF) : "memory", "m0");
}
__device__ __forceinline__ void gload16s(u32x4& dst, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// a wave-uniform pointer the compiler's divergence analysis lost track of (loop-carried through a lambda): back into scalar registers
__device__ __forceinline__ const char* uni(const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

struct Frag { uint4 u; };

// EPI: 0 PLAIN, 1 STATS, 2 BNRELU_BWD, 3 BNRELU_BWD with the dropout mask of the forward re-derived
template <int EPI, int NJ>
__device__ __forceinline__ void bj_epilogue(const gast_gemm_args& a, const BjPlan& pl, unsigned char* smem, const float (&acc)[NJ][4],
                                            int m0, int n0, int mt) {
    constexpr int TN = tn_of(NJ);
    constexpr bool bwd = EPI >= 2, xdrop = EPI == 3;
    constexpr uint32_t OOB = 0x80000000u, RSRC3 = 0x00020000u;
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
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    };
    auto stv = [&](float v, const __amdgpu_buffer_rsrc_t& r, uint32_t off) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, 0);
    };
    // (a tensor that is absent gets a zero-sized descriptor: its loads return 0, its stores are dropped)
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((rowsC - 1) * a.ldc + N) * 4), RSRC3);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd ? a.X : a.C), 0, bwd ? (int)(((rowsC - 1) * a.ldx + N) * 4) : 0, RSRC3);
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(has2 ? a.C2 : a.C, 0, has2 ? (int)(((rowsC - 1) * a.ldc2 + N) * 4) : 0, RSRC3);
    const long rowsAdd = add ? (long)a.B * a.addmap.T_total * a.J : 1;
    const __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc((void*)(add ? a.addend : a.C), 0, add ? (int)(((rowsAdd - 1) * a.ldadd + N) * 4) : 0, RSRC3);
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
    // this k-group finishes the accumulator registers 4 kg .. 4 kg + 3: rows wr*32 + 8 kg + 4 lh + {0..3}
    int crow[4], arow[4];
    float xv[NJ][4], av[NJ][4];
    {
        const int base = wr * 32 + 8 * kg + 4 * lh;
        const int4 c4 = *(const int4*)(sCrow + base);
        crow[0] = c4.x; crow[1] = c4.y; crow[2] = c4.z; crow[3] = c4.w;
        const int4 a4 = *(const int4*)(sAdd + base);
        arow[0] = a4.x; arow[1] = a4.y; arow[2] = a4.z; arow[3] = a4.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t offx = crow[r] >= 0 ? (uint32_t)(crow[r] * a.ldx + col0) * 4u : OOB;
            const uint32_t offa = arow[r] >= 0 ? (uint32_t)(arow[r] * a.ldadd + col0) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                xv[q][r] = bwd ? ldv(rX, offx + 128u * q) : 0.f;
                av[q][r] = ldv(rAdd, offa + 128u * q);            // (no addend: zero-sized descriptor, reads 0)
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cr = crow[r];
        const uint32_t coff = (uint32_t)(cr * a.ldc + col0) * 4u;
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const bool ok = cr >= 0 && nin[q];
            float v = acc[q][r] + bias[q] + av[q][r];
            if (bwd) {
                stv(v, rC2, ok ? (uint32_t)(cr * a.ldc2 + col0) * 4u + 128u * q : OOB);
                const float x = xv[q][r];
                v = fmaf(x, xs[q], xh[q]) > 0.f ? v : 0.f;
                if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)(cr * a.ldx + col0 + 32 * q));
                s1[q] += ok ? v : 0.f;
                s2[q] += ok ? v * x : 0.f;
            } else if (EPI == 1) {
                s1[q] += ok ? v : 0.f;
                s2[q] += ok ? v * v : 0.f;
            }
            stv(v, rC, ok ? coff + 128u * q : OOB);
        }
    }
    if (EPI != 0) {
        // column sums of the block's 64 rows: the two lane halves by shuffle, the eight waves (kg, wr) of a column half through LDS
        // (behind the accumulator exchange), then ONE add per column into the 128-row statistics block
        float* const sRed = (float*)(smem + OFF_RED(NJ));      // [8][TN][2]
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
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g += 2) {
                t1 += sRed[(g * TN + tid) * 2] + sRed[((g + 1) * TN + tid) * 2];
                t2 += sRed[(g * TN + tid) * 2 + 1] + sRed[((g + 1) * TN + tid) * 2 + 1];
            }
            if (!(pl.ablate & 1)) {
                atomicAdd(pp, t1);
                atomicAdd(pp + 1, t2);
            }
        }
    }
}

// PAIR: 1 = bf16 hi/lo pairs (GAST_F32X3), 2 = fp16 pairs (GAST_F32X3H: forward epilogues, images of the f16 kind)
template <int NJ, int PAIR>
__device__ __forceinline__ void bj_body(const gast_gemm_args& a, const BjPlan& pl, int blk, unsigned char* smem) {
    constexpr int TN = tn_of(NJ), W_BYTES = w_bytes(NJ), OFF_TAB = off_tab(NJ), WS = ws_of(NJ), NA = na_of(NJ);
    constexpr int NWP = NJ;                          // 1 KB DMA pieces per wave and K step
    constexpr int DW = WS - 1;                       // the weights of tile t + DW are requested in step t
    constexpr int UNROLL = NJ == 1 ? 4 : 6;          // a common multiple of 2 (activation stages), WS and NA
    static_assert(UNROLL % 2 == 0 && UNROLL % WS == 0 && UNROLL % NA == 0, "unroll period");
    // transfers that may still be in flight at the top of a step: everything requested after the weights of tile t (the activations
    // of tile t + 1 are older): the activation load of that step + (DW - 1) full steps
    constexpr int INFLIGHT = 1 + (DW - 1) * (NWP + 1);
    static_assert(INFLIGHT <= (NA - 1) * (NWP + 1), "the activations of tile t + 1 must be older than the weights of tile t");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kg = w >> 2, wr = (w >> 1) & 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M;
    if (pl.ablate & 256) return;
    // tile order: a contiguous chunk of the logical order runs on one XCD (xcd_remap), i.e. shares one L2.  Every activation row is
    // read by all N / TN column tiles and every weight row by all M / 64 row tiles; the operand that is re-fetched per XCD should be the
    // SMALLER one: M >= N (most GEMMs of the stage: N = 512 / 1024 against 2 176 rows) -> the column tiles of one row tile are
    // consecutive (activations cross the fabric once, the weight panel once per XCD), else (G1: N = 5C + 8) the other way round.
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    int mt, nt;
    if (pl.M >= a.N) { mt = lb / pl.tilesN; nt = lb - mt * pl.tilesN; }
    else { nt = lb / pl.tilesM; mt = lb - nt * pl.tilesM; }
    const int m0 = mt * TM, n0 = nt * TN;

    int* const sCrow = (int*)smem;
    int* const sAdd = sCrow + TM;
    const int nseg = a.nseg;

    // ---- staging duties.  Activations: thread = (row ra_row of the tile, 16-byte chunk c8 of the 64-value step):
    // values 4 c8 .. + 3 = sub-tile c8 >> 2, chunk c8 & 3 of its 16 values
    const int ra_row = tid >> 4, c8 = tid & 15, cA = c8 & 3;
    const int TJ = a.Tn * a.J;
    int pb = -1, pt = 0, pj = 0;
    {
        const int m = m0 + ra_row;
        if (m < M) { pb = m / TJ; const int rem = m - pb * TJ; pt = rem / a.J; pj = rem - pt * a.J; }
    }
    if (c8 == 0) {          // one thread per tile row: output / addend rows
        int crow = -1, arow = -1;
        if (pb >= 0) {
            crow = (int)map_row(a.cmap, pb, pt, pj, a.J);
            if (a.addend) arow = (int)map_row(a.addmap, pb, pt, pj, a.J);
        }
        sCrow[ra_row] = crow;
        sAdd[ra_row] = arow;
    }
    // can a tile of segment s contain a row that must read as zero?  (block-uniform, conservative: the host knows which segments map
    // every position of the domain -- pl.segfull -- and only the last row tile can run past M)
    const bool tile_partial = m0 + TM > M;

    // ---- the three operand streams: {segment, values left in the segment from the current tile on, pointer}
    // activation loads (tile t + 1 + NA at step t)
    int a_seg = 0, a_rem = a.seg[0].K;
    const char* a_ptr = (const char*)a.seg[0].A;
    uint32_t offA = 0;
    int zmask = 0;                                   // bit s: this thread's row reads as zero in segment s
    auto row_of = [&](const gast_gemm_seg& sg, bool& ok) -> uint32_t {
        const int ts = pt * sg.map.t_stride + sg.map.t_off;
        ok = pb >= 0 && ts >= 0 && ts < sg.map.T_total;
        return ok ? (uint32_t)((pb * sg.map.T_total + ts) * a.J + pj) : 0u;
    };
    {
        bool ok;
        const uint32_t srow = row_of(a.seg[0], ok);
        offA = (srow * (uint32_t)a.seg[0].lda + c8 * 4) * 4u;
        if (!ok) zmask |= 1;
    }
    const int kofs0 = c8 * 4;                        // the thread's value offset inside a step
    u32x4 ra[NA];
    auto load_a = [&](u32x4& r) {
        // K tail: a chunk past K re-reads the row's first chunk of the step (its values are zeroed by the conversion); branch-free
        const uint32_t v0 = kofs0 < a_rem ? offA : offA - (uint32_t)c8 * 16u;
        const char* const ap = uni(a_ptr);
        if (!(pl.ablate & 16)) gload16s(r, v0, ap);
        a_ptr += KS * 4;
        a_rem -= KS;
        if (a_rem <= 0) {
            if (a_seg + 1 < nseg) {
                ++a_seg;
                const gast_gemm_seg& sg = a.seg[a_seg];
                a_rem = sg.K;
                a_ptr = (const char*)sg.A;
                bool ok;
                const uint32_t srow = row_of(sg, ok);
                offA = (srow * (uint32_t)sg.lda + c8 * 4) * 4u;
                if (!ok) zmask |= 1 << a_seg;
            } else { a_ptr -= KS * 4; a_rem += KS; }          // past the last tile: the last tile again
        }
    };
    // weights (tile t + DW at step t).  DMA: wave w fills sub-tile w >> 2, the NWP 1 KB pieces from row (w & 3) * TN / 4 on: 16 rows x
    // 64 B each, contiguous in the k-group-major image; lane = (row r16, slot s4), slot s4 receives source chunk s4 ^ key(row)
    const int r16 = lane >> 2, s4 = lane & 3;
    const int subW = __builtin_amdgcn_readfirstlane(w >> 2), rowW = __builtin_amdgcn_readfirstlane((w & 3) * (TN / 4));
    const uint32_t offW = (uint32_t)(n0 + rowW + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);
    const uint32_t sW0 = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + subW * (TN * ROWB) + rowW * ROWB);
    int w_seg = 0, w_rem = a.seg[0].K;
    const char* w_ptr = (const char*)a.seg[0].Wx;
    long w_ldg = (long)a.seg[0].ldwx * 2;                     // bytes per 16-value group of the weight image
    long w_sub = w_ldg * subW;                                // this wave's sub-tile inside a step
    auto dma_w = [&](int stage) {
        // (a sub-tile past the segment's last 16-value group re-reads the step's first one: its activations are written as zeros)
        const char* wbase = uni(w_ptr + (subW * 16 < w_rem ? w_sub : 0));
        const uint32_t sW = sW0 + stage * W_BYTES;
        if (!(pl.ablate & 8)) {
            glds16<0>(offW, wbase, sW);
            if (NWP == 2) glds16<1024>(offW, wbase, sW);
        }
        w_ptr += NSUB * w_ldg;
        w_rem -= KS;
        if (w_rem <= 0) {
            if (w_seg + 1 < nseg) {
                ++w_seg;
                w_rem = a.seg[w_seg].K;
                w_ptr = (const char*)a.seg[w_seg].Wx;
                w_ldg = (long)a.seg[w_seg].ldwx * 2;
                w_sub = w_ldg * subW;
            } else { w_ptr -= NSUB * w_ldg; w_rem += KS; }
        }
    };
    // conversion facts of the tile the NEXT step converts (prologue? zero rows / K tail? scale / shift of the thread's 8 values)
    int c_seg = 0, c_rem = a.seg[0].K, c_tab = pl.taboff[0] >= 0 ? OFF_TAB + pl.taboff[0] * 4 : -1;      // LDS byte address of the tile's first scale
    const int tab_sh = pl.ntab * 4;                           // shift table behind the scale table
    float4 tsc, tsh;
    bool c_pro = false, c_fix = false;
    int c_seg_now = 0, c_rem_now = 0;
    auto conv_facts = [&]() {
        c_pro = c_tab >= 0;
        c_seg_now = c_seg;
        c_rem_now = c_rem;
        c_fix = tile_partial || !((pl.segfull >> c_seg) & 1) || c_rem < KS;
        if (c_pro) {
            const unsigned char* t0 = smem + c_tab + min(kofs0 * 4, (c_rem - 4) * 4);
            tsc = *(const float4*)t0;
            tsh = *(const float4*)(t0 + tab_sh);
            c_tab += KS * 4;
        }
        c_rem -= KS;
        if (c_rem <= 0) {
            if (c_seg + 1 < nseg) {
                ++c_seg;
                c_rem = a.seg[c_seg].K;
                c_tab = pl.taboff[c_seg] >= 0 ? OFF_TAB + pl.taboff[c_seg] * 4 : -1;
            } else { c_rem += KS; if (c_pro) c_tab -= KS * 4; }
        }
    };
    const int wa_key = (ra_row >> 2) & 3;
    const int wa_base = OFF_A + (c8 >> 2) * (TM * ROWB) + ra_row * ROWB + (cA & 1) * 8;
    const int wa_hi = wa_base + (((cA >> 1) ^ wa_key) << 4), wa_lo = wa_base + (((2 + (cA >> 1)) ^ wa_key) << 4);
    auto write_a = [&](int stage, const u32x4& r) {
        float x0 = __uint_as_float(r.x), x1 = __uint_as_float(r.y), x2 = __uint_as_float(r.z), x3 = __uint_as_float(r.w);
        if (c_pro) {         // BN + ReLU prologue
            x0 = fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f);
            x1 = fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f);
            x2 = fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f);
            x3 = fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f);
        }
        if (c_fix) {         // zero rows / the K tail must read as zero (relu(shift) must not leak in)
            const bool z = ((zmask >> c_seg_now) & 1) || kofs0 >= c_rem_now;
            x0 = z ? 0.f : x0; x1 = z ? 0.f : x1; x2 = z ? 0.f : x2; x3 = z ? 0.f : x3;
        }
        uint2 hh, ll;
        split_pair4<PAIR>(x0, x1, x2, x3, hh, ll);
        *(uint2*)(smem + stage * A_BYTES + wa_hi) = hh;
        *(uint2*)(smem + stage * A_BYTES + wa_lo) = ll;
    };

    const int ntile = pl.ntile;

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

    // ---- pipeline fill: W(0), A(0), A(1) first (the wait below needs exactly these), then the rest of the look-ahead
    dma_w(0);
    load_a(ra[0]);
    load_a(ra[1]);
#pragma unroll
    for (int j = 1; j < DW; ++j) {
        dma_w(j);
        if (j + 1 < NA) load_a(ra[j + 1]);
    }
#pragma unroll
    for (int j = DW + 1; j < NA; ++j) load_a(ra[j]);
    {
        float* const sSc = (float*)(smem + OFF_TAB);
        float* const sSh = sSc + pl.ntab;
        for (int s = 0; s < nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
            if (pl.taboff[s] >= 0) {
                const float* sc = a.seg[s].scale;
                const float* sh = a.seg[s].shift;
                for (int k = tid; k < a.seg[s].K; k += NT) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
            }
        }
    }
    gload_wait_n<(DW - 1) * NWP + (NA - 2)>();
    __syncthreads();                                   // tables and row maps complete
    if (pl.ablate & 512) { gload_wait_n<0>(); return; }
    gload_pin(ra[0]);
    gload_pin(ra[1]);
    conv_facts();                                      // tile 0
    write_a(0, ra[0]);
    load_a(ra[0]);                                     // tile NA
    conv_facts();                                      // tile 1 (converted by step 0)

    // One K step (U = t % UNROLL).  At the top, after the counted wait + barrier: LDS holds tile t (activations in stage t & 1,
    // weights in stage t % WS); register set (t + 1) % NA holds tile t + 1's activations; in flight: INFLIGHT transfers at most.
    auto step = [&](auto Uc, int t) {
        constexpr int U = decltype(Uc)::value;
        constexpr int SA = U & 1, SW = U % WS, SET = (U + 1) % NA;
        gload_wait_n<INFLIGHT>();
        __syncthreads();
        gload_pin(ra[SET]);
        Frag ah, al, bh[NJ], bl[NJ];
        if (!(pl.ablate & 64)) {
            ah.u = *(const uint4*)(fA + SA * A_BYTES + ohi);
            al.u = *(const uint4*)(fA + SA * A_BYTES + olo);
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                bh[q].u = *(const uint4*)(fW + SW * W_BYTES + q * 32 * ROWB + ohi);
                bl[q].u = *(const uint4*)(fW + SW * W_BYTES + q * 32 * ROWB + olo);
            }
        }
        if (t + 1 < ntile && !(pl.ablate & 128)) write_a(SA ^ 1, ra[SET]);             // tile t+1: registers -> LDS (the set is then free for tile t+1+NA)
        if (t < ntile && !(pl.ablate & 32)) {
#pragma unroll
            for (int q = 0; q < NJ; ++q) acl[q] = mfma_pair<PAIR>(al.u, bh[q].u, acl[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acc[q] = mfma_pair<PAIR>(ah.u, bh[q].u, acc[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acl[q] = mfma_pair<PAIR>(ah.u, bl[q].u, acl[q]);
        }
        dma_w((U + DW) % WS);                                    // tile t + DW -> the stage of tile t - 1: every wave is past its reads
        load_a(ra[SET]);                                         // tile t + 1 + NA
        conv_facts();                                            // tile t + 2
    };
    for (int t = 0; t < ((pl.ablate & 4) ? 0 : ntile); t += UNROLL) {
        step(std::integral_constant<int, 0>{}, t);
        step(std::integral_constant<int, 1>{}, t + 1);
        step(std::integral_constant<int, 2>{}, t + 2);
        step(std::integral_constant<int, 3>{}, t + 3);
        if constexpr (UNROLL == 6) {
            step(std::integral_constant<int, 4>{}, t + 4);
            step(std::integral_constant<int, 5>{}, t + 5);
        }
    }
    gload_wait_n<0>();                 // (the re-requested tiles past the end: nothing may land in LDS or in registers after this point)
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
    for (int q = 0; q < NJ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] += acl[q][r];
    __syncthreads();

    // ---- the four k-groups' accumulators meet in LDS ([4][16 NJ][256] floats, conflict-free): group g finishes registers 4 g .. 4 g + 3
    // (rows 8 g .. 8 g + 3 (+ 4 for the upper lane half) of the wave tile), summed in ONE fixed order: (g0 + g1) + (g2 + g3)
    float fin[NJ][4];
    {
        float* const xch = (float*)(smem + OFF_A);
        const int t256 = tid & 255;
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[((kg * NJ + q) * 16 + r) * 256 + t256] = acc[q][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * kg + r;
                const float x0 = xch[((0 * NJ + q) * 16 + rr) * 256 + t256], x1 = xch[((1 * NJ + q) * 16 + rr) * 256 + t256];
                const float x2 = xch[((2 * NJ + q) * 16 + rr) * 256 + t256], x3 = xch[((3 * NJ + q) * 16 + rr) * 256 + t256];
                fin[q][r] = (x0 + x1) + (x2 + x3);
            }
    }
    if (pl.ablate & 2) { if (fin[0][0] == 12345.678f) ((float*)a.C)[0] = fin[0][1]; return; }
    const int v = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    if (v == 0) bj_epilogue<0, NJ>(a, pl, smem, fin, m0, n0, mt);
    else if (v == 1) bj_epilogue<1, NJ>(a, pl, smem, fin, m0, n0, mt);
    else if (v == 2) bj_epilogue<2, NJ>(a, pl, smem, fin, m0, n0, mt);
    else bj_epilogue<3, NJ>(a, pl, smem, fin, m0, n0, mt);
}

struct BjBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    BjPlan pl[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int n;
};
static_assert(sizeof(BjBatch) <= 3840, "BjBatch travels as a kernel argument (4 KB limit)");

template <int NJ, int PAIR>
__global__ void __launch_bounds__(NT, 4) gemm_bj_kernel(const gast_gemm_args a, const BjPlan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bj_body<NJ, PAIR>(a, pl, blockIdx.x, smem);
}
template <int NJ, int PAIR>
__global__ void __launch_bounds__(NT, 4) gemm_bj_multi_kernel(const BjBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    bj_body<NJ, PAIR>(b.a[d], b.pl[d], blockIdx.x - b.first[d], smem);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
std::atomic<bool> bj_setup_done[64];

typedef void (*bj_kernel_t)(const gast_gemm_args, const BjPlan);
typedef void (*bj_multi_kernel_t)(const BjBatch);
bj_kernel_t bj_kernel(int nj, int pair) {
    if (pair == 2) return nj == 1 ? gemm_bj_kernel<1, 2> : gemm_bj_kernel<2, 2>;
    return nj == 1 ? gemm_bj_kernel<1, 1> : gemm_bj_kernel<2, 1>;
}
bj_multi_kernel_t bj_multi_kernel(int nj, int pair) {
    if (pair == 2) return nj == 1 ? gemm_bj_multi_kernel<1, 2> : gemm_bj_multi_kernel<2, 2>;
    return nj == 1 ? gemm_bj_multi_kernel<1, 1> : gemm_bj_multi_kernel<2, 1>;
}
int bj_lds_bytes(int ntab, int nj) {
    const int tab = off_tab(nj) + 2 * ntab * 4, red = OFF_RED(nj) + RED_BYTES;
    return tab > red ? tab : red;
}

void bj_setup() {
    int dev = 0;
    hipGetDevice(&dev);
    dev &= 63;
    if (bj_setup_done[dev].load(std::memory_order_acquire)) return;
    for (int nj = 1; nj <= 2; ++nj)
        for (int pair = 1; pair <= 2; ++pair) {
            const hipError_t e1 = hipFuncSetAttribute((const void*)bj_kernel(nj, pair), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BLOCK);
            const hipError_t e2 = hipFuncSetAttribute((const void*)bj_multi_kernel(nj, pair), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BLOCK);
            if (e1 != hipSuccess || e2 != hipSuccess) {
                fprintf(stderr, "gast_hip: gemm_bj set-up failed for NJ %d pair %d: %s\n", nj, pair, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                (void)hipGetLastError();
            }
        }
    bj_setup_done[dev].store(true, std::memory_order_release);
    if (getenv("GAST_GEMM_BJ_DEBUG")) {
        for (int nj = 1; nj <= 2; ++nj) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)bj_kernel(nj, 1), NT, bj_lds_bytes(0, nj));
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void*)bj_kernel(nj, 1));
            fprintf(stderr, "gemm_bj NJ %d: %d blocks/CU at %d B LDS, %d regs, %zu B scratch\n", nj, nb, bj_lds_bytes(0, nj), fa.numRegs, (size_t)fa.localSizeBytes);
        }
    }
}

}  // namespace

// Can this GEMM run on the M = B*J kernel?  Fills the plan when it can (nj = 0: the caller picks the tile width for the launch).
int gast_gemm_bj_plan(const gast_gemm_args& a, BjPlan& pl) {
    static const int enabled = getenv("GAST_GEMM_BJ") ? atoi(getenv("GAST_GEMM_BJ")) : 1;
    static const int max_rows = getenv("GAST_GEMM_BJ_MAX_M") ? atoi(getenv("GAST_GEMM_BJ_MAX_M")) : 8191;
    if (!enabled || (a.dtype != GAST_F32X3 && a.dtype != GAST_F32X3H)) return 0;
    if (a.dtype == GAST_F32X3H && a.epi == GAST_EPI_BNRELU_BWD) return 0;     // (a gradient operand does not fit fp16's range)
    if (a.out_f32 || a.f8_scale) return 0;
    pl.pair = a.dtype == GAST_F32X3H ? 2 : 1;
    const long Ml = (long)a.B * a.Tn * a.J;
    if (Ml < 1 || Ml > max_rows || a.N < 1) return 0;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG || !a.C) return 0;
    int ntab = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.Wx || !aligned16(g.Wx) || g.ldwx % 8 || !g.A || !aligned16(g.A) || g.lda % 4 || g.K % 4 || g.K < 4) return 0;
        if ((long)a.B * g.map.T_total * a.J * g.lda * 4 >= 0xffffffffL) return 0;      // 32-bit byte offsets into the activation tensor
        if (g.pro == GAST_PRO_BNRELU_DROP) return 0;
        pl.taboff[s] = -1;
        if (g.pro == GAST_PRO_BNRELU) {
            if (!g.scale || !g.shift) return 0;
            for (int q = 0; q < s; ++q)
                if (pl.taboff[q] >= 0 && a.seg[q].scale == g.scale && a.seg[q].shift == g.shift && a.seg[q].K == g.K) pl.taboff[s] = pl.taboff[q];
            if (pl.taboff[s] < 0) { pl.taboff[s] = ntab; ntab += (g.K + 3) / 4 * 4; }
        }
    }
    if (ntab > max_tab(2)) return 0;
    if (a.epi < 0 || a.epi > GAST_EPI_BNRELU_BWD) return 0;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return 0;
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    if (rowsC * a.ldc * 4 >= 0x7fffffffL || (a.epi == GAST_EPI_BNRELU_BWD && rowsC * a.ldx * 4 >= 0x7fffffffL)) return 0;
    if (a.C2 && (a.epi != GAST_EPI_BNRELU_BWD || rowsC * a.ldc2 * 4 >= 0x7fffffffL)) return 0;
    if (a.addend && (long)a.B * a.addmap.T_total * a.J * a.ldadd * 4 >= 0x7fffffffL) return 0;
    pl.M = (int)Ml;
    pl.ntile = 0;
    pl.segfull = 0;
    for (int s = 0; s < a.nseg; ++s) {
        pl.ntile += (a.seg[s].K + KS - 1) / KS;
        // does the segment's row map send EVERY frame of the domain to a frame of its tensor?  (then no tile of it has zero rows)
        const gast_rowmap& mp = a.seg[s].map;
        const long lo = mp.t_stride >= 0 ? mp.t_off : (long)(a.Tn - 1) * mp.t_stride + mp.t_off;
        const long hi = mp.t_stride >= 0 ? (long)(a.Tn - 1) * mp.t_stride + mp.t_off : mp.t_off;
        if (lo >= 0 && hi < mp.T_total) pl.segfull |= 1 << s;
    }
    pl.tilesM = (pl.M + TM - 1) / TM;
    pl.ntab = ntab;
    pl.nj = 0;
    pl.tilesN = 0;
    static const int ablate = getenv("GAST_GEMM_BJ_ABLATE") ? atoi(getenv("GAST_GEMM_BJ_ABLATE")) : 0;
    pl.ablate = ablate;
    return 1;
}

// tile width of a launch: 64 columns while the grid fits one round of resident blocks (one per CU), else 128
static int bj_pick_nj(const gast_gemm_args* args, const BjPlan* pls, int n) {
    static const int nj_env = getenv("GAST_GEMM_BJ_NJ") ? atoi(getenv("GAST_GEMM_BJ_NJ")) : 0;
    static const int max_blocks = getenv("GAST_GEMM_BJ_BLOCKS") ? atoi(getenv("GAST_GEMM_BJ_BLOCKS")) : 288;
    if (nj_env == 1 || nj_env == 2) return nj_env;
    long blocks = 0;
    for (int d = 0; d < n; ++d) blocks += (long)pls[d].tilesM * ((args[d].N + 63) / 64);
    return blocks <= max_blocks ? 1 : 2;
}

int gast_gemm_bj_launch_multi(const gast_gemm_args* args, BjPlan* pls, int n, hipStream_t st) {
    bj_setup();
    const int nj = bj_pick_nj(args, pls, n), pair = pls[0].pair;
    BjBatch b;
    b.n = n;
    b.first[0] = 0;
    int ntab = 0;
    for (int d = 0; d < n; ++d) {
        if (pls[d].pair != pair) return GAST_EINVAL;
        pls[d].nj = nj;
        pls[d].tilesN = (args[d].N + tn_of(nj) - 1) / tn_of(nj);
        b.a[d] = args[d];
        b.pl[d] = pls[d];
        b.first[d + 1] = b.first[d] + pls[d].tilesM * pls[d].tilesN;
        if (pls[d].ntab > ntab) ntab = pls[d].ntab;
    }
    if (n == 1) hipLaunchKernelGGL(bj_kernel(nj, pair), dim3(b.first[1]), dim3(NT), bj_lds_bytes(ntab, nj), st, b.a[0], b.pl[0]);
    else hipLaunchKernelGGL(bj_multi_kernel(nj, pair), dim3(b.first[n]), dim3(NT), bj_lds_bytes(ntab, nj), st, b);
    GAST_CHECK_LAUNCH();
    return 0;
}
