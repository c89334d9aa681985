// gast_gemm, big-tile path for GAST_F32X3 (fp32 storage, split-bf16 products on v_mfma_f32_32x32x16_bf16), gfx950.
//
// Same contract as gemm.hip (K segments with row maps = channel concat / temporal taps of reference gast_net.py:28-32,145-148,
// 173-174; BN+ReLU load prologue; STATS / BNRELU_BWD epilogues), different machine mapping -- the 128x128 two-barrier loop of
// gemm.hip runs the split-bf16 products at 19 % of the matrix-core peak (rocprof, profiles/r02_v0_*):
//   * block tile 256x256, 512 threads = 8 waves (2 x 4), wave tile 128x64 = 4x2 MFMA tiles, one block per CU;
//   * K tile = 32 fp32 values per row, held in LDS as a 128-byte row image [32 bf16 hi | 32 bf16 lo]; 16-byte chunks are
//     XOR-swizzled by (row>>1)&7 so the fragment reads (ds_read_b128, 16 lanes per pass) are conflict-free;
//   * TWO LDS stages, ONE barrier per K tile: while tile t is multiplied, tile t+1's weights stream global -> LDS by DMA
//     (global_load_lds_dwordx4 from the pre-split weight image, gast_x3_image) and tile t+2's activations are in flight to
//     registers (inline-asm loads, counted together with the DMA by one s_waitcnt vmcnt(0) at the top of the next iteration);
//     activations go through registers because the BN+ReLU prologue and the hi/lo split are VALU work, and are written to
//     the other stage right after the barrier;
//   * 48 MFMAs per wave per K tile (3 products x 4x2 tiles x 2 k-steps) against 24 ds_read_b128 and 4+4 16-byte loads per
//     thread: the loop is matrix-core bound by construction (21 B/clk/CU of operand traffic, L2 delivers 32);
//   * epilogue straight from the accumulators: in the 32x32 layout a lane owns one column, so a store instruction writes two
//     128-byte row segments of fp32 -- no LDS staging needed; the column statistics of a wave's 128 rows are exactly one
//     128-row statistics block (partials[ceil(M/128)][N][2], the layout gemm.hip and the BatchNorm finalizes share).
#include "common.h"
#include "gemm_big.h"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int ROWB = 128;
constexpr int TILE_BYTES = 256 * ROWB;          // one operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // A | W
constexpr int LDS_FIXED = 2 * STAGE_BYTES + 2 * TM * 4;
constexpr int LDS_MAX = 160 * 1024;
constexpr int MAX_TAB = (LDS_MAX - LDS_FIXED) / 8;   // floats of scale (and as many of shift) the block can keep in LDS

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "memory", "m0");
}

// EPI: 0 PLAIN, 1 STATS, 2 BNRELU_BWD, 3 BNRELU_BWD with the dropout mask of the forward re-derived (compile-time: the
// epilogue is straight-line code per element)
template <int EPI>
__device__ __forceinline__ void big_body(const gast_gemm_args& a, const BigPlan& pl, int blk, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M, N = a.N;
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    const int mt = lb / pl.tilesN, nt = lb - mt * pl.tilesN;
    const int m0 = mt * TM, n0 = nt * TN;

    float* const sSc = (float*)(smem + 2 * STAGE_BYTES);
    float* const sSh = sSc + pl.ntab;
    int* const sCrow = (int*)(sSh + pl.ntab);
    int* const sAdd = sCrow + TM;

    const int TJ = a.Tn * a.J;
    if (tid < TM) {
        const int m = m0 + tid;
        int crow = -1, arow = -1;
        if (m < M) {
            const int b = m / TJ, rem = m - b * TJ, t = rem / a.J, j = rem - t * a.J;
            crow = (int)map_row(a.cmap, b, t, j, a.J);
            if (a.addend) arow = (int)map_row(a.addmap, b, t, j, a.J);
        }
        sCrow[tid] = crow;
        sAdd[tid] = arow;
    }

    // ---- this thread's staging duties.  A (registers): rows rbase + 64 i, 16-byte chunk `c` (4 fp32 values) of the K tile.
    const int c = tid & 7, rbase = tid >> 3;
    int pb[4], pt[4], pj[4];
    bool mvalid[4];
    uint32_t aoff_hi[4], aoff_lo[4];           // byte offsets of this thread's two 8-byte pieces inside an A tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rbase + 64 * i;
        const int m = m0 + row;
        mvalid[i] = m < M;
        const int mm = mvalid[i] ? m : 0;
        pb[i] = mm / TJ;
        const int rem = mm - pb[i] * TJ;
        pt[i] = rem / a.J;
        pj[i] = rem - pt[i] * a.J;
        const int key = (row >> 1) & 7;
        aoff_hi[i] = row * ROWB + (((c >> 1) ^ key) << 4) + (c & 1) * 8;
        aoff_lo[i] = row * ROWB + (((4 + (c >> 1)) ^ key) << 4) + (c & 1) * 8;
    }
    // W (DMA): wave w fills the 8-row pieces (w*4 + i); lane = (row r8, slot s8) and slot s8 receives source chunk s8 ^ key(row)
    const int r8 = lane >> 3, s8 = lane & 7;
    int wrow[4], wchunk[4];
    uint32_t wlds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + r8;
        wchunk[i] = s8 ^ ((row >> 1) & 7);
        const int n = n0 + row;
        wrow[i] = n < N ? n : N - 1;                       // rows past N: clamped (their columns are never stored)
        wlds[i] = (w * 4 + i) * 8 * ROWB;
    }

    // segment state: pointers of the segment entered last
    const float* pA[4];
    const bf16_t* pW[4];
    bool zrow[4];
    int cur_seg = -1;
    auto enter_seg = [&](int s) {
        if (s == cur_seg) return;
        cur_seg = s;
        const gast_gemm_seg& sg = a.seg[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = pt[i] * sg.map.t_stride + sg.map.t_off;
            const bool ok = mvalid[i] && ts >= 0 && ts < sg.map.T_total;
            const long srow = ok ? ((long)pb[i] * sg.map.T_total + ts) * a.J + pj[i] : 0;
            zrow[i] = !ok;                                   // out-of-range tap (or a row past M): reads as zero
            pA[i] = (const float*)sg.A + srow * sg.lda + c * 4;
            pW[i] = (const bf16_t*)sg.Wx + (long)wrow[i] * sg.ldwx + wchunk[i] * 8;
        }
    };
    struct Tile { int seg, k0; };
    int seg_l = 0, k_l = 0;
    auto next_tile = [&](Tile& t) {
        t.seg = seg_l; t.k0 = k_l;
        k_l += TK;
        if (k_l >= a.seg[seg_l].K) { k_l = 0; ++seg_l; }
    };
    int ntile = 0;
    for (int s = 0; s < a.nseg; ++s) ntile += (a.seg[s].K + TK - 1) / TK;

    u32x4 ra[4];
    bool rz[4];
    auto load_a = [&](const Tile& t) {                       // (enter_seg(t.seg) ran before)
        const int K = a.seg[t.seg].K;
        const bool kin = t.k0 + c * 4 < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gload16(ra[i], kin ? pA[i] + t.k0 : pA[i] - c * 4);     // (past the K tail: any valid address, the values are zeroed)
            rz[i] = zrow[i] || !kin;
        }
    };
    auto dma_w = [&](const Tile& t, int stage) {
        const uint32_t sW = lds0 + stage * STAGE_BYTES + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(pW[i] + t.k0 * 2, __builtin_amdgcn_readfirstlane(sW + wlds[i]));
    };
    auto write_a = [&](const Tile& t, int stage) {
        unsigned char* sA = smem + stage * STAGE_BYTES;
        const int toff = pl.taboff[t.seg];
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (toff >= 0) {
            const int k = toff + min(t.k0 + c * 4, a.seg[t.seg].K - 4);
            sc = *(const float4*)(sSc + k);
            sh = *(const float4*)(sSh + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x0 = __uint_as_float(ra[i].x), x1 = __uint_as_float(ra[i].y), x2 = __uint_as_float(ra[i].z), x3 = __uint_as_float(ra[i].w);
            if (toff >= 0) {
                x0 = fmaxf(fmaf(x0, sc.x, sh.x), 0.f);
                x1 = fmaxf(fmaf(x1, sc.y, sh.y), 0.f);
                x2 = fmaxf(fmaf(x2, sc.z, sh.z), 0.f);
                x3 = fmaxf(fmaf(x3, sc.w, sh.w), 0.f);
            }
            if (rz[i]) { x0 = 0.f; x1 = 0.f; x2 = 0.f; x3 = 0.f; }      // (relu(shift) must not leak into zero rows / the K tail)
            uint2 h, l;
            h.x = pack_bf16x2(x0, x1);
            h.y = pack_bf16x2(x2, x3);
            l.x = pack_bf16x2(x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xffff0000u));
            l.y = pack_bf16x2(x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xffff0000u));
            *(uint2*)(sA + aoff_hi[i]) = h;
            *(uint2*)(sA + aoff_lo[i]) = l;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    int offA[4], keyA[4], offB[2], keyB[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { const int row = wr * 128 + mi * 32 + li; offA[mi] = row * ROWB; keyA[mi] = (row >> 1) & 7; }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { const int row = wc * 64 + ni * 32 + li; offB[ni] = row * ROWB; keyB[ni] = (row >> 1) & 7; }

    // ---- pipeline.  Invariant at the top of iteration t (after the wait + barrier): stage t&1 holds tile t; the register set
    // holds the activations of tile t+1.  Tiles 0 and 1 are requested back to back (one exposed latency, not two).
    Tile cur, nxt, nn;
    next_tile(cur);
    enter_seg(cur.seg);
    dma_w(cur, 0);
    load_a(cur);
    bool have_nxt = ntile > 1, have_nn = false;
    u32x4 rb[4];
    bool rzb[4];
    if (have_nxt) {
        next_tile(nxt);
        enter_seg(nxt.seg);
        const int K1 = a.seg[nxt.seg].K;
        const bool kin1 = nxt.k0 + c * 4 < K1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gload16(rb[i], kin1 ? pA[i] + nxt.k0 : pA[i] - c * 4);
            rzb[i] = zrow[i] || !kin1;
        }
    }
    for (int s = 0; s < a.nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
        if (pl.taboff[s] >= 0) {
            const float* sc = a.seg[s].scale;
            const float* sh = a.seg[s].shift;
            for (int k = tid; k < a.seg[s].K; k += 512) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
        }
    }
    gload_wait_n<0>();
    __syncthreads();                                   // tables complete
    write_a(cur, 0);
    if (have_nxt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = rb[i]; rz[i] = rzb[i]; }
    }
    for (int t = 0; t < ntile; ++t) {
        gload_wait_n<0>();
        __syncthreads();
        if (have_nxt) {
            const int st = (t + 1) & 1;
            write_a(nxt, st);
            enter_seg(nxt.seg);                        // (a no-op unless tile t+2's segment moved the pointers on)
            dma_w(nxt, st);
            have_nn = t + 2 < ntile;
            if (have_nn) {
                next_tile(nn);
                enter_seg(nn.seg);
                load_a(nn);
            }
        }
        const unsigned char* sA = smem + (t & 1) * STAGE_BYTES;
        const unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; s16x8 s; } ah[4], al[4], bh[2], bl[2];
            const int ch = ks * 2 + lh;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                ah[mi].u = *(const uint4*)(sA + offA[mi] + ((ch ^ keyA[mi]) << 4));
                al[mi].u = *(const uint4*)(sA + offA[mi] + (((4 + ch) ^ keyA[mi]) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                bh[ni].u = *(const uint4*)(sW + offB[ni] + ((ch ^ keyB[ni]) << 4));
                bl[ni].u = *(const uint4*)(sW + offB[ni] + (((4 + ch) ^ keyB[ni]) << 4));
            }
            // small terms first; consecutive MFMAs on different accumulators
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bl[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
        }
        nxt = nn; have_nxt = have_nn; have_nn = false;
    }

    // ---- epilogue, straight from the accumulators (lane = column li of the MFMA tile; register r = row (r&3) + 8 (r>>2) + 4 lh).
    // Branch-free and batched: per 32-row group `mi` the 2 x 16 X / addend values of the lane are loaded unconditionally from
    // clamped addresses one group AHEAD of their use, rows that must not be stored (past M, unmapped by cmap) and columns past N
    // only predicate the stores and the statistics.
    constexpr bool bwd = EPI >= 2, xdrop = EPI == 3;
    const uint32_t thresh = a.drop.thresh;
    const float inv_keep = a.drop.inv_keep;
    const uint32_t xkey = xdrop ? drop_key(a.drop, a.xsalt) : 0u;
    float* const Cb = (float*)a.C;
    const float* const Addb = (const float*)a.addend;
    const float* const Xb = (const float*)a.X;
    int ncol[2], ncl[2];
    bool nin[2];
    float bias[2], xs[2], xh[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ncol[ni] = n0 + wc * 64 + ni * 32 + li;
        nin[ni] = ncol[ni] < N;
        ncl[ni] = nin[ni] ? ncol[ni] : N - 1;
        bias[ni] = a.bias ? (a.bias_neg ? -a.bias[ncl[ni]] : a.bias[ncl[ni]]) : 0.f;
        xs[ni] = bwd ? a.xscale[ncl[ni]] : 0.f;
        xh[ni] = bwd ? a.xshift[ncl[ni]] : 0.f;
    }
    // unit u = 8 rows of one 32-row group: mi = u >> 1, registers r = 8 (u & 1) .. + 7
    int crow[2][8], arow[2][8];
    float xv[2][2][8], av[2][2][8];
    auto fetch = [&](int u, int buf) {
        const int mi = u >> 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int base = wr * 128 + mi * 32 + 8 * (2 * (u & 1) + q) + 4 * lh;
            const int4 c4 = *(const int4*)(sCrow + base);
            crow[buf][4 * q] = c4.x; crow[buf][4 * q + 1] = c4.y; crow[buf][4 * q + 2] = c4.z; crow[buf][4 * q + 3] = c4.w;
            if (Addb) {
                const int4 a4 = *(const int4*)(sAdd + base);
                arow[buf][4 * q] = a4.x; arow[buf][4 * q + 1] = a4.y; arow[buf][4 * q + 2] = a4.z; arow[buf][4 * q + 3] = a4.w;
            }
        }
        if (bwd) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) xv[buf][ni][r] = Xb[(long)max(crow[buf][r], 0) * a.ldx + ncl[ni]];
        }
        if (Addb) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) av[buf][ni][r] = Addb[(long)max(arow[buf][r], 0) * a.ldadd + ncl[ni]];
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) { arow[buf][r] = -1; av[buf][0][r] = 0.f; av[buf][1][r] = 0.f; }
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int buf = u & 1, mi = u >> 1;
        if (u + 1 < 8) fetch(u + 1, buf ^ 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int cr = crow[buf][r];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bool ok = cr >= 0 && nin[ni];
                float v = acc[mi][ni][8 * (u & 1) + r] + bias[ni];
                v += arow[buf][r] >= 0 ? av[buf][ni][r] : 0.f;
                if (bwd) {
                    const float x = xv[buf][ni][r];
                    v = fmaf(x, xs[ni], xh[ni]) > 0.f ? v : 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)((long)max(cr, 0) * a.ldx + ncl[ni]));
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * x : 0.f;
                } else if (EPI == 1) {
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * v : 0.f;
                }
                if (ok) Cb[(long)cr * a.ldc + ncol[ni]] = v;
            }
        }
    }
    if (EPI != 0 && m0 + wr * 128 < M) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32);
            s2[ni] += __shfl_xor(s2[ni], 32);
            if (lh == 0 && nin[ni]) {
                float* pp = a.partials + ((long)(mt * 2 + wr) * N + ncol[ni]) * 2;
                pp[0] = s1[ni];
                pp[1] = s2[ni];
            }
        }
    }
}

__device__ __forceinline__ int epi_variant(const gast_gemm_args& a) {
    if (a.epi == GAST_EPI_BNRELU_BWD) return (a.xdrop && a.drop.thresh != 0) ? 3 : 2;
    return a.epi;
}

template <int EPI>
__global__ void __launch_bounds__(512) gemm_big_kernel(const gast_gemm_args a, const BigPlan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    big_body<EPI>(a, pl, blockIdx.x, smem);
}

struct BigBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    BigPlan pl[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int n;
};
static_assert(sizeof(BigBatch) <= 3840, "BigBatch travels as a kernel argument (4 KB limit)");
__global__ void __launch_bounds__(512) gemm_big_multi_kernel(const BigBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const int blk = blockIdx.x - b.first[d];
    switch (epi_variant(b.a[d])) {
        case 0: big_body<0>(b.a[d], b.pl[d], blk, smem); break;
        case 1: big_body<1>(b.a[d], b.pl[d], blk, smem); break;
        case 2: big_body<2>(b.a[d], b.pl[d], blk, smem); break;
        default: big_body<3>(b.a[d], b.pl[d], blk, smem); break;
    }
}

// ---- pre-split weight image: img[r][(k>>5)*64 + (k&31)] = bf16 hi(W[r][k]),  + 32: bf16 lo; zero for K <= k < Kp
struct ImageBatch { gast_x3_image_job j[GAST_X3_IMAGE_MAX_BATCH]; int first[GAST_X3_IMAGE_MAX_BATCH + 1]; int n; };
static_assert(sizeof(ImageBatch) <= 3840, "ImageBatch travels as a kernel argument");
__global__ void __launch_bounds__(256) x3_image_kernel(const ImageBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const gast_x3_image_job& j = b.j[d];
    const int Kp4 = (j.K + 31) / 32 * 8;                          // 4-value chunks per padded row
    const long idx = (long)(blockIdx.x - b.first[d]) * 256 + threadIdx.x;
    if (idx >= (long)j.R * Kp4) return;
    const int r = (int)(idx / Kp4), k = (int)(idx - (long)r * Kp4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < j.K) v = *(const float4*)(j.W + (long)r * j.ldw + k);
    uint2 h, l;
    h.x = pack_bf16x2(v.x, v.y);
    h.y = pack_bf16x2(v.z, v.w);
    l.x = pack_bf16x2(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf16x2(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
    bf16_t* o = (bf16_t*)j.img + (long)r * j.ldimg + (k >> 5) * 64 + (k & 31);
    *(uint2*)o = h;
    *(uint2*)(o + 32) = l;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool big_setup_done[64] = {};

}  // namespace

// Can this GEMM run on the big-tile kernel?  Fills the plan when it can.  (Called by gast_gemm_ws / gast_gemm_multi in gemm.hip.)
int gast_gemm_big_plan(const gast_gemm_args& a, BigPlan& pl) {
    static const int enabled = getenv("GAST_GEMM_BIG") ? atoi(getenv("GAST_GEMM_BIG")) : 1;
    static const int min_rows = getenv("GAST_GEMM_BIG_MIN_M") ? atoi(getenv("GAST_GEMM_BIG_MIN_M")) : 8192;
    if (!enabled || a.dtype != GAST_F32X3) return 0;
    const long Ml = (long)a.B * a.Tn * a.J;
    if (Ml < min_rows || Ml > 0x7fffff00L || a.N < 32) return 0;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG) return 0;
    int ntab = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.Wx || !aligned16(g.Wx) || g.ldwx % 8 || !g.A || !aligned16(g.A) || g.lda % 4 || g.K % 4 || g.K < 4) return 0;
        if (g.pro == GAST_PRO_BNRELU_DROP) return 0;
        pl.taboff[s] = -1;
        if (g.pro == GAST_PRO_BNRELU) {
            if (!g.scale || !g.shift) return 0;
            for (int q = 0; q < s; ++q)
                if (pl.taboff[q] >= 0 && a.seg[q].scale == g.scale && a.seg[q].shift == g.shift && a.seg[q].K == g.K) pl.taboff[s] = pl.taboff[q];
            if (pl.taboff[s] < 0) { pl.taboff[s] = ntab; ntab += (g.K + 3) / 4 * 4; }
        }
    }
    if (ntab > MAX_TAB) return 0;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return 0;
    pl.M = (int)Ml;
    pl.tilesM = (pl.M + TM - 1) / TM;
    pl.tilesN = (a.N + TN - 1) / TN;
    pl.ntab = ntab;
    return 1;
}

static int big_lds_bytes(int ntab) { return LDS_FIXED + 2 * ntab * 4; }

static void big_setup() {
    int dev = 0;
    hipGetDevice(&dev);               // function attributes are per device (nn.DataParallel replicas launch on several)
    dev &= 63;
    if (big_setup_done[dev]) return;
    hipFuncSetAttribute((const void*)gemm_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    hipFuncSetAttribute((const void*)gemm_big_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    hipFuncSetAttribute((const void*)gemm_big_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    hipFuncSetAttribute((const void*)gemm_big_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    hipFuncSetAttribute((const void*)gemm_big_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    big_setup_done[dev] = true;
}

int gast_gemm_big_launch(const gast_gemm_args& a, const BigPlan& pl, hipStream_t st) {
    big_setup();
    const dim3 grid(pl.tilesM * pl.tilesN), block(512);
    const int lds = big_lds_bytes(pl.ntab);
    const int v = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    if (v == 0) hipLaunchKernelGGL(gemm_big_kernel<0>, grid, block, lds, st, a, pl);
    else if (v == 1) hipLaunchKernelGGL(gemm_big_kernel<1>, grid, block, lds, st, a, pl);
    else if (v == 2) hipLaunchKernelGGL(gemm_big_kernel<2>, grid, block, lds, st, a, pl);
    else hipLaunchKernelGGL(gemm_big_kernel<3>, grid, block, lds, st, a, pl);
    GAST_CHECK_LAUNCH();
    return 0;
}

int gast_gemm_big_launch_multi(const gast_gemm_args* args, const BigPlan* pls, int n, hipStream_t st) {
    big_setup();
    BigBatch b;
    b.n = n;
    b.first[0] = 0;
    int ntab = 0;
    for (int d = 0; d < n; ++d) {
        b.a[d] = args[d];
        b.pl[d] = pls[d];
        b.first[d + 1] = b.first[d] + pls[d].tilesM * pls[d].tilesN;
        if (pls[d].ntab > ntab) ntab = pls[d].ntab;
    }
    hipLaunchKernelGGL(gemm_big_multi_kernel, dim3(b.first[n]), dim3(512), big_lds_bytes(ntab), st, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" long gast_x3_image_ld(int K) { return (long)((K + 31) / 32) * 64; }

extern "C" int gast_x3_image_multi(const gast_x3_image_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 0) return GAST_EINVAL;
    for (int i0 = 0; i0 < n; i0 += GAST_X3_IMAGE_MAX_BATCH) {
        ImageBatch b;
        b.n = n - i0 < GAST_X3_IMAGE_MAX_BATCH ? n - i0 : GAST_X3_IMAGE_MAX_BATCH;
        b.first[0] = 0;
        for (int d = 0; d < b.n; ++d) {
            const gast_x3_image_job& j = jobs[i0 + d];
            if (!j.W || !j.img || j.R < 1 || j.K < 4) return GAST_EINVAL;
            if (j.K % 4 || j.ldw % 4 || !aligned16(j.W) || !aligned16(j.img) || j.ldimg % 8 || j.ldimg < gast_x3_image_ld(j.K)) return GAST_EALIGN;
            b.j[d] = j;
            const long chunks = (long)j.R * ((j.K + 31) / 32 * 8);
            b.first[d + 1] = b.first[d] + (int)((chunks + 255) / 256);
        }
        hipLaunchKernelGGL(x3_image_kernel, dim3(b.first[b.n]), dim3(256), 0, (hipStream_t)stream, b);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}
