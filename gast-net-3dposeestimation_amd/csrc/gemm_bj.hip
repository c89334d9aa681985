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

namespace {

constexpr int ROWB = 64;                         // LDS row image of one 16-deep k-group: 16 x 16-bit hi | 16 x 16-bit lo
constexpr int TM = 64, NT = 512, KG = 2;
constexpr int OFF_A = 2 * TM * 4;                // crow[TM] | addrow[TM] in front
constexpr int A_BYTES = KG * TM * ROWB;          // 8 KB per stage
constexpr int OFF_W = OFF_A + 2 * A_BYTES;
constexpr int tn_of(int nj) { return 64 * nj; }
constexpr int w_bytes(int nj) { return KG * tn_of(nj) * ROWB; }          // 8 KB / 16 KB per stage
constexpr int off_tab(int nj) { return OFF_W + 3 * w_bytes(nj); }
constexpr int LDS_BLOCK = 80 * 1024;             // two blocks per CU
constexpr int max_tab(int nj) { return (LDS_BLOCK - off_tab(nj)) / 8; }

template <int OFF>
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(OFF) : "memory", "m0");
}
__device__ __forceinline__ void gload16s(u32x4& dst, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

struct Frag { uint4 u; };

// EPI: 0 PLAIN, 1 STATS, 2 BNRELU_BWD, 3 BNRELU_BWD with the dropout mask of the forward re-derived
template <int EPI, int NJ>
__device__ __forceinline__ void bj_epilogue(const gast_gemm_args& a, const BjPlan& pl, unsigned char* smem, const f32x16 (&acc)[NJ],
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
            const uint32_t offx = crow[u][r] >= 0 ? (uint32_t)(crow[u][r] * a.ldx + col0) * 4u : OOB;
            const uint32_t offa = arow[u][r] >= 0 ? (uint32_t)(arow[u][r] * a.ldadd + col0) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                xv[u][q][r] = bwd ? ldv(rX, offx + 128u * q) : 0.f;
                av[u][q][r] = ldv(rAdd, offa + 128u * q);            // (no addend: zero-sized descriptor, reads 0)
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cr = crow[u][r];
            const uint32_t coff = (uint32_t)(cr * a.ldc + col0) * 4u;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const bool ok = cr >= 0 && nin[q];
                float v = acc[q][8 * kg + 4 * u + r] + bias[q] + av[u][q][r];
                if (bwd) {
                    stv(v, rC2, ok ? (uint32_t)(cr * a.ldc2 + col0) * 4u + 128u * q : OOB);
                    const float x = xv[u][q][r];
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
    }
    if (EPI != 0) {
        // column sums of the block's 64 rows: the two lane halves by shuffle, the four waves (kg, wr) of a column half through LDS
        // (the K loop's stages are free by now), then ONE add per column into the 128-row statistics block
        float* const sRed = (float*)(smem + OFF_W + 2 * w_bytes(NJ));      // [4][TN][2]: the third weight stage (the accumulator exchange uses the space in front)
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
            atomicAdd(pp, t1);
            atomicAdd(pp + 1, t2);
        }
    }
}

// PAIR: 1 = bf16 hi/lo pairs (GAST_F32X3), 2 = fp16 pairs (GAST_F32X3H: forward epilogues, images of the f16 kind)
template <int NJ, int PAIR>
__device__ __forceinline__ void bj_body(const gast_gemm_args& a, const BjPlan& pl, int blk, unsigned char* smem) {
    constexpr int TN = tn_of(NJ), W_BYTES = w_bytes(NJ), OFF_TAB = off_tab(NJ);
    constexpr int NW = NJ;                           // 1 KB DMA pieces per wave and K step
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kg = w >> 2, wr = (w >> 1) & 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M;
    // tile order: the row tiles of one column panel are consecutive, and a contiguous chunk of that order runs on one XCD -- the
    // weight panel (the fat operand of this stage) is fetched into one or two L2s, the activation rows are small
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    const int nt = lb / pl.tilesM, mt = lb - nt * pl.tilesM;
    const int m0 = mt * TM, n0 = nt * TN;

    int* const sCrow = (int*)smem;
    int* const sAdd = sCrow + TM;
    float* const sSc = (float*)(smem + OFF_TAB);
    float* const sSh = sSc + pl.ntab;

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

    // ---- staging duties.  Activations: thread = (row ra of the tile, 16-byte chunk c8 of the 32-value step: k-group c8 >> 2,
    // chunk c8 & 3 of its 16 values)
    const int ra_row = tid >> 3, c8 = tid & 7, kgA = c8 >> 2, cA = c8 & 3;
    int pb = -1, pt = 0, pj = 0;
    {
        const int m = m0 + ra_row;
        if (m < M) { pb = m / TJ; const int rem = m - pb * TJ; pt = rem / a.J; pj = rem - pt * a.J; }
    }
    uint32_t offA = 0;
    bool zrow = true;
    int seg_a = -1;
    auto enter_a = [&](int s) {
        if (s == seg_a) return;
        seg_a = s;
        const gast_gemm_seg& sg = a.seg[s];
        const int ts = pt * sg.map.t_stride + sg.map.t_off;
        const bool ok = pb >= 0 && ts >= 0 && ts < sg.map.T_total;
        const uint32_t srow = ok ? (uint32_t)((pb * sg.map.T_total + ts) * a.J + pj) : 0u;
        zrow = !ok;                                      // out-of-range tap (or a row past M): reads as zero
        offA = (srow * (uint32_t)sg.lda + c8 * 4) * 4u;
    };
    // Weights (DMA): wave w fills the 1 KB pieces (w & 3) * NW + i of k-group w >> 2: 16 rows x 64 B each, contiguous in the
    // k-group-major image; lane = (row r16, slot s4), slot s4 receives source chunk s4 ^ key(row)
    const int r16 = lane >> 2, s4 = lane & 3;
    const int kgW = __builtin_amdgcn_readfirstlane(w >> 2), pieceW = __builtin_amdgcn_readfirstlane((w & 3) * NW);      // (wave-uniform: scalar registers)
    const uint32_t offW = (uint32_t)(n0 + pieceW * 16 + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);

    struct Tile { int seg, k0, K, toff; const char* abase; const char* wbase; };
    int ntile = 0;
    for (int s = 0; s < a.nseg; ++s) ntile += (a.seg[s].K + 31) / 32;
    int seg_l = 0, k_l = 0, gen = 0, K_l = a.seg[0].K, toff_l = pl.taboff[0];
    const char* A_l = (const char*)a.seg[0].A;
    const char* W_l = (const char*)a.seg[0].Wx;
    long ldg_l = (long)a.seg[0].ldwx * 2;                     // bytes per k-group of the weight image
    Tile last_tile = {0, 0, K_l, toff_l, A_l, W_l};
    auto next_tile = [&](Tile& t) {                          // tiles in order; past the end: the last tile again
        if (gen >= ntile) { t = last_tile; return; }
        t.seg = seg_l; t.k0 = k_l; t.K = K_l; t.toff = toff_l;
        t.abase = A_l + k_l * 4;
        // this wave's k-group of the step; a step whose second half lies past the segment's last 16-value group re-reads the
        // first one (the activations of that half are written as zeros)
        const int g = (k_l >> 4) + kgW;
        t.wbase = W_l + (long)(g * 16 < K_l ? g : (k_l >> 4)) * ldg_l;
        last_tile = t;
        ++gen;
        k_l += 32;
        if (k_l >= K_l && seg_l + 1 < a.nseg) {
            k_l = 0; ++seg_l;
            K_l = a.seg[seg_l].K; toff_l = pl.taboff[seg_l];
            A_l = (const char*)a.seg[seg_l].A; W_l = (const char*)a.seg[seg_l].Wx; ldg_l = (long)a.seg[seg_l].ldwx * 2;
        }
    };

    u32x4 ra[2];
    bool rz[2];
    auto load_a = [&](const Tile& t, u32x4& r, bool& z) {
        enter_a(t.seg);
        const bool kin = t.k0 + c8 * 4 < t.K;
        gload16s(r, kin ? offA : offA - c8 * 16, t.abase);       // (past the K tail: the step's first chunk, the values are zeroed)
        z = zrow || !kin;
    };
    auto dma_w = [&](const Tile& t, int stage) {
        const uint32_t sW = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + stage * W_BYTES + kgW * (TN * ROWB) + pieceW * 1024);
        glds16<0>(offW, t.wbase, sW);
        if (NW == 2) glds16<1024>(offW, t.wbase, sW);
    };
    float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_tab = [&](const Tile& t) {
        if (t.toff >= 0) {
            const int k = t.toff + min(t.k0 + c8 * 4, t.K - 4);
            tsc = *(const float4*)(sSc + k);
            tsh = *(const float4*)(sSh + k);
        }
    };
    const int wa_key = (ra_row >> 2) & 3;
    const int wa_base = kgA * (TM * ROWB) + ra_row * ROWB + (cA & 1) * 8;
    const int wa_hi = wa_base + (((cA >> 1) ^ wa_key) << 4), wa_lo = wa_base + (((2 + (cA >> 1)) ^ wa_key) << 4);
    auto write_a = [&](const Tile& t, int stage, const u32x4& r, bool z) {
        unsigned char* sA = smem + OFF_A + stage * A_BYTES;
        const bool pro = t.toff >= 0;
        float x0 = __uint_as_float(r.x), x1 = __uint_as_float(r.y), x2 = __uint_as_float(r.z), x3 = __uint_as_float(r.w);
        // BN + ReLU prologue; zero rows / the K tail must read as zero (relu(shift) must not leak in)
        x0 = z ? 0.f : (pro ? fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f) : x0);
        x1 = z ? 0.f : (pro ? fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f) : x1);
        x2 = z ? 0.f : (pro ? fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f) : x2);
        x3 = z ? 0.f : (pro ? fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f) : x3);
        uint2 h, l;
        split_pair4<PAIR>(x0, x1, x2, x3, h, l);
        *(uint2*)(sA + wa_hi) = h;
        *(uint2*)(sA + wa_lo) = l;
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int fkey = (li >> 2) & 3;
    const int ohi = li * ROWB + ((lh ^ fkey) << 4), olo = li * ROWB + (((2 + lh) ^ fkey) << 4);
    const int fa_off = kg * (TM * ROWB) + wr * 32 * ROWB, fw_off = kg * (TN * ROWB) + wc * (TN / 2) * ROWB;

    // ---- pipeline (gemm_big.hip, prefetch distance 2).  Tile j's activations travel in register set j & 1 and LDS stage j & 1,
    // its weights in stage j % 3.  At the top of step t (after the counted wait + barrier): LDS holds tile t; set (t+1) & 1 holds
    // tile t+1's activations; in flight: the weights of tile t+1, the activations of tile t+2.
    Tile dq[3];
    {
        Tile d0;
        next_tile(d0);
        dma_w(d0, 0);
        load_a(d0, ra[0], rz[0]);
        next_tile(dq[0]);
        dma_w(dq[0], 1);
        load_a(dq[0], ra[1], rz[1]);
        for (int s = 0; s < a.nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
            if (pl.taboff[s] >= 0) {
                const float* sc = a.seg[s].scale;
                const float* sh = a.seg[s].shift;
                for (int k = tid; k < a.seg[s].K; k += NT) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
            }
        }
        gload_wait_n<0>();
        __syncthreads();                                   // tables and row maps complete
        gload_pin(ra[0]);
        gload_pin(ra[1]);
        fetch_tab(d0);
        write_a(d0, 0, ra[0], rz[0]);
        next_tile(dq[1]);
        load_a(dq[1], ra[0], rz[0]);
        next_tile(dq[2]);
        fetch_tab(dq[0]);
    }
    int wstage = 0;
    auto step = [&](int t, bool wr_next, bool do_mma, u32x4& r, bool& z) {
        gload_wait_n<NW + 1>();          // the newest step's DMA pieces + activation load stay in flight
        __syncthreads();
        gload_pin(r);
        const unsigned char* sA = smem + OFF_A + (t & 1) * A_BYTES + fa_off;
        const unsigned char* sW = smem + OFF_W + wstage * W_BYTES + fw_off;
        Frag ah, al, bh[NJ], bl[NJ];
        ah.u = *(const uint4*)(sA + ohi);
        al.u = *(const uint4*)(sA + olo);
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            bh[q].u = *(const uint4*)(sW + q * 32 * ROWB + ohi);
            bl[q].u = *(const uint4*)(sW + q * 32 * ROWB + olo);
        }
        if (wr_next) write_a(dq[0], (t + 1) & 1, r, z);          // tile t+1: registers -> LDS (the set is then free for tile t+3)
        if (do_mma) {
            // small terms first
#pragma unroll
            for (int q = 0; q < NJ; ++q) acc[q] = mfma_pair<PAIR>(al.u, bh[q].u, acc[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acc[q] = mfma_pair<PAIR>(ah.u, bl[q].u, acc[q]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) acc[q] = mfma_pair<PAIR>(ah.u, bh[q].u, acc[q]);
        }
        dma_w(dq[1], wstage == 0 ? 2 : wstage - 1);              // tile t + 2 -> stage (t + 2) % 3
        load_a(dq[2], r, z);
        wstage = wstage == 2 ? 0 : wstage + 1;
        dq[0] = dq[1]; dq[1] = dq[2];
        next_tile(dq[2]);
        fetch_tab(dq[0]);
    };
    for (int t = 0; t < ntile; t += 2) {
        step(t, t + 1 < ntile, true, ra[1], rz[1]);
        step(t + 1, t + 2 < ntile, t + 1 < ntile, ra[0], rz[0]);
    }
    gload_wait_n<0>();                 // (the re-requested tiles past the end: nothing may land in LDS or in registers after this point)
    asm volatile("" ::"v"(ra[0]), "v"(ra[1]));
    __syncthreads();

    // ---- the two k-groups' accumulators meet: group g keeps registers 8 g .. 8 g + 7 (rows 16 g .. 16 g + 15 of the wave tile) and
    // hands the other half over through LDS ([8 NJ][256] floats per direction, conflict-free)
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
    const int v = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    if (v == 0) bj_epilogue<0, NJ>(a, pl, smem, acc, m0, n0, mt);
    else if (v == 1) bj_epilogue<1, NJ>(a, pl, smem, acc, m0, n0, mt);
    else if (v == 2) bj_epilogue<2, NJ>(a, pl, smem, acc, m0, n0, mt);
    else bj_epilogue<3, NJ>(a, pl, smem, acc, m0, n0, mt);
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
int bj_lds_bytes(int ntab, int nj) { return off_tab(nj) + 2 * ntab * 4; }

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
    pl.tilesM = (pl.M + TM - 1) / TM;
    pl.ntab = ntab;
    pl.nj = 0;
    pl.tilesN = 0;
    return 1;
}

// tile width of a launch: 64 columns while the grid fits the 512 resident blocks of the chip (two per CU), else 128
static int bj_pick_nj(const gast_gemm_args* args, const BjPlan* pls, int n) {
    static const int nj_env = getenv("GAST_GEMM_BJ_NJ") ? atoi(getenv("GAST_GEMM_BJ_NJ")) : 0;
    static const int max_blocks = getenv("GAST_GEMM_BJ_BLOCKS") ? atoi(getenv("GAST_GEMM_BJ_BLOCKS")) : 560;
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
