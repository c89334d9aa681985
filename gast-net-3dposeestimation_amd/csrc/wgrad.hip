// gast_wgrad: weight gradients of the channel-mixing GEMMs on MFMA (gfx950).
//
//   dW[r, wcol0_s + k] += sum_m P[pmap(m), r] * pro_s(Q_s[map_s(m), k])
//
// P is the output gradient of the forward GEMM ("dC"), Q_s its activation operands with the same BN+ReLU(+dropout)
// load prologue and row maps as in the forward (so a dilated-conv tap or a concat segment is just another
// (Q_s, map_s, wcol0_s)).  The reduction runs over the position axis m = (b,t,j), i.e. over the NON-contiguous axis
// of both operands:
//   * fp32: v_mfma_f32_32x32x2_f32 takes one float per lane per operand, so the [m][col] tiles are used as they
//     are (conflict-free ds_read_b32 fragment reads, row stride 132 floats);
//   * bf16: v_mfma_f32_32x32x16_bf16 wants 8 consecutive m per lane; every thread transposes an 8(m) x 8(col)
//     block in registers while staging, so LDS holds [col][m] rows of 128 B (+16 B pad) and the MFMA loop is the
//     same as gast_gemm's.
// M is split over blockIdx (split-M) and partial 128x128 tiles are combined with fp32 atomics into the zero-filled
// gradient buffer.
#include "common.h"
#include "wgrad_common.h"
#include <type_traits>

namespace {

constexpr int BT = 128;      // output tile (rows of dW x cols of dW)
constexpr int LSTR = 144;    // bf16: LDS row stride in bytes
constexpr int FSTR = 132;    // fp32: LDS row stride in floats

// ------------------------------------------------------------------------------------------------ fp32
__device__ __forceinline__ void wgrad_f32_body(const gast_wgrad_args& a, int M, int tilesS_total, int mchunk, int tile, int sp) {
    constexpr int BKM = 32;
    __shared__ __attribute__((aligned(16))) float sP[BKM * FSTR];
    __shared__ __attribute__((aligned(16))) float sQ[BKM * FSTR];
    __shared__ int sRowP[2][BKM];
    __shared__ int sRowQ[2][BKM];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const TileCoord tc = decode_tile(a, tile, tilesS_total, BT);
    const gast_wgrad_seg& sg = a.seg[tc.seg];
    const int m_begin = sp * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    if (m_begin >= m_end) return;
    const int ntile = (m_end - m_begin + BKM - 1) / BKM;

    const float* Pb = (const float*)a.P;
    const float* Qb = (const float*)sg.Q;
    const int c = tid & 31, rb0 = tid >> 5;
    const int pcol = tc.rt * BT + c * 4;
    const int qcol = tc.st * BT + c * 4;
    const bool pin = pcol < a.R, qin = qcol < sg.S;
    const bool pro = sg.pro != GAST_PRO_NONE;
    const bool drop = sg.pro == GAST_PRO_BNRELU_DROP && a.drop.thresh != 0;
    const uint32_t key = drop ? drop_key(a.drop, sg.salt) : 0u;
    float4 sc = make_float4(0, 0, 0, 0), sh = make_float4(0, 0, 0, 0);
    if (pro && qin) { sc = *(const float4*)(sg.scale + qcol); sh = *(const float4*)(sg.shift + qcol); }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto compute_rows = [&](int it, int buf) {
        if (tid < BKM) {
            int pr, qr;
            int m = m_begin + it * BKM + tid;
            rows_for(a, sg, m < m_end ? m : M, M, pr, qr);
            sRowP[buf][tid] = pr;
            sRowQ[buf][tid] = qr;
        }
    };

    // raw, unconditional (clamped) asm loads; zero rows / column tails are applied when the tile is written to LDS
    u32x4 rp[4], rq[4];
    const int pcolc = pin ? pcol : 0, qcolc = qin ? qcol : 0;
    auto load_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = rb0 + 8 * i;
            int pr = sRowP[buf][r], qr = sRowQ[buf][r];
            gload16(rp[i], Pb + (long)(pr < 0 ? 0 : pr) * a.ldp + pcolc);
            gload16(rq[i], Qb + (long)(qr < 0 ? 0 : qr) * sg.ldq + qcolc);
        }
    };
    auto store_tile = [&](int buf) {
        gload_wait_n<0>();
#pragma unroll
        for (int i = 0; i < 4; ++i) { gload_pin(rp[i]); gload_pin(rq[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = rb0 + 8 * i;
            const int pr_ = sRowP[buf][r], qr_ = sRowQ[buf][r];
            const bool okp = pin && pr_ >= 0, okq = qin && qr_ >= 0;
            float4 pv = make_float4(okp ? __uint_as_float(rp[i].x) : 0.f, okp ? __uint_as_float(rp[i].y) : 0.f,
                                    okp ? __uint_as_float(rp[i].z) : 0.f, okp ? __uint_as_float(rp[i].w) : 0.f);
            float4 q = make_float4(okq ? __uint_as_float(rq[i].x) : 0.f, okq ? __uint_as_float(rq[i].y) : 0.f,
                                   okq ? __uint_as_float(rq[i].z) : 0.f, okq ? __uint_as_float(rq[i].w) : 0.f);
            if (pro && qin) {
                int qr = qr_;
                if (qr >= 0) {
                    q.x = fmaxf(fmaf(q.x, sc.x, sh.x), 0.f);
                    q.y = fmaxf(fmaf(q.y, sc.y, sh.y), 0.f);
                    q.z = fmaxf(fmaf(q.z, sc.z, sh.z), 0.f);
                    q.w = fmaxf(fmaf(q.w, sc.w, sh.w), 0.f);
                    if (drop) {
                        uint32_t e0 = (uint32_t)((long)qr * sg.ldq + qcol);
                        q.x *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0);
                        q.y *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + 1);
                        q.z *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + 2);
                        q.w *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + 3);
                    }
                }
            }
            *(float4*)(sP + r * FSTR + c * 4) = pv;
            *(float4*)(sQ + r * FSTR + c * 4) = q;
        }
    };

    compute_rows(0, 0);
    __syncthreads();
    load_tile(0);
    if (ntile > 1) compute_rows(1, 1);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        store_tile(it & 1);
        __syncthreads();
        if (it + 1 < ntile) load_tile((it + 1) & 1);
        if (it + 2 < ntile) compute_rows(it + 2, it & 1);
#pragma unroll 4
        for (int kk = 0; kk < BKM / 2; ++kk) {
            float fa[2], fb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = sP[(kk * 2 + lh) * FSTR + wr * 64 + mi * 32 + li];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[ni] = sQ[(kk * 2 + lh) * FSTR + wc * 64 + ni * 32 + li];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
        }
    }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int scol = tc.st * BT + wc * 64 + ni * 32 + li;
        if (scol >= sg.S) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrow = tc.rt * BT + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rrow < a.R) atomicAdd(a.dW + (long)rrow * a.ldw + sg.wcol0 + scol, acc[mi][ni][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------ fp32 storage, split-bf16 MFMA
// GAST_F32X3: the operands are fp32 in memory; every thread owns an 8(m) x 4(col) block per 32-row step, applies the prologue in
// fp32, splits each value into bf16 hi + lo (x - hi is exact in fp32) and -- because v_cvt_pk_bf16_f32 packs two DIFFERENT source
// registers -- gets the m-major -> col-major transposition for free: LDS rows are [col][hi: 32 m = 64 B | lo: 64 B] (+16 B pad) and
// the MFMA loop is gast_gemm's split loop: hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
// TILE = edge of the square dW tile: 128 (4 waves as 2x2 of 64x64, three blocks per CU) or 256 (8 waves as 2x4 of 128x64, one block
// per CU: the round-2 form of the wide tile, no longer launched -- wgrad_wide.hip is its pipelined successor).
// DROP: some segment re-derives a dropout mask in its prologue (compile-time: without it the staging pass needs neither the per-row
// element offsets -- eight serialised LDS reads per step -- nor the hash)
template <int TILE, bool DROP>
__device__ __forceinline__ void wgrad_x3_body(unsigned char* smem, const gast_wgrad_args& a, int M, int tilesS_total, int mchunk, int tile, int sp) {
    constexpr int BKM = 32;
    constexpr int WGC = TILE / 64, MI = TILE / 64;     // wave columns; 32-row MFMA tiles per wave (wave tile: TILE/2 rows x 64 columns)
    __shared__ __attribute__((aligned(16))) int sRowP[2][BKM];
    __shared__ __attribute__((aligned(16))) int sRowQ[2][BKM];
    __shared__ int sBad[2];
    unsigned char* const sP = smem;
    unsigned char* const sQ = smem + TILE * LSTR;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w / WGC, wc = w % WGC;
    const int li = lane & 31, lh = lane >> 5;
    const TileCoord tc = decode_tile(a, tile, tilesS_total, TILE);
    const gast_wgrad_seg& sg = a.seg[tc.seg];
    const int m_begin = sp * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    if (m_begin >= m_end) return;
    const int ntile = (m_end - m_begin + BKM - 1) / BKM;

    // staging role: the first TILE threads stage P, the others Q; each owns an 8(m) x 4(col) block of the 32 x TILE step tile
    const int op = tid / TILE, task = tid - op * TILE;
    const int mb = task & 3, rc = task >> 2;
    const float* base = op == 0 ? (const float*)a.P : (const float*)sg.Q;
    const int ld = op == 0 ? a.ldp : sg.ldq;
    const int col = (op == 0 ? tc.rt : tc.st) * TILE + rc * 4;
    const bool cin = col < (op == 0 ? a.R : sg.S);
    const bool pro = op == 1 && sg.pro != GAST_PRO_NONE;
    const bool drop = DROP && op == 1 && sg.pro == GAST_PRO_BNRELU_DROP && a.drop.thresh != 0;
    const uint32_t key = drop ? drop_key(a.drop, sg.salt) : 0u;
    float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (pro && cin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = sg.scale[col + q]; sh[q] = sg.shift[col + q]; }
    }
    unsigned char* const sdst = op == 0 ? sP : sQ;

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto compute_rows = [&](int it, int buf) {
        if (tid < 64) {      // wave 0; lanes 32..63 only take part in the ballot
            int pr = 0, qr = 0;
            if (tid < BKM) {
                const int m = m_begin + it * BKM + tid;
                rows_for(a, sg, m < m_end ? m : M, M, pr, qr);
                sRowP[buf][tid] = pr;
                sRowQ[buf][tid] = qr;
            }
            const unsigned long long bad = __ballot(pr < 0);
            if (tid == 0) sBad[buf] = bad != 0ull;
        }
    };

    u32x4 rl[8];
    const int colc = cin ? col : 0;
    // (the operand a thread stages is the same for its whole wave: the base pointer can live in scalar registers)
    const float* const sbase = (const float*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)base >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)base));
    auto load_tile = [&](int buf) {
        const int4* rp = (const int4*)((op == 0 ? sRowP[buf] : sRowQ[buf]) + mb * 8);
        const int4 r0 = rp[0], r1 = rp[1];
        const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)      // 32-bit byte offset + uniform base (wgrad_check bounds the operands to 4 GB)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rl[i]) : "v"((uint32_t)((rows[i] < 0 ? 0 : rows[i]) * ld + colc) * 4u), "s"(sbase) : "memory");
    };
    auto store_tile = [&](int buf) {
        float x[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i][0] = __uint_as_float(rl[i].x); x[i][1] = __uint_as_float(rl[i].y);
            x[i][2] = __uint_as_float(rl[i].z); x[i][3] = __uint_as_float(rl[i].w);
        }
        const bool bad = __builtin_amdgcn_readfirstlane(sBad[buf]) != 0;
        if (pro && cin) {
            if (DROP && drop) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = sRowQ[buf][mb * 8 + i];
                    const uint32_t e0 = (uint32_t)((long)(row < 0 ? 0 : row) * ld + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        x[i][q] = fmaxf(fmaf(x[i][q], sc[q], sh[q]), 0.f) * drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + q);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[i][q] = fmaxf(fmaf(x[i][q], sc[q], sh[q]), 0.f);
            }
        }
        // rows outside the chunk / the row map must read as zero (after the prologue: relu(shift) must not leak in)
        if (bad) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = op == 0 ? sRowP[buf][mb * 8 + i] : sRowQ[buf][mb * 8 + i];
                if (row < 0) { x[i][0] = 0.f; x[i][1] = 0.f; x[i][2] = 0.f; x[i][3] = 0.f; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const float x0 = x[2 * p2][q], x1 = x[2 * p2 + 1][q];
                h[p2] = pack_bf16x2(x0, x1);
                l[p2] = pack_bf16x2(x0 - __uint_as_float(h[p2] << 16), x1 - __uint_as_float(h[p2] & 0xffff0000u));
            }
            unsigned char* d = sdst + (rc * 4 + q) * LSTR + mb * 16;
            *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
            *(uint4*)(d + 64) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; s16x8 s; } ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const unsigned char* p = sP + (wr * (TILE / 2) + mi * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                ah[mi].u = *(const uint4*)p;
                al[mi].u = *(const uint4*)(p + 64);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const unsigned char* p = sQ + (wc * 64 + ni * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                bh[ni].u = *(const uint4*)p;
                bl[ni].u = *(const uint4*)(p + 64);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bl[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
        }
    };

    compute_rows(0, 0);
    __syncthreads();
    load_tile(0);
    if (ntile > 1) compute_rows(1, 1);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        gload_wait_n<0>();
#pragma unroll
        for (int i = 0; i < 8; ++i) gload_pin(rl[i]);
        store_tile(it & 1);
        __syncthreads();
        if (it + 1 < ntile) load_tile((it + 1) & 1);
        if (it + 2 < ntile) compute_rows(it + 2, it & 1);
        mfma_tile();
    }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int scol = tc.st * TILE + wc * 64 + ni * 32 + li;
        if (scol >= sg.S) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrow = tc.rt * TILE + wr * (TILE / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rrow < a.R) atomicAdd(a.dW + (long)rrow * a.ldw + sg.wcol0 + scol, acc[mi][ni][r]);
            }
    }
}

// Pipelined form of wgrad_x3_body (TILE 128): TWO LDS stages, TWO register sets, ONE barrier per step.  The step of wgrad_x3_body is
//   wait(all loads) -> convert -> LDS write -> barrier -> issue loads -> MFMA -> barrier
// with every phase of a wave serialised behind a barrier; here the MFMAs of tile t (stage t&1) and the wait / prologue / hi-lo
// split / transposing LDS write of tile t+1 (stage (t+1)&1) form ONE branch-free region, followed by a single barrier, and the
// loads of tiles t+2 and t+3 stay in flight across it (counted s_waitcnt vmcnt(8), as in gemm_big).  What makes the region
// branch-free:
//   * the operands are read with raw buffer loads; wave 0 turns the row map of each 32-row step into BYTE offsets and gives rows
//     outside the chunk / the row map an out-of-range offset: the hardware returns zeros for them without touching memory, so
//     neither the step tail nor an out-of-range tap needs a branch (only the dC operand has to read as zero: the product with a
//     finite relu(shift) then vanishes);
//   * the two staging roles are two instantiations of the loop (waves 0-1 stage dC: no prologue; waves 2-3 stage the activation:
//     fma + max against a lower clamp that is -inf for a prologue-free segment);
//   * the step count is padded to an even number and the loads / row tables run two tiles past the end (all out of range).
// 72 KB of LDS: two blocks per CU.
#ifndef WG_ABLATE
#define WG_ABLATE 0
#endif
constexpr int wgrad_x3_pipe_lds_bytes() { return 2 * 2 * BT * LSTR; }
constexpr uint32_t WG_OOB = 0xFFFF0000u;      // buffer size == first out-of-range byte offset (wgrad_check bounds the operands)
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));

// NP = split products per FLOP pair: 3 (hi*hi + hi*lo + lo*hi: fp32-class, the default), 2 (the activation operand as its bf16 hi
// part only: dC_hi*q_hi + dC_lo*q_hi) or 1 (plain bf16 products): GAST_WGRAD_X3_PRODUCTS.  A weight gradient feeds nothing but the
// optimizer -- the forward outputs and the input-gradient chain do not see it -- so fewer products is a precision lever for dW alone.
template <bool DROP, int NP>
__device__ __forceinline__ void wgrad_x3_pipe_body(unsigned char* smem, const gast_wgrad_args& a, int M, int tilesS_total, int mchunk, int tile, int sp) {
    constexpr int BKM = 32;
    constexpr int STAGE = 2 * BT * LSTR;
    __shared__ __attribute__((aligned(16))) uint32_t sOffP[2][BKM];
    __shared__ __attribute__((aligned(16))) uint32_t sOffQ[2][BKM];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const TileCoord tc = decode_tile(a, tile, tilesS_total, BT);
    const gast_wgrad_seg& sg = a.seg[tc.seg];
    const int m_begin = sp * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    if (m_begin >= m_end) return;
    const int ntile = ((m_end - m_begin + BKM - 1) / BKM + 1) & ~1;      // even

    const int op = __builtin_amdgcn_readfirstlane(w >> 1);               // 0: dC (P) staging waves, 1: activation (Q) staging waves
    const int task = tid & (BT - 1);
    // lane -> (8-row block mb, column quad rc): the four lanes of a QUAD read 64 contiguous bytes of ONE row.  The texture addresser
    // coalesces a 16-byte-per-lane load over adjacent lanes only: with the row block as the fastest index (lanes 0..3 = four different
    // rows) every lane was its own request and the launches took 11 % longer (round 4: 166 / 555 / 142 -> 144 / 491 / 128 us on the
    // three stages, scripts/r4_wg_map.sh; 8 or 16 lanes per row piece are no better than 4).  The LDS side does not care: a group
    // of 8 lanes still writes 8 different 4-bank groups.
    const int mb = (task >> 2) & 3, rc = ((task >> 4) << 2) | (task & 3);
    const float* base = op == 0 ? (const float*)a.P : (const float*)sg.Q;
    const int ld = op == 0 ? a.ldp : sg.ldq;
    const int col = (op == 0 ? tc.rt : tc.st) * BT + rc * 4;
    const bool cin = col < (op == 0 ? a.R : sg.S);
    const bool pro = op == 1 && sg.pro != GAST_PRO_NONE;
    const bool drop = DROP && op == 1 && sg.pro == GAST_PRO_BNRELU_DROP && a.drop.thresh != 0;
    const uint32_t key = drop ? drop_key(a.drop, sg.salt) : 0u;
    // prologue constants of this thread's 4 columns; identity (scale 1, shift 0, clamp -inf) without a prologue; columns past S
    // feed only outputs that are never stored
    float sc[4], sh[4];
    const float lowclamp = pro ? 0.f : -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = 1.f; sh[q] = 0.f; }
    if (pro && cin) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = sg.scale[col + q]; sh[q] = sg.shift[col + q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(sc[q]), "+v"(sh[q]));      // (the compiler's wait for these loads lands here, not in the loop)
    const uint32_t colbytes = (uint32_t)(cin ? col : 0) * 4u;
    const int sdst_off = (op == 0 ? 0 : BT * LSTR) + rc * 4 * LSTR + mb * 16;
    // buffer resource of this wave's operand: base, stride 0, WG_OOB bytes, raw 32-bit dwords
    const float* const sbase = (const float*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)base >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)base));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, (int)WG_OOB, 0x00020000);

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const uint32_t ldp4 = (uint32_t)a.ldp * 4u, ldq4 = (uint32_t)sg.ldq * 4u;
    auto compute_rows = [&](int it, int buf) {      // wave 0 only
        if (tid < BKM) {
            const int m = m_begin + it * BKM + tid;
            int pr, qr;
            rows_for(a, sg, m < m_end ? m : M, M, pr, qr);
            sOffP[buf][tid] = pr < 0 ? WG_OOB : (uint32_t)pr * ldp4;
            sOffQ[buf][tid] = qr < 0 ? WG_OOB : (uint32_t)qr * ldq4;
        }
    };

    struct Set { u32x4 rl[8]; uint32_t off[8]; };
    Set S0, S1;
    auto load_tile = [&](Set& S, int buf) {
        const uint4* rp = (const uint4*)((op == 0 ? sOffP[buf] : sOffQ[buf]) + mb * 8);
        const uint4 r0 = rp[0], r1 = rp[1];
        S.off[0] = r0.x + colbytes; S.off[1] = r0.y + colbytes; S.off[2] = r0.z + colbytes; S.off[3] = r0.w + colbytes;
        S.off[4] = r1.x + colbytes; S.off[5] = r1.y + colbytes; S.off[6] = r1.z + colbytes; S.off[7] = r1.w + colbytes;
#if WG_ABLATE == 1        // profiling build: no operand loads at all
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(S.rl[i]));
#elif WG_ABLATE == 2      // profiling build: every load inside one 64 KB window (cache-resident operands)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            S.rl[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(S.off[i] & 0xFFF0u), 0, 0));
#else
#pragma unroll
        for (int i = 0; i < 8; ++i)
            S.rl[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)S.off[i], 0, 0));
#endif
    };
    auto store_tile = [&](const Set& S, unsigned char* stage, auto opc) {
        constexpr int OP = decltype(opc)::value;
#if WG_ABLATE == 4        // profiling build: no conversion, no LDS write (the loads are still waited for)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(S.rl[i]));
        return;
#endif
        float x[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i][0] = __uint_as_float(S.rl[i].x); x[i][1] = __uint_as_float(S.rl[i].y);
            x[i][2] = __uint_as_float(S.rl[i].z); x[i][3] = __uint_as_float(S.rl[i].w);
        }
        if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) x[i][q] = fmaxf(fmaf(x[i][q], sc[q], sh[q]), lowclamp);
            if (DROP && drop) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t e0 = S.off[i] >> 2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[i][q] *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + q);
                }
            }
        }
        constexpr bool LO = NP == 3 || (NP == 2 && OP == 0);      // is this operand's lo part multiplied at all?
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const float x0 = x[2 * p2][q], x1 = x[2 * p2 + 1][q];
                h[p2] = pack_bf16x2(x0, x1);
                if (LO) l[p2] = pack_bf16x2(x0 - __uint_as_float(h[p2] << 16), x1 - __uint_as_float(h[p2] & 0xffff0000u));
            }
            unsigned char* d = stage + sdst_off + q * LSTR;
            *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
            if (LO) *(uint4*)(d + 64) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    auto mfma_tile = [&](const unsigned char* stage) {
#if WG_ABLATE == 3        // profiling build: no fragment reads, no MFMAs
        return;
#endif
        const unsigned char* const sP = stage;
        const unsigned char* const sQ = stage + BT * LSTR;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; s16x8 s; } ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const unsigned char* p = sP + (wr * 64 + mi * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                ah[mi].u = *(const uint4*)p;
                if (NP >= 2) al[mi].u = *(const uint4*)(p + 64);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const unsigned char* p = sQ + (wc * 64 + ni * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                bh[ni].u = *(const uint4*)p;
                if (NP == 3) bl[ni].u = *(const uint4*)(p + 64);
            }
            if (NP >= 2) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
            }
            if (NP == 3) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bl[ni].s, acc[mi][ni], 0, 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bh[ni].s, acc[mi][ni], 0, 0, 0);
        }
    };
    unsigned char* const stage0 = smem;
    unsigned char* const stage1 = smem + STAGE;

    // prologue: offsets of tiles 0,1 -> loads 0,1 -> offsets of tiles 2,3 -> tile 0 into stage 0 -> load 2
    if (w == 0) { compute_rows(0, 0); compute_rows(1, 1); }
    __syncthreads();
    load_tile(S0, 0);
    load_tile(S1, 1);
    __syncthreads();
    if (w == 0) { compute_rows(2, 0); compute_rows(3, 1); }
    auto run = [&](auto opc) {
        store_tile(S0, stage0, opc);
        __syncthreads();
        load_tile(S0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // steady state, two steps per trip.  Step t: MFMA on stage t&1 | tile t+1 (set (t+1)&1) -> stage (t+1)&1; barrier; load
        // tile t+3 into the freed set; offsets of tile t+4.  (Tiles >= ntile: all offsets out of range, zeros, no traffic.)
        for (int it = 0; it < ntile; it += 2) {
            mfma_tile(stage0);
            store_tile(S1, stage1, opc);
            __syncthreads();
            load_tile(S1, 1);
            if (decltype(opc)::value == 0 && w == 0) compute_rows(it + 4, 0);
            __builtin_amdgcn_sched_barrier(0);      // (the next step's conversions must not move up: they would wait for the loads just issued)
            mfma_tile(stage1);
                store_tile(S0, stage0, opc);
            __syncthreads();
            load_tile(S0, 0);
            if (decltype(opc)::value == 0 && w == 0) compute_rows(it + 5, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (op == 0) run(std::integral_constant<int, 0>());
    else run(std::integral_constant<int, 1>());

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int scol = tc.st * BT + wc * 64 + ni * 32 + li;
        if (scol >= sg.S) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrow = tc.rt * BT + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rrow < a.R) atomicAdd(a.dW + (long)rrow * a.ldw + sg.wcol0 + scol, acc[mi][ni][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------ bf16
__device__ __forceinline__ void transpose8x8_bf16(const u32x4 (&in)[8], uint4 (&out)[8]) {
    // in[i] = row i (8 bf16: cols 0..7 packed in 4 u32); out[q] = col q (8 bf16: rows 0..7).  One v_perm_b32 per output word
    // (the shift/mask form costs two VALU instructions per word, and this kernel is VALU-bound in its staging pass).
    uint32_t w[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[i][0] = in[i].x; w[i][1] = in[i].y; w[i][2] = in[i].z; w[i][3] = in[i].w; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t lo = w[2 * p][q >> 1], hi = w[2 * p + 1][q >> 1];
            // (q odd) high halves: (lo >> 16) | (hi & 0xffff0000); (q even) low halves: (lo & 0xffff) | (hi << 16)
            o[p] = __builtin_amdgcn_perm(hi, lo, (q & 1) ? 0x07060302u : 0x05040100u);
        }
        out[q] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// TILE = edge of the square dW tile: 128 (4 waves as 2x2 of 64x64, three blocks per CU) or 256 (8 waves as 2x4 of 128x64, one
// block per CU: half the operand bytes per FLOP -- the 128x128 blocks are bound by their global loads).  2*TILE threads: the
// first TILE stage P, the others Q, one 8(m) x 8(col) block each per 64-row step.
constexpr int WG_BKM = 64;
constexpr int wgrad_bf16_lds_bytes(int tile, int ring = 1) { return 2 * tile * LSTR + 4 * ring * WG_BKM * (int)sizeof(int) + 16 * ring; }

// RING = register sets in flight: 1 (the tile of step t+1 is loaded while step t multiplies; 168 VGPRs, three blocks per CU) or
// 2 (tiles t+1 AND t+2 in flight, the wait before the LDS write is a counted vmcnt(8): the window a load has to land in grows
// from the MFMA phase to a whole step; 2 x 32 more VGPRs, two blocks per CU).  Row-map buffers: 2 * RING.
template <int TILE, int RING = 1>
__device__ __forceinline__ void wgrad_bf16_body(unsigned char* __restrict__ smem, const gast_wgrad_args& a, int M, int tilesS_total,
                                                int mchunk, int tile, int sp) {
    constexpr int BKM = WG_BKM;
    constexpr int WGC = TILE / 64;            // wave columns (each wave: TILE/2 rows x 64 columns of the dW tile)
    constexpr int WROWS = TILE / 2;
    constexpr int MI = WROWS / 32;
    unsigned char* const sP = smem;
    unsigned char* const sQ = smem + TILE * LSTR;
    int (*sRowP)[BKM] = (int (*)[BKM])(smem + 2 * TILE * LSTR);
    constexpr int NRB = 2 * RING;             // row-map buffers
    int (*sRowQ)[BKM] = sRowP + NRB;
    int* const sBad = (int*)(sRowQ + NRB);    // [NRB]: does the 64-row step hold a row that must read as zero?

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w / WGC, wc = w - wr * WGC;
    const int li = lane & 31, lh = lane >> 5;
    const TileCoord tc = decode_tile(a, tile, tilesS_total, TILE);
    const gast_wgrad_seg& sg = a.seg[tc.seg];
    const int m_begin = sp * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    if (m_begin >= m_end) return;
    const int ntile = (m_end - m_begin + BKM - 1) / BKM;

    // staging role: the first TILE threads stage P, the others Q; each thread owns an 8(m) x 8(col) block
    const int op = tid / TILE, task = tid - op * TILE;
    // lane -> (8-row block mb, 8-column group rc): a quad of lanes reads 64 contiguous bytes of one row (see wgrad_x3_pipe_body);
    // column groups 2, 3 (mod 4) store their 16-byte slots with slot bit 1 flipped so that a group of 8 lanes -- four column groups x
    // two row blocks -- writes 8 different 4-bank groups (LSTR = 36 banks: column groups g and g + 2 would share one)
    const int mb = (task >> 2) & 7, rc = ((task >> 5) << 2) | (task & 3);
    const int wslot = mb ^ (((rc >> 1) & 1) << 1);
    const bf16_t* base = op == 0 ? (const bf16_t*)a.P : (const bf16_t*)sg.Q;
    const int ld = op == 0 ? a.ldp : sg.ldq;
    const int col = (op == 0 ? tc.rt : tc.st) * TILE + rc * 8;
    const bool cin = col < (op == 0 ? a.R : sg.S);
    const bool pro = op == 1 && sg.pro != GAST_PRO_NONE;
    const bool drop = op == 1 && sg.pro == GAST_PRO_BNRELU_DROP && a.drop.thresh != 0;
    const uint32_t key = drop ? drop_key(a.drop, sg.salt) : 0u;
    float sc[8], sh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { sc[q] = 0.f; sh[q] = 0.f; }
    if (pro && cin) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = sg.scale[col + q]; sh[q] = sg.shift[col + q]; }
    }
    unsigned char* sdst = op == 0 ? sP : sQ;

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto compute_rows = [&](int it, int buf) {
        if (tid < BKM) {
            int pr, qr;
            int m = m_begin + it * BKM + tid;
            rows_for(a, sg, m < m_end ? m : M, M, pr, qr);
            sRowP[buf][tid] = pr;
            sRowQ[buf][tid] = qr;
            const unsigned long long bad = __ballot(pr < 0);      // tid < 64 is exactly wave 0
            if (tid == 0) sBad[buf] = bad != 0ull;
        }
    };

    u32x4 rl[8];      // the staged 8(m) x 8(col) block: loaded, fixed up in place, transposed into LDS
    u32x4 rl2[8];     // second register set (RING == 2; dead otherwise)
    const int colc = cin ? col : 0;
    auto load_tile = [&](u32x4 (&rl)[8], int buf) {
        // the 8 source rows of this thread's block in two 16-byte LDS reads (one read + wait per row in front of every load
        // serialised eight LDS round trips ahead of the MFMAs)
        const int4* rp = (const int4*)((op == 0 ? sRowP[buf] : sRowQ[buf]) + mb * 8);
        const int4 r0 = rp[0], r1 = rp[1];
        const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) gload16(rl[i], base + (long)(rows[i] < 0 ? 0 : rows[i]) * ld + colc);
    };
    auto store_tile = [&](u32x4 (&rl)[8], int buf) {
        // Rows outside the chunk / the row map must read as zero (they are summed into valid outputs): only the last step of a
        // chunk or an out-of-range tap has any -- a block-uniform branch instead of 32 v_cndmask per step.  Columns beyond R / S
        // need no zeroing: they only reach dW rows / columns that are never written.
        if (__builtin_amdgcn_readfirstlane(sBad[buf])) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = op == 0 ? sRowP[buf][mb * 8 + i] : sRowQ[buf][mb * 8 + i];
                if (row < 0) rl[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (pro && cin) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = sRowQ[buf][mb * 8 + i];
                if (row < 0) continue;
                uint32_t wv[4] = {rl[i].x, rl[i].y, rl[i].z, rl[i].w};
                uint32_t e0 = (uint32_t)((long)row * ld + col);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float lo, hi;
                    h16x2_unpack(wv[p], lo, hi);
                    lo = fmaxf(fmaf(lo, sc[2 * p], sh[2 * p]), 0.f);
                    hi = fmaxf(fmaf(hi, sc[2 * p + 1], sh[2 * p + 1]), 0.f);
                    if (drop) {
                        lo *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + 2 * p);
                        hi *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e0 + 2 * p + 1);
                    }
                    wv[p] = pack_h16x2(lo, hi);
                }
                rl[i] = u32x4{wv[0], wv[1], wv[2], wv[3]};
            }
        }
        uint4 tr[8];
        transpose8x8_bf16(rl, tr);
#pragma unroll
        for (int q = 0; q < 8; ++q) *(uint4*)(sdst + (rc * 8 + q) * LSTR + wslot * 16) = tr[q];
    };

    const int rswz = ((li >> 4) & 1) << 1;      // (the slot swizzle of the staging writes: row = 32 * n + li, column group = row >> 3)
    auto mfma_tile = [&]() {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            union { uint4 u; s16x8 s; } fa[MI], fb[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                fa[mi].u = *(const uint4*)(sP + (wr * WROWS + mi * 32 + li) * LSTR + ((kc * 2 + lh) ^ rswz) * 16);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni].u = *(const uint4*)(sQ + (wc * 64 + ni * 32 + li) * LSTR + ((kc * 2 + lh) ^ rswz) * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = mfma_h16(fa[mi].u, fb[ni].u, acc[mi][ni]);
        }
    };

    if constexpr (RING == 1) {
        compute_rows(0, 0);
        __syncthreads();
        load_tile(rl, 0);
        if (ntile > 1) compute_rows(1, 1);
        for (int it = 0; it < ntile; ++it) {
            __syncthreads();
            gload_wait_n<0>();
#pragma unroll
            for (int i = 0; i < 8; ++i) gload_pin(rl[i]);
            store_tile(rl, it & 1);
            __syncthreads();
            if (it + 1 < ntile) load_tile(rl, (it + 1) & 1);
            if (it + 2 < ntile) compute_rows(it + 2, it & 1);
            mfma_tile();
        }
    } else {
        // step t: tile t sits in set t & 1; its reload (tile t + 2) is issued right after its LDS write; row maps three steps ahead
        auto step = [&](u32x4 (&set)[8], int it) {
            __syncthreads();                                   // every wave is done with the previous LDS tile
            if (it + 1 < ntile) gload_wait_n<8>(); else gload_wait_n<0>();     // the other set's 8 loads stay in flight
#pragma unroll
            for (int i = 0; i < 8; ++i) gload_pin(set[i]);
            store_tile(set, it & 3);
            __syncthreads();
            if (it + 2 < ntile) load_tile(set, (it + 2) & 3);
            if (it + 3 < ntile) compute_rows(it + 3, (it + 3) & 3);
            mfma_tile();
        };
        compute_rows(0, 0);
        if (ntile > 1) compute_rows(1, 1);
        if (ntile > 2) compute_rows(2, 2);
        __syncthreads();
        load_tile(rl, 0);
        if (ntile > 1) load_tile(rl2, 1);
        for (int it = 0; it < ntile; it += 2) {
            step(rl, it);
            if (it + 1 < ntile) step(rl2, it + 1);
        }
    }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int scol = tc.st * TILE + wc * 64 + ni * 32 + li;
        if (scol >= sg.S) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrow = tc.rt * TILE + wr * WROWS + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rrow < a.R) atomicAdd(a.dW + (long)rrow * a.ldw + sg.wcol0 + scol, acc[mi][ni][r]);
            }
    }
}

__global__ void __launch_bounds__(256, 3) wgrad_f32_kernel(const gast_wgrad_args a, int M, int tilesS_total, int splitM, int mchunk) {
    const int tile = blockIdx.x / splitM;
    wgrad_f32_body(a, M, tilesS_total, mchunk, tile, blockIdx.x - tile * splitM);
}
__global__ void __launch_bounds__(256, 3) wgrad_x3_kernel(const gast_wgrad_args a, int M, int tilesS_total, int splitM, int mchunk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BT * LSTR];
    const int tile = blockIdx.x / splitM;
    wgrad_x3_body<BT, true>(smem, a, M, tilesS_total, mchunk, tile, blockIdx.x - tile * splitM);
}
__global__ void __launch_bounds__(256, 3) wgrad_bf16_kernel(const gast_wgrad_args a, int M, int tilesS_total, int splitM, int mchunk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[wgrad_bf16_lds_bytes(128)];
    const int tile = blockIdx.x / splitM;
    wgrad_bf16_body<128>(smem, a, M, tilesS_total, mchunk, tile, blockIdx.x - tile * splitM);
}

__global__ void __launch_bounds__(256, 3) wgrad_f32_multi_kernel(const WgBatch b) {
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_f32_body(b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}
template <bool DROP>
__global__ void __launch_bounds__(256, 3) wgrad_x3_multi_kernel(const WgBatch b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BT * LSTR];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_x3_body<BT, DROP>(smem, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}
template <bool DROP, int NP>
__global__ void __launch_bounds__(256, 2) wgrad_x3_pipe_multi_kernel(const WgBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem_x3p[];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_x3_pipe_body<DROP, NP>(dsmem_x3p, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}
__global__ void __launch_bounds__(256, 3) wgrad_bf16_multi_kernel(const WgBatch b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[wgrad_bf16_lds_bytes(128)];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_bf16_body<128>(smem, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}
// two register sets in flight (GAST_WGRAD_RING=2): 200 VGPRs, two blocks per CU
__global__ void __launch_bounds__(256, 2) wgrad_bf16_multi_ring2_kernel(const WgBatch b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[wgrad_bf16_lds_bytes(128, 2)];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_bf16_body<128, 2>(smem, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}
// 256x256 tiles: 512 threads, 74.8 KB of dynamic LDS, one block per CU (two waves per SIMD: 256 registers each)
__global__ void __launch_bounds__(512, 2) wgrad_bf16_multi256_kernel(const WgBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_bf16_body<256>(dsmem, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// validation shared by gast_wgrad / gast_wgrad_multi; returns 0 and the tile counts, or an error code
int wgrad_check(const gast_wgrad_args& a, int& M, int& tilesR, int& tilesS, int bt = BT) {
    if (a.dtype != GAST_F32 && a.dtype != GAST_BF16 && a.dtype != GAST_F32X3) return GAST_EINVAL;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG || !a.P || !a.dW || a.R < 1 || a.B < 1 || a.Tn < 1 || a.J < 1) return GAST_EINVAL;
    const int epc = a.dtype == GAST_BF16 ? 8 : 4;
    if (a.R % epc || a.ldp % epc || !aligned16(a.P)) return GAST_EALIGN;
    tilesS = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_wgrad_seg& g = a.seg[s];
        if (!g.Q || g.S < 1) return GAST_EINVAL;
        if (g.S % epc || g.ldq % epc || !aligned16(g.Q)) return GAST_EALIGN;
        if (g.pro != GAST_PRO_NONE && (!g.scale || !g.shift || !aligned16(g.scale) || !aligned16(g.shift))) return GAST_EINVAL;
        tilesS += (g.S + bt - 1) / bt;
    }
    long Ml = (long)a.B * a.Tn * a.J;
    if (Ml > 0x7fffff00L) return GAST_ERANGE;
    if (a.dtype == GAST_F32X3) {      // the bf16x3 kernel addresses its operands with 32-bit byte offsets
        if ((long)a.B * a.pmap.T_total * a.J * a.ldp * 4 >= 0xffff0000L || a.R > 16000) return GAST_ERANGE;
        for (int s = 0; s < a.nseg; ++s)
            if ((long)a.B * a.seg[s].map.T_total * a.J * a.seg[s].ldq * 4 >= 0xffff0000L || a.seg[s].S > 16000) return GAST_ERANGE;
    }
    M = (int)Ml;
    tilesR = (a.R + bt - 1) / bt;
    return 0;
}

}  // namespace

extern "C" int gast_wgrad(const gast_wgrad_args* args, gast_stream_t stream) {
    if (!args) return GAST_EINVAL;
    const gast_wgrad_args& a = *args;
    int M, tilesR, tilesS;
    int rc = wgrad_check(a, M, tilesR, tilesS);
    if (rc) return rc;
    const int bkm = a.dtype == GAST_BF16 ? 64 : 32;
    const int tiles = tilesR * tilesS;
    static const int tgt = getenv("GAST_WGRAD_BLOCKS") ? atoi(getenv("GAST_WGRAD_BLOCKS")) : 768;   // 3 resident blocks per CU
    int splitM = gast_deterministic() ? 1 : tgt / tiles;      // (deterministic: one block per output tile, one add per element)
    int maxsplit = (M + bkm * 4 - 1) / (bkm * 4);
    if (splitM > maxsplit) splitM = maxsplit;
    if (splitM < 1) splitM = 1;
    int mchunk = (M + splitM - 1) / splitM;
    mchunk = (mchunk + bkm - 1) / bkm * bkm;
    splitM = (M + mchunk - 1) / mchunk;
    hipStream_t st = (hipStream_t)stream;
    if (a.zero_first) {
        hipError_t e = hipMemsetAsync(a.dW, 0, (size_t)a.R * a.ldw * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(tiles * splitM), block(256);
    if (a.dtype == GAST_F32)
        hipLaunchKernelGGL(wgrad_f32_kernel, grid, block, 0, st, a, M, tilesS, splitM, mchunk);
    else if (a.dtype == GAST_F32X3)
        hipLaunchKernelGGL(wgrad_x3_kernel, grid, block, 0, st, a, M, tilesS, splitM, mchunk);
    else
        hipLaunchKernelGGL(wgrad_bf16_kernel, grid, block, 0, st, a, M, tilesS, splitM, mchunk);
    GAST_CHECK_LAUNCH();
    return 0;
}

// (hipFuncSetAttribute is per device and nn.DataParallel replicas launch from several threads: set on every launch, like graph_ops.hip)
static hipError_t wgrad_dyn_lds(const void* fn, int bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

extern "C" int gast_wgrad_multi(const gast_wgrad_args* args, int n, gast_stream_t stream) {
    if (!args || n < 1 || n > GAST_WGRAD_MAX_BATCH) return GAST_EINVAL;
    WgBatch b;     // 2.4 KB: filled on the host, passed by value (kernel argument)
    int tilesR[GAST_WGRAD_MAX_BATCH];
    for (int d = 0; d < n; ++d)
        if (args[d].dtype != args[0].dtype) return GAST_EINVAL;
    // tile edge (bf16): 128, or 256 with GAST_WGRAD_TILE=256.  Measured on the B=128 step (scripts/wgrad_multi_bench.py): the
    // 256x256 variant halves the operand bytes per FLOP but runs one 8-wave block per CU with a single register set in flight,
    // and its wait -> transpose -> LDS write -> barrier -> issue -> MFMA -> barrier round is then fully exposed: 275 us vs 264 us
    // on the C=256 stage (19 vs 74 tiles), 111 vs 86 us on the C=512 stage, 124 vs 105 us on the C=128 stage.  Kept for the
    // next step (a second LDS stage / loader waves), not selected by default.
    static const int tile_env = getenv("GAST_WGRAD_TILE") ? atoi(getenv("GAST_WGRAD_TILE")) : 128;
    // bf16x3: 256 x 256 tiles (wgrad_wide.hip) when the job set fills them -- at least 3/4 of the tile area carries outputs (the
    // C = 256 and M = B*J stages: 0.93 - 0.95; the C = 128 stage's 128-wide matrices: 0.5, slower there) -- GAST_WGRAD_X3_TILE=128 / 256 forces
    static const int x3_tile_env = getenv("GAST_WGRAD_X3_TILE") ? atoi(getenv("GAST_WGRAD_X3_TILE")) : 0;
    // Round 5: 16-bit storage (GAST_BF16 jobs) takes the same kernel in its one-product form under the same rule (GAST_WGRAD_H16_WIDE=0:
    // the 128 x 128 kernel as in rounds 1-4)
    static const int h16_wide_env = getenv("GAST_WGRAD_H16_WIDE") ? atoi(getenv("GAST_WGRAD_H16_WIDE")) : 1;
    const bool h16_cand = args[0].dtype == GAST_BF16 && h16_wide_env && tile_env != 256;
    bool x3_wide = x3_tile_env == 256 && !h16_cand;
    if ((args[0].dtype == GAST_F32X3 && x3_tile_env == 0) || h16_cand) {
        double used = 0, area = 0;
        for (int d = 0; d < n; ++d) {
            long ts = 0, ssum = 0;
            for (int q = 0; q < args[d].nseg && q < GAST_MAX_SEG; ++q) { ts += (args[d].seg[q].S + 255) / 256; ssum += args[d].seg[q].S; }
            used += (double)args[d].R * ssum;
            area += (double)((args[d].R + 255) / 256) * ts * 65536.0;
        }
        x3_wide = area > 0 && used >= 0.75 * area;
    }
    bool h16_range = true;      // (the wide kernel addresses its operands with 32-bit byte offsets inside 4 GB buffer descriptors)
    for (int d = 0; d < n && h16_cand; ++d) {
        const gast_wgrad_args& a = args[d];
        if (a.nseg < 1 || a.nseg > GAST_MAX_SEG) { h16_range = false; break; }
        if ((long)a.B * a.pmap.T_total * a.J * a.ldp * 2 >= 0xffff0000L || a.R > 16000) h16_range = false;
        for (int q = 0; q < a.nseg; ++q)
            if ((long)a.B * a.seg[q].map.T_total * a.J * a.seg[q].ldq * 2 >= 0xffff0000L || a.seg[q].S > 16000) h16_range = false;
    }
    const bool h16_wide = h16_cand && x3_wide && h16_range;
    const int bt = ((args[0].dtype == GAST_BF16 && tile_env == 256) || (args[0].dtype == GAST_F32X3 && x3_wide) || h16_wide) ? 256 : BT;
    long tile_rows = 0;
    int total_tiles = 0;
    for (int d = 0; d < n; ++d) {
        int rc = wgrad_check(args[d], b.M[d], tilesR[d], b.tilesS[d], bt);
        if (rc) return rc;
        b.a[d] = args[d];
        total_tiles += tilesR[d] * b.tilesS[d];
        tile_rows += (long)tilesR[d] * b.tilesS[d] * b.M[d];
    }
    const int bkm = args[0].dtype == GAST_BF16 ? 64 : 32;
    static const int tgt128 = getenv("GAST_WGRAD_BLOCKS") ? atoi(getenv("GAST_WGRAD_BLOCKS")) : 1024;
    static const int tgt256 = getenv("GAST_WGRAD_BLOCKS256") ? atoi(getenv("GAST_WGRAD_BLOCKS256")) : 512;   // one resident block per CU
    static const int ring = getenv("GAST_WGRAD_RING") ? atoi(getenv("GAST_WGRAD_RING")) : 1;
    const bool wide = (args[0].dtype == GAST_F32X3 && bt == 256) || h16_wide;      // wgrad_wide.hip: one 8-wave block per CU
    static const int tgt_wide = getenv("GAST_WGRAD_BLOCKS_WIDE") ? atoi(getenv("GAST_WGRAD_BLOCKS_WIDE")) : 256;
    const int tgt = wide ? tgt_wide : bt == 256 ? tgt256 : tgt128;
    // one common chunk length (rows of the reduction axis per block) so that every block does the same number of steps
    long chunk = (tile_rows + tgt - 1) / tgt;
    chunk = (chunk + bkm - 1) / bkm * bkm;
    if (chunk < 4 * bkm) chunk = 4 * bkm;
    if (wide) {
        // one resident block per CU: the block count is quantised against the 256 slots (266 blocks take twice as long as 256), so
        // the chunk is the shortest multiple of the kernel's 96-row trip whose launch fits a whole number of rounds
        auto blocks_for = [&](long c) {
            long nb = 0;
            for (int d = 0; d < n; ++d) nb += (long)tilesR[d] * b.tilesS[d] * ((b.M[d] + c - 1) / c);
            return nb;
        };
        long maxM = 0;
        for (int d = 0; d < n; ++d) if (b.M[d] > maxM) maxM = b.M[d];
        const long slots = (long)tgt * ((total_tiles + tgt - 1) / tgt);
        const long trip = h16_wide ? 192 : 96;      // six steps of 16 (16-bit storage: 32) reduction rows
        chunk = trip;
        while (chunk < maxM && blocks_for(chunk) > slots) chunk += trip;
    }
    if (gast_deterministic()) chunk = 1L << 40;      // no split-M: every output tile is reduced by ONE block in row order
    const bool ring2 = ring == 2 && bt == 128 && args[0].dtype == GAST_BF16 && !getenv("GAST_WGRAD_BLOCKS") && !gast_deterministic();
    if (ring2) {
        // two blocks per CU: the block count is quantised against 512 slots (518 blocks run 1.33x slower than 444), so pick
        // the shortest chunk whose launch fits ONE round; if that leaves a quarter of the slots idle (the B*J-row stage: 300 or
        // 600 blocks), the shortest chunk that fits one round and a half.
        auto blocks_for = [&](long c) {
            long nb = 0;
            for (int d = 0; d < n; ++d) nb += (long)tilesR[d] * b.tilesS[d] * ((b.M[d] + c - 1) / c);
            return nb;
        };
        long maxM = 0;
        for (int d = 0; d < n; ++d) if (b.M[d] > maxM) maxM = b.M[d];
        const long cmax = (maxM + bkm - 1) / bkm * bkm;
        auto shortest = [&](long slots) {
            long c = 4 * bkm;
            while (c < cmax && blocks_for(c) > slots) c += bkm;
            return c;
        };
        chunk = shortest(512);
        if (blocks_for(chunk) < 384) {
            const long c2 = shortest(768);
            if (blocks_for(c2) > blocks_for(chunk)) chunk = c2;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    b.n = n;
    b.first[0] = 0;
    b.tfirst[0] = 0;
    int max_split = 1;
    // block order: 1 = chunk-major.  bf16 / fp32 kernels: tile-major (chunk-major measured 3 % slower on the bf16 step); the pipelined
    // bf16x3 kernel: chunk-major (PMC: 2.38 -> 1.07 GB fetched from HBM per launch on the C=256 stage, 535 -> 506 us)
    static const int order = getenv("GAST_WGRAD_ORDER") ? atoi(getenv("GAST_WGRAD_ORDER")) : -1;
    static const int x3_pipe = getenv("GAST_WGRAD_X3_PIPE") ? atoi(getenv("GAST_WGRAD_X3_PIPE")) : 1;
    b.chunk_major = order >= 0 ? order : ((args[0].dtype == GAST_F32X3 && ((bt == BT && x3_pipe) || wide)) || h16_wide) ? 1 : 0;
    for (int d = 0; d < n; ++d) {
        b.mchunk[d] = (int)(chunk < b.M[d] ? chunk : (b.M[d] + bkm - 1) / bkm * bkm);
        b.splitM[d] = (b.M[d] + b.mchunk[d] - 1) / b.mchunk[d];
        b.first[d + 1] = b.first[d] + tilesR[d] * b.tilesS[d] * b.splitM[d];
        b.tfirst[d + 1] = b.tfirst[d] + tilesR[d] * b.tilesS[d];
        if (b.splitM[d] > max_split) max_split = b.splitM[d];
        if (args[d].zero_first) {
            hipError_t e = hipMemsetAsync(args[d].dW, 0, (size_t)args[d].R * args[d].ldw * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    dim3 grid(b.chunk_major ? max_split * b.tfirst[n] : b.first[n]);
    if (args[0].dtype == GAST_F32)
        hipLaunchKernelGGL(wgrad_f32_multi_kernel, grid, dim3(256), 0, st, b);
    else if (args[0].dtype == GAST_F32X3) {
        bool any_drop = false;
        for (int d = 0; d < n; ++d)
            for (int q = 0; q < args[d].nseg; ++q) any_drop |= args[d].seg[q].pro == GAST_PRO_BNRELU_DROP && args[d].drop.thresh != 0;
        if (wide) {
            const int rcw = gast_wgrad_x3_wide_launch(b, grid.x, any_drop, st);
            if (rcw) return rcw;
        } else if (x3_pipe) {
            constexpr int lds = wgrad_x3_pipe_lds_bytes();
            static const int np_env = getenv("GAST_WGRAD_X3_PRODUCTS") ? atoi(getenv("GAST_WGRAD_X3_PRODUCTS")) : 3;
            typedef void (*kern_t)(const WgBatch);
            const kern_t kern = np_env == 1 ? (any_drop ? wgrad_x3_pipe_multi_kernel<true, 1> : wgrad_x3_pipe_multi_kernel<false, 1>)
                              : np_env == 2 ? (any_drop ? wgrad_x3_pipe_multi_kernel<true, 2> : wgrad_x3_pipe_multi_kernel<false, 2>)
                                            : (any_drop ? wgrad_x3_pipe_multi_kernel<true, 3> : wgrad_x3_pipe_multi_kernel<false, 3>);
            const hipError_t at = wgrad_dyn_lds((const void*)kern, lds);
            if (at != hipSuccess) return (int)at;
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
        }
        else if (any_drop) hipLaunchKernelGGL(wgrad_x3_multi_kernel<true>, grid, dim3(256), 0, st, b);
        else hipLaunchKernelGGL(wgrad_x3_multi_kernel<false>, grid, dim3(256), 0, st, b);
    }
    else if (h16_wide) {
        bool any_drop = false;
        for (int d = 0; d < n; ++d)
            for (int q = 0; q < args[d].nseg; ++q) any_drop |= args[d].seg[q].pro == GAST_PRO_BNRELU_DROP && args[d].drop.thresh != 0;
        const int rcw = gast_wgrad_x3_wide_launch(b, grid.x, any_drop, st, true);
        if (rcw) return rcw;
    }
    else if (bt == 256) {
        constexpr int lds = wgrad_bf16_lds_bytes(256);
        const hipError_t attr = wgrad_dyn_lds((const void*)wgrad_bf16_multi256_kernel, lds);
        if (attr != hipSuccess) return (int)attr;
        hipLaunchKernelGGL(wgrad_bf16_multi256_kernel, grid, dim3(512), lds, st, b);
    } else if (ring == 2)
        hipLaunchKernelGGL(wgrad_bf16_multi_ring2_kernel, grid, dim3(256), 0, st, b);
    else
        hipLaunchKernelGGL(wgrad_bf16_multi_kernel, grid, dim3(256), 0, st, b);
    GAST_CHECK_LAUNCH();
    return 0;
}
