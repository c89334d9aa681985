// gast_gemm: multi-segment "TN" GEMM on MFMA for gfx950 (MI355X).
//
//   C[cmap(m), n] = epi( sum_s sum_k pro_s(A_s[map_s(m), k]) * W_s[n, k] + bias[n] + addend[addmap(m), n] )
//
// One launch covers what the reference spreads over conv2d/conv1d/matmul + cat + permute + batch_norm + relu +
// dropout (reference gast_net.py:19,31-32,130,145-148,164,173-174; local_attention.py:37-38,122,142-148;
// global_attention.py:30-35,94,122-125):
//   * K segments read from different tensors  == torch.cat on the channel axis, never materialised;
//   * a row map per segment                    == the temporal taps of the dilated/strided (k,1) convolution;
//   * the A-load prologue                      == BatchNorm2d(apply) + ReLU (+ Dropout) of the producer;
//   * the STATS epilogue                       == the batch statistics of the BatchNorm2d that follows;
//   * the BNRELU_BWD epilogue                  == ReLU/Dropout backward + the two BN-backward column sums.
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA 32x32 tiles
// (v_mfma_f32_32x32x2_f32 for fp32: exact fmaf chain; v_mfma_f32_32x32x16_bf16 for bf16), K tile = 128 bytes per
// row.  Operands are staged global -> registers -> LDS (the prologue runs on the registers), LDS rows are padded to
// 144 B so the 16-byte fragment reads of a 16-lane group hit 16 distinct 4-bank slots (conflict-free).  The next K
// tile's global loads are issued before the MFMAs of the current one.  Blocks are remapped so that the N-tiles of
// one M-panel run on the same XCD (shared L2).
#include "common.h"
#include "gemm_big.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, LSTR = 144;

template <typename T> struct Mma;
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_h16(a, b, acc);      // (bfloat16 or binary16: the storage flavour of the build, common.h)
    }
};

// fp32 storage, split-bf16 arithmetic (GAST_F32X3): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (the difference is exact in
// fp32), and a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the dropped
// a_lo*b_lo term and the rounding of lo are ~2^-17 relative, i.e. fp32-class products at 3/16 of the fp32 MFMA cost.
// GAST_F32X3H (X3 = 2): the same with fp16 pairs on v_mfma_f32_32x32x16_f16 -- 11 + 11 significand bits instead of 8 + 8 (~2^-22
// per product) for operands inside fp16's range (activations, weights; NOT gradients): the forward GEMMs of the bf16x3 plan.
template <int PAIR>
__device__ __forceinline__ void split_x4(const uint4& v, uint2& hi, uint2& lo) {
    split_pair4<PAIR>(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), hi, lo);
}

// apply BN+ReLU(+dropout) to one 16-byte chunk of A held in registers
template <typename T>
__device__ __forceinline__ uint4 prologue(uint4 v, const float* sc, const float* sh, bool drop, uint32_t key,
                                          uint32_t thresh, float inv_keep, uint32_t e0);
template <>
__device__ __forceinline__ uint4 prologue<float>(uint4 v, const float* sc, const float* sh, bool drop, uint32_t key,
                                                 uint32_t thresh, float inv_keep, uint32_t e0) {
    float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float y = fmaxf(fmaf(x[q], sc[q], sh[q]), 0.f);
        if (drop) y *= drop_mul(key, thresh, inv_keep, e0 + q);
        x[q] = y;
    }
    return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
}
template <>
__device__ __forceinline__ uint4 prologue<bf16_t>(uint4 v, const float* sc, const float* sh, bool drop, uint32_t key,
                                                  uint32_t thresh, float inv_keep, uint32_t e0) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float lo, hi;
        h16x2_unpack(w[p], lo, hi);
        lo = fmaxf(fmaf(lo, sc[2 * p], sh[2 * p]), 0.f);
        hi = fmaxf(fmaf(hi, sc[2 * p + 1], sh[2 * p + 1]), 0.f);
        if (drop) {
            lo *= drop_mul(key, thresh, inv_keep, e0 + 2 * p);
            hi *= drop_mul(key, thresh, inv_keep, e0 + 2 * p + 1);
        }
        w[p] = pack_h16x2(lo, hi);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// 8 bf16 values (one 16-byte chunk) -> 8 OCP e4m3 bytes, each multiplied by `s` first (v_cvt_pk_fp8_f32: round to nearest even,
// saturating).  GAST_BF16 with gast_gemm_args.f8_scale: the "mixed fp8" mode of BASELINE.json configs[4] -- bf16 storage, the
// forward channel GEMMs' operands as fp8 on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulation; input / weight gradients stay bf16.
__device__ __forceinline__ uint2 to_fp8x8(const uint4& v, float s) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int q[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float lo, hi;
        h16x2_unpack(w[p], lo, hi);
        lo *= s;
        hi *= s;
        q[p >> 1] = (p & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, q[p >> 1], true) : __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, q[p >> 1], false);
    }
    return make_uint2((uint32_t)q[0], (uint32_t)q[1]);
}

template <typename T, typename TO, int X3 = 0, bool F8 = false>
__device__ __forceinline__ void gemm_body(const gast_gemm_args& a, int M, int gridM, int gridN, int vec_epi, int splitk, float* __restrict__ ws, int blk) {
    static_assert(!X3 || sizeof(T) == 4, "the split-bf16 mode stores fp32");
    static_assert(!F8 || sizeof(T) == 2, "the fp8-operand mode stores bf16");
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LSTR];   // A|B tiles, reused as the C staging tile
    __shared__ int sRow[GAST_MAX_SEG][BM];
    __shared__ int sCrow[BM];
    __shared__ int sAddRow[BM];
    __shared__ float sRed[4][BN][2];
    __shared__ int sBadSeg[GAST_MAX_SEG];      // does the segment map a row of this tile that must read as zero?
    unsigned char* const sA = smem;
    unsigned char* const sB = smem + BM * LSTR;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    // split-K (small-M GEMMs: few output tiles, long K): blockIdx.x = tile * splitk + split
    const int sp = blk % splitk;
    const int logical = xcd_remap(blk / splitk, gridM * gridN);
    const int mt = logical / gridN, nt = logical - mt * gridN;

    if (tid < GAST_MAX_SEG) sBadSeg[tid] = 0;
    __syncthreads();
    if (tid < BM) {
        int m = mt * BM + tid;
        int crow = -1, arow = -1;
        if (m < M) {
            int TJ = a.Tn * a.J;
            int b = m / TJ, rem = m - b * TJ;
            int t = rem / a.J, j = rem - t * a.J;
            for (int s = 0; s < a.nseg; ++s) {
                const int r = (int)map_row(a.seg[s].map, b, t, j, a.J);
                sRow[s][tid] = r;
                if (r < 0) sBadSeg[s] = 1;       // an out-of-range tap inside the tile (rows past M are never stored: no zeroing needed)
            }
            crow = (int)map_row(a.cmap, b, t, j, a.J);
            if (a.addend) arow = (int)map_row(a.addmap, b, t, j, a.J);
        } else {
            for (int s = 0; s < a.nseg; ++s) sRow[s][tid] = -1;
        }
        sCrow[tid] = crow;
        sAddRow[tid] = arow;
    }
    __syncthreads();

    const int chunk = tid & 7, rbase = tid >> 3;
    const uint32_t thresh = a.drop.thresh;
    const float inv_keep = a.drop.inv_keep;
    const uint32_t seedv = (thresh != 0 && a.drop.seed) ? *a.drop.seed : 0u;   // read once: no compiler-tracked VMEM load in the K loop
    const float f8_wscale = F8 ? a.f8_scale[0] : 1.f;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- main loop: 2-deep register-staged pipeline.
    // Two register sets (each: 4 A chunks + 4 W chunks + scale/shift chunks = NL 16-byte loads) hold the tiles t+1 and t+2
    // while tile t is multiplied out of LDS, so ~64 KB per block are in flight across a whole tile period (one tile in flight
    // left the kernel bound by exposed HBM/L2 latency: 3 us per K tile).  Loads are inline-asm (see gload16) and waited for
    // with counted s_waitcnt vmcnt(NL): the newer set stays in flight.  All per-segment fields live in scalar registers and
    // are re-fetched only when the segment changes; addresses are clamped instead of branching, zero rows / K tails are
    // applied when the tile is written to LDS.
    struct SegRegs {
        const T* A; const T* W; const float* scale; const float* shift;
        int lda, ldw, K, pro; uint32_t key; bool drop;
        bool fix;        // block-uniform: the tile needs zeroed rows (out-of-range taps) or a zeroed K tail
        int row[4];      // source rows of this thread's 4 tile rows (-1 = zero row)
    };
    auto fetch_seg = [&](int s, SegRegs& R) {
        const gast_gemm_seg& sg = a.seg[s];
        R.A = (const T*)sg.A; R.W = (const T*)sg.W;
        R.pro = sg.pro;
        R.scale = sg.pro != GAST_PRO_NONE ? sg.scale : (const float*)sg.W;   // dummy but valid address: the load count per set is constant
        R.shift = sg.pro != GAST_PRO_NONE ? sg.shift : (const float*)sg.W;
        R.lda = sg.lda; R.ldw = sg.ldw; R.K = sg.K;
        R.drop = sg.pro == GAST_PRO_BNRELU_DROP && thresh != 0;
        R.key = seedv * 0x9E3779B9u + sg.salt * 0x85EBCA6Bu;
        R.fix = __builtin_amdgcn_readfirstlane(sBadSeg[s]) != 0 || (sg.K % BK) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) R.row[i] = sRow[s][rbase + 32 * i];
    };
    int nrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int n = nt * BN + rbase + 32 * i; nrow[i] = n < a.N ? n : -1; }

    constexpr int NSS = EPC / 4;               // 16-byte chunks of scale (and of shift) per thread
    constexpr int NL = 8 + 2 * NSS;            // loads per register set
    struct RegSet { u32x4 a[4], b[4], sc[NSS], sh[NSS]; };
    RegSet R0;
    SegRegs L, S0;     // L: segment of the next tile to load; S0/S1: segments of the tiles held by R0/R1
    int l_s = 0, l_k0 = 0, k0_0 = 0;

    auto load_set = [&](RegSet& R, SegRegs& S, int& sk0) {
        S = L; sk0 = l_k0;
        const int k = l_k0 + chunk * EPC;
        const int kc_ = k < L.K ? k : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = L.row[i];
            gload16(R.a[i], L.A + (long)(row < 0 ? 0 : row) * L.lda + kc_);
            gload16(R.b[i], L.W + (long)(nrow[i] < 0 ? 0 : nrow[i]) * L.ldw + kc_);
        }
        const int ks = L.pro != GAST_PRO_NONE ? kc_ : 0;
#pragma unroll
        for (int q = 0; q < NSS; ++q) {
            gload16(R.sc[q], L.scale + ks + 4 * q);
            gload16(R.sh[q], L.shift + ks + 4 * q);
        }
        // advance L to the next tile
        l_k0 += BK;
        if (l_k0 >= L.K) { ++l_s; l_k0 = 0; if (l_s < a.nseg) fetch_seg(l_s, L); }
    };

    auto store_set = [&](const RegSet& R, const SegRegs& S, int sk0) {
        const int k = sk0 + chunk * EPC;
        const bool kin = k < S.K;
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int q = 0; q < NSS; ++q) {
            sc[4 * q] = __uint_as_float(R.sc[q].x); sc[4 * q + 1] = __uint_as_float(R.sc[q].y);
            sc[4 * q + 2] = __uint_as_float(R.sc[q].z); sc[4 * q + 3] = __uint_as_float(R.sc[q].w);
            sh[4 * q] = __uint_as_float(R.sh[q].x); sh[4 * q + 1] = __uint_as_float(R.sh[q].y);
            sh[4 * q + 2] = __uint_as_float(R.sh[q].z); sh[4 * q + 3] = __uint_as_float(R.sh[q].w);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rbase + 32 * i;
            const int row = S.row[i];
            uint4 v = make_uint4(R.a[i].x, R.a[i].y, R.a[i].z, R.a[i].w);
            if (S.pro != GAST_PRO_NONE)
                v = prologue<T>(v, sc, sh, S.drop, S.key, thresh, inv_keep, (uint32_t)((long)(row < 0 ? 0 : row) * S.lda + k));
            uint4 wv = make_uint4(R.b[i].x, R.b[i].y, R.b[i].z, R.b[i].w);
            // Zero rows (out-of-range taps) and the K tail must read as zero (relu(shift) must not leak in); a block-uniform
            // branch, taken by few tiles, instead of 32 v_cndmask per K tile.  Rows past M / W rows past N are clamped to row 0 and
            // only reach outputs that are never stored.
            if (S.fix) {
                const bool oka = kin && row >= 0;
                v = make_uint4(oka ? v.x : 0u, oka ? v.y : 0u, oka ? v.z : 0u, oka ? v.w : 0u);
                wv = make_uint4(kin ? wv.x : 0u, kin ? wv.y : 0u, kin ? wv.z : 0u, kin ? wv.w : 0u);
            }
            if (F8) {
                // row image: 64 e4m3 bytes; this thread's 8 k values are bytes chunk*8 .. +8
                *(uint2*)(sA + r * LSTR + chunk * 8) = to_fp8x8(v, 1.f);
                *(uint2*)(sB + r * LSTR + chunk * 8) = to_fp8x8(wv, f8_wscale);
            } else if (X3) {
                // row image: [hi plane: 32 bf16 = 64 B | lo plane: 64 B]; this thread's 4 k values are bytes chunk*8 .. +8 of each
                uint2 h, l;
                split_x4<X3 == 2 ? 2 : 1>(v, h, l);
                *(uint2*)(sA + r * LSTR + chunk * 8) = h;
                *(uint2*)(sA + r * LSTR + 64 + chunk * 8) = l;
                split_x4<X3 == 2 ? 2 : 1>(wv, h, l);
                *(uint2*)(sB + r * LSTR + chunk * 8) = h;
                *(uint2*)(sB + r * LSTR + 64 + chunk * 8) = l;
            } else {
                *(uint4*)(sA + r * LSTR + chunk * 16) = v;
                *(uint4*)(sB + r * LSTR + chunk * 16) = wv;
            }
        }
    };

    auto compute_tile = [&]() {
        if (F8) {
            // K tile = 64 e4m3 values: four 16-deep steps, 8 bytes per lane and operand
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                long fa[2], fb[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const long*)(sA + (wr * 64 + mi * 32 + li) * LSTR + (kc * 2 + lh) * 8);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) fb[ni] = *(const long*)(sB + (wc * 64 + ni * 32 + li) * LSTR + (kc * 2 + lh) * 8);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
            }
            return;
        }
        if (X3) {
            // K tile = 32: two 16-deep MFMA steps; per step the hi and lo fragments of both operands, three products per
            // accumulator, small terms first, consecutive MFMAs on different accumulators
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                constexpr int PAIR = X3 == 2 ? 2 : 1;
                uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const unsigned char* p = sA + (wr * 64 + mi * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                    ah[mi] = *(const uint4*)p;
                    al[mi] = *(const uint4*)(p + 64);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const unsigned char* p = sB + (wc * 64 + ni * 32 + li) * LSTR + (ks * 2 + lh) * 16;
                    bh[ni] = *(const uint4*)p;
                    bl[ni] = *(const uint4*)(p + 64);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_pair<PAIR>(al[mi], bh[ni], acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_pair<PAIR>(ah[mi], bl[ni], acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_pair<PAIR>(ah[mi], bh[ni], acc[mi][ni]);
            }
            return;
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                fa[mi] = *(const uint4*)(sA + (wr * 64 + mi * 32 + li) * LSTR + (kc * 2 + lh) * 16);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni] = *(const uint4*)(sB + (wc * 64 + ni * 32 + li) * LSTR + (kc * 2 + lh) * 16);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) Mma<T>::run(acc[mi][ni], fa[mi], fb[ni]);
        }
    };

    // single register set, three blocks per CU: measured equal to the 2-deep ring per block, and 3 resident blocks overlap the
    // load / stage / MFMA / epilogue phases of different tiles better than 2
    // this block's K-tile range [t_begin, t_end) of the flattened (segment, k0) tile list
    int ntiles_all = 0;
    for (int sgi = 0; sgi < a.nseg; ++sgi) ntiles_all += (a.seg[sgi].K + BK - 1) / BK;
    const int tps = (ntiles_all + splitk - 1) / splitk;
    const int t_begin = sp * tps;
    int t_left = min(ntiles_all, t_begin + tps) - t_begin;
    if (t_left > 0) {
        int skip = t_begin;
        while (true) {     // advance to the segment that contains tile t_begin
            const int nts = (a.seg[l_s].K + BK - 1) / BK;
            if (skip < nts) break;
            skip -= nts;
            ++l_s;
        }
        l_k0 = skip * BK;
        fetch_seg(l_s, L);
        load_set(R0, S0, k0_0);
        while (true) {
            gload_wait_n<0>();
#pragma unroll
            for (int i = 0; i < 4; ++i) { gload_pin(R0.a[i]); gload_pin(R0.b[i]); }
#pragma unroll
            for (int q = 0; q < NSS; ++q) { gload_pin(R0.sc[q]); gload_pin(R0.sh[q]); }
            __syncthreads();                // everyone finished reading the previous tile
            store_set(R0, S0, k0_0);
            __syncthreads();
            --t_left;
            const bool more = t_left > 0;
            if (more) load_set(R0, S0, k0_0);
            compute_tile();
            if (!more) break;
        }
    }
    if (F8) {
        const float ds = a.f8_scale[1];        // 1 / weight scale (a power of two)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= ds;
    }
    if (splitk > 1) {
        // raw fp32 partial tile -> workspace [split][M][N]; bias / addend / epilogue run in splitk_finish_kernel
        float* wsp = ws + (long)sp * M * a.N;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = nt * BN + wc * 64 + ni * 32 + li;
            if (n >= a.N) continue;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mt * BM + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < M) wsp[(long)m * a.N + n] = acc[mi][ni][r];
                }
        }
        return;
    }

    // ------------------------------------------------------------------ epilogue
    const int epi = a.epi;
    const bool xdrop = epi == GAST_EPI_BNRELU_BWD && a.xdrop && thresh != 0;
    uint32_t xkey = 0;
    if (xdrop) xkey = drop_key(a.drop, a.xsalt);
    if (vec_epi) {
        // ---- coalesced epilogue: acc (+bias) -> LDS staging tile (TO) -> 16-byte row chunks -> math -> 16-byte stores.
        // The accumulator layout (lane = column, 16 rows per register file) would give 2-byte scattered stores and
        // scattered loads of X / addend; in the staged layout a wave touches 4 full 256-byte rows per instruction.
        constexpr int EPO = 16 / (int)sizeof(TO);        // elements per 16-byte chunk of the output
        constexpr int CPR = BN / EPO;                    // chunks per tile row
        constexpr int CSTR = BN * (int)sizeof(TO) + 16;  // staging row stride (bytes), 16-byte pad against bank conflicts
        constexpr int NH = sizeof(TO) == 4 ? 2 : 1;      // fp32 tiles are staged in two 64-row halves
        constexpr int RPP = 256 / CPR;                   // rows covered per pass of the 256 threads
        unsigned char* const sC = smem;
        const int cc = tid % CPR, rq = tid / CPR;
        const int n0 = nt * BN + cc * EPO;
        const bool nin = n0 < a.N;
        float xs[EPO], xh[EPO], st1[EPO], st2[EPO];
#pragma unroll
        for (int q = 0; q < EPO; ++q) { xs[q] = 0.f; xh[q] = 0.f; st1[q] = 0.f; st2[q] = 0.f; }
        if (epi == GAST_EPI_BNRELU_BWD && nin) {
#pragma unroll
            for (int q = 0; q < EPO; ++q) { xs[q] = a.xscale[n0 + q]; xh[q] = a.xshift[n0 + q]; }
        }
        const TO* Addv = (const TO*)a.addend;
        const TO* Xv = (const TO*)a.X;
        TO* Cv = (TO*)a.C;
        __syncthreads();   // every wave is done reading the A/B tiles
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (NH == 1 || wr == h) {
                const int rbase_w = NH == 1 ? wr * 64 : 0;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int nl = wc * 64 + ni * 32 + li;
                    const int n = nt * BN + nl;
                    const float bias = (a.bias && n < a.N) ? (a.bias_neg ? -a.bias[n] : a.bias[n]) : 0.f;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = rbase_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            Elem<TO>::st((TO*)(sC + rl * CSTR) + nl, acc[mi][ni][r] + bias);
                        }
                }
            }
            __syncthreads();
            constexpr int ROWS = NH == 1 ? BM : BM / 2;
#pragma unroll 2
            for (int i = 0; i < ROWS / RPP; ++i) {
                const int rl = rq + i * RPP;
                const int ml = (NH == 1 ? 0 : h * 64) + rl;
                const int crow = sCrow[ml];
                if (!nin || crow < 0) continue;
                const uint4 raw = *(const uint4*)(sC + rl * CSTR + cc * 16);
                float v[EPO];
                if (sizeof(TO) == 4) {
                    v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y); v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
                } else {
                    const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) h16x2_unpack(w4[p2], v[(2 * p2) % EPO], v[(2 * p2 + 1) % EPO]);
                }
                if (Addv) {
                    const int arow = sAddRow[ml];
                    if (arow >= 0) {
                        const uint4 ar = *(const uint4*)(Addv + (long)arow * a.ldadd + n0);
                        if (sizeof(TO) == 4) {
                            v[0] += __uint_as_float(ar.x); v[1] += __uint_as_float(ar.y); v[2] += __uint_as_float(ar.z); v[3] += __uint_as_float(ar.w);
                        } else {
                            const uint32_t w4[4] = {ar.x, ar.y, ar.z, ar.w};
#pragma unroll
                            for (int p2 = 0; p2 < 4; ++p2) {
                                float a0, a1;
                                h16x2_unpack(w4[p2], a0, a1);
                                v[(2 * p2) % EPO] += a0;
                                v[(2 * p2 + 1) % EPO] += a1;
                            }
                        }
                    }
                }
                if (epi == GAST_EPI_BNRELU_BWD) {
                    if (a.C2) {          // second output: the value before the mask
                        uint4 o2;
                        if (sizeof(TO) == 4) o2 = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                        else o2 = make_uint4(pack_h16x2(v[0], v[1 % EPO]), pack_h16x2(v[2 % EPO], v[3 % EPO]), pack_h16x2(v[4 % EPO], v[5 % EPO]),
                                             pack_h16x2(v[6 % EPO], v[7 % EPO]));
                        *(uint4*)((TO*)a.C2 + (long)crow * a.ldc2 + n0) = o2;
                    }
                    const uint4 xr = *(const uint4*)(Xv + (long)crow * a.ldx + n0);
                    float x[EPO];
                    if (sizeof(TO) == 4) {
                        x[0] = __uint_as_float(xr.x); x[1] = __uint_as_float(xr.y); x[2] = __uint_as_float(xr.z); x[3] = __uint_as_float(xr.w);
                    } else {
                        const uint32_t w4[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
                        for (int p2 = 0; p2 < 4; ++p2) h16x2_unpack(w4[p2], x[(2 * p2) % EPO], x[(2 * p2 + 1) % EPO]);
                    }
                    const uint32_t e0 = (uint32_t)((long)crow * a.ldx + n0);
#pragma unroll
                    for (int q = 0; q < EPO; ++q) {
                        float y = v[q];
                        if (!(fmaf(x[q], xs[q], xh[q]) > 0.f)) y = 0.f;
                        if (xdrop) y *= drop_mul(xkey, thresh, inv_keep, e0 + q);
                        y = Elem<TO>::rnd(y);
                        v[q] = y;
                        st1[q] += y;
                        st2[q] = fmaf(y, x[q], st2[q]);
                    }
                } else if (epi == GAST_EPI_STATS) {
#pragma unroll
                    for (int q = 0; q < EPO; ++q) {
                        const float y = Elem<TO>::rnd(v[q]);
                        v[q] = y;
                        st1[q] += y;
                        st2[q] = fmaf(y, y, st2[q]);
                    }
                }
                uint4 o;
                if (sizeof(TO) == 4) {
                    o = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                } else {
                    o = make_uint4(pack_h16x2(v[0], v[1 % EPO]), pack_h16x2(v[2 % EPO], v[3 % EPO]), pack_h16x2(v[4 % EPO], v[5 % EPO]),
                                   pack_h16x2(v[6 % EPO], v[7 % EPO]));
                }
                *(uint4*)(Cv + (long)crow * a.ldc + n0) = o;
            }
            if (h + 1 < NH) __syncthreads();
        }
        if (epi != GAST_EPI_PLAIN) {
            // threads with the same column chunk: lanes l, l+CPR, ... of every wave
#pragma unroll
            for (int q = 0; q < EPO; ++q) {
                if (CPR <= 16) { st1[q] += __shfl_xor(st1[q], 16); st2[q] += __shfl_xor(st2[q], 16); }
                st1[q] += __shfl_xor(st1[q], 32);
                st2[q] += __shfl_xor(st2[q], 32);
            }
            if (lane < CPR) {
#pragma unroll
                for (int q = 0; q < EPO; ++q) { sRed[w][lane * EPO + q][0] = st1[q]; sRed[w][lane * EPO + q][1] = st2[q]; }
            }
            __syncthreads();
            if (tid < BN) {
                const int n = nt * BN + tid;
                if (n < a.N) {
                    float* pp = a.partials + ((long)mt * a.N + n) * 2;
                    pp[0] = sRed[0][tid][0] + sRed[1][tid][0] + sRed[2][tid][0] + sRed[3][tid][0];
                    pp[1] = sRed[0][tid][1] + sRed[1][tid][1] + sRed[2][tid][1] + sRed[3][tid][1];
                }
            }
        }
        return;
    }
    // ---- scalar epilogue (fallback for unaligned / narrow outputs such as the N=3 shrink layer)
    TO* Cb = (TO*)a.C;
    const T* Addb = (const T*)a.addend;
    const T* Xb = (const T*)a.X;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int nl = wc * 64 + ni * 32 + li;
        const int n = nt * BN + nl;
        const bool nin = n < a.N;
        float bias = (a.bias && nin) ? (a.bias_neg ? -a.bias[n] : a.bias[n]) : 0.f;
        float xs = 0.f, xh = 0.f;
        if (epi == GAST_EPI_BNRELU_BWD && nin) { xs = a.xscale[n]; xh = a.xshift[n]; }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int crow = sCrow[ml];
                if (!nin || crow < 0) continue;
                float v = acc[mi][ni][r] + bias;
                if (Addb) {
                    int arow = sAddRow[ml];
                    if (arow >= 0) v += Elem<T>::ld(Addb + (long)arow * a.ldadd + n);
                }
                if (epi == GAST_EPI_BNRELU_BWD) {
                    if (a.C2) Elem<TO>::st((TO*)a.C2 + (long)crow * a.ldc2 + n, v);
                    float x = Elem<T>::ld(Xb + (long)crow * a.ldx + n);
                    if (!(fmaf(x, xs, xh) > 0.f)) v = 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)((long)crow * a.ldx + n));
                    v = Elem<TO>::rnd(v);
                    s1[ni] += v;
                    s2[ni] += v * x;
                } else if (epi == GAST_EPI_STATS) {
                    v = Elem<TO>::rnd(v);
                    s1[ni] += v;
                    s2[ni] += v * v;
                }
                Elem<TO>::st(Cb + (long)crow * a.ldc + n, v);
            }
        }
    }
    if (epi != GAST_EPI_PLAIN) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32);
            s2[ni] += __shfl_xor(s2[ni], 32);
            if (lh == 0) {
                sRed[wr][wc * 64 + ni * 32 + li][0] = s1[ni];
                sRed[wr][wc * 64 + ni * 32 + li][1] = s2[ni];
            }
        }
        __syncthreads();
        if (tid < BN) {
            int n = nt * BN + tid;
            if (n < a.N) {
                float* p = a.partials + ((long)mt * a.N + n) * 2;
                p[0] = sRed[0][tid][0] + sRed[1][tid][0];
                p[1] = sRed[0][tid][1] + sRed[1][tid][1];
            }
        }
    }
}

// ---- split-K finish: C = epi(sum_split ws + bias + addend), same epilogue semantics and partial-sum layout as gemm_kernel.
// grid = (N tiles of 128 columns) x (groups of 8 rows); thread = (row, 4 consecutive columns); the column statistics of the 8 rows
// are combined in LDS and added atomically to the row tile's (pre-zeroed) partial row.
template <typename T, typename TO, int X3 = 0, bool F8 = false>
__global__ void __launch_bounds__(256, 3) gemm_kernel(const gast_gemm_args a, int M, int gridM, int gridN, int vec_epi, int splitk, float* __restrict__ ws) {
    gemm_body<T, TO, X3, F8>(a, M, gridM, gridN, vec_epi, splitk, ws, blockIdx.x);
}

// Several independent GEMMs of one plan step in ONE grid (gast_gemm_multi): the K <= 256 launches of a block (G2 / G3, the two
// branch input gradients, ...) have 425-646 blocks each -- 1.7-2.5 per CU at an occupancy of 3 -- and end in a tail; launched
// together they share one tail and one launch.
// Round 3: split-K jobs (the M = B*J stage) ride in the same grid -- the three disjoint-tap input gradients of the last temporal
// level, G2 | G3 and the two branch input gradients of the last block were one launch PAIR each (408 blocks + finish); together they
// are one grid of ~1200 blocks and ONE finish launch (splitk_finish_multi_kernel).  ws_off = the job's slice of the workspace.
struct GemmBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int M[GAST_GEMM_MAX_BATCH], gridM[GAST_GEMM_MAX_BATCH], gridN[GAST_GEMM_MAX_BATCH], vec_epi[GAST_GEMM_MAX_BATCH];
    int splitk[GAST_GEMM_MAX_BATCH];
    long ws_off[GAST_GEMM_MAX_BATCH];
    float* ws;
    int n;
};
static_assert(sizeof(GemmBatch) <= 3712, "GemmBatch travels as a kernel argument (4 KB limit)");
template <typename T, typename TO, int X3 = 0, bool F8 = false>
__global__ void __launch_bounds__(256, 3) gemm_multi_kernel(const GemmBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    gemm_body<T, TO, X3, F8>(b.a[d], b.M[d], b.gridM[d], b.gridN[d], b.vec_epi[d], b.splitk[d], b.splitk[d] > 1 ? b.ws + b.ws_off[d] : nullptr,
                             blockIdx.x - b.first[d]);
}

template <typename T, typename TO>
__device__ __forceinline__ void splitk_finish_body(const gast_gemm_args& a, int M, int gridN, int splitk, const float* __restrict__ ws, int blk) {
    __shared__ float sRed[8][BN][2];
    const int tid = threadIdx.x;
    const int rg = blk / gridN, nt = blk - rg * gridN;
    const int cc = tid & 31, rq = tid >> 5;
    const int n0 = nt * BN + cc * 4;
    const int m = rg * 8 + rq;
    const int epi = a.epi;
    const uint32_t thresh = a.drop.thresh;
    const float inv_keep = a.drop.inv_keep;
    const bool xdrop = epi == GAST_EPI_BNRELU_BWD && a.xdrop && thresh != 0;
    const uint32_t xkey = xdrop ? drop_key(a.drop, a.xsalt) : 0u;
    float st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0};
    if (m < M && n0 < a.N) {
        const int TJ = a.Tn * a.J;
        const int b = m / TJ, rem = m - b * TJ;
        const int t = rem / a.J, j = rem - t * a.J;
        const long crow = map_row(a.cmap, b, t, j, a.J);
        const long arow = a.addend ? map_row(a.addmap, b, t, j, a.J) : -1;
        if (crow >= 0) {
            float v4[4] = {0, 0, 0, 0};
            const bool full = n0 + 3 < a.N && (a.N & 3) == 0;
            for (int s = 0; s < splitk; ++s) {
                const float* p = ws + ((long)s * M + m) * a.N + n0;
                if (full) { const float4 q = *(const float4*)p; v4[0] += q.x; v4[1] += q.y; v4[2] += q.z; v4[3] += q.w; }
                else { for (int q = 0; q < 4; ++q) if (n0 + q < a.N) v4[q] += p[q]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + q;
                if (n >= a.N) continue;
                float v = v4[q];
                if (a.bias) v += a.bias_neg ? -a.bias[n] : a.bias[n];
                if (arow >= 0) v += Elem<T>::ld((const T*)a.addend + arow * a.ldadd + n);
                if (epi == GAST_EPI_BNRELU_BWD) {
                    if (a.C2) Elem<TO>::st((TO*)a.C2 + crow * a.ldc2 + n, v);
                    const float x = Elem<T>::ld((const T*)a.X + crow * a.ldx + n);
                    if (!(fmaf(x, a.xscale[n], a.xshift[n]) > 0.f)) v = 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)(crow * a.ldx + n));
                    v = Elem<TO>::rnd(v);
                    st1[q] = v;
                    st2[q] = v * x;
                } else if (epi == GAST_EPI_STATS) {
                    v = Elem<TO>::rnd(v);
                    st1[q] = v;
                    st2[q] = v * v;
                }
                Elem<TO>::st((TO*)a.C + crow * a.ldc + n, v);
            }
        }
    }
    if (epi != GAST_EPI_PLAIN) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sRed[rq][cc * 4 + q][0] = st1[q]; sRed[rq][cc * 4 + q][1] = st2[q]; }
        __syncthreads();
        if (tid < BN) {
            const int n = nt * BN + tid;
            if (n < a.N) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { t1 += sRed[r][tid][0]; t2 += sRed[r][tid][1]; }
                float* pp = a.partials + ((long)((rg * 8) / BM) * a.N + n) * 2;
                atomicAdd(pp, t1);
                atomicAdd(pp + 1, t2);
            }
        }
    }
}
template <typename T, typename TO>
__global__ void __launch_bounds__(256) splitk_finish_kernel(const gast_gemm_args a, int M, int gridN, int splitk, const float* __restrict__ ws) {
    splitk_finish_body<T, TO>(a, M, gridN, splitk, ws, blockIdx.x);
}
// the finishes of the split-K jobs of one gast_gemm_multi call in one grid
struct FinishBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int M[GAST_GEMM_MAX_BATCH], gridN[GAST_GEMM_MAX_BATCH], splitk[GAST_GEMM_MAX_BATCH];
    long ws_off[GAST_GEMM_MAX_BATCH];
    const float* ws;
    int n;
};
static_assert(sizeof(FinishBatch) <= 3712, "FinishBatch travels as a kernel argument (4 KB limit)");
template <typename T, typename TO>
__global__ void __launch_bounds__(256) splitk_finish_multi_kernel(const FinishBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    splitk_finish_body<T, TO>(b.a[d], b.M[d], b.gridN[d], b.splitk[d], b.ws + b.ws_off[d], blockIdx.x - b.first[d]);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int gast_gemm_row_blocks(int M) { return (M + BM - 1) / BM; }

// fp32 workspace the caller may provide to let small-M GEMMs split their K loop over several blocks
extern "C" long gast_gemm_splitk_ws_bytes(long M, int N) { return 8L * M * N * (long)sizeof(float); }

extern "C" int gast_gemm_ws(const gast_gemm_args* args, void* ws, long ws_bytes, gast_stream_t stream);
extern "C" int gast_gemm(const gast_gemm_args* args, gast_stream_t stream) { return gast_gemm_ws(args, nullptr, 0, stream); }

namespace {
// validation + launch geometry shared by gast_gemm_ws / gast_gemm_multi
int gemm_plan(const gast_gemm_args& a, void* ws, long ws_bytes, int& M, int& gridM, int& gridN, int& vec_epi, int& splitk) {
    if (a.dtype != GAST_F32 && a.dtype != GAST_BF16 && a.dtype != GAST_F32X3 && a.dtype != GAST_F32X3H) return GAST_EINVAL;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG || !a.C || a.N < 1 || a.B < 1 || a.Tn < 1 || a.J < 1) return GAST_EINVAL;
    const int epc = a.dtype == GAST_BF16 ? 8 : 4;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.A || !g.W || g.K < 1) return GAST_EINVAL;
        if (g.K % epc || g.lda % epc || g.ldw % epc || !aligned16(g.A) || !aligned16(g.W)) return GAST_EALIGN;
        if (g.pro != GAST_PRO_NONE && (!g.scale || !g.shift || !aligned16(g.scale) || !aligned16(g.shift))) return GAST_EINVAL;
        if (g.pro < 0 || g.pro > GAST_PRO_BNRELU_DROP) return GAST_EINVAL;
    }
    if (a.epi < 0 || a.epi > GAST_EPI_BNRELU_BWD) return GAST_EINVAL;
    if (a.f8_scale && (a.dtype != GAST_BF16 || a.out_f32)) return GAST_EINVAL;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return GAST_EINVAL;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return GAST_EINVAL;
    if (a.C2 && a.epi != GAST_EPI_BNRELU_BWD) return GAST_EINVAL;
    long Ml = (long)a.B * a.Tn * a.J;
    if (Ml > 0x7fffff00L) return GAST_ERANGE;
    M = (int)Ml;
    gridM = (M + BM - 1) / BM;
    gridN = (a.N + BN - 1) / BN;
    // split-K: few output tiles and a long K loop (the M = B*J rows of the last stage): up to 8 K ranges per tile
    int ntiles = 0;
    for (int s2 = 0; s2 < a.nseg; ++s2) ntiles += (a.seg[s2].K + 8 * epc - 1) / (8 * epc);
    splitk = 1;
    static const int allow_splitk = getenv("GAST_GEMM_SPLITK") ? atoi(getenv("GAST_GEMM_SPLITK")) : 1;   // 0: bisecting aid
    static const int splitk_blocks = getenv("GAST_GEMM_SPLITK_BLOCKS") ? atoi(getenv("GAST_GEMM_SPLITK_BLOCKS")) : 512;
    if (ws && gridM * gridN <= 160 && ntiles >= 4 && allow_splitk && !gast_deterministic()) {
        splitk = splitk_blocks / (gridM * gridN);
        if (splitk > 8) splitk = 8;
        if (splitk > ntiles / 2) splitk = ntiles / 2;
        if (splitk < 1) splitk = 1;
        if ((long)splitk * M * a.N * (long)sizeof(float) > ws_bytes) splitk = 1;
        if (splitk > 1) {
            const int tps = (ntiles + splitk - 1) / splitk;
            splitk = (ntiles + tps - 1) / tps;      // no empty K ranges
        }
    }
    // the coalesced (LDS-staged, 16-byte) epilogue needs same-width in/out element types and 16-byte aligned rows
    vec_epi = !(a.dtype == GAST_BF16 && a.out_f32) && a.N % epc == 0 && a.ldc % epc == 0 && aligned16(a.C);
    if (a.addend && (a.ldadd % epc || !aligned16(a.addend))) vec_epi = 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (a.ldx % epc || !aligned16(a.X))) vec_epi = 0;
    if (a.C2 && (a.ldc2 % epc || !aligned16(a.C2))) vec_epi = 0;
    return 0;
}
}  // namespace

extern "C" int gast_gemm_ws(const gast_gemm_args* args, void* ws, long ws_bytes, gast_stream_t stream) {
    if (!args) return GAST_EINVAL;
    const gast_gemm_args& a = *args;
    int M, gridM, gridN, vec_epi, splitk;
    int rc = gemm_plan(a, ws, ws_bytes, M, gridM, gridN, vec_epi, splitk);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    BigPlan bp;
    if (gast_gemm_big_plan(a, bp)) return gast_gemm_big_launch(a, bp, st);
    BjPlan jp;      // the M = B*J stage (gemm_bj.hip): needs zero-filled `partials`, which only the workspace entry points promise
    if (ws && gast_gemm_bj_plan(a, jp)) return gast_gemm_bj_launch_multi(&a, &jp, 1, st);      // large-M GAST_F32X3 GEMMs: gemm_big.hip
    dim3 grid(gridM * gridN * splitk), block(256);
    if (a.dtype == GAST_F32)
        hipLaunchKernelGGL((gemm_kernel<float, float>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    else if (a.dtype == GAST_F32X3)
        hipLaunchKernelGGL((gemm_kernel<float, float, 1>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    else if (a.dtype == GAST_F32X3H)
        hipLaunchKernelGGL((gemm_kernel<float, float, 2>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    else if (a.f8_scale && !a.out_f32)
        hipLaunchKernelGGL((gemm_kernel<bf16_t, bf16_t, 0, true>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    else if (a.out_f32)
        hipLaunchKernelGGL((gemm_kernel<bf16_t, float>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    else
        hipLaunchKernelGGL((gemm_kernel<bf16_t, bf16_t>), grid, block, 0, st, a, M, gridM, gridN, vec_epi, splitk, (float*)ws);
    GAST_CHECK_LAUNCH();
    if (splitk > 1) {
        // the finish kernel accumulates the column statistics with atomics: `partials` arrives zero-filled (gast_hip.h)
        dim3 fgrid(gridN * ((M + 7) / 8));
        if (a.dtype != GAST_BF16)
            hipLaunchKernelGGL((splitk_finish_kernel<float, float>), fgrid, block, 0, st, a, M, gridN, splitk, (const float*)ws);
        else if (a.out_f32)
            hipLaunchKernelGGL((splitk_finish_kernel<bf16_t, float>), fgrid, block, 0, st, a, M, gridN, splitk, (const float*)ws);
        else
            hipLaunchKernelGGL((splitk_finish_kernel<bf16_t, bf16_t>), fgrid, block, 0, st, a, M, gridN, splitk, (const float*)ws);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int gast_gemm_path(const gast_gemm_args* args) {
    if (!args) return GAST_EINVAL;
    BigPlan bp;
    if (gast_gemm_big_plan(*args, bp)) return 1;
    BjPlan jp;
    return gast_gemm_bj_plan(*args, jp) ? 2 : 0;
}

extern "C" int gast_gemm_multi(const gast_gemm_args* args, int n, void* ws, long ws_bytes, gast_stream_t stream) {
    if (!args || n < 1 || n > GAST_GEMM_MAX_BATCH) return GAST_EINVAL;
    GemmBatch b;
    b.n = 0;
    b.first[0] = 0;
    b.ws = (float*)ws;
    FinishBatch fb;
    fb.n = 0;
    fb.first[0] = 0;
    fb.ws = (const float*)ws;
    long ws_used = 0;
    gast_gemm_args big_a[GAST_GEMM_MAX_BATCH], bj_a[GAST_GEMM_MAX_BATCH];
    BigPlan big_p[GAST_GEMM_MAX_BATCH];
    BjPlan bj_p[GAST_GEMM_MAX_BATCH];
    int nbig = 0, nbj = 0;
    for (int d = 0; d < n; ++d) {
        if (args[d].dtype != args[0].dtype || args[d].out_f32 != args[0].out_f32 || !args[d].f8_scale != !args[0].f8_scale) return GAST_EINVAL;
        int M, gridM, gridN, vec_epi, splitk;
        int rc = gemm_plan(args[d], ws, ws_bytes, M, gridM, gridN, vec_epi, splitk);
        if (rc) return rc;
        if (gast_gemm_big_plan(args[d], big_p[nbig])) { big_a[nbig++] = args[d]; continue; }
        if (ws && gast_gemm_bj_plan(args[d], bj_p[nbj])) { bj_a[nbj++] = args[d]; continue; }
        long off = 0;
        if (splitk > 1) {                       // small-M job: its K ranges join the grid, its slice of the workspace follows the others'
            off = ws_used;
            const long need = (long)splitk * M * args[d].N;
            static const int multi_splitk = getenv("GAST_GEMM_MULTI_SPLITK") ? atoi(getenv("GAST_GEMM_MULTI_SPLITK")) : 1;
            if (!multi_splitk || (off + need) * (long)sizeof(float) > ws_bytes) {      // (0: bisecting aid) own split-K launch pair
                rc = gast_gemm_ws(&args[d], ws, ws_bytes, stream);
                if (rc) return rc;
                continue;
            }
            ws_used += (need + 63) / 64 * 64;
        }
        const int k = b.n++;
        b.a[k] = args[d];
        b.M[k] = M; b.gridM[k] = gridM; b.gridN[k] = gridN; b.vec_epi[k] = vec_epi;
        b.splitk[k] = splitk; b.ws_off[k] = off;
        b.first[k + 1] = b.first[k] + gridM * gridN * splitk;
        if (splitk > 1) {
            const int f = fb.n++;
            fb.a[f] = args[d]; fb.M[f] = M; fb.gridN[f] = gridN; fb.splitk[f] = splitk; fb.ws_off[f] = off;
            fb.first[f + 1] = fb.first[f] + gridN * ((M + 7) / 8);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (nbig) {
        int rc = gast_gemm_big_launch_multi(big_a, big_p, nbig, st);
        if (rc) return rc;
    }
    if (nbj) {
        int rc = gast_gemm_bj_launch_multi(bj_a, bj_p, nbj, st);
        if (rc) return rc;
    }
    if (b.n == 0) return 0;
    dim3 grid(b.first[b.n]), block(256);
    if (args[0].dtype == GAST_F32)
        hipLaunchKernelGGL((gemm_multi_kernel<float, float>), grid, block, 0, st, b);
    else if (args[0].dtype == GAST_F32X3)
        hipLaunchKernelGGL((gemm_multi_kernel<float, float, 1>), grid, block, 0, st, b);
    else if (args[0].dtype == GAST_F32X3H)
        hipLaunchKernelGGL((gemm_multi_kernel<float, float, 2>), grid, block, 0, st, b);
    else if (args[0].f8_scale && !args[0].out_f32)
        hipLaunchKernelGGL((gemm_multi_kernel<bf16_t, bf16_t, 0, true>), grid, block, 0, st, b);
    else if (args[0].out_f32)
        hipLaunchKernelGGL((gemm_multi_kernel<bf16_t, float>), grid, block, 0, st, b);
    else
        hipLaunchKernelGGL((gemm_multi_kernel<bf16_t, bf16_t>), grid, block, 0, st, b);
    GAST_CHECK_LAUNCH();
    if (fb.n) {
        dim3 fgrid(fb.first[fb.n]);
        if (args[0].dtype != GAST_BF16)
            hipLaunchKernelGGL((splitk_finish_multi_kernel<float, float>), fgrid, block, 0, st, fb);
        else if (args[0].out_f32)
            hipLaunchKernelGGL((splitk_finish_multi_kernel<bf16_t, float>), fgrid, block, 0, st, fb);
        else
            hipLaunchKernelGGL((splitk_finish_multi_kernel<bf16_t, bf16_t>), fgrid, block, 0, st, fb);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}


// ---- per-tensor power-of-two scales for the fp8 weight operands: out[0] = 2^floor(log2(448 / max|W|)), out[1] = 1 / out[0].
// Three small launches: zero the slots (a kernel, not one memset node per tensor), max |W| by blocks of 16K elements (atomicMax on the bit pattern: non-negative floats order
// like unsigned integers), then the conversion of every slot.
namespace {
struct F8ScaleBatch { gast_f8_scale_job j[GAST_F8_SCALE_MAX_BATCH]; int first[GAST_F8_SCALE_MAX_BATCH + 1]; int n; };
static_assert(sizeof(F8ScaleBatch) <= 3840, "F8ScaleBatch travels as a kernel argument");
constexpr int F8_CHUNK = 16384;
__global__ void __launch_bounds__(256) f8_absmax_kernel(const F8ScaleBatch b) {
    __shared__ float sred[4];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const gast_f8_scale_job& j = b.j[d];
    const long total = (long)j.R * j.K, i0 = (long)(blockIdx.x - b.first[d]) * F8_CHUNK;
    float m = 0.f;
    for (long i = i0 + threadIdx.x; i < min(total, i0 + F8_CHUNK); i += 256) {
        const int r = (int)(i / j.K), k = (int)(i - (long)r * j.K);
        m = fmaxf(m, fabsf(bf2f(((const bf16_t*)j.W)[(long)r * j.ldw + k])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax((unsigned*)j.out, __float_as_uint(fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]))));
}
__global__ void __launch_bounds__(64) f8_scale_zero_kernel(const F8ScaleBatch b) {
    if ((int)threadIdx.x < b.n) b.j[threadIdx.x].out[0] = 0.f;
}
__global__ void __launch_bounds__(64) f8_scale_finish_kernel(const F8ScaleBatch b) {
    if ((int)threadIdx.x >= b.n) return;
    float* out = b.j[threadIdx.x].out;
    const float m = out[0];
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e = (int)floorf(log2f(448.f / m));
        e = e < -40 ? -40 : (e > 40 ? 40 : e);
        s = exp2f((float)e);
    }
    out[0] = s;
    out[1] = 1.f / s;
}
}  // namespace

extern "C" int gast_f8_scale_multi(const gast_f8_scale_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 0) return GAST_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += GAST_F8_SCALE_MAX_BATCH) {
        F8ScaleBatch b;
        b.n = n - i0 < GAST_F8_SCALE_MAX_BATCH ? n - i0 : GAST_F8_SCALE_MAX_BATCH;
        b.first[0] = 0;
        for (int d = 0; d < b.n; ++d) {
            const gast_f8_scale_job& j = jobs[i0 + d];
            if (!j.W || !j.out || j.R < 1 || j.K < 1) return GAST_EINVAL;
            b.j[d] = j;
            b.first[d + 1] = b.first[d] + (int)(((long)j.R * j.K + F8_CHUNK - 1) / F8_CHUNK);
        }
        hipLaunchKernelGGL(f8_scale_zero_kernel, dim3(1), dim3(64), 0, st, b);
        GAST_CHECK_LAUNCH();
        hipLaunchKernelGGL(f8_absmax_kernel, dim3(b.first[b.n]), dim3(256), 0, st, b);
        GAST_CHECK_LAUNCH();
        hipLaunchKernelGGL(f8_scale_finish_kernel, dim3(1), dim3(64), 0, st, b);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}
