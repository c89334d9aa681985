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
        union { uint4 u; s16x8 s; } ua, ub;
        ua.u = a; ub.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.s, ub.s, acc, 0, 0, 0);
    }
};

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
        float lo = __uint_as_float(w[p] << 16), hi = __uint_as_float(w[p] & 0xffff0000u);
        lo = fmaxf(fmaf(lo, sc[2 * p], sh[2 * p]), 0.f);
        hi = fmaxf(fmaf(hi, sc[2 * p + 1], sh[2 * p + 1]), 0.f);
        if (drop) {
            lo *= drop_mul(key, thresh, inv_keep, e0 + 2 * p);
            hi *= drop_mul(key, thresh, inv_keep, e0 + 2 * p + 1);
        }
        w[p] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T, typename TO>
__global__ void __launch_bounds__(256) gemm_kernel(const gast_gemm_args a, int M, int gridM, int gridN) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    __shared__ __attribute__((aligned(16))) unsigned char sA[BM * LSTR];
    __shared__ __attribute__((aligned(16))) unsigned char sB[BN * LSTR];
    __shared__ int sRow[GAST_MAX_SEG][BM];
    __shared__ int sCrow[BM];
    __shared__ int sAddRow[BM];
    __shared__ float sRed[2][BN][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int logical = xcd_remap(blockIdx.x, gridM * gridN);
    const int mt = logical / gridN, nt = logical - mt * gridN;

    if (tid < BM) {
        int m = mt * BM + tid;
        int crow = -1, arow = -1;
        if (m < M) {
            int TJ = a.Tn * a.J;
            int b = m / TJ, rem = m - b * TJ;
            int t = rem / a.J, j = rem - t * a.J;
            for (int s = 0; s < a.nseg; ++s) sRow[s][tid] = (int)map_row(a.seg[s].map, b, t, j, a.J);
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

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    uint4 ra[4], rb[4];
    float sc[EPC], sh[EPC];
    int cur_s = 0, cur_k0 = 0;  // tile held in registers

    auto load_tile = [&](int s, int k0) {
        const gast_gemm_seg& sg = a.seg[s];
        const int k = k0 + chunk * EPC;
        const bool kin = k < sg.K;
        const T* Ab = (const T*)sg.A;
        const T* Wb = (const T*)sg.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = rbase + 32 * i;
            int row = sRow[s][r];
            ra[i] = make_uint4(0, 0, 0, 0);
            rb[i] = make_uint4(0, 0, 0, 0);
            if (kin && row >= 0) ra[i] = *(const uint4*)(Ab + (long)row * sg.lda + k);
            int n = nt * BN + r;
            if (kin && n < a.N) rb[i] = *(const uint4*)(Wb + (long)n * sg.ldw + k);
        }
        if (sg.pro != GAST_PRO_NONE && kin) {
#pragma unroll
            for (int q = 0; q < EPC; q += 4) {
                float4 s4 = *(const float4*)(sg.scale + k + q);
                float4 h4 = *(const float4*)(sg.shift + k + q);
                sc[q] = s4.x; sc[q + 1] = s4.y; sc[q + 2] = s4.z; sc[q + 3] = s4.w;
                sh[q] = h4.x; sh[q + 1] = h4.y; sh[q + 2] = h4.z; sh[q + 3] = h4.w;
            }
        }
    };

    auto store_tile = [&](int s, int k0) {
        const gast_gemm_seg& sg = a.seg[s];
        const int k = k0 + chunk * EPC;
        const bool kin = k < sg.K;
        const bool pro = sg.pro != GAST_PRO_NONE;
        const bool drop = sg.pro == GAST_PRO_BNRELU_DROP && thresh != 0;
        uint32_t key = 0;
        if (drop) key = drop_key(a.drop, sg.salt);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = rbase + 32 * i;
            uint4 v = ra[i];
            if (pro && kin) {
                int row = sRow[s][r];
                if (row >= 0)
                    v = prologue<T>(v, sc, sh, drop, key, thresh, inv_keep, (uint32_t)((long)row * sg.lda + k));
            }
            *(uint4*)(sA + r * LSTR + chunk * 16) = v;
            *(uint4*)(sB + r * LSTR + chunk * 16) = rb[i];
        }
    };

    load_tile(0, 0);
    while (cur_s < a.nseg) {
        __syncthreads();  // everyone finished reading the previous tile
        store_tile(cur_s, cur_k0);
        __syncthreads();
        int nxt_s = cur_s, nxt_k0 = cur_k0 + BK;
        if (nxt_k0 >= a.seg[cur_s].K) { nxt_s = cur_s + 1; nxt_k0 = 0; }
        if (nxt_s < a.nseg) load_tile(nxt_s, nxt_k0);
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
        cur_s = nxt_s;
        cur_k0 = nxt_k0;
    }

    // ------------------------------------------------------------------ epilogue
    const int epi = a.epi;
    const bool xdrop = epi == GAST_EPI_BNRELU_BWD && a.xdrop && thresh != 0;
    uint32_t xkey = 0;
    if (xdrop) xkey = drop_key(a.drop, a.xsalt);
    TO* Cb = (TO*)a.C;
    const T* Addb = (const T*)a.addend;
    const T* Xb = (const T*)a.X;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int nl = wc * 64 + ni * 32 + li;
        const int n = nt * BN + nl;
        const bool nin = n < a.N;
        float bias = (a.bias && nin) ? a.bias[n] : 0.f;
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

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int gast_gemm_row_blocks(int M) { return (M + BM - 1) / BM; }

extern "C" int gast_gemm(const gast_gemm_args* args, gast_stream_t stream) {
    if (!args) return GAST_EINVAL;
    const gast_gemm_args& a = *args;
    if (a.dtype != GAST_F32 && a.dtype != GAST_BF16) return GAST_EINVAL;
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG || !a.C || a.N < 1 || a.B < 1 || a.Tn < 1 || a.J < 1) return GAST_EINVAL;
    const int epc = a.dtype == GAST_F32 ? 4 : 8;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.A || !g.W || g.K < 1) return GAST_EINVAL;
        if (g.K % epc || g.lda % epc || g.ldw % epc || !aligned16(g.A) || !aligned16(g.W)) return GAST_EALIGN;
        if (g.pro != GAST_PRO_NONE && (!g.scale || !g.shift || !aligned16(g.scale) || !aligned16(g.shift))) return GAST_EINVAL;
        if (g.pro < 0 || g.pro > GAST_PRO_BNRELU_DROP) return GAST_EINVAL;
    }
    if (a.epi < 0 || a.epi > GAST_EPI_BNRELU_BWD) return GAST_EINVAL;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return GAST_EINVAL;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return GAST_EINVAL;
    long Ml = (long)a.B * a.Tn * a.J;
    if (Ml > 0x7fffff00L) return GAST_ERANGE;
    const int M = (int)Ml;
    const int gridM = (M + BM - 1) / BM, gridN = (a.N + BN - 1) / BN;
    dim3 grid(gridM * gridN), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (a.dtype == GAST_F32)
        hipLaunchKernelGGL((gemm_kernel<float, float>), grid, block, 0, st, a, M, gridM, gridN);
    else if (a.out_f32)
        hipLaunchKernelGGL((gemm_kernel<bf16_t, float>), grid, block, 0, st, a, M, gridM, gridN);
    else
        hipLaunchKernelGGL((gemm_kernel<bf16_t, bf16_t>), grid, block, 0, st, a, M, gridM, gridN);
    GAST_CHECK_LAUNCH();
    return 0;
}
