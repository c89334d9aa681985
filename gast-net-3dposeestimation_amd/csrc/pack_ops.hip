// Parameter packing / gradient unpacking for the GAST-Net plan (gfx950).
//
// The reference keeps 165 separate parameter tensors in PyTorch layouts (conv weights (Cout,Cin,k,1), SemCH W (2,Cin,Cout),
// per-head g/theta/phi, ...; state_dict contract).  The GEMM kernels want [N][K] K-contiguous operands in
// the activation dtype, in both orientations (forward and input-gradient), and the additive attention's theta/phi folded
// into one C-vector per head (global_attention.py:60-74 is rank-1 in (theta_i, phi_j)).  Doing that with torch ops costs
// ~600 tiny kernels per step; here ONE launch runs a table of strided 2-D copy jobs (with dtype conversion) and one launch
// runs the fold jobs.  The same copy kernel scatters the packed weight gradients back onto parameter-shaped gradients.
#include "common.h"

namespace {

// job word layout (int64 each): see gast_hip.h
constexpr int CJ_WORDS = 10;
constexpr int TILE = 32;
struct Bases { long v[8]; };   // passed by value: base byte addresses selected by the low 4 bits of every pointer word

__device__ __forceinline__ float load_any(const void* p, long idx, int is_bf16) {
    return is_bf16 ? bf2f(((const bf16_t*)p)[idx]) : ((const float*)p)[idx];
}

// one block = TQ 32x32 tiles (consecutive entries of the tile table, any jobs); 256 threads = 32 x 8 per tile.
// CJ: 10 words per job (gast_strided_copy) or 16 (gast_pack_all: + the pre-split image of the packed operand the tile lands in --
// words 10..15 = image word, ldimg (16-bit elements per k-group), element offset of the job's (0, 0) inside the operand, columns K
// of the operand, fp16-pair flag, reserved; image word 0 = none): the tile is then ALSO written as 16-bit hi/lo pairs in the
// k-group-major layout of gast_x3_image_multi, so the large-M GEMM's weight images no longer need a pass of their own over the
// packed fp32 operands (x3_image_kernel: 30 us per step + its graph-node boundary).
constexpr int CJ_WORDS_X = 16;
// Round 6: a block runs TQ tiles with the loads of all of them in flight together.  One tile per block was bound by its chain of
// dependent loads (tile entry -> job words -> element: ~7 us per block lifetime, 12 564 blocks of 4 KB for pack_all = 44 us per step at
// 1.9 TB/s); the tables and the per-element arithmetic are unchanged, so the operands and images are bit-equal.
constexpr int TQ = 4;
struct TileJob {
    const char* src; char* dst;
    int R, S, r0, c0, flags;
    long srs, scs, drs, dcs;
    const long* j;
};
template <int CJ>
__device__ __forceinline__ TileJob tile_job(const long* __restrict__ jobs, const int* __restrict__ t, const long* bases) {
    TileJob q;
    const long* j = jobs + (long)t[0] * CJ;
    q.j = j;
    q.src = (const char*)(bases[j[0] & 7] + (j[0] >> 4));     // (element-size independent) byte address
    q.dst = (char*)(bases[j[1] & 7] + (j[1] >> 4));
    q.R = (int)j[2]; q.S = (int)j[3];
    q.srs = j[4]; q.scs = j[5]; q.drs = j[6]; q.dcs = j[7];
    q.flags = (int)j[8];
    q.r0 = t[1] * TILE; q.c0 = t[2] * TILE;
    return q;
}
template <int CJ>
__device__ __forceinline__ void copy_tiles(const long* __restrict__ jobs, const int* __restrict__ tiles, int first, int ntiles, const long* bases,
                                           float (*tile)[TILE][TILE + 1]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int nq = min(TQ, ntiles - first);
    TileJob tj[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) tj[q] = tile_job<CJ>(jobs, tiles + (long)(first + (q < nq ? q : 0)) * 3, bases);
    // read with the source-contiguous dimension on tx
    float v[TQ][TILE / 8];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        const TileJob& J = tj[q];
        const int src_bf16 = J.flags & 1, zero_fill = (J.flags >> 3) & 1;
        const bool src_col_fast = (J.scs <= J.srs);
#pragma unroll
        for (int i = 0; i < TILE; i += 8) {
            const int r = src_col_fast ? J.r0 + ty + i : J.r0 + tx;
            const int c = src_col_fast ? J.c0 + tx : J.c0 + ty + i;
            v[q][i / 8] = (q < nq && r < J.R && c < J.S && !zero_fill) ? load_any(J.src, r * J.srs + c * J.scs, src_bf16) : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        const TileJob& J = tj[q];
        const bool src_col_fast = (J.scs <= J.srs);
#pragma unroll
        for (int i = 0; i < TILE; i += 8) {
            const int rl = src_col_fast ? ty + i : tx, cl = src_col_fast ? tx : ty + i;
            tile[q][rl][cl] = v[q][i / 8];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        if (q >= nq) break;
        const TileJob& J = tj[q];
        const int dst_bf16 = (J.flags >> 1) & 1, accumulate = (J.flags >> 2) & 1;
        const int R = J.R, S = J.S, r0 = J.r0, c0 = J.c0;
        const long drs = J.drs, dcs = J.dcs;
        char* dst = J.dst;
        const bool dst_col_fast = (dcs <= drs);
#pragma unroll
        for (int i = 0; i < TILE; i += 8) {
            int r = dst_col_fast ? r0 + ty + i : r0 + tx;
            int c = dst_col_fast ? c0 + tx : c0 + ty + i;
            if (r < R && c < S) {
                float x = tile[q][r - r0][c - c0];
                long o = r * drs + c * dcs;
                if (dst_bf16) ((bf16_t*)dst)[o] = f2bf(x);
                else if (accumulate) ((float*)dst)[o] += x;
                else ((float*)dst)[o] = x;
            }
        }
        if constexpr (CJ == CJ_WORDS_X) {
            const long* j = J.j;
            if (j[10] == 0) continue;
            // image: thread = (line along the operand's row direction, quad of 4 consecutive K values).  The operand's K runs along the
            // job's s when dcs == 1 and along its r when drs == 1 (a transposed twin); 4-aligned quads stay inside one 16-value k-group.
            bf16_t* img = (bf16_t*)(bases[j[10] & 7] + (j[10] >> 4));
            const long ldimg = j[11], rel = j[12];
            const int Kop = (int)j[13], f16 = (int)j[14];
            const int line = threadIdx.x >> 3, quad = threadIdx.x & 7;
            const bool k_on_s = dcs == 1;
            float x[4];
            bool in[4];
            long o0 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rl = k_on_s ? line : quad * 4 + e, cl = k_on_s ? quad * 4 + e : line;
                in[e] = r0 + rl < R && c0 + cl < S;
                x[e] = in[e] ? tile[q][rl][cl] : 0.f;
                if (e == 0) o0 = rel + (long)(r0 + rl) * drs + (long)(c0 + cl) * dcs;
            }
            if (!in[0]) continue;          // (validity is monotone along the quad)
            const long row = o0 / Kop;
            const int k = (int)(o0 - row * Kop);
            if (f16 == 2) {
                // layout image of a 16-BIT operand (round 5 kind 2; fused into this launch in round 6): the stored values themselves,
                // img[(k >> 5) * ldimg + row * 32 + (k & 31)]
                if (in[3] && (k & 3) == 0) {
                    bf16_t* o = img + (long)(k >> 5) * ldimg + row * 32 + (k & 31);
                    *(uint2*)o = make_uint2((uint32_t)f2bf(x[0]) | ((uint32_t)f2bf(x[1]) << 16), (uint32_t)f2bf(x[2]) | ((uint32_t)f2bf(x[3]) << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!in[e]) break;
                        const int kq = k + e;
                        img[(long)(kq >> 5) * ldimg + row * 32 + (kq & 31)] = f2bf(x[e]);
                    }
                }
                continue;
            }
            uint2 h, l;
            if (f16) split_pair4<2>(x[0], x[1], x[2], x[3], h, l);
            else split_pair4<1>(x[0], x[1], x[2], x[3], h, l);
            bf16_t* o = img + (long)(k >> 4) * ldimg + row * 32 + (k & 15);
            if (in[3] && (k & 3) == 0) {
                *(uint2*)o = h;
                *(uint2*)(o + 16) = l;
            } else {
                // ragged quad (a 2-row head block of an 8-channel model, a K offset that is not a multiple of 4): value by value -- the
                // neighbouring K positions belong to other jobs
                const bf16_t* hp = (const bf16_t*)&h;
                const bf16_t* lp = (const bf16_t*)&l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!in[e]) break;
                    const int kq = k + e;
                    bf16_t* oq = img + (long)(kq >> 4) * ldimg + row * 32 + (kq & 15);
                    oq[0] = hp[e];
                    oq[16] = lp[e];
                }
            }
        }
    }
}
__global__ void __launch_bounds__(256) strided_copy_kernel(const long* __restrict__ jobs, const int* __restrict__ tiles, int ntiles,
                                                           const Bases bs) {
    __shared__ float tile[TQ][TILE][TILE + 1];
    copy_tiles<CJ_WORDS>(jobs, tiles, (int)blockIdx.x * TQ, ntiles, bs.v, tile);
}

// fold: v[k] = sum_m W[m][k] * w[m]  (k < C, m < Ci), a = sum_m w[m] * b[m]
//   -> dst_row[k * ds_row]  and dst_col[k * ds_col] (activation dtype), bias_dst (fp32)
// grid = (njobs, ceil(C/32)); 256 threads = 32 columns x 8 m-lanes (each lane sums every 8th m, LDS combine)
// FJ: 12 words per job (gast_fold) or 18 (gast_pack_all: + words 12..17 = image word of the row destination's first element,
// its ldimg, image word of the column destination's first element, its ldimg, fp16-pair flags (bit 0 row, bit 1 column), reserved:
// element k of the row destination is K position k0r + k of one operand row, element k of the column destination is K position
// (fixed) of operand row k -- see gast_hip/binding.py)
constexpr int FJ_WORDS = 12;
constexpr int FJ_WORDS_X = 18;
__device__ __forceinline__ void split1(float v, int f16, bf16_t& hi, bf16_t& lo) {
    uint2 h, l;
    if (f16) split_pair4<2>(v, 0.f, 0.f, 0.f, h, l);
    else split_pair4<1>(v, 0.f, 0.f, 0.f, h, l);
    hi = (bf16_t)(h.x & 0xffffu);
    lo = (bf16_t)(l.x & 0xffffu);
}
template <int FJ>
__device__ __forceinline__ void fold_body(const long* __restrict__ jobs, const long* bases, int job, int by, float (*sred)[32]) {
    const long* j = jobs + (long)job * FJ;
    const float* W = (const float*)(bases[j[0] & 7] + (j[0] >> 4));
    const float* w = (const float*)(bases[j[1] & 7] + (j[1] >> 4));
    const float* b = (const float*)(bases[j[2] & 7] + (j[2] >> 4));
    const int Ci = (int)j[3], C = (int)j[4];
    if (by * 32 >= C) return;
    char* d_row = (char*)(bases[j[5] & 7] + (j[5] >> 4));
    const long ds_row = j[6];
    char* d_col = (char*)(bases[j[7] & 7] + (j[7] >> 4));
    const long ds_col = j[8];
    float* d_bias = (float*)(bases[j[9] & 7] + (j[9] >> 4));
    const int dst_bf16 = (int)j[10];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int k = by * 32 + cx;
    float acc = 0.f;
    if (k < C) {
#pragma unroll 4
        for (int m = ry; m < Ci; m += 8) acc = fmaf(W[(long)m * C + k], w[m], acc);
    }
    sred[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && k < C) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) v += sred[r][cx];
        if (dst_bf16) { ((bf16_t*)d_row)[k * ds_row] = f2bf(v); ((bf16_t*)d_col)[k * ds_col] = f2bf(v); }
        else { ((float*)d_row)[k * ds_row] = v; ((float*)d_col)[k * ds_col] = v; }
        if constexpr (FJ == FJ_WORDS_X) {
            if (j[17]) {         // layout images of 16-bit operands (kind 2): the stored value, 32 K positions per 64-byte row
                if (j[12]) {
                    bf16_t* ir = (bf16_t*)(bases[j[12] & 7] + (j[12] >> 4));
                    ir[(long)(k >> 5) * j[13] + (k & 31)] = f2bf(v);
                }
                if (j[14]) {
                    bf16_t* ic = (bf16_t*)(bases[j[14] & 7] + (j[14] >> 4));
                    ic[(long)k * 32] = f2bf(v);
                }
            } else {
            if (j[12]) {         // images: the row destination is K positions k0 + k of one operand row ...
                bf16_t* ir = (bf16_t*)(bases[j[12] & 7] + (j[12] >> 4));      // (address of K position 0 of that row in k-group 0)
                bf16_t hi, lo;
                split1(v, (int)j[16] & 1, hi, lo);
                bf16_t* o = ir + (long)(k >> 4) * j[13] + (k & 15);
                o[0] = hi;
                o[16] = lo;
            }
            if (j[14]) {         // ... the column destination one K position of operand row k
                bf16_t* ic = (bf16_t*)(bases[j[14] & 7] + (j[14] >> 4));      // (address of that K position in operand row 0)
                bf16_t hi, lo;
                split1(v, ((int)j[16] >> 1) & 1, hi, lo);
                bf16_t* o = ic + (long)k * 32;
                o[0] = hi;
                o[16] = lo;
            }
            }
        }
    }
    if (by == 0 && threadIdx.x == 0) {
        float a = 0.f;
        for (int m = 0; m < Ci; ++m) a = fmaf(w[m], b[m], a);
        *d_bias = a;
    }
}
__global__ void __launch_bounds__(256) fold_kernel(const long* __restrict__ jobs, const Bases bs) {
    __shared__ float sred[8][32];
    fold_body<FJ_WORDS>(jobs, bs.v, blockIdx.x, blockIdx.y, sred);
}

// gast_pack_all: every copy tile (with its image) and every fold block (with its images) of a step in ONE grid -- the three
// launches of the parameter packing (strided_copy, fold, x3_image: 68 us + three graph-node boundaries per step) as one.  The jobs
// are independent: a copy tile and a fold block never write the same element (operand or image).
__global__ void __launch_bounds__(256) pack_all_kernel(const long* __restrict__ cjobs, const int* __restrict__ tiles, int ntiles,
                                                       const long* __restrict__ fjobs, int fold_by, const Bases bs) {
    __shared__ float smem[TQ][TILE][TILE + 1];
    const int ncopy = (ntiles + TQ - 1) / TQ;
    if ((int)blockIdx.x < ncopy) {
        copy_tiles<CJ_WORDS_X>(cjobs, tiles, (int)blockIdx.x * TQ, ntiles, bs.v, smem);
        return;
    }
    const int fb = blockIdx.x - ncopy;
    fold_body<FJ_WORDS_X>(fjobs, bs.v, fb / fold_by, fb % fold_by, (float (*)[32])smem);
}

// unfold (gradient of fold): given dv[k] (fp32, stride 1) and da (fp32 scalar):
//   dW[m][k] (+)= w[m] * dv[k];   dw[m] (+)= sum_k W[m][k] * dv[k] + b[m] * da;   db[m] (+)= w[m] * da
// grid = (njobs, ceil(Ci/4)): one wave per row m, wave-shuffle reduction over k (no barriers)
constexpr int UJ_WORDS = 12;
__global__ void __launch_bounds__(256) unfold_kernel(const long* __restrict__ jobs, const Bases bs) {
    const long* bases = bs.v;
    const long* j = jobs + (long)blockIdx.x * UJ_WORDS;
    const float* dv = (const float*)(bases[j[0] & 7] + (j[0] >> 4));
    const float* da = (const float*)(bases[j[1] & 7] + (j[1] >> 4));
    const float* W = (const float*)(bases[j[2] & 7] + (j[2] >> 4));
    const float* w = (const float*)(bases[j[3] & 7] + (j[3] >> 4));
    const float* b = (const float*)(bases[j[4] & 7] + (j[4] >> 4));
    float* dW = (float*)(bases[j[5] & 7] + (j[5] >> 4));
    float* dw = (float*)(bases[j[6] & 7] + (j[6] >> 4));
    float* db = (float*)(bases[j[7] & 7] + (j[7] >> 4));
    const int Ci = (int)j[8], C = (int)j[9], accumulate = (int)j[10];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = blockIdx.y * 4 + wv;
    if (m >= Ci) return;
    const float dav = *da;
    const float wm = w[m];
    float part = 0.f;
    for (int k = lane; k < C; k += 64) {
        const float d = dv[k];
        const long o = (long)m * C + k;
        part = fmaf(W[o], d, part);
        if (accumulate) dW[o] += wm * d; else dW[o] = wm * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) {
        const float tot = part + b[m] * dav;
        if (accumulate) { dw[m] += tot; db[m] += wm * dav; } else { dw[m] = tot; db[m] = wm * dav; }
    }
}

}  // namespace

extern "C" int gast_strided_copy(const int64_t* jobs, const int32_t* tiles, int ntiles, const int64_t* bases, gast_stream_t stream) {
    if (!jobs || !tiles || !bases || ntiles < 0) return GAST_EINVAL;
    if (ntiles == 0) return 0;
    Bases b;
    for (int i = 0; i < 8; ++i) b.v[i] = bases[i];
    hipLaunchKernelGGL(strided_copy_kernel, dim3((ntiles + TQ - 1) / TQ), dim3(256), 0, (hipStream_t)stream, (const long*)jobs, tiles, ntiles, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_pack_all(const int64_t* cjobs, const int32_t* tiles, int ntiles, const int64_t* fjobs, int nfold, int max_C,
                             const int64_t* bases, gast_stream_t stream) {
    if (!bases || ntiles < 0 || nfold < 0 || (ntiles && (!cjobs || !tiles)) || (nfold && (!fjobs || max_C < 1))) return GAST_EINVAL;
    const int fold_by = nfold ? (max_C + 31) / 32 : 1;
    const long nblk = (long)((ntiles + TQ - 1) / TQ) + (long)nfold * fold_by;
    if (nblk == 0) return 0;
    Bases b;
    for (int i = 0; i < 8; ++i) b.v[i] = bases[i];
    hipLaunchKernelGGL(pack_all_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const long*)cjobs, tiles, ntiles,
                       (const long*)fjobs, fold_by, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_fold(const int64_t* jobs, int njobs, int max_C, const int64_t* bases, gast_stream_t stream) {
    if (!jobs || !bases || njobs < 0 || max_C < 1) return GAST_EINVAL;
    const int max_c32 = (max_C + 31) / 32;
    if (njobs == 0) return 0;
    Bases b;
    for (int i = 0; i < 8; ++i) b.v[i] = bases[i];
    hipLaunchKernelGGL(fold_kernel, dim3(njobs, max_c32), dim3(256), 0, (hipStream_t)stream, (const long*)jobs, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_unfold(const int64_t* jobs, int njobs, int max_Ci, const int64_t* bases, gast_stream_t stream) {
    if (!jobs || !bases || njobs < 0 || max_Ci < 1) return GAST_EINVAL;
    const int max_ci4 = (max_Ci + 3) / 4;
    if (njobs == 0) return 0;
    Bases b;
    for (int i = 0; i < 8; ++i) b.v[i] = bases[i];
    hipLaunchKernelGGL(unfold_kernel, dim3(njobs, max_ci4), dim3(256), 0, (hipStream_t)stream, (const long*)jobs, b);
    GAST_CHECK_LAUNCH();
    return 0;
}
