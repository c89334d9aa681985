// BatchNorm (two-phase), residual, input-side (init_bn + expand_conv) and column-sum kernels for gfx950.
// Reference: every nn.BatchNorm2d(momentum=0.1) of model/gast_net.py:20,58-59,147,149, model/local_attention.py:117-123,
// model/global_attention.py:95; the residual adds gast_net.py:170-174 / :243-247; init_bn + expand_conv :163-164.
// All kernels are pure streaming (HBM-bound): 4 channels per thread (16 B fp32 / 8 B bf16 accesses), channel-major
// coalescing, column statistics reduced per block in LDS and written as partial rows that `gast_bn_finalize`
// combines in double precision (deterministic, no atomics on the statistics).
#include "common.h"
#include "bn_finalize.h"

namespace {
using namespace gastbn;

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

struct RowCfg { int TPR, RB; };
inline RowCfg row_cfg(int N) {
    RowCfg c;
    int n4 = N / 4;
    c.TPR = n4 < 256 ? n4 : 256;
    if (c.TPR < 1) c.TPR = 1;
    c.RB = 256 / c.TPR;
    return c;
}
inline int row_blocks(long rows, int N) {
    RowCfg c = row_cfg(N);
    long nb = (rows + c.RB - 1) / c.RB;
    return (int)(nb < 1024 ? nb : 1024);
}

// reduce NV float4 accumulators over the RB row slots of the block; slot 0 receives the totals
template <int NV>
__device__ __forceinline__ void slot_reduce(float4 (&v)[NV], float (*sred)[4 * NV], int tid, int slot, int ct, int TPR, int RB) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        sred[tid][4 * i] = v[i].x; sred[tid][4 * i + 1] = v[i].y; sred[tid][4 * i + 2] = v[i].z; sred[tid][4 * i + 3] = v[i].w;
    }
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = make_float4(0, 0, 0, 0);
        for (int sl = 0; sl < RB; ++sl) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float* s = &sred[sl * TPR + ct][4 * i];
                v[i].x += s[0]; v[i].y += s[1]; v[i].z += s[2]; v[i].w += s[3];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ BN finalize
// 256 threads = 32 columns x 8 partial-row lanes; each lane sums every 8th partial row (independent loads, unrolled),
// then the 8 lanes are combined in LDS in double precision.  (One thread per column walking all row blocks serially
// exposed one L2 latency per row block: 80 us for 425 blocks.)
constexpr int FIN_COLS = 32, FIN_LANES = 8;          // the fused short-tensor kernel (its apply half wants 32 columns)
// (the stand-alone finalizes' geometry -- 4 columns x 64 lanes per block -- and their bodies live in bn_finalize.h)

template <int COLS, int LANES>
__device__ __forceinline__ void finalize_sums(const float* __restrict__ partials, int nblk, int ncol_total, int col, bool valid,
                                              double (*sred)[COLS][2], int cx, int ry, double& s1, double& s2) {
    double a1 = 0.0, a2 = 0.0;
    if (valid) {
#pragma unroll 8
        for (int b = ry; b < nblk; b += LANES) {
            const float2 p = *(const float2*)(partials + ((long)b * ncol_total + col) * 2);
            a1 += (double)p.x;
            a2 += (double)p.y;
        }
    }
    sred[ry][cx][0] = a1;
    sred[ry][cx][1] = a2;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int r = 0; r < LANES; ++r) { s1 += sred[r][cx][0]; s2 += sred[r][cx][1]; }
}

__device__ __forceinline__ void bn_finalize_body(const gast_bn_fin_job& j) {
    __shared__ double sred[4][FINS_COLS][2];
    if ((int)blockIdx.x * FINS_COLS >= j.N) return;
    bn_finalize_unit(j, blockIdx.x, sred);
}

// up to GAST_BN_MAX_BATCH independent finalizes in one launch: blockIdx.y = job
struct BnFinBatch { gast_bn_fin_job j[GAST_BN_MAX_BATCH]; };
__global__ void __launch_bounds__(256) bn_finalize_multi_kernel(const BnFinBatch b) { bn_finalize_body(b.j[blockIdx.y]); }

// every eval-mode BatchNorm of the model in ONE launch (they depend on parameters and buffers only): blockIdx.y = job
struct BnEvalBatch { gast_bn_eval_job j[GAST_BN_EVAL_MAX_BATCH]; };
__global__ void bn_eval_multi_kernel(const BnEvalBatch b, float eps) {
    const gast_bn_eval_job& j = b.j[blockIdx.y];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= j.N) return;
    const float sc = j.gamma[n] / sqrtf(j.running_var[n] + eps);
    j.scale[n] = sc;
    j.shift[n] = j.centered ? j.beta[n] : j.beta[n] - j.running_mean[n] * sc;
}

__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float eps, int N, float* scale, float* shift, int centered) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float sc = gamma[n] / sqrtf(rv[n] + eps);
    scale[n] = sc;
    shift[n] = centered ? beta[n] : beta[n] - rm[n] * sc;
}

__device__ __forceinline__ void bn_bwd_finalize_body(const gast_bn_bwd_fin_job& j) {
    __shared__ double sred[4][FINS_COLS][2];
    if ((int)blockIdx.x * FINS_COLS >= j.N) return;
    bn_bwd_finalize_unit(j, blockIdx.x, sred);
}

struct BnBwdFinBatch { gast_bn_bwd_fin_job j[GAST_BN_MAX_BATCH]; };
__global__ void __launch_bounds__(256) bn_bwd_finalize_multi_kernel(const BnBwdFinBatch b) { bn_bwd_finalize_body(b.j[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ elementwise
// FR (round 6): rows = B * T_total * J and only the frames whose bit is set in `frames` carry a gradient -- the others are known to be zero
// and are NOT read (they need not even be initialised): the input gradient of the last dilated level reaches 3 of its 19 input frames
// (reference gast_net.py:173, T' = 1), and both the zero fill of the rest and its read-back were at the HBM roofline.
template <typename T, bool FR = false>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(T* __restrict__ dz, int lddz, const T* __restrict__ X, int ldx, long rows, int N,
                                                           const float* ka, const float* kb,
                                                           const float* kc, int TPR, int RB, int T_total = 1, int J = 1,
                                                           unsigned long long frames = 0) {
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    if (slot >= RB) return;
    const int N4 = N >> 2;
    for (int cg = ct; cg < N4; cg += TPR) {
        const int c = cg * 4;
        const float4 a = *(const float4*)(ka + c), b = *(const float4*)(kb + c), k = *(const float4*)(kc + c);
        for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
            const bool live = !FR || ((frames >> (unsigned)((r / J) % T_total)) & 1ull);
            float4 d = live ? ld4(dz + r * lddz + c) : make_float4(0, 0, 0, 0), x = ld4(X + r * ldx + c);
            d.x = fmaf(a.x, d.x, fmaf(b.x, x.x, k.x));
            d.y = fmaf(a.y, d.y, fmaf(b.y, x.y, k.y));
            d.z = fmaf(a.z, d.z, fmaf(b.z, x.z, k.z));
            d.w = fmaf(a.w, d.w, fmaf(b.w, x.w, k.w));
            st4(dz + r * lddz + c, d);
        }
    }
}

// BatchNorm backward for SHORT tensors (the M = B*J rows of the last stage, small models): finalize and apply in ONE launch.
// A block owns 32 columns of one job: it reduces the partial column sums (as bn_bwd_finalize), keeps the three coefficients in
// LDS and rewrites FUSED_ROWS rows of its columns (dz <- ka*dz + kb*x + kc; grid.z walks the rows, every split redoes the cheap
// reduction).  256 threads = 8 column quads x 32 row lanes.
constexpr int FUSED_ROWS = 256;      // rows per block of the fused kernel (grid.z splits the rows)
struct BnBwdFusedBatch { gast_bn_bwd_job j[GAST_BN_MAX_BATCH]; };
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_fused_kernel(const BnBwdFusedBatch b) {
    __shared__ double sred[FIN_LANES][FIN_COLS][2];
    __shared__ float scoef[3][FIN_COLS];
    const gast_bn_bwd_job& j = b.j[blockIdx.y];
    const int N = j.f.N;
    if ((int)blockIdx.x * FIN_COLS >= N) return;
    {
        const int cx = threadIdx.x & (FIN_COLS - 1), ry = threadIdx.x / FIN_COLS;
        const int n = blockIdx.x * FIN_COLS + cx;
        double s1, s2;
        finalize_sums<FIN_COLS, FIN_LANES>(j.f.partials, j.f.nblk, j.f.ncol_total, j.f.col0 + n, n < N, sred, cx, ry, s1, s2);
        if (ry == 0) {
            float a = 0.f, bb = 0.f, c = 0.f;
            if (n < N) {
                const double mu = j.f.mean[n], r = j.f.rstd[n], g = j.f.gamma[n];
                const double dg = r * (s2 - mu * s1), db = s1;
                if (blockIdx.z == 0) {          // every row split recomputes the coefficients, one writes the parameter gradients
                    if (j.f.accumulate) { j.f.dgamma[n] += (float)dg; j.f.dbeta[n] += (float)db; }
                    else { j.f.dgamma[n] = (float)dg; j.f.dbeta[n] = (float)db; }
                }
                const double ad = g * r, bd = -g * r * r * dg / j.f.count;
                a = (float)ad; bb = (float)bd; c = (float)(-bd * mu - ad * db / j.f.count);
            }
            scoef[0][cx] = a; scoef[1][cx] = bb; scoef[2][cx] = c;
        }
    }
    __syncthreads();
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;          // column quad, row lane
    const int c = blockIdx.x * FIN_COLS + q * 4;
    if (c >= N) return;
    const float4 a = *(const float4*)&scoef[0][q * 4], kb = *(const float4*)&scoef[1][q * 4], kc = *(const float4*)&scoef[2][q * 4];
    T* dz = (T*)j.dz;
    const T* X = (const T*)j.X;
    const long r_end = min(j.rows, (long)(blockIdx.z + 1) * FUSED_ROWS);
    for (long r = (long)blockIdx.z * FUSED_ROWS + rl; r < r_end; r += 32) {
        float4 d = ld4(dz + r * j.lddz + c), x = ld4(X + r * j.ldx + c);
        d.x = fmaf(a.x, d.x, fmaf(kb.x, x.x, kc.x));
        d.y = fmaf(a.y, d.y, fmaf(kb.y, x.y, kc.y));
        d.z = fmaf(a.z, d.z, fmaf(kb.z, x.z, kc.z));
        d.w = fmaf(a.w, d.w, fmaf(kb.w, x.w, kc.w));
        st4(dz + r * j.lddz + c, d);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) bnrelu_apply_kernel(const T* __restrict__ X, int ldx, long rows, int N,
                                                           const float* scale, const float* shift,
                                                           T* __restrict__ Y, int ldy, int use_drop, uint32_t salt, gast_dropout drop,
                                                           int TPR, int RB) {
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    if (slot >= RB) return;
    const int N4 = N >> 2;
    const bool dr = use_drop && drop.thresh != 0;
    const uint32_t key = dr ? drop_key(drop, salt) : 0u;
    for (int cg = ct; cg < N4; cg += TPR) {
        const int c = cg * 4;
        const float4 s = *(const float4*)(scale + c), h = *(const float4*)(shift + c);
        for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
            float4 x = ld4(X + r * ldx + c);
            x.x = fmaxf(fmaf(x.x, s.x, h.x), 0.f);
            x.y = fmaxf(fmaf(x.y, s.y, h.y), 0.f);
            x.z = fmaxf(fmaf(x.z, s.z, h.z), 0.f);
            x.w = fmaxf(fmaf(x.w, s.w, h.w), 0.f);
            if (dr) {       // the dropout stream is indexed by the element offset in the SOURCE tensor X
                const uint32_t e0 = (uint32_t)(r * ldx + c);
                x.x *= drop_mul(key, drop.thresh, drop.inv_keep, e0);
                x.y *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 1);
                x.z *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 2);
                x.w *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 3);
            }
            st4(Y + r * ldy + c, x);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) bnrelu_bwd_mask_kernel(const T* dY, int lddy, const T* __restrict__ X, int ldx,
                                                              long rows, int N, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int use_drop, uint32_t salt,
                                                              gast_dropout drop, T* dz, int lddz,
                                                              float* __restrict__ partials, int TPR, int RB) {
    __shared__ float sred[256][8];
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    const int N4 = N >> 2;
    const bool dr = use_drop && drop.thresh != 0;
    const uint32_t key = dr ? drop_key(drop, salt) : 0u;
    for (int cg0 = 0; cg0 < N4; cg0 += TPR) {
        const int cg = cg0 + ct;
        const int c = cg * 4;
        float4 acc[2];
        acc[0] = acc[1] = make_float4(0, 0, 0, 0);
        if (slot < RB && cg < N4) {
            const float4 s = *(const float4*)(scale + c), h = *(const float4*)(shift + c);
            for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
                float4 d = ld4(dY + r * lddy + c), x = ld4(X + r * ldx + c);
                if (!(fmaf(x.x, s.x, h.x) > 0.f)) d.x = 0.f;
                if (!(fmaf(x.y, s.y, h.y) > 0.f)) d.y = 0.f;
                if (!(fmaf(x.z, s.z, h.z) > 0.f)) d.z = 0.f;
                if (!(fmaf(x.w, s.w, h.w) > 0.f)) d.w = 0.f;
                if (dr) {
                    uint32_t e0 = (uint32_t)(r * ldx + c);
                    d.x *= drop_mul(key, drop.thresh, drop.inv_keep, e0);
                    d.y *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 1);
                    d.z *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 2);
                    d.w *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 3);
                }
                d = rnd4(d, (const T*)nullptr);
                st4(dz + r * lddz + c, d);
                acc[0].x += d.x; acc[0].y += d.y; acc[0].z += d.z; acc[0].w += d.w;
                acc[1].x = fmaf(d.x, x.x, acc[1].x); acc[1].y = fmaf(d.y, x.y, acc[1].y);
                acc[1].z = fmaf(d.z, x.z, acc[1].z); acc[1].w = fmaf(d.w, x.w, acc[1].w);
            }
        }
        slot_reduce<2>(acc, sred, tid, slot, ct, TPR, RB);
        if (slot == 0 && cg < N4) {
            float* pp = partials + ((long)blockIdx.x * N + c) * 2;
            pp[0] = acc[0].x; pp[1] = acc[1].x; pp[2] = acc[0].y; pp[3] = acc[1].y;
            pp[4] = acc[0].z; pp[5] = acc[1].z; pp[6] = acc[0].w; pp[7] = acc[1].w;
        }
    }
}

// ------------------------------------------------------------------------------------------------ shrink layer (round 6)
// Reference gast_net.py:99,176-178: shrink = Conv2d(2C * 2^(L-1), 3, 1, bias=False) on the M = B*T'*J rows that survive the temporal
// stages.  With D = 3 output columns it is a row-wise dot product, not a GEMM: on the GEMM kernels it ran 34 blocks with a 32-step
// serial K loop (20 us), its input gradient a K = 8 GEMM whose whole cost is the epilogue (19 us).
//   forward   pred[r, d] = sum_k relu(scale[k] * O[r, k] + shift[k]) * W[d][k]            (fp32 out; a wave owns two rows)
//   backward  dO[r, k] = [scale[k] * O[r, k] + shift[k] > 0] * sum_d dp[r, d] * W[d][k]  + the BatchNorm-backward column sums
//             {sum dO, sum dO * O} of every SHRINK_ROWS-row block (same partial-row contract as GAST_EPI_BNRELU_BWD / gast_bnrelu_bwd_mask)
constexpr int SHRINK_ROWS = 32;          // rows per block of the backward kernel (x 256 columns)
constexpr int SHRINK_MAXD = 4;
template <typename T>
__global__ void __launch_bounds__(256) shrink_fwd_kernel(const T* __restrict__ O, int ldo, long rows, int K, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ W, int ldw, int D,
                                                         float* __restrict__ pred, int ldp) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wv) * 2;
    if (r0 >= rows) return;
    const bool two = r0 + 1 < rows;
    const T* o0 = O + r0 * ldo;
    const T* o1 = O + (two ? r0 + 1 : r0) * ldo;
    float acc[2][SHRINK_MAXD];
#pragma unroll
    for (int d = 0; d < SHRINK_MAXD; ++d) acc[0][d] = acc[1][d] = 0.f;
    const int K4 = K >> 2;
#pragma unroll 4
    for (int k4 = lane; k4 < K4; k4 += 64) {
        const int k = k4 * 4;
        const float4 s = *(const float4*)(scale + k), h = *(const float4*)(shift + k);
        float4 x0 = ld4(o0 + k), x1 = ld4(o1 + k);
        x0.x = fmaxf(fmaf(x0.x, s.x, h.x), 0.f); x0.y = fmaxf(fmaf(x0.y, s.y, h.y), 0.f);
        x0.z = fmaxf(fmaf(x0.z, s.z, h.z), 0.f); x0.w = fmaxf(fmaf(x0.w, s.w, h.w), 0.f);
        x1.x = fmaxf(fmaf(x1.x, s.x, h.x), 0.f); x1.y = fmaxf(fmaf(x1.y, s.y, h.y), 0.f);
        x1.z = fmaxf(fmaf(x1.z, s.z, h.z), 0.f); x1.w = fmaxf(fmaf(x1.w, s.w, h.w), 0.f);
#pragma unroll
        for (int d = 0; d < SHRINK_MAXD; ++d) {
            if (d < D) {
                const float4 w = ld4(W + (long)d * ldw + k);
                acc[0][d] = fmaf(x0.w, w.w, fmaf(x0.z, w.z, fmaf(x0.y, w.y, fmaf(x0.x, w.x, acc[0][d]))));
                acc[1][d] = fmaf(x1.w, w.w, fmaf(x1.z, w.z, fmaf(x1.y, w.y, fmaf(x1.x, w.x, acc[1][d]))));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < SHRINK_MAXD; ++d)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[i][d] += __shfl_xor(acc[i][d], off);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < SHRINK_MAXD; ++d) {
            if (d < D) {
                pred[r0 * ldp + d] = acc[0][d];
                if (two) pred[(r0 + 1) * ldp + d] = acc[1][d];
            }
        }
    }
}

// block = (SHRINK_ROWS rows, 256 columns): 256 threads = 64 column quads x 4 row lanes, 8 rows per thread (all loads in flight)
template <typename T>
__global__ void __launch_bounds__(256) shrink_bwd_kernel(const T* __restrict__ dp, int lddp, const T* __restrict__ W, int ldw, int D,
                                                         const T* __restrict__ O, int ldo, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, long rows, int K, T* __restrict__ dO, int lddo,
                                                         float* __restrict__ partials) {
    __shared__ float sred[4][64][8];
    __shared__ float sdp[SHRINK_ROWS][SHRINK_MAXD];
    const int tid = threadIdx.x, cq = tid & 63, rl = tid >> 6;
    const long rb = (long)blockIdx.x * SHRINK_ROWS;
    const int c = blockIdx.y * 256 + cq * 4;
    if (tid < SHRINK_ROWS * SHRINK_MAXD) {
        const int r = tid / SHRINK_MAXD, d = tid - r * SHRINK_MAXD;
        sdp[r][d] = (rb + r < rows && d < D) ? Elem<T>::ld(dp + (rb + r) * lddp + d) : 0.f;
    }
    __syncthreads();
    float4 a1 = make_float4(0, 0, 0, 0), a2 = make_float4(0, 0, 0, 0);
    if (c < K) {
        const float4 s = *(const float4*)(scale + c), h = *(const float4*)(shift + c);
        float4 w[SHRINK_MAXD];
#pragma unroll
        for (int d = 0; d < SHRINK_MAXD; ++d) w[d] = d < D ? ld4(W + (long)d * ldw + c) : make_float4(0, 0, 0, 0);
        float4 x[SHRINK_ROWS / 4];
#pragma unroll
        for (int i = 0; i < SHRINK_ROWS / 4; ++i) {
            const long r = rb + rl + 4 * i;
            x[i] = r < rows ? ld4(O + r * ldo + c) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < SHRINK_ROWS / 4; ++i) {
            const int rr = rl + 4 * i;
            const long r = rb + rr;
            if (r >= rows) continue;
            float4 g = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int d = 0; d < SHRINK_MAXD; ++d) {
                const float p = sdp[rr][d];
                g.x = fmaf(p, w[d].x, g.x); g.y = fmaf(p, w[d].y, g.y); g.z = fmaf(p, w[d].z, g.z); g.w = fmaf(p, w[d].w, g.w);
            }
            if (!(fmaf(x[i].x, s.x, h.x) > 0.f)) g.x = 0.f;
            if (!(fmaf(x[i].y, s.y, h.y) > 0.f)) g.y = 0.f;
            if (!(fmaf(x[i].z, s.z, h.z) > 0.f)) g.z = 0.f;
            if (!(fmaf(x[i].w, s.w, h.w) > 0.f)) g.w = 0.f;
            g = rnd4(g, (const T*)nullptr);
            st4(dO + r * lddo + c, g);
            a1.x += g.x; a1.y += g.y; a1.z += g.z; a1.w += g.w;
            a2.x = fmaf(g.x, x[i].x, a2.x); a2.y = fmaf(g.y, x[i].y, a2.y); a2.z = fmaf(g.z, x[i].z, a2.z); a2.w = fmaf(g.w, x[i].w, a2.w);
        }
    }
    float* sr = sred[rl][cq];
    sr[0] = a1.x; sr[1] = a2.x; sr[2] = a1.y; sr[3] = a2.y; sr[4] = a1.z; sr[5] = a2.z; sr[6] = a1.w; sr[7] = a2.w;
    __syncthreads();
    if (rl == 0 && c < K) {
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = ((sred[0][cq][q] + sred[1][cq][q]) + sred[2][cq][q]) + sred[3][cq][q];
        float* pp = partials + ((long)blockIdx.x * K + c) * 2;
        *(float4*)pp = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)(pp + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) residual_fwd_kernel(const T* __restrict__ O, int ldo, gast_rowmap omap,
                                                           const float* __restrict__ scO, const float* __restrict__ shO,
                                                           const T* __restrict__ T2, int ldt, const float* sc2,
                                                           const float* sh2, int use_drop, uint32_t salt,
                                                           gast_dropout drop, int Tn, int J, long rows, int N,
                                                           T* __restrict__ Xn, int ldxn, int TPR, int RB) {
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    if (slot >= RB) return;
    const int N4 = N >> 2;
    const bool dr = use_drop && drop.thresh != 0;
    const uint32_t key = dr ? drop_key(drop, salt) : 0u;
    const int TJ = Tn * J;
    for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
        int m = (int)r;
        int b = m / TJ, rem = m - b * TJ;
        int t = rem / J, j = rem - t * J;
        long orow = map_row(omap, b, t, j, J);
        for (int cg = ct; cg < N4; cg += TPR) {
            const int c = cg * 4;
            float4 res = make_float4(0, 0, 0, 0);
            if (orow >= 0) {
                float4 o = ld4(O + orow * ldo + c);
                const float4 s = *(const float4*)(scO + c), h = *(const float4*)(shO + c);
                res.x = fmaxf(fmaf(o.x, s.x, h.x), 0.f); res.y = fmaxf(fmaf(o.y, s.y, h.y), 0.f);
                res.z = fmaxf(fmaf(o.z, s.z, h.z), 0.f); res.w = fmaxf(fmaf(o.w, s.w, h.w), 0.f);
            }
            float4 x = ld4(T2 + r * ldt + c);
            const float4 s = *(const float4*)(sc2 + c), h = *(const float4*)(sh2 + c);
            x.x = fmaxf(fmaf(x.x, s.x, h.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, h.y), 0.f);
            x.z = fmaxf(fmaf(x.z, s.z, h.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, h.w), 0.f);
            if (dr) {
                uint32_t e0 = (uint32_t)(r * ldt + c);
                x.x *= drop_mul(key, drop.thresh, drop.inv_keep, e0);
                x.y *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 1);
                x.z *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 2);
                x.w *= drop_mul(key, drop.thresh, drop.inv_keep, e0 + 3);
            }
            res.x += x.x; res.y += x.y; res.z += x.z; res.w += x.w;
            st4(Xn + r * ldxn + c, res);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ X, int ldx, long rows, int N, float* __restrict__ out,
                                                     int TPR, int RB) {
    __shared__ float sred[256][4];
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    const int N4 = N >> 2;
    for (int cg0 = 0; cg0 < N4; cg0 += TPR) {
        const int cg = cg0 + ct;
        const int c = cg * 4;
        float4 acc[1];
        acc[0] = make_float4(0, 0, 0, 0);
        if (slot < RB && cg < N4) {
            for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
                float4 x = ld4(X + r * ldx + c);
                acc[0].x += x.x; acc[0].y += x.y; acc[0].z += x.z; acc[0].w += x.w;
            }
        }
        slot_reduce<1>(acc, sred, tid, slot, ct, TPR, RB);
        if (slot == 0 && cg < N4) {
            atomicAdd(out + c, acc[0].x); atomicAdd(out + c + 1, acc[0].y);
            atomicAdd(out + c + 2, acc[0].z); atomicAdd(out + c + 3, acc[0].w);
        }
    }
}

// ------------------------------------------------------------------------------------------------ input side
constexpr int IN_ROWS_PER_BLOCK = 1024;      // 4 rows per thread: 58 blocks on the B=128 window batch (15 blocks of 4096 rows took 12 us)

__global__ void __launch_bounds__(256) input_stats_kernel(const float* __restrict__ x, long rows, int F_in, float* __restrict__ partials) {
    // partials[blk][f][2]; F_in <= 8
    __shared__ float sred[256][16];
    const int tid = threadIdx.x;
    float s1[8], s2[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) { s1[f] = 0.f; s2[f] = 0.f; }
    long r0 = (long)blockIdx.x * IN_ROWS_PER_BLOCK;
    long r1 = r0 + IN_ROWS_PER_BLOCK < rows ? r0 + IN_ROWS_PER_BLOCK : rows;
    for (long r = r0 + tid; r < r1; r += 256) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (f < F_in) { float v = x[r * F_in + f]; s1[f] += v; s2[f] = fmaf(v, v, s2[f]); }
        }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) { sred[tid][2 * f] = s1[f]; sred[tid][2 * f + 1] = s2[f]; }
    __syncthreads();
    if (tid < 2 * F_in) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += sred[i][tid];
        partials[(long)blockIdx.x * F_in * 2 + tid] = t;
    }
}

constexpr int KMAX = 16;  // F_in * k0 (2 features x up to 7 taps, or 3 x 5)

template <typename T>
__global__ void __launch_bounds__(256) expand_fwd_kernel(const float* __restrict__ x, int B, int T_in, int J, int F_in, int k0,
                                                         int t_stride, int T_out, const float* __restrict__ W,
                                                         const float* sc0, const float* sh0, int C,
                                                         T* __restrict__ E, int lde, float* __restrict__ partials, int TPR, int RB,
                                                         const float* __restrict__ center) {
    extern __shared__ __attribute__((aligned(16))) float sW[];  // [K0][C] then sred
    __shared__ float sred[256][8];
    __shared__ float sBN0[2][KMAX];
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    const int K0 = F_in * k0;
    // scale / shift of the input features through LDS
    if (tid < F_in) {
        sBN0[0][tid] = sc0[tid];
        sBN0[1][tid] = sh0[tid];
    }
    for (int t = tid; t < K0 * C; t += 256) {
        int c = t / K0, kk = t - c * K0;     // W is [c][f][tap] = [c][kk]
        sW[kk * C + c] = W[t];
    }
    __syncthreads();
    const int C4 = C >> 2;
    const long rows = (long)B * T_out * J;
    const int TJ = T_out * J;
    for (int cg0 = 0; cg0 < C4; cg0 += TPR) {
        const int cg = cg0 + ct;
        const int c = cg * 4;
        float4 acc[2];
        acc[0] = acc[1] = make_float4(0, 0, 0, 0);
        if (slot < RB && cg < C4) {
            for (long r = (long)blockIdx.x * RB + slot; r < rows; r += (long)gridDim.x * RB) {
                int m = (int)r;
                int b = m / TJ, rem = m - b * TJ;
                int t = rem / J, j = rem - t * J;
                float4 e = make_float4(0, 0, 0, 0);
                for (int f = 0; f < F_in; ++f) {
                    float s = sBN0[0][f], h = sBN0[1][f];
                    for (int tap = 0; tap < k0; ++tap) {
                        long xr = ((long)b * T_in + t * t_stride + tap) * J + j;
                        float xv = fmaf(x[xr * F_in + f], s, h);
                        float4 w = *(const float4*)(sW + (f * k0 + tap) * C + c);
                        e.x = fmaf(xv, w.x, e.x); e.y = fmaf(xv, w.y, e.y); e.z = fmaf(xv, w.z, e.z); e.w = fmaf(xv, w.w, e.w);
                    }
                }
                if (center) { const float4 cv = *(const float4*)(center + c); e.x -= cv.x; e.y -= cv.y; e.z -= cv.z; e.w -= cv.w; }
                e = rnd4(e, (const T*)nullptr);
                st4(E + r * lde + c, e);
                acc[0].x += e.x; acc[0].y += e.y; acc[0].z += e.z; acc[0].w += e.w;
                acc[1].x = fmaf(e.x, e.x, acc[1].x); acc[1].y = fmaf(e.y, e.y, acc[1].y);
                acc[1].z = fmaf(e.z, e.z, acc[1].z); acc[1].w = fmaf(e.w, e.w, acc[1].w);
            }
        }
        slot_reduce<2>(acc, sred, tid, slot, ct, TPR, RB);
        if (slot == 0 && cg < C4) {
            float* pp = partials + ((long)blockIdx.x * C + c) * 2;
            pp[0] = acc[0].x; pp[1] = acc[1].x; pp[2] = acc[0].y; pp[3] = acc[1].y;
            pp[4] = acc[0].z; pp[5] = acc[1].z; pp[6] = acc[0].w; pp[7] = acc[1].w;
        }
    }
}

// Accumulators are indexed [f][tap] with compile-time bounds (F_in <= 2 features, k0 <= 8 taps) so they stay in registers and
// no integer division by the runtime filter width runs per row.
constexpr int XF = 2, XT = 8;
// Stage 1: every block reduces its share of the rows into ws[blk][kk][C] (kk = f*k0 + tap for G, kk = F_in*k0 for S): no
// atomics, two rows in flight per thread.  Stage 2 (expand_bwd_finish_kernel) sums the block rows and applies the
// parameter-sized epilogue.  (One pass with <=128 long-running blocks ending in (F_in*k0+1)*C atomics took 115 us for
// 14 MB of dE: 8 bytes in flight per thread.)
// BN (round 5): dE arrives BEFORE the backward of expand_bn (the masked gradient of the first block's input GEMM) and the kernel applies
// dz = ka*dE + kb*E + kc while loading -- the stand-alone gast_bn_bwd_apply pass over dE disappears.  Same fma nesting and rounding as
// bn_bwd_apply_kernel.
template <typename T, bool BN = false>
__global__ void __launch_bounds__(256) expand_bwd_kernel(const T* __restrict__ dE, int ldde, const float* __restrict__ x, int B, int T_in,
                                                         int J, int F_in, int k0, int t_stride, int T_out,
                                                         const float* __restrict__ mean0, const float* __restrict__ rstd0, int C,
                                                         float* __restrict__ ws, int TPR, int RB,
                                                         const T* __restrict__ Epre, int lde, const float* ka, const float* kb, const float* kc) {
    __shared__ float sred[256][4];
    const int tid = threadIdx.x, slot = tid / TPR, ct = tid - slot * TPR;
    const int K0 = F_in * k0;
    const int C4 = C >> 2;
    const long rows = (long)B * T_out * J;
    const int TJ = T_out * J;
    const long step = (long)gridDim.x * RB;
    float mu[XF], rs[XF];
#pragma unroll
    for (int f = 0; f < XF; ++f) { mu[f] = f < F_in ? mean0[f] : 0.f; rs[f] = f < F_in ? rstd0[f] : 0.f; }
    float* wsb = ws + (long)blockIdx.x * (K0 + 1) * C;
    for (int cg0 = 0; cg0 < C4; cg0 += TPR) {
        const int cg = cg0 + ct;
        const int c = cg * 4;
        float4 g[XF][XT], gs = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < XF; ++f)
#pragma unroll
            for (int tap = 0; tap < XT; ++tap) g[f][tap] = make_float4(0, 0, 0, 0);
        if (slot < RB && cg < C4) {
            float4 ca = f4(1.f), cb = f4(0.f), ck = f4(0.f);
            if (BN) { ca = *(const float4*)(ka + c); cb = *(const float4*)(kb + c); ck = *(const float4*)(kc + c); }
            for (long r0 = (long)blockIdx.x * RB + slot; r0 < rows; r0 += 2 * step) {
                const long r1 = r0 + step;
                const bool two = r1 < rows;
                float4 d0 = ld4(dE + r0 * ldde + c);
                float4 d1 = two ? ld4(dE + r1 * ldde + c) : make_float4(0, 0, 0, 0);
                if (BN) {
                    const float4 e0 = ld4(Epre + r0 * lde + c);
                    const float4 e1 = two ? ld4(Epre + r1 * lde + c) : make_float4(0, 0, 0, 0);
                    d0.x = fmaf(ca.x, d0.x, fmaf(cb.x, e0.x, ck.x)); d0.y = fmaf(ca.y, d0.y, fmaf(cb.y, e0.y, ck.y));
                    d0.z = fmaf(ca.z, d0.z, fmaf(cb.z, e0.z, ck.z)); d0.w = fmaf(ca.w, d0.w, fmaf(cb.w, e0.w, ck.w));
                    d1.x = fmaf(ca.x, d1.x, fmaf(cb.x, e1.x, ck.x)); d1.y = fmaf(ca.y, d1.y, fmaf(cb.y, e1.y, ck.y));
                    d1.z = fmaf(ca.z, d1.z, fmaf(cb.z, e1.z, ck.z)); d1.w = fmaf(ca.w, d1.w, fmaf(cb.w, e1.w, ck.w));
                    d0 = rnd4(d0, (const T*)nullptr);
                    d1 = rnd4(d1, (const T*)nullptr);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
                    const float4 d = u ? d1 : d0;
                    const int m = (int)(u ? r1 : r0);
                    const int b = m / TJ, rem = m - b * TJ;
                    const int t = rem / J, j = rem - t * J;
                    gs.x += d.x; gs.y += d.y; gs.z += d.z; gs.w += d.w;
                    const float* xb = x + (((long)b * T_in + t * t_stride) * J + j) * F_in;
#pragma unroll
                    for (int tap = 0; tap < XT; ++tap) {
                        if (tap < k0) {
#pragma unroll
                            for (int f = 0; f < XF; ++f) {
                                if (f < F_in) {
                                    const float xh = (xb[(long)tap * J * F_in + f] - mu[f]) * rs[f];
                                    g[f][tap].x = fmaf(d.x, xh, g[f][tap].x); g[f][tap].y = fmaf(d.y, xh, g[f][tap].y);
                                    g[f][tap].z = fmaf(d.z, xh, g[f][tap].z); g[f][tap].w = fmaf(d.w, xh, g[f][tap].w);
                                }
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < XF; ++f)
#pragma unroll
            for (int tap = 0; tap < XT; ++tap) {
                if (f < F_in && tap < k0) {
                    float4 v[1] = {g[f][tap]};
                    slot_reduce<1>(v, sred, tid, slot, ct, TPR, RB);
                    if (slot == 0 && cg < C4) *(float4*)(wsb + (long)(f * k0 + tap) * C + c) = v[0];
                }
            }
        {
            float4 v[1] = {gs};
            slot_reduce<1>(v, sred, tid, slot, ct, TPR, RB);
            if (slot == 0 && cg < C4) *(float4*)(wsb + (long)K0 * C + c) = v[0];
        }
    }
}

// Stage 2: grid = (channel groups of 8, F_in*k0).  Block (cg, kk) sums G[c][kk] = sum_blk ws[blk][kk][c] and S[c] (column K0) over
// the block rows with 256 threads = 8 channels x 32 block-row lanes, then
//   dW[c][f][tap] (+)= gamma0[f] * G + beta0[f] * S[c]                    (xn = gamma0 * xhat + beta0 feeds the expand conv)
//   dgamma0[f]   += sum_c W[c][f][tap] * G[c][kk];   dbeta0[f] += sum_c W[c][f][tap] * S[c]          (atomics, kk = f*k0 + tap)
constexpr int EF_CH = 8, EF_LANES = 32;
__global__ void __launch_bounds__(256) expand_bwd_finish_kernel(const float* __restrict__ ws, int nb, int C, int F_in, int k0,
                                                                const float* __restrict__ W, const float* __restrict__ gamma0,
                                                                const float* __restrict__ beta0, float* __restrict__ dW,
                                                                float* __restrict__ dgamma0, float* __restrict__ dbeta0, int accumulate, int det) {
    __shared__ float sred[EF_LANES][EF_CH][2];
    __shared__ float sgb[EF_CH][2];
    const int cx = threadIdx.x % EF_CH, ry = threadIdx.x / EF_CH;
    const int K0 = F_in * k0, K1 = K0 + 1;
    // default grid: (channel groups of EF_CH, F_in * k0) -- one (group, kk) per block, dgamma0 / dbeta0 by atomics.
    // GAST_DETERMINISTIC grid: (1, F_in) -- the block walks the taps and channel groups of its input feature in order and adds ONCE.
    const int ngroups = (C + EF_CH - 1) / EF_CH;
    const int kk_lo = det ? blockIdx.y * k0 : blockIdx.y, kk_hi = det ? kk_lo + k0 : kk_lo + 1;
    const int g_lo = det ? 0 : blockIdx.x, g_hi = det ? ngroups : g_lo + 1;
    float tot = 0.f;               // (threads 0 / 1: the running dgamma0 / dbeta0 contribution of this block)
    int f = 0;
    for (int kk = kk_lo; kk < kk_hi; ++kk) {
        f = kk / k0;
        const int tap = kk - f * k0;
        for (int grp = g_lo; grp < g_hi; ++grp) {
            const int c = grp * EF_CH + cx;
            float g = 0.f, sv = 0.f;
            if (c < C) {
#pragma unroll 4
                for (int blk = ry; blk < nb; blk += EF_LANES) {
                    const float* p = ws + (long)blk * K1 * C + c;
                    g += p[(long)kk * C];
                    sv += p[(long)K0 * C];
                }
            }
            __syncthreads();       // (the previous round's readers are done with sred / sgb)
            sred[ry][cx][0] = g;
            sred[ry][cx][1] = sv;
            __syncthreads();
            if (threadIdx.x < EF_CH) {
                const int ch = threadIdx.x, cc = grp * EF_CH + ch;
                float Gv = 0.f, Sc = 0.f;
#pragma unroll 8
                for (int r = 0; r < EF_LANES; ++r) { Gv += sred[r][ch][0]; Sc += sred[r][ch][1]; }
                float dg = 0.f, db = 0.f;
                if (cc < C) {
                    const long o = ((long)cc * F_in + f) * k0 + tap;
                    const float w = W[o];
                    const float v = gamma0[f] * Gv + beta0[f] * Sc;
                    if (accumulate) dW[o] += v; else dW[o] = v;
                    dg = w * Gv;
                    db = w * Sc;
                }
                sgb[ch][0] = dg;
                sgb[ch][1] = db;
            }
            __syncthreads();
            if (threadIdx.x < 2) {
                float v = 0.f;
                for (int i = 0; i < EF_CH; ++i) v += sgb[i][threadIdx.x];
                tot += v;
            }
        }
    }
    if (threadIdx.x < 2) atomicAdd(threadIdx.x ? dbeta0 + f : dgamma0 + f, tot);
}

}  // namespace


extern "C" int gast_rowwise_blocks(long rows, int N) { return row_blocks(rows, N); }

extern "C" int gast_bn_finalize_multi(const gast_bn_fin_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_BN_MAX_BATCH) return GAST_EINVAL;
    BnFinBatch b;
    int maxN = 0;
    for (int d = 0; d < n; ++d) {
        const gast_bn_fin_job& j = jobs[d];
        if (!j.partials || !j.gamma || !j.beta || !j.scale || !j.shift || !j.mean || !j.rstd || j.N < 1 || j.nblk < 1 || j.count <= 0)
            return GAST_EINVAL;
        if ((j.running_mean == nullptr) != (j.running_var == nullptr)) return GAST_EINVAL;
        b.j[d] = j;
        if (j.N > maxN) maxN = j.N;
    }
    hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3((maxN + FINS_COLS - 1) / FINS_COLS, n), dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_finalize(const float* partials, int nblk, int ncol_total, int col0, int N, double count,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* rstd, int centered, gast_stream_t stream) {
    gast_bn_fin_job j = {partials, nblk, ncol_total, col0, N, count, gamma, beta, running_mean, running_var, num_batches_tracked,
                         momentum, eps, scale, shift, mean, rstd, centered};
    return gast_bn_finalize_multi(&j, 1, stream);
}

extern "C" int gast_bn_eval_multi(const gast_bn_eval_job* jobs, int n, float eps, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_BN_EVAL_MAX_BATCH) return GAST_EINVAL;
    BnEvalBatch b;
    int maxN = 0;
    for (int d = 0; d < n; ++d) {
        if (!jobs[d].gamma || !jobs[d].beta || !jobs[d].running_mean || !jobs[d].running_var || !jobs[d].scale || !jobs[d].shift ||
            jobs[d].N < 1)
            return GAST_EINVAL;
        b.j[d] = jobs[d];
        if (jobs[d].N > maxN) maxN = jobs[d].N;
    }
    hipLaunchKernelGGL(bn_eval_multi_kernel, dim3((maxN + 127) / 128, n), dim3(128), 0, (hipStream_t)stream, b, eps);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                            float eps, int N, float* scale, float* shift, int centered, gast_stream_t stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || N < 1) return GAST_EINVAL;
    hipLaunchKernelGGL(bn_eval_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var,
                       eps, N, scale, shift, centered);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_bwd_finalize_multi(const gast_bn_bwd_fin_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_BN_MAX_BATCH) return GAST_EINVAL;
    BnBwdFinBatch b;
    int maxN = 0;
    for (int d = 0; d < n; ++d) {
        const gast_bn_bwd_fin_job& j = jobs[d];
        if (!j.partials || !j.gamma || !j.mean || !j.rstd || !j.dgamma || !j.dbeta || !j.ka || !j.kb || !j.kc || j.N < 1 || j.nblk < 1 ||
            j.count <= 0)
            return GAST_EINVAL;
        b.j[d] = j;
        if (j.N > maxN) maxN = j.N;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_multi_kernel, dim3((maxN + FINS_COLS - 1) / FINS_COLS, n), dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_bwd_fused_multi(int dtype, const gast_bn_bwd_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_BN_MAX_BATCH || (dtype != GAST_F32 && dtype != GAST_BF16)) return GAST_EINVAL;
    BnBwdFusedBatch b;
    int maxN = 0;
    long maxRows = 0;
    for (int d = 0; d < n; ++d) {
        const gast_bn_bwd_job& j = jobs[d];
        if (j.rows > maxRows) maxRows = j.rows;
        if (!j.f.partials || !j.f.gamma || !j.f.mean || !j.f.rstd || !j.f.dgamma || !j.f.dbeta || !j.dz || !j.X || j.f.N < 1 ||
            j.f.nblk < 1 || j.f.count <= 0 || j.rows < 1)
            return GAST_EINVAL;
        if (j.f.N % 4 || j.lddz % 4 || j.ldx % 4) return GAST_EALIGN;
        b.j[d] = j;
        if (j.f.N > maxN) maxN = j.f.N;
    }
    dim3 grid((maxN + FIN_COLS - 1) / FIN_COLS, n, (unsigned)((maxRows + FUSED_ROWS - 1) / FUSED_ROWS));
    if (dtype == GAST_F32) hipLaunchKernelGGL((bn_bwd_fused_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL((bn_bwd_fused_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_bwd_finalize(const float* partials, int nblk, int ncol_total, int col0, int N, double count,
                                    const float* gamma, const float* mean, const float* rstd,
                                    float* dgamma, float* dbeta, float* ka, float* kb, float* kc, gast_stream_t stream) {
    gast_bn_bwd_fin_job j = {partials, nblk, ncol_total, col0, N, count, gamma, mean, rstd, dgamma, dbeta, ka, kb, kc, 0};
    return gast_bn_bwd_finalize_multi(&j, 1, stream);
}

static inline bool bad_dtype(int d) { return d != GAST_F32 && d != GAST_BF16; }

extern "C" int gast_bn_bwd_apply(int dtype, void* dz, int lddz, const void* X, int ldx, long rows, int N,
                                 const float* ka, const float* kb, const float* kc, gast_stream_t stream) {
    if (bad_dtype(dtype) || !dz || !X || !ka || !kb || !kc || rows < 1) return GAST_EINVAL;
    if (N % 4 || lddz % 4 || ldx % 4) return GAST_EALIGN;
    RowCfg c = row_cfg(N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (float*)dz, lddz, (const float*)X, ldx,
                           rows, N, ka, kb, kc, c.TPR, c.RB);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (bf16_t*)dz, lddz, (const bf16_t*)X,
                           ldx, rows, N, ka, kb, kc, c.TPR, c.RB);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bn_bwd_apply_frames(int dtype, void* dz, int lddz, const void* X, int ldx, long rows, int N, const float* ka,
                                        const float* kb, const float* kc, int T_total, int J, unsigned long long frames, gast_stream_t stream) {
    if (bad_dtype(dtype) || !dz || !X || !ka || !kb || !kc || rows < 1 || T_total < 1 || T_total > 64 || J < 1 || rows % ((long)T_total * J))
        return GAST_EINVAL;
    if (N % 4 || lddz % 4 || ldx % 4) return GAST_EALIGN;
    RowCfg c = row_cfg(N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (float*)dz, lddz, (const float*)X, ldx,
                           rows, N, ka, kb, kc, c.TPR, c.RB, T_total, J, frames);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (bf16_t*)dz, lddz, (const bf16_t*)X,
                           ldx, rows, N, ka, kb, kc, c.TPR, c.RB, T_total, J, frames);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bnrelu_apply(int dtype, const void* X, int ldx, long rows, int N, const float* scale, const float* shift,
                                 void* Y, int ldy, int use_drop, uint32_t salt, gast_dropout drop, gast_stream_t stream) {
    if (bad_dtype(dtype) || !X || !Y || !scale || !shift || rows < 1) return GAST_EINVAL;
    if (N % 4 || ldx % 4 || ldy % 4) return GAST_EALIGN;
    RowCfg c = row_cfg(N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((bnrelu_apply_kernel<float>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const float*)X, ldx, rows, N, scale,
                           shift, (float*)Y, ldy, use_drop, salt, drop, c.TPR, c.RB);
    else
        hipLaunchKernelGGL((bnrelu_apply_kernel<bf16_t>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const bf16_t*)X, ldx, rows, N, scale,
                           shift, (bf16_t*)Y, ldy, use_drop, salt, drop, c.TPR, c.RB);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_shrink_fwd(int dtype, const void* O, int ldo, long rows, int K, const float* scale, const float* shift, const void* W,
                               int ldw, int D, float* pred, int ldp, gast_stream_t stream) {
    if (bad_dtype(dtype) || !O || !scale || !shift || !W || !pred || rows < 1 || K < 4 || D < 1 || D > SHRINK_MAXD || ldp < D) return GAST_EINVAL;
    const int epc = dtype == GAST_F32 ? 4 : 8;       // elements per 16 bytes
    if (K % 4 || ldo % epc || ldw % epc || (((uintptr_t)O) & 15) || (((uintptr_t)W) & 15) || (((uintptr_t)scale) & 15) || (((uintptr_t)shift) & 15))
        return GAST_EALIGN;
    const unsigned grid = (unsigned)((rows + 7) / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((shrink_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)O, ldo, rows, K, scale, shift, (const float*)W,
                           ldw, D, pred, ldp);
    else
        hipLaunchKernelGGL((shrink_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)O, ldo, rows, K, scale, shift,
                           (const bf16_t*)W, ldw, D, pred, ldp);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_shrink_bwd_blocks(long rows) { return (int)((rows + SHRINK_ROWS - 1) / SHRINK_ROWS); }

extern "C" int gast_shrink_bwd(int dtype, const void* dp, int lddp, const void* W, int ldw, int D, const void* O, int ldo, const float* scale,
                               const float* shift, long rows, int K, void* dO, int lddo, float* partials, gast_stream_t stream) {
    if (bad_dtype(dtype) || !dp || !W || !O || !scale || !shift || !dO || !partials || rows < 1 || K < 4 || D < 1 || D > SHRINK_MAXD || lddp < D)
        return GAST_EINVAL;
    const int epc = dtype == GAST_F32 ? 4 : 8;
    if (K % 4 || ldo % epc || lddo % epc || ldw % epc || (((uintptr_t)O) & 15) || (((uintptr_t)dO) & 15) || (((uintptr_t)W) & 15) ||
        (((uintptr_t)scale) & 15) || (((uintptr_t)shift) & 15) || (((uintptr_t)partials) & 15))
        return GAST_EALIGN;
    const dim3 grid((unsigned)gast_shrink_bwd_blocks(rows), (unsigned)((K + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((shrink_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dp, lddp, (const float*)W, ldw, D, (const float*)O, ldo,
                           scale, shift, rows, K, (float*)dO, lddo, partials);
    else
        hipLaunchKernelGGL((shrink_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dp, lddp, (const bf16_t*)W, ldw, D, (const bf16_t*)O,
                           ldo, scale, shift, rows, K, (bf16_t*)dO, lddo, partials);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_bnrelu_bwd_mask(int dtype, const void* dY, int lddy, const void* X, int ldx, long rows, int N,
                                    const float* scale, const float* shift, int use_drop, uint32_t salt, gast_dropout drop,
                                    void* dz, int lddz, float* partials, gast_stream_t stream) {
    if (bad_dtype(dtype) || !dY || !X || !scale || !shift || !dz || !partials || rows < 1) return GAST_EINVAL;
    if (N % 4 || lddy % 4 || ldx % 4 || lddz % 4) return GAST_EALIGN;
    RowCfg c = row_cfg(N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((bnrelu_bwd_mask_kernel<float>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const float*)dY, lddy,
                           (const float*)X, ldx, rows, N, scale, shift, use_drop, salt, drop, (float*)dz, lddz, partials, c.TPR, c.RB);
    else
        hipLaunchKernelGGL((bnrelu_bwd_mask_kernel<bf16_t>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const bf16_t*)dY, lddy,
                           (const bf16_t*)X, ldx, rows, N, scale, shift, use_drop, salt, drop, (bf16_t*)dz, lddz, partials, c.TPR, c.RB);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_residual_fwd(int dtype, const void* O, int ldo, gast_rowmap omap, const float* scO, const float* shO,
                                 const void* T2, int ldt, const float* sc2, const float* sh2,
                                 int use_drop, uint32_t salt, gast_dropout drop,
                                 int B, int Tn, int J, int N, void* Xn, int ldxn, gast_stream_t stream) {
    if (bad_dtype(dtype) || !O || !scO || !shO || !T2 || !sc2 || !sh2 || !Xn || B < 1 || Tn < 1 || J < 1) return GAST_EINVAL;
    if (N % 4 || ldo % 4 || ldt % 4 || ldxn % 4) return GAST_EALIGN;
    long rows = (long)B * Tn * J;
    RowCfg c = row_cfg(N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((residual_fwd_kernel<float>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const float*)O, ldo, omap, scO, shO,
                           (const float*)T2, ldt, sc2, sh2, use_drop, salt, drop, Tn, J, rows, N, (float*)Xn, ldxn, c.TPR, c.RB);
    else
        hipLaunchKernelGGL((residual_fwd_kernel<bf16_t>), dim3(row_blocks(rows, N)), dim3(256), 0, st, (const bf16_t*)O, ldo, omap, scO, shO,
                           (const bf16_t*)T2, ldt, sc2, sh2, use_drop, salt, drop, Tn, J, rows, N, (bf16_t*)Xn, ldxn, c.TPR, c.RB);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_colsum(int dtype, const void* X, int ldx, long rows, int N, float* out, int zero_first, gast_stream_t stream) {
    if (bad_dtype(dtype) || !X || !out || rows < 1) return GAST_EINVAL;
    if (N % 4 || ldx % 4) return GAST_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (zero_first) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    RowCfg c = row_cfg(N);
    int nb = row_blocks(rows, N);
    if (nb > 1024) nb = 1024;
    if (gast_deterministic()) nb = 1;      // one block: one add per column into the zeroed / running destination
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((colsum_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)X, ldx, rows, N, out, c.TPR, c.RB);
    else
        hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)X, ldx, rows, N, out, c.TPR, c.RB);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_input_stats_blocks(long rows) { return (int)((rows + IN_ROWS_PER_BLOCK - 1) / IN_ROWS_PER_BLOCK); }

extern "C" int gast_input_stats(const float* x, long rows, int F_in, float* partials, int* nblk_out, gast_stream_t stream) {
    if (!x || !partials || rows < 1 || F_in < 1 || F_in > 8) return GAST_EINVAL;
    int nb = gast_input_stats_blocks(rows);
    if (nblk_out) *nblk_out = nb;
    hipLaunchKernelGGL(input_stats_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, rows, F_in, partials);
    GAST_CHECK_LAUNCH();
    return 0;
}

static inline int conv_t_out(int T_in, int k0, int t_stride) { return (T_in - k0) / t_stride + 1; }

extern "C" int gast_expand_fwd(int dtype, const float* x, int B, int T_in, int J, int F_in, int k0, int t_stride,
                               const float* W, const float* sc0, const float* sh0, int C,
                               void* E, int lde, float* partials, const float* center, gast_stream_t stream) {
    if (bad_dtype(dtype) || !x || !W || !sc0 || !sh0 || !E || !partials) return GAST_EINVAL;
    if (F_in < 1 || k0 < 1 || F_in * k0 > KMAX || t_stride < 1 || T_in < k0 || B < 1 || J < 1) return GAST_ERANGE;
    if (C % 4 || lde % 4) return GAST_EALIGN;
    size_t smem = (size_t)F_in * k0 * C * sizeof(float);
    if (smem > 96 * 1024) return GAST_ERANGE;
    int T_out = conv_t_out(T_in, k0, t_stride);
    long rows = (long)B * T_out * J;
    RowCfg c = row_cfg(C);
    hipStream_t st = (hipStream_t)stream;
    int nb = row_blocks(rows, C);
    if (smem > 48 * 1024) {
        hipError_t e = dtype == GAST_F32
            ? hipFuncSetAttribute((const void*)expand_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
            : hipFuncSetAttribute((const void*)expand_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((expand_fwd_kernel<float>), dim3(nb), dim3(256), smem, st, x, B, T_in, J, F_in, k0, t_stride, T_out, W, sc0, sh0,
                           C, (float*)E, lde, partials, c.TPR, c.RB, center);
    else
        hipLaunchKernelGGL((expand_fwd_kernel<bf16_t>), dim3(nb), dim3(256), smem, st, x, B, T_in, J, F_in, k0, t_stride, T_out, W, sc0,
                           sh0, C, (bf16_t*)E, lde, partials, c.TPR, c.RB, center);
    GAST_CHECK_LAUNCH();
    return 0;
}

static int expand_bwd_blocks(long rows, int C) {
    int nb = row_blocks(rows, C) / 2;       // two rows in flight per thread
    if (nb > 1024) nb = 1024;
    return nb < 1 ? 1 : nb;
}

extern "C" long gast_expand_bwd_ws_floats(long rows, int C, int F_in, int k0) {
    return (long)expand_bwd_blocks(rows, C) * (F_in * k0 + 1) * C;
}

extern "C" int gast_expand_bwd(int dtype, const void* dE, int ldde, const float* x, int B, int T_in, int J, int F_in, int k0,
                               int t_stride, const float* mean0, const float* rstd0, int C, const float* W, const float* gamma0,
                               const float* beta0, float* dW, float* dgamma0, float* dbeta0, float* ws, int accumulate,
                               gast_stream_t stream) {
    return gast_expand_bwd_bn(dtype, dE, ldde, nullptr, 0, nullptr, nullptr, nullptr, x, B, T_in, J, F_in, k0, t_stride, mean0, rstd0, C, W,
                              gamma0, beta0, dW, dgamma0, dbeta0, ws, accumulate, stream);
}
extern "C" int gast_expand_bwd_bn(int dtype, const void* dE, int ldde, const void* Epre, int lde, const float* ka, const float* kb,
                                  const float* kc, const float* x, int B, int T_in, int J, int F_in, int k0,
                                  int t_stride, const float* mean0, const float* rstd0, int C, const float* W, const float* gamma0,
                                  const float* beta0, float* dW, float* dgamma0, float* dbeta0, float* ws, int accumulate,
                                  gast_stream_t stream) {
    const bool bn = Epre != nullptr;
    if (bn && (!ka || !kb || !kc || lde % 4)) return GAST_EINVAL;
    if (bad_dtype(dtype) || !dE || !x || !mean0 || !rstd0 || !W || !gamma0 || !beta0 || !dW || !dgamma0 || !dbeta0 || !ws)
        return GAST_EINVAL;
    if (F_in < 1 || k0 < 1 || F_in > XF || k0 > XT || t_stride < 1 || T_in < k0 || B < 1 || J < 1) return GAST_ERANGE;
    if (C % 4 || ldde % 4) return GAST_EALIGN;
    int T_out = conv_t_out(T_in, k0, t_stride);
    long rows = (long)B * T_out * J;
    RowCfg c = row_cfg(C);
    hipStream_t st = (hipStream_t)stream;
    const int nb = expand_bwd_blocks(rows, C);
    if (dtype == GAST_F32)
        if (bn)
            hipLaunchKernelGGL((expand_bwd_kernel<float, true>), dim3(nb), dim3(256), 0, st, (const float*)dE, ldde, x, B, T_in, J, F_in, k0, t_stride,
                               T_out, mean0, rstd0, C, ws, c.TPR, c.RB, (const float*)Epre, lde, ka, kb, kc);
        else
            hipLaunchKernelGGL((expand_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)dE, ldde, x, B, T_in, J, F_in, k0, t_stride,
                               T_out, mean0, rstd0, C, ws, c.TPR, c.RB, (const float*)nullptr, 0, nullptr, nullptr, nullptr);
    else if (bn)
        hipLaunchKernelGGL((expand_bwd_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dE, ldde, x, B, T_in, J, F_in, k0,
                           t_stride, T_out, mean0, rstd0, C, ws, c.TPR, c.RB, (const bf16_t*)Epre, lde, ka, kb, kc);
    else
        hipLaunchKernelGGL((expand_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dE, ldde, x, B, T_in, J, F_in, k0,
                           t_stride, T_out, mean0, rstd0, C, ws, c.TPR, c.RB, (const bf16_t*)nullptr, 0, nullptr, nullptr, nullptr);
    GAST_CHECK_LAUNCH();
    // (GAST_DETERMINISTIC: one block per input feature walks its taps and channel groups in order -- one add per address)
    const dim3 fgrid = gast_deterministic() ? dim3(1, F_in) : dim3((C + EF_CH - 1) / EF_CH, F_in * k0);
    hipLaunchKernelGGL(expand_bwd_finish_kernel, fgrid, dim3(256), 0, st, ws, nb, C, F_in, k0, W, gamma0, beta0, dW,
                       dgamma0, dbeta0, accumulate, gast_deterministic() ? 1 : 0);
    GAST_CHECK_LAUNCH();
    return 0;
}

#ifdef GAST_H16_F16
extern "C" const char* gast_version(void) { return "gast_hip 0.2 gfx950 (16-bit storage: IEEE binary16)"; }
#else
extern "C" const char* gast_version(void) { return "gast_hip 0.2 gfx950 (16-bit storage: bfloat16)"; }
#endif
