// BatchNorm finalize as device functions (the bodies of the stand-alone finalize kernels of norm_ops.hip).
// Reference: nn.BatchNorm2d(momentum=0.1) in train mode, model/gast_net.py:20,58-59,147,149,
// model/local_attention.py:117-123, model/global_attention.py:95, and its autograd backward.
#pragma once
#include "common.h"

namespace gastbn {

// Round 3 geometry of the stand-alone finalizes: a 256-thread block owns 4 columns; 64 lanes per column sum every 64th partial row (all
// loads of a lane in flight at once), the 16 lanes of a column inside a wave are combined with shuffles, the four waves through LDS --
// always in the same order: the statistics are deterministic.
constexpr int FINS_COLS = 4, FINS_LANES = 64;
typedef double (*fin_red_t)[FINS_COLS][2];          // [4 waves][FINS_COLS][2]

// thread = (column cx = tid & 3, lane ry = tid >> 2) of the first 256 threads of the block; every thread of the block must call
__device__ __forceinline__ void finalize_sums_wide(const float* __restrict__ partials, int nblk, int ncol_total, int col, bool valid,
                                                   fin_red_t sred, int cx, int ry, double& s1, double& s2) {
    double a1 = 0.0, a2 = 0.0;
    if (valid) {
#pragma unroll 8
        for (int b = ry; b < nblk; b += FINS_LANES) {
            const float2 p = *(const float2*)(partials + ((long)b * ncol_total + col) * 2);
            a1 += (double)p.x;
            a2 += (double)p.y;
        }
    }
#pragma unroll
    for (int o = FINS_COLS; o < 64; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }      // lanes of equal cx inside the wave
    const int w = threadIdx.x >> 6;
    if (w < 4 && (threadIdx.x & 63) < FINS_COLS) { sred[w][cx][0] = a1; sred[w][cx][1] = a2; }
    __syncthreads();
    s1 = (sred[0][cx][0] + sred[1][cx][0]) + (sred[2][cx][0] + sred[3][cx][0]);
    s2 = (sred[0][cx][1] + sred[1][cx][1]) + (sred[2][cx][1] + sred[3][cx][1]);
}

// columns cb*4 .. cb*4+3 of one forward finalize: scale / shift / mean / rstd and the running statistics (nn.BatchNorm2d, train mode).
// Called by every thread of the block (blocks of 256 or 512 threads; the first 256 work); ends with a barrier: sred may be reused.
__device__ __forceinline__ void bn_finalize_unit(const gast_bn_fin_job& j, int cb, fin_red_t sred) {
    const int N = j.N;
    const bool act = threadIdx.x < 256;
    const int cx = threadIdx.x & (FINS_COLS - 1), ry = (threadIdx.x & 255) / FINS_COLS;
    const int n = cb * FINS_COLS + cx;
    double s1, s2;
    finalize_sums_wide(j.partials, j.nblk, j.ncol_total, j.col0 + n, act && n < N, sred, cx, ry, s1, s2);
    if (act && ry == 0 && n < N) {
        // Every multiply-add below is an EXPLICIT fma: left to the compiler's contraction pass `a*b + c*d` rounds differently from one
        // inlining context to the next, and the statistics must not depend on which kernel inlines this.
        const double count = j.count;
        const double mean = s1 / count;
        double var = fma(-mean, mean, s2 / count);
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)j.eps));
        const float sc = j.gamma[n] * rstd;
        const float meanf = (float)mean;
        j.scale[n] = sc;
        j.shift[n] = fmaf(-meanf, sc, j.beta[n]);
        j.mean[n] = meanf;
        j.rstd[n] = rstd;
        if (j.running_mean) {
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            const float momentum = j.momentum, keep = 1.f - momentum;
            // centred storage: the statistics are those of x - running_mean, so the true batch mean is running_mean + mean
            j.running_mean[n] = j.centered ? fmaf(momentum, meanf, j.running_mean[n]) : fmaf(momentum, meanf, keep * j.running_mean[n]);
            j.running_var[n] = fmaf(momentum, (float)unb, keep * j.running_var[n]);
        }
        if (n == 0 && j.num_batches_tracked) *j.num_batches_tracked += 1;
    }
    __syncthreads();
}

// columns cb*4 .. cb*4+3 of one backward finalize: dgamma / dbeta and the coefficients of dx = ka*dz + kb*x + kc
__device__ __forceinline__ void bn_bwd_finalize_unit(const gast_bn_bwd_fin_job& j, int cb, fin_red_t sred) {
    const int N = j.N;
    const bool act = threadIdx.x < 256;
    const int cx = threadIdx.x & (FINS_COLS - 1), ry = (threadIdx.x & 255) / FINS_COLS;
    const int n = cb * FINS_COLS + cx;
    double s1, s2;
    finalize_sums_wide(j.partials, j.nblk, j.ncol_total, j.col0 + n, act && n < N, sred, cx, ry, s1, s2);
    if (act && ry == 0 && n < N) {
        const double mu = j.mean[n], r = j.rstd[n], g = j.gamma[n], count = j.count;
        const double dg = r * fma(-mu, s1, s2);   // sum dz * xhat   (explicit fma: see bn_finalize_unit)
        const double db = s1;
        if (j.accumulate) { j.dgamma[n] += (float)dg; j.dbeta[n] += (float)db; }      // gradient destinations: plain read-modify-write, one owner per element
        else { j.dgamma[n] = (float)dg; j.dbeta[n] = (float)db; }
        const double a = g * r;
        const double b = -(g * r) * r * dg / count;
        j.ka[n] = (float)a;
        j.kb[n] = (float)b;
        j.kc[n] = (float)fma(-b, mu, -(a * db / count));
    }
    __syncthreads();
}

}  // namespace gastbn
