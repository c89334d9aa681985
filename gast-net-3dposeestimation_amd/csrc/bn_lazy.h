// BatchNorm finalize as device functions, shared by the stand-alone finalize kernels (norm_ops.hip) and by every launch that runs a
// finalize LAZILY in front of its own work (gast_bn_lazy, include/gast_hip.h): the streaming consumers of norm_ops.hip and the GEMM
// kernels of gemm.hip / gemm_big.hip.  Reference: nn.BatchNorm2d(momentum=0.1) in train mode, model/gast_net.py:20,58-59,147,149,
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
        // Every multiply-add below is an EXPLICIT fma: this code is inlined into many kernels, and left to the compiler's contraction
        // pass `a*b + c*d` rounds differently from one inlining context to the next -- the lazy finalize must reproduce the stand-alone
        // launch bit for bit.
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

// ---- lazy finalize (gast_bn_lazy): called by EVERY thread of EVERY block of a 1-D grid before the first read of a coefficient.
// The first nfin blocks run the finalize units (4 columns each; more than one per block when the grid is smaller than the unit
// count) and publish: results written back at device scope, then ONE word per block, flag[block] = 1.  Every block waits until all
// nfin words are set.  NO shared counter: same-address atomics serialise at 50 - 170 ns apiece on this chip, so counting 128 blocks on
// one address took longer than the finalize launch it replaces (measured: +12 us per lazy launch); nfin independent stores do not queue.
// lazy_red: 256 bytes of LDS scratch ([4][FINS_COLS][2] doubles) nobody else touches during the call.
constexpr int LAZY_MAX_FIN = 256;       // finalizing blocks (= words of lz.flag that must arrive zeroed: GAST_BN_LAZY_FLAG_WORDS)
__device__ __forceinline__ void bn_lazy_sync_with(const gast_bn_lazy& lz, fin_red_t lazy_red) {
    if (lz.kind == 0) return;
    const int N0 = lz.kind == GAST_BN_LAZY_FWD ? lz.fwd[0].N : lz.bwd[0].N;
    const int N1 = lz.n > 1 ? (lz.kind == GAST_BN_LAZY_FWD ? lz.fwd[1].N : lz.bwd[1].N) : 0;
    const int u0 = (N0 + FINS_COLS - 1) / FINS_COLS, U = u0 + (N1 + FINS_COLS - 1) / FINS_COLS;
    int nfin = U < (int)gridDim.x ? U : (int)gridDim.x;
    if (nfin > LAZY_MAX_FIN) nfin = LAZY_MAX_FIN;
    if ((int)blockIdx.x < nfin) {
        // (one loop per job, each with a compile-time job index: a run-time index into the by-value argument -- also the one the
        //  compiler re-creates by merging two branches of ONE loop -- parks the jobs' `count` fields in scratch memory, and a kernel
        //  that uses scratch pays for its set-up at every dispatch)
        int u = blockIdx.x;
        for (; u < u0; u += nfin) {
            if (lz.kind == GAST_BN_LAZY_FWD) bn_finalize_unit(lz.fwd[0], u, lazy_red);
            else bn_bwd_finalize_unit(lz.bwd[0], u, lazy_red);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (; u < U; u += nfin) {
            if (lz.kind == GAST_BN_LAZY_FWD) bn_finalize_unit(lz.fwd[1], u - u0, lazy_red);
            else bn_bwd_finalize_unit(lz.bwd[1], u - u0, lazy_red);
        }
        // publish: the coefficient writers are threads 0..3 (ry == 0), i.e. wave 0 -- its device-scope release writes the results back
        // from this XCD's L2 before the block's word is set (a device-scope store: it goes through to memory)
        if (threadIdx.x < 64) {
            __threadfence();
            if (threadIdx.x == 0) __hip_atomic_store(lz.flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // wait: wave 0 reads the nfin words with device-scope loads (they bypass this XCD's L2, which may cache earlier polls), one word per
    // lane, with a growing pause between polls.  No cache invalidation afterwards: the launch started with clean caches (the kernel
    // boundary invalidates them) and nobody reads a coefficient before every word is set, so no stale copy of those lines can exist on
    // this XCD -- the first read after the barrier misses and fetches what the finalizing blocks wrote back.  (A device-scope ACQUIRE
    // here -- buffer_inv sc1 by every wave of every block, also of the blocks that start long after the finalize is over -- kept emptying
    // the L2 under the running kernel: measured +2 ms per training step.)
    if (threadIdx.x < 64) {
        int pause = 0;
        while (true) {
            bool ok = true;
            for (int i = threadIdx.x; i < nfin; i += 64)
                ok = ok && __hip_atomic_load(lz.flag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (__all(ok)) break;
            if (pause == 0) __builtin_amdgcn_s_sleep(8);
            else if (pause == 1) __builtin_amdgcn_s_sleep(24);
            else __builtin_amdgcn_s_sleep(64);
            ++pause;
        }
    }
    __syncthreads();
}

// the same with a static scratch array of its own (NOT for kernels whose dynamic LDS request sits at the per-block limit or whose
// occupancy is LDS-bound: gemm_big.hip passes a free piece of its own dynamic segment instead)
__device__ __forceinline__ void bn_lazy_sync(const gast_bn_lazy& lz) {
    __shared__ double lazy_red_static[4][FINS_COLS][2];
    bn_lazy_sync_with(lz, lazy_red_static);
}

// host side: by-value copy for a kernel argument (kind 0 = nothing to do)
static inline gast_bn_lazy lazy_arg(const gast_bn_lazy* lz) {
    gast_bn_lazy r;
    if (lz) r = *lz;
    else { r.kind = 0; r.n = 0; r.flag = nullptr; }
    return r;
}
static inline int lazy_check(const gast_bn_lazy* lz) {
    if (!lz) return 0;
    if ((lz->kind != GAST_BN_LAZY_FWD && lz->kind != GAST_BN_LAZY_BWD) || lz->n < 1 || lz->n > GAST_BN_LAZY_MAX || !lz->flag) return GAST_EINVAL;
    for (int i = 0; i < lz->n; ++i) {
        if (lz->kind == GAST_BN_LAZY_FWD) {
            const gast_bn_fin_job& j = lz->fwd[i];
            if (!j.partials || j.nblk < 1 || j.N < 1 || !j.gamma || !j.beta || !j.scale || !j.shift || !j.mean || !j.rstd || j.count <= 0) return GAST_EINVAL;
        } else {
            const gast_bn_bwd_fin_job& j = lz->bwd[i];
            if (!j.partials || j.nblk < 1 || j.N < 1 || !j.gamma || !j.mean || !j.rstd || !j.dgamma || !j.dbeta || !j.ka || !j.kb || !j.kc) return GAST_EINVAL;
        }
    }
    return 0;
}

}  // namespace gastbn
