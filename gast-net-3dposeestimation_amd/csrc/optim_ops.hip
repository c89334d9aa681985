// Flat Adam / AMSGrad step for the GAST-Net training step (gfx950).
//
// The reference trains with torch.optim.Adam(lr, amsgrad=True) over 165 parameter tensors (reference trainval.py:78,
// main.py:238).  torch's fused multi-tensor Adam spends 0.33 ms per step on them (8 launches whose blocks are sized by
// the largest tensors); with parameters, gradients and moments each living in ONE flat fp32 buffer the update is a pure
// 36-byte-per-parameter stream: one launch, 16-byte accesses, ~50 us for the 6.9 M parameters of the BASELINE model.
// Arithmetic follows torch/optim/adam.py (_single_tensor_adam) term by term.
#include "common.h"

namespace {

struct AdamCoef { float step_size, inv_sqrt_bc2; };

__device__ __forceinline__ AdamCoef adam_coef(int step, float lr, float beta1, float beta2) {
    const double t = (double)step;
    const double bc1 = 1.0 - exp(t * log((double)beta1));
    const double bc2 = 1.0 - exp(t * log((double)beta2));
    AdamCoef c;
    c.step_size = (float)((double)lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return c;
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float& vmax, bool amsgrad, float beta1, float beta2,
                                         float eps, float wd, AdamCoef c) {
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = fmaf(1.f - beta1, g - m, m);              // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(beta2, v, (1.f - beta2) * g * g);
    float vv = v;
    if (amsgrad) { vmax = fmaxf(vmax, v); vv = vmax; }
    const float denom = sqrtf(vv) * c.inv_sqrt_bc2 + eps;
    p -= c.step_size * (m / denom);
}

// GAST_NONFINITE_FLAGS per-block verdicts of gast_nonfinite_scan (every block writes its own word on every call: nothing to reset);
// block-uniform: is any of them set?  (nullptr: no guard)
__device__ __forceinline__ bool skip_any(const int* __restrict__ skip) {
    if (!skip) return false;
    int bad = 0;
    for (int i = threadIdx.x & 63; i < GAST_NONFINITE_FLAGS; i += 64) bad |= skip[i];
    return __any(bad != 0);
}

// flags[b] = 1 when block b's slice of g holds an inf / NaN (exponent bits all ones), else 0
__global__ void __launch_bounds__(256) nonfinite_scan_kernel(const float* __restrict__ g, long n, int* __restrict__ flags) {
    __shared__ int sbad;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    const long n4 = n >> 2;
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint4 v = ((const uint4*)g)[i];
        bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
               ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
    if (bad) sbad = 1;               // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = sbad;
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ vmax, long n, const int* __restrict__ step,
                                                   float lr, float beta1, float beta2, float eps, float wd, float gscale,
                                                   const int* __restrict__ skip) {
    if (skip_any(skip)) return;         // (non-finite gradient detected by gast_nonfinite_scan: the whole update is skipped, loss-scaled 16-bit mode)
    // (Round 4 tried to fold the step increment into this kernel -- every block computing with step + 1 and the last block to take a
    // ticket storing it back: 4096 device-scope atomics on ONE address cost 194 us (~47 ns each, serialised at the memory side), against
    // 4.6 us for the one-thread increment kernel.  Same-address atomics from every block of a grid are never cheap on this chip.)
    const AdamCoef c = adam_coef(*step, lr, beta1, beta2);
    const bool ams = vmax != nullptr;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
        float4 xx = ams ? ((float4*)vmax)[i] : make_float4(0, 0, 0, 0);
        adam_one(pp.x, gg.x * gscale, mm.x, vv.x, xx.x, ams, beta1, beta2, eps, wd, c);
        adam_one(pp.y, gg.y * gscale, mm.y, vv.y, xx.y, ams, beta1, beta2, eps, wd, c);
        adam_one(pp.z, gg.z * gscale, mm.z, vv.z, xx.z, ams, beta1, beta2, eps, wd, c);
        adam_one(pp.w, gg.w * gscale, mm.w, vv.w, xx.w, ams, beta1, beta2, eps, wd, c);
        ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
        if (ams) ((float4*)vmax)[i] = xx;
    }
    if (blockIdx.x == 0) {
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            float x = ams ? vmax[i] : 0.f;
            adam_one(p[i], g[i] * gscale, m[i], v[i], x, ams, beta1, beta2, eps, wd, c);
            if (ams) vmax[i] = x;
        }
    }
}

__global__ void __launch_bounds__(64) step_inc_kernel(int* step, const int* skip, long long* skipped) {
    const bool sk = skip_any(skip);
    if (threadIdx.x == 0) {
        if (!sk) *step += 1;
        else if (skipped) *skipped += 1;
    }
}

// ---- pass prologue ("prep"): everything a forward or backward pass needs before its first real kernel, as ONE launch -- the
// zero fills of the accumulation arenas / gradient buffers (up to GAST_PREP_MAX_ZERO regions), the dropout seed bump + its
// per-pass copy, and the 3 -> 8 column padding of d loss / d pred that the shrink layer's gradient GEMMs read.  These were 4 + 3
// separate graph nodes per step (torch fills, an add, a clone, a slice copy) at ~4.7 us of boundary each.
constexpr int PREP_CHUNK = 256 * 16 * 16;      // bytes one block zeroes: 256 threads x 16 B x 16 rounds
struct PrepArgs {
    gast_prep_args a;
    int first[GAST_PREP_MAX_ZERO + 1];         // first block of every zero job; first[nzero] = first block of the pad job
    int nblk_pad;
};
__global__ void __launch_bounds__(256) prep_kernel(const PrepArgs p) {
    const gast_prep_args& a = p.a;
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk == 0 && tid == 0 && a.seed_ctr) {
        const uint32_t s = *a.seed_ctr + 1u;
        *a.seed_ctr = s;
        if (a.seed_out) *a.seed_out = s;
    }
    if (blk < p.first[a.nzero]) {
        int d = 0;
        while (d + 1 < a.nzero && blk >= p.first[d + 1]) ++d;
        char* base = (char*)a.zero[d].ptr;
        const long off0 = (long)(blk - p.first[d]) * PREP_CHUNK, end = a.zero[d].bytes;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const long o = off0 + ((long)i * 256 + tid) * 16;
            if (o < end) *(uint4*)(base + o) = z;
        }
        return;
    }
    // pad job: dst[r][c] = c < cols_src ? src[r][c] : 0, one thread per destination row
    const long r = (long)(blk - p.first[a.nzero]) * 256 + tid;
    if (r < a.pad_rows) {
        const float sc = a.pad_scale != 0.f ? a.pad_scale : 1.f;
        for (int c = 0; c < a.pad_cols_dst; ++c) {
            const float v = c < a.pad_cols_src ? sc * a.pad_src[r * a.pad_cols_src + c] : 0.f;
            if (a.pad_dst_h16) ((bf16_t*)a.pad_dst)[r * a.pad_cols_dst + c] = f2bf(v);
            else a.pad_dst[r * a.pad_cols_dst + c] = v;
        }
    }
}
__global__ void null_kernel() {}

// mpjpe (reference common/loss.py:5-11): mean over rows of ||pred[r,:] - target[r,:]||_2, D <= 4 components per row.
// One block (deterministic summation order); also emits dirs[r,:] = (pred - target) / (norm * rows), the gradient of the loss
// w.r.t. pred, so that backward is a single scaling.  A zero-length difference has gradient 0 (torch: subgradient 0).
__global__ void __launch_bounds__(1024) mpjpe_kernel(const float* __restrict__ pred, const float* __restrict__ target, long rows, int D,
                                                     float* __restrict__ loss, float* __restrict__ dirs) {
    __shared__ double sred[1024];
    double acc = 0.0;
    const float inv_rows = 1.f / (float)rows;
    for (long r = threadIdx.x; r < rows; r += 1024) {
        float d[4], ss = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            d[q] = q < D ? pred[r * D + q] - target[r * D + q] : 0.f;
            ss = fmaf(d[q], d[q], ss);
        }
        const float nrm = sqrtf(ss);
        acc += (double)nrm;
        const float sc = nrm > 0.f ? inv_rows / nrm : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < D) dirs[r * D + q] = d[q] * sc;
    }
    sred[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(sred[0] / (double)rows);
}

}  // namespace

extern "C" int gast_adam_step(float* p, const float* g, float* m, float* v, float* vmax, long n, int* step, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, gast_stream_t stream) {
    return gast_adam_step_guarded(p, g, m, v, vmax, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale, nullptr, nullptr, stream);
}

extern "C" int gast_nonfinite_scan(const float* g, long n, int* flags, gast_stream_t stream) {
    if (!g || !flags || n < 1 || ((uintptr_t)g & 15)) return GAST_EINVAL;
    hipLaunchKernelGGL(nonfinite_scan_kernel, dim3(GAST_NONFINITE_FLAGS), dim3(256), 0, (hipStream_t)stream, g, n, flags);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_adam_step_guarded(float* p, const float* g, float* m, float* v, float* vmax, long n, int* step, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, float grad_scale, const int* skip, long long* skipped,
                                      gast_stream_t stream) {
    if (!p || !g || !m || !v || !step || n < 1) return GAST_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) return GAST_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, st, step, skip, skipped);
    long nb = ((n >> 2) + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, st, p, g, m, v, vmax, n, step, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, skip);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_prep(const gast_prep_args* args, gast_stream_t stream) {
    if (!args || args->nzero < 0 || args->nzero > GAST_PREP_MAX_ZERO) return GAST_EINVAL;
    PrepArgs p;
    p.a = *args;
    int nb = 0;
    for (int d = 0; d < args->nzero; ++d) {
        const gast_zero_job& z = args->zero[d];
        if (!z.ptr || z.bytes < 0) return GAST_EINVAL;
        if (((uintptr_t)z.ptr & 15) || (z.bytes & 15)) return GAST_EALIGN;
        p.first[d] = nb;
        nb += (int)((z.bytes + PREP_CHUNK - 1) / PREP_CHUNK);
    }
    p.first[args->nzero] = nb;
    p.nblk_pad = 0;
    if (args->pad_rows > 0) {
        if (!args->pad_src || !args->pad_dst || args->pad_cols_src < 1 || args->pad_cols_dst < args->pad_cols_src) return GAST_EINVAL;
        p.nblk_pad = (int)((args->pad_rows + 255) / 256);
    }
    nb += p.nblk_pad;
    if (nb == 0) {
        if (!args->seed_ctr) return 0;
        nb = 1;                                   // (seed bump only: block 0 falls through to an empty pad job)
    }
    hipLaunchKernelGGL(prep_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_mpjpe(const float* pred, const float* target, long rows, int D, float* loss, float* dirs, gast_stream_t stream) {
    if (!pred || !target || !loss || !dirs || rows < 1) return GAST_EINVAL;
    if (D < 1 || D > 4) return GAST_ERANGE;
    hipLaunchKernelGGL(mpjpe_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, rows, D, loss, dirs);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_null_launch(gast_stream_t stream) {
    hipLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    GAST_CHECK_LAUNCH();
    return 0;
}
