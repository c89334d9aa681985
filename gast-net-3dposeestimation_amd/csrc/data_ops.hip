// Device-side ChunkedGenerator gather (reference common/generators.py:93-159).
//
// The reference builds every training batch in a Python loop (one np.pad + fancy-index flip per sample) from numpy sequences on
// the host, then converts float64 -> float32 and copies to the GPU.  Here all sequences stay resident in HBM (concatenated,
// fp32) and ONE launch builds the batch from the epoch's (shuffled) pair table:
//   batch_2d[i, t, j, :] = poses_2d[seq][clamp(start_3d - pad - causal_shift + t, 0, len-1)][perm_j]      (edge padding == clamp)
//   flipped samples: x coordinate negated, left/right joints swapped (perm tables built by the host from kps_left/right)
//   batch_3d likewise with its own joint permutation, batch_cam with coefficients 2 and 7 negated.
// A gather of 0.5 MB per batch: launch-latency bound; the point is that nothing per-sample happens on the host any more.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) chunk_gather_kernel(const float* __restrict__ src, const long* __restrict__ seq_off,
                                                           const int* __restrict__ pairs, long first_pair, int B, int T_out,
                                                           int shift, int J, int F, const int* __restrict__ perm,
                                                           float* __restrict__ out) {
    const long per_sample = (long)T_out * J * F;
    const long total = (long)B * per_sample;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx / per_sample);
        long rem = idx - (long)i * per_sample;
        const int t = (int)(rem / (J * F));
        rem -= (long)t * J * F;
        const int j = (int)(rem / F), f = (int)(rem - (long)j * F);
        const int* p = pairs + (first_pair + i) * 4;
        const int seq = p[0], start = p[1], flip = p[3];
        const long off = seq_off[seq], len = seq_off[seq + 1] - off;
        long fr = (long)start + shift + t;            // shift = -(pad + causal_shift) for the 2D window, 0 for the 3D chunk
        fr = fr < 0 ? 0 : (fr >= len ? len - 1 : fr);
        const int js = flip ? perm[j] : j;
        float v = src[((off + fr) * J + js) * F + f];
        if (flip && f == 0) v = -v;
        out[idx] = v;
    }
}

__global__ void cam_gather_kernel(const float* __restrict__ cams, int ncam, const int* __restrict__ pairs, long first_pair, int B,
                                  float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * ncam) return;
    const int i = idx / ncam, k = idx - i * ncam;
    const int* p = pairs + (first_pair + i) * 4;
    float v = cams[(long)p[0] * ncam + k];
    if (p[3] && (k == 2 || k == 7)) v = -v;          // horizontal distortion coefficients (generators.py:144-147)
    out[idx] = v;
}

// ---- causal streaming (gast_hip/streaming.py): the per-level frame windows advance by one frame per step.  One launch for every
// window of the model: thread = (job, b, x) owns column x of batch entry b through the window's frames, so the in-place shift
// buf[b][t] <- buf[b][t + 1] (t = 0 .. Tb - 2), buf[b][Tb - 1] <- newest[b] needs no second buffer and no synchronisation.
struct ShiftBatch { gast_stream_shift_job j[GAST_STREAM_SHIFT_MAX]; int first[GAST_STREAM_SHIFT_MAX + 1]; int n; };
__global__ void __launch_bounds__(256) stream_shift_kernel(const ShiftBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const gast_stream_shift_job& j = b.j[d];
    const long idx = (long)(blockIdx.x - b.first[d]) * 256 + threadIdx.x;        // over B * X / 4 (16-byte pieces)
    const long X4 = j.X / 4;
    if (idx >= (long)j.B * X4) return;
    const long bb = idx / X4, x = (idx - bb * X4) * 4;
    float* base = (float*)j.buf + bb * (long)j.Tb * j.X + x;
    float4 nxt = j.Tb > 1 ? *(const float4*)(base + j.X) : make_float4(0, 0, 0, 0);
    for (int t = 0; t + 1 < j.Tb; ++t) {
        const float4 cur = nxt;
        if (t + 2 < j.Tb) nxt = *(const float4*)(base + (long)(t + 2) * j.X);
        *(float4*)(base + (long)t * j.X) = cur;
    }
    *(float4*)(base + (long)(j.Tb - 1) * j.X) = *(const float4*)((const float*)j.newest + bb * (long)j.ldnew + x);
}

}  // namespace

extern "C" int gast_stream_shift_multi(const gast_stream_shift_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_STREAM_SHIFT_MAX) return GAST_EINVAL;
    ShiftBatch b;
    b.n = n;
    b.first[0] = 0;
    for (int d = 0; d < n; ++d) {
        const gast_stream_shift_job& j = jobs[d];
        if (!j.buf || !j.newest || j.B < 1 || j.Tb < 1 || j.X < 4) return GAST_EINVAL;
        if (j.X % 4 || j.ldnew % 4 || (((uintptr_t)j.buf) & 15) || (((uintptr_t)j.newest) & 15)) return GAST_EALIGN;
        b.j[d] = j;
        b.first[d + 1] = b.first[d] + (int)(((long)j.B * (j.X / 4) + 255) / 256);
    }
    hipLaunchKernelGGL(stream_shift_kernel, dim3(b.first[n]), dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_chunk_gather(const float* poses2d, const float* poses3d, const float* cams, const int64_t* seq_off,
                                 const int32_t* pairs, long first_pair, int B, int chunk, int pad, int causal_shift, int J2, int F2,
                                 int J3, int F3, int ncam, const int32_t* perm2d, const int32_t* perm3d, float* out2d, float* out3d,
                                 float* outcam, gast_stream_t stream) {
    if (!poses2d || !seq_off || !pairs || !perm2d || !out2d || B < 1 || chunk < 1 || pad < 0 || J2 < 1 || F2 < 1 || first_pair < 0)
        return GAST_EINVAL;
    if ((poses3d != nullptr) != (out3d != nullptr) || (cams != nullptr) != (outcam != nullptr)) return GAST_EINVAL;
    if (poses3d && (!perm3d || J3 < 1 || F3 < 1)) return GAST_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int T2 = chunk + 2 * pad;
    long n2 = (long)B * T2 * J2 * F2;
    int g2 = (int)((n2 + 255) / 256 < 2048 ? (n2 + 255) / 256 : 2048);
    hipLaunchKernelGGL(chunk_gather_kernel, dim3(g2), dim3(256), 0, st, poses2d, (const long*)seq_off, pairs, first_pair, B, T2,
                       -(pad + causal_shift), J2, F2, perm2d, out2d);
    GAST_CHECK_LAUNCH();
    if (poses3d) {
        long n3 = (long)B * chunk * J3 * F3;
        int g3 = (int)((n3 + 255) / 256 < 2048 ? (n3 + 255) / 256 : 2048);
        hipLaunchKernelGGL(chunk_gather_kernel, dim3(g3), dim3(256), 0, st, poses3d, (const long*)seq_off, pairs, first_pair, B, chunk, 0,
                           J3, F3, perm3d, out3d);
        GAST_CHECK_LAUNCH();
    }
    if (cams) {
        if (ncam < 1) return GAST_EINVAL;
        hipLaunchKernelGGL(cam_gather_kernel, dim3((B * ncam + 255) / 256), dim3(256), 0, st, cams, ncam, pairs, first_pair, B, outcam);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}
