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

}  // namespace

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
