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

// one block = one 32x32 tile of one job; 256 threads = 32 x 8
__global__ void __launch_bounds__(256) strided_copy_kernel(const long* __restrict__ jobs, const int* __restrict__ tiles,
                                                           const Bases bs) {
    const long* bases = bs.v;
    __shared__ float tile[TILE][TILE + 1];
    const int* t = tiles + (long)blockIdx.x * 3;
    const long* j = jobs + (long)t[0] * CJ_WORDS;
    const char* src = (const char*)(bases[j[0] & 7] + (j[0] >> 4));     // (element-size independent) byte address
    char* dst = (char*)(bases[j[1] & 7] + (j[1] >> 4));
    const int R = (int)j[2], S = (int)j[3];
    const long srs = j[4], scs = j[5], drs = j[6], dcs = j[7];
    const int flags = (int)j[8];
    const int src_bf16 = flags & 1, dst_bf16 = (flags >> 1) & 1, accumulate = (flags >> 2) & 1, zero_fill = (flags >> 3) & 1;
    const int r0 = t[1] * TILE, c0 = t[2] * TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // read with the source-contiguous dimension on tx
    const bool src_col_fast = (scs <= srs);
#pragma unroll
    for (int i = 0; i < TILE; i += 8) {
        int r = src_col_fast ? r0 + ty + i : r0 + tx;
        int c = src_col_fast ? c0 + tx : c0 + ty + i;
        float v = 0.f;
        if (r < R && c < S && !zero_fill) v = load_any(src, r * srs + c * scs, src_bf16);
        tile[r - r0][c - c0] = v;
    }
    __syncthreads();
    const bool dst_col_fast = (dcs <= drs);
#pragma unroll
    for (int i = 0; i < TILE; i += 8) {
        int r = dst_col_fast ? r0 + ty + i : r0 + tx;
        int c = dst_col_fast ? c0 + tx : c0 + ty + i;
        if (r < R && c < S) {
            float v = tile[r - r0][c - c0];
            long o = r * drs + c * dcs;
            if (dst_bf16) ((bf16_t*)dst)[o] = f2bf(v);
            else if (accumulate) ((float*)dst)[o] += v;
            else ((float*)dst)[o] = v;
        }
    }
}

// fold: v[k] = sum_m W[m][k] * w[m]  (k < C, m < Ci), a = sum_m w[m] * b[m]
//   -> dst_row[k * ds_row]  and dst_col[k * ds_col] (activation dtype), bias_dst (fp32)
// grid = (njobs, ceil(C/32)); 256 threads = 32 columns x 8 m-lanes (each lane sums every 8th m, LDS combine)
constexpr int FJ_WORDS = 12;
__global__ void __launch_bounds__(256) fold_kernel(const long* __restrict__ jobs, const Bases bs) {
    const long* bases = bs.v;
    __shared__ float sred[8][32];
    const long* j = jobs + (long)blockIdx.x * FJ_WORDS;
    const float* W = (const float*)(bases[j[0] & 7] + (j[0] >> 4));
    const float* w = (const float*)(bases[j[1] & 7] + (j[1] >> 4));
    const float* b = (const float*)(bases[j[2] & 7] + (j[2] >> 4));
    const int Ci = (int)j[3], C = (int)j[4];
    if ((int)blockIdx.y * 32 >= C) return;
    char* d_row = (char*)(bases[j[5] & 7] + (j[5] >> 4));
    const long ds_row = j[6];
    char* d_col = (char*)(bases[j[7] & 7] + (j[7] >> 4));
    const long ds_col = j[8];
    float* d_bias = (float*)(bases[j[9] & 7] + (j[9] >> 4));
    const int dst_bf16 = (int)j[10];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int k = blockIdx.y * 32 + cx;
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
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float a = 0.f;
        for (int m = 0; m < Ci; ++m) a = fmaf(w[m], b[m], a);
        *d_bias = a;
    }
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
    hipLaunchKernelGGL(strided_copy_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const long*)jobs, tiles, b);
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
