// Graph kernels of the GAST-Net hot path on gfx950: the channel-wise semantic graph convolution's masked-softmax
// adjacency + neighbour aggregation (reference model/local_attention.py:35-53) and the additive global joint attention
// (reference model/global_attention.py:52-82).  All of them are HBM/latency bound (AI < 5 F/B): coalesced channel-major
// accesses, the J x J tiles live in LDS, no MFMA.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int JMAX = 19;   // largest supported skeleton (Human3.6M + toes)

struct Pat {
    int J, nnz, Dr, Dc;
    const int32_t *row_ptr, *col, *col_ptr, *crow, *cedge;
    const int32_t *ell_rj, *ell_rk, *ell_ci, *ell_ck;   // padded fixed-degree views: [J][Dr] / [J][Dc]; padding -> edge id nnz
};
__device__ __host__ __forceinline__ Pat make_pat(const int32_t* p, int J, int nnz) {
    Pat q;
    q.J = J; q.nnz = nnz;
    q.row_ptr = p + 2;
    q.col = q.row_ptr + (J + 1);
    q.col_ptr = q.col + nnz;
    q.crow = q.col_ptr + (J + 1);
    q.cedge = q.crow + nnz;
    const int32_t* e = q.cedge + nnz;
    q.Dr = e[0]; q.Dc = e[1];
    q.ell_rj = e + 2;
    q.ell_rk = q.ell_rj + J * q.Dr;
    q.ell_ci = q.ell_rk + J * q.Dr;
    q.ell_ck = q.ell_ci + J * q.Dc;
    return q;
}

// ------------------------------------------------------------------------------------------------ adjacency softmax
// one thread per (channel c, row i): A_t[k][c] = exp(e[c][k] - max) / sum over the edges k of row i
__device__ __forceinline__ void semch_adj_fwd_body(const float* __restrict__ e, int C, const int32_t* __restrict__ pat,
                                                   float* __restrict__ A_t) {
    const int J = pat[0], nnz = pat[1];
    const Pat p = make_pat(pat, J, nnz);
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * J) return;
    int i = idx / C, c = idx - i * C;
    int k0 = p.row_ptr[i], k1 = p.row_ptr[i + 1];
    float mx = -3.0e38f;
    for (int k = k0; k < k1; ++k) mx = fmaxf(mx, e[(long)c * nnz + k]);
    float sum = 0.f;
    for (int k = k0; k < k1; ++k) sum += expf(e[(long)c * nnz + k] - mx);
    float inv = 1.f / sum;
    for (int k = k0; k < k1; ++k) A_t[(long)k * C + c] = expf(e[(long)c * nnz + k] - mx) * inv;
    if (i == 0) A_t[(long)nnz * C + c] = 0.f;   // the all-zero weight row that padded (ELL) edge slots point at
}
__global__ void semch_adj_fwd_kernel(const float* __restrict__ e, int C, const int32_t* __restrict__ pat, float* __restrict__ A_t) {
    semch_adj_fwd_body(e, C, pat, A_t);
}
// all adjacency softmaxes of a pass in one launch (they depend on parameters only): blockIdx.y = job
struct AdjBatch { gast_adj_job j[GAST_ADJ_MAX_BATCH]; };
__global__ void semch_adj_fwd_multi_kernel(const AdjBatch b) {
    const gast_adj_job& j = b.j[blockIdx.y];
    semch_adj_fwd_body(j.e, j.C, j.pat, j.A_t);
}

__device__ __forceinline__ void semch_adj_bwd_body(const float* __restrict__ dA_t, const float* __restrict__ A_t, int C,
                                                   const int32_t* __restrict__ pat, float* __restrict__ de, int accumulate = 0) {
    const int J = pat[0], nnz = pat[1];
    const Pat p = make_pat(pat, J, nnz);
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * J) return;
    int i = idx / C, c = idx - i * C;
    int k0 = p.row_ptr[i], k1 = p.row_ptr[i + 1];
    float dot = 0.f;
    for (int k = k0; k < k1; ++k) dot += A_t[(long)k * C + c] * dA_t[(long)k * C + c];
    for (int k = k0; k < k1; ++k) {
        const float v = A_t[(long)k * C + c] * (dA_t[(long)k * C + c] - dot);
        if (accumulate) de[(long)c * nnz + k] += v; else de[(long)c * nnz + k] = v;
    }
}
__global__ void semch_adj_bwd_kernel(const float* __restrict__ dA_t, const float* __restrict__ A_t, int C,
                                     const int32_t* __restrict__ pat, float* __restrict__ de) {
    semch_adj_bwd_body(dA_t, A_t, C, pat, de);
}
__global__ void semch_adj_bwd_multi_kernel(const AdjBatch b, int accumulate) {
    const gast_adj_job& j = b.j[blockIdx.y];
    semch_adj_bwd_body(j.dA_t, j.A_t, j.C, j.pat, j.e, accumulate);     // j.e = de (output) in the backward form
}

// ------------------------------------------------------------------------------------------------ neighbour aggregation
// thread <-> (frame slot, group of 4 channels).  TPF = threads per frame = min(C/4, 256), FB = 256 / TPF frame slots.
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <typename T>
__global__ void __launch_bounds__(256) semch_agg_fwd_kernel(const T* __restrict__ H, int ldh, int F, int J, int C,
                                                            const float* __restrict__ A_sym, const int32_t* __restrict__ pat_sym,
                                                            const float* __restrict__ A_con, const int32_t* __restrict__ pat_con,
                                                            T* __restrict__ Y, int ldy, float* __restrict__ partials, int TPF, int FB,
                                                            const float* __restrict__ ctr_s, const float* __restrict__ ctr_c) {
    __shared__ float sred[256][8];
    const int tid = threadIdx.x;
    const int slot = tid / TPF, ct = tid - slot * TPF;
    const int C4 = C >> 2;
    const bool active = slot < FB;
    for (int cg0 = 0; cg0 < C4; cg0 += TPF) {
        const int cg = cg0 + ct;
        const bool cin = active && cg < C4;
        const int c = cg * 4;
        float4 s1[2], s2[2];
        s1[0] = s1[1] = s2[0] = s2[1] = make_float4(0, 0, 0, 0);
        if (cin) {
            for (int f = blockIdx.x * FB + slot; f < F; f += gridDim.x * FB) {
                const T* Hf = H + (long)f * J * ldh;
                T* Yf = Y + (long)f * J * ldy;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int32_t* pat = g == 0 ? pat_sym : pat_con;
                    const float* A = g == 0 ? A_sym : A_con;
                    const Pat p = make_pat(pat, J, pat[1]);
                    const int h0c = g * 2 * C + c, h1c = g * 2 * C + C + c;
                    for (int i = 0; i < J; ++i) {
                        float4 acc = make_float4(0, 0, 0, 0);
                        for (int k = p.row_ptr[i]; k < p.row_ptr[i + 1]; ++k) {
                            int j = p.col[k];
                            float4 av = *(const float4*)(A + (long)k * C + c);
                            float4 hv = ld4(Hf + (long)j * ldh + (j == i ? h0c : h1c));
                            acc = fma4(av, hv, acc);
                        }
                        const float* ctr = g == 0 ? ctr_s : ctr_c;
                        if (ctr) { const float4 c4v = *(const float4*)(ctr + c); acc.x -= c4v.x; acc.y -= c4v.y; acc.z -= c4v.z; acc.w -= c4v.w; }
                        acc = rnd4(acc, (const T*)nullptr);
                        st4(Yf + (long)i * ldy + g * C + c, acc);
                        s1[g] = add4(s1[g], acc);
                        s2[g] = fma4(acc, acc, s2[g]);
                    }
                }
            }
        }
        // reduce over the frame slots of this block, then one partial row per block
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            __syncthreads();
            sred[tid][0] = s1[g].x; sred[tid][1] = s1[g].y; sred[tid][2] = s1[g].z; sred[tid][3] = s1[g].w;
            sred[tid][4] = s2[g].x; sred[tid][5] = s2[g].y; sred[tid][6] = s2[g].z; sred[tid][7] = s2[g].w;
            __syncthreads();
            if (slot == 0 && cg < C4) {
                float t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = 0.f;
                for (int sl = 0; sl < FB; ++sl)
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] += sred[sl * TPF + ct][q];
                float* pp = partials + ((long)blockIdx.x * 2 * C + g * C + c) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) { pp[2 * q] = t[q]; pp[2 * q + 1] = t[4 + q]; }
            }
        }
    }
}

// ---- fixed-degree variants: every row (column) of the pattern is padded to D slots (padding -> zero weight row nnz), so
// the edge loops have compile-time trip counts, all (j,k) lookups are hoisted and the D weight/feature loads of a row are
// issued back to back (the CSR version exposes one scalar-load + one vector-load latency per edge).
template <typename T, int D>
__device__ __forceinline__ void agg_rows_ell(const T* __restrict__ Hf, int ldh, T* __restrict__ Yf, int ldy, int J, int C,
                                             const float* __restrict__ A, int lda, int acol, const int32_t* __restrict__ ell_j,
                                             const int32_t* __restrict__ ell_k, int h0c, int h1c, int yc, float4& s1, float4& s2,
                                             float4 ctr, int i0, int istep) {
#pragma unroll 4
    for (int i = i0; i < J; i += istep) {
        int jj[D], kk[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { jj[d] = ell_j[i * D + d]; kk[d] = ell_k[i * D + d]; }
        float4 av[D], hv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            av[d] = *(const float4*)(A + kk[d] * lda + acol);
            hv[d] = ld4(Hf + (long)jj[d] * ldh + (jj[d] == i ? h0c : h1c));
        }
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < D; ++d) acc = fma4(av[d], hv[d], acc);
        acc.x -= ctr.x; acc.y -= ctr.y; acc.z -= ctr.z; acc.w -= ctr.w;
        acc = rnd4(acc, (const T*)nullptr);
        st4(Yf + (long)i * ldy + yc, acc);
        s1 = add4(s1, acc);
        s2 = fma4(acc, acc, s2);
    }
}

// Block = (frame block fb, chunk of CC <= 64 channels): the softmax coefficients of the chunk ([nnz + 1][CC] fp32 per pattern)
// are staged in LDS once per block -- read from global per (frame, joint, slot) they were 2/3 of the kernel's L2 traffic
// (136 padded slots x C x 4 bytes per frame against 136 x C x 2 bytes of features).  Thread = (frame slot, 4 channels).
template <typename T, int DS, int DC>
__global__ void __launch_bounds__(256) semch_agg_fwd_ell_kernel(const T* __restrict__ H, int ldh, int F, int J, int C,
                                                                const float* __restrict__ A_sym, const int32_t* __restrict__ pat_sym,
                                                                const float* __restrict__ A_con, const int32_t* __restrict__ pat_con,
                                                                T* __restrict__ Y, int ldy, float* __restrict__ partials, int TPF, int FB,
                                                                const float* __restrict__ ctr_s, const float* __restrict__ ctr_c,
                                                                int nchunk, int CC) {
    extern __shared__ __attribute__((aligned(16))) float sAgg[];      // [nnz_s + 1][CC] | [nnz_c + 1][CC]
    __shared__ float sred[256][8];
    const int tid = threadIdx.x;
    const int slot = tid / TPF, ct = tid - slot * TPF;
    const int fb = blockIdx.x / nchunk, ch = blockIdx.x - fb * nchunk, nfb = gridDim.x / nchunk;
    const Pat ps = make_pat(pat_sym, J, pat_sym[1]), pc = make_pat(pat_con, J, pat_con[1]);
    float* sAs = sAgg;
    float* sAc = sAgg + (ps.nnz + 1) * CC;
    const int CC4 = CC >> 2;
    for (int t = tid; t < (ps.nnz + 1) * CC4; t += 256) {
        const int k = t / CC4, q = t - k * CC4;
        *(float4*)(sAs + k * CC + q * 4) = *(const float4*)(A_sym + (long)k * C + ch * CC + q * 4);
    }
    for (int t = tid; t < (pc.nnz + 1) * CC4; t += 256) {
        const int k = t / CC4, q = t - k * CC4;
        *(float4*)(sAc + k * CC + q * 4) = *(const float4*)(A_con + (long)k * C + ch * CC + q * 4);
    }
    __syncthreads();
    const int cl = ct * 4, c = ch * CC + cl;
    const bool cin = slot < FB && cl < CC && c < C;
    float4 s1[2], s2[2];
    s1[0] = s1[1] = s2[0] = s2[1] = make_float4(0, 0, 0, 0);
    if (cin) {
        for (int f = fb * FB + slot; f < F; f += nfb * FB) {
            const T* Hf = H + (long)f * J * ldh;
            T* Yf = Y + (long)f * J * ldy;
            const float4 z4 = make_float4(0, 0, 0, 0);
            // gridDim.y = joint split (few frames: the rows i = blockIdx.y, blockIdx.y + gridDim.y, ... of every frame)
            agg_rows_ell<T, DS>(Hf, ldh, Yf, ldy, J, C, sAs, CC, cl, ps.ell_rj, ps.ell_rk, c, C + c, c, s1[0], s2[0],
                                ctr_s ? *(const float4*)(ctr_s + c) : z4, blockIdx.y, gridDim.y);
            agg_rows_ell<T, DC>(Hf, ldh, Yf, ldy, J, C, sAc, CC, cl, pc.ell_rj, pc.ell_rk, 2 * C + c, 3 * C + c, C + c, s1[1], s2[1],
                                ctr_c ? *(const float4*)(ctr_c + c) : z4, blockIdx.y, gridDim.y);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        __syncthreads();
        sred[tid][0] = s1[g].x; sred[tid][1] = s1[g].y; sred[tid][2] = s1[g].z; sred[tid][3] = s1[g].w;
        sred[tid][4] = s2[g].x; sred[tid][5] = s2[g].y; sred[tid][6] = s2[g].z; sred[tid][7] = s2[g].w;
        __syncthreads();
        if (slot == 0 && cl < CC && c < C) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = 0.f;
            for (int sl = 0; sl < FB; ++sl)
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] += sred[sl * TPF + ct][q];
            float* pp = partials + (((long)blockIdx.y * nfb + fb) * 2 * C + g * C + c) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { pp[2 * q] = t[q]; pp[2 * q + 1] = t[4 + q]; }
        }
    }
}

// backward, fixed-degree column view: for column j the slots (i, k) give dh0 / dh1
template <typename T, int D>
__device__ __forceinline__ void agg_cols_ell(const T* __restrict__ dYf, int ldy, T* __restrict__ dHf, int lddh, int J, int C,
                                             const float* __restrict__ A, int lda, int acol, const int32_t* __restrict__ ell_i,
                                             const int32_t* __restrict__ ell_k, int h0c, int h1c, int yc, int j0, int jstep) {
#pragma unroll 4
    for (int j = j0; j < J; j += jstep) {
        int ii[D], kk[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { ii[d] = ell_i[j * D + d]; kk[d] = ell_k[j * D + d]; }
        float4 av[D], dv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            av[d] = *(const float4*)(A + kk[d] * lda + acol);
            dv[d] = ld4(dYf + (long)ii[d] * ldy + yc);
        }
        float4 d0 = make_float4(0, 0, 0, 0), d1 = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (ii[d] == j) d0 = fma4(av[d], dv[d], d0); else d1 = fma4(av[d], dv[d], d1);
        }
        st4(dHf + (long)j * lddh + h0c, d0);
        st4(dHf + (long)j * lddh + h1c, d1);
    }
}

// Phase A: dh0/dh1, thread = (frame slot, 4 channels).  Phase B: dA[k][c] = sum_f dY[f,i_k,c] * h[f,j_k,c], thread = (edge k,
// 4 channels) walking the block's frames with 4 independent frame loads in flight -- one owner per (k, c): no atomics, no LDS
// (the first versions used one global / LDS atomic per thread, edge and channel and were bound by atomic contention).
template <typename T, int DS, int DC>
__global__ void __launch_bounds__(256) semch_agg_bwd_ell_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ H, int ldh,
                                                                int F, int J, int C,
                                                                const float* __restrict__ A_sym, const int32_t* __restrict__ pat_sym,
                                                                const float* __restrict__ A_con, const int32_t* __restrict__ pat_con,
                                                                T* __restrict__ dH, int lddh, float* __restrict__ part, int nfb,
                                                                int nchunk, int CC, int TPF, int FB) {
    // gridDim.y = joint split (few frames: the per-thread chains over 17 joints / 82 edges are the critical path, so the joints
    // of phase A and the edges of phase B are dealt to gridDim.y blocks)
    const int jpart = blockIdx.y, jsplit = gridDim.y;
    __shared__ int s_ei[2 * JMAX * JMAX], s_ej[2 * JMAX * JMAX];     // (i, j) of every edge, sym edges first
    extern __shared__ __attribute__((aligned(16))) float sAgg[];     // coefficient slices [nnz_s + 1][CC] | [nnz_c + 1][CC] (as forward)
    const int tid = threadIdx.x;
    const int fb = blockIdx.x / nchunk, ch = blockIdx.x - fb * nchunk;
    const int slot = tid / TPF, ct = tid - slot * TPF;
    const Pat ps = make_pat(pat_sym, J, pat_sym[1]), pc = make_pat(pat_con, J, pat_con[1]);
    const int nnz_s = ps.nnz, nnz_c = pc.nnz, nnz_t = nnz_s + nnz_c;
    float* sAs = sAgg;
    float* sAc = sAgg + (nnz_s + 1) * CC;
    for (int t = tid; t < (nnz_s + 1) * (CC >> 2); t += 256) {
        const int k = t / (CC >> 2), q = t - k * (CC >> 2);
        *(float4*)(sAs + k * CC + q * 4) = *(const float4*)(A_sym + (long)k * C + ch * CC + q * 4);
    }
    for (int t = tid; t < (nnz_c + 1) * (CC >> 2); t += 256) {
        const int k = t / (CC >> 2), q = t - k * (CC >> 2);
        *(float4*)(sAc + k * CC + q * 4) = *(const float4*)(A_con + (long)k * C + ch * CC + q * 4);
    }
    __syncthreads();
    for (int t = tid; t < 2 * J; t += 256) {
        const Pat& p = t < J ? ps : pc;
        const int i = t < J ? t : t - J, base = t < J ? 0 : nnz_s;
        for (int k = p.row_ptr[i]; k < p.row_ptr[i + 1]; ++k) { s_ei[base + k] = i; s_ej[base + k] = p.col[k]; }
    }
    const int cl = ct * 4;
    const int c = ch * CC + cl;
    if (slot < FB && cl < CC && c < C) {
        for (int f = fb * FB + slot; f < F; f += nfb * FB) {
            const T* dYf = dY + (long)f * J * ldy;
            T* dHf = dH + (long)f * J * lddh;
            agg_cols_ell<T, DS>(dYf, ldy, dHf, lddh, J, C, sAs, CC, cl, ps.ell_ci, ps.ell_ck, c, C + c, c, jpart, jsplit);
            agg_cols_ell<T, DC>(dYf, ldy, dHf, lddh, J, C, sAc, CC, cl, pc.ell_ci, pc.ell_ck, 2 * C + c, 3 * C + c, C + c, jpart, jsplit);
        }
    }
    __syncthreads();
    const int CC4 = CC >> 2;
    const int fstep = nfb;          // frames of this block: fb, fb + nfb, ... in units of single frames (slot-interleaved)
    for (int pidx = tid; pidx < nnz_t * CC4; pidx += 256) {
        const int k = pidx / CC4, c4 = pidx - k * CC4;
        const int cg = ch * CC + c4 * 4;
        if (cg >= C || (k % jsplit) != jpart) continue;
        const int g = k < nnz_s ? 0 : 1;
        const int i = s_ei[k], j = s_ej[k];
        const int yc = g * C + cg;
        const int hc = g * 2 * C + (i == j ? 0 : C) + cg;
        float4 acc0 = make_float4(0, 0, 0, 0), acc1 = acc0, acc2 = acc0, acc3 = acc0;
        // the block's frames are f = (fb + q * nfb) * FB + slot, slot < FB
        for (int fq = fb * FB; fq < F; fq += fstep * FB) {
            const int nf = min(FB, F - fq);
            int s0 = 0;
            for (; s0 + 4 <= nf; s0 += 4) {
                const long f0 = fq + s0;
                const float4 d0 = ld4(dY + ((f0 + 0) * J + i) * ldy + yc), h0 = ld4(H + ((f0 + 0) * J + j) * ldh + hc);
                const float4 d1 = ld4(dY + ((f0 + 1) * J + i) * ldy + yc), h1 = ld4(H + ((f0 + 1) * J + j) * ldh + hc);
                const float4 d2 = ld4(dY + ((f0 + 2) * J + i) * ldy + yc), h2 = ld4(H + ((f0 + 2) * J + j) * ldh + hc);
                const float4 d3 = ld4(dY + ((f0 + 3) * J + i) * ldy + yc), h3 = ld4(H + ((f0 + 3) * J + j) * ldh + hc);
                acc0 = fma4(d0, h0, acc0); acc1 = fma4(d1, h1, acc1); acc2 = fma4(d2, h2, acc2); acc3 = fma4(d3, h3, acc3);
            }
            for (; s0 < nf; ++s0) {
                const long f0 = fq + s0;
                acc0 = fma4(ld4(dY + (f0 * J + i) * ldy + yc), ld4(H + (f0 * J + j) * ldh + hc), acc0);
            }
        }
        float4 r = add4(add4(acc0, acc1), add4(acc2, acc3));
        *(float4*)(part + ((long)fb * nnz_t + k) * C + cg) = r;
    }
}

// Round 4: the same backward with the frame tiles staged in LDS.  The kernel above reads every dY row once per neighbour in phase A and
// once per edge in phase B (and every H row once per edge) -- from L2 / HBM, since a block's frames do not stay in the cache between its
// phases: 311 MB fetched + written per launch against 249 algorithmic, each as its own 16-byte load.  Here a block stages U frames of its
// 32-channel chunk (dY: [U][J][2][CC], H: [U][J][4][CC] fp32) with one coalesced load per element, and both phases read LDS:
//   phase A  thread = (frame u, column j, 4 channels): dh0 / dh1 over the D padded (row, edge) slots of the column;
//   phase B  thread = up to NPB fixed (edge k, 4 channels) pairs whose dA accumulators live in registers across ALL frames of the block
//            (one owner per pair: no atomics; the block writes one partial row at the end);
// the next group's elements are requested into registers before the current group is processed (without that prefetch the kernel
// was 5 % SLOWER than the one above: two barriers per group with nothing in flight).
#ifndef GAST_AGG_LDS_U
#define GAST_AGG_LDS_U 2
#endif
constexpr int AGG_LDS_U = GAST_AGG_LDS_U, AGG_LDS_CC = 32, AGG_LDS_NPB = 5;
// BN (round 5): dY arrives BEFORE the BatchNorm backward of bn_1 | bn_2 (the masked gradient the preceding GEMM epilogue wrote) and the
// kernel applies dy = ka*dY + kb*Y + kc (Y = the pre-BatchNorm aggregation output of the forward) while staging -- the stand-alone
// gast_bn_bwd_apply pass over dY (a read-modify-write of P x 2C, at the HBM roofline) disappears.  Same fma nesting as
// bn_bwd_apply_kernel: the values are bit-equal to the two-launch form.
template <typename T, int DS, int DC, bool BN = false>
__global__ void __launch_bounds__(256) semch_agg_bwd_lds_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ H, int ldh,
                                                                int F, int J, int C,
                                                                const float* __restrict__ A_sym, const int32_t* __restrict__ pat_sym,
                                                                const float* __restrict__ A_con, const int32_t* __restrict__ pat_con,
                                                                T* __restrict__ dH, int lddh, float* __restrict__ part, int nfb, int nchunk,
                                                                const T* __restrict__ Yp, int ldyp, const float* ka, const float* kb,
                                                                const float* kc) {
    constexpr int U = AGG_LDS_U, CC = AGG_LDS_CC, CC4 = CC / 4, NPB = AGG_LDS_NPB;
    __shared__ int s_ei[2 * JMAX * JMAX], s_ej[2 * JMAX * JMAX];     // (i, j) of every edge, sym edges first
    __shared__ __attribute__((aligned(16))) float sK[BN ? 3 : 1][2][CC];      // BN: [ka | kb | kc][sym | con half of dY][channel of the chunk]
    extern __shared__ __attribute__((aligned(16))) float sAggL[];
    const int tid = threadIdx.x;
    const int fb = blockIdx.x / nchunk, ch = blockIdx.x - fb * nchunk;
    const Pat ps = make_pat(pat_sym, J, pat_sym[1]), pc = make_pat(pat_con, J, pat_con[1]);
    const int nnz_s = ps.nnz, nnz_c = pc.nnz, nnz_t = nnz_s + nnz_c;
    float* const sAs = sAggL;
    float* const sAc = sAs + (nnz_s + 1) * CC;
    float* const sDY = sAc + (nnz_c + 1) * CC;
    float* const sH = sDY + U * J * 2 * CC;
    const int c0 = ch * CC;
    for (int t = tid; t < (nnz_s + 1) * CC4; t += 256) {
        const int k = t / CC4, q = t - k * CC4;
        *(float4*)(sAs + k * CC + q * 4) = c0 + q * 4 < C ? *(const float4*)(A_sym + (long)k * C + c0 + q * 4) : make_float4(0, 0, 0, 0);
    }
    for (int t = tid; t < (nnz_c + 1) * CC4; t += 256) {
        const int k = t / CC4, q = t - k * CC4;
        *(float4*)(sAc + k * CC + q * 4) = c0 + q * 4 < C ? *(const float4*)(A_con + (long)k * C + c0 + q * 4) : make_float4(0, 0, 0, 0);
    }
    for (int t = tid; t < 2 * J; t += 256) {
        const Pat& p = t < J ? ps : pc;
        const int i = t < J ? t : t - J, base = t < J ? 0 : nnz_s;
        for (int k = p.row_ptr[i]; k < p.row_ptr[i + 1]; ++k) { s_ei[base + k] = i; s_ej[base + k] = p.col[k]; }
    }
    if (BN) {
        for (int t = tid; t < 3 * 2 * CC; t += 256) {
            const int which = t / (2 * CC), g = (t / CC) & 1, cl = t % CC;
            const float* src = which == 0 ? ka : which == 1 ? kb : kc;
            sK[which][g][cl] = c0 + cl < C ? src[g * C + c0 + cl] : 0.f;
        }
    }
    __syncthreads();
    // phase-B ownership: pairs tid + 256 q
    int pdy[NPB], ph[NPB];        // float offsets of the pair's dY / H values inside frame 0 of the tiles (-1: no pair)
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        const int p = tid + 256 * q;
        pdy[q] = -1; ph[q] = 0;
        if (p < nnz_t * CC4) {
            const int k = p / CC4, c4 = p - k * CC4;
            const int g = k < nnz_s ? 0 : 1, i = s_ei[k], j = s_ej[k];
            pdy[q] = (i * 2 + g) * CC + c4 * 4;
            ph[q] = (j * 4 + g * 2 + (i == j ? 0 : 1)) * CC + c4 * 4;
        }
    }
    float4 acc[NPB];
#pragma unroll
    for (int q = 0; q < NPB; ++q) acc[q] = make_float4(0, 0, 0, 0);

    // element t of a tile group: (frame row rj = u * J + j, tensor slice w: 0, 1 = dY sym / con; 2 .. 5 = H h0 / h1 sym, h0 / h1 con, 4 channels);
    // the NEXT group's elements are requested into registers before the current group is processed
    constexpr int NPRE = (U * JMAX * 6 * CC4 + 255) / 256;
    float4 pre[NPRE];
    float4 prey[BN ? NPRE : 1];        // BN: the Y values next to the dY slots
    unsigned bnmask = 0;               // BN: slots that hold a real dY element (a frame past F stays zero: kc must not leak in)
    auto request = [&](int f0) {
        bnmask = 0;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int t = tid + 256 * q;
            pre[q] = make_float4(0, 0, 0, 0);
            if (BN) prey[q] = make_float4(0, 0, 0, 0);
            if (t < U * J * 6 * CC4) {
                const int c4 = t % CC4, r = t / CC4;
                const int w = r % 6, rj = r / 6;
                const int u = rj / J, j = rj - u * J;
                const long f = f0 + u;
                const int c = c0 + c4 * 4;
                if (f < F && c < C) {
                    pre[q] = w < 2 ? ld4(dY + (f * J + j) * ldy + w * C + c) : ld4(H + (f * J + j) * ldh + (w - 2) * C + c);
                    if (BN && w < 2) { prey[q] = ld4(Yp + (f * J + j) * ldyp + w * C + c); bnmask |= 1u << q; }
                }
            }
        }
    };
    request(fb * U);
    for (int f0 = fb * U; f0 < F; f0 += nfb * U) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int t = tid + 256 * q;
            if (t < U * J * 6 * CC4) {
                const int c4 = t % CC4, r = t / CC4;
                const int w = r % 6, rj = r / 6;
                if (w < 2) {
                    float4 d = pre[q];
                    if (BN && (bnmask >> q & 1u)) {
                        const float4 a = *(const float4*)&sK[0][w][c4 * 4], b = *(const float4*)&sK[BN ? 1 : 0][w][c4 * 4],
                                     k = *(const float4*)&sK[BN ? 2 : 0][w][c4 * 4], x = prey[q];
                        d.x = fmaf(a.x, d.x, fmaf(b.x, x.x, k.x));
                        d.y = fmaf(a.y, d.y, fmaf(b.y, x.y, k.y));
                        d.z = fmaf(a.z, d.z, fmaf(b.z, x.z, k.z));
                        d.w = fmaf(a.w, d.w, fmaf(b.w, x.w, k.w));
                        d = rnd4(d, (const T*)nullptr);          // (16-bit storage: the stand-alone pass rounds what it stores)
                    }
                    *(float4*)(sDY + (rj * 2 + w) * CC + c4 * 4) = d;
                }
                else *(float4*)(sH + (rj * 4 + (w - 2)) * CC + c4 * 4) = pre[q];
            }
        }
        __syncthreads();
        request(f0 + nfb * U);          // (past the last frame: zeros, no traffic)
        // phase A
        auto cols = [&](auto dtag, const Pat& p, const float* sA, int g) {
            constexpr int D = decltype(dtag)::value;
            for (int t = tid; t < U * J * CC4; t += 256) {
                const int c4 = t % CC4, rj = t / CC4;
                const int u = rj / J, j = rj - u * J;
                const long f = f0 + u;
                const int c = c0 + c4 * 4;
                if (f >= F || c >= C) continue;
                float4 d0 = make_float4(0, 0, 0, 0), d1 = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int ii = p.ell_ci[j * D + d], kk = p.ell_ck[j * D + d];
                    const float4 av = *(const float4*)(sA + kk * CC + c4 * 4);
                    const float4 dv = *(const float4*)(sDY + ((u * J + ii) * 2 + g) * CC + c4 * 4);
                    if (ii == j) d0 = fma4(av, dv, d0); else d1 = fma4(av, dv, d1);
                }
                T* o = dH + (f * J + j) * lddh + g * 2 * C + c;
                st4(o, d0);
                st4(o + C, d1);
            }
        };
        cols(std::integral_constant<int, DS>(), ps, sAs, 0);
        cols(std::integral_constant<int, DC>(), pc, sAc, 1);
        // phase B (frames past F were staged as zeros)
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            if (pdy[q] < 0) continue;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 dv = *(const float4*)(sDY + u * J * 2 * CC + pdy[q]);
                const float4 hv = *(const float4*)(sH + u * J * 4 * CC + ph[q]);
                acc[q] = fma4(dv, hv, acc[q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        const int p = tid + 256 * q;
        if (p >= nnz_t * CC4) continue;
        const int k = p / CC4, c4 = p - k * CC4;
        const int cg = c0 + c4 * 4;
        if (cg < C) *(float4*)(part + ((long)fb * nnz_t + k) * C + cg) = acc[q];
    }
}

// backward: dH[:, 0:4C] and dA.
// Block = (frame block fb, channel chunk of CC <= 128 channels); thread = (frame slot, 4 channels).  dA[k][c] sums over ALL
// frames: every thread adds its products into a per-block LDS accumulator [nnz_sym+nnz_con][CC] with ds_add_f32
// (conflict-free: consecutive lanes hit consecutive channels), the block then stores one partial row to global memory
// and `reduce_rows_kernel` combines the partial rows.  (The first version issued one global atomic per thread, edge and
// channel: 34 M atomics per launch at B=128, 0.66 ms.)
template <typename T>
__global__ void __launch_bounds__(256) semch_agg_bwd_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ H, int ldh,
                                                            int F, int J, int C,
                                                            const float* __restrict__ A_sym, const int32_t* __restrict__ pat_sym,
                                                            const float* __restrict__ A_con, const int32_t* __restrict__ pat_con,
                                                            T* __restrict__ dH, int lddh, float* __restrict__ part, int nfb,
                                                            int nchunk, int CC, int TPF, int FB) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];   // [nnz_sym + nnz_con][CC]
    const int tid = threadIdx.x;
    const int fb = blockIdx.x / nchunk, ch = blockIdx.x - fb * nchunk;
    const int slot = tid / TPF, ct = tid - slot * TPF;
    const int nnz_s = pat_sym[1], nnz_c = pat_con[1], nnz_t = nnz_s + nnz_c;
    for (int t = tid; t < nnz_t * CC; t += 256) sacc[t] = 0.f;
    __syncthreads();
    const int cl = ct * 4;               // channel inside the chunk
    const int c = ch * CC + cl;          // global channel
    if (slot < FB && cl < CC && c < C) {
        for (int f = fb * FB + slot; f < F; f += nfb * FB) {
            const T* dYf = dY + (long)f * J * ldy;
            const T* Hf = H + (long)f * J * ldh;
            T* dHf = dH + (long)f * J * lddh;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int32_t* pat = g == 0 ? pat_sym : pat_con;
                const float* A = g == 0 ? A_sym : A_con;
                const Pat p = make_pat(pat, J, pat[1]);
                const int h0c = g * 2 * C + c, h1c = g * 2 * C + C + c;
                const int yc = g * C + c;
                float* acc = sacc + (g == 0 ? 0 : nnz_s) * CC + cl;
                for (int j = 0; j < J; ++j) {
                    float4 d0 = make_float4(0, 0, 0, 0), d1 = make_float4(0, 0, 0, 0);
                    for (int q = p.col_ptr[j]; q < p.col_ptr[j + 1]; ++q) {
                        int i = p.crow[q], k = p.cedge[q];
                        float4 av = *(const float4*)(A + (long)k * C + c);
                        float4 dv = ld4(dYf + (long)i * ldy + yc);
                        float4 hv = ld4(Hf + (long)j * ldh + (i == j ? h0c : h1c));
                        if (i == j) d0 = fma4(av, dv, d0); else d1 = fma4(av, dv, d1);
                        float* a4 = acc + k * CC;
                        atomicAdd(a4, dv.x * hv.x); atomicAdd(a4 + 1, dv.y * hv.y);
                        atomicAdd(a4 + 2, dv.z * hv.z); atomicAdd(a4 + 3, dv.w * hv.w);
                    }
                    st4(dHf + (long)j * lddh + h0c, d0);
                    st4(dHf + (long)j * lddh + h1c, d1);
                }
            }
        }
    }
    __syncthreads();
    // one partial row per frame block: part[fb][k][c]
    for (int t = tid; t < nnz_t * CC; t += 256) {
        int k = t / CC, cc = t - k * CC;
        int cg = ch * CC + cc;
        if (cg < C) part[((long)fb * nnz_t + k) * C + cg] = sacc[t];
    }
}

// out[n] = sum_r part[r][n]   (256 threads = 32 columns x 8 row lanes)
__global__ void __launch_bounds__(256) reduce_rows_kernel(const float* __restrict__ part, int nrow, long ncol, float* __restrict__ out) {
    __shared__ float sred[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const long n = (long)blockIdx.x * 32 + cx;
    float a = 0.f;
    if (n < ncol) {
#pragma unroll 4
        for (int r = ry; r < nrow; r += 8) a += part[(long)r * ncol + n];
    }
    sred[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && n < ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sred[r][cx];
        out[n] = t;
    }
}

// ------------------------------------------------------------------------------------------------ global attention
// Work unit = (frame f, head h).  A block owns one head (h = blockIdx.x % nheads) and walks frames UB at a time.
constexpr int UB = 4;       // max frames per block iteration (runtime `ub` <= UB, chosen so the g tiles fit in LDS)
constexpr int JP = 20;      // padded row length of the J x J tiles (J <= 19 -> multiple of 4 floats, 16-byte rows)

// att[i][j] = softmax_j(leaky(a_i + c_j)) (+ C_k); also returns p (softmax part) and the LeakyReLU slope mask
template <typename T>
__device__ __forceinline__ void attn_rows(const T* __restrict__ AC, int ldac, const float* __restrict__ Ck, int F, int J, int nheads, int h,
                                          int fbase, int ub, float (*sp)[JMAX][JP], float (*satt)[JMAX][JP], float (*sslope)[JMAX][JP],
                                          float (*sa)[JMAX], float (*sc)[JMAX]) {
    const int tid = threadIdx.x;
    // stage a_i, c_j of the UB frames
    for (int t = tid; t < ub * J; t += blockDim.x) {
        int u = t / J, i = t - u * J;
        int f = fbase + u;
        float av = 0.f, cv = 0.f;
        if (f < F) {
            av = Elem<T>::ld(AC + ((long)f * J + i) * ldac + h);
            cv = Elem<T>::ld(AC + ((long)f * J + i) * ldac + nheads + h);
        }
        sa[u][i] = av;
        sc[u][i] = cv;
    }
    __syncthreads();
    for (int t = tid; t < ub * J; t += blockDim.x) {
        int u = t / J, i = t - u * J;
        float a = sa[u][i];
        float mx = -3.0e38f;
        for (int j = 0; j < J; ++j) {
            float s = a + sc[u][j];
            s = s > 0.f ? s : 0.2f * s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < J; ++j) {
            float s = a + sc[u][j];
            float sl = s > 0.f ? 1.f : 0.2f;
            s *= sl;
            float ex = expf(s - mx);
            sp[u][i][j] = ex;
            if (sslope) sslope[u][i][j] = sl;
            sum += ex;
        }
        float inv = 1.f / sum;
        for (int j = 0; j < J; ++j) {
            float pv = sp[u][i][j] * inv;
            sp[u][i][j] = pv;
            satt[u][i][j] = pv + Ck[((long)h * J + i) * J + j];
        }
    }
    __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ AC, int ldac,
                                                       const float* __restrict__ Ck, int F, int J, int C, int nheads,
                                                       T* __restrict__ Y, int ldy, int ub) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ci = C / nheads;
    const int cpr = (Ci + 3) / 4;
    const int GS = cpr * 4 + 4;  // LDS row stride of the g tile (floats): 16-byte rows, +16 B pad against bank conflicts
    float (*sp)[JMAX][JP] = (float (*)[JMAX][JP])smem;
    float (*satt)[JMAX][JP] = (float (*)[JMAX][JP])(smem + UB * JMAX * JP);
    float (*sa)[JMAX] = (float (*)[JMAX])(smem + 2 * UB * JMAX * JP);
    float (*scc)[JMAX] = (float (*)[JMAX])(smem + 2 * UB * JMAX * JP + UB * JMAX);
    float* sg = smem + 2 * UB * JMAX * JP + 2 * UB * JMAX + 8;  // [UB][J][GS]
    const int tid = threadIdx.x;
    const int h = blockIdx.x % nheads;
    const int bstride = gridDim.x / nheads;
    for (int fbase = (blockIdx.x / nheads) * ub; fbase < F; fbase += bstride * ub) {
        attn_rows<T>(AC, ldac, Ck, F, J, nheads, h, fbase, ub, sp, satt, nullptr, sa, scc);
        // stage g_h of the ub frames: [u][j][c]
        const int nchunk = ub * J * cpr;
        for (int t = tid; t < nchunk; t += blockDim.x) {
            int c4 = t % cpr, rj = t / cpr;
            int u = rj / J, j = rj - u * J;
            int f = fbase + u;
            float v[4] = {0, 0, 0, 0};
            if (f < F) {
                const T* src = G + ((long)f * J + j) * ldg + h * Ci + c4 * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (c4 * 4 + q < Ci) v[q] = Elem<T>::ld(src + q);
            }
            float* d = sg + ((long)u * J + j) * GS + c4 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = v[q];
        }
        __syncthreads();
        // y[u][i][c4] = sum_j att[u][i][j] * g[u][j][c4]
        for (int t = tid; t < nchunk; t += blockDim.x) {
            int c4 = t % cpr, ri = t / cpr;
            int u = ri / J, i = ri - u * J;
            int f = fbase + u;
            if (f >= F) continue;
            float4 acc = make_float4(0, 0, 0, 0);
            for (int j = 0; j < J; ++j) {
                float a = satt[u][i][j];
                float4 gv = *(const float4*)(sg + ((long)u * J + j) * GS + c4 * 4);
                acc.x = fmaf(a, gv.x, acc.x); acc.y = fmaf(a, gv.y, acc.y);
                acc.z = fmaf(a, gv.z, acc.z); acc.w = fmaf(a, gv.w, acc.w);
            }
            T* dst = Y + ((long)f * J + i) * ldy + h * Ci + c4 * 4;
            float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c4 * 4 + q < Ci) Elem<T>::st(dst + q, o[q]);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_kernel(const T* __restrict__ dY, int lddy, const T* __restrict__ G, int ldg,
                                                       const T* __restrict__ AC, int ldac, const float* __restrict__ Ck,
                                                       int F, int J, int C, int nheads, T* __restrict__ dG, int lddg,
                                                       T* __restrict__ dAC, int lddac, float* __restrict__ dCk,
                                                       float* __restrict__ dbias_ac, int ub) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ci = C / nheads;
    const int cpr = (Ci + 3) / 4;
    const int GS = cpr * 4 + 4;
    const int TILE = UB * JMAX * JP;
    float (*sp)[JMAX][JP] = (float (*)[JMAX][JP])smem;
    float (*satt)[JMAX][JP] = (float (*)[JMAX][JP])(smem + TILE);
    float (*sslope)[JMAX][JP] = (float (*)[JMAX][JP])(smem + 2 * TILE);
    float (*sdat)[JMAX][JP] = (float (*)[JMAX][JP])(smem + 3 * TILE);
    float (*sa)[JMAX] = (float (*)[JMAX])(smem + 4 * TILE);
    float (*scc)[JMAX] = (float (*)[JMAX])(smem + 4 * TILE + UB * JMAX);
    float* sg = smem + 4 * TILE + 2 * UB * JMAX + 8;   // [UB][J][GS]
    float* sdy = sg + ub * J * GS;                     // [ub][J][GS]
    const int tid = threadIdx.x;
    const int h = blockIdx.x % nheads;
    const int bstride = gridDim.x / nheads;
    const int nchunk = ub * J * cpr;
    // per-thread accumulators of dC_k[h][i][j] for entries t = tid, tid + 256 (J*J <= 361 < 512)
    float ck_acc0 = 0.f, ck_acc1 = 0.f;
    float da_sum = 0.f, dc_sum = 0.f;     // fp32 sums of da / dc over this thread's rows (bias gradients of theta / phi)
    for (int fbase = (blockIdx.x / nheads) * ub; fbase < F; fbase += bstride * ub) {
        attn_rows<T>(AC, ldac, Ck, F, J, nheads, h, fbase, ub, sp, satt, sslope, sa, scc);
        for (int t = tid; t < nchunk; t += blockDim.x) {
            int c4 = t % cpr, rj = t / cpr;
            int u = rj / J, j = rj - u * J;
            int f = fbase + u;
            float v[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
            if (f < F) {
                const T* src = G + ((long)f * J + j) * ldg + h * Ci + c4 * 4;
                const T* dsrc = dY + ((long)f * J + j) * lddy + h * Ci + c4 * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c4 * 4 + q < Ci) { v[q] = Elem<T>::ld(src + q); d[q] = Elem<T>::ld(dsrc + q); }
            }
            float* pg = sg + ((long)u * J + j) * GS + c4 * 4;
            float* pd = sdy + ((long)u * J + j) * GS + c4 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) { pg[q] = v[q]; pd[q] = d[q]; }
        }
        __syncthreads();
        // datt[u][i][j] = sum_c dy[u][i][c] * g[u][j][c]
        for (int t = tid; t < ub * J * J; t += blockDim.x) {
            int u = t / (J * J), r = t - u * J * J;
            int i = r / J, j = r - i * J;
            const float* pd = sdy + ((long)u * J + i) * GS;
            const float* pg = sg + ((long)u * J + j) * GS;
            float acc = 0.f;
            for (int c4 = 0; c4 < cpr; ++c4) {
                float4 a = *(const float4*)(pd + c4 * 4);
                float4 b = *(const float4*)(pg + c4 * 4);
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
            sdat[u][i][j] = acc;
        }
        __syncthreads();
        // dC_k accumulation (sum over the frames of this iteration)
        {
            int t = tid;
            if (t < J * J) {
                int i = t / J, j = t - i * J;
                for (int u = 0; u < ub; ++u) if (fbase + u < F) ck_acc0 += sdat[u][i][j];
            }
            t = tid + 256;
            if (t < J * J) {
                int i = t / J, j = t - i * J;
                for (int u = 0; u < ub; ++u) if (fbase + u < F) ck_acc1 += sdat[u][i][j];
            }
        }
        // dg[u][j][c4] = sum_i att[u][i][j] * dy[u][i][c4]
        for (int t = tid; t < nchunk; t += blockDim.x) {
            int c4 = t % cpr, rj = t / cpr;
            int u = rj / J, j = rj - u * J;
            int f = fbase + u;
            if (f >= F) continue;
            float4 acc = make_float4(0, 0, 0, 0);
            for (int i = 0; i < J; ++i) {
                float a = satt[u][i][j];
                float4 dv = *(const float4*)(sdy + ((long)u * J + i) * GS + c4 * 4);
                acc.x = fmaf(a, dv.x, acc.x); acc.y = fmaf(a, dv.y, acc.y);
                acc.z = fmaf(a, dv.z, acc.z); acc.w = fmaf(a, dv.w, acc.w);
            }
            T* dst = dG + ((long)f * J + j) * lddg + h * Ci + c4 * 4;
            float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c4 * 4 + q < Ci) Elem<T>::st(dst + q, o[q]);
        }
        // softmax + LeakyReLU backward, row-wise: ds[i][j] = p_ij (datt_ij - sum_j' p_ij' datt_ij') * slope_ij
        for (int t = tid; t < ub * J; t += blockDim.x) {
            int u = t / J, i = t - u * J;
            float dot = 0.f;
            for (int j = 0; j < J; ++j) dot += sp[u][i][j] * sdat[u][i][j];
            float da = 0.f;
            for (int j = 0; j < J; ++j) {
                float ds = sp[u][i][j] * (sdat[u][i][j] - dot) * sslope[u][i][j];
                sslope[u][i][j] = ds;   // reuse the slope tile for ds
                da += ds;
            }
            int f = fbase + u;
            if (f < F) { Elem<T>::st(dAC + ((long)f * J + i) * lddac + h, da); da_sum += da; }
        }
        __syncthreads();
        for (int t = tid; t < ub * J; t += blockDim.x) {
            int u = t / J, j = t - u * J;
            float dc = 0.f;
            for (int i = 0; i < J; ++i) dc += sslope[u][i][j];
            int f = fbase + u;
            if (f < F) { Elem<T>::st(dAC + ((long)f * J + j) * lddac + nheads + h, dc); dc_sum += dc; }
        }
        __syncthreads();
    }
    if (tid < J * J) atomicAdd(dCk + (long)h * J * J + tid, ck_acc0);
    if (tid + 256 < J * J) atomicAdd(dCk + (long)h * J * J + tid + 256, ck_acc1);
    if (dbias_ac) {
        // only threads < ub*J hold non-zero sums; wave-level reductions, the block's waves combined in a fixed order, ONE atomic per
        // block and address (with one block per head -- GAST_DETERMINISTIC -- the result does not depend on scheduling)
        __shared__ float swave[8][2];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { da_sum += __shfl_xor(da_sum, off); dc_sum += __shfl_xor(dc_sum, off); }
        if ((tid & 63) == 0) { swave[tid >> 6][0] = da_sum; swave[tid >> 6][1] = dc_sum; }
        __syncthreads();
        if (tid < 2) {
            float v = 0.f;
            for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += swave[wv][tid];
            atomicAdd(dbias_ac + (tid ? nheads : 0) + h, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ global attention, wave-per-unit
// One WAVE owns a (frame, head) unit at a time and never meets a block barrier: the J x J tiles and the fp32 copies of the
// g / dy tiles live in a per-wave LDS region, the phases of a unit are separated by wave-local waits only, and the four waves
// of a block drift freely against each other.  16-byte (fp32) / 8-byte (bf16) global accesses.  (The block-per-head kernels
// above -- kept for head widths other than 32/64/128 channels -- spent 83 us on 60 MB: 2-byte loads, ~8 block barriers per
// 4 frames, 68 of 256 threads busy in the row phases.)
// A wave keeps ONE head for its whole life (the wave count is a multiple of nheads), so the C_k row of a lane, the dC_k
// accumulators and the bias-gradient sums stay in registers across units; they leave through a per-wave partial row
// [C | 2*nheads | nheads*J*J] that attn_bwd_finish_kernel reduces without atomics.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// unroll factor of the two J-long product loops of the JT kernels (full unrolling hoists all 17 row loads: 220 - 350 VGPRs in the backward)
#ifndef GAST_ATTN_PROD_UNROLL
#define GAST_ATTN_PROD_UNROLL 4
#endif
template <int CI4> struct AttnW {
    static constexpr int CI = CI4 * 4, GS = CI + 4, RG = 64 / CI4, NR = (JMAX + RG - 1) / RG;
    static constexpr int FWD_FLOATS = JMAX * JP + JMAX * GS + 32;
    static constexpr int BWD_FLOATS = 4 * JMAX * JP + 2 * JMAX * GS + 32;
};

// rows of att = softmax_j(leaky(a_i + c_j)) + C_k for lane i < J; optionally p and the LeakyReLU slopes.
// JT = the skeleton's joint count as a compile-time constant (15 / 17 / 19: the shipped skeletons; 0 = run-time J).  A unit is ONE
// wave's dependent instruction chain, so its length is the kernel's time: with JT the 17-lane row phases are straight-line code --
// the exponentials stay in registers between the two passes, the row of c_j is read with 16-byte LDS loads, no loop branches.
template <int JT>
__device__ __forceinline__ void attn_row(int lane, int J, float a_i, const float* __restrict__ sc, const float* ckrow,
                                         float (*satt)[JP], float (*sp)[JP], float (*sslope)[JP]) {
    if constexpr (JT > 0) {
        if (lane < JT) {
            constexpr int J4 = (JT + 3) / 4;
            float s[J4 * 4], sl[J4 * 4];
#pragma unroll
            for (int q = 0; q < J4; ++q) {      // (sc has 32 floats: the tail of the last quad is finite garbage that is never used)
                const float4 c4 = *(const float4*)(sc + 4 * q);
                s[4 * q] = c4.x; s[4 * q + 1] = c4.y; s[4 * q + 2] = c4.z; s[4 * q + 3] = c4.w;
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const float v = a_i + s[j];
                sl[j] = v > 0.f ? 1.f : 0.2f;
                s[j] = v * sl[j];
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                s[j] = expf(s[j] - mx);
                sum += s[j];
            }
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = JT; j < J4 * 4; ++j) { s[j] = 0.f; sl[j] = 0.f; }
            // rows are JP = 20 floats (80 B, 16-byte aligned): whole quads; columns JT .. 4 J4 - 1 of a tile are padding nobody reads
#pragma unroll
            for (int q = 0; q < J4; ++q) {
                const float4 pv = make_float4(s[4 * q] * inv, s[4 * q + 1] * inv, s[4 * q + 2] * inv, s[4 * q + 3] * inv);
                if (sp) *(float4*)(&sp[lane][4 * q]) = pv;
                if (sslope) *(float4*)(&sslope[lane][4 * q]) = make_float4(sl[4 * q], sl[4 * q + 1], sl[4 * q + 2], sl[4 * q + 3]);
                *(float4*)(&satt[lane][4 * q]) = make_float4(pv.x + ckrow[4 * q], pv.y + ckrow[4 * q + 1], pv.z + ckrow[4 * q + 2],
                                                             4 * q + 3 < JMAX ? pv.w + ckrow[4 * q + 3 < JMAX ? 4 * q + 3 : 0] : 0.f);
            }
        }
        return;
    }
    if (lane < J) {
        float mx = -3.0e38f;
        for (int j = 0; j < J; ++j) {
            float s = a_i + sc[j];
            s = s > 0.f ? s : 0.2f * s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < J; ++j) {
            float s = a_i + sc[j];
            const float sl = s > 0.f ? 1.f : 0.2f;
            const float ex = expf(s * sl - mx);
            satt[lane][j] = ex;
            if (sslope) sslope[lane][j] = sl;
            sum += ex;
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (j < J) {
                const float pv = satt[lane][j] * inv;
                if (sp) sp[lane][j] = pv;
                satt[lane][j] = pv + ckrow[j];
            }
        }
    }
}

template <typename T, int CI4, int JT>
__global__ void __launch_bounds__(256) attn_fwd_wave_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ AC, int ldac,
                                                            const float* __restrict__ Ck, int F, int Jrt, int nheads,
                                                            T* __restrict__ Y, int ldy) {
    using W = AttnW<CI4>;
    const int J = JT ? JT : Jrt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* base = smem + w * W::FWD_FLOATS;
    float (*satt)[JP] = (float (*)[JP])base;
    float* sg = base + JMAX * JP;
    float* sc = sg + JMAX * W::GS;
    const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
    const int h = gw % nheads;
    float ckrow[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) ckrow[j] = (lane < J && j < J) ? Ck[((long)h * J + lane) * J + j] : 0.f;
    const int rg = lane / CI4, c4 = lane - rg * CI4;
    // the next unit's tiles are requested before this unit's dependent chain starts (registers -> LDS at the top of the next trip)
    constexpr int NPF = (JMAX * CI4 + 63) / 64;
    float4 pg[NPF];
    float pa = 0.f, pc = 0.f;
    auto prefetch = [&](int u) {
        const int f = u / nheads;
        if (lane < J) {
            const T* acp = AC + ((long)f * J + lane) * ldac;
            pa = Elem<T>::ld(acp + h);
            pc = Elem<T>::ld(acp + nheads + h);
        }
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int t = lane + 64 * q, j = t / CI4, cc = t - j * CI4;
            if (t < J * CI4) pg[q] = ld4(G + ((long)f * J + j) * ldg + h * W::CI + cc * 4);
        }
    };
    // (only where the look-ahead registers cost no occupancy: at 32-channel heads they cross 168 VGPRs -- a resident block per CU, 53 -> 73 us
    // in the backward; at 128-channel heads, 308 VGPRs, they would spill)
    constexpr bool LOOKAHEAD = CI4 == 16;
    if (LOOKAHEAD && gw < F * nheads) prefetch(gw);
    for (int u = gw; u < F * nheads; u += nw) {
        const int f = u / nheads;
        float a_i = 0.f;
        if constexpr (LOOKAHEAD) {
            a_i = lane < J ? pa : 0.f;
            if (lane < J) sc[lane] = pc;
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int t = lane + 64 * q, j = t / CI4, cc = t - j * CI4;
                if (t < J * CI4) *(float4*)(sg + j * W::GS + cc * 4) = pg[q];
            }
            if (u + nw < F * nheads) prefetch(u + nw);
        } else {
            if (lane < J) {
                const T* acp = AC + ((long)f * J + lane) * ldac;
                a_i = Elem<T>::ld(acp + h);
                sc[lane] = Elem<T>::ld(acp + nheads + h);
            }
            for (int t = lane; t < J * CI4; t += 64) {
                const int j = t / CI4, cc = t - j * CI4;
                *(float4*)(sg + j * W::GS + cc * 4) = ld4(G + ((long)f * J + j) * ldg + h * W::CI + cc * 4);
            }
        }
        wave_lds_sync();
        attn_row<JT>(lane, J, a_i, sc, ckrow, satt, nullptr, nullptr);
        wave_lds_sync();
        float4 acc[W::NR];
#pragma unroll
        for (int q = 0; q < W::NR; ++q) acc[q] = make_float4(0, 0, 0, 0);
#pragma unroll(JT ? GAST_ATTN_PROD_UNROLL : 1)
        for (int j = 0; j < J; ++j) {
            const float4 gv = *(const float4*)(sg + j * W::GS + c4 * 4);
#pragma unroll
            for (int q = 0; q < W::NR; ++q) {
                const float a = satt[rg + W::RG * q][j];      // rows >= J: never stored below
                acc[q].x = fmaf(a, gv.x, acc[q].x); acc[q].y = fmaf(a, gv.y, acc[q].y);
                acc[q].z = fmaf(a, gv.z, acc[q].z); acc[q].w = fmaf(a, gv.w, acc[q].w);
            }
        }
#pragma unroll
        for (int q = 0; q < W::NR; ++q) {
            const int i = rg + W::RG * q;
            if (i < J) st4(Y + ((long)f * J + i) * ldy + h * W::CI + c4 * 4, acc[q]);
        }
        wave_lds_sync();
    }
}

template <typename T, int CI4, int JT>
__global__ void __launch_bounds__(256) attn_bwd_wave_kernel(const T* __restrict__ dY, int lddy, const T* __restrict__ G, int ldg,
                                                            const T* __restrict__ AC, int ldac, const float* __restrict__ Ck,
                                                            int F, int Jrt, int nheads, T* __restrict__ dG, int lddg,
                                                            T* __restrict__ dAC, int lddac, float* __restrict__ ws, int ncol) {
    using W = AttnW<CI4>;
    const int J = JT ? JT : Jrt;
    constexpr int NP = (JMAX * JMAX + 63) / 64;      // (i, j) pairs per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* base = smem + w * W::BWD_FLOATS;
    float (*satt)[JP] = (float (*)[JP])base;
    float (*sp)[JP] = (float (*)[JP])(base + JMAX * JP);
    float (*sds)[JP] = (float (*)[JP])(base + 2 * JMAX * JP);     // LeakyReLU slopes, then ds
    float (*sdat)[JP] = (float (*)[JP])(base + 3 * JMAX * JP);
    float* sg = base + 4 * JMAX * JP;
    float* sdy = sg + JMAX * W::GS;
    float* sc = sdy + JMAX * W::GS;
    const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
    const int h = gw % nheads;
    const int C = nheads * W::CI;
    float ckrow[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) ckrow[j] = (lane < J && j < J) ? Ck[((long)h * J + lane) * J + j] : 0.f;
    const int rg = lane / CI4, c4 = lane - rg * CI4;
    float ckacc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) ckacc[q] = 0.f;
    float4 gsum = make_float4(0, 0, 0, 0);
    float da_sum = 0.f, dc_sum = 0.f;
    // the next unit's tiles are requested before this unit's dependent chain starts (registers -> LDS at the top of the next trip)
    constexpr int NPF = (JMAX * CI4 + 63) / 64;
    float4 pg[NPF], pdy[NPF];
    float pa = 0.f, pc = 0.f;
    auto prefetch = [&](int u) {
        const int f = u / nheads;
        if (lane < J) {
            const T* acp = AC + ((long)f * J + lane) * ldac;
            pa = Elem<T>::ld(acp + h);
            pc = Elem<T>::ld(acp + nheads + h);
        }
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int t = lane + 64 * q, j = t / CI4, cc = t - j * CI4;
            if (t < J * CI4) {
                pg[q] = ld4(G + ((long)f * J + j) * ldg + h * W::CI + cc * 4);
                pdy[q] = ld4(dY + ((long)f * J + j) * lddy + h * W::CI + cc * 4);
            }
        }
    };
    // (only where the look-ahead registers cost no occupancy: at 32-channel heads they cross 168 VGPRs -- a resident block per CU, 53 -> 73 us
    // in the backward; at 128-channel heads, 308 VGPRs, they would spill)
    constexpr bool LOOKAHEAD = CI4 == 16;
    if (LOOKAHEAD && gw < F * nheads) prefetch(gw);
    for (int u = gw; u < F * nheads; u += nw) {
        const int f = u / nheads;
        float a_i = 0.f;
        if constexpr (LOOKAHEAD) {
            a_i = lane < J ? pa : 0.f;
            if (lane < J) sc[lane] = pc;
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int t = lane + 64 * q, j = t / CI4, cc = t - j * CI4;
                if (t < J * CI4) {
                    *(float4*)(sg + j * W::GS + cc * 4) = pg[q];
                    *(float4*)(sdy + j * W::GS + cc * 4) = pdy[q];
                }
            }
            if (u + nw < F * nheads) prefetch(u + nw);
        } else {
            if (lane < J) {
                const T* acp = AC + ((long)f * J + lane) * ldac;
                a_i = Elem<T>::ld(acp + h);
                sc[lane] = Elem<T>::ld(acp + nheads + h);
            }
            for (int t = lane; t < J * CI4; t += 64) {
                const int j = t / CI4, cc = t - j * CI4;
                *(float4*)(sg + j * W::GS + cc * 4) = ld4(G + ((long)f * J + j) * ldg + h * W::CI + cc * 4);
                *(float4*)(sdy + j * W::GS + cc * 4) = ld4(dY + ((long)f * J + j) * lddy + h * W::CI + cc * 4);
            }
        }
        wave_lds_sync();
        attn_row<JT>(lane, J, a_i, sc, ckrow, satt, sp, sds);
        // datt[i][j] = sum_c dy[i][c] * g[j][c]   (pairs t = lane + 64 q)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int t = lane + 64 * q;
            if (t < J * J) {
                const int i = t / J, j = t - i * J;
                const float* pd = sdy + i * W::GS;
                const float* pg = sg + j * W::GS;
                float acc = 0.f;
#pragma unroll 4
                for (int cc = 0; cc < CI4; ++cc) {
                    const float4 x = *(const float4*)(pd + cc * 4);
                    const float4 y = *(const float4*)(pg + cc * 4);
                    acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
                }
                sdat[i][j] = acc;
                ckacc[q] += acc;
            }
        }
        wave_lds_sync();
        // softmax + LeakyReLU backward, row i = lane: ds_ij = p_ij (datt_ij - <p_i, datt_i>) slope_ij;  da_i = sum_j ds_ij
        if constexpr (JT > 0) {
            // straight-line row phase: the rows of p / datt / slope come in with 16-byte loads, two partial sums break the FMA chain
            if (lane < JT) {
                constexpr int J4 = (JT + 3) / 4;
                float pv[J4 * 4], dv[J4 * 4], sv[J4 * 4];
#pragma unroll
                for (int q = 0; q < J4; ++q) {
                    const float4 a4 = *(const float4*)(&sp[lane][4 * q]), b4 = *(const float4*)(&sdat[lane][4 * q]), c4v = *(const float4*)(&sds[lane][4 * q]);
                    pv[4 * q] = a4.x; pv[4 * q + 1] = a4.y; pv[4 * q + 2] = a4.z; pv[4 * q + 3] = a4.w;
                    dv[4 * q] = b4.x; dv[4 * q + 1] = b4.y; dv[4 * q + 2] = b4.z; dv[4 * q + 3] = b4.w;
                    sv[4 * q] = c4v.x; sv[4 * q + 1] = c4v.y; sv[4 * q + 2] = c4v.z; sv[4 * q + 3] = c4v.w;
                }
                float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    if (j & 1) dot1 = fmaf(pv[j], dv[j], dot1);
                    else dot0 = fmaf(pv[j], dv[j], dot0);
                }
                const float dot = dot0 + dot1;
                float da0 = 0.f, da1 = 0.f;
#pragma unroll
                for (int j = 0; j < J4 * 4; ++j) {
                    const float ds = j < JT ? pv[j] * (dv[j] - dot) * sv[j] : 0.f;
                    sv[j] = ds;
                    if (j & 1) da1 += ds;
                    else da0 += ds;
                }
#pragma unroll
                for (int q = 0; q < J4; ++q) *(float4*)(&sds[lane][4 * q]) = make_float4(sv[4 * q], sv[4 * q + 1], sv[4 * q + 2], sv[4 * q + 3]);
                const float da = da0 + da1;
                Elem<T>::st(dAC + ((long)f * J + lane) * lddac + h, da);
                da_sum += da;
            }
            wave_lds_sync();
            if (lane < JT) {
                float dc0 = 0.f, dc1 = 0.f;
#pragma unroll
                for (int i = 0; i < JT; ++i) {
                    if (i & 1) dc1 += sds[i][lane];
                    else dc0 += sds[i][lane];
                }
                const float dc = dc0 + dc1;
                Elem<T>::st(dAC + ((long)f * J + lane) * lddac + nheads + h, dc);
                dc_sum += dc;
            }
        } else {
        if (lane < J) {
            float dot = 0.f;
            for (int j = 0; j < J; ++j) dot = fmaf(sp[lane][j], sdat[lane][j], dot);
            float da = 0.f;
            for (int j = 0; j < J; ++j) {
                const float ds = sp[lane][j] * (sdat[lane][j] - dot) * sds[lane][j];
                sds[lane][j] = ds;
                da += ds;
            }
            Elem<T>::st(dAC + ((long)f * J + lane) * lddac + h, da);
            da_sum += da;
        }
        wave_lds_sync();
        if (lane < J) {
            float dc = 0.f;
            for (int i = 0; i < J; ++i) dc += sds[i][lane];
            Elem<T>::st(dAC + ((long)f * J + lane) * lddac + nheads + h, dc);
            dc_sum += dc;
        }
        }
        // dg[j][c] = sum_i att[i][j] * dy[i][c]
        float4 acc[W::NR];
#pragma unroll
        for (int q = 0; q < W::NR; ++q) acc[q] = make_float4(0, 0, 0, 0);
#pragma unroll(JT ? GAST_ATTN_PROD_UNROLL : 1)
        for (int i = 0; i < J; ++i) {
            const float4 dv = *(const float4*)(sdy + i * W::GS + c4 * 4);
#pragma unroll
            for (int q = 0; q < W::NR; ++q) {
                const int jq = rg + W::RG * q;
                const float a = satt[i][jq < JMAX ? jq : 0];
                acc[q].x = fmaf(a, dv.x, acc[q].x); acc[q].y = fmaf(a, dv.y, acc[q].y);
                acc[q].z = fmaf(a, dv.z, acc[q].z); acc[q].w = fmaf(a, dv.w, acc[q].w);
            }
        }
#pragma unroll
        for (int q = 0; q < W::NR; ++q) {
            const int j = rg + W::RG * q;
            if (j < J) {
                st4(dG + ((long)f * J + j) * lddg + h * W::CI + c4 * 4, acc[q]);
                gsum.x += acc[q].x; gsum.y += acc[q].y; gsum.z += acc[q].z; gsum.w += acc[q].w;
            }
        }
        wave_lds_sync();
    }
    // per-wave partial row: [g bias sums (C) | da sums (nheads) | dc sums (nheads) | dC_k (nheads*J*J)]; a wave writes only the
    // slices of its head, the nheads waves that share a row index write disjoint slices
    float* row = ws + (long)(gw / nheads) * ncol;
    for (int off = CI4; off < 64; off <<= 1) {
        gsum.x += __shfl_xor(gsum.x, off); gsum.y += __shfl_xor(gsum.y, off);
        gsum.z += __shfl_xor(gsum.z, off); gsum.w += __shfl_xor(gsum.w, off);
    }
    if (rg == 0) *(float4*)(row + h * W::CI + c4 * 4) = gsum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { da_sum += __shfl_xor(da_sum, off); dc_sum += __shfl_xor(dc_sum, off); }
    if (lane == 0) { row[C + h] = da_sum; row[C + nheads + h] = dc_sum; }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int t = lane + 64 * q;
        if (t < J * J) row[C + 2 * nheads + h * J * J + t] = ckacc[q];
    }
}

// ------------------------------------------------------------------------------------------------ bf16 attention on MFMA
// Same decomposition (one (frame, head) unit per wave, no block barriers), but the three J x J x Ci products run on
// v_mfma_f32_32x32x16_bf16 with J padded to 32:
//   forward   y[i][c]   = sum_j att[i][j] g[j][c]       A = att   (rows i, k = j)   B = g^T  (cols c, k = j)
//   backward  datt[i][j] = sum_c dy[i][c] g[j][c]       A = dy    (rows i, k = c)   B = g    (cols j, k = c)   straight from global
//             dg[j][c]   = sum_i att[i][j] dy[i][c]     A = att^T (rows j, k = i)   B = dy^T (cols c, k = i)
// The row tiles of g / dy are loaded in MFMA-fragment shape (lane = (row, 8-channel half of a 16-channel step)) and the
// operands whose reduction index is the joint are written transposed to LDS with 2-byte stores; the padding rows / columns of
// the LDS tiles are zeroed once and never written.  The VALU versions above spend ~340 (forward) / ~900 (backward) FMA-class
// instructions per lane and unit on these products; att is rounded to bf16 here (the operands of the other two already are).
template <int CI> struct AttnM {
    static constexpr int KS = CI / 16, NT = CI / 32;
    static constexpr int T_BYTES = CI * 64;                                              // transposed operand tile [CI][32] bf16
    static constexpr int FWD_BYTES = T_BYTES + 32 * 64 + 128;                             // g^T | att [32][32] | c_j
    static constexpr int BWD_BYTES = T_BYTES + 32 * 64 + 3 * JMAX * JP * 4 + 128;         // dy^T | att^T | p | slope/ds | datt | c_j
};

__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, f32x16 c) {
    return mfma_h16(a, b, c);       // (the 16-bit storage flavour of the build: common.h)
}
// 8 bf16 of a fragment register -> column `col` of rows c0 .. c0+7 of a [*][32] bf16 tile
__device__ __forceinline__ void scatter8_bf16(bf16_t* __restrict__ tile, int c0, int col, const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        tile[(c0 + 2 * e) * 32 + col] = (bf16_t)(w[e] & 0xffffu);
        tile[(c0 + 2 * e + 1) * 32 + col] = (bf16_t)(w[e] >> 16);
    }
}

template <int CI>
__global__ void __launch_bounds__(256) attn_fwd_mfma_kernel(const bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ AC, int ldac,
                                                            const float* __restrict__ Ck, int F, int J, int nheads,
                                                            bf16_t* __restrict__ Y, int ldy) {
    using M = AttnM<CI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned char* base = smemb + w * M::FWD_BYTES;
    bf16_t* sgT = (bf16_t*)base;                       // [CI][32]: g^T, columns j >= J stay zero
    bf16_t* satb = (bf16_t*)(base + M::T_BYTES);       // [32][32]: att[i][j], rows / columns >= J stay zero
    float* sc = (float*)(base + M::T_BYTES + 2048);
    for (int t = lane; t < (M::T_BYTES + 2048) / 16; t += 64) ((uint4*)base)[t] = make_uint4(0u, 0u, 0u, 0u);
    const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
    const int h = gw % nheads;
    float ckrow[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) ckrow[j] = (lane < J && j < J) ? Ck[((long)h * J + lane) * J + j] : 0.f;
    const int jr = li < J ? li : J - 1;                // source row of this lane's fragments (clamped: rows >= J are never used)
    wave_lds_sync();
    for (int u = gw; u < F * nheads; u += nw) {
        const int f = u / nheads;
        float a_i = 0.f;
        if (lane < J) {
            const bf16_t* acp = AC + ((long)f * J + lane) * ldac;
            a_i = bf2f(acp[h]);
            sc[lane] = bf2f(acp[nheads + h]);
        }
        uint4 gfr[M::KS];
#pragma unroll
        for (int ks = 0; ks < M::KS; ++ks) gfr[ks] = *(const uint4*)(G + ((long)f * J + jr) * ldg + h * CI + ks * 16 + lh * 8);
        if (li < J) {
#pragma unroll
            for (int ks = 0; ks < M::KS; ++ks) scatter8_bf16(sgT, ks * 16 + lh * 8, li, gfr[ks]);
        }
        wave_lds_sync();
        if (lane < J) {        // att row i = lane (same arithmetic as attn_row)
            float mx = -3.0e38f;
            for (int j = 0; j < J; ++j) {
                float sv = a_i + sc[j];
                sv = sv > 0.f ? sv : 0.2f * sv;
                mx = fmaxf(mx, sv);
            }
            float ex[JMAX];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                ex[j] = 0.f;
                if (j < J) {
                    const float sv = a_i + sc[j];
                    const float sl = sv > 0.f ? 1.f : 0.2f;
                    ex[j] = expf(sv * sl - mx);
                    sum += ex[j];
                }
            }
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                if (j < J) satb[lane * 32 + j] = f2bf(ex[j] * inv + ckrow[j]);
        }
        wave_lds_sync();
#pragma unroll
        for (int nt = 0; nt < M::NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                acc = mfma_bf16(*(const uint4*)(satb + li * 32 + k2 * 16 + lh * 8), *(const uint4*)(sgT + (nt * 32 + li) * 32 + k2 * 16 + lh * 8), acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < J) Y[((long)f * J + i) * ldy + h * CI + nt * 32 + li] = f2bf(acc[r]);
            }
        }
        wave_lds_sync();
    }
}

template <int CI>
__global__ void __launch_bounds__(256) attn_bwd_mfma_kernel(const bf16_t* __restrict__ dY, int lddy, const bf16_t* __restrict__ G, int ldg,
                                                            const bf16_t* __restrict__ AC, int ldac, const float* __restrict__ Ck,
                                                            int F, int J, int nheads, bf16_t* __restrict__ dG, int lddg,
                                                            bf16_t* __restrict__ dAC, int lddac, float* __restrict__ ws, int ncol) {
    using M = AttnM<CI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned char* base = smemb + w * M::BWD_BYTES;
    bf16_t* sdyT = (bf16_t*)base;                         // [CI][32]: dy^T, columns i >= J stay zero
    bf16_t* sattT = (bf16_t*)(base + M::T_BYTES);         // [32][32]: att^T[j][i], rows / columns >= J stay zero
    float (*sp)[JP] = (float (*)[JP])(base + M::T_BYTES + 2048);
    float (*sds)[JP] = sp + JMAX;                         // LeakyReLU slopes, then ds
    float (*sdat)[JP] = sds + JMAX;
    float* sc = (float*)(sdat + JMAX);
    for (int t = lane; t < (M::T_BYTES + 2048) / 16; t += 64) ((uint4*)base)[t] = make_uint4(0u, 0u, 0u, 0u);
    const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
    const int h = gw % nheads;
    const int C = nheads * CI;
    float ckrow[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) ckrow[j] = (lane < J && j < J) ? Ck[((long)h * J + lane) * J + j] : 0.f;
    const int jr = li < J ? li : J - 1;
    f32x16 ckacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) ckacc[r] = 0.f;
    float gsum[M::NT];
#pragma unroll
    for (int nt = 0; nt < M::NT; ++nt) gsum[nt] = 0.f;
    float da_sum = 0.f, dc_sum = 0.f;
    wave_lds_sync();
    for (int u = gw; u < F * nheads; u += nw) {
        const int f = u / nheads;
        float a_i = 0.f;
        if (lane < J) {
            const bf16_t* acp = AC + ((long)f * J + lane) * ldac;
            a_i = bf2f(acp[h]);
            sc[lane] = bf2f(acp[nheads + h]);
        }
        uint4 dfr[M::KS], gfr[M::KS];
#pragma unroll
        for (int ks = 0; ks < M::KS; ++ks) {
            dfr[ks] = *(const uint4*)(dY + ((long)f * J + jr) * lddy + h * CI + ks * 16 + lh * 8);
            gfr[ks] = *(const uint4*)(G + ((long)f * J + jr) * ldg + h * CI + ks * 16 + lh * 8);
        }
        if (li < J) {
#pragma unroll
            for (int ks = 0; ks < M::KS; ++ks) scatter8_bf16(sdyT, ks * 16 + lh * 8, li, dfr[ks]);
        }
        wave_lds_sync();
        if (lane < J) {        // row i = lane: p, LeakyReLU slopes, att^T (same arithmetic as attn_row)
            float mx = -3.0e38f;
            for (int j = 0; j < J; ++j) {
                float sv = a_i + sc[j];
                sv = sv > 0.f ? sv : 0.2f * sv;
                mx = fmaxf(mx, sv);
            }
            float ex[JMAX];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                ex[j] = 0.f;
                if (j < J) {
                    const float sv = a_i + sc[j];
                    const float sl = sv > 0.f ? 1.f : 0.2f;
                    ex[j] = expf(sv * sl - mx);
                    sds[lane][j] = sl;
                    sum += ex[j];
                }
            }
            const float inv = 1.f / sum;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                if (j < J) {
                    const float pv = ex[j] * inv;
                    sp[lane][j] = pv;
                    sattT[j * 32 + lane] = f2bf(pv + ckrow[j]);
                }
            }
        }
        // datt[i][j] = sum_c dy[i][c] g[j][c]: fragments straight from the registers
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < M::KS; ++ks) d = mfma_bf16(dfr[ks], gfr[ks], d);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ckacc[r] += d[r];                         // (entries with i or j >= J hold finite junk that is never written out)
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < J && li < J) sdat[i][li] = d[r];
        }
        wave_lds_sync();
        // softmax + LeakyReLU backward, row i = lane
        if (lane < J) {
            float dot = 0.f;
            for (int j = 0; j < J; ++j) dot = fmaf(sp[lane][j], sdat[lane][j], dot);
            float da = 0.f;
            for (int j = 0; j < J; ++j) {
                const float ds = sp[lane][j] * (sdat[lane][j] - dot) * sds[lane][j];
                sds[lane][j] = ds;
                da += ds;
            }
            dAC[((long)f * J + lane) * lddac + h] = f2bf(da);
            da_sum += da;
        }
        wave_lds_sync();
        if (lane < J) {
            float dc = 0.f;
            for (int i = 0; i < J; ++i) dc += sds[i][lane];
            dAC[((long)f * J + lane) * lddac + nheads + h] = f2bf(dc);
            dc_sum += dc;
        }
        // dg[j][c] = sum_i att[i][j] dy[i][c]
#pragma unroll
        for (int nt = 0; nt < M::NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                acc = mfma_bf16(*(const uint4*)(sattT + li * 32 + k2 * 16 + lh * 8), *(const uint4*)(sdyT + (nt * 32 + li) * 32 + k2 * 16 + lh * 8), acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (j < J) {
                    dG[((long)f * J + j) * lddg + h * CI + nt * 32 + li] = f2bf(acc[r]);
                    gsum[nt] += acc[r];
                }
            }
        }
        wave_lds_sync();
    }
    // per-wave partial row (layout of attn_bwd_wave_kernel): [g bias sums (C) | da sums | dc sums | dC_k (nheads*J*J)]
    float* row = ws + (long)(gw / nheads) * ncol;
#pragma unroll
    for (int nt = 0; nt < M::NT; ++nt) {
        const float t = gsum[nt] + __shfl_xor(gsum[nt], 32);
        if (lh == 0) row[h * CI + nt * 32 + li] = t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { da_sum += __shfl_xor(da_sum, off); dc_sum += __shfl_xor(dc_sum, off); }
    if (lane == 0) { row[C + h] = da_sum; row[C + nheads + h] = dc_sum; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < J && li < J) row[C + 2 * nheads + h * J * J + i * J + li] = ckacc[r];
    }
}

// dbias[n] += sum_r ws[r][n] (n < nb);  dCk[n - nb] += sum_r ws[r][n] (n >= nb)      (256 threads = 32 columns x 8 row lanes)
__global__ void __launch_bounds__(256) attn_bwd_finish_kernel(const float* __restrict__ ws, int nrow, int ncol, int nb,
                                                              float* __restrict__ dbias, float* __restrict__ dCk) {
    __shared__ float sred[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cx;
    float a = 0.f;
    if (n < ncol) {
#pragma unroll 4
        for (int r = ry; r < nrow; r += 8) a += ws[(long)r * ncol + n];
    }
    sred[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && n < ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sred[r][cx];
        if (n < nb) { if (dbias) dbias[n] += t; }
        else dCk[n - nb] += t;
    }
}

// gast_rowsum_multi: the deferred finishes of a backward pass (attention: dbias / dC_k += column sums of the per-wave partial rows;
// aggregation: dA = column sums of the per-block partial rows) as ONE launch, blockIdx.y = job -- they were six launches per step
// (256 threads = 32 columns x 8 row lanes, as the two kernels above)
struct RowsumBatch { gast_rowsum_job j[GAST_ROWSUM_MAX_BATCH]; };
__global__ void __launch_bounds__(256) rowsum_multi_kernel(const RowsumBatch b) {
    __shared__ float sred[8][32];
    const gast_rowsum_job& j = b.j[blockIdx.y];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const long n = (long)blockIdx.x * 32 + cx;
    if ((long)blockIdx.x * 32 >= j.ncol) return;
    float a = 0.f;
    if (n < j.ncol) {
#pragma unroll 4
        for (int r = ry; r < j.nrow; r += 8) a += j.ws[(long)r * j.ncol + n];
    }
    sred[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && n < j.ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sred[r][cx];
        float* o = n < j.nb ? (j.out0 ? j.out0 + n : nullptr) : (j.out1 ? j.out1 + (n - j.nb) : nullptr);
        if (o) *o = j.accumulate ? *o + t : t;
    }
}

}  // namespace

extern "C" int gast_rowsum_multi(const gast_rowsum_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_ROWSUM_MAX_BATCH) return GAST_EINVAL;
    RowsumBatch b;
    long maxcol = 0;
    for (int d = 0; d < n; ++d) {
        const gast_rowsum_job& j = jobs[d];
        if (!j.ws || j.nrow < 1 || j.ncol < 1 || j.nb < 0 || j.nb > j.ncol || (!j.out0 && !j.out1)) return GAST_EINVAL;
        b.j[d] = j;
        if (j.ncol > maxcol) maxcol = j.ncol;
    }
    hipLaunchKernelGGL(rowsum_multi_kernel, dim3((unsigned)((maxcol + 31) / 32), n), dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_semch_adj_fwd(const float* e, int C, const int32_t* pat, float* A_t, gast_stream_t stream) {
    if (!e || !pat || !A_t || C < 1) return GAST_EINVAL;
    // J is read on the device; launch for the maximum J supported
    int n = C * JMAX;
    hipLaunchKernelGGL(semch_adj_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e, C, pat, A_t);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_semch_adj_bwd(const float* dA_t, const float* A_t, int C, const int32_t* pat, float* de, gast_stream_t stream) {
    if (!dA_t || !A_t || !pat || !de || C < 1) return GAST_EINVAL;
    int n = C * JMAX;
    hipLaunchKernelGGL(semch_adj_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dA_t, A_t, C, pat, de);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int gast_semch_adj_multi(const gast_adj_job* jobs, int n, int backward, gast_stream_t stream) {
    if (!jobs || n < 1 || n > GAST_ADJ_MAX_BATCH) return GAST_EINVAL;
    AdjBatch b;
    int maxC = 0;
    for (int d = 0; d < n; ++d) {
        if (!jobs[d].e || !jobs[d].pat || !jobs[d].A_t || jobs[d].C < 1 || (backward && !jobs[d].dA_t)) return GAST_EINVAL;
        b.j[d] = jobs[d];
        if (jobs[d].C > maxC) maxC = jobs[d].C;
    }
    const int nthr = maxC * JMAX;
    if (backward)
        hipLaunchKernelGGL(semch_adj_bwd_multi_kernel, dim3((nthr + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, b, backward == 2);
    else
        hipLaunchKernelGGL(semch_adj_fwd_multi_kernel, dim3((nthr + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, b);
    GAST_CHECK_LAUNCH();
    return 0;
}

// forward aggregation: blocks of (FB frames) x (a chunk of CC <= 64 channels); TPF = CC / 4 threads per frame
static int agg_fwd_cc(int C) { return C < 64 ? C : 64; }
static int agg_fwd_tpf(int C) { return agg_fwd_cc(C) / 4; }
static int agg_fwd_frame_blocks(int F, int C) {
    const int CC = agg_fwd_cc(C), nchunk = (C + CC - 1) / CC;
    int FB = 256 / agg_fwd_tpf(C);
    int nb = (F + FB - 1) / FB, cap = 1024 / nchunk;
    if (cap < 1) cap = 1;
    return nb < cap ? nb : cap;
}
// few frames (the M = B*J stage): the rows of a frame are dealt to 4 blocks so that the launch covers the chip; the partial-sum
// rows are then [joint part][frame block]
static int agg_fwd_joint_split(int F, int C) {
    const int CC = agg_fwd_cc(C);
    const int nblk = agg_fwd_frame_blocks(F, C) * ((C + CC - 1) / CC);
    // (round 3: splitting the joints further at the large stages -- more, shorter blocks -- is SLOWER: 3.87 / 3.89 vs 3.86 ms per step
    //  with 2 / 4 parts forward, 3.92 / 3.95 backward; the rule stays)
    return nblk <= 128 ? 4 : nblk <= 512 ? 2 : 1;
}
extern "C" int gast_semch_agg_blocks(int F, int C) { return agg_fwd_frame_blocks(F, C) * agg_fwd_joint_split(F, C); }

extern "C" int gast_semch_agg_fwd(int dtype, const void* H, int ldh, int F, int J, int C,
                                  const float* A_sym, const int32_t* pat_sym, int deg_sym, const float* A_con,
                                  const int32_t* pat_con, int deg_con, void* Y, int ldy, float* partials,
                                  const float* center_sym, const float* center_con, gast_stream_t stream) {
    if (!H || !A_sym || !A_con || !pat_sym || !pat_con || !Y || !partials) return GAST_EINVAL;
    if (dtype != GAST_F32 && dtype != GAST_BF16) return GAST_EINVAL;
    if (C % 4 || ldh % 4 || ldy % 4 || J < 1 || J > JMAX || F < 1) return GAST_EALIGN;
    const int CC = agg_fwd_cc(C), nchunk = (C + CC - 1) / CC;
    int TPF = agg_fwd_tpf(C), FB = 256 / TPF;
    const int nb = agg_fwd_frame_blocks(F, C), jsplit = agg_fwd_joint_split(F, C);
    const dim3 grid_ell(nb * nchunk, jsplit);
    // LDS for the coefficient slices: at most J * deg + 1 rows per pattern (the tables are padded to the fixed degree)
    const size_t smem = (size_t)(J * deg_sym + 1 + J * deg_con + 1) * CC * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define AGG_FWD_ELL(DS, DC)                                                                                                  \
    do {                                                                                                                     \
        if (dtype == GAST_F32)                                                                                               \
            hipLaunchKernelGGL((semch_agg_fwd_ell_kernel<float, DS, DC>), grid_ell, dim3(256), smem, st, (const float*)H, ldh, F, J, C, \
                               A_sym, pat_sym, A_con, pat_con, (float*)Y, ldy, partials, TPF, FB, center_sym, center_con, nchunk, CC); \
        else                                                                                                                 \
            hipLaunchKernelGGL((semch_agg_fwd_ell_kernel<bf16_t, DS, DC>), grid_ell, dim3(256), smem, st, (const bf16_t*)H, ldh, F, J, \
                               C, A_sym, pat_sym, A_con, pat_con, (bf16_t*)Y, ldy, partials, TPF, FB, center_sym, center_con, nchunk, CC); \
    } while (0)
    // the fixed-degree kernels walk exactly DS / DC padded slots per row: the pattern tables must have been built with
    // these degrees, which is the case for deg_sym == 2 (every supported skeleton) and deg_con in {5, 6}
    if (deg_sym == 2 && deg_con == 5) AGG_FWD_ELL(2, 5);
    else if (deg_sym == 2 && deg_con == 6) AGG_FWD_ELL(2, 6);
    else {
        // generic CSR kernel: it fills the first nb partial rows only; the joint-split rows stay zero
        if (jsplit > 1) {
            hipError_t e = hipMemsetAsync(partials + (size_t)nb * 2 * C * 2, 0, (size_t)nb * (jsplit - 1) * 2 * C * 2 * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
        if (dtype == GAST_F32)
            hipLaunchKernelGGL((semch_agg_fwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)H, ldh, F, J, C, A_sym, pat_sym,
                               A_con, pat_con, (float*)Y, ldy, partials, TPF, FB, center_sym, center_con);
        else
            hipLaunchKernelGGL((semch_agg_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)H, ldh, F, J, C, A_sym,
                               pat_sym, A_con, pat_con, (bf16_t*)Y, ldy, partials, TPF, FB, center_sym, center_con);
    }
#undef AGG_FWD_ELL
    GAST_CHECK_LAUNCH();
    return 0;
}

struct AggBwdCfg { int CC, nchunk, TPF, FB, nfb, lds; };
// lds_variant: the LDS-staged kernel (fp32 storage only: with 16-bit storage a 32-channel row piece is 64 bytes and the staging pass
// costs more than the re-reads it saves -- the f16 step was 2.80 vs 2.66 ms with it)
static AggBwdCfg agg_bwd_cfg(int F, int C, bool lds_variant) {
    AggBwdCfg c;
    // (measured in the B = 128 step, per launch of the three stages: 19.3 / 104.0 / 70.7 us against the ELL kernel's 34.6 / 127.4 / 76.3;
    //  block budget 512 = two blocks per CU: 384 / 768 / 1024 are slower; 3 frames per group the same, 4 spill)
    static const int lds_env = getenv("GAST_AGG_BWD_LDS") ? atoi(getenv("GAST_AGG_BWD_LDS")) : 1;
    c.lds = lds_env && lds_variant;
    if (c.lds) {          // semch_agg_bwd_lds_kernel: 32-channel chunks, U frames per iteration
        c.CC = AGG_LDS_CC;
        c.nchunk = (C + c.CC - 1) / c.CC;
        c.TPF = c.CC / 4;
        c.FB = AGG_LDS_U;
        static const int want_lds = getenv("GAST_AGG_BWD_BLOCKS") ? atoi(getenv("GAST_AGG_BWD_BLOCKS")) : 512;
        int want = want_lds / c.nchunk;
        if (want < 1) want = 1;
        const int maxfb = (F + c.FB - 1) / c.FB;
        c.nfb = want < maxfb ? want : maxfb;
        return c;
    }
    c.CC = C < 64 ? C : 64;           // channel chunk of a block (its coefficient slices live in LDS)
    c.nchunk = (C + c.CC - 1) / c.CC;
    c.TPF = c.CC / 4;
    c.FB = 256 / c.TPF;
    static const int want_env = getenv("GAST_AGG_BWD_BLOCKS") ? atoi(getenv("GAST_AGG_BWD_BLOCKS")) : 1024;   // 512: 58 us avg at B=128, 1024: 51, 2048: 51
    int want = want_env / c.nchunk;
    if (want < 1) want = 1;
    int maxfb = (F + c.FB - 1) / c.FB;
    c.nfb = want < maxfb ? want : maxfb;
    return c;
}

extern "C" long gast_semch_agg_bwd_ws_floats(int F, int C, int nnz_sym, int nnz_con) {
    const AggBwdCfg a = agg_bwd_cfg(F, C, false), b = agg_bwd_cfg(F, C, true);      // (the caller does not say which storage type: the larger of the two)
    return (long)(a.nfb > b.nfb ? a.nfb : b.nfb) * (nnz_sym + nnz_con) * C;
}

// the BatchNorm-backward apply of dY fused into the staging pass (semch_agg_bwd_lds_kernel<.., BN = true>): Yp != null
struct AggBwdBn { const void* Yp; int ldyp; const float* ka; const float* kb; const float* kc; };

static bool agg_bwd_takes_lds(int dtype, int F, int C, int nnz_sym, int cdeg_sym, int nnz_con, int cdeg_con) {
    const AggBwdCfg c = agg_bwd_cfg(F, C, dtype == GAST_F32 && (cdeg_sym == 2 && (cdeg_con == 5 || cdeg_con == 6)));
    return c.lds && (long)(nnz_sym + nnz_con) * (AGG_LDS_CC / 4) <= 256L * AGG_LDS_NPB;
}

static int semch_agg_bwd_impl(int dtype, const void* dY, int ldy, const void* H, int ldh, int F, int J, int C,
                              const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                              const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                              gast_rowsum_job* finish, gast_stream_t stream, const AggBwdBn* bn = nullptr) {
    if (!dY || !H || !A_sym || !A_con || !pat_sym || !pat_con || !dH || !dA || !ws) return GAST_EINVAL;
    if (dtype != GAST_F32 && dtype != GAST_BF16) return GAST_EINVAL;
    if (C % 4 || ldh % 4 || ldy % 4 || lddh % 4 || J < 1 || J > JMAX || F < 1 || nnz_sym < 1 || nnz_con < 1) return GAST_EALIGN;
    AggBwdCfg c = agg_bwd_cfg(F, C, dtype == GAST_F32 && (cdeg_sym == 2 && (cdeg_con == 5 || cdeg_con == 6)));
    const int nnz_t = nnz_sym + nnz_con;
    if (bn) {
        if (!bn->Yp || !bn->ka || !bn->kb || !bn->kc || bn->ldyp % 4) return GAST_EINVAL;
        if (!agg_bwd_takes_lds(dtype, F, C, nnz_sym, cdeg_sym, nnz_con, cdeg_con)) return GAST_EINVAL;      // (ask gast_semch_agg_bwd_fuses_bn first)
    }
    // fixed-degree kernel: LDS for the coefficient slices, at most J * deg + 1 rows per pattern
    size_t smem = (size_t)(J * cdeg_sym + 1 + J * cdeg_con + 1) * c.CC * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(c.nfb * c.nchunk);
    // few frames (the M = B*J stage): split the joints / edges over 4 blocks so that the launch covers the chip
    dim3 grid_ell(c.nfb * c.nchunk, c.nfb * c.nchunk <= 128 ? 4 : 1);
#define AGG_BWD_ELL(DS, DC)                                                                                                   \
    do {                                                                                                                      \
        if (dtype == GAST_F32) {                                                                                              \
            if (smem > 48 * 1024) hipFuncSetAttribute((const void*)semch_agg_bwd_ell_kernel<float, DS, DC>,                  \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            hipLaunchKernelGGL((semch_agg_bwd_ell_kernel<float, DS, DC>), grid_ell, dim3(256), smem, st, (const float*)dY, ldy, \
                               (const float*)H, ldh, F, J, C, A_sym, pat_sym, A_con, pat_con, (float*)dH, lddh, ws, c.nfb,    \
                               c.nchunk, c.CC, c.TPF, c.FB);                                                                  \
        } else {                                                                                                              \
            if (smem > 48 * 1024) hipFuncSetAttribute((const void*)semch_agg_bwd_ell_kernel<bf16_t, DS, DC>,                 \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            hipLaunchKernelGGL((semch_agg_bwd_ell_kernel<bf16_t, DS, DC>), grid_ell, dim3(256), smem, st, (const bf16_t*)dY, ldy, \
                               (const bf16_t*)H, ldh, F, J, C, A_sym, pat_sym, A_con, pat_con, (bf16_t*)dH, lddh, ws, c.nfb,  \
                               c.nchunk, c.CC, c.TPF, c.FB);                                                                  \
        }                                                                                                                     \
    } while (0)
#define AGG_BWD_LDS(DS, DC)                                                                                                   \
    do {                                                                                                                      \
        const size_t sm = ((size_t)(J * cdeg_sym + 1 + J * cdeg_con + 1) * AGG_LDS_CC + (size_t)AGG_LDS_U * J * 6 * AGG_LDS_CC) * sizeof(float); \
        if (dtype == GAST_F32) {                                                                                              \
            if (sm > 48 * 1024) hipFuncSetAttribute((const void*)semch_agg_bwd_lds_kernel<float, DS, DC>,                    \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);                     \
            if (bn) {                                                                                                        \
                if (sm > 48 * 1024) hipFuncSetAttribute((const void*)semch_agg_bwd_lds_kernel<float, DS, DC, true>,              \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);                     \
                hipLaunchKernelGGL((semch_agg_bwd_lds_kernel<float, DS, DC, true>), grid, dim3(256), sm, st, (const float*)dY, ldy, \
                                   (const float*)H, ldh, F, J, C, A_sym, pat_sym, A_con, pat_con, (float*)dH, lddh, ws, c.nfb, c.nchunk, \
                                   (const float*)bn->Yp, bn->ldyp, bn->ka, bn->kb, bn->kc);                                   \
            } else                                                                                                            \
            hipLaunchKernelGGL((semch_agg_bwd_lds_kernel<float, DS, DC>), grid, dim3(256), sm, st, (const float*)dY, ldy,     \
                               (const float*)H, ldh, F, J, C, A_sym, pat_sym, A_con, pat_con, (float*)dH, lddh, ws, c.nfb, c.nchunk, \
                               (const float*)nullptr, 0, nullptr, nullptr, nullptr);                                          \
        } else {                                                                                                              \
            if (sm > 48 * 1024) hipFuncSetAttribute((const void*)semch_agg_bwd_lds_kernel<bf16_t, DS, DC>,                   \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);                     \
            hipLaunchKernelGGL((semch_agg_bwd_lds_kernel<bf16_t, DS, DC>), grid, dim3(256), sm, st, (const bf16_t*)dY, ldy,   \
                               (const bf16_t*)H, ldh, F, J, C, A_sym, pat_sym, A_con, pat_con, (bf16_t*)dH, lddh, ws, c.nfb, c.nchunk, \
                               (const bf16_t*)nullptr, 0, nullptr, nullptr, nullptr);                                         \
        }                                                                                                                     \
    } while (0)
    bool fast = true;
    const bool lds_ok = c.lds && (long)nnz_t * (AGG_LDS_CC / 4) <= 256L * AGG_LDS_NPB;
    if (cdeg_sym == 2 && cdeg_con == 5) { if (lds_ok) AGG_BWD_LDS(2, 5); else AGG_BWD_ELL(2, 5); }
    else if (cdeg_sym == 2 && cdeg_con == 6) { if (lds_ok) AGG_BWD_LDS(2, 6); else AGG_BWD_ELL(2, 6); }
    else fast = false;
#undef AGG_BWD_ELL
#undef AGG_BWD_LDS
    if (fast) {
        GAST_CHECK_LAUNCH();
        long ncol_f = (long)nnz_t * C;
        if (finish) {            // deferred: the caller sums the partial rows later (gast_rowsum_multi)
            *finish = gast_rowsum_job{ws, c.nfb, ncol_f, ncol_f, dA, nullptr, 0};
            return 0;
        }
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((ncol_f + 31) / 32)), dim3(256), 0, st, ws, c.nfb, ncol_f, dA);
        GAST_CHECK_LAUNCH();
        return 0;
    }
    smem = (size_t)nnz_t * c.CC * sizeof(float);
    if (smem > 48 * 1024) {
        hipError_t e = dtype == GAST_F32
            ? hipFuncSetAttribute((const void*)semch_agg_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
            : hipFuncSetAttribute((const void*)semch_agg_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((semch_agg_bwd_kernel<float>), grid, dim3(256), smem, st, (const float*)dY, ldy, (const float*)H, ldh, F, J,
                           C, A_sym, pat_sym, A_con, pat_con, (float*)dH, lddh, ws, c.nfb, c.nchunk, c.CC, c.TPF, c.FB);
    else
        hipLaunchKernelGGL((semch_agg_bwd_kernel<bf16_t>), grid, dim3(256), smem, st, (const bf16_t*)dY, ldy, (const bf16_t*)H, ldh, F,
                           J, C, A_sym, pat_sym, A_con, pat_con, (bf16_t*)dH, lddh, ws, c.nfb, c.nchunk, c.CC, c.TPF, c.FB);
    GAST_CHECK_LAUNCH();
    long ncol = (long)nnz_t * C;
    if (finish) {
        *finish = gast_rowsum_job{ws, c.nfb, ncol, ncol, dA, nullptr, 0};
        return 0;
    }
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((ncol + 31) / 32)), dim3(256), 0, st, ws, c.nfb, ncol, dA);
    GAST_CHECK_LAUNCH();
    return 0;
}
extern "C" int gast_semch_agg_bwd(int dtype, const void* dY, int ldy, const void* H, int ldh, int F, int J, int C,
                                  const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                                  const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                                  gast_stream_t stream) {
    return semch_agg_bwd_impl(dtype, dY, ldy, H, ldh, F, J, C, A_sym, pat_sym, nnz_sym, cdeg_sym, A_con, pat_con, nnz_con, cdeg_con, dH, lddh,
                              dA, ws, nullptr, stream);
}
extern "C" int gast_semch_agg_bwd_deferred(int dtype, const void* dY, int ldy, const void* H, int ldh, int F, int J, int C,
                                           const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                                           const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                                           gast_rowsum_job* finish, gast_stream_t stream) {
    if (!finish) return GAST_EINVAL;
    finish->ws = nullptr;
    return semch_agg_bwd_impl(dtype, dY, ldy, H, ldh, F, J, C, A_sym, pat_sym, nnz_sym, cdeg_sym, A_con, pat_con, nnz_con, cdeg_con, dH, lddh,
                              dA, ws, finish, stream);
}

extern "C" int gast_semch_agg_bwd_fuses_bn(int dtype, int F, int J, int C, int nnz_sym, int cdeg_sym, int nnz_con, int cdeg_con) {
    if (J < 1 || J > JMAX || F < 1 || C % 4) return 0;
    return agg_bwd_takes_lds(dtype, F, C, nnz_sym, cdeg_sym, nnz_con, cdeg_con) ? 1 : 0;
}
extern "C" int gast_semch_agg_bwd_bn(int dtype, const void* dY, int ldy, const void* Ypre, int ldyp, const float* ka, const float* kb,
                                     const float* kc, const void* H, int ldh, int F, int J, int C,
                                     const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                                     const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                                     gast_rowsum_job* finish, gast_stream_t stream) {
    const AggBwdBn bn = {Ypre, ldyp, ka, kb, kc};
    if (finish) finish->ws = nullptr;
    return semch_agg_bwd_impl(dtype, dY, ldy, H, ldh, F, J, C, A_sym, pat_sym, nnz_sym, cdeg_sym, A_con, pat_con, nnz_con, cdeg_con, dH, lddh,
                              dA, ws, finish, stream, &bn);
}

static int attn_grid(int F, int nheads, int ub) {
    int per_head = (F + ub - 1) / ub;
    if (per_head > 512) per_head = 512;
    return per_head * nheads;
}

// wave-per-unit path: head width 32 / 64 / 128 channels, vector-aligned operands, wave count a multiple of nheads
static int attn_wave_ci4(int dtype, int C, int nheads, int ldg, int ldy, const void* G, const void* Y) {
    if (C % nheads) return 0;
    const int Ci = C / nheads;
    if (Ci != 32 && Ci != 64 && Ci != 128) return 0;
    if (4 % nheads) return 0;                         // 4 waves per block: every wave keeps one head
    const int al = dtype == GAST_F32 ? 16 : 8;        // ld4 granularity in bytes
    const int es = dtype == GAST_F32 ? 4 : 2;
    if ((ldg * es) % al || (ldy * es) % al || ((uintptr_t)G % al) || ((uintptr_t)Y % al)) return 0;
    return Ci / 4;
}
// Grid of the wave-per-unit kernels: one unit per wave until the cap.  A unit is a ~15 us dependent chain, so the cap is ONE round of
// resident blocks, 256 CUs x the occupancy the runtime reports for the kernel (LDS and registers): at 768 blocks the 64-channel-head
// backward (66 KB of LDS per block, two per CU) ran a second, half-empty round -- 85 vs 71 us; the fp32 forward kernels fit 3-5
// blocks per CU.  per_cu = 0: the historical 768 (the bf16 kernels, tuned at that value).
constexpr int ATTN_MAX_GRID = 2048;
static int attn_wave_grid(int F, int nheads, int per_cu = 0) {
    long units = (long)F * nheads;
    long g = (units + 3) / 4;
    long cap = per_cu > 0 ? 256L * per_cu : 768;
    if (cap > ATTN_MAX_GRID) cap = ATTN_MAX_GRID;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}
static int attn_blocks_per_cu(const void* kernel, size_t smem) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, smem) != hipSuccess || nb < 1) nb = 1;
    return nb;
}
// The occupancy of one kernel instantiation (its dynamic LDS size is a compile-time constant of the instantiation), cached PER DEVICE:
// nn.DataParallel replicas launch from several threads on several devices.  A racing first call computes the same value twice.
struct AttnOcc {
    std::atomic<int> v[64];
    AttnOcc() { for (auto& x : v) x.store(0, std::memory_order_relaxed); }
    int get(const void* kernel, size_t smem) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        int nb = v[dev].load(std::memory_order_relaxed);
        if (nb == 0) { nb = attn_blocks_per_cu(kernel, smem); v[dev].store(nb, std::memory_order_relaxed); }
        return nb;
    }
};

// bf16 + 16-byte aligned row tiles: the MFMA kernels (GAST_ATTN_MFMA=0 keeps the VALU wave kernels)
static bool attn_mfma_ok(int ld_a, const void* a, int ld_b, const void* b) {
    static const bool on = getenv("GAST_ATTN_MFMA") ? atoi(getenv("GAST_ATTN_MFMA")) != 0 : true;
    return on && ld_a % 8 == 0 && ld_b % 8 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
}

template <typename T, int CI4, int JT>
static int launch_attn_fwd_wave_j(const void* G, int ldg, const void* AC, int ldac, const float* C_k, int F, int J, int nheads, void* Y,
                                  int ldy, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (attn_mfma_ok(ldg, G, ldg, G)) {
            constexpr int CI = CI4 * 4;
            const size_t smem = (size_t)4 * AttnM<CI>::FWD_BYTES;
            if (smem > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<CI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL((attn_fwd_mfma_kernel<CI>), dim3(attn_wave_grid(F, nheads)), dim3(256), smem, st, (const bf16_t*)G, ldg,
                               (const bf16_t*)AC, ldac, C_k, F, J, nheads, (bf16_t*)Y, ldy);
            GAST_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t smem = (size_t)4 * AttnW<CI4>::FWD_FLOATS * sizeof(float);
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_wave_kernel<T, CI4, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    static AttnOcc occ;
    const int per_cu = sizeof(T) == 4 ? occ.get((const void*)attn_fwd_wave_kernel<T, CI4, JT>, smem) : 0;
    hipLaunchKernelGGL((attn_fwd_wave_kernel<T, CI4, JT>), dim3(attn_wave_grid(F, nheads, per_cu)), dim3(256), smem, st, (const T*)G, ldg, (const T*)AC,
                       ldac, C_k, F, J, nheads, (T*)Y, ldy);
    GAST_CHECK_LAUNCH();
    return 0;
}
// the shipped skeletons (Human3.6M 17, + toes 19, HumanEva 15) get the straight-line fp32 kernels, everything else the run-time-J ones
// (GAST_ATTN_JT=0: run-time J everywhere, the A/B and bisecting switch)
static bool attn_jt_enabled() {
    static const int on = getenv("GAST_ATTN_JT") ? atoi(getenv("GAST_ATTN_JT")) : 1;
    return on != 0;
}
template <typename T, int CI4>
static int launch_attn_fwd_wave(const void* G, int ldg, const void* AC, int ldac, const float* C_k, int F, int J, int nheads, void* Y,
                                int ldy, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        if (attn_jt_enabled()) {
            if (J == 17) return launch_attn_fwd_wave_j<T, CI4, 17>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, st);
            if (J == 19) return launch_attn_fwd_wave_j<T, CI4, 19>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, st);
            if (J == 15) return launch_attn_fwd_wave_j<T, CI4, 15>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, st);
        }
    }
    return launch_attn_fwd_wave_j<T, CI4, 0>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, st);
}

template <typename T, int CI4, int JT>
static int launch_attn_bwd_wave_j(const void* dY, int lddy, const void* G, int ldg, const void* AC, int ldac, const float* C_k, int F, int J,
                                  int nheads, void* dG, int lddg, void* dAC, int lddac, float* dC_k, float* dbias, float* ws,
                                  hipStream_t st, gast_rowsum_job* finish) {
    static AttnOcc occ;
    const int per_cu = sizeof(T) == 4 ? occ.get((const void*)attn_bwd_wave_kernel<T, CI4, JT>, (size_t)4 * AttnW<CI4>::BWD_FLOATS * sizeof(float)) : 0;
    const int grid = attn_wave_grid(F, nheads, per_cu);
    const int C = nheads * CI4 * 4;
    const int nb = C + 2 * nheads, ncol = nb + nheads * J * J;
    bool mfma = false;
    // (32-channel heads: the transposed staging costs more than the two 2-step products save -- 63 vs 48 us at B=128)
    if constexpr (sizeof(T) == 2) mfma = CI4 * 4 >= 64 && attn_mfma_ok(lddy, dY, ldg, G);
    if (mfma) {
        if constexpr (sizeof(T) == 2) {
            constexpr int CI = CI4 * 4;
            const size_t smem = (size_t)4 * AttnM<CI>::BWD_BYTES;
            if (smem > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<CI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL((attn_bwd_mfma_kernel<CI>), dim3(grid), dim3(256), smem, st, (const bf16_t*)dY, lddy, (const bf16_t*)G, ldg,
                               (const bf16_t*)AC, ldac, C_k, F, J, nheads, (bf16_t*)dG, lddg, (bf16_t*)dAC, lddac, ws, ncol);
        }
    } else {
        const size_t smem = (size_t)4 * AttnW<CI4>::BWD_FLOATS * sizeof(float);
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_wave_kernel<T, CI4, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((attn_bwd_wave_kernel<T, CI4, JT>), dim3(grid), dim3(256), smem, st, (const T*)dY, lddy, (const T*)G, ldg, (const T*)AC,
                           ldac, C_k, F, J, nheads, (T*)dG, lddg, (T*)dAC, lddac, ws, ncol);
    }
    GAST_CHECK_LAUNCH();
    if (finish) {            // deferred: dbias / dC_k += column sums of the partial rows, later (gast_rowsum_multi)
        *finish = gast_rowsum_job{ws, grid * 4 / nheads, (long)ncol, (long)nb, dbias, dC_k, 1};
        return 0;
    }
    hipLaunchKernelGGL(attn_bwd_finish_kernel, dim3((ncol + 31) / 32), dim3(256), 0, st, ws, grid * 4 / nheads, ncol, nb, dbias, dC_k);
    GAST_CHECK_LAUNCH();
    return 0;
}
template <typename T, int CI4>
static int launch_attn_bwd_wave(const void* dY, int lddy, const void* G, int ldg, const void* AC, int ldac, const float* C_k, int F, int J,
                                int nheads, void* dG, int lddg, void* dAC, int lddac, float* dC_k, float* dbias, float* ws,
                                hipStream_t st, gast_rowsum_job* finish) {
    if constexpr (sizeof(T) == 4) {
        if (attn_jt_enabled()) {
#define GAST_BWD_J(JT_) if (J == JT_) return launch_attn_bwd_wave_j<T, CI4, JT_>(dY, lddy, G, ldg, AC, ldac, C_k, F, J, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, st, finish);
            GAST_BWD_J(17) GAST_BWD_J(19) GAST_BWD_J(15)
#undef GAST_BWD_J
        }
    }
    return launch_attn_bwd_wave_j<T, CI4, 0>(dY, lddy, G, ldg, AC, ldac, C_k, F, J, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, st, finish);
}

extern "C" int gast_attn_fwd(int dtype, const void* G, int ldg, const void* AC, int ldac, const float* C_k,
                             int F, int J, int C, int nheads, void* Y, int ldy, gast_stream_t stream) {
    if (!G || !AC || !C_k || !Y) return GAST_EINVAL;
    if (dtype != GAST_F32 && dtype != GAST_BF16) return GAST_EINVAL;
    if (J < 1 || J > JMAX || nheads < 1 || C % nheads || F < 1) return GAST_EINVAL;
    switch (attn_wave_ci4(dtype, C, nheads, ldg, ldy, G, Y)) {
#define GAST_FWD_WAVE(CI4_)                                                                                                         \
        case CI4_: return dtype == GAST_F32                                                                                         \
            ? launch_attn_fwd_wave<float, CI4_>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, (hipStream_t)stream)                  \
            : launch_attn_fwd_wave<bf16_t, CI4_>(G, ldg, AC, ldac, C_k, F, J, nheads, Y, ldy, (hipStream_t)stream);
        GAST_FWD_WAVE(8) GAST_FWD_WAVE(16) GAST_FWD_WAVE(32)
#undef GAST_FWD_WAVE
        default: break;
    }
    const int Ci = C / nheads, GS = (Ci + 3) / 4 * 4 + 4;
    int ub = UB;
    size_t smem = 0;
    for (;; ub >>= 1) {
        smem = (size_t)(2 * UB * JMAX * JP + 2 * UB * JMAX + 8 + ub * J * GS) * sizeof(float);
        if (smem <= 64 * 1024 || ub == 1) break;
    }
    if (smem > 160 * 1024) return GAST_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    int grid = attn_grid(F, nheads, ub);
    if (smem > 48 * 1024) {
        hipError_t e = dtype == GAST_F32
            ? hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
            : hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((attn_fwd_kernel<float>), dim3(grid), dim3(256), smem, st, (const float*)G, ldg, (const float*)AC, ldac, C_k,
                           F, J, C, nheads, (float*)Y, ldy, ub);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(grid), dim3(256), smem, st, (const bf16_t*)G, ldg, (const bf16_t*)AC, ldac,
                           C_k, F, J, C, nheads, (bf16_t*)Y, ldy, ub);
    GAST_CHECK_LAUNCH();
    return 0;
}

extern "C" long gast_attn_bwd_ws_floats(int F, int J, int C, int nheads) {
    if (nheads < 1 || F < 1 || J < 1) return 0;
    // (an upper bound: the backward's grid depends on the occupancy of the kernel variant it picks, at most ATTN_MAX_GRID blocks)
    long g = ((long)F * nheads + 3) / 4;
    if (g > ATTN_MAX_GRID) g = ATTN_MAX_GRID;
    return (g * 4 / nheads + 1) * (long)(C + 2 * nheads + nheads * J * J);
}

static int attn_bwd_impl(int dtype, const void* dY, int ldy, const void* G, int ldg, const void* AC, int ldac,
                         const float* C_k, int F, int J, int C, int nheads,
                         void* dG, int lddg, void* dAC, int lddac, float* dC_k, float* dbias, float* ws, gast_rowsum_job* finish,
                         gast_stream_t stream) {
    if (!dY || !G || !AC || !C_k || !dG || !dAC || !dC_k) return GAST_EINVAL;
    if (dtype != GAST_F32 && dtype != GAST_BF16) return GAST_EINVAL;
    if (J < 1 || J > JMAX || nheads < 1 || C % nheads || F < 1) return GAST_EINVAL;
    if (ws && attn_wave_ci4(dtype, C, nheads, ldg, ldy, G, dY) && attn_wave_ci4(dtype, C, nheads, lddg, lddg, dG, dG)) {
        switch (C / nheads / 4) {
#define GAST_BWD_WAVE(CI4_)                                                                                                         \
            case CI4_: return dtype == GAST_F32                                                                                     \
                ? launch_attn_bwd_wave<float, CI4_>(dY, ldy, G, ldg, AC, ldac, C_k, F, J, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, \
                                                    (hipStream_t)stream, finish)                                                    \
                : launch_attn_bwd_wave<bf16_t, CI4_>(dY, ldy, G, ldg, AC, ldac, C_k, F, J, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, \
                                                     (hipStream_t)stream, finish);
            GAST_BWD_WAVE(8) GAST_BWD_WAVE(16) GAST_BWD_WAVE(32)
#undef GAST_BWD_WAVE
            default: break;
        }
    }
    float* dbias_ac = dbias ? dbias + C : nullptr;
    const int Ci = C / nheads, GS = (Ci + 3) / 4 * 4 + 4;
    int ub = UB;
    size_t smem = 0;
    for (;; ub >>= 1) {
        smem = (size_t)(4 * UB * JMAX * JP + 2 * UB * JMAX + 8 + 2 * ub * J * GS) * sizeof(float);
        if (smem <= 64 * 1024 || ub == 1) break;
    }
    if (smem > 160 * 1024) return GAST_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    int grid = gast_deterministic() ? nheads : attn_grid(F, nheads, ub);      // (deterministic: one block per head adds dC_k / the bias sums once)
    if (smem > 48 * 1024) {
        hipError_t e = dtype == GAST_F32
            ? hipFuncSetAttribute((const void*)attn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
            : hipFuncSetAttribute((const void*)attn_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (dtype == GAST_F32)
        hipLaunchKernelGGL((attn_bwd_kernel<float>), dim3(grid), dim3(256), smem, st, (const float*)dY, ldy, (const float*)G, ldg,
                           (const float*)AC, ldac, C_k, F, J, C, nheads, (float*)dG, lddg, (float*)dAC, lddac, dC_k, dbias_ac, ub);
    else
        hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), dim3(grid), dim3(256), smem, st, (const bf16_t*)dY, ldy, (const bf16_t*)G, ldg,
                           (const bf16_t*)AC, ldac, C_k, F, J, C, nheads, (bf16_t*)dG, lddg, (bf16_t*)dAC, lddac, dC_k, dbias_ac, ub);
    GAST_CHECK_LAUNCH();
    if (dbias) return gast_colsum(dtype, dG, lddg, (long)F * J, C, dbias, 0, stream);      // g-bias part: column sums of dG
    return 0;
}
extern "C" int gast_attn_bwd(int dtype, const void* dY, int ldy, const void* G, int ldg, const void* AC, int ldac,
                             const float* C_k, int F, int J, int C, int nheads,
                             void* dG, int lddg, void* dAC, int lddac, float* dC_k, float* dbias, float* ws, gast_stream_t stream) {
    return attn_bwd_impl(dtype, dY, ldy, G, ldg, AC, ldac, C_k, F, J, C, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, nullptr, stream);
}
extern "C" int gast_attn_bwd_deferred(int dtype, const void* dY, int ldy, const void* G, int ldg, const void* AC, int ldac,
                                      const float* C_k, int F, int J, int C, int nheads, void* dG, int lddg, void* dAC, int lddac,
                                      float* dC_k, float* dbias, float* ws, gast_rowsum_job* finish, gast_stream_t stream) {
    if (!finish) return GAST_EINVAL;
    finish->ws = nullptr;         // (stays null when a generic kernel, which needs no finish, took the call)
    return attn_bwd_impl(dtype, dY, ldy, G, ldg, AC, ldac, C_k, F, J, C, nheads, dG, lddg, dAC, lddac, dC_k, dbias, ws, finish, stream);
}

