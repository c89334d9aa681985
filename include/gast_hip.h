/*
 * gast_hip.h -- C ABI of the MI355X (gfx950) GAST-Net spatio-temporal hot path.
 *
 * Drop-in boundary.  The reference has no FFI: its hot path is the Python module `model.gast_net`
 * (reference model/gast_net.py:84-104,159-177,235-251; model/local_attention.py:35-53,130-151;
 * model/global_attention.py:52-82,103-130), which issues only stock ATen ops.  The entry points below are what
 * a binding for that path binds instead of ATen: each one replaces a cited group of reference lines.  The host
 * side (the model package under gast-net-3dposeestimation_amd/) keeps the reference's class names, constructor signatures and
 * state_dict and calls these through ctypes (gast-net-3dposeestimation_amd/gast_hip/binding.py).
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (torch tensors).  Kernels never allocate or free.
 *   - Activations are "position-major": one row per (b, t, j) position, channels contiguous; `ld*` are row strides
 *     in ELEMENTS.  Activation/weight element type is selected by `dtype` (GAST_F32 / GAST_BF16); statistics,
 *     scale/shift vectors, partial sums and parameter gradients are always fp32.
 *   - Every function enqueues work on `stream` and returns immediately: 0 on success, a hipError_t code (>0) on a
 *     launch failure, or a negative GAST_E* code on invalid arguments.  No exceptions cross the boundary, no
 *     internal synchronisation, no global mutable state.
 *   - A row map addresses a (B, T_total, J) tensor from a (B, Tn, J) iteration domain:
 *       row(b, t, j) = (b * T_total + t * t_stride + t_off) * J + j,  "invalid" (reads as zero / not written)
 *     when t * t_stride + t_off falls outside [0, T_total).  This is how the temporal taps of the dilated /
 *     strided convolution (gast_net.py:145-146,173,222,246) and the residual slices (:170,:243) are expressed
 *     without materialising anything.
 */
#ifndef GAST_HIP_H
#define GAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gast_stream_t; /* hipStream_t */

#define GAST_F32 0
#define GAST_BF16 1
/* fp32 storage like GAST_F32; the MFMA GEMMs (gast_gemm*, gast_wgrad*) split every operand into a bf16 hi/lo pair in
 * registers and accumulate hi*hi + hi*lo + lo*hi on the bf16 matrix cores in fp32 ("bf16x3": ~2^-17 relative per product,
 * 3/16 of the fp32 MFMA cost).  Every other entry point treats it as GAST_F32. */
#define GAST_F32X3 2
/* fp32 storage like GAST_F32X3, products on FP16 hi/lo pairs (v_mfma_f32_32x32x16_f16): 11 + 11 significand bits instead of
 * 8 + 8, ~2^-22 relative per product at the same matrix-core rate -- for operands inside fp16's range (|x| < 65504; parts below
 * 6e-8 are lost), i.e. the FORWARD GEMMs (post-BatchNorm activations, weights), not gradient operands.  gast_gemm / gast_gemm_ws /
 * gast_gemm_multi only (gast_wgrad* reject it); Wx images must be of the fp16 kind (gast_x3_image_job.f16). */
#define GAST_F32X3H 3

#define GAST_EINVAL (-1)   /* bad argument (null pointer, bad dtype, bad size) */
#define GAST_EALIGN (-2)   /* pointer or leading dimension not 16-byte aligned */
#define GAST_ERANGE (-3)   /* size outside the supported range */

#define GAST_MAX_SEG 8

#define GAST_PRO_NONE 0        /* operand used as stored */
#define GAST_PRO_BNRELU 1      /* relu(x * scale[k] + shift[k])             (BatchNorm2d + ReLU applied on load) */
#define GAST_PRO_BNRELU_DROP 2 /* relu(x * scale[k] + shift[k]) * keep / (1-p)   (+ nn.Dropout)                 */

#define GAST_EPI_PLAIN 0       /* C = acc (+bias) (+addend)                                                     */
#define GAST_EPI_STATS 1       /* as PLAIN, and per-column partial sums {sum c, sum c^2} for the following BN    */
#define GAST_EPI_BNRELU_BWD 2  /* C = (acc + addend) * [xscale*X+xshift > 0] * keep/(1-p); partial sums {sum C, sum C*X} */

typedef struct {
    int T_total;  /* frames per sequence of the addressed tensor */
    int t_stride; /* t_src = t * t_stride + t_off */
    int t_off;
} gast_rowmap;

/* Dropout stream: element e (linear element offset inside its tensor) of tensor `salt` is kept iff
 * bits16(hash32((e >> 1) ^ (seed*0x9E3779B9 + salt*0x85EBCA6B)), e & 1) >= thresh; kept values are multiplied by
 * inv_keep.  thresh = round(p * 65536), inv_keep = 65536 / (65536 - thresh).  `seed` lives in device memory so
 * a captured hipGraph can be replayed with a fresh mask per step. */
typedef struct {
    const uint32_t* seed; /* device pointer to one uint32 (may be null when thresh == 0) */
    uint32_t thresh;
    float inv_keep;
} gast_dropout;

typedef struct {
    const void* A;       /* [rows][lda]  activation operand of this K segment */
    int lda;
    int K;               /* depth of the segment (multiple of 4 for f32, 8 for bf16) */
    gast_rowmap map;     /* row map from the GEMM's (B,Tn,J) domain into A */
    const void* W;       /* [N][ldw]  weight segment, K contiguous ("B transposed") */
    int ldw;
    int pro;             /* GAST_PRO_* */
    const float* scale;  /* [K] when pro != NONE */
    const float* shift;  /* [K] */
    uint32_t salt;       /* dropout stream id of tensor A (pro == BNRELU_DROP) */
    const void* Wx;      /* optional (GAST_F32X3): pre-split bf16 image of W made by gast_x3_image_multi, [N][ldwx]; lets the GEMM
                          * take the large-M path (gemm_big.hip) whose weight tiles stream global -> LDS without passing registers.
                          * GAST_BF16 (16-bit storage, round 5): the k-group-major LAYOUT image of the 16-bit operand (image kind 2) */
    int ldwx;
} gast_gemm_seg;

/* gast_gemm: the channel-mixing GEMM family on MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16).
 *   C[cmap(m), n] = epi( sum_s sum_k pro_s(A_s[map_s(m), k]) * W_s[n, k] + bias[n] + addend[addmap(m), n] )
 * Replaces every Conv2d 1x1 / (k,1) dilated / strided conv, Conv1d 1x1 and X.W matmul of the path:
 * gast_net.py:19,31-32,60,99,130,145-148,164,173-174,207,222; local_attention.py:37-38,122,143;
 * global_attention.py:30-35,41,56-72,94,122; and, with transposed weights, their input gradients. */
typedef struct {
    int dtype;            /* element type of A, W, addend, X (and C unless out_f32) */
    int out_f32;          /* store C as fp32 regardless of dtype */
    int B, Tn, J;         /* iteration domain: M = B*Tn*J rows */
    int N;                /* output columns */
    int nseg;
    gast_gemm_seg seg[GAST_MAX_SEG];
    void* C;
    int ldc;
    gast_rowmap cmap;     /* where row m is stored */
    const float* bias;    /* [N] or null */
    int bias_neg;         /* subtract bias instead of adding it ("centred" storage of a pre-BN tensor: C = acc - running_mean) */
    const void* addend;   /* optional [rows][ldadd], added before the epilogue non-linearity */
    int ldadd;
    gast_rowmap addmap;
    int epi;              /* GAST_EPI_* */
    float* partials;      /* epi != PLAIN: [ceil(M/128)][N][2] fp32; fully overwritten by gast_gemm, ACCUMULATED into by the
                           * split-K paths of gast_gemm_ws -- the cross-block one and the in-block one of the M = B*J kernel
                           * (gemm_bj.hip, two 64-row tiles add into one 128-row entry) -- so pass it zero-filled there */
    const void* X;        /* epi == BNRELU_BWD: pre-BN tensor addressed with cmap, [rows][ldx] */
    int ldx;
    const float* xscale;  /* [N] */
    const float* xshift;  /* [N] */
    int xdrop;            /* forward applied dropout to relu(bn(X)) */
    uint32_t xsalt;
    gast_dropout drop;
    void* C2;             /* epi == BNRELU_BWD, optional: a second output [rows][ldc2] addressed with cmap that receives the value BEFORE the
                           * mask, acc (+bias) (+addend).  The input gradient of a block feeds two consumers -- the residual path takes it
                           * as it is, the branch through drop(relu(bn(T2))) takes it masked, with the BatchNorm-backward sums: one GEMM
                           * writes both, the stand-alone gast_bnrelu_bwd_mask pass (reference gast_net.py:174 backward) disappears */
    int ldc2;
    const float* f8_scale; /* optional (GAST_BF16, bf16 output): run this GEMM's operands as OCP e4m3 on v_mfma_f32_32x32x16_fp8_fp8
                            * ("mixed fp8", BASELINE.json configs[4]).  Device pointer to {s, 1/s}: every weight is multiplied by s
                            * (a power of two, gast_f8_scale_multi) before the conversion, the accumulators by 1/s; activations are
                            * converted as they are (post-BatchNorm values are O(1)).  All segments share the scale. */
} gast_gemm_args;

int gast_gemm(const gast_gemm_args* args, gast_stream_t stream);
/* Same, with an fp32 workspace: GEMMs with few output tiles and a long K loop (the M = B*J rows of the last stage) split K.
 * Round 6: with pre-split weight images (GAST_F32X3 / GAST_F32X3H), M <= 8191 and at most 600 64x64 tiles the split happens INSIDE
 * the block (two k-groups of four waves, accumulators exchanged through LDS in a fixed order, epilogue in the same launch; the
 * workspace pointer only enables the path, nothing is written to it).  Otherwise K splits over up to 8 blocks per tile and a
 * finish kernel applies bias / addend / epilogue.  ws_bytes >= gast_gemm_splitk_ws_bytes(M, N)
 * enables every split the heuristic may pick; a smaller or null workspace simply disables splitting. */
int gast_gemm_ws(const gast_gemm_args* args, void* ws, long ws_bytes, gast_stream_t stream);
/* n <= GAST_GEMM_MAX_BATCH independent GEMMs (same dtype / out_f32) in ONE grid: one launch and one tail for the thin GEMMs of a
 * plan step.  Jobs the split-K heuristic picks (small M, long K) ride in the same grid -- each with its own slice of the workspace --
 * and share ONE finish launch (round 3; when the workspace cannot hold all their partial tiles the overflowing jobs are launched on
 * their own, as before).  Per-job semantics of gast_gemm_ws. */
#define GAST_GEMM_MAX_BATCH 3
int gast_gemm_multi(const gast_gemm_args* args, int n, void* ws, long ws_bytes, gast_stream_t stream);
long gast_gemm_splitk_ws_bytes(long M, int N);
/* Per-tensor fp8 scales: out[0] = 2^floor(log2(448 / max |W|)) (1 for an all-zero tensor), out[1] = 1 / out[0], for the bf16
 * operands W[R][ldw] (K columns used); n jobs in one launch (GAST_F8_SCALE_MAX_BATCH per launch). */
#define GAST_F8_SCALE_MAX_BATCH 64
typedef struct { const void* W; int R, K, ldw; float* out; } gast_f8_scale_job;
int gast_f8_scale_multi(const gast_f8_scale_job* jobs, int n, gast_stream_t stream);
/* which kernel gast_gemm_ws would launch for these arguments: 0 = the 128x128-tile kernel, 1 = the large-M kernel (GAST_F32X3 /
 * GAST_F32X3H, and GAST_BF16 with 16-bit output and no fp8 scale; needs every segment's Wx image, >= 8192 rows, N >= 32, no
 * dropout prologue) */
int gast_gemm_path(const gast_gemm_args* args);
/* Pre-split weight image for GAST_F32X3, k-group-major: for every group g of 16 K values and every row r of the fp32 operand
 * W[R][ldw] (K columns used), 16 bf16 "hi" = bf16(w) followed by 16 bf16 "lo" = bf16(w - hi):
 *   img[g * ldimg + r * 32 + (k & 15)] = hi,  img[... + 16] = lo,  g = k >> 4,  zero for K <= k < 16 * ceil(K / 16),
 * (fp16 halves instead when the job's f16 flag is set: the image a GAST_F32X3H GEMM reads)
 * so the 256 x 16 weight tile of one K step of the large-M GEMM is ONE contiguous 16 KB block.  ldimg (bf16 elements per
 * k-group) >= gast_x3_image_ld(R) = 32 * (round_up(R, 16) + 256): the rows past R must exist and be ZERO-FILLED by the caller (a
 * tile may start at any row and always spans 256).  A row slice W[r0:] has the image img + 32 * r0, a column slice W[:, k0:]
 * with k0 % 16 == 0 the image img + (k0 / 16) * ldimg (same ldimg).  gast_gemm_seg.Wx / ldwx carry img / ldimg.
 * Image kind 2 (round 5): W is a 16-BIT operand [R][ldw] of the build's storage type (K, ldw multiples of 8) and the image is a pure
 * layout change for the one-product large-M kernel of GAST_BF16 GEMMs: groups of 32 K values,
 *   img[g * ldimg + r * 32 + (k & 31)] = W[r][k],  g = k >> 5,  zero for K <= k < 32 * ceil(K / 32)
 * (same 64-byte rows, same ldimg rule, same slicing with 32 in the place of 16).
 * n jobs in one launch (GAST_X3_IMAGE_MAX_BATCH per launch). */
#define GAST_X3_IMAGE_MAX_BATCH 64
typedef struct { const float* W; int R, K, ldw; void* img; int ldimg;
                 int f16;   /* image kind -- 0: bf16 hi/lo pairs (GAST_F32X3), 1: fp16 hi/lo pairs (GAST_F32X3H), same layout and size;
                             * 2: layout image of a 16-bit operand (W points at 16-bit values) */
} gast_x3_image_job;
int gast_x3_image_multi(const gast_x3_image_job* jobs, int n, gast_stream_t stream);
long gast_x3_image_ld(int R);
/* number of row blocks (first dimension of `partials`) gast_gemm uses for a domain of M rows */
int gast_gemm_row_blocks(int M);

typedef struct {
    const void* Q;       /* [rows][ldq] activation operand (forward "A") */
    int ldq;
    int S;               /* columns of Q used = K extent of the weight segment */
    gast_rowmap map;
    int pro;
    const float* scale;
    const float* shift;
    uint32_t salt;
    int wcol0;           /* first column of dW this segment fills */
} gast_wgrad_seg;

/* gast_wgrad: weight gradient  dW[r, wcol0_s + k] (+)= sum_m P[pmap(m), r] * pro_s(Q_s[map_s(m), k])
 * (autograd of the convs/matmuls listed under gast_gemm; reference main.py:237 `loss.backward()`).
 * dW is fp32 [R][ldw]; the function zero-fills it (hipMemsetAsync) and accumulates split-M partial tiles with
 * fp32 atomics. */
typedef struct {
    int dtype;
    int B, Tn, J;
    const void* P;       /* [rows][ldp]  output-gradient operand */
    int ldp;
    int R;               /* columns of P used = rows of dW */
    gast_rowmap pmap;
    int nseg;
    gast_wgrad_seg seg[GAST_MAX_SEG];
    float* dW;
    int ldw;
    int zero_first;      /* memset dW[R][ldw] before accumulating */
    gast_dropout drop;
} gast_wgrad_args;

int gast_wgrad(const gast_wgrad_args* args, gast_stream_t stream);
/* n <= GAST_WGRAD_MAX_BATCH weight gradients (same dtype) in ONE launch: the block budget, hence the split-M atomic volume and
 * the launch tail, is shared by all of them.  Same per-job semantics as gast_wgrad. */
#define GAST_WGRAD_MAX_BATCH 8
int gast_wgrad_multi(const gast_wgrad_args* args, int n, gast_stream_t stream);

/* ---- semantic channel-wise graph convolution (local_attention.py:10-56) ------------------------------------
 * Pattern (device int32 array, built once per skeleton by the host from local_attention.py:92-114):
 *   pat[0]=J, pat[1]=nnz, then row_ptr[J+1], col[nnz] (edges sorted by row i, CSR), col_ptr[J+1], crow[nnz],
 *   cedge[nnz] (CSC view: for column j the rows i and the CSR edge ids), then Dr, Dc (max row / column degree) and the
 *   padded fixed-degree views ell_rj[J][Dr], ell_rk[J][Dr], ell_ci[J][Dc], ell_ck[J][Dc] whose padding slots carry the
 *   edge id nnz.  Edge k <-> e[:, k] exactly as the reference's boolean-mask assignment enumerates (i, j) row-major
 *   (local_attention.py:41).  Adjacency buffers A_t have nnz+1 rows: row nnz is all zero (target of the padding). */

/* A_t[k][c] = softmax over the edges of row i(k) of e[c][k]   (local_attention.py:40-42); A_t[nnz][c] = 0 */
int gast_semch_adj_fwd(const float* e, int C, const int32_t* pat, float* A_t, gast_stream_t stream);
/* de[c][k] = A (dA - sum_row A dA)   (softmax backward) */
int gast_semch_adj_bwd(const float* dA_t, const float* A_t, int C, const int32_t* pat, float* de, gast_stream_t stream);

/* All adjacency softmaxes of a pass in ONE launch (they depend on parameters only): forward jobs {e, C, pat, A_t}, backward jobs
 * {dA_t, A_t, C, pat, e = de (output)}; n <= GAST_ADJ_MAX_BATCH.  backward: 0 = forward, 1 = backward (de written), 2 = backward
 * accumulating into de. */
#define GAST_ADJ_MAX_BATCH 8
typedef struct { float* e; int C; const int32_t* pat; float* A_t; const float* dA_t; } gast_adj_job;
int gast_semch_adj_multi(const gast_adj_job* jobs, int n, int backward, gast_stream_t stream);

/* Y[f*J+i, p*C + c] = A_p[c,i,i] h0_p[f*J+i, c] + sum_{j in N_p(i), j != i} A_p[c,i,j] h1_p[f*J+j, c]
 * for the two graphs p = 0 (symmetry) and 1 (connection); H holds [h0_sym | h1_sym | h0_con | h1_con] in columns
 * [0,4C) (local_attention.py:44-48).  Also emits BatchNorm partial sums over the 2C output columns
 * (bn_1 / bn_2, local_attention.py:139-140): partials[nblk][2C][2]; returns nblk through *nblk_out (host mirror:
 * gast_semch_agg_blocks). */
int gast_semch_agg_fwd(int dtype, const void* H, int ldh, int F, int J, int C,
                       const float* A_sym, const int32_t* pat_sym, int deg_sym, const float* A_con, const int32_t* pat_con,
                       int deg_con, void* Y, int ldy, float* partials, const float* center_sym, const float* center_con,
                       gast_stream_t stream);   /* deg_* = Dr of the tables; center_* (nullable, [C]) are subtracted from the outputs */
int gast_semch_agg_blocks(int F, int C);
/* Backward: dH columns [0,4C) and dA = [dA_sym (nnz_sym rows) ; dA_con (nnz_con rows)] x C, fully written (no zero-fill
 * needed).  ws: workspace of gast_semch_agg_bwd_ws_floats() floats for the per-block partial rows. */
int gast_semch_agg_bwd(int dtype, const void* dY, int ldy, const void* H, int ldh, int F, int J, int C,
                       const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                       const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                       gast_stream_t stream);   /* cdeg_* = Dc of the tables */
long gast_semch_agg_bwd_ws_floats(int F, int C, int nnz_sym, int nnz_con);

/* ---- global additive joint attention, 4 heads (global_attention.py:52-82, App. A.2 of SURVEY.md) -----------
 * G = base pointer of the g columns ([rows][ldg], C columns, head h owns columns [h*Ci,(h+1)*Ci));
 * AC = base pointer of 2*nheads columns: a_h (nheads) then c_h (nheads), a_i = w_theta.theta_i, c_j = w_phi.phi_j.
 * att = softmax_j(LeakyReLU_0.2(a_i + c_j)) + C_k[h,i,j];  Y[i, hCi+c] = sum_j att_ij g[j, hCi+c]. */
int gast_attn_fwd(int dtype, const void* G, int ldg, const void* AC, int ldac, const float* C_k,
                  int F, int J, int C, int nheads, void* Y, int ldy, gast_stream_t stream);
/* Backward: dG (C cols), dAC (2*nheads cols) written; dC_k[nheads][J][J] and (optional) dbias[C + 2*nheads] = column sums of
 * [dG | dAC] over all rows (bias gradients of g / theta / phi, global_attention.py:29-34) are ACCUMULATED (caller zeroes).
 * ws: gast_attn_bwd_ws_floats() floats of scratch (per-wave partial rows), or null to force the generic kernels. */
int gast_attn_bwd(int dtype, const void* dY, int ldy, const void* G, int ldg, const void* AC, int ldac,
                  const float* C_k, int F, int J, int C, int nheads,
                  void* dG, int lddg, void* dAC, int lddac, float* dC_k, float* dbias, float* ws, gast_stream_t stream);
long gast_attn_bwd_ws_floats(int F, int J, int C, int nheads);

/* Deferred finishes (round 4).  gast_attn_bwd and gast_semch_agg_bwd end with a small reduction of per-wave / per-block partial
 * rows (dbias, dC_k += column sums; dA = column sums): six launches per training step that nothing in the backward pass waits for
 * except the adjacency-softmax backward at its very end.  The *_deferred forms skip that launch and describe it in *finish (ws ==
 * null when the call needed none); the caller keeps the workspace alive and runs all of them as ONE gast_rowsum_multi launch.
 *   job: out0[c] (+)= sum_r ws[r][c] for c < nb,  out1[c - nb] (+)= sum_r ws[r][c] for nb <= c < ncol   (a null output is skipped) */
typedef struct { const float* ws; int nrow; long ncol; long nb; float* out0; float* out1; int accumulate; } gast_rowsum_job;
#define GAST_ROWSUM_MAX_BATCH 8
int gast_rowsum_multi(const gast_rowsum_job* jobs, int n, gast_stream_t stream);
/* Aggregation backward with the BatchNorm backward of its input fused in (round 5).  dY is the gradient BEFORE the backward of
 * bn_1 | bn_2 (what the preceding GEMM's BNRELU_BWD epilogue wrote), Ypre the pre-BatchNorm aggregation output of the forward
 * ([F*J][2C]: sym half, con half, like dY), ka / kb / kc the [2C] coefficients of dx = ka*dz + kb*x + kc; the kernel applies them while
 * it stages dY, so the stand-alone gast_bn_bwd_apply pass over dY (local_attention.py:139-141 backward) never runs.
 * finish: nullable, as gast_semch_agg_bwd_deferred.
 * Only where gast_semch_agg_bwd_fuses_bn(...) returns 1 (the LDS-staged kernel: fp32 storage, column degrees of the shipped skeletons). */
int gast_semch_agg_bwd_fuses_bn(int dtype, int F, int J, int C, int nnz_sym, int cdeg_sym, int nnz_con, int cdeg_con);
int gast_semch_agg_bwd_bn(int dtype, const void* dY, int ldy, const void* Ypre, int ldyp, const float* ka, const float* kb,
                          const float* kc, const void* H, int ldh, int F, int J, int C,
                          const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                          const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                          gast_rowsum_job* finish, gast_stream_t stream);
int gast_attn_bwd_deferred(int dtype, const void* dY, int ldy, const void* G, int ldg, const void* AC, int ldac,
                           const float* C_k, int F, int J, int C, int nheads, void* dG, int lddg, void* dAC, int lddac,
                           float* dC_k, float* dbias, float* ws, gast_rowsum_job* finish, gast_stream_t stream);
int gast_semch_agg_bwd_deferred(int dtype, const void* dY, int ldy, const void* H, int ldh, int F, int J, int C,
                                const float* A_sym, const int32_t* pat_sym, int nnz_sym, int cdeg_sym, const float* A_con,
                                const int32_t* pat_con, int nnz_con, int cdeg_con, void* dH, int lddh, float* dA, float* ws,
                                gast_rowsum_job* finish, gast_stream_t stream);

/* ---- BatchNorm2d (momentum 0.1, eps 1e-5; gast_net.py:20,58-59,147,149 etc.) as a two-phase scheme ---------
 * Producers emit per-row-block partial sums; `gast_bn_finalize` turns them into the per-channel scale/shift that
 * consumers apply on load, and updates the running statistics exactly as nn.BatchNorm2d does in train mode. */
/* centered != 0: the statistics belong to a tensor stored as x - running_mean (value of running_mean before this call): the
 * running mean update becomes running_mean += momentum * mean', scale/shift/mean refer to the stored (centred) values. */
int gast_bn_finalize(const float* partials, int nblk, int ncol_total, int col0, int N, double count,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float momentum, float eps,
                     float* scale, float* shift, float* mean, float* rstd, int centered, gast_stream_t stream);
/* Job forms of the two finalizes and their multi-job launches (n <= GAST_BN_MAX_BATCH jobs, one launch): the plan finalizes
 * bn_1 + bn_2, lcat_bn + gcat_bn (and their backward twins) together. */
#define GAST_BN_MAX_BATCH 4
typedef struct {
    const float* partials; int nblk, ncol_total, col0, N; double count;
    const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* num_batches_tracked;
    float momentum, eps; float* scale; float* shift; float* mean; float* rstd; int centered;
} gast_bn_fin_job;
typedef struct {
    const float* partials; int nblk, ncol_total, col0, N; double count;
    const float* gamma; const float* mean; const float* rstd; float* dgamma; float* dbeta; float* ka; float* kb; float* kc;
    int accumulate;      /* dgamma / dbeta: += instead of = (gradient buffers that already hold a sum) */
} gast_bn_bwd_fin_job;
/* finalize + in-place apply (dz <- ka*dz + kb*x + kc) in ONE launch, for short tensors (a block streams all rows of its 32
 * columns): f.ka / f.kb / f.kc are not written. */
typedef struct { gast_bn_bwd_fin_job f; void* dz; int lddz; const void* X; int ldx; long rows; } gast_bn_bwd_job;
int gast_bn_bwd_fused_multi(int dtype, const gast_bn_bwd_job* jobs, int n, gast_stream_t stream);
/* all eval-mode BatchNorms of a model in one launch (scale/shift from the running statistics; inference path, row f4) */
#define GAST_BN_EVAL_MAX_BATCH 32
typedef struct { const float* gamma; const float* beta; const float* running_mean; const float* running_var; int N; float* scale; float* shift; int centered; } gast_bn_eval_job;
int gast_bn_eval_multi(const gast_bn_eval_job* jobs, int n, float eps, gast_stream_t stream);
int gast_bn_finalize_multi(const gast_bn_fin_job* jobs, int n, gast_stream_t stream);
int gast_bn_bwd_finalize_multi(const gast_bn_bwd_fin_job* jobs, int n, gast_stream_t stream);
int gast_bn_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                 float eps, int N, float* scale, float* shift, int centered, gast_stream_t stream);
/* partials hold {sum dz, sum dz*x}; writes dgamma, dbeta and the per-channel coefficients of
 * dx = ka*dz + kb*x + kc. */
int gast_bn_bwd_finalize(const float* partials, int nblk, int ncol_total, int col0, int N, double count,
                         const float* gamma, const float* mean, const float* rstd,
                         float* dgamma, float* dbeta, float* ka, float* kb, float* kc, gast_stream_t stream);
/* dz <- ka*dz + kb*x + kc (in place) */
int gast_bn_bwd_apply(int dtype, void* dz, int lddz, const void* X, int ldx, long rows, int N,
                      const float* ka, const float* kb, const float* kc, gast_stream_t stream);
/* The same for a tensor of rows = B * T_total * J of which only the frames t with bit t of `frames` set carry a gradient (T_total <= 64): the
 * other rows of dz are taken as zero WITHOUT being read -- they may be uninitialised -- and receive kb*X + kc.  The input gradient of
 * the last dilated temporal level (reference gast_net.py:173 with T' = 1) reaches k of its input frames only. */
int gast_bn_bwd_apply_frames(int dtype, void* dz, int lddz, const void* X, int ldx, long rows, int N, const float* ka, const float* kb,
                             const float* kc, int T_total, int J, unsigned long long frames, gast_stream_t stream);
/* Y = drop(relu(scale*X + shift)); dropout (use_drop != 0) uses stream `salt` indexed by the element offset in X.  Materialises
 * the post-activation of the local / global branch (gast_net.py:24-27) once, so that the G4 GEMM, its weight gradient and the
 * branch input gradients need neither the BatchNorm prologue nor the dropout hash (mask = [Y > 0], see epi_scale). */
int gast_bnrelu_apply(int dtype, const void* X, int ldx, long rows, int N, const float* scale, const float* shift,
                      void* Y, int ldy, int use_drop, uint32_t salt, gast_dropout drop, gast_stream_t stream);
/* dz = dY * [scale*X+shift > 0] * keep/(1-p)  + partial sums {sum dz, sum dz*x}: partials[nblk][N][2] */
int gast_bnrelu_bwd_mask(int dtype, const void* dY, int lddy, const void* X, int ldx, long rows, int N,
                         const float* scale, const float* shift, int use_drop, uint32_t salt, gast_dropout drop,
                         void* dz, int lddz, float* partials, gast_stream_t stream);
int gast_rowwise_blocks(long rows, int N);
/* Shrink layer (reference gast_net.py:99,176-178: Conv2d(2C * 2^(L-1), 3, 1, bias=False) on the rows the temporal stages leave) as a
 * row-wise dot product -- D <= 4 output columns are not a GEMM (round 6; gast_gemm accepts the same operation and is what the
 * streaming path and the 16-bit fp8 experiments still use):
 *   pred[r, d] = sum_k relu(scale[k] * O[r, k] + shift[k]) * W[d][k]      pred fp32 [rows][ldp], W [D][ldw] of the activation type
 * and its input gradient with the BatchNorm-backward column sums of GAST_EPI_BNRELU_BWD:
 *   dO[r, k] = [scale[k] * O[r, k] + shift[k] > 0] * sum_d dp[r, d] * W[d][k]
 *   partials[gast_shrink_bwd_blocks(rows)][K][2] = {sum dO, sum dO * O} over each row block, fully overwritten.
 * K % 4 == 0, 16-byte aligned rows. */
int gast_shrink_fwd(int dtype, const void* O, int ldo, long rows, int K, const float* scale, const float* shift, const void* W, int ldw,
                    int D, float* pred, int ldp, gast_stream_t stream);
int gast_shrink_bwd_blocks(long rows);
int gast_shrink_bwd(int dtype, const void* dp, int lddp, const void* W, int ldw, int D, const void* O, int ldo, const float* scale,
                    const float* shift, long rows, int K, void* dO, int lddo, float* partials, gast_stream_t stream);
/* Residual of a temporal block (gast_net.py:170-174 / :243-247):
 * Xn[m] = relu(scO*O[omap(m)] + shO) + keep/(1-p) * relu(sc2*T2[m] + sh2) */
int gast_residual_fwd(int dtype, const void* O, int ldo, gast_rowmap omap, const float* scO, const float* shO,
                      const void* T2, int ldt, const float* sc2, const float* sh2,
                      int use_drop, uint32_t salt, gast_dropout drop,
                      int B, int Tn, int J, int N, void* Xn, int ldxn, gast_stream_t stream);

/* ---- input side: init_bn + expand_conv (gast_net.py:58,130-131,163-164,207) -------------------------------- */
/* partial sums {sum x, sum x^2} of the (rows, F_in) fp32 network input: partials[nblk][F_in][2] */
int gast_input_stats(const float* x, long rows, int F_in, float* partials, int* nblk_out, gast_stream_t stream);
int gast_input_stats_blocks(long rows);
/* E[(b,t,j), c] = sum_{f,tap} W[c][f][tap] * (sc0[f]*x[(b, t*t_stride+tap, j), f] + sh0[f]) + partial sums for expand_bn */
int gast_expand_fwd(int dtype, const float* x, int B, int T_in, int J, int F_in, int k0, int t_stride,
                    const float* W, const float* sc0, const float* sh0, int C,
                    void* E, int lde, float* partials, const float* center, gast_stream_t stream);   /* center: nullable [C], subtracted */
/* Backward of init_bn + expand_conv w.r.t. their parameters (reference gast_net.py:163-164; the input needs no gradient).
 * With G[c][f][tap] = sum_m dE[m,c]*xhat[(b,t*ts+tap,j), f] and S[c] = sum_m dE[m,c]:
 *   dW[c][f][tap] = gamma0[f]*G + beta0[f]*S[c]      (written, or += when accumulate != 0)
 *   dgamma0[f] += sum_{c,tap} W*G,  dbeta0[f] += sum_{c,tap} W*S[c]      (atomics: pass them zero-filled)
 * ws: gast_expand_bwd_ws_floats(rows = B*T_out*J, C, F_in, k0) floats of scratch. */
int gast_expand_bwd(int dtype, const void* dE, int ldde, const float* x, int B, int T_in, int J, int F_in, int k0,
                    int t_stride, const float* mean0, const float* rstd0, int C, const float* W, const float* gamma0,
                    const float* beta0, float* dW, float* dgamma0, float* dbeta0, float* ws, int accumulate, gast_stream_t stream);
long gast_expand_bwd_ws_floats(long rows, int C, int F_in, int k0);
/* The same with the backward of expand_bn fused in (round 5): dE is the gradient BEFORE that BatchNorm backward, Epre ([rows][lde]) the
 * pre-BatchNorm output of the expand conv, ka / kb / kc the [C] coefficients of dz = ka*dE + kb*Epre + kc applied on load.
 * Epre == NULL: exactly gast_expand_bwd. */
int gast_expand_bwd_bn(int dtype, const void* dE, int ldde, const void* Epre, int lde, const float* ka, const float* kb, const float* kc,
                       const float* x, int B, int T_in, int J, int F_in, int k0,
                       int t_stride, const float* mean0, const float* rstd0, int C, const float* W, const float* gamma0,
                       const float* beta0, float* dW, float* dgamma0, float* dbeta0, float* ws, int accumulate, gast_stream_t stream);

/* out[n] (+)= sum_m X[m, n]   (bias gradients of the g / theta / phi 1x1 convs, global_attention.py:30-35) */
int gast_colsum(int dtype, const void* X, int ldx, long rows, int N, float* out, int zero_first, gast_stream_t stream);

/* ---- parameter packing / gradient unpacking (replaces ~600 tiny ATen kernels per step; reference layouts: App. D) ----
 * Pointer words: (byte_offset << 4) | base_id, address = bases[base_id] + byte_offset; `bases` is a HOST array of 8 byte
 * addresses copied into the launch (base 0 is conventionally 0, i.e. absolute addresses; the others let a table built once
 * address buffers that move between calls, e.g. the gradient buffer).  Strides are in elements of the respective type.
 * copy job (10 int64): src word, dst word, R, S, src_row_stride, src_col_stride, dst_row_stride, dst_col_stride,
 *   flags (1 = src bf16, 2 = dst bf16, 4 = accumulate into fp32 dst, 8 = zero-fill dst), reserved.
 *   `tiles`: int32 triples (job, row tile, col tile) of 32x32 tiles, one block each.
 * fold job (12 int64): W word [Ci][C] fp32, w word [Ci], b word [Ci], Ci, C, dst_row word, dst_row stride, dst_col word,
 *   dst_col stride, bias_dst word (fp32), dst_bf16, reserved:   v[k] = sum_m W[m][k] w[m];  bias = sum_m w[m] b[m]
 *   (theta/phi folding of the additive attention, global_attention.py:60-74).
 * unfold job (12 int64): dv word, da word, W, w, b, dW, dw, db words, Ci, C, accumulate, reserved. */
int gast_strided_copy(const int64_t* jobs, const int32_t* tiles, int ntiles, const int64_t* bases, gast_stream_t stream);
/* The whole parameter packing of a step in ONE launch (round 4; was gast_strided_copy + gast_fold + gast_x3_image_multi): copy jobs of
 * 16 int64 = the 10 words above + {image word (0: none), ldimg, element offset of the job's element (0, 0) inside its packed operand,
 * columns K of that operand, image kind (gast_x3_image_job.f16: 0 bf16 pairs, 1 fp16 pairs, 2 = round 6: the layout image of a 16-bit
 * operand), reserved} -- the tile is also written into the operand's image (layout of gast_x3_image_multi; the destination must be
 * K-contiguous or a transposed twin, i.e. one of its strides 1) -- and fold jobs of 18 int64 = the 12 words above + {image word of K
 * position 0 of the destination row (k-group 0), its ldimg, image word of the destination column's K position in operand row 0,
 * reserved, fp16-pair flags (bit 0 row, bit 1 column), 1 = both images are layout images of 16-bit operands (kind 2)}. */
int gast_pack_all(const int64_t* cjobs, const int32_t* tiles, int ntiles, const int64_t* fjobs, int nfold, int max_C,
                  const int64_t* bases, gast_stream_t stream);
int gast_fold(const int64_t* jobs, int njobs, int max_C, const int64_t* bases, gast_stream_t stream);      /* max_C  = largest C of the jobs */
int gast_unfold(const int64_t* jobs, int njobs, int max_Ci, const int64_t* bases, gast_stream_t stream);  /* max_Ci = largest Ci */

/* library identification: returns a static string "gast_hip <version> gfx950" */
/* ---- training-step tail: loss and optimiser on flat fp32 buffers ------------------------------------ */
/* mpjpe (reference common/loss.py:5-11): *loss = mean_r ||pred[r,:] - target[r,:]||_2 over `rows` rows of D <= 4 components;
 * dirs[r,:] = d loss / d pred[r,:] (zero where the difference vanishes).  All fp32, contiguous. */
int gast_mpjpe(const float* pred, const float* target, long rows, int D, float* loss, float* dirs, gast_stream_t stream);
/* One Adam / AMSGrad step (torch.optim.Adam semantics, reference trainval.py:78) over flat fp32 buffers of n elements:
 * p, m, v (and vmax: non-null selects amsgrad) updated in place from g * grad_scale; *step (device int32) is incremented first
 * and supplies the bias corrections.  16-byte aligned buffers. */
int gast_adam_step(float* p, const float* g, float* m, float* v, float* vmax, long n, int* step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, gast_stream_t stream);
/* The same behind a guard (the loss-scaled 16-bit mode; like torch.cuda.amp.GradScaler's skipped steps -- the reference trains in fp32 and
 * has no such step).  gast_nonfinite_scan: flags[b] = 1 when block b's slice of the flat gradient holds an inf / NaN, else 0, for all
 * GAST_NONFINITE_FLAGS blocks (every word is rewritten on every call).  gast_adam_step_guarded: `skip` (nullable) points at those words;
 * when any is set the call is a no-op -- parameters, moments and *step untouched -- and *skipped (nullable, device int64) is incremented. */
#define GAST_NONFINITE_FLAGS 256
int gast_nonfinite_scan(const float* g, long n, int* flags, gast_stream_t stream);
int gast_adam_step_guarded(float* p, const float* g, float* m, float* v, float* vmax, long n, int* step, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float grad_scale, const int* skip, long long* skipped,
                           gast_stream_t stream);

/* Pass prologue, ONE launch: zero-fill up to GAST_PREP_MAX_ZERO regions (16-byte aligned, sizes multiples of 16: the accumulation
 * arena of the pass, the flat gradient buffer, the packed-gradient scratch), optionally advance the dropout seed (*seed_out =
 * ++*seed_ctr: the per-pass copy the kernels of this pass -- and its backward -- read) and optionally pad d loss / d pred from
 * pad_cols_src to pad_cols_dst columns (pad_dst[r][c] = c < pad_cols_src ? pad_scale * pad_src[r][c] : 0, contiguous, fp32 source, pad_rows rows;
 * reference main.py:231-237: the 3 output coordinates, 8 columns for the shrink layer's gradient GEMMs).  Replaces the torch
 * fill / add / clone / slice-copy nodes at the head of a forward and of a backward pass. */
#define GAST_PREP_MAX_ZERO 6
typedef struct { void* ptr; long bytes; } gast_zero_job;
typedef struct {
    gast_zero_job zero[GAST_PREP_MAX_ZERO];
    int nzero;
    uint32_t* seed_ctr;      /* nullable */
    uint32_t* seed_out;      /* nullable */
    const float* pad_src;    /* nullable (pad_rows == 0) */
    float* pad_dst;
    long pad_rows;
    int pad_cols_src, pad_cols_dst;
    float pad_scale;         /* round 6: the copied values are multiplied by this (0 means 1: the loss scale of GAST_HIP_DTYPE=f16) */
    int pad_dst_h16;         /* round 6: pad_dst holds the library's 16-bit storage type (bfloat16 / binary16 by build flavour) instead of fp32 */
} gast_prep_args;
int gast_prep(const gast_prep_args* args, gast_stream_t stream);

/* ---- causal streaming inference (reference gen_skes.py:43-69, tools/inference.py:73-91 re-run the whole
 * receptive field per frame): the per-level frame windows of gast_hip/streaming.py advance by one frame -- every window of the model
 * in ONE launch.  buf: [B][Tb][X] fp32 (X = J * channels, a multiple of 4), shifted in place by one frame towards t = 0; newest:
 * [B][ldnew >= X] fp32, copied into frame Tb - 1. */
#define GAST_STREAM_SHIFT_MAX 8
typedef struct {
    void* buf;
    const void* newest;
    int B, Tb, X, ldnew;
} gast_stream_shift_job;
int gast_stream_shift_multi(const gast_stream_shift_job* jobs, int n, gast_stream_t stream);

/* ---- data side: device-resident ChunkedGenerator (reference common/generators.py:93-159) ---------------- */
/* Builds one training batch from sequences resident in HBM.  poses2d: all sequences concatenated, [sum_len][J2][F2] fp32; seq_off:
 * nseq+1 frame offsets; pairs: the epoch's (seq, start_3d, end_3d, flip) int32 rows (generators.py:33-42, already shuffled),
 * the batch being rows [first_pair, first_pair + B).  out2d[i] = the window [start_3d - pad - causal_shift, end_3d + pad -
 * causal_shift) with edge replication outside the sequence (np.pad "edge", generators.py:100-110); flipped samples get x -> -x and
 * the joint permutation perm2d (destination joint j reads source joint perm2d[j]; built from kps_left/kps_right,
 * generators.py:112-115).  poses3d / out3d (chunk frames, perm3d) and cams / outcam (ncam coefficients, 2 and 7 negated when
 * flipped) are optional (both null or both non-null). */
int gast_chunk_gather(const float* poses2d, const float* poses3d, const float* cams, const int64_t* seq_off, const int32_t* pairs,
                      long first_pair, int B, int chunk, int pad, int causal_shift, int J2, int F2, int J3, int F3, int ncam,
                      const int32_t* perm2d, const int32_t* perm3d, float* out2d, float* out3d, float* outcam, gast_stream_t stream);

/* launches an empty kernel: the fixed dispatch cost an event pair sees around any launch (bench.py calibration) */
int gast_null_launch(gast_stream_t stream);
const char* gast_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GAST_HIP_H */
