// gast_wgrad, bf16x3 arithmetic, 256 x 256 output tiles: ONE eight-wave block per CU (two waves per SIMD), three tiles of operand
// rows in flight per thread.
//
//   dW[r, wcol0_s + k] += sum_m P[pmap(m), r] * pro_s(Q_s[map_s(m), k])          (wgrad.hip has the contract)
//
// Why a second kernel.  The 128 x 128-tile kernel (wgrad_x3_pipe_body) moves 32 KB of operands per 3.1 MFLOP-triple, and the round-4
// ablation builds (scripts/r4_wg_ablate.sh, WG_ABLATE below) say the operand loads are what it waits for: without them a launch takes
// 55 % of its time, a third of the MFMA work buys 2 %.  A 256 x 256 tile halves the loads per FLOP:
//   * a wave owns 128 x 64 of the tile (8 accumulator tiles of 32 x 32 = 128 registers); a step is 16 reduction rows: 24 MFMAs per
//     wave against 12 fragment reads, 4 operand loads and the conversion of 16 values per thread;
//   * 32 KB of operands per step and CU for 6.3 MFLOP-triples, three steps ahead in registers (WD sets of 4 x 16 bytes);
//   * the next tile of a set is requested BEFORE the step's barrier (the set is thread-private; only the LDS stages need the
//     barrier), from row offsets that wave 0 wrote one step earlier.
// Measured (scripts/wgrad_multi_bench.py, one box): C = 256 stage 490 -> 390 us, M = B*J stage 128 -> 99 us; the C = 128 stage's
// matrices are 128 wide and stay on the narrow kernel (207 vs 139 us).  What was tried on the way: FOUR waves of 128 x 128 with
// 256 accumulator registers each (one wave per SIMD, 512 registers: correct, 438 us -- a lone wave issues in order, so its
// conversion and its MFMAs only overlap as far as the compiler interleaves them, and every LDS / barrier latency is exposed);
// 4 or 5 sets in flight (no change: the loads are not latency-bound); cache-policy bits on the loads (sc0 / nt / sc1: no change);
// an explicit sched_group_barrier interleave (worse: the conversions end up behind the MFMAs).
// Staging as in the narrow kernel: waves 0-3 stage dC (P), waves 4-7 the activation (Q, with the BN+ReLU(+dropout) prologue); a
// thread owns a 4(m) x 4(col) block per step, a QUAD of lanes reads 64 contiguous bytes of one row, rows outside the chunk / the
// row map get an out-of-range byte offset (the buffer load returns zeros without touching memory), the hi/lo split packs PAIRS of
// reduction rows so that the m-major -> col-major transposition is free.  LDS: two stages of [P cols | Q cols] x 80 B
// ([hi: 16 m = 32 B | lo: 32 B] + 16 B pad) = 80 KB, plus six generations of row offsets.
#include "common.h"
#include "wgrad_common.h"
#include <type_traits>

namespace {

#ifndef WG_ABLATE
#define WG_ABLATE 0                         // profiling builds (EXTRA_FLAGS): 1 no operand loads, 2 loads inside one 64 KB window, 3 no MFMA, 4 no conversion
#endif
constexpr int WT = 256;                     // tile edge
constexpr int WK = 16;                      // reduction rows per step (bf16x3; 16-bit storage: 32, see H16 below)
constexpr int WSTR = 80;                    // LDS bytes per tile column and step
constexpr int WSTAGE = 2 * WT * WSTR;       // P columns, then Q columns
constexpr int WD = 3;                       // register sets (tiles in flight)
constexpr int WUNROLL = 6;                  // lcm(WD, 2 stages); also the number of row-offset generations
constexpr uint32_t WW_OOB = 0xFFFF0000u;    // buffer size == first out-of-range byte offset (wgrad_check bounds the operands)

// H16 (round 5): the same kernel on 16-BIT STORAGE (GAST_BF16 tensors: bfloat16, or binary16 in the -DGAST_H16_F16 build), one product.
// A step covers 32 reduction rows: the LDS image of a tile column is [32 m x 16 bit] = the 64 bytes that hold [16 hi | 16 lo] in the
// split kernel, so the fragment addresses do not change -- the first 32 bytes feed the first 16-deep MFMA, the second 32 the second
// (16 MFMAs per wave and step for twice the rows).  A thread owns 4(m) x 8(col): the same four 16-byte loads per set, the same eight
// 8-byte LDS writes; without a prologue the m-major -> col-major transposition is eight v_perm pairs, with one the values pass
// through fp32 (BN + ReLU (+ dropout)) and are packed in PAIRS of reduction rows like the split kernel's hi / lo words.
template <bool DROP, bool H16>
__device__ __forceinline__ void wgrad_x3_wide_body(unsigned char* smem, const gast_wgrad_args& a, int M, int tilesS_total, int mchunk, int tile, int sp) {
    constexpr int WKS = H16 ? 32 : WK;          // reduction rows per step
    constexpr int CPT = H16 ? 8 : 4;            // columns per thread (one 16-byte load per row)
    constexpr uint32_t ESZ = H16 ? 2u : 4u;     // bytes per stored element
    __shared__ __attribute__((aligned(16))) uint32_t sOffP[WUNROLL][32];      // byte offsets of the rows of tile j: generation j % 6
    __shared__ __attribute__((aligned(16))) uint32_t sOffQ[WUNROLL][32];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;      // 8 waves: 2 x 4 of 128 x 64
    const int li = lane & 31, lh = lane >> 5;
    const TileCoord tc = decode_tile(a, tile, tilesS_total, WT);
    const gast_wgrad_seg& sg = a.seg[tc.seg];
    const int m_begin = sp * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    if (m_begin >= m_end) return;
    const int ntile = ((m_end - m_begin + WKS - 1) / WKS + WUNROLL - 1) / WUNROLL * WUNROLL;

    const int op = __builtin_amdgcn_readfirstlane(w >> 2);               // 0: dC (P) staging waves, 1: activation (Q) staging waves
    const int task = tid & 255;
    // 4(m) x 4(col) per thread (H16: 4 x 8); a quad of lanes = 64 contiguous bytes of one row
    const int mb = H16 ? (task >> 2) & 7 : (task >> 2) & 3, rc = H16 ? ((task >> 5) << 2) | (task & 3) : ((task >> 4) << 2) | (task & 3);
    const float* base = op == 0 ? (const float*)a.P : (const float*)sg.Q;
    const int col = (op == 0 ? tc.rt : tc.st) * WT + rc * CPT;
    const bool cin = col < (op == 0 ? a.R : sg.S);
    const bool pro = op == 1 && sg.pro != GAST_PRO_NONE;
    const bool drop = DROP && op == 1 && sg.pro == GAST_PRO_BNRELU_DROP && a.drop.thresh != 0;
    const uint32_t key = drop ? drop_key(a.drop, sg.salt) : 0u;
    // prologue constants of this thread's 4 columns; identity (scale 1, shift 0, clamp -inf) without a prologue; columns past S
    // feed only outputs that are never stored
    float sc[CPT], sh[CPT];
    const float lowclamp = pro ? 0.f : -__builtin_inff();
#pragma unroll
    for (int q = 0; q < CPT; ++q) { sc[q] = 1.f; sh[q] = 0.f; }
    if (pro && cin) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) { sc[q] = sg.scale[col + q]; sh[q] = sg.shift[col + q]; }
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) asm volatile("" : "+v"(sc[q]), "+v"(sh[q]));      // (the compiler's wait for these loads lands here, not in the loop)
    const uint32_t colbytes = (uint32_t)(cin ? col : 0) * ESZ;
    const int sdst_off = (op == 0 ? 0 : WT * WSTR) + rc * CPT * WSTR + mb * 8;
    // buffer resource of this wave's operand: base, stride 0, WW_OOB bytes, raw 32-bit dwords
    const float* const sbase = (const float*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)base >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)base));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, (int)WW_OOB, 0x00020000);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const uint32_t ldp4 = (uint32_t)a.ldp * ESZ, ldq4 = (uint32_t)sg.ldq * ESZ;
    auto compute_rows = [&](int it, int gen) {      // wave 0 only: byte offsets of the 16 (32) rows of tile `it`
        if (tid < WKS) {
            const int m = m_begin + it * WKS + tid;
            int pr, qr;
            rows_for(a, sg, m < m_end ? m : M, M, pr, qr);
            sOffP[gen][tid] = pr < 0 ? WW_OOB : (uint32_t)pr * ldp4;
            sOffQ[gen][tid] = qr < 0 ? WW_OOB : (uint32_t)qr * ldq4;
        }
    };

    u32x4 S[WD][4];
    auto load_tile = [&](u32x4 (&s)[4], int gen) {
        const uint4 r0 = *(const uint4*)((op == 0 ? sOffP[gen] : sOffQ[gen]) + mb * 4);
        const uint32_t o[4] = {r0.x + colbytes, r0.y + colbytes, r0.z + colbytes, r0.w + colbytes};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if WG_ABLATE == 1
            asm volatile("" : "=v"(s[i]));
#elif WG_ABLATE == 2
            s[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o[i] & 0xFFF0u), 0, 0));
#else
            s[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o[i], 0, 0));
#endif
        }
    };
    auto store_tile = [&](const u32x4 (&s)[4], int gen, unsigned char* stage, auto opc) {
        constexpr int OP = decltype(opc)::value;
#if WG_ABLATE == 4
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(s[i]));
        return;
#endif
        if constexpr (H16) {
            unsigned char* const d0 = stage + sdst_off;
            if (OP == 0 || !pro) {
                // no prologue: a pure 4 x 8 transposition of 16-bit values -- low halves of word p of rows (0,1) / (2,3) = column 2p, high = 2p+1
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t r0 = s[0][p], r1 = s[1][p], r2 = s[2][p], r3 = s[3][p];
                    *(uint2*)(d0 + (2 * p) * WSTR) = make_uint2(__builtin_amdgcn_perm(r1, r0, 0x05040100u), __builtin_amdgcn_perm(r3, r2, 0x05040100u));
                    *(uint2*)(d0 + (2 * p + 1) * WSTR) = make_uint2(__builtin_amdgcn_perm(r1, r0, 0x07060302u), __builtin_amdgcn_perm(r3, r2, 0x07060302u));
                }
                return;
            }
            float x[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) h16x2_unpack(s[i][p], x[i][2 * p], x[i][2 * p + 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) x[i][q] = fmaxf(fmaf(x[i][q], sc[q], sh[q]), lowclamp);
            if (DROP && drop) {
                const uint4 r0 = *(const uint4*)(sOffQ[gen] + mb * 4);
                const uint32_t e[4] = {(r0.x + colbytes) >> 1, (r0.y + colbytes) >> 1, (r0.z + colbytes) >> 1, (r0.w + colbytes) >> 1};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[i][q] *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e[i] + q);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *(uint2*)(d0 + q * WSTR) = make_uint2(pack_h16x2(x[0][q], x[1][q]), pack_h16x2(x[2][q], x[3][q]));
            return;
        } else {
        float x[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i][0] = __uint_as_float(s[i].x); x[i][1] = __uint_as_float(s[i].y);
            x[i][2] = __uint_as_float(s[i].z); x[i][3] = __uint_as_float(s[i].w);
        }
        if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) x[i][q] = fmaxf(fmaf(x[i][q], sc[q], sh[q]), lowclamp);
            if (DROP && drop) {
                // (the element index of the dropout stream = byte offset / 4: re-read from the tile's offset generation, which
                //  lives until the tile after next is converted -- cheaper than four more registers per set)
                const uint4 r0 = *(const uint4*)(sOffQ[gen] + mb * 4);
                const uint32_t e[4] = {(r0.x + colbytes) >> 2, (r0.y + colbytes) >> 2, (r0.z + colbytes) >> 2, (r0.w + colbytes) >> 2};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[i][q] *= drop_mul(key, a.drop.thresh, a.drop.inv_keep, e[i] + q);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t h[2], l[2];
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const float x0 = x[2 * p2][q], x1 = x[2 * p2 + 1][q];
                h[p2] = pack_bf16x2(x0, x1);
                l[p2] = pack_bf16x2(x0 - __uint_as_float(h[p2] << 16), x1 - __uint_as_float(h[p2] & 0xffff0000u));
            }
            unsigned char* d = stage + sdst_off + q * WSTR;
            *(uint2*)d = make_uint2(h[0], h[1]);
            *(uint2*)(d + 32) = make_uint2(l[0], l[1]);
        }
        }
    };
    auto mfma_tile = [&](const unsigned char* stage) {
#if WG_ABLATE == 3
        return;
#endif
        const unsigned char* const sP = stage + (wr * 128 + li) * WSTR + lh * 16;
        const unsigned char* const sQ = stage + WT * WSTR + (wc * 64 + li) * WSTR + lh * 16;
        union F { uint4 u; s16x8 s; };
        F ah[4], al[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            ah[mi].u = *(const uint4*)(sP + mi * 32 * WSTR);
            al[mi].u = *(const uint4*)(sP + mi * 32 * WSTR + 32);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            F bh, bl;
            bh.u = *(const uint4*)(sQ + ni * 32 * WSTR);
            bl.u = *(const uint4*)(sQ + ni * 32 * WSTR + 32);
            if constexpr (H16) {      // reduction rows 0..15 (bytes 0..31 of the column image), then 16..31
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma_h16(ah[mi].u, bh.u, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma_h16(al[mi].u, bl.u, acc[mi][ni]);
                continue;
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi].s, bh.s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bl.s, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi].s, bh.s, acc[mi][ni], 0, 0, 0);
        }
    };
    unsigned char* const stage0 = smem;
    unsigned char* const stage1 = smem + WSTAGE;

    auto run = [&](auto opc) {
        // prologue: tiles 0 .. WD requested (tile 0 converted into stage 0 on the way, its set reused for tile WD), offsets of tile WD+1
#pragma unroll
        for (int j = 0; j <= WD; ++j) {
            if (w == 0) compute_rows(j, j);
            __syncthreads();
            if (j == WD) store_tile(S[0], 0, stage0, opc);
            load_tile(S[j % WD], j);
        }
        if (w == 0) compute_rows(WD + 1, WD + 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // Step t = it + u (top: stage t & 1 holds tile t, the sets hold tiles t+1 .. t+WD): the MFMAs of tile t and the conversion of
        // tile t+1 (set (t+1) % WD -> stage (t+1) & 1) as ONE region; tile t+1+WD requested into the freed set; wave 0 writes the
        // offsets of tile t+2+WD (generation (t+5) % 6 = tile t-1's, last read when that tile was converted in step t-2); barrier.
        // Tiles past the chunk: every offset out of range, zeros, no traffic.
        for (int it = 0; it < ntile; it += WUNROLL) {
#pragma unroll
            for (int u = 0; u < WUNROLL; ++u) {
                mfma_tile((u & 1) ? stage1 : stage0);
                store_tile(S[(u + 1) % WD], (u + 1) % WUNROLL, (u & 1) ? stage0 : stage1, opc);
                load_tile(S[(u + 1) % WD], (u + 1 + WD) % WUNROLL);
                if (decltype(opc)::value == 0 && w == 0) compute_rows(it + u + 2 + WD, (u + 2 + WD) % WUNROLL);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);      // (the next step's conversions must not move up: they would wait for the loads just issued)
            }
        }
    };
    if (op == 0) run(std::integral_constant<int, 0>());
    else run(std::integral_constant<int, 1>());

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int scol = tc.st * WT + wc * 64 + ni * 32 + li;
        if (scol >= sg.S) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrow = tc.rt * WT + wr * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rrow < a.R) atomicAdd(a.dW + (long)rrow * a.ldw + sg.wcol0 + scol, acc[mi][ni][r]);
            }
    }
}

template <bool DROP, bool H16 = false>
__global__ void __launch_bounds__(512, 2) wgrad_x3_wide_multi_kernel(const WgBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem_wide[];
    int d, tile, sp;
    if (!wg_decode(b, d, tile, sp)) return;
    wgrad_x3_wide_body<DROP, H16>(dsmem_wide, b.a[d], b.M[d], b.tilesS[d], b.mchunk[d], tile, sp);
}

}  // namespace

int gast_wgrad_x3_wide_launch(const WgBatch& b, unsigned grid, bool any_drop, hipStream_t st, bool h16) {
    typedef void (*kern_t)(const WgBatch);
    const kern_t kern = h16 ? (any_drop ? wgrad_x3_wide_multi_kernel<true, true> : wgrad_x3_wide_multi_kernel<false, true>)
                            : (any_drop ? wgrad_x3_wide_multi_kernel<true> : wgrad_x3_wide_multi_kernel<false>);
    // (hipFuncSetAttribute is per device and nn.DataParallel replicas launch from several threads: set on every launch)
    const hipError_t at = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE);
    if (at != hipSuccess) return (int)at;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * WSTAGE, st, b);
    GAST_CHECK_LAUNCH();
    return 0;
}
