// PROTOTYPE (not built, not part of the product): the wave-specialised variant of csrc/gemm_big.hip tried in round 2 --
// 512 threads, one block per CU, waves 0-3 issue only MFMAs (fragments of the next 16-deep half prefetched under the current
// half's 24 MFMAs), waves 4-7 stage operands (weights by LDS-DMA into a ring of six half buffers, activations through two register
// sets into three LDS stages, prefetch distance 2).  It is CORRECT (it passed tests/test_kernels_gpu.py -k gemm_big as the
// product kernel) and measured, MI355X, B = 128: G4 s1 (41344 x 512, K = 768) 119 us vs 130 us for the product kernel (2 blocks
// per CU), K = 2048 172 vs 178 us, but G1 s0 (K = 128) 63 vs 50 us and G1 s1 (N = 1288, K = 256) 153 vs 124 us: with one
// block per CU the epilogues are not covered.  Both variants move ~9 TB/s of operand traffic L2 -> LDS at a 128 x 256 tile,
// which is what bounds them at ~0.8 PF-eq/s.  Drop-in for csrc/gemm_big.hip (same interface).
// gast_gemm, large-M path for GAST_F32X3 (fp32 storage, split-bf16 products on v_mfma_f32_32x32x16_bf16), gfx950.
//
// Same contract as gemm.hip (K segments with row maps = channel concat / temporal taps of reference gast_net.py:28-32,145-148,
// 173-174; BN+ReLU load prologue; STATS / BNRELU_BWD epilogues), different machine mapping.  The 128x128 two-barrier loop of
// gemm.hip runs the split-bf16 products at 19 % of the matrix-core peak (rocprof, profiles/r02_v0_*); a first 256x256-tile,
// one-block-per-CU version of this file only tied it: with a prefetch distance of ONE K tile every iteration waited out the
// memory latency (3.1 us per K tile against 1.3 us of MFMA work) and with one block per CU the 256 KB epilogues of all CUs ran
// in lockstep with idle matrix cores.  Hence:
//   * block tile 128 x 256, 256 threads = 4 waves (2 x 2), wave tile 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers),
//     TWO blocks per CU (2 waves per SIMD, 256 VGPRs each): the blocks drift apart, so one block's epilogue / barrier stalls are
//     covered by the other's MFMAs;
//   * K step = 16 fp32 values per row, held in LDS as a 64-byte row image [16 bf16 hi | 16 bf16 lo]; the 16-byte chunks are
//     XOR-swizzled by (row>>2)&3 so the fragment reads (ds_read_b128) are conflict-free;
//   * prefetch distance TWO for both operands, one barrier per K step: weights stream global -> LDS by DMA
//     (global_load_lds_dwordx4 from the pre-split weight image, gast_x3_image_multi) into a ring of three stages; activations
//     pass through two register sets (the BN+ReLU prologue and the hi/lo split are VALU work) into two LDS stages; one counted
//     s_waitcnt vmcnt(6) per iteration leaves the newest step's 4 DMA + 2 loads in flight (every iteration issues exactly
//     that many -- past the last tile they re-request the last one -- so the count is exact);
//   * 24 MFMAs per wave per K step against 12 ds_read_b128;
//   * epilogue straight from the accumulators (in the 32x32 layout a lane owns one column: a store instruction writes two
//     128-byte row segments), X / addend values fetched one 4-row unit ahead; the column statistics of a block's 128 rows
//     are exactly one statistics block (partials[ceil(M/128)][N][2], the layout gemm.hip and the BatchNorm finalizes share).
#include "common.h"
#include "gemm_big.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int TM = 128, TN = 256, TK = 32;    // K tile = two 16-deep halves
constexpr int ROWB = 64;                        // LDS row image of a half: 16 bf16 hi | 16 bf16 lo
constexpr int A_HALF = TM * ROWB;               // 8 KB
constexpr int A_BYTES = 2 * A_HALF;             // activation stage = one K tile = two halves
constexpr int W_BYTES = TN * ROWB;              // weight buffer = one half: 16 KB
constexpr int OFF_ROWS = 0;                     // crow[128] | addrow[128]
constexpr int OFF_A = 2 * TM * 4;               // three activation stages
constexpr int OFF_W = OFF_A + 3 * A_BYTES;      // six weight half buffers
constexpr int OFF_TAB = OFF_W + 6 * W_BYTES;    // packed producer row positions [4][256] | scale | shift tables
constexpr int LDS_BLOCK = 160 * 1024;           // one block per CU
constexpr int MAX_TAB = (LDS_BLOCK - OFF_TAB - 4 * 256 * 4) / 8;

// 16 bytes per lane global -> LDS (DMA): address = sbase + voff + OFF; lands at lds_wave_base + 16 * lane
template <int OFF>
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(OFF) : "memory", "m0");
}
// 16 bytes per lane global -> registers, address = sbase + voff (scalar base: the per-step advance costs no VALU)
__device__ __forceinline__ void gload16s(u32x4& dst, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// EPI: 0 PLAIN, 1 STATS, 2 BNRELU_BWD, 3 BNRELU_BWD with the dropout mask of the forward re-derived (compile-time: the
// epilogue is straight-line code per element); ADD: an addend tensor is present
template <int EPI, bool ADD>
__device__ __forceinline__ void big_body(const gast_gemm_args& a, const BigPlan& pl, int blk, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool producer = w >= 4;                          // waves 0-3: MFMA (2 x 2 of 64 x 128), waves 4-7: operand staging
    const int cw = w & 3, wr = cw >> 1, wc = cw & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M, N = a.N;
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    const int mt = lb / pl.tilesN, nt = lb - mt * pl.tilesN;
    const int m0 = mt * TM, n0 = nt * TN;

    int* const sCrow = (int*)(smem + OFF_ROWS);
    int* const sAdd = sCrow + TM;
    int* const sPos = (int*)(smem + OFF_TAB);              // [4][256]: packed (b, t, j) of the producer threads' rows, -1: row past M
    float* const sSc = (float*)(smem + OFF_TAB + 4 * 256 * 4);
    float* const sSh = sSc + pl.ntab;

    const int TJ = a.Tn * a.J;
    if (tid < TM) {
        const int m = m0 + tid;
        int crow = -1, arow = -1;
        if (m < M) {
            const int b = m / TJ, rem = m - b * TJ, t = rem / a.J, j = rem - t * a.J;
            crow = (int)map_row(a.cmap, b, t, j, a.J);
            if (a.addend) arow = (int)map_row(a.addmap, b, t, j, a.J);
        }
        sCrow[tid] = crow;
        sAdd[tid] = arow;
    }
    int ntile = 0;                                          // K tiles of 32 (two 16-deep halves each) over all segments
    for (int s = 0; s < a.nseg; ++s) ntile += (a.seg[s].K + TK - 1) / TK;

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (producer) {
        // ================================================================================ producer waves
        const int ptid = tid - 256, pw = ptid >> 6;
        // activations (registers): rows rbase + 32 i (i < 4), 16-byte chunk c (4 fp32 values) of the 32-wide K tile: a thread's
        // chunk is one quarter of a 64-byte half row, a wave instruction covers 8 full 128-byte lines
        const int c = ptid & 7, rbase = ptid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rbase + 32 * i;
            int packed = -1;
            if (m < M) { const int b = m / TJ, rem = m - b * TJ, t = rem / a.J, j = rem - t * a.J; packed = (b << 16) | (t << 5) | j; }
            sPos[i * 256 + ptid] = packed;
        }
        uint32_t offA[4];
        bool zrow[4];
        int seg_a = -1;
        auto enter_a = [&](int s) {
            if (s == seg_a) return;
            seg_a = s;
            const gast_gemm_seg& sg = a.seg[s];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pk = sPos[i * 256 + ptid];
                const int b = pk >> 16, t = (pk >> 5) & 0x7ff, j = pk & 31;
                const int ts = t * sg.map.t_stride + sg.map.t_off;
                const bool ok = pk >= 0 && ts >= 0 && ts < sg.map.T_total;
                const uint32_t srow = ok ? (uint32_t)((b * sg.map.T_total + ts) * a.J + j) : 0u;
                zrow[i] = !ok;                                   // out-of-range tap (or a row past M): reads as zero
                offA[i] = (srow * (uint32_t)sg.lda + c * 4) * 4u;
            }
        };
        // weights (DMA): wave pw fills the 16-row pieces (pw*4 + i) of a 256 x 64 B half tile = ONE contiguous 16 KB block of the
        // k-group-major image; lane = (row r16, slot s4), slot s4 receives source chunk s4 ^ key(row)
        const int r16 = lane >> 2, s4 = lane & 3;
        const uint32_t offW = (uint32_t)(n0 + pw * 64 + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);
        // tile descriptors: everything about the tile's segment comes from the kernel arguments only when the generator enters
        // a new segment (a dependent s_load per use costs ~200 clk)
        struct Tile { int seg, k0, K, toff; const float* abase; const char* wbase0; const char* wbase1; };
        int seg_l = 0, k_l = 0, gen = 0, K_l = a.seg[0].K, toff_l = pl.taboff[0];
        const float* A_l = (const float*)a.seg[0].A;
        const char* W_l = (const char*)a.seg[0].Wx;
        long ldg_l = (long)a.seg[0].ldwx * 2;                 // bytes per k-group of the weight image
        Tile last_tile;
        auto next_tile = [&](Tile& t) {                      // tiles in order; past the end: the last tile again
            if (gen >= ntile) { t = last_tile; return; }
            t.seg = seg_l; t.k0 = k_l; t.K = K_l; t.toff = toff_l;
            t.abase = A_l + k_l;
            const int g0 = k_l >> 4, glast = (K_l - 1) >> 4;  // (a K tail may have only one 16-group: the second half re-reads it,
            t.wbase0 = W_l + (long)g0 * ldg_l;                //  its activations are zero)
            t.wbase1 = W_l + (long)min(g0 + 1, glast) * ldg_l;
            last_tile = t;
            ++gen;
            k_l += TK;
            if (k_l >= K_l && seg_l + 1 < a.nseg) {
                k_l = 0; ++seg_l;
                K_l = a.seg[seg_l].K; toff_l = pl.taboff[seg_l];
                A_l = (const float*)a.seg[seg_l].A; W_l = (const char*)a.seg[seg_l].Wx; ldg_l = (long)a.seg[seg_l].ldwx * 2;
            }
        };
        u32x4 ra0[4], ra1[4];
        bool rz0[4], rz1[4];
        auto load_a = [&](const Tile& t, u32x4 (&ra)[4], bool (&rz)[4]) {
            enter_a(t.seg);
            const bool kin = t.k0 + c * 4 < t.K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gload16s(ra[i], kin ? offA[i] : offA[i] - c * 16, t.abase);     // (past the K tail: any valid address, the values are zeroed)
                rz[i] = zrow[i] || !kin;
            }
        };
        auto dma_half = [&](const char* wbase, int buf) {
            // (the instruction offset of an LDS-DMA load moves the global AND the LDS address: one M0 base serves the four pieces)
            const uint32_t sW = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + buf * W_BYTES + pw * 4 * 16 * ROWB);
            glds16<0>(offW, wbase, sW);
            glds16<1024>(offW, wbase, sW);
            glds16<2048>(offW, wbase, sW);
            glds16<3072>(offW, wbase, sW);
        };
        // write_a: BN + ReLU prologue, hi/lo split, into the tile's LDS stage [half][128 rows][64 B]; chunk c lies in half c >> 2
        const int cc = c & 3, wa_key = (rbase >> 2) & 3;       // (rows rbase + 32 i share the swizzle key)
        const int wa_hi = (c >> 2) * A_HALF + rbase * ROWB + (((cc >> 1) ^ wa_key) << 4) + (cc & 1) * 8;
        const int wa_lo = (c >> 2) * A_HALF + rbase * ROWB + (((2 + (cc >> 1)) ^ wa_key) << 4) + (cc & 1) * 8;
        auto write_a = [&](const Tile& t, int stage, const u32x4 (&ra)[4], const bool (&rz)[4]) {
            unsigned char* sA = smem + OFF_A + stage * A_BYTES;
            const bool pro = t.toff >= 0;
            float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pro) {
                const int k = t.toff + min(t.k0 + c * 4, t.K - 4);
                tsc = *(const float4*)(sSc + k);
                tsh = *(const float4*)(sSh + k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x0 = __uint_as_float(ra[i].x), x1 = __uint_as_float(ra[i].y), x2 = __uint_as_float(ra[i].z), x3 = __uint_as_float(ra[i].w);
                // zero rows / the K tail must read as zero (relu(shift) must not leak in)
                x0 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f) : x0);
                x1 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f) : x1);
                x2 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f) : x2);
                x3 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f) : x3);
                uint2 h, l;
                h.x = pack_bf16x2(x0, x1);
                h.y = pack_bf16x2(x2, x3);
                l.x = pack_bf16x2(x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xffff0000u));
                l.y = pack_bf16x2(x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xffff0000u));
                *(uint2*)(sA + wa_hi + i * 32 * ROWB) = h;
                *(uint2*)(sA + wa_lo + i * 32 * ROWB) = l;
            }
        };
        // ---- prologue.  Invariant at barrier(t): A tiles t, t+1 are in LDS (stages j % 3); weight halves up to 2t+2 have landed
        // (buffers h % 6); in flight: halves 2t+3, 2t+4 and the activations of tile t+3 (12 VMEM ops); tile t+2 is in its
        // register set (j & 1).
        Tile e0, e1, e2, e3, e4;        // tiles t+2 (to write), t+3, t+4 (to load); weight halves run one tile further
        next_tile(e0); next_tile(e1);
        load_a(e0, ra0, rz0);
        load_a(e1, ra1, rz1);
        for (int s = 0; s < a.nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
            if (pl.taboff[s] >= 0) {
                const float* sc = a.seg[s].scale;
                const float* sh = a.seg[s].shift;
                for (int k = ptid; k < a.seg[s].K; k += 256) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
            }
        }
        gload_wait_n<0>();
        __builtin_amdgcn_s_barrier();                       // P1 (producers + consumers): tables, crow complete
        write_a(e0, 0, ra0, rz0);
        write_a(e1, 1, ra1, rz1);
        dma_half(e0.wbase0, 0); dma_half(e0.wbase1, 1); dma_half(e1.wbase0, 2);
        next_tile(e2);                                      // tile 2
        load_a(e2, ra0, rz0);
        dma_half(e1.wbase1, 3); dma_half(e2.wbase0, 4);
        next_tile(e3);                                      // tile 3
        load_a(e3, ra1, rz1);
        // e2 = tile t+2, e3 = tile t+3 for t = 0; the weight halves 2t+5, 2t+6 = second half of tile t+2, first half of tile t+3
        gload_wait_n<12>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // barrier(0)
        int hbuf = 5;                                       // buffer of half 2t+5
        auto pstep = [&](int t, u32x4 (&ra)[4], bool (&rz)[4]) {
            write_a(e2, (t + 2) % 3, ra, rz);               // tile t+2: registers -> LDS (its set is then free for tile t+4)
            dma_half(e2.wbase1, hbuf);                      // halves 2t+5, 2t+6
            dma_half(e3.wbase0, hbuf == 5 ? 0 : hbuf + 1);
            hbuf = hbuf >= 4 ? hbuf - 4 : hbuf + 2;
            next_tile(e4);
            load_a(e4, ra, rz);                             // tile t+4
            e2 = e3; e3 = e4;
            gload_wait_n<12>();                             // everything but this step's 8 DMA + 4 loads has arrived
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // barrier(t+1)
        };
        for (int t = 0; t < ntile; t += 2) {
            pstep(t, ra0, rz0);
            if (t + 1 < ntile) pstep(t + 1, ra1, rz1);
        }
        gload_wait_n<0>();                                  // (re-requested tiles past the end: nothing may land in LDS later)
        __builtin_amdgcn_s_barrier();                       // E1
        if (EPI != 0) __builtin_amdgcn_s_barrier();         // E2 (statistics reduce of the consumers)
        return;
    }
    // ==================================================================================== consumer waves
    // fragment reads: row = (multiple of 32) + li, so the swizzle key is (li >> 2) & 3 for every MFMA tile; the lane's hi / lo
    // chunk offsets are two registers and the tile / stage bases are immediates
    const int fkey = (li >> 2) & 3;
    const int ohi = li * ROWB + ((lh ^ fkey) << 4), olo = li * ROWB + (((2 + lh) ^ fkey) << 4);
    union Frag { uint4 u; s16x8 s; };
    struct Frags { Frag ah[2], al[2], bh[4], bl[4]; };
    auto read_frags = [&](Frags& f, int h) {                // half h: A stage (h >> 1) % 3, half h & 1; W buffer h % 6
        const unsigned char* sA = smem + OFF_A + ((h >> 1) % 3) * A_BYTES + (h & 1) * A_HALF + wr * 64 * ROWB;
        const unsigned char* sW = smem + OFF_W + (h % 6) * W_BYTES + wc * 128 * ROWB;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            f.ah[mi].u = *(const uint4*)(sA + mi * 32 * ROWB + ohi);
            f.al[mi].u = *(const uint4*)(sA + mi * 32 * ROWB + olo);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            f.bh[ni].u = *(const uint4*)(sW + ni * 32 * ROWB + ohi);
            f.bl[ni].u = *(const uint4*)(sW + ni * 32 * ROWB + olo);
        }
    };
    auto mma = [&](const Frags& f) {                        // small terms first; consecutive MFMAs on different accumulators
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mi].s, f.bh[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi].s, f.bl[ni].s, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mi].s, f.bh[ni].s, acc[mi][ni], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                           // P1
    __builtin_amdgcn_s_barrier();                           // barrier(0)
    Frags fa, fb;
    read_frags(fa, 0);
    for (int t = 0; t < ntile; ++t) {
        // the fragments of the NEXT half are requested before the MFMAs of the current one: the matrix pipe never waits for LDS
        read_frags(fb, 2 * t + 1);
        mma(fa);
        read_frags(fa, 2 * t + 2);                          // (first half of tile t+1: complete since barrier(t))
        mma(fb);
        // every LDS read of this step has returned before the barrier releases the producers onto the buffers it read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // barrier(t+1)
    }
    __builtin_amdgcn_s_barrier();                           // E1: the producers' last transfers have landed, LDS is free

    // ---- epilogue, straight from the accumulators (lane = column li of the MFMA tile; register r = row (r&3) + 8 (r>>2) + 4 lh).
    // Branch-free: all global accesses are BUFFER loads / stores with the tensors' true extents as bounds -- an element that must
    // not be touched (row past M or unmapped by cmap, column past N) simply gets an out-of-range offset (loads return 0, stores
    // are dropped by the address unit).  Unit = 4 consecutive rows x the lane's 4 columns (one voffset per row, the columns are
    // immediate offsets of 128 bytes); the X / addend values of a unit are requested one unit AHEAD of their use.
    constexpr bool bwd = EPI >= 2, xdrop = EPI == 3;
    constexpr uint32_t OOB = 0x80000000u, RSRC3 = 0x00020000u;
    const uint32_t thresh = a.drop.thresh;
    const float inv_keep = a.drop.inv_keep;
    const uint32_t xkey = xdrop ? drop_key(a.drop, a.xsalt) : 0u;
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((rowsC - 1) * a.ldc + N) * 4), RSRC3);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd ? a.X : a.C), 0, bwd ? (int)(((rowsC - 1) * a.ldx + N) * 4) : 0, RSRC3);
    const long rowsAdd = ADD ? (long)a.B * a.addmap.T_total * a.J : 1;
    const __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc((void*)(ADD ? a.addend : a.C), 0, ADD ? (int)(((rowsAdd - 1) * a.ldadd + N) * 4) : 0, RSRC3);
    const int col0 = n0 + wc * 128 + li;               // the lane's first column; the others are + 32 ni
    bool nin[4];
    float bias[4], xs[4], xh[4], s1[4], s2[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = col0 + 32 * ni;
        nin[ni] = n < N;
        const int ncl = nin[ni] ? n : N - 1;
        bias[ni] = a.bias ? (a.bias_neg ? -a.bias[ncl] : a.bias[ncl]) : 0.f;
        xs[ni] = bwd ? a.xscale[ncl] : 0.f;
        xh[ni] = bwd ? a.xshift[ncl] : 0.f;
        s1[ni] = 0.f; s2[ni] = 0.f;
    }
    // unit u = (mi = u >> 2, q = u & 3): rows wr*64 + mi*32 + 8 q + 4 lh + {0..3} = accumulator registers 4 q .. 4 q + 3
    int crow[2][4];
    float xv[2][4][4], av[2][4][4];
    auto fetch = [&](int u, int buf) {
        const int base = wr * 64 + (u >> 2) * 32 + 8 * (u & 3) + 4 * lh;
        const int4 c4 = *(const int4*)(sCrow + base);
        crow[buf][0] = c4.x; crow[buf][1] = c4.y; crow[buf][2] = c4.z; crow[buf][3] = c4.w;
        if (bwd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t off = crow[buf][r] >= 0 ? (uint32_t)(crow[buf][r] * a.ldx + col0) * 4u : OOB;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    xv[buf][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, off + 128 * ni, 0, 0));
            }
        }
        if (ADD) {
            const int4 a4 = *(const int4*)(sAdd + base);
            const int ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t off = ar[r] >= 0 ? (uint32_t)(ar[r] * a.ldadd + col0) * 4u : OOB;     // unmapped addend row: reads 0
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    av[buf][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rAdd, off + 128 * ni, 0, 0));
            }
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int buf = u & 1, mi = u >> 2, q = u & 3;
        if (u + 1 < 8) fetch(u + 1, buf ^ 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cr = crow[buf][r];
            const uint32_t coff = (uint32_t)(cr * a.ldc + col0) * 4u;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const bool ok = cr >= 0 && nin[ni];
                float v = acc[mi][ni][4 * q + r] + bias[ni];
                if (ADD) v += av[buf][ni][r];
                if (bwd) {
                    const float x = xv[buf][ni][r];
                    v = fmaf(x, xs[ni], xh[ni]) > 0.f ? v : 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)(cr * a.ldx + col0 + 32 * ni));
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * x : 0.f;
                } else if (EPI == 1) {
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * v : 0.f;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rC, ok ? coff + 128 * ni : OOB, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the units apart: interleaving them only adds register pressure
    }
    if (EPI != 0) {
        // the two row-halves of the block (waves wr = 0 / 1) are one 128-row statistics block: combine through LDS
        float* const sRed = (float*)(smem + OFF_A);      // [wr][256][2]
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32);
            s2[ni] += __shfl_xor(s2[ni], 32);
            if (lh == 0) {
                const int cl = wc * 128 + ni * 32 + li;
                sRed[(wr * TN + cl) * 2] = s1[ni];
                sRed[(wr * TN + cl) * 2 + 1] = s2[ni];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // E2
        const int n = n0 + tid;                             // (the 256 consumer threads: one column each)
        if (n < N) {
            float* pp = a.partials + ((long)mt * N + n) * 2;
            pp[0] = sRed[tid * 2] + sRed[(TN + tid) * 2];
            pp[1] = sRed[tid * 2 + 1] + sRed[(TN + tid) * 2 + 1];
        }
    }
}

__host__ __device__ __forceinline__ int epi_variant(const gast_gemm_args& a) {      // EPI * 2 + ADD
    const int e = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    return e * 2 + (a.addend ? 1 : 0);
}

template <int EPI, bool ADD>
__global__ void __launch_bounds__(512) gemm_big_kernel(const gast_gemm_args a, const BigPlan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    big_body<EPI, ADD>(a, pl, blockIdx.x, smem);
}

struct BigBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    BigPlan pl[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int n;
};
static_assert(sizeof(BigBatch) <= 3840, "BigBatch travels as a kernel argument (4 KB limit)");
// several jobs with the SAME epilogue variant in one grid (one launch, one tail): G2 | G3 of a block, ...
template <int EPI, bool ADD>
__global__ void __launch_bounds__(512) gemm_big_multi_kernel(const BigBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    big_body<EPI, ADD>(b.a[d], b.pl[d], blockIdx.x - b.first[d], smem);
}

// ---- pre-split weight image, k-group-major: img[(k>>4) * ldimg + r * 32 + (k&15)] = bf16 hi(W[r][k]),  + 16: bf16 lo;
// zero for K <= k < Kp (rows past R are never written: the caller provides them zero-filled)
struct ImageBatch { gast_x3_image_job j[GAST_X3_IMAGE_MAX_BATCH]; int first[GAST_X3_IMAGE_MAX_BATCH + 1]; int n; };
static_assert(sizeof(ImageBatch) <= 3840, "ImageBatch travels as a kernel argument");
__global__ void __launch_bounds__(256) x3_image_kernel(const ImageBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const gast_x3_image_job& j = b.j[d];
    const int Kp4 = (j.K + 15) / 16 * 4;                          // 4-value chunks per padded row
    const long idx = (long)(blockIdx.x - b.first[d]) * 256 + threadIdx.x;
    if (idx >= (long)j.R * Kp4) return;
    const int r = (int)(idx / Kp4), k = (int)(idx - (long)r * Kp4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < j.K) v = *(const float4*)(j.W + (long)r * j.ldw + k);
    uint2 h, l;
    h.x = pack_bf16x2(v.x, v.y);
    h.y = pack_bf16x2(v.z, v.w);
    l.x = pack_bf16x2(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf16x2(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
    bf16_t* o = (bf16_t*)j.img + (long)(k >> 4) * j.ldimg + (long)r * 32 + (k & 15);
    *(uint2*)o = h;
    *(uint2*)(o + 16) = l;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool big_setup_done[64] = {};

}  // namespace

// Can this GEMM run on the large-M kernel?  Fills the plan when it can.  (Called by gast_gemm_ws / gast_gemm_multi in gemm.hip.)
int gast_gemm_big_plan(const gast_gemm_args& a, BigPlan& pl) {
    static const int enabled = getenv("GAST_GEMM_BIG") ? atoi(getenv("GAST_GEMM_BIG")) : 1;
    static const int min_rows = getenv("GAST_GEMM_BIG_MIN_M") ? atoi(getenv("GAST_GEMM_BIG_MIN_M")) : 8192;
    if (!enabled || a.dtype != GAST_F32X3) return 0;
    const long Ml = (long)a.B * a.Tn * a.J;
    static const int all_shapes = getenv("GAST_GEMM_BIG_ALL") ? atoi(getenv("GAST_GEMM_BIG_ALL")) : 0;
    if (Ml < min_rows || Ml > 0x7fffff00L || a.N < 32) return 0;
    if (!all_shapes) {
        // measured on MI355X (scripts/gemm_table.py bf16x3, B = 128): the 128 x 256 tile loses to gemm.hip's 128 x 128 tile when
        // half of it is empty (N <= 128), and its BNRELU_BWD epilogue (X / addend gathered per lane) only pays on long K loops
        int ksum = 0;
        for (int s = 0; s < a.nseg; ++s) ksum += a.seg[s].K;
        if (a.N < 256) return 0;
        if (a.epi == GAST_EPI_BNRELU_BWD && ksum < 768) return 0;
    }
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG) return 0;
    int ntab = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.Wx || !aligned16(g.Wx) || g.ldwx % 8 || !g.A || !aligned16(g.A) || g.lda % 4 || g.K % 4 || g.K < 4) return 0;
        if ((long)a.B * g.map.T_total * a.J * g.lda * 4 >= 0xffffffffL) return 0;      // 32-bit byte offsets into the activation tensor
        if (g.pro == GAST_PRO_BNRELU_DROP) return 0;
        pl.taboff[s] = -1;
        if (g.pro == GAST_PRO_BNRELU) {
            if (!g.scale || !g.shift) return 0;
            for (int q = 0; q < s; ++q)
                if (pl.taboff[q] >= 0 && a.seg[q].scale == g.scale && a.seg[q].shift == g.shift && a.seg[q].K == g.K) pl.taboff[s] = pl.taboff[q];
            if (pl.taboff[s] < 0) { pl.taboff[s] = ntab; ntab += (g.K + 3) / 4 * 4; }
        }
    }
    if (ntab > MAX_TAB) return 0;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return 0;
    // the epilogue addresses C / X / addend with 32-bit byte offsets inside buffer descriptors
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    if (rowsC * a.ldc * 4 >= 0x7fffffffL || (a.epi == GAST_EPI_BNRELU_BWD && rowsC * a.ldx * 4 >= 0x7fffffffL)) return 0;
    if (a.addend && (long)a.B * a.addmap.T_total * a.J * a.ldadd * 4 >= 0x7fffffffL) return 0;
    static const int ablate = getenv("GAST_GEMM_BIG_ABLATE") ? atoi(getenv("GAST_GEMM_BIG_ABLATE")) : 0;
    pl.ablate = ablate;
    pl.M = (int)Ml;
    pl.tilesM = (pl.M + TM - 1) / TM;
    pl.tilesN = (a.N + TN - 1) / TN;
    pl.ntab = ntab;
    return 1;
}

static int big_lds_bytes(int ntab) { return OFF_TAB + 4 * 256 * 4 + 2 * ntab * 4; }

typedef void (*big_kernel_t)(const gast_gemm_args, const BigPlan);
static big_kernel_t big_kernel(int v) {
    switch (v) {
        case 0: return gemm_big_kernel<0, false>;
        case 1: return gemm_big_kernel<0, true>;
        case 2: return gemm_big_kernel<1, false>;
        case 3: return gemm_big_kernel<1, true>;
        case 4: return gemm_big_kernel<2, false>;
        case 5: return gemm_big_kernel<2, true>;
        case 6: return gemm_big_kernel<3, false>;
        default: return gemm_big_kernel<3, true>;
    }
}

typedef void (*big_multi_kernel_t)(const BigBatch);
static big_multi_kernel_t big_multi_kernel(int v) {
    switch (v) {
        case 0: return gemm_big_multi_kernel<0, false>;
        case 1: return gemm_big_multi_kernel<0, true>;
        case 2: return gemm_big_multi_kernel<1, false>;
        case 3: return gemm_big_multi_kernel<1, true>;
        case 4: return gemm_big_multi_kernel<2, false>;
        case 5: return gemm_big_multi_kernel<2, true>;
        case 6: return gemm_big_multi_kernel<3, false>;
        default: return gemm_big_multi_kernel<3, true>;
    }
}

static void big_setup() {
    int dev = 0;
    hipGetDevice(&dev);               // function attributes are per device (nn.DataParallel replicas launch on several)
    dev &= 63;
    if (big_setup_done[dev]) return;
    for (int v = 0; v < 8; ++v) hipFuncSetAttribute((const void*)big_kernel(v), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BLOCK);
    for (int v = 0; v < 8; ++v) hipFuncSetAttribute((const void*)big_multi_kernel(v), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BLOCK);
    big_setup_done[dev] = true;
    if (getenv("GAST_GEMM_BIG_DEBUG")) {
        for (int v = 0; v < 8; v += 2) {
            int nb = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)big_kernel(v), 512, big_lds_bytes(0));
            hipFuncAttributes fa;
            hipFuncGetAttributes(&fa, (const void*)big_kernel(v));
            fprintf(stderr, "gemm_big variant %d: %d blocks/CU at %d B LDS, %d regs, %zu B scratch\n", v, nb, big_lds_bytes(0), fa.numRegs, (size_t)fa.localSizeBytes);
        }
    }
}

int gast_gemm_big_launch(const gast_gemm_args& a, const BigPlan& pl, hipStream_t st) {
    big_setup();
    hipLaunchKernelGGL(big_kernel(epi_variant(a)), dim3(pl.tilesM * pl.tilesN), dim3(512), big_lds_bytes(pl.ntab), st, a, pl);
    GAST_CHECK_LAUNCH();
    return 0;
}

int gast_gemm_big_launch_multi(const gast_gemm_args* args, const BigPlan* pls, int n, hipStream_t st) {
    big_setup();
    bool done[GAST_GEMM_MAX_BATCH] = {};
    for (int d0 = 0; d0 < n; ++d0) {          // one grid per epilogue variant present in the batch
        if (done[d0]) continue;
        const int v = epi_variant(args[d0]);
        BigBatch b;
        b.n = 0;
        b.first[0] = 0;
        int ntab = 0;
        for (int d = d0; d < n; ++d) {
            if (done[d] || epi_variant(args[d]) != v) continue;
            done[d] = true;
            const int k = b.n++;
            b.a[k] = args[d];
            b.pl[k] = pls[d];
            b.first[k + 1] = b.first[k] + pls[d].tilesM * pls[d].tilesN;
            if (pls[d].ntab > ntab) ntab = pls[d].ntab;
        }
        if (b.n == 1) hipLaunchKernelGGL(big_kernel(v), dim3(b.first[1]), dim3(512), big_lds_bytes(ntab), st, b.a[0], b.pl[0]);
        else hipLaunchKernelGGL(big_multi_kernel(v), dim3(b.first[b.n]), dim3(512), big_lds_bytes(ntab), st, b);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" long gast_x3_image_ld(int R) { return (long)((R + 15) / 16 * 16 + 256) * 32; }

extern "C" int gast_x3_image_multi(const gast_x3_image_job* jobs, int n, gast_stream_t stream) {
    if (!jobs || n < 0) return GAST_EINVAL;
    for (int i0 = 0; i0 < n; i0 += GAST_X3_IMAGE_MAX_BATCH) {
        ImageBatch b;
        b.n = n - i0 < GAST_X3_IMAGE_MAX_BATCH ? n - i0 : GAST_X3_IMAGE_MAX_BATCH;
        b.first[0] = 0;
        for (int d = 0; d < b.n; ++d) {
            const gast_x3_image_job& j = jobs[i0 + d];
            if (!j.W || !j.img || j.R < 1 || j.K < 4) return GAST_EINVAL;
            if (j.K % 4 || j.ldw % 4 || !aligned16(j.W) || !aligned16(j.img) || j.ldimg % 8 || j.ldimg < (long)j.R * 32) return GAST_EALIGN;
            b.j[d] = j;
            const long chunks = (long)j.R * ((j.K + 15) / 16 * 4);
            b.first[d + 1] = b.first[d] + (int)((chunks + 255) / 256);
        }
        hipLaunchKernelGGL(x3_image_kernel, dim3(b.first[b.n]), dim3(256), 0, (hipStream_t)stream, b);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}
