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
#include <atomic>

namespace {

// NI = 32-column MFMA tiles per wave: 4 (block tile 128 x 256) or 2 (128 x 128: the N <= 128 GEMMs, for which half of the wide
// tile would be empty, and the short-K input gradients whose epilogue then holds half the registers)
// MW = 64-row wave rows per block: 2 (128-row block tile, 256 threads, two or three blocks per CU) or 4 (256-row block tile, 512
// threads = 8 waves, ONE block per CU: the 256 x 64 B weight tile of a K step then serves 256 rows -- 2/3 of the operand bytes per
// FLOP of the 128 x 256 tile)
constexpr int TK = 16;
constexpr int ROWB = 64;                        // LDS row image: 16 bf16 hi | 16 bf16 lo
constexpr int OFF_ROWS = 0;                     // crow[TM] | addrow[TM]
constexpr int tm_of(int mw) { return 64 * mw; }
constexpr int nt_of(int mw) { return 128 * mw; }                       // threads per block
constexpr int a_bytes(int mw) { return tm_of(mw) * ROWB; }             // 8 KB / 16 KB
constexpr int off_a(int mw) { return 2 * tm_of(mw) * 4; }              // two activation stages
constexpr int off_w(int mw) { return off_a(mw) + 2 * a_bytes(mw); }    // three weight stages
constexpr int lds_block(int mw) { return mw == 2 ? 80 * 1024 : 160 * 1024; }
constexpr int tn_of(int ni) { return 64 * ni; }
constexpr int w_bytes(int ni) { return tn_of(ni) * ROWB; }             // 16 KB / 8 KB
constexpr int off_tab(int ni, int mw, int d = 2) { return off_w(mw) + (d + 1) * w_bytes(ni); }      // scale | shift tables behind the d + 1 weight stages
constexpr int max_tab(int ni, int mw, int d = 2) { return (lds_block(mw) - off_tab(ni, mw, d) - 6 * nt_of(mw) * 4) / 8; }

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
// PAIR: 1 = bf16 hi/lo pairs (GAST_F32X3), 2 = fp16 pairs (GAST_F32X3H: the forward GEMMs; weight images of the f16 kind)
// D = prefetch distance in K steps (both operands): 2 (the large-M launches: two or three resident blocks per CU cover each other) or 4
// (NI = 2 only, the M = B*J stage: at most a block or two per CU, so a block has to cover the memory latency by itself -- five weight
// stages, four activation register sets, 57 KB of LDS)
template <int EPI, bool ADD, int NI, int MW, int PAIR, int D = 2>
__device__ __forceinline__ void big_body(const gast_gemm_args& a, const BigPlan& pl, int blk, unsigned char* smem) {
    constexpr int TM = tm_of(MW), NT = nt_of(MW), A_BYTES = a_bytes(MW), OFF_A = off_a(MW), OFF_W = off_w(MW);
    constexpr int TN = tn_of(NI), W_BYTES = w_bytes(NI), OFF_TAB = off_tab(NI, MW, D);
    // PAIR = 3: 16-bit STORAGE (GAST_BF16: bfloat16, or binary16 in the -DGAST_H16_F16 build), ONE product.  A K step then covers 32
    // values: a row image is [32 x 16 bit] = the same 64 bytes, four 16-byte chunks of 8 values; the chunk a lane reads for the first
    // 16-deep MFMA is the one that holds the hi halves in the split layout (chunk lh), for the second the lo one (chunk 2 + lh) -- the
    // weight DMA, the fragment reads and the LDS swizzle are the split kernel's, the products per K step drop from 3 x 8 to 2 x 8
    // per column half, and the activation path converts (prologue only) instead of splitting.
    constexpr bool H16 = PAIR == 3;
    constexpr int ESZ = H16 ? 2 : 4;                // bytes per stored element
    constexpr int CV = 16 / ESZ;                    // values per 16-byte chunk: 4 / 8
    constexpr int TKV = H16 ? 32 : TK;              // K values per step
    constexpr int WS = D + 1;                       // weight stages in the ring
    constexpr int WROWS = TN / (2 * MW);            // weight rows a wave's DMA fills per K step
    constexpr int WPIECES = WROWS / 16;             // ... in 16-row (1 KB) pieces: 4, 2 or 1
    static_assert(WPIECES >= 1, "tile too narrow for the wave count");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int M = pl.M, N = a.N;
    const int lb = xcd_remap(blk, pl.tilesM * pl.tilesN);
    const int mt = lb / pl.tilesN, nt = lb - mt * pl.tilesN;
    const int m0 = mt * TM, n0 = nt * TN;

    int* const sCrow = (int*)(smem + OFF_ROWS);
    int* const sAdd = sCrow + TM;
    float* const sSc = (float*)(smem + OFF_TAB);
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

    // ---- staging duties.  Activations (registers): rows rbase + TM/2 i (i < 2), 16-byte chunk c (4 fp32 values) of the K step.
    // Per-row facts that are only needed when a stream enters a new K segment (the (b, t, j) position of the thread's rows, the
    // weight row / chunk of its DMA pieces) are recomputed or re-read from LDS there instead of living in registers: the K loop
    // runs at 250 of 256 VGPRs and a spill inside it costs an s_waitcnt vmcnt(0), i.e. the whole prefetch.
    const int c = tid & 3, rbase = tid >> 2;
    int* const sPos = (int*)(smem + OFF_TAB) + 2 * pl.ntab;      // [2][3][NT]: b, t, j of this thread's two rows (-1: row past M)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + rbase + (TM / 2) * i;
        int b = -1, t = 0, j = 0;
        if (m < M) { b = m / TJ; const int rem = m - b * TJ; t = rem / a.J; j = rem - t * a.J; }
        sPos[(i * 3 + 0) * NT + tid] = b;
        sPos[(i * 3 + 1) * NT + tid] = t;
        sPos[(i * 3 + 2) * NT + tid] = j;
    }
    const int r16 = lane >> 2, s4 = lane & 3;

    // Activation stream: per-thread byte offsets of its two rows inside the segment's tensor (they change with the segment's
    // row map), added to a scalar base that advances with k
    uint32_t offA[2];
    bool zrow[2];
    int seg_a = -1;
    auto enter_a = [&](int s) {
        if (s == seg_a) return;
        seg_a = s;
        const gast_gemm_seg& sg = a.seg[s];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = sPos[(i * 3 + 0) * NT + tid], t = sPos[(i * 3 + 1) * NT + tid], j = sPos[(i * 3 + 2) * NT + tid];
            const int ts = t * sg.map.t_stride + sg.map.t_off;
            const bool ok = b >= 0 && ts >= 0 && ts < sg.map.T_total;
            const uint32_t srow = ok ? (uint32_t)((b * sg.map.T_total + ts) * a.J + j) : 0u;
            zrow[i] = !ok;                                   // out-of-range tap (or a row past M): reads as zero
            offA[i] = (srow * (uint32_t)sg.lda + c * CV) * (uint32_t)ESZ;
        }
    };
    // Weight stream (DMA): wave w fills the 16-row pieces (w*4 + i) of the 256 x 64 B tile, which is ONE contiguous 16 KB block
    // of the k-group-major image; lane = (row r16, slot s4) and slot s4 receives source chunk s4 ^ key(row).  Per thread: one
    // byte offset (the pieces are immediates), per step: a scalar base.
    const uint32_t offW = (uint32_t)(n0 + w * WROWS + r16) * 64u + (uint32_t)((s4 ^ ((r16 >> 2) & 3)) << 4);
    // A tile descriptor carries everything the K loop needs to know about its segment (K, table offset, operand bases), fetched
    // from the kernel arguments only when the generator enters a new segment: a dependent s_load per use costs ~200 clk.
    struct Tile { int seg, k0, K, toff; const char* abase; const char* wbase; };
    int ntile = 0;
    for (int s = 0; s < a.nseg; ++s) ntile += (a.seg[s].K + TKV - 1) / TKV;
    int seg_l = 0, k_l = 0, gen = 0, K_l = a.seg[0].K, toff_l = pl.taboff[0];
    const char* A_l = (const char*)a.seg[0].A;
    const char* W_l = (const char*)a.seg[0].Wx;
    long ldg_l = (long)a.seg[0].ldwx * 2;                     // bytes per k-group of the weight image
    Tile last_tile = {0, 0, K_l, toff_l, A_l, W_l};
    auto next_tile = [&](Tile& t) {                          // tiles in order; past the end: the last tile again
        if (gen >= ntile) { t = last_tile; return; }
        t.seg = seg_l; t.k0 = k_l; t.K = K_l; t.toff = toff_l;
        t.abase = A_l + k_l * ESZ;
        t.wbase = W_l + (long)(k_l / TKV) * ldg_l;
        last_tile = t;
        ++gen;
        k_l += TKV;
        if (k_l >= K_l && seg_l + 1 < a.nseg) {
            k_l = 0; ++seg_l;
            K_l = a.seg[seg_l].K; toff_l = pl.taboff[seg_l];
            A_l = (const char*)a.seg[seg_l].A; W_l = (const char*)a.seg[seg_l].Wx; ldg_l = (long)a.seg[seg_l].ldwx * 2;
        }
    };

    // NA = D register sets of activation tiles.  (Round 4, large-M launches: a THIRD set with the weights still at distance 2 changed
    // nothing -- 3.759 / 3.784 vs 3.750 / 3.774 ms -- the activation stream's latency is not what those launches wait for.)
    constexpr int NA = D;
    static_assert(D == 2 || D == 4, "prefetch distance 2 or 4");
    u32x4 ra[NA][2];
    bool rz[NA][2];
#ifdef GAST_GEMM_BIG_ABLATION
    const int abl = pl.ablate;          // profiling build only (build.sh ABLATION=1): runtime switches split the K loop's basic blocks
#else
    constexpr int abl = 0;
#endif
    auto load_a = [&](const Tile& t, u32x4 (&ra)[2], bool (&rz)[2]) {
        enter_a(t.seg);
        if (abl & 8) return;
        const bool kin = t.k0 + c * CV < t.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gload16s(ra[i], kin ? offA[i] : offA[i] - c * 16, t.abase);     // (past the K tail: any valid address, the values are zeroed)
            rz[i] = zrow[i] || !kin;
        }
    };
    auto dma_w = [&](const Tile& t, int stage) {
        if (abl & 4) return;
        const uint32_t sW = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + stage * W_BYTES + w * WROWS * ROWB);
        // (the instruction offset of an LDS-DMA load moves the global AND the LDS address: one M0 base serves the four pieces)
        glds16<0>(offW, t.wbase, sW);
        if (WPIECES >= 2) glds16<1024>(offW, t.wbase, sW);
        if (WPIECES == 4) {
            glds16<2048>(offW, t.wbase, sW);
            glds16<3072>(offW, t.wbase, sW);
        }
    };
    // scale / shift of a tile's prologue (this thread's 4 K values), fetched from the LDS tables one step before write_a uses them
    float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tsc2 = tsc, tsh2 = tsh;                   // (PAIR = 3: values 4 .. 7 of the thread's chunk)
    auto fetch_tab = [&](const Tile& t) {
        if (t.toff >= 0) {
            const int k = t.toff + min(t.k0 + c * CV, t.K - CV);
            tsc = *(const float4*)(sSc + k);
            tsh = *(const float4*)(sSh + k);
            if constexpr (H16) {
                tsc2 = *(const float4*)(sSc + k + 4);
                tsh2 = *(const float4*)(sSh + k + 4);
            }
        }
    };
    const int wa_key = (rbase >> 2) & 3;             // (rows rbase and rbase + TM/2 share the swizzle key)
    const int wa_hi = rbase * ROWB + (((c >> 1) ^ wa_key) << 4) + (c & 1) * 8, wa_lo = rbase * ROWB + (((2 + (c >> 1)) ^ wa_key) << 4) + (c & 1) * 8;
    const int wa_h16 = rbase * ROWB + ((c ^ wa_key) << 4);      // PAIR = 3: chunk c of the row, whole
    auto write_a = [&](const Tile& t, int stage, const u32x4 (&ra)[2], const bool (&rz)[2]) {
        if (abl & 16) return;
        unsigned char* sA = smem + OFF_A + stage * A_BYTES;
        const bool pro = t.toff >= 0;
        if constexpr (H16) {
            // 8 stored values per row: pass through, or BN + ReLU on the unpacked values and round back to the storage type
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t w4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                const float sc8[8] = {tsc.x, tsc.y, tsc.z, tsc.w, tsc2.x, tsc2.y, tsc2.z, tsc2.w};
                const float sh8[8] = {tsh.x, tsh.y, tsh.z, tsh.w, tsh2.x, tsh2.y, tsh2.z, tsh2.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float lo, hi;
                    h16x2_unpack(w4[p], lo, hi);
                    const uint32_t pw = pack_h16x2(fmaxf(fmaf(lo, sc8[2 * p], sh8[2 * p]), 0.f), fmaxf(fmaf(hi, sc8[2 * p + 1], sh8[2 * p + 1]), 0.f));
                    w4[p] = rz[i] ? 0u : (pro ? pw : w4[p]);
                }
                *(uint4*)(sA + wa_h16 + i * (TM / 2) * ROWB) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float x0 = __uint_as_float(ra[i].x), x1 = __uint_as_float(ra[i].y), x2 = __uint_as_float(ra[i].z), x3 = __uint_as_float(ra[i].w);
            // BN + ReLU prologue; zero rows / the K tail must read as zero (relu(shift) must not leak in)
            x0 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x0, tsc.x, tsh.x), 0.f) : x0);
            x1 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x1, tsc.y, tsh.y), 0.f) : x1);
            x2 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x2, tsc.z, tsh.z), 0.f) : x2);
            x3 = rz[i] ? 0.f : (pro ? fmaxf(fmaf(x3, tsc.w, tsh.w), 0.f) : x3);
            uint2 h, l;
            split_pair4<PAIR>(x0, x1, x2, x3, h, l);
            *(uint2*)(sA + wa_hi + i * (TM / 2) * ROWB) = h;
            *(uint2*)(sA + wa_lo + i * (TM / 2) * ROWB) = l;
        }
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    // fragment reads: row = (multiple of 32) + li, so the swizzle key is (li >> 2) & 3 for every MFMA tile; the lane's hi / lo
    // chunk offsets are two registers and the tile / stage bases are immediates
    const int fkey = (li >> 2) & 3;
    const int ohi = li * ROWB + ((lh ^ fkey) << 4), olo = li * ROWB + (((2 + lh) ^ fkey) << 4);

    // ---- pipeline.  Tile j's activations travel in register set j % D and LDS stage j & 1, its weights in stage j % (D + 1).
    // Invariant at the top of iteration t (after the counted wait + barrier): LDS holds tile t; set (t+1) % D holds tile t+1's
    // activations; in flight: the weights of tiles t+1 .. t+D-1 and the activations of tiles t+2 .. t+D.
    // dq[i] = tile t+1+i at the top of step t: dq[0] is written to LDS, dq[D-1]'s weights are requested, dq[D]'s activations are loaded
    Tile dq[NA + 1];
    {
        Tile d0;
        next_tile(d0);
        dma_w(d0, 0);
        load_a(d0, ra[0], rz[0]);
#pragma unroll
        for (int i = 1; i < NA; ++i) {             // tiles 1 .. NA-1 into sets 1 .. NA-1
            next_tile(dq[i - 1]);
            dma_w(dq[i - 1], i);
            load_a(dq[i - 1], ra[i], rz[i]);
        }
        for (int s = 0; s < a.nseg; ++s) {                 // scale / shift tables (while the first tiles are in flight)
            if (pl.taboff[s] >= 0) {
                const float* sc = a.seg[s].scale;
                const float* sh = a.seg[s].shift;
                for (int k = tid; k < a.seg[s].K; k += NT) { sSc[pl.taboff[s] + k] = sc[k]; sSh[pl.taboff[s] + k] = sh[k]; }
            }
        }
        gload_wait_n<0>();
        __syncthreads();                                   // tables complete
        // The loads are inline asm the compiler does not track and the wait above does not name their registers: nothing but this
        // statement keeps the conversion's VALU instructions -- pure register code -- BEHIND the wait (volatile asm statements keep their
        // order, and the values write_a reads now come out of this one).  Round 5: a build of this kernel WITHOUT the prologue's fma / max
        // had nothing else anchoring them -- hipcc hoisted the whole conversion above the wait, and the GEMM multiplied whatever the
        // registers held (scripts/asm_load_hazard.py --strict reports it).  With the prologue the scale / shift reads happen to anchor the
        // code; the pin makes that a property of the source.
        asm volatile("" : "+v"(ra[0][0]), "+v"(ra[0][1]));
        fetch_tab(d0);
        write_a(d0, 0, ra[0], rz[0]);
        next_tile(dq[NA - 1]);                     // tile NA into the freed set 0
        load_a(dq[NA - 1], ra[0], rz[0]);
        next_tile(dq[NA]);
        fetch_tab(dq[0]);
    }
    // One K step.  Order after the barrier: the fragment reads of tile t go out first (their LDS latency is covered by the VALU
    // work of write_a), then tile t+1's activations are written, the transfers of tiles t+2 / t+3 requested, and the MFMAs of
    // tile t issued; the second half of the weight fragments is read under the first half's MFMAs.
    struct Frag { uint4 u; };
    auto mma3 = [&](int nh, const Frag (&ah)[2], const Frag (&al)[2], const Frag (&bh)[2], const Frag (&bl)[2]) {
        if (abl & 1) {      // keep the fragment reads alive without the matrix cores
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[0][nh * 2 + q][0] += __uint_as_float(bh[q].u.x ^ bl[q].u.y ^ ah[q].u.z ^ al[q].u.w);
            return;
        }
        if constexpr (H16) {      // two 16-deep products of the 32-value step: k 0..15 (chunks 0 / 1), k 16..31 (chunks 2 / 3)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[mi][nh * 2 + q] = mfma_h16(ah[mi].u, bh[q].u, acc[mi][nh * 2 + q]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[mi][nh * 2 + q] = mfma_h16(al[mi].u, bl[q].u, acc[mi][nh * 2 + q]);
            return;
        }
        // small terms first; consecutive MFMAs on different accumulators
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[mi][nh * 2 + q] = mfma_pair<PAIR>(al[mi].u, bh[q].u, acc[mi][nh * 2 + q]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[mi][nh * 2 + q] = mfma_pair<PAIR>(ah[mi].u, bl[q].u, acc[mi][nh * 2 + q]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[mi][nh * 2 + q] = mfma_pair<PAIR>(ah[mi].u, bh[q].u, acc[mi][nh * 2 + q]);
    };
    int wstage = 0;                    // t % (D + 1): the weight stage of the tile the step multiplies
    auto step = [&](int t, bool wr_next, bool do_mma, u32x4 (&ra)[2], bool (&rz)[2]) {
        // (the DMA pieces + 2 activation loads of the newest D - 1 steps stay in flight: everything step t - D requested -- the weights
        //  of tile t, the activations of tile t + 1 -- is older and therefore complete)
        gload_wait_n<(D - 1) * (WPIECES + 2)>();
        __syncthreads();
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]));        // (the set this step converts is complete NOW: see the pipeline fill above)
        const unsigned char* sA = smem + OFF_A + (t & 1) * A_BYTES + wr * 64 * ROWB;
        const unsigned char* sW = smem + OFF_W + wstage * W_BYTES + wc * (TN / 2) * ROWB;
        Frag ah[2], al[2], bh[2], bl[2];
        if (!(abl & 2)) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                ah[mi].u = *(const uint4*)(sA + mi * 32 * ROWB + ohi);
                al[mi].u = *(const uint4*)(sA + mi * 32 * ROWB + olo);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bh[q].u = *(const uint4*)(sW + q * 32 * ROWB + ohi);
                bl[q].u = *(const uint4*)(sW + q * 32 * ROWB + olo);
            }
        }
        if (wr_next) write_a(dq[0], (t + 1) & 1, ra, rz);       // tile t+1: registers -> LDS (its set is then free for tile t+1+NA)
        if (!(abl & 2)) {
            if (do_mma) mma3(0, ah, al, bh, bl);
            if (NI == 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {                   // second column half: read under the first half's MFMAs
                    bh[q].u = *(const uint4*)(sW + (2 + q) * 32 * ROWB + ohi);
                    bl[q].u = *(const uint4*)(sW + (2 + q) * 32 * ROWB + olo);
                }
                if (do_mma) mma3(1, ah, al, bh, bl);
            }
        }
        // the transfers of tiles t+2 / t+3 are requested AFTER the step's MFMAs have been issued: when the memory system pushes
        // back, a wave stalls at the ISSUE of a VMEM instruction, and everything behind it in program order waits with it
        dma_w(dq[D - 1], wstage == 0 ? WS - 1 : wstage - 1);     // tile t + D -> stage (t + D) % (D + 1) = the stage before tile t's
        load_a(dq[NA], ra, rz);
        wstage = wstage == WS - 1 ? 0 : wstage + 1;
#pragma unroll
        for (int i = 0; i < NA; ++i) dq[i] = dq[i + 1];
        next_tile(dq[NA]);
        fetch_tab(dq[0]);                                       // (for the next step's write_a)
    };
    for (int t = 0; t < ntile; t += NA) {
        // step t + u: tile t+u+1 lives in set (u + 1) % NA.  Past the last tile the remaining steps of the trip still run their
        // (re-requested, unused) transfers so that the loop has ONE exit and the counted waits stay exact.
        step(t, t + 1 < ntile, true, ra[1 % NA], rz[1 % NA]);
        step(t + 1, t + 2 < ntile, t + 1 < ntile, ra[2 % NA], rz[2 % NA]);
        if constexpr (NA == 4) {
            step(t + 2, t + 3 < ntile, t + 2 < ntile, ra[3], rz[3]);
            step(t + 3, t + 4 < ntile, t + 3 < ntile, ra[0], rz[0]);
        }
    }
    gload_wait_n<0>();                 // (the re-requested tiles past the end: nothing may land in LDS after this point)
    // ... nor in REGISTERS: the activation loads of the tiles past the end are never consumed, so the compiler considers their
    // destination registers dead from the moment the load is issued and may park an epilogue value there -- computed ABOVE the
    // wait (only memory operations keep their place relative to an asm statement) and overwritten when the load lands.  Seen
    // with four sets in flight (round 4: a column's scale pointer, i.e. a memory fault).  An empty asm that reads every set
    // after the wait keeps them allocated until here.
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        asm volatile("" ::"v"(ra[i][0]), "v"(ra[i][1]));
    }
    __syncthreads();

    if (abl & 32) { if (acc[0][0][0] == 12345.678f) ((float*)a.C)[0] = acc[1][NI - 1][5] + acc[0][1][2] + acc[1][0][1] + acc[1][1][1]; return; }
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
    // (PAIR = 3: C / C2 / X / addend are tensors of the 16-bit storage type -- two-byte buffer accesses, values rounded to the storage
    //  type BEFORE they enter the column sums, as in gemm.hip: the statistics are those of the tensor that is stored)
    constexpr uint32_t ESO = H16 ? 2u : 4u, NISTEP = 32u * ESO;
    auto ldv = [&](const __amdgpu_buffer_rsrc_t& r, uint32_t off) -> float {
        if constexpr (H16) return bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    };
    auto stv = [&](float v, const __amdgpu_buffer_rsrc_t& r, uint32_t off) {
        if constexpr (H16) __builtin_amdgcn_raw_buffer_store_b16(f2bf(v), r, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, off, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((rowsC - 1) * a.ldc + N) * ESO), RSRC3);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd ? a.X : a.C), 0, bwd ? (int)(((rowsC - 1) * a.ldx + N) * ESO) : 0, RSRC3);
    // second output of the BNRELU_BWD epilogues: the value before the mask (a null C2 gets a zero-sized descriptor: its stores are dropped).
    // Only in the variants WITHOUT an addend (what the plan needs: gast_gemm_big_plan): with one the extra descriptor and store push the
    // narrow-tile kernel from 167 to 173 registers, i.e. from three to two blocks per CU.
    constexpr bool C2OK = bwd && !ADD;
    const bool has2 = C2OK && a.C2 != nullptr;
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(has2 ? a.C2 : a.C, 0, has2 ? (int)(((rowsC - 1) * a.ldc2 + N) * ESO) : 0, RSRC3);
    const long rowsAdd = ADD ? (long)a.B * a.addmap.T_total * a.J : 1;
    const __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc((void*)(ADD ? a.addend : a.C), 0, ADD ? (int)(((rowsAdd - 1) * a.ldadd + N) * ESO) : 0, RSRC3);
    const int col0 = n0 + wc * (TN / 2) + li;          // the lane's first column; the others are + 32 ni
    bool nin[NI];
    float bias[NI], xs[NI], xh[NI], s1[NI], s2[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
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
    float xv[2][NI][4], av[2][NI][4];
    auto fetch = [&](int u, int buf) {
        const int base = wr * 64 + (u >> 2) * 32 + 8 * (u & 3) + 4 * lh;
        const int4 c4 = *(const int4*)(sCrow + base);
        crow[buf][0] = c4.x; crow[buf][1] = c4.y; crow[buf][2] = c4.z; crow[buf][3] = c4.w;
        if (bwd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t off = crow[buf][r] >= 0 ? (uint32_t)(crow[buf][r] * a.ldx + col0) * ESO : OOB;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) xv[buf][ni][r] = ldv(rX, off + NISTEP * ni);
            }
        }
        if (ADD) {
            const int4 a4 = *(const int4*)(sAdd + base);
            const int ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t off = ar[r] >= 0 ? (uint32_t)(ar[r] * a.ldadd + col0) * ESO : OOB;     // unmapped addend row: reads 0
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) av[buf][ni][r] = ldv(rAdd, off + NISTEP * ni);
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
            const uint32_t coff = (uint32_t)(cr * a.ldc + col0) * ESO;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const bool ok = cr >= 0 && nin[ni];
                float v = acc[mi][ni][4 * q + r] + bias[ni];
                if (ADD) v += av[buf][ni][r];
                if (bwd) {
                    if constexpr (C2OK) stv(v, rC2, ok ? (uint32_t)(cr * a.ldc2 + col0) * ESO + NISTEP * ni : OOB);
                    const float x = xv[buf][ni][r];
                    v = fmaf(x, xs[ni], xh[ni]) > 0.f ? v : 0.f;
                    if (xdrop) v *= drop_mul(xkey, thresh, inv_keep, (uint32_t)(cr * a.ldx + col0 + 32 * ni));
                    if constexpr (H16) v = bf2f(f2bf(v));
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * x : 0.f;
                } else if (EPI == 1) {
                    if constexpr (H16) v = bf2f(f2bf(v));
                    s1[ni] += ok ? v : 0.f;
                    s2[ni] += ok ? v * v : 0.f;
                }
                stv(v, rC, ok ? coff + NISTEP * ni : OOB);
            }
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the units apart: interleaving them only adds register pressure
    }
    if (EPI != 0) {
        // the two row-halves of the block (waves wr = 0 / 1) are one 128-row statistics block: combine through LDS
        float* const sRed = (float*)(smem + OFF_A);      // [wr][TN][2]
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32);
            s2[ni] += __shfl_xor(s2[ni], 32);
            if (lh == 0) {
                const int cl = wc * (TN / 2) + ni * 32 + li;
                sRed[(wr * TN + cl) * 2] = s1[ni];
                sRed[(wr * TN + cl) * 2 + 1] = s2[ni];
            }
        }
        __syncthreads();
        if constexpr (MW == 2) {
            const int n = n0 + tid;
            if (tid < TN && n < N) {
                float* pp = a.partials + ((long)mt * N + n) * 2;
                pp[0] = sRed[tid * 2] + sRed[(TN + tid) * 2];
                pp[1] = sRed[tid * 2 + 1] + sRed[(TN + tid) * 2 + 1];
            }
        } else {
            // a 256-row block is two 128-row statistics blocks (wave rows 0-1 and 2-3)
            const int hb = tid / TN, cl = tid - hb * TN;
            const int n = n0 + cl;
            const int sblk = mt * (MW / 2) + hb;
            if (hb < MW / 2 && n < N && (long)sblk * 128 < M) {
                float* pp = a.partials + ((long)sblk * N + n) * 2;
                pp[0] = sRed[((2 * hb) * TN + cl) * 2] + sRed[((2 * hb + 1) * TN + cl) * 2];
                pp[1] = sRed[((2 * hb) * TN + cl) * 2 + 1] + sRed[((2 * hb + 1) * TN + cl) * 2 + 1];
            }
        }
    }
}

__host__ __device__ __forceinline__ int epi_variant(const gast_gemm_args& a) {      // EPI * 2 + ADD
    const int e = a.epi == GAST_EPI_BNRELU_BWD ? ((a.xdrop && a.drop.thresh != 0) ? 3 : 2) : a.epi;
    return e * 2 + (a.addend ? 1 : 0);
}

template <int EPI, bool ADD, int NI, int MW, int PAIR, int D = 2>
__global__ void __launch_bounds__(128 * MW, MW == 2 ? 2 : 1) gemm_big_kernel(const gast_gemm_args a, const BigPlan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    big_body<EPI, ADD, NI, MW, PAIR, D>(a, pl, blockIdx.x, smem);
}

struct BigBatch {
    gast_gemm_args a[GAST_GEMM_MAX_BATCH];
    BigPlan pl[GAST_GEMM_MAX_BATCH];
    int first[GAST_GEMM_MAX_BATCH + 1];
    int n;
};
static_assert(sizeof(BigBatch) <= 3840, "BigBatch travels as a kernel argument (4 KB limit)");
// several jobs with the SAME epilogue variant in one grid (one launch, one tail): G2 | G3 of a block, ...
template <int EPI, bool ADD, int NI, int MW, int PAIR, int D = 2>
__global__ void __launch_bounds__(128 * MW, MW == 2 ? 2 : 1) gemm_big_multi_kernel(const BigBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    big_body<EPI, ADD, NI, MW, PAIR, D>(b.a[d], b.pl[d], blockIdx.x - b.first[d], smem);
}

// ---- pre-split weight image, k-group-major: img[(k>>4) * ldimg + r * 32 + (k&15)] = bf16 hi(W[r][k]),  + 16: bf16 lo;
// zero for K <= k < Kp (rows past R are never written: the caller provides them zero-filled)
struct ImageBatch { gast_x3_image_job j[GAST_X3_IMAGE_MAX_BATCH]; int first[GAST_X3_IMAGE_MAX_BATCH + 1]; int n; };
static_assert(sizeof(ImageBatch) <= 3840, "ImageBatch travels as a kernel argument");
__global__ void __launch_bounds__(256) x3_image_kernel(const ImageBatch b) {
    int d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const gast_x3_image_job& j = b.j[d];
    if (j.f16 == 2) {
        // 16-bit storage, one product (PAIR = 3): a LAYOUT change only -- img[(k >> 5) * ldimg + r * 32 + (k & 31)] = W[r][k], zero
        // for K <= k < 32 * ceil(K / 32); one 16-byte chunk (8 values) per thread
        const int Kp8 = (j.K + 31) / 32 * 4;
        const long idx = (long)(blockIdx.x - b.first[d]) * 256 + threadIdx.x;
        if (idx >= (long)j.R * Kp8) return;
        const int r = (int)(idx / Kp8), k = (int)(idx - (long)r * Kp8) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k < j.K) v = *(const uint4*)((const bf16_t*)j.W + (long)r * j.ldw + k);
        *(uint4*)((bf16_t*)j.img + (long)(k >> 5) * j.ldimg + (long)r * 32 + (k & 31)) = v;
        return;
    }
    const int Kp4 = (j.K + 15) / 16 * 4;                          // 4-value chunks per padded row
    const long idx = (long)(blockIdx.x - b.first[d]) * 256 + threadIdx.x;
    if (idx >= (long)j.R * Kp4) return;
    const int r = (int)(idx / Kp4), k = (int)(idx - (long)r * Kp4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < j.K) v = *(const float4*)(j.W + (long)r * j.ldw + k);
    uint2 h, l;
    if (j.f16) split_pair4<2>(v.x, v.y, v.z, v.w, h, l);
    else split_pair4<1>(v.x, v.y, v.z, v.w, h, l);
    bf16_t* o = (bf16_t*)j.img + (long)(k >> 4) * j.ldimg + (long)r * 32 + (k & 15);
    *(uint2*)o = h;
    *(uint2*)(o + 16) = l;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

std::atomic<bool> big_setup_done[64];      // (zero-initialised: static storage)

}  // namespace

// Can this GEMM run on the large-M kernel?  Fills the plan when it can.  (Called by gast_gemm_ws / gast_gemm_multi in gemm.hip.)
int gast_gemm_big_plan(const gast_gemm_args& a, BigPlan& pl) {
    static const int enabled = getenv("GAST_GEMM_BIG") ? atoi(getenv("GAST_GEMM_BIG")) : 1;
    static const int min_rows = getenv("GAST_GEMM_BIG_MIN_M") ? atoi(getenv("GAST_GEMM_BIG_MIN_M")) : 8192;
    // Round 5: 16-bit storage (GAST_BF16 tensors: bfloat16, or binary16 in the f16 build) with ONE product per value (pair = 3),
    // every epilogue; 16-bit output only, no fp8 operands.  GAST_GEMM_BIG_H16=0: those GEMMs stay on gemm.hip's kernel.
    static const int h16_enabled = getenv("GAST_GEMM_BIG_H16") ? atoi(getenv("GAST_GEMM_BIG_H16")) : 1;
    const bool h16 = a.dtype == GAST_BF16;
    if (!enabled || (a.dtype != GAST_F32X3 && a.dtype != GAST_F32X3H && !h16)) return 0;
    if (h16 && (!h16_enabled || a.out_f32 || a.f8_scale)) return 0;
    // fp16 pairs: forward epilogues only (a gradient operand does not fit fp16's range; gemm.hip has every variant)
    if (a.dtype == GAST_F32X3H && a.epi == GAST_EPI_BNRELU_BWD) return 0;
    pl.pair = h16 ? 3 : a.dtype == GAST_F32X3H ? 2 : 1;
    const int esz = h16 ? 2 : 4, cv = 16 / esz;
    const long Ml = (long)a.B * a.Tn * a.J;
    static const int all_shapes = getenv("GAST_GEMM_BIG_ALL") ? atoi(getenv("GAST_GEMM_BIG_ALL")) : 0;
    // (the M = B*J stage stays on gemm.hip's split-K path: a split-K version of this kernel was built and measured slower on every
    // shape of that stage -- DESIGN.md, round 2b; it also cost the BNRELU_BWD variants 50 registers, and is gone)
    // Round 3: the short-K GEMMs of the M = B*J stage may take this kernel WITHOUT split-K (GAST_GEMM_BIG_SMALL_MIN_M rows and up, sum K
    // <= GAST_GEMM_BIG_SMALL_MAX_K): a lone block needs ~0.6 us per 16-deep K step, so 32 steps cost what gemm.hip's split-K kernel +
    // finish pair costs, without the partial-tile round trip
    // Round 4: with prefetch distance FOUR (GAST_GEMM_BIG_DEEP, five weight stages and four activation register sets at the 128 x 128 tile)
    // a lone block covers the memory latency by itself
    static const int small_min_m = getenv("GAST_GEMM_BIG_SMALL_MIN_M") ? atoi(getenv("GAST_GEMM_BIG_SMALL_MIN_M")) : 0;
    static const int small_max_k = getenv("GAST_GEMM_BIG_SMALL_MAX_K") ? atoi(getenv("GAST_GEMM_BIG_SMALL_MAX_K")) : 512;
    static const int deep = getenv("GAST_GEMM_BIG_DEEP") ? atoi(getenv("GAST_GEMM_BIG_DEEP")) : 0;
    if (Ml > 0x7fffff00L || a.N < 32) return 0;
    static const int deep_min_m = getenv("GAST_GEMM_BIG_DEEP_MIN_M") ? atoi(getenv("GAST_GEMM_BIG_DEEP_MIN_M")) : 1024;
    static const int deep_max_k = getenv("GAST_GEMM_BIG_DEEP_MAX_K") ? atoi(getenv("GAST_GEMM_BIG_DEEP_MAX_K")) : 1 << 30;
    const bool small = Ml < min_rows;
    bool want_deep = false;
    if (small) {
        int ks = 0;
        for (int s = 0; s < a.nseg; ++s) ks += a.seg[s].K;
        // (GAST_GEMM_BIG_DEEP_MIN_TILES: only GEMMs whose 128 x 128 tiles fill the chip without split-K)
        static const int deep_min_tiles = getenv("GAST_GEMM_BIG_DEEP_MIN_TILES") ? atoi(getenv("GAST_GEMM_BIG_DEEP_MIN_TILES")) : 0;
        const long tiles128 = ((Ml + 127) / 128) * ((a.N + 127) / 128);
        want_deep = deep && !h16 && Ml >= deep_min_m && ks <= deep_max_k && tiles128 >= deep_min_tiles;
        if (!want_deep && !(small_min_m > 0 && Ml >= small_min_m && ks <= small_max_k)) return 0;
    }
    pl.depth = 2;
    // tile width, measured on MI355X (scripts/gemm_table.py bf16x3, B = 128): 128 x 128 (NI = 2, 143-167 VGPRs, three blocks per
    // CU) for N <= 192 -- half of the wide tile would be empty -- and for every BNRELU_BWD epilogue (its X / addend values are
    // gathered per lane from the accumulator layout: at NI = 4 that epilogue spills 27-60 registers; at NI = 2 the short-K input
    // gradients run 10-16 % faster than on gemm.hip's kernel, the K = 768 one 7 % faster than at NI = 4); 128 x 256 (NI = 4)
    // otherwise (a wash on the forward GEMMs, 5 % better on the K = 1800 input gradient).  GAST_GEMM_BIG_NI=2|4 forces one width
    // (kernel tests), GAST_GEMM_BIG_BWD_NI=4|0 restores the round-2a rule for the BNRELU_BWD epilogue (4: wide tile for K >= 768,
    // gemm.hip below; 0: same, spelled as "no narrow BWD"), GAST_GEMM_BIG_NARROW=0 keeps N <= 192 on gemm.hip.
    static const int ni_env = getenv("GAST_GEMM_BIG_NI") ? atoi(getenv("GAST_GEMM_BIG_NI")) : 0;
    static const int bwd_ni = getenv("GAST_GEMM_BIG_BWD_NI") ? atoi(getenv("GAST_GEMM_BIG_BWD_NI")) : 2;
    static const int narrow = getenv("GAST_GEMM_BIG_NARROW") ? atoi(getenv("GAST_GEMM_BIG_NARROW")) : 1;
    int ksum = 0;
    for (int s = 0; s < a.nseg; ++s) ksum += a.seg[s].K;
    const bool bwd_epi = a.epi == GAST_EPI_BNRELU_BWD;
    pl.ni = (ni_env == 2 || ni_env == 4) ? ni_env : (a.N <= 192 || (bwd_epi && bwd_ni == 2)) ? 2 : 4;
    // block rows: 256 (MW = 4, one 8-wave block per CU) at the wide tile with GAST_GEMM_BIG_MW=4, else 128
    static const int mw_env = getenv("GAST_GEMM_BIG_MW") ? atoi(getenv("GAST_GEMM_BIG_MW")) : 2;
    static const int mw_min_k = getenv("GAST_GEMM_BIG_MW_MIN_K") ? atoi(getenv("GAST_GEMM_BIG_MW_MIN_K")) : 0;
    pl.mw = (mw_env == 4 && pl.ni == 4 && ksum >= mw_min_k && !h16) ? 4 : 2;
    if (want_deep) { pl.ni = 2; pl.mw = 2; pl.depth = 4; }
    if (!all_shapes) {
        if (a.N <= 192 && !narrow) return 0;
        if (bwd_epi && bwd_ni != 2 && ksum < 768) return 0;
    }
    if (a.nseg < 1 || a.nseg > GAST_MAX_SEG) return 0;
    int ntab = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const gast_gemm_seg& g = a.seg[s];
        if (!g.Wx || !aligned16(g.Wx) || g.ldwx % 8 || !g.A || !aligned16(g.A) || g.lda % cv || g.K % cv || g.K < cv) return 0;
        if ((long)a.B * g.map.T_total * a.J * g.lda * esz >= 0xffffffffL) return 0;      // 32-bit byte offsets into the activation tensor
        if (g.pro == GAST_PRO_BNRELU_DROP) return 0;
        pl.taboff[s] = -1;
        if (g.pro == GAST_PRO_BNRELU) {
            if (!g.scale || !g.shift) return 0;
            for (int q = 0; q < s; ++q)
                if (pl.taboff[q] >= 0 && a.seg[q].scale == g.scale && a.seg[q].shift == g.shift && a.seg[q].K == g.K) pl.taboff[s] = pl.taboff[q];
            if (pl.taboff[s] < 0) { pl.taboff[s] = ntab; ntab += (g.K + 3) / 4 * 4; }
        }
    }
    if (ntab > max_tab(pl.ni, pl.mw, pl.depth)) return 0;
    if (a.epi != GAST_EPI_PLAIN && !a.partials) return 0;
    if (a.epi == GAST_EPI_BNRELU_BWD && (!a.X || !a.xscale || !a.xshift)) return 0;
    // the epilogue addresses C / X / addend with 32-bit byte offsets inside buffer descriptors
    const long rowsC = (long)a.B * a.cmap.T_total * a.J;
    if (rowsC * a.ldc * esz >= 0x7fffffffL || (a.epi == GAST_EPI_BNRELU_BWD && rowsC * a.ldx * esz >= 0x7fffffffL)) return 0;
    if (a.C2 && (a.epi != GAST_EPI_BNRELU_BWD || a.addend || rowsC * a.ldc2 * esz >= 0x7fffffffL)) return 0;
    if (a.addend && (long)a.B * a.addmap.T_total * a.J * a.ldadd * esz >= 0x7fffffffL) return 0;
    static const int ablate = getenv("GAST_GEMM_BIG_ABLATE") ? atoi(getenv("GAST_GEMM_BIG_ABLATE")) : 0;
    pl.ablate = ablate;
    pl.M = (int)Ml;
    pl.tilesM = (pl.M + tm_of(pl.mw) - 1) / tm_of(pl.mw);
    pl.tilesN = (a.N + tn_of(pl.ni) - 1) / tn_of(pl.ni);
    pl.ntab = ntab;
    return 1;
}

static int big_lds_bytes(int ntab, int ni, int mw, int depth) { return off_tab(ni, mw, depth) + 2 * ntab * 4 + 6 * nt_of(mw) * 4; }

typedef void (*big_kernel_t)(const gast_gemm_args, const BigPlan);
template <int NI, int MW, int D = 2>
static big_kernel_t big_kernel_ni(int v, int pair) {
    if (pair == 2) {                // (variants 0..3: gast_gemm_big_plan keeps the BNRELU_BWD epilogues on bf16 pairs)
        switch (v) {
            case 0: return gemm_big_kernel<0, false, NI, MW, 2, D>;
            case 1: return gemm_big_kernel<0, true, NI, MW, 2, D>;
            case 2: return gemm_big_kernel<1, false, NI, MW, 2, D>;
            case 3: return gemm_big_kernel<1, true, NI, MW, 2, D>;
            default: return nullptr;
        }
    }
    if (pair == 3) {
        // (16-bit storage: the 128-row block tile at prefetch distance 2 only -- gast_gemm_big_plan never asks for the opt-in shapes)
        if constexpr (MW != 2 || D != 2) return nullptr;
        else switch (v) {
            case 0: return gemm_big_kernel<0, false, NI, MW, 3, D>;
            case 1: return gemm_big_kernel<0, true, NI, MW, 3, D>;
            case 2: return gemm_big_kernel<1, false, NI, MW, 3, D>;
            case 3: return gemm_big_kernel<1, true, NI, MW, 3, D>;
            case 4: return gemm_big_kernel<2, false, NI, MW, 3, D>;
            case 5: return gemm_big_kernel<2, true, NI, MW, 3, D>;
            case 6: return gemm_big_kernel<3, false, NI, MW, 3, D>;
            default: return gemm_big_kernel<3, true, NI, MW, 3, D>;
        }
    }
    switch (v) {
        case 0: return gemm_big_kernel<0, false, NI, MW, 1, D>;
        case 1: return gemm_big_kernel<0, true, NI, MW, 1, D>;
        case 2: return gemm_big_kernel<1, false, NI, MW, 1, D>;
        case 3: return gemm_big_kernel<1, true, NI, MW, 1, D>;
        case 4: return gemm_big_kernel<2, false, NI, MW, 1, D>;
        case 5: return gemm_big_kernel<2, true, NI, MW, 1, D>;
        case 6: return gemm_big_kernel<3, false, NI, MW, 1, D>;
        default: return gemm_big_kernel<3, true, NI, MW, 1, D>;
    }
}
// (the 256-row block tile exists at the wide tile only)
static big_kernel_t big_kernel(int v, int ni, int mw, int pair, int depth) {
    if (depth == 4) return big_kernel_ni<2, 2, 4>(v, pair);      // (gast_gemm_big_plan: depth 4 comes with NI = 2, MW = 2)
    return mw == 4 ? big_kernel_ni<4, 4>(v, pair) : ni == 2 ? big_kernel_ni<2, 2>(v, pair) : big_kernel_ni<4, 2>(v, pair);
}

typedef void (*big_multi_kernel_t)(const BigBatch);
template <int NI, int MW, int D = 2>
static big_multi_kernel_t big_multi_kernel_ni(int v, int pair) {
    if (pair == 2) {
        switch (v) {
            case 0: return gemm_big_multi_kernel<0, false, NI, MW, 2, D>;
            case 1: return gemm_big_multi_kernel<0, true, NI, MW, 2, D>;
            case 2: return gemm_big_multi_kernel<1, false, NI, MW, 2, D>;
            case 3: return gemm_big_multi_kernel<1, true, NI, MW, 2, D>;
            default: return nullptr;
        }
    }
    if (pair == 3) {
        // (16-bit storage: the 128-row block tile at prefetch distance 2 only -- gast_gemm_big_plan never asks for the opt-in shapes)
        if constexpr (MW != 2 || D != 2) return nullptr;
        else switch (v) {
            case 0: return gemm_big_multi_kernel<0, false, NI, MW, 3, D>;
            case 1: return gemm_big_multi_kernel<0, true, NI, MW, 3, D>;
            case 2: return gemm_big_multi_kernel<1, false, NI, MW, 3, D>;
            case 3: return gemm_big_multi_kernel<1, true, NI, MW, 3, D>;
            case 4: return gemm_big_multi_kernel<2, false, NI, MW, 3, D>;
            case 5: return gemm_big_multi_kernel<2, true, NI, MW, 3, D>;
            case 6: return gemm_big_multi_kernel<3, false, NI, MW, 3, D>;
            default: return gemm_big_multi_kernel<3, true, NI, MW, 3, D>;
        }
    }
    switch (v) {
        case 0: return gemm_big_multi_kernel<0, false, NI, MW, 1, D>;
        case 1: return gemm_big_multi_kernel<0, true, NI, MW, 1, D>;
        case 2: return gemm_big_multi_kernel<1, false, NI, MW, 1, D>;
        case 3: return gemm_big_multi_kernel<1, true, NI, MW, 1, D>;
        case 4: return gemm_big_multi_kernel<2, false, NI, MW, 1, D>;
        case 5: return gemm_big_multi_kernel<2, true, NI, MW, 1, D>;
        case 6: return gemm_big_multi_kernel<3, false, NI, MW, 1, D>;
        default: return gemm_big_multi_kernel<3, true, NI, MW, 1, D>;
    }
}
static big_multi_kernel_t big_multi_kernel(int v, int ni, int mw, int pair, int depth) {
    if (depth == 4) return big_multi_kernel_ni<2, 2, 4>(v, pair);
    return mw == 4 ? big_multi_kernel_ni<4, 4>(v, pair) : ni == 2 ? big_multi_kernel_ni<2, 2>(v, pair) : big_multi_kernel_ni<4, 2>(v, pair);
}

static void big_setup() {
    int dev = 0;
    hipGetDevice(&dev);               // function attributes are per device (nn.DataParallel replicas launch on several)
    dev &= 63;
    if (big_setup_done[dev].load(std::memory_order_acquire)) return;      // (idempotent set-up: a racing first call repeats it)
    for (int c = 0; c < 4; ++c) {          // (NI, MW, D) = (2, 2, 2), (4, 2, 2), (4, 4, 2), (2, 2, 4)
        const int ni = (c == 0 || c == 3) ? 2 : 4, mw = c == 2 ? 4 : 2, depth = c == 3 ? 4 : 2;
        for (int pair = 1; pair <= 3; ++pair)
            for (int v = 0; v < (pair == 2 ? 4 : 8); ++v) {
                if (pair == 3 && (mw != 2 || depth != 2)) continue;
                const hipError_t e1 = hipFuncSetAttribute((const void*)big_kernel(v, ni, mw, pair, depth), hipFuncAttributeMaxDynamicSharedMemorySize, lds_block(mw));
                const hipError_t e2 = hipFuncSetAttribute((const void*)big_multi_kernel(v, ni, mw, pair, depth), hipFuncAttributeMaxDynamicSharedMemorySize, lds_block(mw));
                if (e1 != hipSuccess || e2 != hipSuccess) {      // (e.g. static + dynamic LDS beyond the 160 KB of a CU: say so here, not at some later launch)
                    fprintf(stderr, "gast_hip: gemm_big set-up failed for variant %d NI %d MW %d pair %d depth %d: %s\n", v, ni, mw, pair, depth,
                            hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                    (void)hipGetLastError();
                }
            }
    }
    big_setup_done[dev].store(true, std::memory_order_release);
    if (getenv("GAST_GEMM_BIG_DEBUG")) {
        for (int c = 0; c < 4; ++c) {
            const int ni = (c == 0 || c == 3) ? 2 : 4, mw = c == 2 ? 4 : 2, depth = c == 3 ? 4 : 2;
            for (int pv = 0; pv < 12; pv += 2) {
                const int pair = pv < 8 ? 1 : 2, v = pv < 8 ? pv : pv - 8;
                int nb = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)big_kernel(v, ni, mw, pair, depth), nt_of(mw), big_lds_bytes(0, ni, mw, depth));
                hipFuncAttributes fa;
                (void)hipFuncGetAttributes(&fa, (const void*)big_kernel(v, ni, mw, pair, depth));
                fprintf(stderr, "gemm_big variant %d NI %d MW %d pair %d depth %d: %d blocks/CU at %d B LDS, %d regs, %zu B scratch\n", v, ni, mw, pair, depth, nb, big_lds_bytes(0, ni, mw, depth), fa.numRegs, (size_t)fa.localSizeBytes);
            }
        }
    }
}

int gast_gemm_big_launch(const gast_gemm_args& a, const BigPlan& pl, hipStream_t st) {
    big_setup();
    hipLaunchKernelGGL(big_kernel(epi_variant(a), pl.ni, pl.mw, pl.pair, pl.depth), dim3(pl.tilesM * pl.tilesN), dim3(nt_of(pl.mw)), big_lds_bytes(pl.ntab, pl.ni, pl.mw, pl.depth), st, a, pl);
    GAST_CHECK_LAUNCH();
    return 0;
}

int gast_gemm_big_launch_multi(const gast_gemm_args* args, const BigPlan* pls, int n, hipStream_t st) {
    big_setup();
    bool done[GAST_GEMM_MAX_BATCH] = {};
    for (int d0 = 0; d0 < n; ++d0) {          // one grid per (epilogue variant, tile width) present in the batch
        if (done[d0]) continue;
        const int v = epi_variant(args[d0]), ni = pls[d0].ni, mw = pls[d0].mw, pair = pls[d0].pair, depth = pls[d0].depth;
        BigBatch b;
        b.n = 0;
        b.first[0] = 0;
        int ntab = 0;
        for (int d = d0; d < n; ++d) {
            if (done[d] || epi_variant(args[d]) != v || pls[d].ni != ni || pls[d].mw != mw || pls[d].pair != pair || pls[d].depth != depth) continue;
            done[d] = true;
            const int k = b.n++;
            b.a[k] = args[d];
            b.pl[k] = pls[d];
            b.first[k + 1] = b.first[k] + pls[d].tilesM * pls[d].tilesN;
            if (pls[d].ntab > ntab) ntab = pls[d].ntab;
        }
        if (b.n == 1) hipLaunchKernelGGL(big_kernel(v, ni, mw, pair, depth), dim3(b.first[1]), dim3(nt_of(mw)), big_lds_bytes(ntab, ni, mw, depth), st, b.a[0], b.pl[0]);
        else hipLaunchKernelGGL(big_multi_kernel(v, ni, mw, pair, depth), dim3(b.first[b.n]), dim3(nt_of(mw)), big_lds_bytes(ntab, ni, mw, depth), st, b);
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
            const int cv = j.f16 == 2 ? 8 : 4;          // values per 16-byte chunk of the source operand (f16 == 2: a 16-bit operand)
            if (!j.W || !j.img || j.R < 1 || j.K < cv || j.f16 < 0 || j.f16 > 2) return GAST_EINVAL;
            if (j.K % cv || j.ldw % cv || !aligned16(j.W) || !aligned16(j.img) || j.ldimg % 8 || j.ldimg < (long)j.R * 32) return GAST_EALIGN;
            b.j[d] = j;
            const long chunks = j.f16 == 2 ? (long)j.R * ((j.K + 31) / 32 * 4) : (long)j.R * ((j.K + 15) / 16 * 4);
            b.first[d + 1] = b.first[d] + (int)((chunks + 255) / 256);
        }
        hipLaunchKernelGGL(x3_image_kernel, dim3(b.first[b.n]), dim3(256), 0, (hipStream_t)stream, b);
        GAST_CHECK_LAUNCH();
    }
    return 0;
}
