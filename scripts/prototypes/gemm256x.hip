// PROTOTYPE, second step (not part of the product, not built by __graft_entry__.build()): the 256x256 two-stage loop of
// gemm256.hip carrying the FEATURES of the production kernel (csrc/gemm.hip), so that next round's integration starts from a
// checked skeleton:
//   * K segments (concat on the channel axis, temporal taps) with a row map (T_total, t_stride, t_off) each; out-of-range taps
//     read as zero rows;
//   * per segment: A either straight to LDS by DMA (identity map, no prologue) or through ONE register set loaded two tiles ahead
//     (row map / BN+ReLU prologue; zero rows masked in registers), W always by DMA;
//   * epilogue on the LDS-staged bf16 tile: + bias, 16-byte stores with M / N tails, optional per-128-row-block column
//     statistics {sum y, sum y^2} of the rounded outputs (the production STATS epilogue: partials[mt128][N][2]).
// Build:  hipcc --offload-arch=gfx950 -O3 gemm256x.hip -o gemm256x        Run: ./gemm256x   (self-checks, then timings)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;

static inline bf16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static inline float bf2f_host(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return *(bf16_t*)&b; }

constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128, TILE_BYTES = 256 * ROWB, LDS_TILES = 4 * TILE_BYTES;
constexpr int MAX_SEG = 8;

struct RowMap { int T_total, t_stride, t_off; };
struct Seg { const bf16_t* A; int lda; const bf16_t* W; int ldw; int K; RowMap map; int pro; const float* scale; const float* shift; };
struct Args {
    int B, Tn, J, N, nseg;
    Seg seg[MAX_SEG];
    bf16_t* C; int ldc;
    const float* bias;
    int stats; float* partials;      // [ceil(M/128)][N][2]
    // tail split: the tiles past `full_tiles` are cut into `tsplit` K ranges; partial accumulators meet in `slabs`
    // ([tail tile][part][256*256] fp32), the last arriver (ticket in `tickets[tail tile]`, zeroed before the launch) reduces
    int full_tiles, tsplit; float* slabs; unsigned* tickets;
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
__device__ __forceinline__ void glds16(const bf16_t* g, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "memory", "m0");
}

__global__ void __launch_bounds__(512, 2) gemm256x_kernel(const Args a, int M, int ktab_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int N = a.N;
    const int tilesN = (N + BN - 1) / BN, tilesM = (M + BM - 1) / BM;
    int lb, part = 0, nparts = 1;
    if ((int)blockIdx.x < a.full_tiles || a.tsplit <= 1) lb = (int)blockIdx.x < tilesM * tilesN ? xcd_remap(blockIdx.x, min(a.full_tiles, tilesM * tilesN)) : 0;
    else { const int q = blockIdx.x - a.full_tiles; lb = a.full_tiles + q / a.tsplit; part = q - (q / a.tsplit) * a.tsplit; nparts = a.tsplit; }
    const int mt = lb / tilesN, nt = lb - mt * tilesN;
    const int m0 = mt * BM, n0 = nt * BN;

    // scale / shift of all prologue segments, concatenated in segment order, in LDS behind the stages
    float* sSc = (float*)(smem + LDS_TILES);
    float* sSh = sSc + ktab_floats;
    {
        int off = 0;
        for (int s = 0; s < a.nseg; ++s) {
            if (a.seg[s].pro) {
                for (int k = tid; k < a.seg[s].K; k += 512) { sSc[off + k] = a.seg[s].scale[k]; sSh[off + k] = a.seg[s].shift[k]; }
                off += a.seg[s].K;
            }
        }
    }
    // this thread's 4 staging rows: (b, t, j) of the output position; the mapped source row is recomputed per segment
    const int r8 = lane >> 3, s8 = lane & 7;
    int arow[4], achunk[4], ldsoff[4], pb[4], pt[4], pj[4];
    bool mvalid[4];
    const bf16_t* gW[4];
    int wrow_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + r8;
        arow[i] = row;
        achunk[i] = s8 ^ ((row >> 1) & 7);
        ldsoff[i] = (w * 4 + i) * 8 * ROWB;
        const int m = m0 + row;
        mvalid[i] = m < M;
        const int mm = mvalid[i] ? m : 0;
        const int TJ = a.Tn * a.J;
        pb[i] = mm / TJ;
        const int rem = mm - pb[i] * TJ;
        pt[i] = rem / a.J;
        pj[i] = rem - pt[i] * a.J;
        const int n = n0 + row;
        wrow_ok[i] = n < N ? n : N - 1;               // W rows past N: clamped (their columns are never stored)
    }
    // segment state (uniform) + per-thread source pointers
    int seg_l = 0, k_l = 0, off_l = 0;                 // the tile that will be LOADED next: segment, k offset, scale-table offset
    const bf16_t* pA[4];
    bool zrow[4];
    auto enter_seg = [&](int s) {
        const Seg& sg = a.seg[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = pt[i] * sg.map.t_stride + sg.map.t_off;
            const bool ok = mvalid[i] && ts >= 0 && ts < sg.map.T_total;
            const long srow = ok ? ((long)pb[i] * sg.map.T_total + ts) * a.J + pj[i] : 0;
            zrow[i] = mvalid[i] && !ok;               // an out-of-range tap of a stored row must read as zero
            pA[i] = sg.A + srow * sg.lda + achunk[i] * 8;
            gW[i] = sg.W + (long)wrow_ok[i] * sg.ldw + achunk[i] * 8;
        }
    };
    // a tile's description, carried along with the data it refers to
    struct Tile { int seg, k0, soff; bool direct; };
    auto next_tile = [&](Tile& t) {                    // returns the tile (seg_l, k_l) and advances
        t.seg = seg_l; t.k0 = k_l; t.soff = off_l;
        const Seg& sg = a.seg[seg_l];
        t.direct = !sg.pro && sg.map.t_stride == 1 && sg.map.t_off == 0 && sg.map.T_total == a.Tn;
        k_l += BK;
        if (k_l >= sg.K) {
            if (sg.pro) off_l += sg.K;
            k_l = 0; ++seg_l;
        }
    };
    int ntile_all = 0;
    for (int s = 0; s < a.nseg; ++s) ntile_all += a.seg[s].K / BK;
    const int t_begin = part * ntile_all / nparts, t_end = (part + 1) * ntile_all / nparts;
    const int ntile = t_end - t_begin;

    u32x4 ra[4];
    bool rz[4];                                        // zero-row flags of the rows held in ra
    Tile treg;                                         // the tile held in ra
    auto load_a = [&](const Tile& t) {                 // (pA / zrow are those of segment t.seg: enter_seg ran before)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra[i]) : "v"(pA[i] + t.k0) : "memory");
            rz[i] = zrow[i];
        }
    };
    auto dma_a = [&](const Tile& t, int stage) {
        const uint32_t sA = lds0 + stage * 2 * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(pA[i] + t.k0, __builtin_amdgcn_readfirstlane(sA + ldsoff[i]));
    };
    auto dma_w = [&](const Tile& t, int stage) {
        const uint32_t sW = lds0 + stage * 2 * TILE_BYTES + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(gW[i] + t.k0, __builtin_amdgcn_readfirstlane(sW + ldsoff[i]));
    };
    auto write_a = [&](const Tile& t, int stage) {
        unsigned char* sA = smem + stage * 2 * TILE_BYTES;
        const bool pro = a.seg[t.seg].pro != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t wv[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            if (pro) {
                const int k = t.soff + t.k0 + achunk[i] * 8;
                const float4 s0 = *(const float4*)(sSc + k), s1 = *(const float4*)(sSc + k + 4);
                const float4 h0 = *(const float4*)(sSh + k), h1 = *(const float4*)(sSh + k + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float lo = __uint_as_float(wv[p] << 16), hi = __uint_as_float(wv[p] & 0xffff0000u);
                    lo = fmaxf(fmaf(lo, sc[2 * p], sh[2 * p]), 0.f);
                    hi = fmaxf(fmaf(hi, sc[2 * p + 1], sh[2 * p + 1]), 0.f);
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
                    wv[p] = *(uint32_t*)&v;
                }
            }
            if (rz[i]) { wv[0] = 0u; wv[1] = 0u; wv[2] = 0u; wv[3] = 0u; }      // zero rows stay zero (relu(shift) must not leak in)
            *(uint4*)(sA + arow[i] * ROWB + s8 * 16) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    int offA[4], offB[2], keyA[4], keyB[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { const int row = wr * 128 + mi * 32 + li; offA[mi] = row * ROWB; keyA[mi] = (row >> 1) & 7; }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { const int row = wc * 64 + ni * 32 + li; offB[ni] = row * ROWB; keyB[ni] = (row >> 1) & 7; }

    __syncthreads();                                   // scale / shift table complete
    // ---- pipeline.  Invariant at the top of iteration t: stage t&1 holds tile t (after the wait + barrier); the register set
    // holds tile t+1 if that tile goes through registers.  The tile descriptors are generated in load order.
    Tile cur, nxt, nn;                                 // tiles t, t+1, t+2
    int cur_seg_entered = -1;
    auto ensure_seg = [&](int s) { if (s != cur_seg_entered) { enter_seg(s); cur_seg_entered = s; } };
    for (int sk = 0; sk < t_begin; ++sk) next_tile(cur);       // (a K-range block starts in the middle of the tile list)
    next_tile(cur);
    ensure_seg(cur.seg);
    dma_w(cur, 0);
    if (cur.direct) dma_a(cur, 0);
    else { load_a(cur); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); write_a(cur, 0); }
    bool have_nxt = ntile > 1, have_nn = false;
    if (have_nxt) {
        next_tile(nxt);
        ensure_seg(nxt.seg);
        if (!nxt.direct) { load_a(nxt); treg = nxt; }
    }
    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (have_nxt) {
            const int st = (t + 1) & 1;
            // tile t+1: registers -> LDS first (see gemm256.hip: the compiler's own wait must find nothing outstanding), then DMA
            if (!nxt.direct) write_a(nxt, st);
            ensure_seg(nxt.seg);                       // (pointers of nxt's segment; a no-op unless tile t+2 moved them on)
            dma_w(nxt, st);
            if (nxt.direct) dma_a(nxt, st);
            // tile t+2 into the register set
            have_nn = t + 2 < ntile;
            if (have_nn) {
                next_tile(nn);
                ensure_seg(nn.seg);
                if (!nn.direct) load_a(nn);
            }
        }
        const unsigned char* sA = smem + (t & 1) * 2 * TILE_BYTES;
        const unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            union { uint4 u; s16x8 s; } fa[4], fb[2];
            const int chunk = kc * 2 + lh;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi].u = *(const uint4*)(sA + offA[mi] + ((chunk ^ keyA[mi]) << 4));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[ni].u = *(const uint4*)(sW + offB[ni] + ((chunk ^ keyB[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi].s, fb[ni].s, acc[mi][ni], 0, 0, 0);
        }
        cur = nxt; nxt = nn; have_nxt = have_nn; have_nn = false;
    }
    // ---- tail split: fp32 partial tiles meet in global slabs; the last arriver of a tile adds the others to its registers
    if (nparts > 1) {
        __shared__ int s_last;
        const int tt = lb - a.full_tiles;
        float* mine = a.slabs + ((long)tt * nparts + part) * (BM * BN);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[(wr * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * 64 + ni * 32 + li] = acc[mi][ni][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(a.tickets + tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = ticket == (unsigned)(nparts - 1);
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!s_last) return;
        for (int p = 0; p < nparts; ++p) {
            if (p == part) continue;
            const float* other = a.slabs + ((long)tt * nparts + p) * (BM * BN);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[mi][ni][r] += __builtin_nontemporal_load(other + (wr * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * 64 + ni * 32 + li);
        }
    }
    // ---- epilogue: acc (+ bias) -> bf16 tile in LDS -> statistics, 16-byte stores
    __syncthreads();
    bf16_t* sC = (bf16_t*)smem;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = wc * 64 + ni * 32 + li;
        const float bias = (a.bias && n0 + col < N) ? a.bias[n0 + col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bf16_t yb = f2bf(acc[mi][ni][r] + bias);
                sC[row * BN + col] = yb;
                if (a.stats == 2 && m0 + row < M) { const float y = bf2f(yb); s1 += y; s2 = fmaf(y, y, s2); }
            }
        if (a.stats == 2) {      // statistics straight from the accumulators: a wave's 128 rows are exactly one statistics block
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0 && n0 + col < N && m0 + wr * 128 < M) {
                float* pp = a.partials + ((long)(mt * 2 + wr) * N + n0 + col) * 2;
                pp[0] = s1; pp[1] = s2;
            }
        }
    }
    __syncthreads();
    if (a.stats == 1) {  // thread = (column, 128-row half): sums of the ROUNDED outputs over the valid rows
        const int col = tid & 255, half = tid >> 8;
        if (n0 + col < N && m0 + half * 128 < M) {
            float s1 = 0.f, s2 = 0.f;
            const int rows = min(128, M - (m0 + half * 128));
            for (int r = 0; r < rows; ++r) {
                const float y = bf2f(sC[(half * 128 + r) * BN + col]);
                s1 += y;
                s2 = fmaf(y, y, s2);
            }
            float* pp = a.partials + ((long)(mt * 2 + half) * N + n0 + col) * 2;
            pp[0] = s1; pp[1] = s2;
        }
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int idx = i * 512 + tid;
        const int row = idx >> 5, chunk = idx & 31;
        if (m0 + row < M && n0 + chunk * 8 < N) *(uint4*)(a.C + (long)(m0 + row) * a.ldc + n0 + chunk * 8) = *(const uint4*)(sC + row * BN + chunk * 8);
    }
}

// ------------------------------------------------------------------------------------------------ reference + harness
__global__ void naive_kernel(const Args a, int M, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= a.N || m >= M) return;
    const int TJ = a.Tn * a.J;
    const int b = m / TJ, rem = m - b * TJ, t = rem / a.J, j = rem - t * a.J;
    float acc = 0.f;
    for (int s = 0; s < a.nseg; ++s) {
        const Seg& sg = a.seg[s];
        const int ts = t * sg.map.t_stride + sg.map.t_off;
        if (ts < 0 || ts >= sg.map.T_total) continue;
        const long srow = ((long)b * sg.map.T_total + ts) * a.J + j;
        for (int k = 0; k < sg.K; ++k) {
            float x = bf2f(sg.A[srow * sg.lda + k]);
            if (sg.pro) x = bf2f(f2bf(fmaxf(fmaf(x, sg.scale[k], sg.shift[k]), 0.f)));
            acc += x * bf2f(sg.W[(long)n * sg.ldw + k]);
        }
    }
    out[(long)m * a.N + n] = acc + (a.bias ? a.bias[n] : 0.f);
}

static unsigned g_seed = 12345;
static std::vector<bf16_t> rnd_bf16(size_t n) {
    std::vector<bf16_t> v(n);
    for (size_t i = 0; i < n; ++i) { g_seed = g_seed * 1664525u + 1013904223u; v[i] = f2bf_host(((g_seed >> 9) & 0xffff) / 65536.f - 0.5f); }
    return v;
}
template <typename T> static T* to_dev(const std::vector<T>& h) {
    T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d;
}

struct SegSpec { int K, T_total, t_stride, t_off, pro, share; };     // share >= 0: reuse the A tensor of that earlier segment

static int run(const char* tag, int B, int Tn, int J, int N, std::vector<SegSpec> segs, bool bias, int stats, bool check, int tsplit = 1) {
    const int M = B * Tn * J;
    Args a; memset(&a, 0, sizeof a);
    a.B = B; a.Tn = Tn; a.J = J; a.N = N; a.nseg = (int)segs.size();
    int ktab = 0;
    double flops = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const SegSpec& sp = segs[s];
        Seg& sg = a.seg[s];
        const long rowsA = (long)B * sp.T_total * J;
        sg.K = sp.K; sg.lda = sp.K; sg.ldw = sp.K; sg.map = {sp.T_total, sp.t_stride, sp.t_off}; sg.pro = sp.pro;
        sg.A = sp.share >= 0 ? a.seg[sp.share].A : to_dev(rnd_bf16((size_t)rowsA * sp.K));
        sg.W = to_dev(rnd_bf16((size_t)N * sp.K));
        if (sp.pro) {
            std::vector<float> hs(sp.K), hh(sp.K);
            for (int k = 0; k < sp.K; ++k) { hs[k] = 0.5f + ((k + s) % 7) * 0.25f; hh[k] = (((k + s) % 5) - 2) * 0.05f; }
            sg.scale = to_dev(hs); sg.shift = to_dev(hh);
            ktab += sp.K;
        }
        flops += 2.0 * M * N * sp.K;
    }
    bf16_t* dC; hipMalloc(&dC, (size_t)M * N * 2); hipMemset(dC, 0, (size_t)M * N * 2);
    a.C = dC; a.ldc = N;
    if (bias) { std::vector<float> hb(N); for (int n = 0; n < N; ++n) hb[n] = ((n % 9) - 4) * 0.1f; a.bias = to_dev(hb); }
    const int nb128 = (M + 127) / 128;
    float* dP = nullptr;
    if (stats) { hipMalloc(&dP, (size_t)nb128 * N * 2 * 4); hipMemset(dP, 0, (size_t)nb128 * N * 2 * 4); a.stats = stats; a.partials = dP; }
    const int lds = LDS_TILES + 2 * ktab * 4;
    hipFuncSetAttribute((const void*)gemm256x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    a.full_tiles = tiles; a.tsplit = 1;
    int grid = tiles;
    if (tsplit > 1 && tiles > 256 && tiles % 256) {      // full rounds as they are, the tiles of the last partial round cut in tsplit K ranges
        a.full_tiles = tiles / 256 * 256; a.tsplit = tsplit;
        const int tail = tiles - a.full_tiles;
        grid = a.full_tiles + tail * tsplit;
        hipMalloc(&a.slabs, (size_t)tail * tsplit * BM * BN * 4);
        hipMalloc(&a.tickets, tail * 4);
    }
    auto launch = [&]() {
        if (a.tsplit > 1) hipMemsetAsync(a.tickets, 0, (tiles - a.full_tiles) * 4, 0);
        hipLaunchKernelGGL(gemm256x_kernel, dim3(grid), dim3(512), lds, 0, a, M, ktab);
    };
    launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", tag, hipGetErrorString(e)); return 1; }
    int rc = 0;
    if (check) {
        float* dR; hipMalloc(&dR, (size_t)M * N * 4);
        hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, a, M, dR);
        std::vector<float> hR((size_t)M * N); std::vector<bf16_t> hC((size_t)M * N);
        hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0, ref = 0; size_t bad = 0;
        for (size_t i = 0; i < hR.size(); ++i) {
            const double d = fabs((double)bf2f_host(hC[i]) - hR[i]);
            if (d > worst) worst = d;
            if (fabs(hR[i]) > ref) ref = fabs(hR[i]);
            if (d > 0.02 * (fabs(hR[i]) + 1.0)) ++bad;
        }
        size_t sbad = 0; double sworst = 0;
        if (stats) {       // statistics of the kernel's OWN rounded outputs: must match sums over hC exactly up to fp32 summation order
            std::vector<float> hP((size_t)nb128 * N * 2);
            hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost);
            for (int blk = 0; blk < nb128; ++blk)
                for (int n = 0; n < N; ++n) {
                    double s1 = 0, s2 = 0;
                    for (int m = blk * 128; m < M && m < blk * 128 + 128; ++m) { const double y = bf2f_host(hC[(size_t)m * N + n]); s1 += y; s2 += y * y; }
                    const double d1 = fabs(hP[((size_t)blk * N + n) * 2] - s1), d2 = fabs(hP[((size_t)blk * N + n) * 2 + 1] - s2);
                    const double tol = 1e-4 * (fabs(s1) + s2 + 1.0);
                    if (d1 > sworst) sworst = d1;
                    if (d1 > tol || d2 > tol) ++sbad;
                }
        }
        printf("check %-28s M=%5d N=%4d: max abs err %.4f (max |ref| %.2f), %zu bad outputs, %zu bad statistics (worst %.2e)\n", tag, M, N, worst,
               ref, bad, sbad, sworst);
        hipFree(dR);
        if (bad || sbad) rc = 2;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("time  %-28s M=%5d N=%4d blocks %4d  %7.1f us  %6.1f TF/s\n", tag, M, N, grid, us, flops / us / 1e6);
    return rc;
}

int main() {
    int rc = 0;
    // small self-checks (naive reference): taps + prologue + statistics + bias; concat of plain segments; zero-row taps; N tail
    rc |= run("conv taps pro stats", 8, 19, 17, 256, {{256, 25, 1, 0, 1, -1}, {256, 25, 1, 3, 1, 0}, {256, 25, 1, 6, 1, 0}}, true, 1, true);
    rc |= run("concat plain stats", 10, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 1, true);
    rc |= run("dgrad taps zero rows", 6, 25, 17, 256, {{256, 19, 1, 0, 0, -1}, {256, 19, 1, -3, 0, 0}, {256, 19, 1, -6, 0, 0}}, false, 0, true);
    rc |= run("N tail 648 bias", 9, 25, 17, 648, {{128, 25, 1, 0, 0, -1}}, true, 0, true);
    rc |= run("mixed pro|plain stats", 7, 19, 17, 256, {{256, 19, 1, 0, 1, -1}, {256, 19, 1, 0, 0, -1}}, false, 1, true);
    rc |= run("conv taps pro stats(acc)", 8, 19, 17, 256, {{256, 25, 1, 0, 1, -1}, {256, 25, 1, 3, 1, 0}, {256, 25, 1, 6, 1, 0}}, true, 2, true);
    rc |= run("concat plain stats(acc)", 10, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 2, true);
    rc |= run("tail split x3 (check)", 18, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 1, -1}}, true, 2, true, 3);      // 23 x 2 = 46 tiles: no split (<= 256)
    rc |= run("tail split x4 (check)", 104, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 1, -1}}, true, 2, true, 4);     // 132 x 2 = 264 tiles: 8 tail tiles x 4
    if (rc) { printf("SELF-CHECK FAILED\n"); return 1; }
    // the step's shapes (B = 128)
    run("G4 s1 [X|ZLG] stats", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 1, false);
    run("conv1 taps pro stats", 128, 19, 17, 256, {{256, 25, 1, 0, 1, -1}, {256, 25, 1, 3, 1, 0}, {256, 25, 1, 6, 1, 0}}, true, 1, false);
    run("G1 s1 N=1288 bias", 128, 19, 17, 1288, {{256, 19, 1, 0, 0, -1}}, true, 0, false);
    run("G1 s0 N=648 bias", 128, 25, 17, 648, {{128, 25, 1, 0, 0, -1}}, true, 0, false);
    run("1x1 s1 pro stats", 128, 19, 17, 256, {{256, 19, 1, 0, 1, -1}}, true, 1, false);
    // statistics from the accumulators instead of a second pass over the staged tile
    run("G4 s1 stats(acc)", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 2, false);
    run("G4 s1 no stats", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 0, false);
    run("conv1 taps pro stats(acc)", 128, 19, 17, 256, {{256, 25, 1, 0, 1, -1}, {256, 25, 1, 3, 1, 0}, {256, 25, 1, 6, 1, 0}}, true, 2, false);
    run("G4 s1 stats(acc) tail x2", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 2, false, 2);
    run("G4 s1 stats(acc) tail x3", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 2, false, 3);
    run("G4 s1 stats(acc) tail x4", 128, 19, 17, 512, {{256, 19, 1, 0, 0, -1}, {512, 19, 1, 0, 0, -1}}, false, 2, false, 4);
    // tile-count quantisation: the same GEMM with M = 32768 (256 tiles: one round), 65536 (512: two rounds), 49152 (384: 1.5 rounds)
    run("G4-like M=32768 (256 tiles)", 128, 16, 16, 512, {{256, 16, 1, 0, 0, -1}, {512, 16, 1, 0, 0, -1}}, false, 2, false);
    run("G4-like M=49152 (384 tiles)", 192, 16, 16, 512, {{256, 16, 1, 0, 0, -1}, {512, 16, 1, 0, 0, -1}}, false, 2, false);
    run("G4-like M=65536 (512 tiles)", 256, 16, 16, 512, {{256, 16, 1, 0, 0, -1}, {512, 16, 1, 0, 0, -1}}, false, 2, false);
    return 0;
}
