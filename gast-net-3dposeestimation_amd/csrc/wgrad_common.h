// Shared by wgrad.hip and wgrad_wide.hip: output-tile / row-map decoding and the job table of a multi-job weight-gradient launch.
#pragma once
#include "common.h"

namespace gastwg {

struct TileCoord { int rt, seg, st; };

__device__ __forceinline__ TileCoord decode_tile(const gast_wgrad_args& a, int tile, int tilesS_total, int bt) {
    TileCoord c;
    c.rt = tile / tilesS_total;
    int rem = tile - c.rt * tilesS_total;
    c.seg = 0;
    for (int s = 0; s < a.nseg; ++s) {
        int ts = (a.seg[s].S + bt - 1) / bt;
        if (rem < ts) { c.seg = s; break; }
        rem -= ts;
    }
    c.st = rem;
    return c;
}

__device__ __forceinline__ bool is_ident(const gast_rowmap& mp, int Tn) { return mp.t_stride == 1 && mp.t_off == 0 && mp.T_total == Tn; }

__device__ __forceinline__ void rows_for(const gast_wgrad_args& a, const gast_wgrad_seg& sg, int m, int M,
                                         int& prow, int& qrow) {
    prow = -1; qrow = -1;
    if (is_ident(a.pmap, a.Tn) && is_ident(sg.map, a.Tn)) {     // (most jobs: no integer divisions on the per-step path)
        if (m < M) { prow = m; qrow = m; }
        return;
    }
    if (m < M) {
        int TJ = a.Tn * a.J;
        int b = m / TJ, rem = m - b * TJ;
        int t = rem / a.J, j = rem - t * a.J;
        prow = (int)map_row(a.pmap, b, t, j, a.J);
        qrow = (int)map_row(sg.map, b, t, j, a.J);
        if (prow < 0 || qrow < 0) { prow = -1; qrow = -1; }
    }
}

// Several weight gradients in ONE launch (gast_wgrad_multi): the split-M atomics cost 30-60 % of a stand-alone weight-gradient
// launch because every launch needs >= 768 blocks by itself, i.e. 768 partial 128x128 tiles (50 MB of fp32 atomics) whatever
// its size.  Sharing the block budget among all the weight gradients of a stage divides that volume by their number and
// leaves one tail instead of one per launch.
struct WgBatch {
    gast_wgrad_args a[GAST_WGRAD_MAX_BATCH];
    int first[GAST_WGRAD_MAX_BATCH + 1];     // first block of each job
    int M[GAST_WGRAD_MAX_BATCH], tilesS[GAST_WGRAD_MAX_BATCH], splitM[GAST_WGRAD_MAX_BATCH], mchunk[GAST_WGRAD_MAX_BATCH];
    int tfirst[GAST_WGRAD_MAX_BATCH + 1];    // chunk-major order: first output tile of each job among all tiles of the batch
    int n, chunk_major;
};
// block -> (job, output tile, M chunk).  chunk_major: logical blocks are ordered (M chunk, job, tile) and an XCD owns a contiguous
// logical range (xcd_remap), so the blocks that run together on one L2 reduce over the SAME rows: the P / Q panels that the
// tiles of a job -- and the jobs of a stage -- share are read from HBM once.  (PMC: 684 MB per launch for 244 MB of operands in
// tile-major order.)  Measured: the step is 3 % SLOWER with it (3.29 vs 3.20 ms) -- opt-in via GAST_WGRAD_ORDER=1.
__device__ __forceinline__ bool wg_decode(const WgBatch& b, int& d, int& tile, int& sp) {
    if (b.chunk_major) {
        const int lb = xcd_remap(blockIdx.x, gridDim.x);
        const int total = b.tfirst[b.n];
        sp = lb / total;
        const int r = lb - sp * total;
        d = 0;
        while (d + 1 < b.n && r >= b.tfirst[d + 1]) ++d;
        tile = r - b.tfirst[d];
        return sp < b.splitM[d];
    }
    d = 0;
    while (d + 1 < b.n && (int)blockIdx.x >= b.first[d + 1]) ++d;
    const int lb = blockIdx.x - b.first[d];
    tile = lb / b.splitM[d];
    sp = lb - tile * b.splitM[d];
    return true;
}
static_assert(sizeof(WgBatch) <= 8192, "WgBatch travels by value in the HSA kernarg segment (4.4 KB; no 4 KB CUDA-style limit on gfx950)");

}  // namespace gastwg
using namespace gastwg;

// wgrad_wide.hip: the 256 x 256-tile bf16x3 weight gradient (one 8-wave block per CU).  Returns a hipError_t as int.
// h16: the one-product variant on 16-bit storage (GAST_BF16 jobs; 32 reduction rows per step)
__attribute__((visibility("hidden"))) int gast_wgrad_x3_wide_launch(const WgBatch& b, unsigned grid, bool any_drop, hipStream_t st, bool h16 = false);
