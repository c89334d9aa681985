// Internal interface between gemm.hip (argument validation, dispatch), gemm_big.hip (the 128x256 / 128x128-tile pipelined GAST_F32X3 kernel)
// and gemm_bj.hip (the 64-row-tile kernel of the M = B*J stage).
#pragma once
#include "common.h"

struct BigPlan {
    int M, tilesM, tilesN;
    int ni;                       // 32-column MFMA tiles per wave: 4 = 256-column block tile, 2 = 128
    int mw;                       // 64-row wave rows per block: 2 = 128-row block tile (256 threads), 4 = 256 rows (512 threads)
    int depth;                    // prefetch distance in K steps: 2, or 4 (NI = 2, MW = 2 only: the M = B*J stage, GAST_GEMM_BIG_DEEP)
    int pair;                     // operand pairs: 1 = bf16 hi/lo (GAST_F32X3), 2 = fp16 hi/lo (GAST_F32X3H, forward epilogues only)
    int ntab;                     // floats in the scale table (= in the shift table) a block keeps in LDS
    int taboff[GAST_MAX_SEG];     // offset of the segment's scale/shift in the tables (-1: no prologue)
    int ablate;                   // GAST_GEMM_BIG_ABLATE (profiling aid, results are wrong when set): 1 no MFMA, 2 no fragment reads,
                                  // 4 no weight DMA, 8 no activation loads, 16 no activation LDS writes, 32 no epilogue
};

// (internal to the library: hidden, so that the dynamic symbol table holds the C ABI of include/gast_hip.h and nothing else)
// 1 when the GEMM can run on the big-tile kernel (fills the plan), else 0: the caller uses the 128x128 kernel of gemm.hip
__attribute__((visibility("hidden"))) int gast_gemm_big_plan(const gast_gemm_args& a, BigPlan& pl);
__attribute__((visibility("hidden"))) int gast_gemm_big_launch(const gast_gemm_args& a, const BigPlan& pl, hipStream_t st);
__attribute__((visibility("hidden"))) int gast_gemm_big_launch_multi(const gast_gemm_args* args, const BigPlan* pls, int n, hipStream_t st);

// gemm_bj.hip: the M = B*J stage (rows < GAST_GEMM_BIG_MIN_M) in GAST_F32X3 / GAST_F32X3H -- 64-row tiles, the K loop split over two
// k-groups of waves INSIDE the block, no workspace and no finish launch.  `partials` must arrive zero-filled (as on gemm.hip's
// split-K path, which it replaces): the caller only takes this path when it was given a workspace, i.e. through gast_gemm_ws / _multi.
struct BjPlan {
    int M, tilesM, tilesN;
    int nj;                       // 32-column MFMA tiles per wave: 1 = 64-column block tile, 2 = 128 (picked per launch)
    int pair;                     // 1 = bf16 hi/lo, 2 = fp16 hi/lo
    int ntab;
    int taboff[GAST_MAX_SEG];
    int fast;                     // every segment's K a multiple of 128 and no zero rows: the lean K loop (bj_body_fast)
    int ntile32;                  // 32-value K steps over all segments
    int ablate;                   // GAST_GEMM_BJ_ABLATE (profiling aid, results are wrong when set): 1 no statistics atomics, 2 no epilogue, 4 no K loop
};
__attribute__((visibility("hidden"))) int gast_gemm_bj_plan(const gast_gemm_args& a, BjPlan& pl);
__attribute__((visibility("hidden"))) int gast_gemm_bj_launch_multi(const gast_gemm_args* args, BjPlan* pls, int n, hipStream_t st);
