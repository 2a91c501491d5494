// Two launches that each carry two independent pieces of a frame set's front work, for gfx950.
//
// After the pass that writes level 0 and level 1 of a frame set, the detector (needs level 0) and the pyrDown step
// level 1 -> 2 are independent of each other, and so are the per-cell ranking of the detector's candidates and the
// step level 2 -> 3.  On one queue they would run one after the other, five small launches whose time is mostly the
// latency of each; here the workgroups of both pieces share a launch (the first `level_blocks` workgroups filter
// pyramid tiles, the others run detector tiles, one per wavefront, or rank cells, four per workgroup).  The pieces
// are the same device functions the separate kernels call, so the results are the same bytes.
#pragma once
#include "ft_kernels_fast.hpp"
#include "ft_kernels_pyramid.hpp"

namespace ftd {

constexpr size_t FRONT_LDS_BYTES = sizeof(FastTileLds) * 4 > (size_t)(LV_LDS_ROWS * LV_LDS_PITCH) ? sizeof(FastTileLds) * 4 : (size_t)(LV_LDS_ROWS * LV_LDS_PITCH);

// detector tiles of level 0 (four per workgroup) + pyrDown tiles level -> level + 1 (lw x lh: size of `level`)
__global__ void __launch_bounds__(256) front_detect_level_kernel(ArenaDesc A, ImgList L, FastGrid G, uint32_t* __restrict__ cand, int* __restrict__ cand_count,
                                                                int level, int lw, int lh, int level_blocks) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[FRONT_LDS_BYTES];
    if ((int)blockIdx.x < level_blocks) {   // (the same for the whole workgroup)
        const int total = ((lw + LV_TX - 1) / LV_TX) * ((lh + LV_TY - 1) / LV_TY) * L.n;
        const int item = xcd_item(blockIdx.x, total);
        if (item >= total) return;
        FramePtrs none;   // never read in this form
        level_tile<false>(A, L, none, 0, level, lw, lh, (uint64_t)0, 0, 1, item, threadIdx.x, lds);
        return;
    }
    const int total = fast_tiles_x(G.w) * fast_tiles_y(G.h) * L.n;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int item = 4 * xcd_item((int)blockIdx.x - level_blocks, (total + 3) / 4) + wave;
    if (item >= total) return;              // (the same for the whole wavefront; the tiles' wavefronts never meet at a barrier)
    fast_detect_tile(A, L, G, cand, cand_count, item, threadIdx.x & 63, reinterpret_cast<FastTileLds*>(lds)[wave]);
}

// per-cell ranking of the detector's candidates (four cells per workgroup) + pyrDown tiles level -> level + 1
__global__ void __launch_bounds__(256) front_topk_level_kernel(FastGrid G, const uint32_t* __restrict__ cand, int* __restrict__ cand_count,
                                                              ft_keypoint* __restrict__ out, int* __restrict__ out_count, int* __restrict__ surv_count,
                                                              int total_cells, ArenaDesc A, ImgList L, int level, int lw, int lh, int level_blocks) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LV_LDS_ROWS * LV_LDS_PITCH];
    if ((int)blockIdx.x < level_blocks) {
        const int total = ((lw + LV_TX - 1) / LV_TX) * ((lh + LV_TY - 1) / LV_TY) * L.n;
        const int item = xcd_item(blockIdx.x, total);
        if (item >= total) return;
        FramePtrs none;
        level_tile<false>(A, L, none, 0, level, lw, lh, (uint64_t)0, 0, 1, item, threadIdx.x, lds);
        return;
    }
    const int gcell = ((int)blockIdx.x - level_blocks) * 4 + ((int)threadIdx.x >> 6);
    if (gcell >= total_cells) return;
    fast_topk_cell(G, cand, cand_count, out, out_count, surv_count, gcell, threadIdx.x & 63);
}

}  // namespace ftd
