// FAST-9/16 corner detection with per-cell 3x3 non-max suppression and per-cell top-k, for gfx950.
//
// Replaces, per grid cell, cv::FAST(img(roi), pts, threshold, true) + std::sort + truncation
// (reference: src/track/Grider_FAST.h:106-123).  A cell is a cv::Mat ROI in the reference, so no corner can lie
// in the 3-pixel band inside each cell border and NMS never sees a neighbouring cell's score: the score map is
// therefore zero outside the cell insets, and a single image-wide 3x3 strict-maximum test equals per-cell NMS.
//
//   fast_score_kernel : u8 image -> u8 score map (0 = not a corner), HBM-bound (reads level 0 once).
//   fast_nms_kernel   : strict 3x3 maxima -> per-cell candidate lists (unordered, one atomic per survivor).
//   fast_topk_kernel  : one workgroup per cell extracts the first `per_cell` candidates in the total order
//                       (score desc, y asc, x asc).
#pragma once
#include "ft_common.hpp"

namespace ftd {

struct FastGrid {
    int w, h;
    int size_x, size_y;     // cell size   (Grider_FAST.h:80-81)
    int ct_cols, ct_rows;   // cell counts (Grider_FAST.h:91-92)
    int threshold;
    int per_cell;
    int cand_cap;           // candidate slots per cell
};

// strength = max over the 16 arcs of 9 contiguous circle pixels of min(d) (darker arc) or min(-d) (brighter arc),
// d[k] = v - p[k].  The pixel is a corner iff strength > threshold; its OpenCV score is strength - 1.
__device__ inline int fast_strength(const int (&d)[16]) {
    int mn2[16], mx2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        best = max(best, max(mn9, -mx9));
    }
    return best;
}

__global__ void __launch_bounds__(256) fast_score_kernel(ArenaDesc A, ImgList L, FastGrid G, uint8_t* __restrict__ score, uint32_t sc_pitch, uint64_t sc_img_bytes) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= G.w || y >= G.h) return;
    const int slot = L.idx[blockIdx.z];
    uint8_t* sc = score + (uint64_t)blockIdx.z * sc_img_bytes + (size_t)y * sc_pitch + x;
    // position inside the cell; pixels right of / below the last full cell belong to no cell
    const int cxi = x / G.size_x, cyi = y / G.size_y;
    const int lx = x - cxi * G.size_x, ly = y - cyi * G.size_y;
    uint8_t out = 0;
    if (cxi < G.ct_cols && cyi < G.ct_rows && lx >= 3 && lx < G.size_x - 3 && ly >= 3 && ly < G.size_y - 3) {
        const uint8_t* __restrict__ img = level_img(A, slot, 0);
        const uint32_t pitch = A.lv[0].img_pitch;
        const uint8_t* p = img + (size_t)y * pitch + x;
        const int v = p[0];
        const int t = G.threshold;
        // quick reject: a 9-arc contains one pixel of every opposite pair
        const int p0 = p[3 * (int)pitch], p8 = p[-3 * (int)pitch], p4 = p[3], p12 = p[-3];
        const int lo = v - t, hi = v + t;
        const bool dark = (p0 < lo || p8 < lo) && (p4 < lo || p12 < lo);
        const bool bright = (p0 > hi || p8 > hi) && (p4 > hi || p12 > hi);
        if (dark || bright) {
            int d[16];
            const int ip = (int)pitch;
            d[0] = v - p0;               d[1] = v - p[3 * ip + 1];   d[2] = v - p[2 * ip + 2];   d[3] = v - p[ip + 3];
            d[4] = v - p4;               d[5] = v - p[-ip + 3];      d[6] = v - p[-2 * ip + 2];  d[7] = v - p[-3 * ip + 1];
            d[8] = v - p8;               d[9] = v - p[-3 * ip - 1];  d[10] = v - p[-2 * ip - 2]; d[11] = v - p[-ip - 3];
            d[12] = v - p12;             d[13] = v - p[ip - 3];      d[14] = v - p[2 * ip - 2];  d[15] = v - p[3 * ip - 1];
            const int s = fast_strength(d);
            if (s > t) out = (uint8_t)(s - 1);
        }
    }
    *sc = out;
}

// candidate key: score in the top byte, then inverted cell-local y, then inverted cell-local x, so that a larger
// key is earlier in (score desc, y asc, x asc)
__device__ inline uint32_t fast_key(int score, int ly, int lx) {
    return ((uint32_t)score << 24) | ((uint32_t)(4095 - ly) << 12) | (uint32_t)(4095 - lx);
}

__global__ void __launch_bounds__(256) fast_nms_kernel(FastGrid G, const uint8_t* __restrict__ score, uint32_t sc_pitch, uint64_t sc_img_bytes,
                                                      uint32_t* __restrict__ cand, int* __restrict__ cand_count) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < 1 || y < 1 || x >= G.w - 1 || y >= G.h - 1) return;
    const uint8_t* sc = score + (uint64_t)blockIdx.z * sc_img_bytes + (size_t)y * sc_pitch + x;
    const int s = sc[0];
    if (s == 0) return;   // a corner's score is >= threshold; threshold 0 corners with score 0 never pass strict NMS
    const int ip = (int)sc_pitch;
    if (s > sc[-1] && s > sc[1] && s > sc[-ip - 1] && s > sc[-ip] && s > sc[-ip + 1] && s > sc[ip - 1] && s > sc[ip] && s > sc[ip + 1]) {
        const int cxi = x / G.size_x, cyi = y / G.size_y;
        const int cell = cyi * G.ct_cols + cxi;
        const int gcell = blockIdx.z * (G.ct_cols * G.ct_rows) + cell;
        const int slot = atomicAdd(&cand_count[gcell], 1);
        if (slot < G.cand_cap) cand[(uint64_t)gcell * G.cand_cap + slot] = fast_key(s, y - cyi * G.size_y, x - cxi * G.size_x);
    }
}

// one workgroup per (image, cell): repeated "largest key below the previous one"
__global__ void __launch_bounds__(256) fast_topk_kernel(FastGrid G, const uint32_t* __restrict__ cand, const int* __restrict__ cand_count,
                                                       ft_keypoint* __restrict__ out, int* __restrict__ out_count) {
    __shared__ uint32_t red[4];
    const int gcell = blockIdx.x;
    const int cell = gcell % (G.ct_cols * G.ct_rows);
    const int n = min(cand_count[gcell], G.cand_cap);
    const uint32_t* c = cand + (uint64_t)gcell * G.cand_cap;
    const int keep = min(n, G.per_cell);
    const int cell_x = (cell % G.ct_cols) * G.size_x, cell_y = (cell / G.ct_cols) * G.size_y;
    uint32_t bound = 0xffffffffu;
    for (int r = 0; r < keep; r++) {
        uint32_t best = 0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const uint32_t k = c[i];
            if (k < bound && k > best) best = k;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, off));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
        __syncthreads();
        best = max(max(red[0], red[1]), max(red[2], red[3]));
        __syncthreads();
        if (threadIdx.x == 0) {
            ft_keypoint kp;
            kp.x = (float)(cell_x + 4095 - (int)(best & 0xfff));
            kp.y = (float)(cell_y + 4095 - (int)((best >> 12) & 0xfff));
            kp.response = (float)(best >> 24);
            out[(uint64_t)gcell * G.per_cell + r] = kp;
        }
        bound = best;
    }
    if (threadIdx.x == 0) out_count[gcell] = keep;
}

}  // namespace ftd
