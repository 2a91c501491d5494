// FAST-9/16 corner detection with per-cell 3x3 non-max suppression and per-cell top-k, for gfx950.
//
// Replaces, per grid cell, cv::FAST(img(roi), pts, threshold, true) + std::sort + truncation
// (reference: src/track/Grider_FAST.h:106-123).  A cell is a cv::Mat ROI in the reference, so no corner can lie
// in the 3-pixel band inside each cell border and NMS never sees a neighbouring cell's score: the score map is
// therefore zero outside the cell insets, and a single image-wide 3x3 strict-maximum test equals per-cell NMS.
//
//   fast_detect_kernel : u8 image -> per-cell candidate lists (score + position keys, unordered); reads level 0
//                        once, keeps pixels and scores in LDS, one global atomic per NMS survivor.
//   fast_topk_kernel   : one workgroup per cell extracts the first `per_cell` candidates in the total order
//                        (score desc, y asc, x asc).
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
// Sliding 9-wide min/max as two rounds of 3-input min/max (v_min3 / v_max3).
__device__ inline int fast_strength(const int (&d)[16]) {
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        mx3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    }
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
        const int mx9 = max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]);
        best = max(max(best, mn9), -mx9);
    }
    return best;
}

// candidate key: score in the top byte, then inverted cell-local y, then inverted cell-local x, so that a larger
// key is earlier in (score desc, y asc, x asc)
__device__ inline uint32_t fast_key(int score, int ly, int lx) {
    return ((uint32_t)score << 24) | ((uint32_t)(4095 - ly) << 12) | (uint32_t)(4095 - lx);
}

constexpr int FD_TX = 64, FD_TY = 16;            // output tile
constexpr int FD_PW = FD_TX + 8, FD_PH = FD_TY + 8;   // pixel tile: 4-pixel apron (3 for the circle + 1 for NMS)
constexpr int FD_SW = FD_TX + 2, FD_SH = FD_TY + 2;   // score tile: 1-pixel apron
constexpr int FD_SPITCH = 68;

// Fused detector: one 256-thread workgroup owns a 64x16 tile.  The pixel tile is staged in LDS once; a cheap
// compass test compacts the candidates; the exact corner strength is evaluated only for those (dense lanes);
// strict 3x3 maxima of the LDS score tile go straight to the per-cell candidate lists.  No score map in HBM.
__global__ void __launch_bounds__(256) fast_detect_kernel(ArenaDesc A, ImgList L, FastGrid G, uint32_t* __restrict__ cand, int* __restrict__ cand_count) {
    __shared__ __attribute__((aligned(16))) uint8_t pix[FD_PH * FD_PW];
    __shared__ uint8_t sc[FD_SH * FD_SPITCH];
    __shared__ uint16_t list[FD_SW * FD_SH];
    __shared__ uint8_t col_ok[FD_SW], row_ok[FD_SH];   // column / row lies in the inset of some cell
    __shared__ int list_n;
    const int t = threadIdx.x;
    // work list: (image, tile row, tile column), images outermost
    const int tiles_x = (G.w + FD_TX - 1) / FD_TX, tiles_y = (G.h + FD_TY - 1) / FD_TY;
    const int item = xcd_item(blockIdx.x, tiles_x * tiles_y * L.n);
    if (item >= tiles_x * tiles_y * L.n) return;
    const int img_i = item / (tiles_x * tiles_y), tile_i = item - img_i * (tiles_x * tiles_y);
    const int slot = L.idx[img_i];
    const int X0 = (tile_i % tiles_x) * FD_TX, Y0 = (tile_i / tiles_x) * FD_TY;
    const uint8_t* __restrict__ img = level_img(A, slot, 0);
    const uint32_t pitch = A.lv[0].img_pitch;

    // ---- stage pixels: 24 rows x 18 dwords; coordinates are clamped (only insets are ever evaluated, and an
    //      inset pixel's circle lies inside the image) ----
    for (int i = t; i < FD_PH * (FD_PW / 4); i += 256) {
        const int rr = i / (FD_PW / 4), d = i - rr * (FD_PW / 4);
        const int gy = min(max(Y0 - 4 + rr, 0), G.h - 1);
        const int gxd = min(max((X0 - 4) / 4 + d, 0), (int)(pitch / 4) - 1);
        reinterpret_cast<uint32_t*>(pix)[rr * (FD_PW / 4) + d] = reinterpret_cast<const uint32_t*>(img + (size_t)gy * pitch)[gxd];
    }
    for (int i = t; i < FD_SH * FD_SPITCH / 4; i += 256) reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if (t == 0) list_n = 0;
    // cell membership is separable: one division per column and per row of the tile instead of two per pixel
    if (t < FD_SW) {
        const int gx = X0 - 1 + t;
        const int cxi = gx >= 0 ? gx / G.size_x : 0, lx = gx - cxi * G.size_x;
        col_ok[t] = (gx >= 0 && gx < G.w && cxi < G.ct_cols && lx >= 3 && lx < G.size_x - 3) ? 1 : 0;
    } else if (t >= 128 && t < 128 + FD_SH) {
        const int gy = Y0 - 1 + (t - 128);
        const int cyi = gy >= 0 ? gy / G.size_y : 0, ly = gy - cyi * G.size_y;
        row_ok[t - 128] = (gy >= 0 && gy < G.h && cyi < G.ct_rows && ly >= 3 && ly < G.size_y - 3) ? 1 : 0;
    }
    __syncthreads();

    // ---- compass test on the 66x18 score region: a 9-arc holds one pixel of every opposite pair ----
    const int thr = G.threshold;
    for (int i = t; i < FD_SW * FD_SH; i += 256) {
        const int sj = i / FD_SW, si = i - sj * FD_SW;
        if (!(col_ok[si] & row_ok[sj])) continue;
        const uint8_t* p = &pix[(sj + 3) * FD_PW + si + 3];
        const int v = p[0], lo = v - thr, hi = v + thr;
        const int p0 = p[3 * FD_PW], p8 = p[-3 * FD_PW], p4 = p[3], p12 = p[-3];
        const bool dark = (p0 < lo || p8 < lo) && (p4 < lo || p12 < lo);
        const bool bright = (p0 > hi || p8 > hi) && (p4 > hi || p12 > hi);
        if (dark || bright) list[atomicAdd(&list_n, 1)] = (uint16_t)i;
    }
    __syncthreads();

    // ---- exact strength for the compacted candidates ----
    const int n_list = list_n;
    for (int c = t; c < n_list; c += 256) {
        const int i = list[c];
        const int sj = i / FD_SW, si = i - sj * FD_SW;
        const uint8_t* p = &pix[(sj + 3) * FD_PW + si + 3];
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * FD_PW];      d[1] = v - p[3 * FD_PW + 1];   d[2] = v - p[2 * FD_PW + 2];   d[3] = v - p[FD_PW + 3];
        d[4] = v - p[3];              d[5] = v - p[-FD_PW + 3];      d[6] = v - p[-2 * FD_PW + 2];  d[7] = v - p[-3 * FD_PW + 1];
        d[8] = v - p[-3 * FD_PW];     d[9] = v - p[-3 * FD_PW - 1];  d[10] = v - p[-2 * FD_PW - 2]; d[11] = v - p[-FD_PW - 3];
        d[12] = v - p[-3];            d[13] = v - p[FD_PW - 3];      d[14] = v - p[2 * FD_PW - 2];  d[15] = v - p[3 * FD_PW - 1];
        const int s = fast_strength(d);
        if (s > thr) sc[sj * FD_SPITCH + si] = (uint8_t)(s - 1);
    }
    __syncthreads();

    // ---- strict 3x3 maxima -> per-cell candidate lists ----
    const int cells = G.ct_cols * G.ct_rows;
#pragma unroll
    for (int r = 0; r < FD_TY / 4; r++) {
        const int ox = t & 63, oy = (t >> 6) + 4 * r;
        const uint8_t* q = &sc[(oy + 1) * FD_SPITCH + ox + 1];
        const int s = q[0];
        if (s == 0) continue;   // a score of 0 can never be a strict maximum
        if (s > q[-1] && s > q[1] && s > q[-FD_SPITCH - 1] && s > q[-FD_SPITCH] && s > q[-FD_SPITCH + 1] &&
            s > q[FD_SPITCH - 1] && s > q[FD_SPITCH] && s > q[FD_SPITCH + 1]) {
            const int gx = X0 + ox, gy = Y0 + oy;
            const int cxi = gx / G.size_x, cyi = gy / G.size_y;
            const int gcell = img_i * cells + cyi * G.ct_cols + cxi;
            const int at = atomicAdd(&cand_count[gcell], 1);
            if (at < G.cand_cap) cand[(uint64_t)gcell * G.cand_cap + at] = fast_key(s, gy - cyi * G.size_y, gx - cxi * G.size_x);
        }
    }
}

// one workgroup per (image, cell): repeated "largest key below the previous one".  Also reports the number of NMS
// survivors of the cell and leaves the cell's candidate counter at zero for the next detector pass.
__global__ void __launch_bounds__(256) fast_topk_kernel(FastGrid G, const uint32_t* __restrict__ cand, int* __restrict__ cand_count,
                                                       ft_keypoint* __restrict__ out, int* __restrict__ out_count, int* __restrict__ surv_count) {
    __shared__ uint32_t red[4];
    const int gcell = blockIdx.x;
    const int cell = gcell % (G.ct_cols * G.ct_rows);
    const int survivors = cand_count[gcell];
    const int n = min(survivors, G.cand_cap);
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
    __syncthreads();   // every thread has read the counter
    if (threadIdx.x == 0) {
        out_count[gcell] = keep;
        surv_count[gcell] = survivors;
        cand_count[gcell] = 0;
    }
}

}  // namespace ftd
