// Pyramid level kernel for gfx950: reads level l once and emits (a) its Scharr gradient image and (b) level l+1;
// in its frame-ingest form it reads the caller's frame instead and also writes level 0.
//
// Replaces cv::pyrDown inside cv::buildOpticalFlowPyramid (reference call src/track/TrackKLT.cpp:51-56), the frame
// copy before it (img.clone(), src/track/TrackKLT.cpp:46-47) and calcSharrDeriv inside cv::calcOpticalFlowPyrLK
// (reference calls src/track/TrackKLT.cpp:306,:402).
//
// One 256-thread workgroup owns a 128x32 source tile.  The tile plus its halo is staged once in LDS with 8-byte
// loads (three rows per wavefront load); the few halo columns outside the image are filled separately (REFLECT_101).
//   - pyrDown: a thread produces 4 pixels of one half-resolution row.  The 5x5 binomial filter runs on
//     v_dot4_u32_u8: the row weight is folded into the byte weights (1 4 6 4)*wr plus a fifth-tap word, so five
//     source rows cost 40 dot products and no separate vertical pass.
//   - Scharr (optional): a thread writes 4x4 gradient pixels as four 16-byte stores (int16 Ix,Iy interleaved).
// Algorithmic bytes per source pixel: 1 (read) + 0.25 (next level) [+ 4 (gradient)] [+ 1 (level 0, ingest form)].
#pragma once
#include "ft_common.hpp"

namespace ftd {

constexpr int LV_TX = 128;           // source tile width
constexpr int LV_TY = 32;            // source tile height
constexpr int LV_LDS_PITCH = 144;    // bytes: source columns X0-8 .. X0+135
constexpr int LV_LDS_ROWS = 36;      // source rows Y0-2 .. Y0+33 (the last one is only loaded, never used)
constexpr int LV_GROUPS = LV_LDS_PITCH / 8;   // 8-byte groups per staged row

constexpr int FRAME_PTRS = 128;
struct FramePtrs {
    const uint8_t* p[FRAME_PTRS];
};
struct __attribute__((packed)) U2Unaligned { uint32_t a, b; };

// FROM_FRAME: the source is frame F.p[i] (row stride sstride, any alignment) and the tile is also written to level 0
// one tile (work item `item` of the list (image, tile row, tile column), images outermost) on one 256-thread workgroup;
// tile: LV_LDS_ROWS * LV_LDS_PITCH bytes of LDS, 16-byte aligned
template <bool FROM_FRAME>
__device__ __forceinline__ void level_tile(const ArenaDesc& A, const ImgList& L, const FramePtrs& F, int first, int level, int w, int h, uint64_t sstride,
                                           int do_grad, int do_down, int item, int t, uint8_t* __restrict__ tile) {
    const int tiles_x = (w + LV_TX - 1) / LV_TX, tiles_y = (h + LV_TY - 1) / LV_TY;
    const int img_i = item / (tiles_x * tiles_y), tile_i = item - img_i * (tiles_x * tiles_y);
    const int slot = L.idx[first + img_i];
    const uint8_t* __restrict__ src = FROM_FRAME ? F.p[img_i] : level_img(A, slot, level);
    const uint64_t spitch = FROM_FRAME ? sstride : (uint64_t)A.lv[level].img_pitch;
    const int X0 = (tile_i % tiles_x) * LV_TX, Y0 = (tile_i / tiles_x) * LV_TY;

    // ---- stage the tile: a wavefront load covers 3 rows x 18 groups of 8 bytes; 3 rounds x 4 waves = 36 rows.
    //      Only groups that lie completely inside the image row are loaded here. ----
    {
        const int lane = t & 63, wv = t >> 6;
        const int rg = lane / LV_GROUPS, c8 = lane - rg * LV_GROUPS;
        const int gx = X0 - 8 + 8 * c8;
        const bool act = lane < 3 * LV_GROUPS && gx >= 0 && gx + 7 < w;
        uint2 v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int row = wv * 3 + rg + 12 * k;
            const int sy = reflect101(Y0 - 2 + min(row, LV_LDS_ROWS - 1), h);
            v[k] = make_uint2(0u, 0u);
            if (act) {
                const U2Unaligned u = *reinterpret_cast<const U2Unaligned*>(src + (uint64_t)sy * spitch + gx);
                v[k] = make_uint2(u.a, u.b);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int row = wv * 3 + rg + 12 * k;
            if (lane < 3 * LV_GROUPS) *reinterpret_cast<uint2*>(&tile[row * LV_LDS_PITCH + 8 * c8]) = v[k];
        }
    }
    // columns the filters need that the groups above did not bring: left of the image (first tile column) and from
    // the first group that crosses the right border on (last tile column); byte by byte with REFLECT_101
    {
        const int need_lo = X0 - 2, need_hi = min(X0 + LV_TX + 1, w + 1);                 // filter footprint of this tile
        const int first_bad = w < 8 ? X0 - 8 : (X0 - 8) + 8 * (((w - 8) - (X0 - 8)) / 8 + 1);   // first group start with gx + 7 >= w
        const int n_left = X0 == 0 ? 2 : 0;
        const int r_lo = max(first_bad, need_lo), n_right = max(need_hi - r_lo + 1, 0);
        const int n_irr = n_left + n_right;
        if (n_irr > 0) {   // tile-uniform
            __syncthreads();   // the zero fill of unloaded groups above must not overtake these bytes
            for (int i = t; i < LV_LDS_ROWS * n_irr; i += 256) {
                const int row = i / n_irr, j = i - row * n_irr;
                const int x = j < n_left ? need_lo + j : r_lo + (j - n_left);
                const int sy = reflect101(Y0 - 2 + row, h);
                tile[row * LV_LDS_PITCH + (x - (X0 - 8))] = src[(uint64_t)sy * spitch + reflect101(x, w)];
            }
        }
    }
    __syncthreads();

    // ---- ingest form: the tile interior is level 0 ----
    if (FROM_FRAME) {
        const int row = t >> 3, seg = t & 7;
        const int gx = X0 + 16 * seg, gy = Y0 + row;
        if (gx < w && gy < h) {
            const uint8_t* lr = &tile[(row + 2) * LV_LDS_PITCH + 8 + 16 * seg];
            const uint2 lo = *reinterpret_cast<const uint2*>(lr), hi = *reinterpret_cast<const uint2*>(lr + 8);
            uint8_t* drow = level_img_w(A, slot, 0) + (size_t)gy * A.lv[0].img_pitch + gx;
            if (gx + 16 <= w) {
                *reinterpret_cast<uint4*>(drow) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                for (int k = 0; gx + k < w; k++) drow[k] = lr[k];
            }
        }
    }

    // ---- Scharr: thread -> 4 columns x 4 rows ----
    if (do_grad) {
        const int tx = t & 31, ty = t >> 5;
        const int gx0 = X0 + 4 * tx, gy0 = Y0 + 4 * ty;
        if (gx0 < w && gy0 < h) {
            uint8_t* __restrict__ gdst = level_grad_w(A, slot, level);
            const uint32_t gpitch = A.lv[level].grad_pitch;
            // source columns gx0-1 .. gx0+4 live at LDS bytes 4tx+7 .. 4tx+12 of rows (4ty+1+j), j = 0..5
            int px[6][6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const uint8_t* lr = &tile[(4 * ty + 1 + j) * LV_LDS_PITCH + 4 * tx + 4];
                const uint32_t a = *reinterpret_cast<const uint32_t*>(lr);
                const uint32_t b = *reinterpret_cast<const uint32_t*>(lr + 4);
                const uint32_t c = *reinterpret_cast<const uint32_t*>(lr + 8);
                px[j][0] = a >> 24;
                px[j][1] = b & 0xff; px[j][2] = (b >> 8) & 0xff; px[j][3] = (b >> 16) & 0xff; px[j][4] = b >> 24;
                px[j][5] = c & 0xff;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gy = gy0 + j;
                if (gy >= h) break;
                int t0[6], t1[6];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    t0[c] = (px[j][c] + px[j + 2][c]) * 3 + px[j + 1][c] * 10;
                    t1[c] = px[j + 2][c] - px[j][c];
                }
                uint32_t o[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int ix = t0[c + 2] - t0[c];
                    const int iy = (t1[c + 2] + t1[c]) * 3 + t1[c + 1] * 10;
                    o[c] = ((uint32_t)ix & 0xffffu) | ((uint32_t)iy << 16);
                }
                uint8_t* drow = gdst + (size_t)gy * gpitch + (size_t)gx0 * 4;
                if (gx0 + 3 < w) {
                    // written once and not read again by this launch or the next ones of a frame set: a streaming store
                    // keeps the 4 bytes per pixel out of the caches (64 x 1280x720: 4.3 -> 6.2 TB/s; level 0 written
                    // that way instead made the kernels that read it next slower)
                    typedef uint32_t lv_u4 __attribute__((ext_vector_type(4)));
                    lv_u4 v4 = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(v4, reinterpret_cast<lv_u4*>(drow));
                } else {
                    for (int c = 0; c < 4 && gx0 + c < w; c++) reinterpret_cast<uint32_t*>(drow)[c] = o[c];
                }
            }
        }
    }

    // ---- pyrDown: thread -> 4 output columns of one output row ----
    if (do_down) {
        const int dw = half_up(w), dh = half_up(h);
        const int tx = t & 15, ty = t >> 4;
        const int ox0 = (X0 >> 1) + 4 * tx, oy = (Y0 >> 1) + ty;
        if (ox0 < dw && oy < dh) {
            uint32_t acc[4] = {128u, 128u, 128u, 128u};   // rounding term of the final >> 8
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint32_t wj = (j == 0 || j == 4) ? 1u : ((j == 2) ? 6u : 4u);
                const uint32_t w4 = wj * 0x04060401u;     // taps 1 4 6 4 of this row, one per byte
                const uint32_t w5lo = wj, w5mid = wj << 16;   // the fifth tap as byte 0 / byte 2 of the following word
                // source columns 2*ox0-2 .. 2*ox0+9 are tile bytes 8tx+6 .. 8tx+17: words d0..d3 from byte 8tx+4
                const uint8_t* lr = &tile[(2 * ty + j) * LV_LDS_PITCH + 8 * tx + 4];
                const uint32_t d0 = *reinterpret_cast<const uint32_t*>(lr);
                const uint2 d12 = *reinterpret_cast<const uint2*>(lr + 4);
                const uint32_t d3 = *reinterpret_cast<const uint32_t*>(lr + 12);
                const uint32_t a0 = __builtin_amdgcn_alignbyte(d12.x, d0, 2), a2 = __builtin_amdgcn_alignbyte(d12.y, d12.x, 2);
                acc[0] = __builtin_amdgcn_udot4(d12.x, w5mid, __builtin_amdgcn_udot4(a0, w4, acc[0], false), false);
                acc[1] = __builtin_amdgcn_udot4(d12.y, w5lo, __builtin_amdgcn_udot4(d12.x, w4, acc[1], false), false);
                acc[2] = __builtin_amdgcn_udot4(d12.y, w5mid, __builtin_amdgcn_udot4(a2, w4, acc[2], false), false);
                acc[3] = __builtin_amdgcn_udot4(d3, w5lo, __builtin_amdgcn_udot4(d12.y, w4, acc[3], false), false);
            }
            uint8_t* __restrict__ dst = level_img_w(A, slot, level + 1);
            const uint32_t dpitch = A.lv[level + 1].img_pitch;
            uint8_t* drow = dst + (size_t)oy * dpitch + ox0;
            const uint32_t packed = (acc[0] >> 8) | ((acc[1] >> 8) << 8) | ((acc[2] >> 8) << 16) | ((acc[3] >> 8) << 24);
            if (ox0 + 3 < dw) {
                *reinterpret_cast<uint32_t*>(drow) = packed;
            } else {
                for (int k = 0; k < 4 && ox0 + k < dw; k++) drow[k] = (uint8_t)(packed >> (8 * k));
            }
        }
    }
}

// GRAD = false: the form without gradient images, as the tracking pipeline launches it (do_grad is ignored).  It is a
// kernel of its own for its registers: the Scharr step keeps a 6 x 6 pixel block per thread, and a kernel that only might
// run it is allocated for it; without it a wavefront takes few enough registers for several of them to find room on a
// SIMD that four tracking wavefronts of another queue have filled but for 64 registers.
template <bool FROM_FRAME, bool GRAD = true>
__global__ void __launch_bounds__(256) level_kernel(ArenaDesc A, ImgList L, FramePtrs F, int first, int level, int w, int h, uint64_t sstride,
                                                    int do_grad, int do_down) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[LV_LDS_ROWS * LV_LDS_PITCH];
    const int tiles = ((w + LV_TX - 1) / LV_TX) * ((h + LV_TY - 1) / LV_TY);
    const int n_img = FROM_FRAME ? min(FRAME_PTRS, L.n - first) : L.n;
    const int item = xcd_item(blockIdx.x, tiles * n_img);
    if (item >= tiles * n_img) return;
    level_tile<FROM_FRAME>(A, L, F, first, level, w, h, sstride, GRAD ? do_grad : 0, do_down, item, threadIdx.x, tile);
}

// ---- batched frame ingest: n device-resident frames -> level 0 of their slots, one launch ----
struct __attribute__((packed)) U4Unaligned { uint32_t a, b, c, d; };

__global__ void __launch_bounds__(256) ingest_kernel(ArenaDesc A, ImgList L, FramePtrs F, int first, int w, int h, uint64_t sstride) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 16;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* __restrict__ src = F.p[blockIdx.z] + (uint64_t)y * sstride + x;
    uint8_t* __restrict__ dst = level_img_w(A, L.idx[first + blockIdx.z], 0) + (size_t)y * A.lv[0].img_pitch + x;
    if (x + 16 <= w) {
        const U4Unaligned v = *reinterpret_cast<const U4Unaligned*>(src);   // source rows may have any alignment
        *reinterpret_cast<uint4*>(dst) = make_uint4(v.a, v.b, v.c, v.d);
    } else {
        for (int k = 0; x + k < w; k++) dst[k] = src[k];
    }
}

// Staged frames (pinned host memory read over the bus: tightly packed rows, w a multiple of 16) -> level 0.  Every wave
// reads 1 KB of consecutive bytes: in that shape a kernel reads host memory at the rate of a DMA copy (56 GB/s measured),
// while the 144-byte row pieces of a tile-shaped pass reach 25-35 GB/s.  grid = (chunks of 256 x 16 bytes, image).
__global__ void __launch_bounds__(256) ingest_linear_kernel(ArenaDesc A, ImgList L, FramePtrs F, int first, int chunks_per_row, int n16) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n16) return;
    const int row = g / chunks_per_row, c = g - row * chunks_per_row;
    const uint4 v = reinterpret_cast<const uint4*>(F.p[blockIdx.y])[g];
    *reinterpret_cast<uint4*>(level_img_w(A, L.idx[first + blockIdx.y], 0) + (size_t)row * A.lv[0].img_pitch + 16 * c) = v;
}

}  // namespace ftd
