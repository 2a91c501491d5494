// Pyramid level kernel for gfx950: reads level l once and emits (a) its Scharr gradient image and (b) level l+1.
//
// Replaces cv::pyrDown inside cv::buildOpticalFlowPyramid (reference call src/track/TrackKLT.cpp:51-56) and
// calcSharrDeriv inside cv::calcOpticalFlowPyrLK (reference calls src/track/TrackKLT.cpp:306,:402).
//
// HBM-bound streaming stencil.  One 256-thread workgroup owns a 128x32 source tile: the tile plus its halo
// (REFLECT_101 applied while loading) is staged once in LDS with aligned dword loads, then
//   - every thread writes 4x4 gradient pixels as four 16-byte stores (int16 Ix,Iy interleaved), and
//   - every thread writes 4 pixels of the half-resolution level as one dword store.
// Algorithmic bytes per source pixel: 1 (read) + 4 (gradient) + 0.25 (next level).
#pragma once
#include "ft_common.hpp"

namespace ftd {

constexpr int LV_TX = 128;           // source tile width
constexpr int LV_TY = 32;            // source tile height
constexpr int LV_LDS_PITCH = 136;    // bytes: source columns X0-4 .. X0+131
constexpr int LV_LDS_ROWS = LV_TY + 3;   // source rows Y0-2 .. Y0+32

__device__ inline uint32_t ld_src_dword(const uint8_t* __restrict__ row, int c0, int w) {
    if (c0 >= 0 && c0 + 3 < w) return *reinterpret_cast<const uint32_t*>(row + c0);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)row[reflect101(c0 + k, w)] << (8 * k);
    return v;
}

__global__ void __launch_bounds__(256) level_kernel(ArenaDesc A, ImgList L, int level, int w, int h, int do_grad, int do_down) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[LV_LDS_ROWS * LV_LDS_PITCH];
    // work list: (image, tile row, tile column), images outermost
    const int tiles_x = (w + LV_TX - 1) / LV_TX, tiles_y = (h + LV_TY - 1) / LV_TY;
    const int item = xcd_item(blockIdx.x, tiles_x * tiles_y * L.n);
    if (item >= tiles_x * tiles_y * L.n) return;
    const int img_i = item / (tiles_x * tiles_y), tile_i = item - img_i * (tiles_x * tiles_y);
    const int slot = L.idx[img_i];
    const uint8_t* __restrict__ src = level_img(A, slot, level);
    const uint32_t spitch = A.lv[level].img_pitch;
    const int X0 = (tile_i % tiles_x) * LV_TX, Y0 = (tile_i / tiles_x) * LV_TY;
    const int t = threadIdx.x;

    // ---- stage the tile: 35 rows x 34 dwords ----
    for (int i = t; i < LV_LDS_ROWS * (LV_LDS_PITCH / 4); i += 256) {
        const int rr = i / (LV_LDS_PITCH / 4), d = i - rr * (LV_LDS_PITCH / 4);
        const int sy = reflect101(Y0 - 2 + rr, h);
        const uint32_t v = ld_src_dword(src + (size_t)sy * spitch, X0 - 4 + 4 * d, w);
        *reinterpret_cast<uint32_t*>(&tile[rr * LV_LDS_PITCH + 4 * d]) = v;
    }
    __syncthreads();

    // ---- Scharr: thread -> 4 columns x 4 rows ----
    if (do_grad) {
        const int tx = t & 31, ty = t >> 5;
        const int gx0 = X0 + 4 * tx, gy0 = Y0 + 4 * ty;
        if (gx0 < w && gy0 < h) {
            uint8_t* __restrict__ gdst = level_grad_w(A, slot, level);
            const uint32_t gpitch = A.lv[level].grad_pitch;
            // source columns gx0-1 .. gx0+4 live at LDS bytes 4tx+3 .. 4tx+8 of rows (4ty+1+j), j = 0..5
            int px[6][6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const uint8_t* lr = &tile[(4 * ty + 1 + j) * LV_LDS_PITCH + 4 * tx];
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
                    *reinterpret_cast<uint4*>(drow) = make_uint4(o[0], o[1], o[2], o[3]);
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
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
                const uint8_t* lr = &tile[(2 * ty + j) * LV_LDS_PITCH + 8 * tx];
                const uint2 lo = *reinterpret_cast<const uint2*>(lr);
                const uint2 hi = *reinterpret_cast<const uint2*>(lr + 8);
                int s[12];   // source columns 2*ox0-2 .. 2*ox0+9 = LDS bytes 8tx+2 .. 8tx+13
                s[0] = (lo.x >> 16) & 0xff; s[1] = lo.x >> 24;
                s[2] = lo.y & 0xff; s[3] = (lo.y >> 8) & 0xff; s[4] = (lo.y >> 16) & 0xff; s[5] = lo.y >> 24;
                s[6] = hi.x & 0xff; s[7] = (hi.x >> 8) & 0xff; s[8] = (hi.x >> 16) & 0xff; s[9] = hi.x >> 24;
                s[10] = hi.y & 0xff; s[11] = (hi.y >> 8) & 0xff;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int hsum = s[2 * k] + 4 * s[2 * k + 1] + 6 * s[2 * k + 2] + 4 * s[2 * k + 3] + s[2 * k + 4];
                    acc[k] += wj * hsum;
                }
            }
            uint8_t* __restrict__ dst = level_img_w(A, slot, level + 1);
            const uint32_t dpitch = A.lv[level + 1].img_pitch;
            uint8_t* drow = dst + (size_t)oy * dpitch + ox0;
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= (uint32_t)((acc[k] + 128) >> 8) << (8 * k);
            if (ox0 + 3 < dw) {
                *reinterpret_cast<uint32_t*>(drow) = packed;
            } else {
                for (int k = 0; k < 4 && ox0 + k < dw; k++) drow[k] = (uint8_t)(packed >> (8 * k));
            }
        }
    }
}

// ---- batched frame ingest: n device-resident frames -> level 0 of their slots, one launch ----
constexpr int FRAME_PTRS = 128;
struct FramePtrs {
    const uint8_t* p[FRAME_PTRS];
};
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

}  // namespace ftd
