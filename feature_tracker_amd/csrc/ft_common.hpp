// Shared device-side descriptors for the gfx950 front-end kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ft_device.h"

namespace ftd {

// Where level l of a pyramid slot lives, relative to the start of that slot.  Offsets and pitches are fixed at
// context creation from the LARGEST frame the context accepts, so a kernel needs only (slot index, w0, h0).
struct LevelDesc {
    uint64_t img_off;     // bytes, u8 image
    uint64_t grad_off;    // bytes, interleaved int16 (Ix,Iy)
    uint32_t img_pitch;   // bytes per image row (multiple of 256)
    uint32_t grad_pitch;  // bytes per gradient row (multiple of 256)
};

struct ArenaDesc {
    uint8_t* base;
    uint64_t slot_bytes;
    LevelDesc lv[FT_MAX_LEVELS];
};

// images of one batched launch: linear slot index = stream * n_slots + slot
struct ImgList {
    int n;
    int idx[FT_MAX_BATCH];
};

__host__ __device__ inline int half_up(int v) { return (v + 1) >> 1; }

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A launch over a linear work list of `total`
// items uses xcd_grid(total) workgroups and workgroup b handles item xcd_item(b, total) (or nothing if that is
// >= total): every XCD then walks one contiguous eighth of the list, so tiles that share image lines (neighbours in
// the list) meet in one L2 instead of being fetched from memory once per XCD.  Only speed depends on the dealing
// order; any order gives the same result.
constexpr int XCDS = 8;
__host__ __device__ inline int xcd_share(int total) { return (total + XCDS - 1) / XCDS; }
__host__ inline int xcd_grid(int total) { return xcd_share(total) * XCDS; }
__device__ inline int xcd_item(int block, int total) { return (block % XCDS) * xcd_share(total) + block / XCDS; }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__device__ inline const uint8_t* level_img(const ArenaDesc& A, int slot_idx, int level) {
    return A.base + (uint64_t)slot_idx * A.slot_bytes + A.lv[level].img_off;
}
__device__ inline uint8_t* level_img_w(const ArenaDesc& A, int slot_idx, int level) {
    return A.base + (uint64_t)slot_idx * A.slot_bytes + A.lv[level].img_off;
}
__device__ inline const uint8_t* level_grad(const ArenaDesc& A, int slot_idx, int level) {
    return A.base + (uint64_t)slot_idx * A.slot_bytes + A.lv[level].grad_off;
}
__device__ inline uint8_t* level_grad_w(const ArenaDesc& A, int slot_idx, int level) {
    return A.base + (uint64_t)slot_idx * A.slot_bytes + A.lv[level].grad_off;
}

}  // namespace ftd
