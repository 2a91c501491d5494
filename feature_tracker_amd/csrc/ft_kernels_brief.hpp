// 256-bit binary descriptors at given keypoints, for gfx950: the extraction step of the descriptor tracker
// (reference: orb0->compute(img0, pts0_ext, desc0_ext), src/track/TrackDescriptor.cpp:218-219).
//
// cv::ORB::compute compares smoothed intensities at 256 point pairs taken from a table inside OpenCV, rotated by the
// keypoint's intensity-centroid angle.  That table is not available here, so the descriptor is this build's own,
// fully specified BRIEF-style one with ORB's geometry (parity unpinned):
//   - the image is smoothed with the 5x5 binomial kernel [1 4 6 4 1]^2, (sum + 128) >> 8 (the pyrDown kernel without the
//     decimation; ORB uses a 7x7 Gaussian);
//   - test i compares the smoothed values at two offsets (x1, y1), (x2, y2) in [-15, 15]^2 around the keypoint (ORB's
//     31x31 patch), offset coordinate k of test i = mix32(0xB5297A4D + 4 i + k) mod 31 - 15; bit i = S(p1) < S(p2),
//     stored in byte i / 8 at bit i % 8; no rotation;
//   - a keypoint closer than 17 pixels to the image border (patch radius 15 + smoothing radius 2) gets no descriptor
//     (valid = 0, 32 zero bytes), as ORB drops keypoints whose patch leaves the image.
//
// One wavefront per keypoint: the 35x35 pixel patch goes to LDS, the 31x31 smoothed patch is computed there in two
// separable passes, then every lane evaluates four tests and the bits are gathered with ballots.
#pragma once
#include "ft_common.hpp"

namespace ftd {

constexpr int BRIEF_R = 15;                       // patch radius
constexpr int BRIEF_P = 2 * BRIEF_R + 1;          // 31
constexpr int BRIEF_PP = BRIEF_P + 4;             // 35: with the smoothing apron
constexpr int BRIEF_BORDER = BRIEF_R + 2;         // 17

__host__ __device__ inline uint32_t brief_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// offset coordinate k (0: x1, 1: y1, 2: x2, 3: y2) of test i
__host__ __device__ inline int brief_offset(int i, int k) { return (int)(brief_mix32(0xB5297A4Du + 4u * (uint32_t)i + (uint32_t)k) % 31u) - BRIEF_R; }

__global__ void __launch_bounds__(64) brief_kernel(ArenaDesc A, int slot, int w, int h, const float* __restrict__ xy, int n,
                                                   uint8_t* __restrict__ desc, uint8_t* __restrict__ valid) {
    __shared__ uint8_t pix[BRIEF_PP * BRIEF_PP];          // rows y-17 .. y+17, columns x-17 .. x+17
    __shared__ uint16_t hsum[BRIEF_PP * BRIEF_P];         // horizontal pass: 35 rows x 31 columns, values <= 16 * 255
    __shared__ uint8_t smooth[BRIEF_P * BRIEF_P];
    const int kp = blockIdx.x, lane = threadIdx.x;
    if (kp >= n) return;
    const int cx = (int)xy[2 * kp], cy = (int)xy[2 * kp + 1];   // detector output: whole pixels
    uint32_t* out = reinterpret_cast<uint32_t*>(desc + (size_t)kp * 32);
    if (cx < BRIEF_BORDER || cy < BRIEF_BORDER || cx >= w - BRIEF_BORDER || cy >= h - BRIEF_BORDER) {   // uniform for the wave
        if (lane < 8) out[lane] = 0u;
        if (lane == 0) valid[kp] = 0;
        return;
    }
    const uint8_t* img = level_img(A, slot, 0);
    const uint32_t pitch = A.lv[0].img_pitch;
    for (int i = lane; i < BRIEF_PP * BRIEF_PP; i += 64) {
        const int r = i / BRIEF_PP, c = i - r * BRIEF_PP;
        pix[i] = img[(uint32_t)(cy - BRIEF_BORDER + r) * pitch + (uint32_t)(cx - BRIEF_BORDER + c)];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < BRIEF_PP * BRIEF_P; i += 64) {
        const int r = i / BRIEF_P, c = i - r * BRIEF_P;
        const uint8_t* p = &pix[r * BRIEF_PP + c];
        hsum[i] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < BRIEF_P * BRIEF_P; i += 64) {
        const int r = i / BRIEF_P, c = i - r * BRIEF_P;
        const uint16_t* p = &hsum[r * BRIEF_P + c];
        const int s = p[0] + 4 * p[BRIEF_P] + 6 * p[2 * BRIEF_P] + 4 * p[3 * BRIEF_P] + p[4 * BRIEF_P];
        smooth[i] = (uint8_t)((s + 128) >> 8);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // test i = 64 * q + lane: the ballot of round q holds bits 64 q .. 64 q + 63, i.e. descriptor words 2 q and 2 q + 1
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = 64 * q + lane;
        const int a = smooth[(brief_offset(i, 1) + BRIEF_R) * BRIEF_P + brief_offset(i, 0) + BRIEF_R];
        const int b = smooth[(brief_offset(i, 3) + BRIEF_R) * BRIEF_P + brief_offset(i, 2) + BRIEF_R];
        const uint64_t bits = __ballot(a < b);
        if (lane == 0) { out[2 * q] = (uint32_t)bits; out[2 * q + 1] = (uint32_t)(bits >> 32); }
    }
    if (lane == 0) valid[kp] = 1;
}

}  // namespace ftd
