// Brute-force Hamming 2-nearest-neighbour search on 256-bit binary descriptors for gfx950: the device part of
// TrackDescriptor::robust_match (reference: src/track/TrackDescriptor.cpp:249-265), i.e. of
// cv::BFMatcher(cv::NORM_HAMMING).knnMatch(query, train, matches, 2).
//
// One wavefront per query descriptor, four queries per workgroup.  A lane walks the train set with stride 64 and
// keeps its two best candidates as keys (distance << 20 | train index): a smaller key is a closer descriptor and,
// at equal distance, the earlier one.  Two wave-wide minima give the query's two neighbours.  The train set is a
// few tens of kilobytes and stays in the caches; the work is popcounts, so the kernel is bound by vector issue.
#pragma once
#include "ft_common.hpp"

namespace ftd {

constexpr uint32_t MATCH_NONE = 0xffffffffu;   // "no neighbour" key
constexpr int MATCH_IDX_BITS = 20;             // train sets of up to 2^20 descriptors; distances need 9 bits

__device__ __forceinline__ uint32_t match_wave_min(uint32_t v) {
    const int none = (int)MATCH_NONE;
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast15 into rows 1,3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(none, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast31 into rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// best[q] = (key of the nearest train descriptor, key of the second nearest), MATCH_NONE where there is none
__global__ void __launch_bounds__(256) hamming_knn2_kernel(const uint4* __restrict__ query, int n_query, const uint4* __restrict__ train,
                                                          int n_train, uint2* __restrict__ best) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n_query) return;   // uniform for the wave
    const uint4 a0 = query[2 * q], a1 = query[2 * q + 1];
    uint32_t k1 = MATCH_NONE, k2 = MATCH_NONE;
    for (int j = lane; j < n_train; j += 64) {
        const uint4 b0 = train[2 * j], b1 = train[2 * j + 1];
        const uint32_t d = __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) + __builtin_popcount(a0.z ^ b0.z) +
                           __builtin_popcount(a0.w ^ b0.w) + __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
                           __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
        const uint32_t key = (d << MATCH_IDX_BITS) | (uint32_t)j;
        k2 = min(k2, max(k1, key));   // the larger of (old best, new) competes for second place
        k1 = min(k1, key);
    }
    const uint32_t m1 = match_wave_min(k1);
    const uint32_t m2 = match_wave_min(k1 == m1 ? k2 : k1);   // keys are unique: the lane that owns m1 offers its runner-up
    if (lane == 0) best[q] = make_uint2(m1, m2);
}

}  // namespace ftd
