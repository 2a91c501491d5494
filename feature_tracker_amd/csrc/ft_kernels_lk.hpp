// Pyramidal Lucas-Kanade for gfx950: one 64-lane wavefront per feature, all pyramid levels in one launch.
//
// Replaces cv::calcOpticalFlowPyrLK / LKTrackerInvoker as called at src/track/TrackKLT.cpp:306 and :402.
//
// Lane layout: the win_w x win_h window is cut into row segments of SEGLEN pixels; lane = row * SEGS + segment.
//  * Both image regions a level needs are staged in LDS with aligned dword loads issued together at the start of
//    the level (REFLECT_101 applied while staging): the template region of I (window + 1 pixel of Scharr apron) and
//    the part of J the window can reach (window + a margin of LK_MARGIN pixels on every side).
//  * Template: each lane samples its SEGLEN pixels of I (x32) and evaluates the Scharr gradients of its samples from
//    the staged I region (same arithmetic as the gradient image kernel, so no gradient image has to be read: a
//    window touches a few bytes of every 128-byte line it crosses, and the 4-byte-per-pixel gradient lines were the
//    largest part of this kernel's memory traffic).  The kernel is bound by vector-instruction issue, so all of
//    this runs on packed 16-bit pairs: v_pk_* for the Scharr taps (|values| <= 4080), v_dot2_i32_i16 for every
//    bilinear blend and for the normal-equation sums.  Samples stay in registers for the whole level.
//  * Target: every Newton iteration reads two LDS rows per lane, so the dependent chain position -> pixels ->
//    (b1,b2) -> position never waits on memory.  A window that walks out of the staged region re-stages it.
//  * Sums are exact integers: lanes accumulate in int32, the wave reduction runs on bit fields narrow enough not
//    to overflow, with DPP adds (no LDS traffic), and the total is rounded to float once, so results do not depend
//    on the reduction order.
#pragma once
#include "ft_common.hpp"

namespace ftd {

struct LKJobDev {
    int img_prev, img_next;   // linear slot indices
    int w0, h0;               // level-0 size (both images)
    int use_H;                // 1: initial guess = pi(H * [prev;1])
    float H[9];
    int first, count;         // this job's points in the batch arrays
    double calib[16];         // fx fy cx cy k1 k2 p1 p2 of the template camera, then of the target camera (outlier rejection, stereo)
    double stereo[12];        // R_cam0_cam1 (row-major), t_cam0_cam1
    int use_stereo;           // 1: stereo match (seed from the extrinsic rotation, epipolar check after tracking)
    int pad_;
};

struct LKArgs {
    ArenaDesc A;
    const LKJobDev* jobs;
    const uint16_t* pt_job;   // job index of every point
    const float* prev_xy;     // template points (forward) / original points (backward)
    const float* guess_xy;    // forward: initial guesses (unused for jobs with a homography)
    float* next_xy;           // forward: results; backward: template points (read only)
    uint8_t* status;
    float* err;
    int* iters;               // optional [n][nlev]
    int total;                // number of points
    int nlev;                 // levels per iters row
    int win_w, win_h, segs;
    int segs_inv;             // ceil(65536 / segs): lane / segs == (lane * segs_inv) >> 16 for lane < 64
    int max_level;            // already clamped to the pyramids
    int max_iter;
    int flags;
    int backward;             // 0: forward pass, 1: forward-backward verification pass
    // When the input block above is read in place from pinned host memory and a kernel that also needs the job records
    // and the template points runs behind this one (the outlier rejection), the tracking wavefronts leave device copies:
    // wavefront pt copies its template point and the job records pt, pt + total, ...  Null: no copies.
    LKJobDev* echo_jobs;
    float* echo_prev_xy;
    int n_jobs;
    float fb_thresh2;
    double epsilon;           // squared
    double min_eig;
};

constexpr int LK_MARGIN = 5;                                     // staged margin around the target window, pixels
constexpr int LK_RW_DWORDS = (32 + 1 + 2 * LK_MARGIN + 3 + 3) / 4;   // widest staged row: 12 dwords
// LDS row pitch in dwords, per segment length.  Lane (row r, segment s) reads dwords pitch*r + (SEGLEN/4)*s + q of a
// region, 32 lanes per bank group.  4-pixel segments (15x15: four lanes per row) are conflict-free at the row width
// itself.  8-pixel segments (21x21: three lanes per row, two dwords apart) at pitch 12 touch only even banks and
// only 8 distinct row offsets, three addresses per bank; no pitch makes 32 such lanes conflict-free (ten rows and
// two lanes do not tile 32 banks), 19 leaves one bank group clean and the other with a single two-way conflict.
template <int SEGLEN> struct LkPitch { static constexpr int value = SEGLEN == 4 ? 12 : (SEGLEN == 8 ? 19 : 13); };

typedef short lk_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lk_v2s v2s(uint32_t x) { return __builtin_bit_cast(lk_v2s, x); }
__device__ __forceinline__ uint32_t u32(lk_v2s x) { return __builtin_bit_cast(uint32_t, x); }

// ---- wave-wide exact integer sum (result uniform in all lanes) ----
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

// three wave-wide sums at once, stage by stage: the chains are independent, so the data-hazard wait states of one
// DPP add are filled by the other two
__device__ __forceinline__ void wave_sum3_i32(int& a, int& b, int& c) {
#define LK_DPP_STAGE(ctrl, rmask)                                                     \
    {                                                                                  \
        const int ta = __builtin_amdgcn_update_dpp(0, a, ctrl, rmask, 0xF, false);     \
        const int tb = __builtin_amdgcn_update_dpp(0, b, ctrl, rmask, 0xF, false);     \
        const int tc = __builtin_amdgcn_update_dpp(0, c, ctrl, rmask, 0xF, false);     \
        a += ta; b += tb; c += tc;                                                     \
    }
    LK_DPP_STAGE(0xB1, 0xF)    // quad_perm [1,0,3,2]
    LK_DPP_STAGE(0x4E, 0xF)    // quad_perm [2,3,0,1]
    LK_DPP_STAGE(0x141, 0xF)   // row_half_mirror
    LK_DPP_STAGE(0x140, 0xF)   // row_mirror
    LK_DPP_STAGE(0x142, 0xA)   // row_bcast15 into rows 1,3
    LK_DPP_STAGE(0x143, 0xC)   // row_bcast31 into rows 2,3
#undef LK_DPP_STAGE
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63);
}

// the same for two chains
__device__ __forceinline__ void wave_sum2_i32(int& a, int& b) {
#define LK_DPP_STAGE2(ctrl, rmask)                                                    \
    {                                                                                  \
        const int ta = __builtin_amdgcn_update_dpp(0, a, ctrl, rmask, 0xF, false);     \
        const int tb = __builtin_amdgcn_update_dpp(0, b, ctrl, rmask, 0xF, false);     \
        a += ta; b += tb;                                                              \
    }
    LK_DPP_STAGE2(0xB1, 0xF)
    LK_DPP_STAGE2(0x4E, 0xF)
    LK_DPP_STAGE2(0x141, 0xF)
    LK_DPP_STAGE2(0x140, 0xF)
    LK_DPP_STAGE2(0x142, 0xA)
    LK_DPP_STAGE2(0x143, 0xC)
#undef LK_DPP_STAGE2
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
}

// sums over the wave of two int32 values per lane with |v| < 2^30, exactly, as floats rounded once.
// Usual case, decided for the whole wave: every |v| < 2^25, so the 64-lane sums fit an int32 and two plain chains
// do, with one int -> float conversion each.  Otherwise the low 10 bits of both travel in one register (two 16-bit
// fields: 64 * 1023 < 2^16) and the high parts in one register each.
__device__ __forceinline__ bool lk_any_lane_big(uint32_t biased_or) {   // biased_or: OR of (v + 2^25) over the lane's values
    return __builtin_amdgcn_ballot_w64((biased_or >> 26) != 0u) != 0ull;
}
__device__ __forceinline__ void wave_sum_exact2(int v1, int v2, float& s1, float& s2) {
    if (!lk_any_lane_big((uint32_t)(v1 + (1 << 25)) | (uint32_t)(v2 + (1 << 25)))) {
        wave_sum2_i32(v1, v2);
        s1 = (float)v1; s2 = (float)v2;
        return;
    }
    int slo = (v1 & 1023) | ((v2 & 1023) << 16), sh1 = v1 >> 10, sh2 = v2 >> 10;
    wave_sum3_i32(slo, sh1, sh2);
    s1 = (float)((double)sh1 * 1024.0 + (double)(slo & 0xffff));
    s2 = (float)((double)sh2 * 1024.0 + (double)((uint32_t)slo >> 16));
}
// the same for three values: two registers of low parts, three of high parts
__device__ __forceinline__ void wave_sum_exact3(int v1, int v2, int v3, float& s1, float& s2, float& s3) {
    if (!lk_any_lane_big((uint32_t)(v1 + (1 << 25)) | (uint32_t)(v2 + (1 << 25)) | (uint32_t)(v3 + (1 << 25)))) {
        wave_sum3_i32(v1, v2, v3);
        s1 = (float)v1; s2 = (float)v2; s3 = (float)v3;
        return;
    }
    int slo = (v1 & 1023) | ((v2 & 1023) << 16), sh1 = v1 >> 10, sh2 = v2 >> 10;
    int slo3 = v3 & 1023, sh3 = v3 >> 10;
    wave_sum3_i32(slo, sh1, sh2);
    wave_sum2_i32(slo3, sh3);
    s1 = (float)((double)sh1 * 1024.0 + (double)(slo & 0xffff));
    s2 = (float)((double)sh2 * 1024.0 + (double)((uint32_t)slo >> 16));
    s3 = (float)((double)sh3 * 1024.0 + (double)slo3);
}

struct BilinW { int w00, w01, w10, w11; };

__device__ __forceinline__ BilinW bilin_weights(float a, float b) {
    BilinW r;
    r.w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * 16384.f);
    r.w01 = (int)__builtin_rintf(a * (1.f - b) * 16384.f);
    r.w10 = (int)__builtin_rintf((1.f - a) * b * 16384.f);
    r.w11 = 16384 - r.w00 - r.w01 - r.w10;
    return r;
}

__device__ __forceinline__ int dot2_acc(uint32_t pair, uint32_t wpair, int acc) {
    return __builtin_amdgcn_sdot2(v2s(pair), v2s(wpair), acc, false);
}
// The same product with the sum going to a new register (three-operand form).  The builtin above becomes the
// accumulate-in-place form, which costs a register copy whenever the addend has to survive, as the template terms kept
// for all iterations of a level do.  Not to be used for a value that a DPP instruction reads next (the compiler
// cannot see through the assembly to place the wait states that needs).
__device__ __forceinline__ int dot2_to(uint32_t pair, uint32_t wpair, int addend) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(pair), "v"(wpair), "v"(addend));
    return r;
}
// the same with a wave-uniform addend taken from a scalar register or an inline constant: no vector register has to
// be set up with the constant first
__device__ __forceinline__ int dot2_to_uniform(uint32_t pair, uint32_t wpair, int addend) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(pair), "v"(wpair), "s"(addend));
    return r;
}
__device__ __forceinline__ int dot2_to_zero(uint32_t pair, uint32_t wpair) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "v"(wpair));
    return r;
}
// upper 16-bit field of lo followed by the lower field of hi: the pair one element further along
__device__ __forceinline__ uint32_t next_pair(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbyte(hi, lo, 2); }
// low halves of two ints as one 16-bit pair
__device__ __forceinline__ uint32_t pack_lo16(int second, int first) {
    return __builtin_amdgcn_perm((uint32_t)second, (uint32_t)first, 0x05040100u);
}

// integer part of a position that is the same in all lanes, in a scalar register.  (The empty statement hides where
// the value comes from: where the compiler can prove it uniform it drops the readfirstlane and then does the bounds
// checks that follow with vector compares on the vector copy.)
__device__ __forceinline__ int lk_uniform_floor(float v) {
    int vi = (int)floorf(v);
    asm("" : "+v"(vi));
    return __builtin_amdgcn_readfirstlane(vi);
}

// ---- image regions in LDS ----
struct Region {
    int rx0, ry0;     // image coordinates of the region origin (rx0 is a multiple of 4)
    int rwd, rh;      // size in dwords x rows
};

// single-reflection REFLECT_101: exact for -len < p < 2*len - 1
__device__ __forceinline__ int reflect101_once(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

struct __attribute__((packed)) LkU32Unaligned { uint32_t v; };

// The interior case of staging in two halves, so that a caller can have the loads of several regions in flight before
// the first store waits for its data.  A staged row is 48 bytes: lane (q, r) = (lane % 3, lane / 3) moves the 16
// bytes q of rows r, r + 21, ...
__device__ __forceinline__ Region region_of(int x_first, int x_last, int y_first, int n_rows) {
    Region R;
    R.rx0 = (x_first >> 2) << 2;
    R.ry0 = y_first;
    R.rwd = (x_last - R.rx0 + 4) >> 2;     // <= LK_RW_DWORDS
    R.rh = n_rows;
    return R;
}
// all staged columns are image columns / all staged rows are image rows
__device__ __forceinline__ bool region_cols_inside(const Region& R, int w) { return R.rx0 >= 0 && R.rx0 + 4 * LK_RW_DWORDS <= w; }
__device__ __forceinline__ bool region_rows_inside(const Region& R, int h) { return R.ry0 >= 0 && R.ry0 + R.rh <= h; }
struct __attribute__((packed, aligned(4))) LkU128Unaligned { uint32_t a, b, c, d; };
static_assert(sizeof(LKJobDev) % 8 == 0 && sizeof(LKJobDev) / 8 <= 64, "a job record is copied by one wavefront, eight bytes per lane");
static_assert(LK_RW_DWORDS == 12, "three 16-byte pieces per staged row");
template <int MAXR> struct LkInteriorRows { LkU128Unaligned v[(4 * MAXR + 20) / 21]; };
// Columns inside the image.  ROWS_INSIDE: so are the rows; otherwise rows are mirrored (REFLECT_101, one reflection:
// the caller checks that the level is tall enough for that).
template <int MAXR, bool ROWS_INSIDE>
__device__ __forceinline__ void load_interior(LkInteriorRows<MAXR>& rows, const uint8_t* __restrict__ img, uint32_t pitch, int h, const Region& R, int lane) {
    const int r = (lane * 21846) >> 16, q = lane - 3 * r;      // lane / 3, lane % 3 (exact for lane < 64)
    const uint32_t xoff = (uint32_t)(R.rx0 + 16 * q);
#pragma unroll
    for (int k = 0; k < (4 * MAXR + 20) / 21; k++) {
        int sy = R.ry0 + r + 21 * k;
        if (ROWS_INSIDE) {
            sy = min(sy, h - 1);                               // rows past the region (not stored) stay inside the image
        } else {
            sy = max(sy, -sy);
            sy = max(min(sy, 2 * h - 2 - sy), 0);              // the mirror row; rows past the region: any image row
        }
        rows.v[k] = *reinterpret_cast<const LkU128Unaligned*>(img + (__umul24(sy, pitch) + xoff));
    }
}
// The LDS array holds 4*MAXR rows and one more that takes the rows past them, so that no store needs a predicate.
template <int MAXR, int LK_PITCH>
__device__ __forceinline__ void store_interior(uint32_t* __restrict__ lds, const LkInteriorRows<MAXR>& rows, int lane) {
    const int r = (lane * 21846) >> 16, q = lane - 3 * r;
#pragma unroll
    for (int k = 0; k < (4 * MAXR + 20) / 21; k++) {
        uint32_t* d = lds + min(r + 21 * k, 4 * MAXR) * LK_PITCH + 4 * q;
        d[0] = rows.v[k].a; d[1] = rows.v[k].b; d[2] = rows.v[k].c; d[3] = rows.v[k].d;
    }
}

// load image columns [x_first, x_last] x rows [y_first, y_first + n_rows) into LDS (REFLECT_101 outside the image).
// Border cases: lane (c, rr) = (lane & 15, lane >> 4) owns dword column c of rows rr, rr+4, ...; the LDS array holds
// 4*MAXR rows and one surplus row.  A lane issues all its loads before the first one is consumed, so a region costs
// one memory round trip.  The caller orders the LDS writes before its reads with lds_fence().
template <int MAXR, int LK_PITCH>
__device__ __forceinline__ Region stage_region(uint32_t* __restrict__ lds, const uint8_t* __restrict__ img, uint32_t pitch, int w, int h,
                                               int x_first, int x_last, int y_first, int n_rows, int lane, int a_reach_w, int a_reach_h) {
    Region R = region_of(x_first, x_last, y_first, n_rows);     // rwd <= LK_RW_DWORDS, rh <= 4 * MAXR
    const int c = lane & 15, rr = lane >> 4;
    const int gx = R.rx0 + 4 * c;
    // wave-uniform cases, cheapest first
    if (region_cols_inside(R, w) && (region_rows_inside(R, h) || h >= a_reach_h)) {
        // every staged column is an image column: no byte shuffling; rows outside the image (one reflection is exact
        // on a level of that height) are read from their mirror rows
        LkInteriorRows<MAXR> v;
        if (region_rows_inside(R, h)) load_interior<MAXR, true>(v, img, pitch, h, R, lane);
        else load_interior<MAXR, false>(v, img, pitch, h, R, lane);
        store_interior<MAXR, LK_PITCH>(lds, v, lane);
        return R;
    }
    // The region reaches at most a window plus its margins beyond the image.  On a level larger than that, one
    // reflection per axis is exact, and a dword is assembled from two loads with a byte permute: A = the dword
    // itself (its bytes inside the image), B = the dword that holds the mirrored bytes (REFLECT_101 reverses their
    // order).  rx0 is a multiple of 4, so a dword never straddles the left border, only the right one.
    const bool single_reflection = w >= a_reach_w && h >= a_reach_h;
    if (single_reflection) {
        if (c < LK_RW_DWORDS) {
            const int gxc = min(gx, R.rx0 + 4 * (R.rwd - 1));    // columns past the region are never read
            const int n_in = w - gxc;                            // bytes of the dword left of the right border
            uint32_t offA, offB, sel;
            if (gxc < 0) { offB = (uint32_t)(-gxc - 3); offA = offB; sel = 0x04050607u; }
            else if (n_in >= 4) { offA = (uint32_t)gxc; offB = offA; sel = 0x03020100u; }
            else {
                offB = (uint32_t)(2 * w - 5 - gxc);
                offA = n_in > 0 ? (uint32_t)gxc : offB;
                sel = n_in <= 0 ? 0x04050607u : (n_in == 1 ? 0x04050600u : (n_in == 2 ? 0x04050100u : 0x04020100u));
            }
            uint32_t va[MAXR], vb[MAXR];
#pragma unroll
            for (int k = 0; k < MAXR; k++) {
                const uint32_t ro = __umul24(reflect101_once(R.ry0 + min(rr + 4 * k, R.rh - 1), h), pitch);
                va[k] = reinterpret_cast<const LkU32Unaligned*>(img + (ro + offA))->v;
                vb[k] = reinterpret_cast<const LkU32Unaligned*>(img + (ro + offB))->v;
            }
            uint32_t* dst = lds + rr * LK_PITCH + c;
#pragma unroll
            for (int k = 0; k < MAXR; k++) dst[4 * k * LK_PITCH] = __builtin_amdgcn_perm(vb[k], va[k], sel);
        }
        return R;
    }
    if (c < R.rwd) {   // degenerate level (smaller than a window's reach): byte by byte, general reflection
#pragma unroll 1
        for (int row = rr; row < R.rh; row += 4) {
            const uint8_t* src = img + (size_t)reflect101(R.ry0 + row, h) * pitch;
            uint32_t t = 0;
#pragma unroll 1
            for (int b = 0; b < 4; b++) t |= (uint32_t)src[reflect101(gx + b, w)] << (8 * b);
            lds[row * LK_PITCH + c] = t;
        }
    }
    return R;
}

// one wave per workgroup: LDS operations of a wave complete in order, this only pins the compiler's ordering
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool region_holds(const Region& R, int inx, int iny, int ext_w, int win_h) {
    return inx >= R.rx0 && inx + ext_w + 1 <= R.rx0 + 4 * R.rwd && iny >= R.ry0 && iny + win_h + 1 <= R.ry0 + R.rh;
}

// 2*NP consecutive bytes of one staged row starting at byte offset byte_off, as NP pairs of 16-bit values:
// P[m] = (byte 2m, byte 2m+1)
template <int NP, int LK_PITCH>
__device__ __forceinline__ void read_row_pairs_even(const uint32_t* __restrict__ lds, int row, int byte_off, uint32_t (&P)[NP]) {
    constexpr int NE = (2 * NP + 3) / 4;
    const uint32_t* p = lds + row * LK_PITCH + (byte_off >> 2);
    const uint32_t sh = byte_off & 3;
    uint32_t d[NE + 1];
#pragma unroll
    for (int q = 0; q <= NE; q++) d[q] = p[q];
#pragma unroll
    for (int q = 0; q < NE; q++) {
        const uint32_t e = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
        P[2 * q] = __builtin_amdgcn_perm(0u, e, 0x0c010c00u);
        if (2 * q + 1 < NP) P[2 * q + 1] = __builtin_amdgcn_perm(0u, e, 0x0c030c02u);
    }
}

// (J[k], J[k+1]) pairs, k = 0..SEGLEN-1, of one staged row as packed 16-bit pairs for v_dot2
template <int SEGLEN, int LK_PITCH>
__device__ __forceinline__ void read_row_pairs(const uint32_t* __restrict__ lds, int row, int byte_off, uint32_t (&pairs)[SEGLEN]) {
    constexpr int NDW = SEGLEN / 4 + 1;
    const uint32_t* p = lds + row * LK_PITCH + (byte_off >> 2);
    const uint32_t sh = byte_off & 3;
    uint32_t d[NDW + 1];
#pragma unroll
    for (int q = 0; q <= NDW; q++) d[q] = p[q];
    uint32_t e[NDW];   // bytes realigned to the window column
#pragma unroll
    for (int q = 0; q < NDW; q++) e[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
#pragma unroll
    for (int k = 0; k < SEGLEN; k++) {
        const int q = k >> 2, b = k & 3;
        // byte b and byte b+1 of the 8-byte string {e[q+1], e[q]} into the low halves of the two 16-bit lanes
        if (b < 3) pairs[k] = __builtin_amdgcn_perm(0u, e[q], 0x0c000c00u | (uint32_t)b | ((uint32_t)(b + 1) << 16));
        else pairs[k] = __builtin_amdgcn_perm(e[q + 1], e[q], 0x0c040c03u);
    }
}

// The same in two steps: the row's bytes realigned to the window column (kept across Newton iterations that stay within
// one pixel), and their unpacking into pairs.
template <int SEGLEN, int LK_PITCH>
__device__ __forceinline__ void read_row_aligned(const uint32_t* __restrict__ lds, int row, int byte_off, uint32_t (&e)[SEGLEN / 4 + 1]) {
    constexpr int NDW = SEGLEN / 4 + 1;
    const uint32_t* p = lds + row * LK_PITCH + (byte_off >> 2);
    const uint32_t sh = byte_off & 3;
    uint32_t d[NDW + 1];
#pragma unroll
    for (int q = 0; q <= NDW; q++) d[q] = p[q];
#pragma unroll
    for (int q = 0; q < NDW; q++) e[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
}
template <int SEGLEN>
__device__ __forceinline__ void unpack_row_pairs(const uint32_t (&e)[SEGLEN / 4 + 1], uint32_t (&pairs)[SEGLEN]) {
#pragma unroll
    for (int k = 0; k < SEGLEN; k++) {
        const int q = k >> 2, b = k & 3;
        if (b < 3) pairs[k] = __builtin_amdgcn_perm(0u, e[q], 0x0c000c00u | (uint32_t)b | ((uint32_t)(b + 1) << 16));
        else pairs[k] = __builtin_amdgcn_perm(e[q + 1], e[q], 0x0c040c03u);
    }
}

// One row of Scharr gradients on packed pairs, blended into the per-sample sums.  U, M, L: pixel pairs of the rows
// above, at and below the gradient row; column index c of a pair array is image column ipx + x0 - 1 + c.
// EDGE: zero the gradients that lie outside the image (the derivative image is zero-padded).
template <int SEGLEN, bool EDGE>
__device__ __forceinline__ void scharr_row_blend(const uint32_t (&U)[SEGLEN / 2 + 2], const uint32_t (&M)[SEGLEN / 2 + 2],
                                                 const uint32_t (&L)[SEGLEN / 2 + 2], uint32_t wpair, int (&sX)[SEGLEN], int (&sY)[SEGLEN],
                                                 bool row_in, int col0, int w) {
    constexpr int NP = SEGLEN / 2 + 2;
    lk_v2s T0[NP], T1[NP];
#pragma unroll
    for (int m = 0; m < NP; m++) {
        T0[m] = (v2s(U[m]) + v2s(L[m])) * (short)3 + v2s(M[m]) * (short)10;   // <= 4080
        T1[m] = v2s(L[m]) - v2s(U[m]);
    }
    // pairs starting at odd columns c = 2m+1: (G[2m+1], G[2m+2])
    uint32_t PX[NP - 1], PY[NP - 1];
#pragma unroll
    for (int m = 0; m < NP - 1; m++) {
        PX[m] = u32(T0[m + 1] - T0[m]);
        const lk_v2s mid = v2s(next_pair(u32(T1[m + 1]), u32(T1[m])));
        PY[m] = u32((T1[m] + T1[m + 1]) * (short)3 + mid * (short)10);
    }
#pragma unroll
    for (int k = 0; k < SEGLEN; k++) {
        // sample k blends gradient columns c = k+1 and k+2
        uint32_t px, py;
        if ((k & 1) == 0) { px = PX[k / 2]; py = PY[k / 2]; }
        else { px = next_pair(PX[(k + 1) / 2], PX[(k - 1) / 2]); py = next_pair(PY[(k + 1) / 2], PY[(k - 1) / 2]); }
        if (EDGE) {
            const int gx = col0 + k + 1;
            const uint32_t keep = ((row_in && gx >= 0 && gx < w) ? 0xffffu : 0u) | ((row_in && gx + 1 >= 0 && gx + 1 < w) ? 0xffff0000u : 0u);
            px &= keep; py &= keep;
        }
        sX[k] = dot2_acc(px, wpair, sX[k]);
        sY[k] = dot2_acc(py, wpair, sY[k]);
    }
}

// ---- what a wavefront works out once (LkWave), what it loads for its feature (LkPoint), and a level's geometry ----
template <int SEGLEN> struct LkWave {
    int lane, ww, wh, r, s, x0, rc, ext_w, reach_w, reach_h, n_valid;
    bool row_ok;
    float half_x, half_y, eps_f, eps_band, inv_area, min_eig_f;
    uint32_t keep[SEGLEN / 2];
};
struct LkPoint {
    float tx, ty;        // template point, level-0 coordinates
    float gx0, gy0;      // first guess
    bool use_init;
    int img_I, img_J;    // linear slot indices of the template and the target pyramid
    int w0, h0;          // level-0 size
};
// local names for the members of a LkWave (the blocks below were written against these names)
#define LK_WAVE_LOCALS(W)                                                                                                  \
    const int lane = (W).lane, ww = (W).ww, wh = (W).wh, r = (W).r, s = (W).s, x0 = (W).x0, rc = (W).rc, ext_w = (W).ext_w, \
              reach_w = (W).reach_w, reach_h = (W).reach_h, n_valid = (W).n_valid;                                         \
    const bool row_ok = (W).row_ok;                                                                                        \
    const float half_x = (W).half_x, half_y = (W).half_y, eps_f = (W).eps_f, eps_band = (W).eps_band, inv_area = (W).inv_area, \
                min_eig_f = (W).min_eig_f;                                                                                 \
    const uint32_t(&keep)[SEGLEN / 2] = (W).keep;                                                                          \
    constexpr float FLT_SCALE = 1.f / (1 << 20);                                                                           \
    (void)lane; (void)ww; (void)wh; (void)r; (void)s; (void)x0; (void)rc; (void)ext_w; (void)reach_w; (void)reach_h; (void)n_valid; \
    (void)row_ok; (void)half_x; (void)half_y; (void)eps_f; (void)eps_band; (void)inv_area; (void)min_eig_f; (void)keep; (void)FLT_SCALE;

// WIN > 0: the window is WIN x WIN, known when the kernel is compiled (the usual sizes have kernels of their own: masks,
// task counts and lane geometry then fold into constants); 0: the size comes with the launch arguments.
template <int SEGLEN, int WIN>
__device__ __forceinline__ void lk_prologue(const LKArgs& a, int pt, int lane, LkWave<SEGLEN>& W, LkPoint& P, bool echo = true) {
    const int ww = WIN ? WIN : a.win_w, wh = WIN ? WIN : a.win_h;
    const int segs = WIN ? (WIN + SEGLEN - 1) / SEGLEN : a.segs;
    const int segs_inv = WIN ? (65536 + segs - 1) / segs : a.segs_inv;
    const int r = (lane * segs_inv) >> 16, s = lane - r * segs;   // lane / segs by reciprocal: exact for lane < 64
    const int x0 = s * SEGLEN;
    const bool row_ok = r < wh;
    const int rc = row_ok ? r : wh - 1;          // lanes below the window reuse its last row (results masked)
    const int ext_w = segs * SEGLEN;             // columns the lanes touch: [0, ext_w]
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    // smallest level size for which a staged region needs at most one reflection per axis (with some slack)
    const int reach_w = ext_w + 2 * LK_MARGIN + 8, reach_h = wh + 2 * LK_MARGIN + 8;
    // which of the lane's samples belong to the window: 16-bit masks for the packed (sample 2m, sample 2m+1) pairs
    const int n_valid = row_ok ? min(max(ww - x0, 0), SEGLEN) : 0;
    uint32_t keep[SEGLEN / 2];
#pragma unroll
    for (int m = 0; m < SEGLEN / 2; m++) keep[m] = (n_valid > 2 * m ? 0xffffu : 0u) | (n_valid > 2 * m + 1 ? 0xffff0000u : 0u);

    // forward: template = prev pyramid at prev_xy, target = next pyramid, guess from guess_xy / H / prev
    // backward: template = next pyramid at the forward result, target = prev pyramid, guess = original point
    // Everything indexed by the point is loaded in one batch, then every field of the job record the kernel uses in a
    // second one: a wave starts with two memory round trips behind the kernel arguments instead of one per field.
    const float* t_src = a.backward ? a.next_xy : a.prev_xy;
    const float* g_src = a.backward ? a.prev_xy : a.guess_xy;
    const int job_i = a.pt_job[pt];
    const float tx = t_src[2 * pt], ty = t_src[2 * pt + 1];
    float gx0 = g_src[2 * pt], gy0 = g_src[2 * pt + 1];
    const LKJobDev* job = a.jobs + job_i;
    const int img_I = a.backward ? job->img_next : job->img_prev;
    const int img_J = a.backward ? job->img_prev : job->img_next;
    const int job_use_H = job->use_H, job_use_stereo = job->use_stereo;
    float H[9];
#pragma unroll
    for (int k = 0; k < 9; k++) H[k] = job->H[k];
    const int w0 = job->w0, h0 = job->h0;      // level-0 size (a load inside the level loop would be a round trip per level)
    // (the empty statement uses all of them here, so that the loads are issued together rather than where a branch
    // first needs each)
    asm volatile("" ::"s"(img_I), "s"(img_J), "s"(job_use_H), "s"(job_use_stereo), "s"(w0), "s"(h0), "s"(H[0]), "s"(H[1]), "s"(H[2]), "s"(H[3]),
                 "s"(H[4]), "s"(H[5]), "s"(H[6]), "s"(H[7]), "s"(H[8]));
    bool use_init = true;
    if (!a.backward) {
        if (job_use_H) {
            const float p0 = H[0] * tx + H[1] * ty + H[2];
            const float p1 = H[3] * tx + H[4] * ty + H[5];
            const float p2 = H[6] * tx + H[7] * ty + H[8];
            gx0 = p0 / p2; gy0 = p1 / p2;
        } else {
            use_init = (a.flags & FT_LK_USE_INITIAL_FLOW) != 0 || job_use_stereo != 0;   // stereo guesses were written on the device
        }
    }

    if (a.echo_jobs && echo) {   // (the same for the whole wavefront)
        if (lane == 0) { a.echo_prev_xy[2 * pt] = a.prev_xy[2 * pt]; a.echo_prev_xy[2 * pt + 1] = a.prev_xy[2 * pt + 1]; }
        for (int j = pt; j < a.n_jobs; j += a.total)
            if (lane < (int)(sizeof(LKJobDev) / 8)) reinterpret_cast<uint64_t*>(a.echo_jobs + j)[lane] = reinterpret_cast<const uint64_t*>(a.jobs + j)[lane];
    }
    const float eps_f = (float)a.epsilon, eps_band = eps_f * 1e-4f + 1e-30f;
    const float inv_area = 1.f / (float)(2 * ww * wh), min_eig_f = (float)a.min_eig;
    W.lane = lane; W.ww = ww; W.wh = wh; W.r = r; W.s = s; W.x0 = x0; W.rc = rc; W.ext_w = ext_w; W.reach_w = reach_w; W.reach_h = reach_h;
    W.n_valid = n_valid; W.row_ok = row_ok; W.half_x = half_x; W.half_y = half_y; W.eps_f = eps_f; W.eps_band = eps_band;
    W.inv_area = inv_area; W.min_eig_f = min_eig_f;
#pragma unroll
    for (int m = 0; m < SEGLEN / 2; m++) W.keep[m] = keep[m];
    P.tx = tx; P.ty = ty; P.gx0 = gx0; P.gy0 = gy0; P.use_init = use_init; P.img_I = img_I; P.img_J = img_J; P.w0 = w0; P.h0 = h0;
}

// size, positions and images of one level; advances the running estimate (outx, outy) to this level's scale
struct LkLevel {
    int w, h;
    float px, py;        // template window corner (sub-pixel), this level
    float nx, ny;        // target window corner at the start of the level
    int ipx, ipy;        // integer part of (px, py)
    bool tmpl_out;       // the template window lies outside the image
    uint32_t ipitch;
    const uint8_t* imgI;
    const uint8_t* imgJ;
};
template <int SEGLEN>
__device__ __forceinline__ LkLevel lk_level(const LKArgs& a, const LkWave<SEGLEN>& W, const LkPoint& P, int level, float& outx, float& outy) {
    LkLevel G;
    int w = P.w0, h = P.h0;
    for (int l = 0; l < level; l++) { w = half_up(w); h = half_up(h); }
    const float scale = __builtin_bit_cast(float, (uint32_t)(127 - level) << 23);   // 2^-level
    float px = P.tx * scale, py = P.ty * scale;
    float nx, ny;
    if (level == a.max_level) {
        if (P.use_init) { nx = outx * scale; ny = outy * scale; }
        else { nx = px; ny = py; }
    } else {
        nx = outx * 2.f; ny = outy * 2.f;
    }
    outx = nx; outy = ny;
    px -= W.half_x; py -= W.half_y;
    // positions are identical in all lanes: keep the integer parts in scalar registers
    G.ipx = lk_uniform_floor(px); G.ipy = lk_uniform_floor(py);
    G.tmpl_out = G.ipx < -W.ww || G.ipx >= w || G.ipy < -W.wh || G.ipy >= h;
    G.w = w; G.h = h; G.px = px; G.py = py; G.nx = nx - W.half_x; G.ny = ny - W.half_y;
    G.ipitch = a.A.lv[level].img_pitch;
    G.imgI = level_img(a.A, P.img_I, level);
    G.imgJ = level_img(a.A, P.img_J, level);
    return G;
}

// The template of a level from the staged I region RI: per-lane samples Cw, gradient pairs GX / GY, the 2x2 matrix.
template <int SEGLEN, bool TALL>
__device__ __forceinline__ void lk_template(const LkWave<SEGLEN>& W, const uint32_t* __restrict__ ilds, const Region& RI, float px, float py, int ipx,
                                            int ipy, int w, int h, int (&Cw)[SEGLEN], uint32_t (&GX)[SEGLEN / 2], uint32_t (&GY)[SEGLEN / 2],
                                            float& A11, float& A12, float& A22) {
    LK_WAVE_LOCALS(W)
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int NP = SEGLEN / 2 + 2;                              // pixel pairs a lane reads per template row
    constexpr int LK_PITCH = LkPitch<SEGLEN>::value;
    // ---- template: bilinear samples of I (x32) and of its Scharr gradients, all from the staged region ----
    // Kept for the iterations: Cw[k] = 256 - 512 * I(k) (folds the subtraction of the template sample into the
    // accumulator of the target blend), and the gradients as pairs GX[m] = (Ix(2m), Ix(2m+1)), GY[m] likewise.
    {
        const BilinW bw = bilin_weights(px - ipx, py - ipy);
        const uint32_t wtop = (uint32_t)bw.w00 | ((uint32_t)bw.w01 << 16), wbot = (uint32_t)bw.w10 | ((uint32_t)bw.w11 << 16);
        int sX[SEGLEN], sY[SEGLEN];
#pragma unroll
        for (int k = 0; k < SEGLEN; k++) { sX[k] = 1 << 13; sY[k] = 1 << 13; }   // rounding term of the 14-bit descale
        const bool inside = ipx >= 0 && ipx + ext_w < w && ipy >= 0 && ipy + wh < h;   // every gradient tap is in the image
        uint32_t Rb[NP], Rc[NP];   // pixel pairs of the two rows the lane's template samples blend (old path: read below)
        if constexpr (SEGLEN == 8 && !TALL) {
            // Two steps through LDS.  (1) The integer Scharr gradients of the (wh + 1) x (ww + 1) patch are computed
            // once, a half row of 12 columns per lane (22 x 2 tasks for 21 x 21), instead of 2.6 times over in the
            // lanes that blend them; (2) every lane blends its eight samples from two rows of that patch.
            constexpr int GP = 19;                               // dwords per patch row (pairs of columns), odd
            __shared__ uint32_t gxl[(WH_MAX + 1) * GP], gyl[(WH_MAX + 1) * GP];
            const int tpr = (ww + 1 + 11) / 12;                  // tasks per patch row
            const int n_tasks = (wh + 1) * tpr;
            for (int task = lane; task < n_tasks; task += 64) {
                const int ra = tpr == 2 ? task >> 1 : (tpr == 1 ? task : task / 3), ha = task - ra * tpr;
                const int boffA = ipx - 1 - RI.rx0 + 12 * ha;    // byte 0: image column ipx + 12 ha - 1
                uint32_t U[7], M[7], L[7];
                read_row_pairs_even<7, LK_PITCH>(ilds, ra, boffA, U);
                read_row_pairs_even<7, LK_PITCH>(ilds, ra + 1, boffA, M);
                read_row_pairs_even<7, LK_PITCH>(ilds, ra + 2, boffA, L);
                lk_v2s T0[7], T1[7];
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    T0[m] = (v2s(U[m]) + v2s(L[m])) * (short)3 + v2s(M[m]) * (short)10;   // <= 4080
                    T1[m] = v2s(L[m]) - v2s(U[m]);
                }
                const int gyA = ipy + ra, gxA = ipx + 12 * ha;
                const bool row_in = gyA >= 0 && gyA < h;
#pragma unroll
                for (int m = 0; m < 6; m++) {                    // patch columns 12 ha + 2 m, + 1
                    uint32_t px = u32(T0[m + 1] - T0[m]);
                    const lk_v2s mid = v2s(next_pair(u32(T1[m + 1]), u32(T1[m])));
                    uint32_t py = u32((T1[m] + T1[m + 1]) * (short)3 + mid * (short)10);
                    if (!inside) {                               // the derivative image is zero outside the image
                        const int gx = gxA + 2 * m;
                        const uint32_t in = ((row_in && gx >= 0 && gx < w) ? 0xffffu : 0u) | ((row_in && gx + 1 >= 0 && gx + 1 < w) ? 0xffff0000u : 0u);
                        px &= in; py &= in;
                    }
                    gxl[ra * GP + 6 * ha + m] = px;
                    gyl[ra * GP + 6 * ha + m] = py;
                }
            }
            lds_fence();
            // the lane's samples k = 0..7 are patch columns x0 + k, x0 + k + 1 of patch rows rc, rc + 1
            {
                const uint32_t* ga = &gxl[rc * GP + 4 * s];
                const uint32_t* gb = &gyl[rc * GP + 4 * s];
                uint32_t xa[5], xb[5], ya[5], yb[5];
#pragma unroll
                for (int q = 0; q < 5; q++) { xa[q] = ga[q]; xb[q] = ga[GP + q]; ya[q] = gb[q]; yb[q] = gb[GP + q]; }
#pragma unroll
                for (int k = 0; k < SEGLEN; k++) {
                    const int q = k >> 1;
                    const uint32_t pxa = (k & 1) ? next_pair(xa[q + 1], xa[q]) : xa[q], pxb = (k & 1) ? next_pair(xb[q + 1], xb[q]) : xb[q];
                    const uint32_t pya = (k & 1) ? next_pair(ya[q + 1], ya[q]) : ya[q], pyb = (k & 1) ? next_pair(yb[q + 1], yb[q]) : yb[q];
                    sX[k] = dot2_acc(pxb, wbot, dot2_to_uniform(pxa, wtop, 1 << 13));   // 1 << 13: rounding term of the 14-bit descale
                    sY[k] = dot2_acc(pyb, wbot, dot2_to_uniform(pya, wtop, 1 << 13));
                }
            }
            // pixel pairs for the template samples: region rows rc + 1, rc + 2, from image column ipx + x0 - 1
            read_row_pairs_even<NP, LK_PITCH>(ilds, rc + 1, ipx + x0 - 1 - RI.rx0, Rb);
            read_row_pairs_even<NP, LK_PITCH>(ilds, rc + 2, ipx + x0 - 1 - RI.rx0, Rc);
        } else {
        // the lane interpolates between image rows ipy+rc and ipy+rc+1; their gradients need rows ipy+rc-1 .. +2,
        // which are region rows rc .. rc+3; columns ipx+x0-1 .. ipx+x0+SEGLEN+2
        const int boff = ipx + x0 - 1 - RI.rx0;
        uint32_t Ra[NP], Rd[NP];
        read_row_pairs_even<NP, LK_PITCH>(ilds, rc, boff, Ra);
        read_row_pairs_even<NP, LK_PITCH>(ilds, rc + 1, boff, Rb);
        read_row_pairs_even<NP, LK_PITCH>(ilds, rc + 2, boff, Rc);
        const int gy = ipy + rc;
        // the two gradient rows one after the other (the scheduling fences keep their temporaries from overlapping)
        // (the column passes through an opaque assembly statement inside the border branch: otherwise the compiler
        // hoists the border masks' comparisons above the branch and every window pays for them)
        if (inside) scharr_row_blend<SEGLEN, false>(Ra, Rb, Rc, wtop, sX, sY, true, 0, w);
        else { int col0 = ipx + x0 - 1; asm volatile("" : "+v"(col0)); scharr_row_blend<SEGLEN, true>(Ra, Rb, Rc, wtop, sX, sY, gy >= 0 && gy < h, col0, w); }
        __builtin_amdgcn_sched_barrier(0);
        read_row_pairs_even<NP, LK_PITCH>(ilds, rc + 3, boff, Rd);
        if (inside) scharr_row_blend<SEGLEN, false>(Rb, Rc, Rd, wbot, sX, sY, true, 0, w);
        else { int col0 = ipx + x0 - 1; asm volatile("" : "+v"(col0)); scharr_row_blend<SEGLEN, true>(Rb, Rc, Rd, wbot, sX, sY, gy + 1 >= 0 && gy + 1 < h, col0, w); }
        __builtin_amdgcn_sched_barrier(0);
        }
        int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int m = 0; m < SEGLEN / 2; m++) {
            GX[m] = pack_lo16(sX[2 * m + 1] >> 14, sX[2 * m] >> 14) & keep[m];
            GY[m] = pack_lo16(sY[2 * m + 1] >> 14, sY[2 * m] >> 14) & keep[m];
            a11 = m == 0 ? dot2_to_zero(GX[m], GX[m]) : dot2_acc(GX[m], GX[m], a11);
            a12 = m == 0 ? dot2_to_zero(GX[m], GY[m]) : dot2_acc(GX[m], GY[m], a12);
            a22 = m == 0 ? dot2_to_zero(GY[m], GY[m]) : dot2_acc(GY[m], GY[m], a22);
        }
#pragma unroll
        for (int k = 0; k < SEGLEN; k++) {
            // sample k blends pixel columns c = k+1, k+2 of region rows rc+1, rc+2
            const uint32_t pb = (k & 1) ? Rb[(k + 1) / 2] : next_pair(Rb[k / 2 + 1], Rb[k / 2]);
            const uint32_t pc = (k & 1) ? Rc[(k + 1) / 2] : next_pair(Rc[k / 2 + 1], Rc[k / 2]);
            const int iw = dot2_acc(pc, wbot, dot2_to(pb, wtop, 256)) >> 9;
            Cw[k] = 256 - (iw << 9);
        }
        wave_sum_exact3(a11, a22, a12, A11, A22, A12);
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
    }
}

// The smaller eigenvalue of the matrix against the threshold, and its determinant: 0 = the level can be tracked,
// 2 = flat or degenerate template.  The eigenvalue's value is a result only when the caller asks for it (level 0,
// eig_value); otherwise only the outcome of the comparison counts, and the hardware's square root and a reciprocal
// multiply (together within 1e-6 of the trace over the area, relative) decide it whenever the value is clear of the
// threshold by four times that; the correctly rounded square root and division run only in that band.
template <int SEGLEN>
__device__ __forceinline__ int lk_matrix_test(const LKArgs& a, const LkWave<SEGLEN>& W, int bwd, int level, float A11, float A12, float A22, float& eig_value) {
    const float D = A11 * A22 - A12 * A12;
    const float eig_tr = A22 + A11, eig_disc = (A11 - A22) * (A11 - A22) + 4.f * A12 * A12;
    bool eig_low;
    const float approx = (eig_tr - __builtin_amdgcn_sqrtf(eig_disc)) * W.inv_area;
    const bool want_value = (a.flags & FT_LK_GET_MIN_EIGENVALS) && level == 0 && !bwd;
    if (!want_value && fabsf(approx - W.min_eig_f) > (eig_tr * W.inv_area + W.min_eig_f) * 4e-6f) {
        eig_low = approx < W.min_eig_f;
    } else {
        const float minEig = (eig_tr - sqrtf(eig_disc)) / (float)(2 * W.ww * W.wh);
        eig_value = minEig;
        eig_low = (double)minEig < a.min_eig;
    }
    return (eig_low || D < 1.1920928955078125e-07f) ? 2 : 0;
}

// Newton iterations of one level on the staged J region R (re-staged when the window walks out of it), the per-level
// iteration count, and at level 0 the residual at the final position.
template <int SEGLEN, bool TALL, bool HOLD = false>
__device__ __forceinline__ void lk_track_level(const LKArgs& a, const LkWave<SEGLEN>& W, int bwd, int pt, int level, uint32_t* __restrict__ jlds,
                                               const uint8_t* __restrict__ imgJ, uint32_t ipitch, int w, int h, const int (&Cw)[SEGLEN],
                                               const uint32_t (&GX)[SEGLEN / 2], const uint32_t (&GY)[SEGLEN / 2], float A11, float A12, float A22,
                                               Region& R, float nx, float ny, float& outx, float& outy, int& st, float& errv) {
    LK_WAVE_LOCALS(W)
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int J_ROWS = (WH_MAX + 1 + 2 * LK_MARGIN + 3) / 4;    // staged rows per lane, target region
    constexpr int LK_PITCH = LkPitch<SEGLEN>::value;
    float D = A11 * A22 - A12 * A12;
    D = 1.f / D;
    int n_iter;
    // ---- Newton iterations on the target image, fed from LDS ----
    float pdx = 0.f, pdy = 0.f;
    n_iter = 0;
    // What a lane keeps across Newton iterations that stay within one pixel: with 4-pixel segments the pixel pairs of its
    // two rows (8 registers, and 16 unpacking instructions less per such iteration: 15 x 15 on 9 531 features 67.4 -> 65.9 us);
    // with longer segments only the realigned rows: the pairs of 8-pixel segments (16 registers) take the kernel from 112
    // to 122 registers, which makes it 2 % faster alone (79.9 -> 78.1 us) and the 64-stream pipeline 1 % slower, because
    // the other queues' kernels then find no registers left beside four tracking wavefronts on a SIMD.  HOLD: the form
    // with the pairs held all the same, for a context whose tracking launches have the device to themselves (one stream)
    constexpr bool HOLD_PAIRS = SEGLEN <= 4 || HOLD;
    uint32_t e0[SEGLEN / 4 + 1], e1[SEGLEN / 4 + 1];
    uint32_t p0[SEGLEN], p1[SEGLEN];           // (J[k], J[k+1]) pairs of the lane's two rows
    int held_x = INT_MIN, held_y = INT_MIN;    // integer position whose rows are held
    for (int j = 0; j < a.max_iter; j++) {
        const int inx = lk_uniform_floor(nx), iny = lk_uniform_floor(ny);
        if (inx < -ww || inx >= w || iny < -wh || iny >= h) {
            if (level == 0) st = 0;
            break;
        }
        n_iter = j + 1;
        if (!region_holds(R, inx, iny, ext_w, wh)) {
            R = stage_region<J_ROWS, LK_PITCH>(jlds, imgJ, ipitch, w, h, inx - LK_MARGIN, inx + ext_w + LK_MARGIN, iny - LK_MARGIN, wh + 1 + 2 * LK_MARGIN, lane, reach_w, reach_h);
            lds_fence();
        }
        const BilinW jw = bilin_weights(nx - inx, ny - iny);
        const uint32_t wtop = (uint32_t)jw.w00 | ((uint32_t)jw.w01 << 16), wbot = (uint32_t)jw.w10 | ((uint32_t)jw.w11 << 16);
        // most steps after the first stay within one pixel: the rows are then those of the last iteration
        if (inx != held_x || iny != held_y) {
            const int boff = inx - R.rx0 + x0, row = iny - R.ry0 + rc;
            read_row_aligned<SEGLEN, LK_PITCH>(jlds, row, boff, e0);
            read_row_aligned<SEGLEN, LK_PITCH>(jlds, row + 1, boff, e1);
            if (HOLD_PAIRS) {
                unpack_row_pairs<SEGLEN>(e0, p0);
                unpack_row_pairs<SEGLEN>(e1, p1);
            }
            held_x = inx; held_y = iny;
        }
        if (!HOLD_PAIRS) {
            unpack_row_pairs<SEGLEN>(e0, p0);
            unpack_row_pairs<SEGLEN>(e1, p1);
        }
        int b1 = 0, b2 = 0;
#pragma unroll
        for (int m = 0; m < SEGLEN / 2; m++) {
            const int d0 = dot2_acc(p1[2 * m], wbot, dot2_to(p0[2 * m], wtop, Cw[2 * m])) >> 9;           // J - I, sample 2m
            const int d1 = dot2_acc(p1[2 * m + 1], wbot, dot2_to(p0[2 * m + 1], wtop, Cw[2 * m + 1])) >> 9;
            const uint32_t dp = pack_lo16(d1, d0);
            b1 = m == 0 ? dot2_to_zero(dp, GX[m]) : dot2_acc(dp, GX[m], b1);
            b2 = m == 0 ? dot2_to_zero(dp, GY[m]) : dot2_acc(dp, GY[m], b2);
        }
        float fb1, fb2;
        wave_sum_exact2(b1, b2, fb1, fb2);
        fb1 *= FLT_SCALE; fb2 *= FLT_SCALE;
        const float dx = (A12 * fb2 - A22 * fb1) * D;
        const float dy = (A12 * fb1 - A11 * fb2) * D;
        nx += dx; ny += dy;
        outx = nx + half_x; outy = ny + half_y;
        // delta . delta <= epsilon in double precision, as the reference computes it.  The float value decides
        // whenever it is clear of the bound by far more than its rounding error (a few 1e-7 relative); the double
        // arithmetic, at half the issue rate, runs only in the band around the bound.  Uniform for the wave.
        {
            const float s = dx * dx + dy * dy;
            const bool far = fabsf(s - eps_f) > eps_band;
            if (far ? s < eps_f : (double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) break;
        }
        // |dx + pdx| < 0.01 and |dy + pdy| < 0.01 in double precision: for a float, "< 0.01" is "< the float above 0.01"
        if (j > 0 && fabsf(dx + pdx) < 0.010000000707805157f && fabsf(dy + pdy) < 0.010000000707805157f) {
            outx -= dx * 0.5f; outy -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
    if (a.iters && !bwd && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;

    // ---- residual at the final position (level 0 only) ----
    if (st && level == 0 && !(a.flags & FT_LK_GET_MIN_EIGENVALS) && !bwd) {
        const float qx = outx - half_x, qy = outy - half_y;
        const int iqx = lk_uniform_floor(qx), iqy = lk_uniform_floor(qy);
        if (iqx < -ww || iqx >= w || iqy < -wh || iqy >= h) {
            st = 0;
        } else if (!(a.flags & FT_LK_NO_ERROR)) {
            if (!region_holds(R, iqx, iqy, ext_w, wh)) {
                R = stage_region<J_ROWS, LK_PITCH>(jlds, imgJ, ipitch, w, h, iqx - LK_MARGIN, iqx + ext_w + LK_MARGIN, iqy - LK_MARGIN, wh + 1 + 2 * LK_MARGIN, lane, reach_w, reach_h);
                lds_fence();
            }
            const BilinW qw = bilin_weights(qx - iqx, qy - iqy);
            const uint32_t wtop = (uint32_t)qw.w00 | ((uint32_t)qw.w01 << 16), wbot = (uint32_t)qw.w10 | ((uint32_t)qw.w11 << 16);
            uint32_t p0[SEGLEN], p1[SEGLEN];
            const int boff = iqx - R.rx0 + x0, row = iqy - R.ry0 + rc;
            read_row_pairs<SEGLEN, LK_PITCH>(jlds, row, boff, p0);
            read_row_pairs<SEGLEN, LK_PITCH>(jlds, row + 1, boff, p1);
            int e = 0;
#pragma unroll
            for (int k = 0; k < SEGLEN; k++) {
                const int diff = dot2_acc(p1[k], wbot, dot2_to(p0[k], wtop, Cw[k])) >> 9;
                if (k < n_valid) e += abs(diff);
            }
            const float errsum = (float)wave_sum_i32(e);   // < 2^24: exact
            errv = errsum * 1.f / (float)(32 * ww * wh);
        }
    }
}

__device__ __forceinline__ void lk_write_result(const LKArgs& a, int pt, const LkPoint& P, float outx, float outy, int st, float errv) {
    if (!a.backward) {
        a.next_xy[2 * pt] = outx;
        a.next_xy[2 * pt + 1] = outy;
        a.status[pt] = (uint8_t)st;
        a.err[pt] = errv;
    } else {
        // forward-backward consistency: the backward track must succeed and return to the original point
        const float dx = outx - P.gx0, dy = outy - P.gy0;
        const float d2 = dx * dx + dy * dy;
        if (!st || !(d2 <= a.fb_thresh2)) a.status[pt] = 0;
    }
}

// Forward and backward pass of a feature in one launch (FUSED): the backward pass of a feature needs nothing but that
// feature's forward result, so the wavefront that found it goes straight on, with the result in registers: template =
// the target pyramid at the forward result, target = the template pyramid, first guess = the original point.  The
// values are those a second launch reads back from memory, so every result is the same; what goes away is the second
// launch with its start and its end.  A feature whose forward track failed keeps status 0 whatever a backward pass
// finds, so it gets none.
// The forward result (x, y, residual) waits in LDS, written and read back by lane 0 of the same wavefront; the
// positions that steer the backward pass are the same in all lanes and stay in scalar registers.
__device__ __forceinline__ float lk_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ void lk_turn_around(LkPoint& P, float outx, float outy) {
    const float ox = P.tx, oy = P.ty;
    P.tx = lk_uniform(outx); P.ty = lk_uniform(outy); P.gx0 = ox; P.gy0 = oy; P.use_init = true;
    const int t = P.img_I; P.img_I = P.img_J; P.img_J = t;
}
__device__ __forceinline__ void lk_write_fused(const LKArgs& a, int pt, const LkPoint& P, const float* fwd, float outx, float outy, int st) {
    const float dx = outx - P.gx0, dy = outy - P.gy0;
    const float d2 = dx * dx + dy * dy;
    a.next_xy[2 * pt] = fwd[0];
    a.next_xy[2 * pt + 1] = fwd[1];
    a.status[pt] = (uint8_t)((st && d2 <= a.fb_thresh2) ? 1 : 0);
    a.err[pt] = fwd[2];
}
#undef LK_WAVE_LOCALS

// All levels of one feature on one wavefront.  SEGLEN: pixels per lane segment; TALL: window height above 21 rows (more
// staged rows per lane); FUSED: forward pass, then the backward pass of the same feature (launched with a.backward = 0).
// jlds / ilds: the wavefront's LDS arrays for the target and the template region.
template <int SEGLEN, bool TALL, bool FUSED, int WIN, bool HOLD = false>
__device__ __forceinline__ void lk_one_point(const LKArgs& a, int pt, int lane, uint32_t* __restrict__ jlds, uint32_t* __restrict__ ilds, float* __restrict__ fwd) {
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int I_ROWS = (WH_MAX + 3 + 3) / 4;                    // staged rows per lane, template region
    constexpr int J_ROWS = (WH_MAX + 1 + 2 * LK_MARGIN + 3) / 4;    // staged rows per lane, target region
    constexpr int LK_PITCH = LkPitch<SEGLEN>::value;
    LkWave<SEGLEN> W;
    LkPoint P;
    lk_prologue<SEGLEN, WIN>(a, pt, lane, W, P);
    int bwd = a.backward;               // direction of the pass that runs now
#pragma nounroll
    for (;;) {
    float outx = P.gx0, outy = P.gy0;   // nextPts[i]
    int st = 1;
    float errv = 0.f;

    for (int level = a.max_level; level >= 0; level--) {
        const LkLevel G = lk_level<SEGLEN>(a, W, P, level, outx, outy);
        if (G.tmpl_out) {
            if (level == 0) { st = 0; errv = 0.f; }
            if (a.iters && !bwd && lane == 0) a.iters[(size_t)pt * a.nlev + level] = -1;
            continue;
        }
        // ---- stage both regions of this level: their loads are in flight together ----
        const int w = G.w, h = G.h;
        const int inx0 = lk_uniform_floor(G.nx), iny0 = lk_uniform_floor(G.ny);
        Region RI = region_of(G.ipx - 1, G.ipx + W.ext_w + 1, G.ipy - 1, W.wh + 3);
        Region R = region_of(inx0 - LK_MARGIN, inx0 + W.ext_w + LK_MARGIN, iny0 - LK_MARGIN, W.wh + 1 + 2 * LK_MARGIN);
        if (region_cols_inside(RI, w) && region_cols_inside(R, w) && h >= W.reach_h && !(iny0 < -W.wh || iny0 >= h)) {
            // the usual case: one memory round trip for both regions and no byte shuffling; rows outside the image
            // (regions near the upper or lower border) are read from their mirror rows
            LkInteriorRows<I_ROWS> vi;
            LkInteriorRows<J_ROWS> vj;
            if (region_rows_inside(RI, h) && region_rows_inside(R, h)) {
                load_interior<I_ROWS, true>(vi, G.imgI, G.ipitch, h, RI, lane);
                load_interior<J_ROWS, true>(vj, G.imgJ, G.ipitch, h, R, lane);
            } else {
                load_interior<I_ROWS, false>(vi, G.imgI, G.ipitch, h, RI, lane);
                load_interior<J_ROWS, false>(vj, G.imgJ, G.ipitch, h, R, lane);
            }
            store_interior<I_ROWS, LK_PITCH>(ilds, vi, lane);
            store_interior<J_ROWS, LK_PITCH>(jlds, vj, lane);
        } else {
            RI = stage_region<I_ROWS, LK_PITCH>(ilds, G.imgI, G.ipitch, w, h, G.ipx - 1, G.ipx + W.ext_w + 1, G.ipy - 1, W.wh + 3, lane, W.reach_w, W.reach_h);
            R.rx0 = 0; R.ry0 = 0; R.rwd = 0; R.rh = 0;   // nothing staged yet
            if (!(inx0 < -W.ww || inx0 >= w || iny0 < -W.wh || iny0 >= h))
                R = stage_region<J_ROWS, LK_PITCH>(jlds, G.imgJ, G.ipitch, w, h, inx0 - LK_MARGIN, inx0 + W.ext_w + LK_MARGIN, iny0 - LK_MARGIN, W.wh + 1 + 2 * LK_MARGIN, lane, W.reach_w, W.reach_h);
        }
        lds_fence();

        int Cw[SEGLEN];
        uint32_t GX[SEGLEN / 2], GY[SEGLEN / 2];
        float A11, A12, A22;
        lk_template<SEGLEN, TALL>(W, ilds, RI, G.px, G.py, G.ipx, G.ipy, w, h, Cw, GX, GY, A11, A12, A22);
        float eig_value = 0.f;
        const int code = lk_matrix_test<SEGLEN>(a, W, bwd, level, A11, A12, A22, eig_value);
        if ((a.flags & FT_LK_GET_MIN_EIGENVALS) && level == 0 && !bwd) errv = eig_value;
        if (code != 0) {
            if (level == 0) st = 0;
            if (a.iters && !bwd && lane == 0) a.iters[(size_t)pt * a.nlev + level] = -1;
            continue;
        }
        lk_track_level<SEGLEN, TALL, HOLD>(a, W, bwd, pt, level, jlds, G.imgJ, G.ipitch, w, h, Cw, GX, GY, A11, A12, A22, R, G.nx, G.ny, outx, outy, st, errv);
        __builtin_amdgcn_wave_barrier();   // all LDS reads of this level precede the next level's staging
    }
    if (!FUSED) {
        if (lane == 0) lk_write_result(a, pt, P, outx, outy, st, errv);
        break;
    }
    if (bwd) {
        if (lane == 0) lk_write_fused(a, pt, P, fwd, outx, outy, st);
        break;
    }
    if (!__builtin_amdgcn_readfirstlane(st)) {   // (the same in all lanes)
        if (lane == 0) lk_write_result(a, pt, P, outx, outy, 0, errv);
        break;
    }
    if (lane == 0) { fwd[0] = outx; fwd[1] = outy; fwd[2] = errv; }
    lk_turn_around(P, outx, outy);
    bwd = 1;
    }
}

// One wavefront per feature, one single-wavefront workgroup each: the hardware dispatcher hands a free wave slot its
// next feature.  Measured against resident wavefronts that loop over features (9 531 features, 21 x 21, 4 levels;
// 83.7 us as launched here): features taken from per-XCD lists through atomic counters 119 us (some 1 200 dependent
// atomics per counter, performed memory-side, cost more than the dispatches they replace), features dealt to the
// resident wavefronts in fixed strides 88 us (no counters, but no balancing either).  Wavefronts that stay warm gain
// nothing: the launch is bound by vector issue between its ends, and its ends by one wavefront's dependent chain.
template <int SEGLEN, bool TALL, bool FUSED, int WIN, bool HOLD = false>
__global__ void __launch_bounds__(64) lk_kernel(LKArgs a) {
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int I_ROWS = (WH_MAX + 3 + 3) / 4;
    constexpr int J_ROWS = (WH_MAX + 1 + 2 * LK_MARGIN + 3) / 4;
    constexpr int LK_PITCH = LkPitch<SEGLEN>::value;
    __shared__ uint32_t jlds[(4 * J_ROWS + 1) * LK_PITCH + 4];  // target region (+ a row for the staging's surplus rows, + slack: row reads are dword-granular)
    __shared__ uint32_t ilds[(4 * I_ROWS + 1) * LK_PITCH + 4];  // template region: window + Scharr apron
    const int pt = xcd_item(blockIdx.x, a.total);   // points of one image pair stay on one XCD
    if (pt >= a.total) return;
    __shared__ float fwd[3];                                     // FUSED: the forward result during the backward pass
    lk_one_point<SEGLEN, TALL, FUSED, WIN, HOLD>(a, pt, threadIdx.x, jlds, ilds, fwd);
}

// The same tracking with two wavefronts per feature.  The template of a level (staging of the I region, Scharr patch,
// blends, the 2x2 matrix and its tests) does not depend on the tracking result, only the Newton iterations on J do.
// Wave 1 (the templater) therefore works one level ahead of wave 0 (the tracker): while the tracker iterates on level
// l, the templater prepares level l - 1 and hands it over through LDS (the per-lane template registers Cw, GX, GY, the
// matrix, the outcome of its tests).  Both waves pass the same two workgroup barriers in every level, whatever the
// level's outcome: the first says "the template is in LDS", the second "the tracker has taken it"; in between only
// the tracker touches the hand-over area, outside only the templater.  The chain of dependent steps of a feature gets
// a third shorter; the arithmetic, and so every result, is that of lk_kernel.  FUSED: after the forward pass the
// tracker hands its result to the templater through LDS (one more barrier for both) and both go through the levels
// again for the backward pass.
template <int SEGLEN, bool TALL, bool FUSED, int WIN, bool HOLD = false>
__global__ void __launch_bounds__(128) lk_pair_kernel(LKArgs a) {
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int I_ROWS = (WH_MAX + 3 + 3) / 4;                    // staged rows per lane, template region
    constexpr int J_ROWS = (WH_MAX + 1 + 2 * LK_MARGIN + 3) / 4;    // staged rows per lane, target region
    constexpr int LK_PITCH = LkPitch<SEGLEN>::value;
    __shared__ uint32_t jlds[(4 * J_ROWS + 1) * LK_PITCH + 4];  // target region: the tracker's
    __shared__ uint32_t ilds[(4 * I_ROWS + 1) * LK_PITCH + 4];  // template region: the templater's
    __shared__ uint32_t hand[2 * SEGLEN * 64];                  // hand-over: Cw[SEGLEN], GX[SEGLEN/2], GY[SEGLEN/2] of every lane
    __shared__ float hand_f[4];                                 // A11, A12, A22, smaller eigenvalue (when asked for)
    __shared__ int hand_code;                                   // 0: template ready, 1: window outside the image, 2: matrix tests failed
    __shared__ float turn_f[3];                                 // FUSED: forward result (x, y, residual) ...
    __shared__ int turn_st;                                     // ... and status, from the tracker to the templater
    const int pt = xcd_item(blockIdx.x, a.total);   // points of one image pair stay on one XCD
    if (pt >= a.total) return;
    const int lane = threadIdx.x & 63;
    const bool templater = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) != 0;
    LkWave<SEGLEN> W;
    LkPoint P;
    lk_prologue<SEGLEN, WIN>(a, pt, lane, W, P, !templater);
    int bwd = a.backward;               // direction of the pass that runs now
#pragma nounroll
    for (;;) {
    float outx = P.gx0, outy = P.gy0;   // nextPts[i]
    int st = 1;
    float errv = 0.f;

    for (int level = a.max_level; level >= 0; level--) {
        const LkLevel G = lk_level<SEGLEN>(a, W, P, level, outx, outy);
        const int w = G.w, h = G.h;
        Region R;
        R.rx0 = 0; R.ry0 = 0; R.rwd = 0; R.rh = 0;   // nothing staged yet
        int Cw[SEGLEN];
        uint32_t GX[SEGLEN / 2], GY[SEGLEN / 2];
        float A11 = 0.f, A12 = 0.f, A22 = 0.f;

        // ---- before the hand-over: the templater builds this level's template, the tracker stages its J region ----
        if (templater) {
            int code = 1;
            float eig_value = 0.f;
            if (!G.tmpl_out) {
                const Region RI = stage_region<I_ROWS, LK_PITCH>(ilds, G.imgI, G.ipitch, w, h, G.ipx - 1, G.ipx + W.ext_w + 1, G.ipy - 1, W.wh + 3, lane, W.reach_w, W.reach_h);
                lds_fence();
                lk_template<SEGLEN, TALL>(W, ilds, RI, G.px, G.py, G.ipx, G.ipy, w, h, Cw, GX, GY, A11, A12, A22);
                code = lk_matrix_test<SEGLEN>(a, W, bwd, level, A11, A12, A22, eig_value);
#pragma unroll
                for (int k = 0; k < SEGLEN; k++) hand[k * 64 + lane] = (uint32_t)Cw[k];
#pragma unroll
                for (int m = 0; m < SEGLEN / 2; m++) {
                    hand[(SEGLEN + m) * 64 + lane] = GX[m];
                    hand[(SEGLEN + SEGLEN / 2 + m) * 64 + lane] = GY[m];
                }
            }
            if (lane == 0) { hand_f[0] = A11; hand_f[1] = A12; hand_f[2] = A22; hand_f[3] = eig_value; hand_code = code; }
        } else if (!G.tmpl_out) {
            const int inx0 = lk_uniform_floor(G.nx), iny0 = lk_uniform_floor(G.ny);
            if (!(inx0 < -W.ww || inx0 >= w || iny0 < -W.wh || iny0 >= h)) {
                R = stage_region<J_ROWS, LK_PITCH>(jlds, G.imgJ, G.ipitch, w, h, inx0 - LK_MARGIN, inx0 + W.ext_w + LK_MARGIN, iny0 - LK_MARGIN, W.wh + 1 + 2 * LK_MARGIN, lane, W.reach_w, W.reach_h);
                lds_fence();
            }
        }
        __syncthreads();   // the template of this level is in LDS
        int code = 1;
        float eig_value = 0.f;
        if (!templater) {
            code = hand_code;
            if (code == 0) {
#pragma unroll
                for (int k = 0; k < SEGLEN; k++) Cw[k] = (int)hand[k * 64 + lane];
#pragma unroll
                for (int m = 0; m < SEGLEN / 2; m++) {
                    GX[m] = hand[(SEGLEN + m) * 64 + lane];
                    GY[m] = hand[(SEGLEN + SEGLEN / 2 + m) * 64 + lane];
                }
            }
            A11 = hand_f[0]; A12 = hand_f[1]; A22 = hand_f[2]; eig_value = hand_f[3];
        }
        __syncthreads();   // ... and the tracker has taken it: the templater goes on to the next level
        if (templater) continue;

        if (code == 1) {            // the template window lies outside the image
            if (level == 0) { st = 0; errv = 0.f; }
            if (a.iters && !bwd && lane == 0) a.iters[(size_t)pt * a.nlev + level] = -1;
        } else {
            if ((a.flags & FT_LK_GET_MIN_EIGENVALS) && level == 0 && !bwd) errv = eig_value;
            if (code == 2) {        // flat or degenerate template
                if (level == 0) st = 0;
                if (a.iters && !bwd && lane == 0) a.iters[(size_t)pt * a.nlev + level] = -1;
            } else {
                lk_track_level<SEGLEN, TALL, HOLD>(a, W, bwd, pt, level, jlds, G.imgJ, G.ipitch, w, h, Cw, GX, GY, A11, A12, A22, R, G.nx, G.ny, outx, outy, st, errv);
                __builtin_amdgcn_wave_barrier();   // all LDS reads of this level precede the next level's staging
            }
        }
    }
    if (!FUSED) {
        if (!templater && lane == 0) lk_write_result(a, pt, P, outx, outy, st, errv);
        break;
    }
    if (bwd) {
        if (!templater && lane == 0) lk_write_fused(a, pt, P, turn_f, outx, outy, st);
        break;
    }
    // the forward result from the tracker to the templater (written once per feature, so one barrier is enough)
    if (!templater && lane == 0) { turn_f[0] = outx; turn_f[1] = outy; turn_f[2] = errv; turn_st = st; }
    __syncthreads();
    if (!turn_st) {
        if (!templater && lane == 0) lk_write_result(a, pt, P, outx, outy, 0, errv);
        break;
    }
    lk_turn_around(P, turn_f[0], turn_f[1]);
    bwd = 1;
    }
}

// input block of a tracking batch: pinned host memory (read through its device address) -> device memory, 16 bytes
// per lane.  A kernel instead of a copy command: enqueueing it never waits on the copy path of the runtime.
__global__ void __launch_bounds__(256) fetch_block_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// standalone point warp p' = pi(H * [p;1]) (gyro-predicted flow)
__global__ void __launch_bounds__(256) predict_flow_kernel(const float* __restrict__ H9, const float* __restrict__ xy, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = xy[2 * i], y = xy[2 * i + 1];
    const float p0 = H9[0] * x + H9[1] * y + H9[2];
    const float p1 = H9[3] * x + H9[4] * y + H9[5];
    const float p2 = H9[6] * x + H9[7] * y + H9[8];
    out[2 * i] = p0 / p2;
    out[2 * i + 1] = p1 / p2;
}

}  // namespace ftd
