// Pyramidal Lucas-Kanade for gfx950: one 64-lane wavefront per feature, all pyramid levels in one launch.
//
// Replaces cv::calcOpticalFlowPyrLK / LKTrackerInvoker as called at src/track/TrackKLT.cpp:306 and :402.
//
// Lane layout: the win_w x win_h window is cut into row segments of SEGLEN pixels; lane = row * SEGS + segment.
//  * Template: each lane samples its SEGLEN pixels of I (x32) and of the Scharr gradients once per level and keeps
//    them in registers.
//  * Target: the part of J the window can reach (window + a margin of LK_MARGIN pixels on every side) is staged in
//    LDS once per level with aligned dword loads (REFLECT_101 applied while staging); every Newton iteration then
//    reads two LDS rows per lane, so the dependent chain position -> pixels -> (b1,b2) -> position never waits on
//    HBM.  A window that walks out of the staged region re-stages it around the current position.
//  * Sums are exact integers: lanes accumulate in int32, the wave reduction runs on 16-bit halves with DPP adds
//    (no LDS traffic), and the total is rounded to float once, so results do not depend on the reduction order.
#pragma once
#include "ft_common.hpp"

namespace ftd {

struct LKJobDev {
    int img_prev, img_next;   // linear slot indices
    int w0, h0;               // level-0 size (both images)
    int use_H;                // 1: initial guess = pi(H * [prev;1])
    float H[9];
};

struct LKArgs {
    ArenaDesc A;
    const LKJobDev* jobs;
    const uint16_t* pt_job;   // job index of every point
    const float* prev_xy;     // template points (forward) / original points (backward)
    float* next_xy;           // forward: in = guess, out = result; backward: template points (read only)
    uint8_t* status;
    float* err;
    int* iters;               // optional [n][nlev]
    int nlev;                 // levels per iters row
    int win_w, win_h, segs;
    int max_level;            // already clamped to the pyramids
    int max_iter;
    int flags;
    int backward;             // 0: forward pass, 1: forward-backward verification pass
    float fb_thresh2;
    double epsilon;           // squared
    double min_eig;
};

constexpr int LK_MARGIN = 5;                                     // staged margin around the target window, pixels
constexpr int LK_RW_DWORDS = (32 + 1 + 2 * LK_MARGIN + 3 + 3) / 4;   // widest staged row: 12 dwords
constexpr int LK_RH_MAX = 32 + 1 + 2 * LK_MARGIN;                // tallest staged region: 43 rows

// ---- wave-wide exact integer sum (result uniform in all lanes) ----
__device__ inline int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

// sum of one int32 per lane whose total may exceed 32 bits: reduce the two 16-bit halves separately
__device__ inline double wave_sum_exact(int v) {
    const int lo = v & 0xffff, hi = v >> 16;
    const int slo = wave_sum_i32(lo), shi = wave_sum_i32(hi);
    return (double)shi * 65536.0 + (double)slo;
}

__device__ inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct BilinW { int w00, w01, w10, w11; };

__device__ inline BilinW bilin_weights(float a, float b) {
    BilinW r;
    r.w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * 16384.f);
    r.w01 = (int)__builtin_rintf(a * (1.f - b) * 16384.f);
    r.w10 = (int)__builtin_rintf((1.f - a) * b * 16384.f);
    r.w11 = 16384 - r.w00 - r.w01 - r.w10;
    return r;
}

typedef short lk_v2s __attribute__((ext_vector_type(2)));
__device__ inline int dot2_acc(uint32_t pair, uint32_t wpair, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_v2s, pair), __builtin_bit_cast(lk_v2s, wpair), acc, false);
}

struct __attribute__((packed)) U32Unaligned { uint32_t v; };

// ---- template loads: SEGLEN+1 pixels of row y from column x ----
template <int SEGLEN>
__device__ inline void load_row_u8(const uint8_t* __restrict__ img, uint32_t pitch, int w, int h, int x, int y, bool interior, int (&out)[SEGLEN + 1]) {
    if (interior) {   // wave-uniform: the whole window lies inside the image
        const uint8_t* p = img + (size_t)y * pitch + x;
#pragma unroll
        for (int q = 0; q < SEGLEN / 4; q++) {
            const uint32_t v = reinterpret_cast<const U32Unaligned*>(p + 4 * q)->v;
            out[4 * q] = v & 0xff; out[4 * q + 1] = (v >> 8) & 0xff; out[4 * q + 2] = (v >> 16) & 0xff; out[4 * q + 3] = v >> 24;
        }
        out[SEGLEN] = p[SEGLEN];
    } else {
        const uint8_t* row = img + (size_t)reflect101(y, h) * pitch;
#pragma unroll
        for (int k = 0; k <= SEGLEN; k++) out[k] = row[reflect101(x + k, w)];
    }
}

// SEGLEN+1 gradient samples of row y from column x; zero outside the image
template <int SEGLEN>
__device__ inline void load_row_grad(const uint8_t* __restrict__ grad, uint32_t pitch, int w, int h, int x, int y, bool interior, int (&gx)[SEGLEN + 1], int (&gy)[SEGLEN + 1]) {
    if (interior) {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(grad + (size_t)y * pitch) + x;
#pragma unroll
        for (int k = 0; k <= SEGLEN; k++) {
            const uint32_t v = row[k];
            gx[k] = (int)(int16_t)(v & 0xffff);
            gy[k] = (int)v >> 16;
        }
    } else {
        const bool row_in = (y >= 0 && y < h);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(grad + (size_t)(row_in ? y : 0) * pitch);
#pragma unroll
        for (int k = 0; k <= SEGLEN; k++) {
            const int xx = x + k;
            uint32_t v = 0;
            if (row_in && xx >= 0 && xx < w) v = row[xx];
            gx[k] = (int)(int16_t)(v & 0xffff);
            gy[k] = (int)v >> 16;
        }
    }
}

// ---- target region in LDS ----
struct JRegion {
    int rx0, ry0;     // image coordinates of the region origin (rx0 is a multiple of 4)
    int rwd, rh;      // size in dwords x rows
};

// stage the region that a window at integer origin (inx, iny) can reach within LK_MARGIN pixels
__device__ inline JRegion stage_region(uint32_t* __restrict__ lds, const uint8_t* __restrict__ img, uint32_t pitch, int w, int h,
                                       int inx, int iny, int ext_w, int win_h, int lane) {
    JRegion R;
    R.rx0 = ((inx - LK_MARGIN) >> 2) << 2;
    R.ry0 = iny - LK_MARGIN;
    R.rwd = (inx + ext_w + 1 + LK_MARGIN - R.rx0 + 3) >> 2;    // <= LK_RW_DWORDS
    R.rh = win_h + 1 + 2 * LK_MARGIN;                          // <= LK_RH_MAX
    const int c = lane & 15, rr = lane >> 4;
    if (c < R.rwd) {
        const int gx = R.rx0 + 4 * c;
        const bool col_in = gx >= 0 && gx + 3 < w;
        for (int row = rr; row < R.rh; row += 4) {
            const uint8_t* src = img + (size_t)reflect101(R.ry0 + row, h) * pitch;
            uint32_t v;
            if (col_in) {
                v = *reinterpret_cast<const uint32_t*>(src + gx);
            } else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) v |= (uint32_t)src[reflect101(gx + k, w)] << (8 * k);
            }
            lds[row * LK_RW_DWORDS + c] = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes are done (one wave per workgroup)
    __builtin_amdgcn_wave_barrier();
    return R;
}

__device__ inline bool region_holds(const JRegion& R, int inx, int iny, int ext_w, int win_h) {
    return inx >= R.rx0 && inx + ext_w + 1 <= R.rx0 + 4 * R.rwd && iny >= R.ry0 && iny + win_h + 1 <= R.ry0 + R.rh;
}

// (J[k], J[k+1]) pairs, k = 0..SEGLEN-1, of one staged row as packed 16-bit pairs for v_dot2
template <int SEGLEN>
__device__ inline void read_row_pairs(const uint32_t* __restrict__ lds, int row, int byte_off, uint32_t (&pairs)[SEGLEN]) {
    constexpr int NDW = SEGLEN / 4 + 1;
    const uint32_t* p = lds + row * LK_RW_DWORDS + (byte_off >> 2);
    const uint32_t sh = byte_off & 3;
    uint32_t d[NDW + 1];
#pragma unroll
    for (int q = 0; q < NDW; q++) d[q] = p[q];
    d[NDW] = 0;
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

template <int SEGLEN>
__global__ void __launch_bounds__(64) lk_kernel(LKArgs a) {
    __shared__ uint32_t jlds[LK_RH_MAX * LK_RW_DWORDS + 4];   // + slack: row reads are dword-granular
    const int pt = blockIdx.x;
    const int lane = threadIdx.x;
    const LKJobDev* job = a.jobs + a.pt_job[pt];
    const int ww = a.win_w, wh = a.win_h;
    const int r = lane / a.segs, s = lane - r * a.segs;
    const int x0 = s * SEGLEN;
    const bool row_ok = r < wh;
    const int rc = row_ok ? r : wh - 1;          // lanes below the window reuse its last row (results masked)
    const int ext_w = a.segs * SEGLEN;           // columns the lanes touch: [0, ext_w]
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;

    // forward: template = prev pyramid at prev_xy, target = next pyramid, guess from next_xy / H / prev
    // backward: template = next pyramid at the forward result, target = prev pyramid, guess = original point
    const int img_I = a.backward ? job->img_next : job->img_prev;
    const int img_J = a.backward ? job->img_prev : job->img_next;
    float tx, ty, gx0, gy0;
    bool use_init;
    if (!a.backward) {
        tx = a.prev_xy[2 * pt]; ty = a.prev_xy[2 * pt + 1];
        if (job->use_H) {
            const float* H = job->H;
            const float p0 = H[0] * tx + H[1] * ty + H[2];
            const float p1 = H[3] * tx + H[4] * ty + H[5];
            const float p2 = H[6] * tx + H[7] * ty + H[8];
            gx0 = p0 / p2; gy0 = p1 / p2;
            use_init = true;
        } else {
            gx0 = a.next_xy[2 * pt]; gy0 = a.next_xy[2 * pt + 1];
            use_init = (a.flags & FT_LK_USE_INITIAL_FLOW) != 0;
        }
    } else {
        tx = a.next_xy[2 * pt]; ty = a.next_xy[2 * pt + 1];
        gx0 = a.prev_xy[2 * pt]; gy0 = a.prev_xy[2 * pt + 1];
        use_init = true;
    }

    float outx = gx0, outy = gy0;   // nextPts[i]
    int st = 1;
    float errv = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = a.max_level; level >= 0; level--) {
        int w = job->w0, h = job->h0;
        for (int l = 0; l < level; l++) { w = half_up(w); h = half_up(h); }
        const float scale = (float)(1. / (1 << level));
        float px = tx * scale, py = ty * scale;
        float nx, ny;
        if (level == a.max_level) {
            if (use_init) { nx = outx * scale; ny = outy * scale; }
            else { nx = px; ny = py; }
        } else {
            nx = outx * 2.f; ny = outy * 2.f;
        }
        outx = nx; outy = ny;
        int n_iter = -1;

        px -= half_x; py -= half_y;
        // positions are identical in all lanes: keep the integer parts in scalar registers
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px)), ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
        if (ipx < -ww || ipx >= w || ipy < -wh || ipy >= h) {
            if (level == 0) { st = 0; errv = 0.f; }
            if (a.iters && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;
            continue;
        }

        // ---- template: bilinear samples of I (x32) and of the Scharr gradients ----
        int Iw[SEGLEN];
        int Gx[SEGLEN], Gy[SEGLEN];
        const uint32_t ipitch = a.A.lv[level].img_pitch, gpitch = a.A.lv[level].grad_pitch;
        float A11, A12, A22;
        {
            const BilinW bw = bilin_weights(px - ipx, py - ipy);
            const uint8_t* imgI = level_img(a.A, img_I, level);
            const uint8_t* grdI = level_grad(a.A, img_I, level);
            const bool interior = ipx >= 0 && ipy >= 0 && ipx + ext_w < w && ipy + wh < h;
            int r0[SEGLEN + 1], r1[SEGLEN + 1];
            load_row_u8<SEGLEN>(imgI, ipitch, w, h, ipx + x0, ipy + rc, interior, r0);
            load_row_u8<SEGLEN>(imgI, ipitch, w, h, ipx + x0, ipy + rc + 1, interior, r1);
            int gx0r[SEGLEN + 1], gy0r[SEGLEN + 1], gx1r[SEGLEN + 1], gy1r[SEGLEN + 1];
            load_row_grad<SEGLEN>(grdI, gpitch, w, h, ipx + x0, ipy + rc, interior, gx0r, gy0r);
            load_row_grad<SEGLEN>(grdI, gpitch, w, h, ipx + x0, ipy + rc + 1, interior, gx1r, gy1r);
            int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
            for (int k = 0; k < SEGLEN; k++) {
                const bool ok = row_ok && (x0 + k < ww);
                const int iv = descale(r0[k] * bw.w00 + r0[k + 1] * bw.w01 + r1[k] * bw.w10 + r1[k + 1] * bw.w11, 9);
                int ix = descale(gx0r[k] * bw.w00 + gx0r[k + 1] * bw.w01 + gx1r[k] * bw.w10 + gx1r[k + 1] * bw.w11, 14);
                int iy = descale(gy0r[k] * bw.w00 + gy0r[k + 1] * bw.w01 + gy1r[k] * bw.w10 + gy1r[k + 1] * bw.w11, 14);
                if (!ok) { ix = 0; iy = 0; }
                Iw[k] = iv; Gx[k] = ix; Gy[k] = iy;
                a11 += ix * ix; a12 += ix * iy; a22 += iy * iy;
            }
            A11 = (float)wave_sum_exact(a11) * FLT_SCALE;
            A12 = (float)wave_sum_exact(a12) * FLT_SCALE;
            A22 = (float)wave_sum_exact(a22) * FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
        if ((a.flags & FT_LK_GET_MIN_EIGENVALS) && level == 0 && !a.backward) errv = minEig;
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-07f) {
            if (level == 0) st = 0;
            if (a.iters && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;
            continue;
        }
        D = 1.f / D;

        // ---- Newton iterations on the target image, fed from LDS ----
        const uint8_t* imgJ = level_img(a.A, img_J, level);
        nx -= half_x; ny -= half_y;
        float pdx = 0.f, pdy = 0.f;
        n_iter = 0;
        JRegion R;
        R.rx0 = 0; R.ry0 = 0; R.rwd = 0; R.rh = 0;   // nothing staged yet
        for (int j = 0; j < a.max_iter; j++) {
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(nx)), iny = __builtin_amdgcn_readfirstlane((int)floorf(ny));
            if (inx < -ww || inx >= w || iny < -wh || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            n_iter = j + 1;
            if (!region_holds(R, inx, iny, ext_w, wh)) R = stage_region(jlds, imgJ, ipitch, w, h, inx, iny, ext_w, wh, lane);
            const BilinW jw = bilin_weights(nx - inx, ny - iny);
            const uint32_t wtop = (uint32_t)jw.w00 | ((uint32_t)jw.w01 << 16), wbot = (uint32_t)jw.w10 | ((uint32_t)jw.w11 << 16);
            uint32_t p0[SEGLEN], p1[SEGLEN];
            const int boff = inx - R.rx0 + x0, row = iny - R.ry0 + rc;
            read_row_pairs<SEGLEN>(jlds, row, boff, p0);
            read_row_pairs<SEGLEN>(jlds, row + 1, boff, p1);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < SEGLEN; k++) {
                const int diff = (dot2_acc(p1[k], wbot, dot2_acc(p0[k], wtop, 256)) >> 9) - Iw[k];
                b1 += diff * Gx[k];
                b2 += diff * Gy[k];
            }
            const float fb1 = (float)wave_sum_exact(b1) * FLT_SCALE;
            const float fb2 = (float)wave_sum_exact(b2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            outx = nx + half_x; outy = ny + half_y;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (a.iters && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;

        // ---- residual at the final position (level 0 only) ----
        if (st && level == 0 && !(a.flags & FT_LK_GET_MIN_EIGENVALS) && !a.backward) {
            const float qx = outx - half_x, qy = outy - half_y;
            const int iqx = __builtin_amdgcn_readfirstlane((int)floorf(qx)), iqy = __builtin_amdgcn_readfirstlane((int)floorf(qy));
            if (iqx < -ww || iqx >= w || iqy < -wh || iqy >= h) {
                st = 0;
            } else {
                if (!region_holds(R, iqx, iqy, ext_w, wh)) R = stage_region(jlds, imgJ, ipitch, w, h, iqx, iqy, ext_w, wh, lane);
                const BilinW qw = bilin_weights(qx - iqx, qy - iqy);
                const uint32_t wtop = (uint32_t)qw.w00 | ((uint32_t)qw.w01 << 16), wbot = (uint32_t)qw.w10 | ((uint32_t)qw.w11 << 16);
                uint32_t p0[SEGLEN], p1[SEGLEN];
                const int boff = iqx - R.rx0 + x0, row = iqy - R.ry0 + rc;
                read_row_pairs<SEGLEN>(jlds, row, boff, p0);
                read_row_pairs<SEGLEN>(jlds, row + 1, boff, p1);
                int e = 0;
#pragma unroll
                for (int k = 0; k < SEGLEN; k++) {
                    const int diff = (dot2_acc(p1[k], wbot, dot2_acc(p0[k], wtop, 256)) >> 9) - Iw[k];
                    if (row_ok && (x0 + k < ww)) e += abs(diff);
                }
                const float errsum = (float)wave_sum_i32(e);   // < 2^24: exact
                errv = errsum * 1.f / (float)(32 * ww * wh);
            }
        }
        __builtin_amdgcn_wave_barrier();   // all LDS reads of this level precede the next level's staging
    }

    if (lane == 0) {
        if (!a.backward) {
            a.next_xy[2 * pt] = outx;
            a.next_xy[2 * pt + 1] = outy;
            a.status[pt] = (uint8_t)st;
            a.err[pt] = errv;
        } else {
            // forward-backward consistency: the backward track must succeed and return to the original point
            const float dx = outx - gx0, dy = outy - gy0;
            const float d2 = dx * dx + dy * dy;
            if (!st || !(d2 <= a.fb_thresh2)) a.status[pt] = 0;
        }
    }
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
