// Pyramidal Lucas-Kanade for gfx950: one 64-lane wavefront per feature, all pyramid levels in one launch.
//
// Replaces cv::calcOpticalFlowPyrLK / LKTrackerInvoker as called at src/track/TrackKLT.cpp:306 and :402.
//
// Lane layout: the win_w x win_h window is cut into row segments of SEGLEN pixels; lane = row * SEGS + segment.
// Each lane keeps its SEGLEN template samples (I*32, Ix, Iy) in registers for the whole level, so every Newton
// iteration only fetches the two target-image rows the lane needs and one wave-wide reduction of (b1, b2).
// All sums are exact integers: lanes accumulate in int32, the wave reduction runs on 16-bit halves with DPP
// adds (no LDS traffic), and the total is rounded to float once.
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

// SEGLEN+1 pixels of image row y starting at column x; REFLECT_101 outside the image
template <int SEGLEN>
__device__ inline void load_row_u8(const uint8_t* __restrict__ img, uint32_t pitch, int w, int h, int x, int y, int (&out)[SEGLEN + 1]) {
    const uint8_t* row = img + (size_t)reflect101(y, h) * pitch;
    if (x >= 0 && x + SEGLEN < w) {
#pragma unroll
        for (int k = 0; k <= SEGLEN; k++) out[k] = row[x + k];
    } else {
#pragma unroll
        for (int k = 0; k <= SEGLEN; k++) out[k] = row[reflect101(x + k, w)];
    }
}

// SEGLEN+1 gradient samples of row y starting at column x; zero outside the image
template <int SEGLEN>
__device__ inline void load_row_grad(const uint8_t* __restrict__ grad, uint32_t pitch, int w, int h, int x, int y, int (&gx)[SEGLEN + 1], int (&gy)[SEGLEN + 1]) {
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

template <int SEGLEN>
__global__ void __launch_bounds__(64) lk_kernel(LKArgs a) {
    const int pt = blockIdx.x;
    const int lane = threadIdx.x;
    const LKJobDev* job = a.jobs + a.pt_job[pt];
    const int ww = a.win_w, wh = a.win_h;
    const int r = lane / a.segs, s = lane - r * a.segs;
    const int x0 = s * SEGLEN;
    const bool row_ok = r < wh;
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

    int lw[FT_MAX_LEVELS], lh[FT_MAX_LEVELS];
    lw[0] = job->w0; lh[0] = job->h0;
#pragma unroll
    for (int l = 1; l < FT_MAX_LEVELS; l++) { lw[l] = half_up(lw[l - 1]); lh[l] = half_up(lh[l - 1]); }

    float outx = gx0, outy = gy0;   // nextPts[i]
    int st = 1;
    float errv = 0.f;

    for (int level = a.max_level; level >= 0; level--) {
        int w = lw[0], h = lh[0];
#pragma unroll
        for (int l = 1; l < FT_MAX_LEVELS; l++) if (l == level) { w = lw[l]; h = lh[l]; }
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
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= w || ipy < -wh || ipy >= h) {
            if (level == 0) { st = 0; errv = 0.f; }
            if (a.iters && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;
            continue;
        }

        // ---- template: bilinear samples of I (x32) and of the Scharr gradients ----
        int Iw[SEGLEN], Gx[SEGLEN], Gy[SEGLEN];
        {
            const BilinW bw = bilin_weights(px - ipx, py - ipy);
            const uint8_t* imgI = level_img(a.A, img_I, level);
            const uint8_t* grdI = level_grad(a.A, img_I, level);
            const uint32_t ipitch = a.A.lv[level].img_pitch, gpitch = a.A.lv[level].grad_pitch;
            int r0[SEGLEN + 1], r1[SEGLEN + 1];
            load_row_u8<SEGLEN>(imgI, ipitch, w, h, ipx + x0, ipy + r, r0);
            load_row_u8<SEGLEN>(imgI, ipitch, w, h, ipx + x0, ipy + r + 1, r1);
            int gx0r[SEGLEN + 1], gy0r[SEGLEN + 1], gx1r[SEGLEN + 1], gy1r[SEGLEN + 1];
            load_row_grad<SEGLEN>(grdI, gpitch, w, h, ipx + x0, ipy + r, gx0r, gy0r);
            load_row_grad<SEGLEN>(grdI, gpitch, w, h, ipx + x0, ipy + r + 1, gx1r, gy1r);
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
            // normal matrix
            const float FLT_SCALE = 1.f / (1 << 20);
            const float A11 = (float)wave_sum_exact(a11) * FLT_SCALE;
            const float A12 = (float)wave_sum_exact(a12) * FLT_SCALE;
            const float A22 = (float)wave_sum_exact(a22) * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
            if ((a.flags & FT_LK_GET_MIN_EIGENVALS) && level == 0 && !a.backward) errv = minEig;
            if ((double)minEig < a.min_eig || D < 1.1920928955078125e-07f) {
                if (level == 0) st = 0;
                if (a.iters && lane == 0) a.iters[(size_t)pt * a.nlev + level] = n_iter;
                continue;
            }
            D = 1.f / D;

            // ---- Newton iterations on the target image ----
            const uint8_t* imgJ = level_img(a.A, img_J, level);
            nx -= half_x; ny -= half_y;
            float pdx = 0.f, pdy = 0.f;
            n_iter = 0;
            for (int j = 0; j < a.max_iter; j++) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -ww || inx >= w || iny < -wh || iny >= h) {
                    if (level == 0) st = 0;
                    break;
                }
                n_iter = j + 1;
                const BilinW jw = bilin_weights(nx - inx, ny - iny);
                int j0[SEGLEN + 1], j1[SEGLEN + 1];
                load_row_u8<SEGLEN>(imgJ, ipitch, w, h, inx + x0, iny + r, j0);
                load_row_u8<SEGLEN>(imgJ, ipitch, w, h, inx + x0, iny + r + 1, j1);
                int b1 = 0, b2 = 0;
#pragma unroll
                for (int k = 0; k < SEGLEN; k++) {
                    const int diff = descale(j0[k] * jw.w00 + j0[k + 1] * jw.w01 + j1[k] * jw.w10 + j1[k + 1] * jw.w11, 9) - Iw[k];
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
                const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
                if (iqx < -ww || iqx >= w || iqy < -wh || iqy >= h) {
                    st = 0;
                } else {
                    const BilinW qw = bilin_weights(qx - iqx, qy - iqy);
                    int j0[SEGLEN + 1], j1[SEGLEN + 1];
                    load_row_u8<SEGLEN>(imgJ, ipitch, w, h, iqx + x0, iqy + r, j0);
                    load_row_u8<SEGLEN>(imgJ, ipitch, w, h, iqx + x0, iqy + r + 1, j1);
                    int e = 0;
#pragma unroll
                    for (int k = 0; k < SEGLEN; k++) {
                        const int diff = descale(j0[k] * qw.w00 + j0[k + 1] * qw.w01 + j1[k] * qw.w10 + j1[k + 1] * qw.w11, 9) - Iw[k];
                        if (row_ok && (x0 + k < ww)) e += abs(diff);
                    }
                    const float errsum = (float)wave_sum_i32(e);   // < 2^24: exact
                    errv = errsum * 1.f / (float)(32 * ww * wh);
                }
            }
        }
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
