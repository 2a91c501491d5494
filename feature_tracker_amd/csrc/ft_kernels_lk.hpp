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
//    largest part of this kernel's memory traffic).  Samples stay in registers for the whole level.
//  * Target: every Newton iteration reads two LDS rows per lane, so the dependent chain position -> pixels ->
//    (b1,b2) -> position never waits on memory.  A window that walks out of the staged region re-stages it.
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
    const float* guess_xy;    // forward: initial guesses (unused for jobs with a homography)
    float* next_xy;           // forward: results; backward: template points (read only)
    uint8_t* status;
    float* err;
    int* iters;               // optional [n][nlev]
    int total;                // number of points
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

// ---- image regions in LDS ----
struct Region {
    int rx0, ry0;     // image coordinates of the region origin (rx0 is a multiple of 4)
    int rwd, rh;      // size in dwords x rows
};

// single-reflection REFLECT_101: exact for -len < p < 2*len - 1
__device__ inline int reflect101_once(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

constexpr int LK_REACH = FT_MAX_WIN + 1 + LK_MARGIN + 4;   // farthest a staged region extends beyond a window origin

// load image columns [x_first, x_last] x rows [y_first, y_first + n_rows) into LDS (REFLECT_101 outside the image).
// Lane (c, rr) = (lane & 15, lane >> 4) owns dword column c of rows rr, rr+4, ...  On levels larger than the reach
// of a region (all but degenerate ones) a lane issues all its loads before the first one is consumed, so a region
// costs one memory round trip.  The caller orders the LDS writes before its reads with lds_fence().
template <int MAXR>
__device__ __forceinline__ Region stage_region(uint32_t* __restrict__ lds, const uint8_t* __restrict__ img, uint32_t pitch, int w, int h,
                                               int x_first, int x_last, int y_first, int n_rows, int lane) {
    Region R;
    R.rx0 = (x_first >> 2) << 2;
    R.ry0 = y_first;
    R.rwd = (x_last - R.rx0 + 4) >> 2;     // <= LK_RW_DWORDS
    R.rh = n_rows;                         // <= 4 * MAXR
    const int c = lane & 15, rr = lane >> 4;
    const int gx = R.rx0 + 4 * c;
    const bool col_used = c < R.rwd;
    const bool col_in = gx >= 0 && gx + 3 < w;
    const bool roomy = w > 2 * LK_REACH && h > 2 * LK_REACH;   // wave-uniform
    if (roomy && (col_in || !col_used)) {
        const uint32_t xoff = col_used ? (uint32_t)gx : 0u;     // unused lanes load something harmless
        uint32_t v[MAXR];
#pragma unroll
        for (int k = 0; k < MAXR; k++) {
            const int sy = reflect101_once(R.ry0 + min(rr + 4 * k, R.rh - 1), h);
            v[k] = *reinterpret_cast<const uint32_t*>(img + ((uint32_t)sy * pitch + xoff));   // uniform base + 32-bit offset
        }
        if (col_used) {
#pragma unroll
            for (int k = 0; k < MAXR; k++)
                if (rr + 4 * k < R.rh) lds[(rr + 4 * k) * LK_RW_DWORDS + c] = v[k];
        }
    } else if (col_used) {   // dword straddles the image border, or a degenerate level: byte by byte, general reflection
#pragma unroll 1
        for (int row = rr; row < R.rh; row += 4) {
            const uint8_t* src = img + (size_t)reflect101(R.ry0 + row, h) * pitch;
            uint32_t t = 0;
#pragma unroll 1
            for (int b = 0; b < 4; b++) t |= (uint32_t)src[reflect101(gx + b, w)] << (8 * b);
            lds[row * LK_RW_DWORDS + c] = t;
        }
    }
    return R;
}

// one wave per workgroup: LDS operations of a wave complete in order, this only pins the compiler's ordering
__device__ inline void lds_fence() {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

__device__ inline bool region_holds(const Region& R, int inx, int iny, int ext_w, int win_h) {
    return inx >= R.rx0 && inx + ext_w + 1 <= R.rx0 + 4 * R.rwd && iny >= R.ry0 && iny + win_h + 1 <= R.ry0 + R.rh;
}

// N consecutive bytes of one staged row starting at byte offset byte_off, as ints
template <int N>
__device__ inline void read_row_bytes(const uint32_t* __restrict__ lds, int row, int byte_off, int (&out)[N]) {
    constexpr int NDW = (N + 3 + 3) / 4;
    const uint32_t* p = lds + row * LK_RW_DWORDS + (byte_off >> 2);
    const uint32_t sh = byte_off & 3;
    uint32_t d[NDW + 1];
#pragma unroll
    for (int q = 0; q < NDW; q++) d[q] = p[q];
    d[NDW] = 0;
#pragma unroll
    for (int q = 0; q < (N + 3) / 4; q++) {
        const uint32_t e = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (4 * q + b < N) out[4 * q + b] = (e >> (8 * b)) & 0xff;
    }
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

// SEGLEN: pixels per lane segment; TALL: window height above 21 rows (more staged rows per lane)
template <int SEGLEN, bool TALL>
__global__ void __launch_bounds__(64) lk_kernel(LKArgs a) {
    constexpr int WH_MAX = TALL ? 32 : 21;
    constexpr int I_ROWS = (WH_MAX + 3 + 3) / 4;                    // staged rows per lane, template region
    constexpr int J_ROWS = (WH_MAX + 1 + 2 * LK_MARGIN + 3) / 4;    // staged rows per lane, target region
    __shared__ uint32_t jlds[LK_RH_MAX * LK_RW_DWORDS + 4];         // target region (+ slack: row reads are dword-granular)
    __shared__ uint32_t ilds[(32 + 3) * LK_RW_DWORDS + 4];          // template region: window + Scharr apron
    const int pt = xcd_item(blockIdx.x, a.total);   // points of one image pair stay on one XCD
    if (pt >= a.total) return;
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
            gx0 = a.guess_xy[2 * pt]; gy0 = a.guess_xy[2 * pt + 1];
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

        // ---- stage both regions of this level: their loads are in flight together ----
        const uint32_t ipitch = a.A.lv[level].img_pitch;
        const uint8_t* imgI = level_img(a.A, img_I, level);
        const uint8_t* imgJ = level_img(a.A, img_J, level);
        nx -= half_x; ny -= half_y;
        const int inx0 = __builtin_amdgcn_readfirstlane((int)floorf(nx)), iny0 = __builtin_amdgcn_readfirstlane((int)floorf(ny));
        const Region RI = stage_region<I_ROWS>(ilds, imgI, ipitch, w, h, ipx - 1, ipx + ext_w + 1, ipy - 1, wh + 3, lane);
        Region R;
        R.rx0 = 0; R.ry0 = 0; R.rwd = 0; R.rh = 0;   // nothing staged yet
        if (!(inx0 < -ww || inx0 >= w || iny0 < -wh || iny0 >= h))
            R = stage_region<J_ROWS>(jlds, imgJ, ipitch, w, h, inx0 - LK_MARGIN, inx0 + ext_w + LK_MARGIN, iny0 - LK_MARGIN, wh + 1 + 2 * LK_MARGIN, lane);
        lds_fence();

        // ---- template: bilinear samples of I (x32) and of its Scharr gradients, all from the staged region ----
        int Iw[SEGLEN];
        int Gx[SEGLEN], Gy[SEGLEN];
        float A11, A12, A22;
        {
            const BilinW bw = bilin_weights(px - ipx, py - ipy);
            // the lane interpolates between image rows ipy+rc and ipy+rc+1; their gradients need rows ipy+rc-1 .. +2.
            // Rows are consumed one at a time (rolling three-row window) and folded into the weighted sums at once.
            const int boff = ipx + x0 - 1 - RI.rx0;             // columns ipx+x0-1 .. ipx+x0+SEGLEN+1
            int ra[SEGLEN + 3], rb[SEGLEN + 3], rcw[SEGLEN + 3];
            read_row_bytes<SEGLEN + 3>(ilds, rc, boff, ra);
            read_row_bytes<SEGLEN + 3>(ilds, rc + 1, boff, rb);
            int sI[SEGLEN], sX[SEGLEN], sY[SEGLEN];
#pragma unroll
            for (int k = 0; k < SEGLEN; k++) { sI[k] = rb[k + 1] * bw.w00 + rb[k + 2] * bw.w01; sX[k] = 0; sY[k] = 0; }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                read_row_bytes<SEGLEN + 3>(ilds, rc + 2 + j, boff, rcw);
                const int wl = j == 0 ? bw.w00 : bw.w10, wr = j == 0 ? bw.w01 : bw.w11;
                const int gy = ipy + rc + j;
                const bool row_in = gy >= 0 && gy < h;
                if (j == 1) {
#pragma unroll
                    for (int k = 0; k < SEGLEN; k++) sI[k] += rb[k + 1] * bw.w10 + rb[k + 2] * bw.w11;
                }
                int t0p = 0, t1p = 0, t0c = 0, t1c = 0;   // vertical sums of the two previous columns
#pragma unroll
                for (int c = 0; c < SEGLEN + 3; c++) {
                    const int t0 = (ra[c] + rcw[c]) * 3 + rb[c] * 10;
                    const int t1 = rcw[c] - ra[c];
                    if (c >= 2) {
                        const int k = c - 2;                    // sample column ipx + x0 + k
                        const int gxp = ipx + x0 + k;
                        const bool in = row_in && gxp >= 0 && gxp < w;
                        const int ix = in ? t0 - t0p : 0;                       // zero outside the image
                        const int iy = in ? (t1 + t1p) * 3 + t1c * 10 : 0;
                        if (k < SEGLEN) { sX[k] += ix * wl; sY[k] += iy * wl; }
                        if (k > 0) { sX[k - 1] += ix * wr; sY[k - 1] += iy * wr; }
                    }
                    t0p = t0c; t1p = t1c; t0c = t0; t1c = t1;
                }
#pragma unroll
                for (int c = 0; c < SEGLEN + 3; c++) { ra[c] = rb[c]; rb[c] = rcw[c]; }
            }
            int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
            for (int k = 0; k < SEGLEN; k++) {
                const bool ok = row_ok && (x0 + k < ww);
                const int ix = ok ? descale(sX[k], 14) : 0;
                const int iy = ok ? descale(sY[k], 14) : 0;
                Iw[k] = descale(sI[k], 9); Gx[k] = ix; Gy[k] = iy;
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
        float pdx = 0.f, pdy = 0.f;
        n_iter = 0;
        for (int j = 0; j < a.max_iter; j++) {
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(nx)), iny = __builtin_amdgcn_readfirstlane((int)floorf(ny));
            if (inx < -ww || inx >= w || iny < -wh || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            n_iter = j + 1;
            if (!region_holds(R, inx, iny, ext_w, wh)) {
                R = stage_region<J_ROWS>(jlds, imgJ, ipitch, w, h, inx - LK_MARGIN, inx + ext_w + LK_MARGIN, iny - LK_MARGIN, wh + 1 + 2 * LK_MARGIN, lane);
                lds_fence();
            }
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
                if (!region_holds(R, iqx, iqy, ext_w, wh)) {
                    R = stage_region<J_ROWS>(jlds, imgJ, ipitch, w, h, iqx - LK_MARGIN, iqx + ext_w + LK_MARGIN, iqy - LK_MARGIN, wh + 1 + 2 * LK_MARGIN, lane);
                    lds_fence();
                }
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
