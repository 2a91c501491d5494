/*
 * ft_oracle.cpp — scalar CPU restatement of the visual front-end hot path (TEST INFRASTRUCTURE).
 *
 * PARITY UNPINNED (see ft_oracle.h): OpenCV 3 is an un-vendored dependency of the reference
 * (build_pc/CMakeLists.txt:13) and is absent here, and the reference holds no golden vectors.  Every function
 * restates the generic x86 code path of the OpenCV 3.4 routine named in its comment, as called by the reference
 * at the cited site.
 *
 * One deliberate deviation, applied identically by the device kernels so both sides agree bit for bit: OpenCV
 * sums the LK normal-equation terms in float, which makes the result depend on SIMD width and summation order.
 * Every term is an exact integer, so this restatement sums them exactly (int64) and rounds to float once.
 *
 * Build with -ffp-contract=off: the float expressions below must not be fused.
 */
#include "ft_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

struct Level {
    int w = 0, h = 0, pad_x = 0, pad_y = 0;
    size_t stride = 0;
    std::vector<uint8_t> buf;
    uint8_t* origin() { return buf.data() + (size_t)pad_y * stride + pad_x; }
    const uint8_t* origin() const { return buf.data() + (size_t)pad_y * stride + pad_x; }
};

// copyMakeBorder(BORDER_REFLECT_101 | BORDER_ISOLATED): fill the padding from the level's own pixels
void fill_reflect_border(Level& L) {
    uint8_t* o = L.origin();
    for (int y = 0; y < L.h; y++) {
        uint8_t* row = o + (ptrdiff_t)y * (ptrdiff_t)L.stride;
        for (int x = 1; x <= L.pad_x; x++) {
            row[-x] = row[reflect101(-x, L.w)];
            row[L.w - 1 + x] = row[reflect101(L.w - 1 + x, L.w)];
        }
    }
    for (int y = 1; y <= L.pad_y; y++) {
        const uint8_t* top = o + (ptrdiff_t)reflect101(-y, L.h) * (ptrdiff_t)L.stride - L.pad_x;
        const uint8_t* bot = o + (ptrdiff_t)reflect101(L.h - 1 + y, L.h) * (ptrdiff_t)L.stride - L.pad_x;
        memcpy(o + (ptrdiff_t)(-y) * (ptrdiff_t)L.stride - L.pad_x, top, (size_t)L.w + 2 * L.pad_x);
        memcpy(o + (ptrdiff_t)(L.h - 1 + y) * (ptrdiff_t)L.stride - L.pad_x, bot, (size_t)L.w + 2 * L.pad_x);
    }
}

void alloc_level(Level& L, int w, int h, int pad_x, int pad_y) {
    L.w = w; L.h = h; L.pad_x = pad_x; L.pad_y = pad_y;
    L.stride = (size_t)w + 2 * (size_t)pad_x;
    L.buf.assign(L.stride * ((size_t)h + 2 * (size_t)pad_y), 0);
}

inline int cv_round(float v) { return (int)lrintf(v); }   // cvRound: round half to even
inline int cv_floor(float v) { return (int)floorf(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

}  // namespace

struct fto_pyramid {
    int win_w = 0, win_h = 0;
    std::vector<Level> lv;
};

extern "C" {

// cv::pyrDown, uchar, 1 channel: 5x5 [1 4 6 4 1]^2, (sum + 128) >> 8, REFLECT_101 on the source itself
void fto_pyr_down(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    std::vector<int> rowbuf((size_t)dw * 5);
    for (int y = 0; y < dh; y++) {
        // horizontal pass of the five contributing source rows
        for (int j = 0; j < 5; j++) {
            const uint8_t* srow = src + (size_t)reflect101(2 * y + j - 2, h) * sstride;
            int* r = rowbuf.data() + (size_t)j * dw;
            for (int x = 0; x < dw; x++) {
                int s = 0;
                for (int i = 0; i < 5; i++) s += k[i] * srow[reflect101(2 * x + i - 2, w)];
                r[x] = s;
            }
        }
        uint8_t* drow = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int s = 0;
            for (int j = 0; j < 5; j++) s += k[j] * rowbuf[(size_t)j * dw + x];
            drow[x] = (uint8_t)((s + 128) >> 8);
        }
    }
}

// cv::buildOpticalFlowPyramid without derivatives (reference call: TrackKLT.cpp:51-56)
fto_pyramid* fto_pyramid_build(const uint8_t* img, int w, int h, size_t stride, int win_w, int win_h, int max_level) {
    if (!img || w <= 0 || h <= 0 || win_w <= 2 || win_h <= 2 || max_level < 0) return nullptr;
    fto_pyramid* P = new fto_pyramid();
    P->win_w = win_w; P->win_h = win_h;
    int sw = w, sh = h;
    for (int level = 0; level <= max_level; level++) {
        P->lv.emplace_back();
        Level& L = P->lv.back();
        alloc_level(L, sw, sh, win_w, win_h);
        if (level == 0) {
            for (int y = 0; y < sh; y++) memcpy(L.origin() + (size_t)y * L.stride, img + (size_t)y * stride, (size_t)sw);
        } else {
            const Level& prev = P->lv[P->lv.size() - 2];
            fto_pyr_down(prev.origin(), prev.w, prev.h, prev.stride, L.origin(), L.stride);
        }
        fill_reflect_border(L);
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= win_w || sh <= win_h) break;   // early stop: pyramid ends at this level
    }
    return P;
}

void fto_pyramid_free(fto_pyramid* p) { delete p; }
int fto_pyramid_levels(const fto_pyramid* p) { return p ? (int)p->lv.size() : 0; }

int fto_pyramid_level(const fto_pyramid* p, int level, int* w, int* h, size_t* stride, const uint8_t** data) {
    if (!p || level < 0 || level >= (int)p->lv.size()) return -1;
    const Level& L = p->lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (stride) *stride = L.stride;
    if (data) *data = L.origin();
    return 0;
}

// calcSharrDeriv (internal to cv::calcOpticalFlowPyrLK)
void fto_scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst) {
    std::vector<int> t0v((size_t)w + 2), t1v((size_t)w + 2);
    int* t0 = t0v.data() + 1;
    int* t1 = t1v.data() + 1;
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * sstride;
        const uint8_t* r1 = src + (size_t)y * sstride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * sstride;
        for (int x = 0; x < w; x++) {
            t0[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x] = r2[x] - r0[x];
        }
        const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        t0[-1] = t0[x0]; t0[w] = t0[x1];
        t1[-1] = t1[x0]; t1[w] = t1[x1];
        int16_t* d = dst + (size_t)y * 2 * (size_t)w;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 1] - t0[x - 1]);
            d[2 * x + 1] = (int16_t)((t1[x + 1] + t1[x - 1]) * 3 + t1[x] * 10);
        }
    }
}

}  // extern "C"

namespace {

// derivative image of one level embedded in a zero border of win (copyMakeBorder BORDER_CONSTANT)
struct DerivLevel {
    int w, h, pad_x, pad_y;
    size_t stride;  // in shorts
    std::vector<int16_t> buf;
    const int16_t* origin() const { return buf.data() + (size_t)pad_y * stride + 2 * (size_t)pad_x; }
};

void make_deriv(const Level& L, int win_w, int win_h, DerivLevel& D) {
    D.w = L.w; D.h = L.h; D.pad_x = win_w; D.pad_y = win_h;
    D.stride = 2 * ((size_t)L.w + 2 * (size_t)win_w);
    D.buf.assign(D.stride * ((size_t)L.h + 2 * (size_t)win_h), 0);
    std::vector<int16_t> tmp((size_t)L.w * L.h * 2);
    fto_scharr(L.origin(), L.w, L.h, L.stride, tmp.data());
    int16_t* o = D.buf.data() + (size_t)win_h * D.stride + 2 * (size_t)win_w;
    for (int y = 0; y < L.h; y++) memcpy(o + (size_t)y * D.stride, tmp.data() + (size_t)y * 2 * L.w, sizeof(int16_t) * 2 * (size_t)L.w);
}

struct LKCtx {
    const fto_pyramid* prev;
    const fto_pyramid* next;
    const std::vector<DerivLevel>* derivs;
    const float* prev_xy;
    float* next_xy;
    uint8_t* status;
    float* err;
    int* iters;
    int nlev_out;
    int max_level, max_iter, flags, win_w, win_h;
    double epsilon;   // already squared
    double min_eig;
};

// LKTrackerInvoker::operator() for one point at one level
void lk_point_level(const LKCtx& c, int level, int i, std::vector<int16_t>& Ibuf, std::vector<int16_t>& dIbuf) {
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const Level& I = c.prev->lv[level];
    const Level& J = c.next->lv[level];
    const DerivLevel& dI = (*c.derivs)[level];
    const int ww = c.win_w, wh = c.win_h;
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;

    float px = c.prev_xy[2 * i] * (float)(1. / (1 << level));
    float py = c.prev_xy[2 * i + 1] * (float)(1. / (1 << level));
    float nx, ny;
    if (level == c.max_level) {
        if (c.flags & FTO_LK_USE_INITIAL_FLOW) {
            nx = c.next_xy[2 * i] * (float)(1. / (1 << level));
            ny = c.next_xy[2 * i + 1] * (float)(1. / (1 << level));
        } else {
            nx = px; ny = py;
        }
    } else {
        nx = c.next_xy[2 * i] * 2.f;
        ny = c.next_xy[2 * i + 1] * 2.f;
    }
    c.next_xy[2 * i] = nx;
    c.next_xy[2 * i + 1] = ny;
    if (c.iters) c.iters[(size_t)i * c.nlev_out + level] = -1;

    px -= half_x; py -= half_y;
    int ipx = cv_floor(px), ipy = cv_floor(py);
    if (ipx < -ww || ipx >= I.w || ipy < -wh || ipy >= I.h) {
        if (level == 0) { c.status[i] = 0; c.err[i] = 0; }
        return;
    }
    float a = px - ipx, b = py - ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    const size_t stepI = I.stride, stepJ = J.stride, dstep = dI.stride;
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    // FTO_LK_FLOAT_SUMS: float accumulators in the order of OpenCV's SSE2 code path (x86, acctype = float): four lanes
    // across x take the pixels x = 0..4*floor(ww/4)-1 (lane = x mod 4), a scalar accumulator takes the rest of every
    // row, and after the last row the four lanes are added to the scalar one, lane 0 first.  Every single product is
    // an integer below 2^24, so only the additions round.
    const bool fsum = (c.flags & FTO_LK_FLOAT_SUMS) != 0;
    const int ww4 = ww - ww % 4;
    float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0}, tA11 = 0.f, tA12 = 0.f, tA22 = 0.f;
    for (int y = 0; y < wh; y++) {
        const uint8_t* src = I.origin() + (ptrdiff_t)(y + ipy) * (ptrdiff_t)stepI + ipx;
        const int16_t* dsrc = dI.origin() + (ptrdiff_t)(y + ipy) * (ptrdiff_t)dstep + 2 * (ptrdiff_t)ipx;
        int16_t* Ip = Ibuf.data() + (size_t)y * ww;
        int16_t* dIp = dIbuf.data() + (size_t)y * ww * 2;
        for (int x = 0; x < ww; x++) {
            int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS - 5);
            int ixval = descale(dsrc[2 * x] * iw00 + dsrc[2 * x + 2] * iw01 + dsrc[2 * x + dstep] * iw10 + dsrc[2 * x + dstep + 2] * iw11, W_BITS);
            int iyval = descale(dsrc[2 * x + 1] * iw00 + dsrc[2 * x + 3] * iw01 + dsrc[2 * x + dstep + 1] * iw10 + dsrc[2 * x + dstep + 3] * iw11, W_BITS);
            Ip[x] = (int16_t)ival;
            dIp[2 * x] = (int16_t)ixval;
            dIp[2 * x + 1] = (int16_t)iyval;
            sA11 += (int64_t)ixval * ixval;
            sA12 += (int64_t)ixval * iyval;
            sA22 += (int64_t)iyval * iyval;
            if (fsum) {
                const float fx = (float)ixval, fy = (float)iyval;
                if (x < ww4) { qA11[x & 3] += fx * fx; qA12[x & 3] += fx * fy; qA22[x & 3] += fy * fy; }
                else { tA11 += (float)(ixval * ixval); tA12 += (float)(ixval * iyval); tA22 += (float)(iyval * iyval); }
            }
        }
    }
    float A11 = (float)(double)sA11 * FLT_SCALE;
    float A12 = (float)(double)sA12 * FLT_SCALE;
    float A22 = (float)(double)sA22 * FLT_SCALE;
    if (fsum) {
        tA11 += qA11[0] + qA11[1] + qA11[2] + qA11[3];
        tA12 += qA12[0] + qA12[1] + qA12[2] + qA12[3];
        tA22 += qA22[0] + qA22[1] + qA22[2] + qA22[3];
        A11 = tA11 * FLT_SCALE; A12 = tA12 * FLT_SCALE; A22 = tA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
    if ((c.flags & FTO_LK_GET_MIN_EIGENVALS) && level == 0) c.err[i] = minEig;
    if ((double)minEig < c.min_eig || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {
        if (level == 0) c.status[i] = 0;
        return;
    }
    D = 1.f / D;
    nx -= half_x; ny -= half_y;
    float pdx = 0.f, pdy = 0.f;
    int it = 0;
    for (int j = 0; j < c.max_iter; j++) {
        int inx = cv_floor(nx), iny = cv_floor(ny);
        if (inx < -ww || inx >= J.w || iny < -wh || iny >= J.h) {
            if (level == 0) c.status[i] = 0;
            break;
        }
        it = j + 1;
        a = nx - inx; b = ny - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t sb1 = 0, sb2 = 0;
        // FTO_LK_FLOAT_SUMS: OpenCV's SSE2 path takes eight pixels at a time; pixels k and k+4 of a group are
        // multiplied and added as integers (pmaddwd), the pair sum is converted to float and added to one of four float
        // lanes per component (k = 0, 1 -> first register, k = 2, 3 -> second register); the rest of the row goes to a
        // scalar float accumulator; at the end the two registers are added and their lanes folded into the scalars.
        const int ww8 = ww - ww % 8;
        float qb1[4] = {0, 0, 0, 0}, qb2[4] = {0, 0, 0, 0}, tb1 = 0.f, tb2 = 0.f;
        for (int y = 0; y < wh; y++) {
            const uint8_t* Jp = J.origin() + (ptrdiff_t)(y + iny) * (ptrdiff_t)stepJ + inx;
            const int16_t* Ip = Ibuf.data() + (size_t)y * ww;
            const int16_t* dIp = dIbuf.data() + (size_t)y * ww * 2;
            int dgrp[8];
            for (int x = 0; x < ww; x++) {
                int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                sb1 += (int64_t)diff * dIp[2 * x];
                sb2 += (int64_t)diff * dIp[2 * x + 1];
                if (fsum) {
                    if (x < ww8) {
                        dgrp[x & 7] = diff;
                        if ((x & 7) == 7) {
                            const int16_t* g = dIp + 2 * (x - 7);
                            for (int k = 0; k < 4; k++) {
                                qb1[k] += (float)(dgrp[k] * g[2 * k] + dgrp[k + 4] * g[2 * (k + 4)]);
                                qb2[k] += (float)(dgrp[k] * g[2 * k + 1] + dgrp[k + 4] * g[2 * (k + 4) + 1]);
                            }
                        }
                    } else {
                        tb1 += (float)(diff * dIp[2 * x]);
                        tb2 += (float)(diff * dIp[2 * x + 1]);
                    }
                }
            }
        }
        float b1 = (float)(double)sb1 * FLT_SCALE;
        float b2 = (float)(double)sb2 * FLT_SCALE;
        if (fsum) {
            // registers: {b1(k=0), b2(k=0), b1(k=1), b2(k=1)} + {b1(k=2), b2(k=2), b1(k=3), b2(k=3)}, then lanes 0+2 and 1+3
            tb1 += (qb1[0] + qb1[2]) + (qb1[1] + qb1[3]);
            tb2 += (qb2[0] + qb2[2]) + (qb2[1] + qb2[3]);
            b1 = tb1 * FLT_SCALE; b2 = tb2 * FLT_SCALE;
        }
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        c.next_xy[2 * i] = nx + half_x;
        c.next_xy[2 * i + 1] = ny + half_y;
        if ((double)dx * (double)dx + (double)dy * (double)dy <= c.epsilon) break;
        if (j > 0 && std::abs((double)(dx + pdx)) < 0.01 && std::abs((double)(dy + pdy)) < 0.01) {
            c.next_xy[2 * i] -= dx * 0.5f;
            c.next_xy[2 * i + 1] -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
    if (c.iters) c.iters[(size_t)i * c.nlev_out + level] = it;

    if (c.status[i] && level == 0 && !(c.flags & FTO_LK_GET_MIN_EIGENVALS)) {
        float qx = c.next_xy[2 * i] - half_x, qy = c.next_xy[2 * i + 1] - half_y;
        int iqx = cv_floor(qx), iqy = cv_floor(qy);
        if (iqx < -ww || iqx >= J.w || iqy < -wh || iqy >= J.h) {
            c.status[i] = 0;
            return;
        }
        if (c.flags & FTO_LK_NO_ERROR) return;   // the caller does not read the residual (it stays 0)
        float aa = qx - iqx, bb = qy - iqy;
        iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        float errval = 0.f;   // every partial sum is an integer below 2^24: exact in float for win <= 32x32
        for (int y = 0; y < wh; y++) {
            const uint8_t* Jp = J.origin() + (ptrdiff_t)(y + iqy) * (ptrdiff_t)stepJ + iqx;
            const int16_t* Ip = Ibuf.data() + (size_t)y * ww;
            for (int x = 0; x < ww; x++) {
                int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                errval += std::abs((float)diff);
            }
        }
        c.err[i] = errval * 1.f / (float)(32 * ww * wh);
    }
}

int lk_one_direction(const fto_pyramid* prev, const fto_pyramid* next, const float* prev_xy, float* next_xy,
                     uint8_t* status, float* err, int n, const fto_lk_params* prm, int n_threads, int* iters_out) {
    int max_level = std::min(prm->max_level, std::min((int)prev->lv.size(), (int)next->lv.size()) - 1);
    if (max_level < 0) return -1;
    const int nlev_out = std::min(prm->max_level, (int)prev->lv.size() - 1) + 1;
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; }
    if (iters_out) for (size_t k = 0; k < (size_t)n * nlev_out; k++) iters_out[k] = -1;

    // calcSharrDeriv of every level of the prev pyramid, once per call
    std::vector<DerivLevel> derivs((size_t)max_level + 1);
    for (int l = 0; l <= max_level; l++) make_deriv(prev->lv[l], prm->win_w, prm->win_h, derivs[l]);

    LKCtx c;
    c.prev = prev; c.next = next; c.derivs = &derivs;
    c.prev_xy = prev_xy; c.next_xy = next_xy; c.status = status; c.err = err; c.iters = iters_out; c.nlev_out = nlev_out;
    c.max_level = max_level;
    c.max_iter = std::min(std::max(prm->max_iter, 0), 100);
    double eps = std::min(std::max(prm->eps, 0.), 10.);
    c.epsilon = eps * eps;
    c.min_eig = prm->min_eig;
    c.flags = prm->flags; c.win_w = prm->win_w; c.win_h = prm->win_h;

    auto work = [&](int begin, int end) {
        std::vector<int16_t> Ibuf((size_t)c.win_w * c.win_h), dIbuf((size_t)c.win_w * c.win_h * 2);
        for (int level = max_level; level >= 0; level--)
            for (int i = begin; i < end; i++) lk_point_level(c, level, i, Ibuf, dIbuf);
    };
    if (n_threads <= 1 || n < 2 * n_threads) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) {
            int b = (int)((int64_t)n * t / n_threads), e = (int)((int64_t)n * (t + 1) / n_threads);
            th.emplace_back(work, b, e);
        }
        for (auto& t : th) t.join();
    }
    return max_level;
}

}  // namespace

extern "C" int fto_lk_track(const fto_pyramid* prev, const fto_pyramid* next, const float* prev_xy, float* next_xy,
                            uint8_t* status, float* err, int n, const fto_lk_params* prm, int n_threads, int* iters_out) {
    if (!prev || !next || !prm || n < 0) return -1;
    if (prm->win_w <= 2 || prm->win_h <= 2) return -1;
    if (prev->win_w < prm->win_w || prev->win_h < prm->win_h || next->win_w < prm->win_w || next->win_h < prm->win_h) return -2;
    if (n == 0) return 0;
    std::vector<float> guess_unused;
    fto_lk_params p = *prm;
    p.flags &= ~FTO_LK_FORWARD_BACKWARD;
    int rc = lk_one_direction(prev, next, prev_xy, next_xy, status, err, n, &p, n_threads, iters_out);
    if (rc < 0) return rc;
    if (prm->flags & FTO_LK_FORWARD_BACKWARD) {
        // build-defined consistency check: track next -> prev starting from the original point
        std::vector<float> back(prev_xy, prev_xy + 2 * (size_t)n);
        std::vector<uint8_t> st_b((size_t)n);
        std::vector<float> err_b((size_t)n);
        p.flags |= FTO_LK_USE_INITIAL_FLOW;
        p.flags &= ~FTO_LK_GET_MIN_EIGENVALS;
        int rc2 = lk_one_direction(next, prev, next_xy, back.data(), st_b.data(), err_b.data(), n, &p, n_threads, nullptr);
        if (rc2 < 0) return rc2;
        const float thr2 = prm->fb_thresh * prm->fb_thresh;
        for (int i = 0; i < n; i++) {
            float dx = back[2 * i] - prev_xy[2 * i], dy = back[2 * i + 1] - prev_xy[2 * i + 1];
            float d2 = dx * dx + dy * dy;
            if (!st_b[i] || !(d2 <= thr2)) status[i] = 0;
        }
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------------------
// FAST-9/16 (cv::FAST_t<16> + cornerScore<16>)
// ------------------------------------------------------------------------------------------------------------
namespace {

const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// segment test: 9 contiguous circle pixels all darker than v-t or all brighter than v+t
bool fast_is_corner(const uint8_t* p, size_t stride, int t) {
    const int v = p[0];
    int cls[25];
    for (int k = 0; k < 25; k++) {
        const int q = p[(ptrdiff_t)kCircle[k & 15][1] * (ptrdiff_t)stride + kCircle[k & 15][0]];
        cls[k] = q < v - t ? 1 : (q > v + t ? 2 : 0);
    }
    for (int want = 1; want <= 2; want++) {
        int run = 0;
        for (int k = 0; k < 25; k++) {
            if (cls[k] == want) { if (++run > 8) return true; }
            else run = 0;
        }
    }
    return false;
}

// cornerScore<16>: largest threshold for which the pixel is still a corner
int fast_corner_score(const uint8_t* p, size_t stride, int threshold) {
    const int v = p[0];
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - p[(ptrdiff_t)kCircle[k & 15][1] * (ptrdiff_t)stride + kCircle[k & 15][0]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int m = 4; m <= 8; m++) a = std::min(a, d[k + m]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        for (int m = 6; m <= 8; m++) b = std::max(b, d[k + m]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

void fast_detect(const uint8_t* img, int w, int h, size_t stride, int threshold, bool nonmax, std::vector<fto_keypoint>& out) {
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return;
    // three rolling score rows, as in FAST_t: a corner of row i-1 is emitted while row i is scanned
    std::vector<uint8_t> sc((size_t)w * 3, 0);
    std::vector<std::vector<int>> cpos(3);
    for (int i = 3; i < h - 2; i++) {
        uint8_t* curr = sc.data() + (size_t)((i - 3) % 3) * w;
        std::vector<int>& ccorn = cpos[(i - 3) % 3];
        memset(curr, 0, (size_t)w);
        ccorn.clear();
        if (i < h - 3) {
            const uint8_t* row = img + (size_t)i * stride;
            for (int j = 3; j < w - 3; j++) {
                if (fast_is_corner(row + j, stride, threshold)) {
                    ccorn.push_back(j);
                    if (nonmax) curr[j] = (uint8_t)fast_corner_score(row + j, stride, threshold);
                }
            }
        }
        if (i == 3) continue;
        const uint8_t* prev = sc.data() + (size_t)((i - 4 + 3) % 3) * w;
        const uint8_t* pprev = sc.data() + (size_t)((i - 5 + 3) % 3) * w;
        const std::vector<int>& pc = cpos[(i - 4 + 3) % 3];
        for (int j : pc) {
            int score = prev[j];
            if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                            score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                out.push_back(fto_keypoint{(float)j, (float)(i - 1), (float)score});
            }
        }
    }
}

}  // namespace

extern "C" {

int fto_fast(const uint8_t* img, int w, int h, size_t stride, int threshold, int nonmax, fto_keypoint* out, int cap) {
    std::vector<fto_keypoint> v;
    fast_detect(img, w, h, stride, threshold, nonmax != 0, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

void fto_fast_score_map(const uint8_t* img, int w, int h, size_t stride, int threshold, uint8_t* score, size_t sc_stride) {
    threshold = std::min(std::max(threshold, 0), 255);
    for (int y = 0; y < h; y++) memset(score + (size_t)y * sc_stride, 0, (size_t)w);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, stride, threshold)) score[(size_t)y * sc_stride + x] = (uint8_t)fast_corner_score(p, stride, threshold);
        }
}

// Grider_FAST::perform_griding (Grider_FAST.h:76-132) with the per-cell keep count given directly
int fto_grid_fast_cells(const uint8_t* img, int w, int h, size_t stride, int per_cell, int grid_x, int grid_y, int threshold,
                        int nonmax, int tie_mode, int n_threads, fto_keypoint* out, int cap, int* cell_counts) {
    if (grid_x <= 0 || grid_y <= 0 || per_cell < 0) return -1;
    const int size_x = w / grid_x, size_y = h / grid_y;                  // :80-81
    if (size_x <= 0 || size_y <= 0) return -1;
    const int ct_cols = w / size_x, ct_rows = h / size_y;               // :91-92
    std::vector<std::vector<fto_keypoint>> collection((size_t)ct_cols * ct_rows);
    std::vector<int> found((size_t)ct_cols * ct_rows, 0);
    auto cell = [&](int r) {
        const int x = r % ct_cols * size_x, y = r / ct_cols * size_y;   // :98-99
        if (x + size_x > w || y + size_y > h) return;                   // :102-103
        std::vector<fto_keypoint> pts;
        fast_detect(img + (size_t)y * stride + x, size_x, size_y, stride, threshold, nonmax != 0, pts);   // :110
        found[r] = (int)pts.size();
        if (tie_mode == 1) {
            std::sort(pts.begin(), pts.end(), [](fto_keypoint a, fto_keypoint b) { return a.response > b.response; });   // :113
        } else {
            std::sort(pts.begin(), pts.end(), [](const fto_keypoint& a, const fto_keypoint& b) {
                if (a.response != b.response) return a.response > b.response;
                if (a.y != b.y) return a.y < b.y;
                return a.x < b.x;
            });
        }
        for (size_t i = 0; i < (size_t)per_cell && i < pts.size(); i++) {   // :118-123
            fto_keypoint k = pts[i];
            k.x += (float)x; k.y += (float)y;
            collection[r].push_back(k);
        }
    };
    const int ncell = ct_cols * ct_rows;
    if (n_threads <= 1) {
        for (int r = 0; r < ncell; r++) cell(r);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++)
            th.emplace_back([&, t]() { for (int r = t; r < ncell; r += n_threads) cell(r); });
        for (auto& t : th) t.join();
    }
    int n = 0;
    for (auto& c : collection)                                           // :128-130
        for (auto& k : c) { if (n < cap) out[n] = k; n++; }
    if (cell_counts) for (int r = 0; r < ncell; r++) cell_counts[r] = found[r];
    return std::min(n, cap);
}

int fto_grid_fast(const uint8_t* img, int w, int h, size_t stride, int num_features, int grid_x, int grid_y,
                  int threshold, int nonmax, int tie_mode, int n_threads, fto_keypoint* out, int cap) {
    if (grid_x <= 0 || grid_y <= 0) return -1;
    const int per_cell = num_features / (grid_x * grid_y) + 1;          // :88
    return fto_grid_fast_cells(img, w, h, stride, per_cell, grid_x, grid_y, threshold, nonmax, tie_mode, n_threads, out, cap, nullptr);
}

// mean-gyro rotation (TrackPropagator.h:107-153)
int fto_integrate_gyro(const double* imu_t, const double* gyro, int n, double t_last, double t_cur,
                       const double* R_imu_to_cam, float* R_p_c) {
    int b = 0;
    while (b < n && (imu_t[b] - t_last) < -0.01) b++;                    // :113-118
    int e = b;
    while (e < n && (imu_t[e] - t_cur) < 0.005) e++;                     // :121-126, end test on the end iterator
    float m[3] = {0.f, 0.f, 0.f};
    for (int k = b; k < e; k++) for (int a = 0; a < 3; a++) m[a] += (float)gyro[3 * k + a];   // :128-130
    if (e - b > 0) { float s = 1.0f / (float)(e - b); for (int a = 0; a < 3; a++) m[a] *= s; }   // :132-133
    // angular velocity in the camera frame (:137-138): (R_cam_imu)^T * w = R_imu_to_cam * w, evaluated in float
    float wc[3];
    for (int r = 0; r < 3; r++)
        wc[r] = (float)R_imu_to_cam[3 * r] * m[0] + (float)R_imu_to_cam[3 * r + 1] * m[1] + (float)R_imu_to_cam[3 * r + 2] * m[2];
    const double dt = t_cur - t_last;                                    // :141
    // Vec3f * double: each component is multiplied in double and stored back as float (:142)
    const double rx = (double)(float)((double)wc[0] * dt), ry = (double)(float)((double)wc[1] * dt), rz = (double)(float)((double)wc[2] * dt);
    // cv::Rodrigues (computed in double, stored as float), then transposed (:142-145)
    const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    double R[9];
    if (theta < 2.220446049250313e-16) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
        const double cth = std::cos(theta), sth = std::sin(theta), c1 = 1. - cth;
        const double kx = rx / theta, ky = ry / theta, kz = rz / theta;
        R[0] = cth + c1 * kx * kx;      R[1] = c1 * kx * ky - sth * kz; R[2] = c1 * kx * kz + sth * ky;
        R[3] = c1 * kx * ky + sth * kz; R[4] = cth + c1 * ky * ky;      R[5] = c1 * ky * kz - sth * kx;
        R[6] = c1 * kx * kz - sth * ky; R[7] = c1 * ky * kz + sth * kx; R[8] = cth + c1 * kz * kz;
    }
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R_p_c[3 * r + cc] = (float)R[3 * cc + r];
    return e - b;
}

void fto_homography(const float* K4, const float* R, float* H) {
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    // M = R * K^-1 with K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]
    const float ifx = 1.f / fx, ify = 1.f / fy, ncx = -cx / fx, ncy = -cy / fy;
    float M[9];
    for (int r = 0; r < 3; r++) {
        M[3 * r] = R[3 * r] * ifx;
        M[3 * r + 1] = R[3 * r + 1] * ify;
        M[3 * r + 2] = R[3 * r] * ncx + R[3 * r + 1] * ncy + R[3 * r + 2];
    }
    // H = K * M
    for (int c = 0; c < 3; c++) {
        H[c] = fx * M[c] + cx * M[6 + c];
        H[3 + c] = fy * M[3 + c] + cy * M[6 + c];
        H[6 + c] = M[6 + c];
    }
}

void fto_predict_flow(const float* K4, const float* R_p_c, const float* xy, float* xy_out, int n) {
    float H[9];
    fto_homography(K4, R_p_c, H);
    for (int i = 0; i < n; i++) {
        const float x = xy[2 * i], y = xy[2 * i + 1];
        const float p0 = H[0] * x + H[1] * y + H[2];
        const float p1 = H[3] * x + H[4] * y + H[5];
        const float p2 = H[6] * x + H[7] * y + H[8];
        xy_out[2 * i] = p0 / p2;
        xy_out[2 * i + 1] = p1 / p2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// outlier rejection (see ft_oracle.h)
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// what TrackBase::undistort_point does (cv::undistortPoints for one point): remove K, five fixed-point iterations of
// the inverse radial-tangential model
inline void undistort_to_double(float u, float v, const double* c, double& x, double& y) {
    const double ifx = 1. / c[0], ify = 1. / c[1];
    const double x0 = ((double)u - c[2]) * ifx, y0 = ((double)v - c[3]) * ify;
    x = x0; y = y0;
    if (c[4] != 0. || c[5] != 0. || c[6] != 0. || c[7] != 0.) {
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1. + (c[5] * r2 + c[4]) * r2);
            const double dx = 2. * c[6] * x * y + c[7] * (r2 + 2. * x * x);
            const double dy = c[6] * (r2 + 2. * y * y) + 2. * c[7] * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    }
}

// ... with the result rounded to float like cv::Point2f
inline void undistort_to_float(float u, float v, const double* c, float& xn, float& yn) {
    double x, y;
    undistort_to_double(u, v, c, x, y);
    xn = (float)x; yn = (float)y;
}

// Null vector of the 8x9 system by Gauss-Jordan elimination with complete pivoting; false if a pivot is below 1e-12.
// Rows and columns are never moved: a step takes the largest remaining entry (ties: the lowest row, then the lowest
// column), clears its column in every other row, and retires its row and column.  After eight steps every row holds
// its own pivot and the one column that was never chosen, whose unknown is set to 1.
bool null_vector_8x9(double (&A)[8][9], double (&f)[9]) {
    bool row_used[8] = {false, false, false, false, false, false, false, false};
    bool col_used[9] = {false, false, false, false, false, false, false, false, false};
    int piv_col[8];
    for (int step = 0; step < 8; step++) {
        int pr = -1, pc = -1;
        double best = -1.;
        for (int i = 0; i < 8; i++) {
            if (row_used[i]) continue;
            for (int j = 0; j < 9; j++) {
                if (col_used[j]) continue;
                const double a = std::fabs(A[i][j]);
                if (a > best) { best = a; pr = i; pc = j; }   // first maximum in row-major order
            }
        }
        if (!(best >= 1e-12)) return false;
        row_used[pr] = true; col_used[pc] = true; piv_col[pr] = pc;
        const double inv = 1. / A[pr][pc];
        for (int i = 0; i < 8; i++) {
            if (i == pr) continue;
            const double m = A[i][pc] * inv;
            for (int j = 0; j < 9; j++)
                if (!col_used[j]) A[i][j] = A[i][j] - m * A[pr][j];
        }
    }
    int free_col = 0;
    for (int j = 0; j < 9; j++) if (!col_used[j]) free_col = j;
    for (int i = 0; i < 8; i++) f[piv_col[i]] = -A[i][free_col] / A[i][piv_col[i]];
    f[free_col] = 1.;
    return true;
}

inline bool is_inlier(const double (&F)[9], float x1, float y1, float x2, float y2, float thr2) {
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const float err = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
    return err <= thr2;
}

// Hypotheses to try before a sample of eight inliers has been drawn with probability 0.999 (the confidence handed to
// cv::findFundamentalMat at src/track/TrackKLT.cpp:418), given the best model so far has `inliers` of n points: the
// smallest N with (1 - w^8)^N <= 0.001, at most max_h.  OpenCV's RANSAC lowers its iteration count in the same manner
// whenever it finds a better model, but with w^7 (its minimal sample has seven points; below 15 points it runs LMedS
// instead); here samples have eight points and the count comes from plain multiplications in a fixed order.
int hypotheses_needed(int inliers, int n, int max_h) {
    const double w = (double)inliers / (double)n;
    const double w2 = w * w, w4 = w2 * w2, w8 = w4 * w4;
    const double q = 1. - w8;
    double p = 1.;
    int N = 0;
    while (N < max_h) {
        p = p * q;
        N++;
        if (p <= 0.001) break;
    }
    return N;
}

}  // namespace

extern "C" {

int fto_reject_outliers(const float* prev_xy, const float* next_xy, uint8_t* status, int n, const double* calib8_prev,
                        const double* calib8_next, int hypotheses, uint32_t seed, double* F_out) {
    if (!prev_xy || !next_xy || !status || !calib8_prev || !calib8_next || n < 0) return -1;
    if (hypotheses <= 0) hypotheses = 1000;   // OpenCV's iteration limit
    if (n < 8) {
        for (int i = 0; i < n; i++) status[i] = 0;
        return -1;
    }
    std::vector<float> m((size_t)n * 4);   // x1 y1 x2 y2, normalized
    for (int i = 0; i < n; i++) {
        undistort_to_float(prev_xy[2 * i], prev_xy[2 * i + 1], calib8_prev, m[4 * i], m[4 * i + 1]);
        undistort_to_float(next_xy[2 * i], next_xy[2 * i + 1], calib8_next, m[4 * i + 2], m[4 * i + 3]);
    }
    const double fmax = std::max(std::max(calib8_prev[0], calib8_prev[1]), std::max(calib8_next[0], calib8_next[1]));
    const float thr = (float)(1. / fmax), thr2 = thr * thr;
    int best_h = -1, best_count = 7;   // a model needs at least 8 inliers
    double best_F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int stop = hypotheses;   // lowered whenever a better model is found
    for (int h = 0; h < stop; h++) {
        int idx[8];
        uint32_t draw = 0;
        for (int k = 0; k < 8; k++) {
            for (;;) {
                const uint32_t r = mix32(seed ^ mix32((uint32_t)h * 0x9E3779B9u + 0x85EBCA6Bu * draw));
                draw++;
                const int cand = (int)(r % (uint32_t)n);
                bool dup = false;
                for (int q = 0; q < k; q++) dup |= idx[q] == cand;
                if (!dup) { idx[k] = cand; break; }
            }
        }
        double A[8][9], F[9];
        for (int k = 0; k < 8; k++) {
            const double x1 = m[4 * idx[k]], y1 = m[4 * idx[k] + 1], x2 = m[4 * idx[k] + 2], y2 = m[4 * idx[k] + 3];
            A[k][0] = x2 * x1; A[k][1] = x2 * y1; A[k][2] = x2;
            A[k][3] = y2 * x1; A[k][4] = y2 * y1; A[k][5] = y2;
            A[k][6] = x1;      A[k][7] = y1;      A[k][8] = 1.;
        }
        if (!null_vector_8x9(A, F)) continue;
        int count = 0;
        for (int i = 0; i < n; i++) count += is_inlier(F, m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3], thr2) ? 1 : 0;
        if (count > best_count) {
            best_count = count; best_h = h;
            for (int q = 0; q < 9; q++) best_F[q] = F[q];
            stop = std::min(stop, hypotheses_needed(count, n, hypotheses));
        }
    }
    if (best_h < 0) {
        for (int i = 0; i < n; i++) status[i] = 0;
        return -1;
    }
    for (int i = 0; i < n; i++)
        if (!is_inlier(best_F, m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3], thr2)) status[i] = 0;
    if (F_out) for (int q = 0; q < 9; q++) F_out[q] = best_F[q];
    return best_h;
}

void fto_stereo_seed(const float* cam0_xy, int n, const double* c0, const double* c1, const double* Rt, float* cam1_xy) {
    for (int i = 0; i < n; i++) {
        double x, y;
        undistort_to_double(cam0_xy[2 * i], cam0_xy[2 * i + 1], c0, x, y);
        const double X = Rt[0] * x + Rt[1] * y + Rt[2], Y = Rt[3] * x + Rt[4] * y + Rt[5], W = Rt[6] * x + Rt[7] * y + Rt[8];
        const double iw = 1. / W;
        // cv::Point2f between the two calls.  Through volatile: g++ 11 -O3 otherwise drops this rounding when it
        // vectorizes the loop.
        volatile float xf = (float)(X * iw), yf = (float)(Y * iw);
        const double xr = (double)xf, yr = (double)yf;
        const double r2 = xr * xr + yr * yr, r4 = r2 * r2;
        const double cdist = 1. + c1[4] * r2 + c1[5] * r4;
        const double a1 = 2. * xr * yr, a2 = r2 + 2. * xr * xr, a3 = r2 + 2. * yr * yr;
        const double xd = xr * cdist + c1[6] * a1 + c1[7] * a2;
        const double yd = yr * cdist + c1[6] * a3 + c1[7] * a1;
        cam1_xy[2 * i] = (float)(xd * c1[0] + c1[2]);
        cam1_xy[2 * i + 1] = (float)(yd * c1[1] + c1[3]);
    }
}

void fto_stereo_check(const float* cam0_xy, const float* cam1_xy, uint8_t* status, int n, const double* c0, const double* c1,
                      const double* Rt, int w, int h, double threshold) {
    const double* R = Rt;
    const double* t = Rt + 9;
    // E = [t]x R
    const double E[9] = {-t[2] * R[3] + t[1] * R[6], -t[2] * R[4] + t[1] * R[7], -t[2] * R[5] + t[1] * R[8],
                         t[2] * R[0] - t[0] * R[6],  t[2] * R[1] - t[0] * R[7],  t[2] * R[2] - t[0] * R[8],
                         -t[1] * R[0] + t[0] * R[3], -t[1] * R[1] + t[0] * R[4], -t[1] * R[2] + t[0] * R[5]};
    const double unit = 4.0 / (c0[0] + c0[1] + c1[0] + c1[1]);
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;
        const float nx = cam1_xy[2 * i], ny = cam1_xy[2 * i + 1];
        if (ny < 0.f || ny > (float)(h - 1) || nx < 0.f || nx > (float)(w - 1)) { status[i] = 0; continue; }
        if (!(threshold > 0.)) continue;
        float fx0, fy0, fx1, fy1;
        undistort_to_float(cam0_xy[2 * i], cam0_xy[2 * i + 1], c0, fx0, fy0);
        undistort_to_float(nx, ny, c1, fx1, fy1);
        const double x0 = fx0, y0 = fy0, x1 = fx1, y1 = fy1;
        const double l0 = E[0] * x0 + E[1] * y0 + E[2], l1 = E[3] * x0 + E[4] * y0 + E[5], l2 = E[6] * x0 + E[7] * y0 + E[8];
        const double error = std::fabs(x1 * l0 + y1 * l1 + l2) / std::sqrt(l0 * l0 + l1 * l1);
        if (error > threshold * unit) status[i] = 0;
    }
}

int fto_compute_descriptors(const uint8_t* img, int w, int h, size_t stride, const float* xy, int n, uint8_t* desc, uint8_t* valid) {
    if (!img || n < 0 || (n > 0 && (!xy || !desc || !valid))) return -1;
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    auto off = [&](int i, int k) { return (int)(mix(0xB5297A4Du + 4u * (uint32_t)i + (uint32_t)k) % 31u) - 15; };
    // 5x5 binomial smoothing at one pixel (the whole window lies inside the image for every sample of a valid keypoint)
    auto smooth = [&](int x, int y) {
        static const int k5[5] = {1, 4, 6, 4, 1};
        int s = 0;
        for (int j = -2; j <= 2; j++)
            for (int i = -2; i <= 2; i++) s += k5[j + 2] * k5[i + 2] * img[(size_t)(y + j) * stride + (size_t)(x + i)];
        return (s + 128) >> 8;
    };
    for (int p = 0; p < n; p++) {
        const int cx = (int)xy[2 * p], cy = (int)xy[2 * p + 1];
        uint8_t* d = desc + (size_t)p * 32;
        for (int b = 0; b < 32; b++) d[b] = 0;
        if (cx < 17 || cy < 17 || cx >= w - 17 || cy >= h - 17) { valid[p] = 0; continue; }
        valid[p] = 1;
        for (int i = 0; i < 256; i++) {
            const int a = smooth(cx + off(i, 0), cy + off(i, 1)), b = smooth(cx + off(i, 2), cy + off(i, 3));
            if (a < b) d[i >> 3] |= (uint8_t)(1u << (i & 7));
        }
    }
    return 0;
}

int fto_match_descriptors(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, int* out_qtd, int cap) {
    struct Two { int i1 = -1, d1 = 0, i2 = -1, d2 = 0; };
    auto knn2 = [](const uint8_t* q, int nq, const uint8_t* t, int nt) {
        std::vector<Two> r((size_t)nq);
        for (int i = 0; i < nq; i++) {
            Two b;
            for (int j = 0; j < nt; j++) {
                int d = 0;
                for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(q[32 * i + k] ^ t[32 * j + k]));
                if (b.i1 < 0 || d < b.d1) { b.i2 = b.i1; b.d2 = b.d1; b.i1 = j; b.d1 = d; }   // strict: the earlier index keeps its place
                else if (b.i2 < 0 || d < b.d2) { b.i2 = j; b.d2 = d; }
            }
            r[(size_t)i] = b;
        }
        return r;
    };
    const std::vector<Two> a = knn2(desc0, n0, desc1, n1), b = knn2(desc1, n1, desc0, n0);
    auto surviving = [knn_ratio](const Two& m) { return (m.i2 < 0 || (float)m.d1 / (float)m.d2 > knn_ratio) ? -1 : m.i1; };
    int n = 0;
    for (int i = 0; i < n0; i++) {
        const int j = surviving(a[(size_t)i]);
        if (j < 0 || surviving(b[(size_t)j]) != i) continue;
        if (n < cap) { out_qtd[3 * n] = i; out_qtd[3 * n + 1] = j; out_qtd[3 * n + 2] = a[(size_t)i].d1; }
        n++;
    }
    return n;
}

}  // extern "C"
