// Outlier rejection after LK for gfx950: undistortion of both point sets and a seeded RANSAC on the fundamental
// matrix, one 256-thread workgroup per LK job (= per image pair), one hypothesis per thread.
//
// Replaces the tail of TrackKLT::perform_matching (reference: src/track/TrackKLT.cpp:405-424): undistort_point on
// both point sets (src/track/TrackBase.h:289-332), cv::findFundamentalMat(FM_RANSAC, 1/max_focal, 0.999) and the
// final mask = mask_klt & mask_ransac.  OpenCV's sample sequence cannot be reproduced without OpenCV, so the sampling
// is the build's own (a counter-based hash); model, error measure and threshold are OpenCV's.  The arithmetic is
// double precision in a fixed order, so the result equals the CPU checker's bit for bit.
//
// The work is tiny next to LK (256 eliminations of an 8x9 system and 256 x n error evaluations per image pair); the
// kernel exists so that the tracking results never leave the device before the final mask is known.  It also moves
// the results of the batch into the pinned host block (positions, residuals, final status).
#pragma once
#include "ft_kernels_lk.hpp"

namespace ftd {

struct OutlierArgs {
    const LKJobDev* jobs;
    const float* prev_xy;      // device
    const float* next_xy;      // device, LK results
    const float* err;          // device
    const uint8_t* status;     // device, LK status
    float4* norm;              // device scratch: x1 y1 x2 y2 normalized, one per point
    float* out_xy;             // pinned host
    float* out_err;            // pinned host
    uint8_t* out_status;       // pinned host
    int hypotheses;            // <= 256
    uint32_t seed;
};

__device__ __forceinline__ uint32_t ro_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// TrackBase::undistort_point: remove K, five fixed-point iterations of the inverse radial-tangential model
__device__ __forceinline__ float2 ro_undistort(float u, float v, const double* c) {
    const double ifx = 1. / c[0], ify = 1. / c[1];
    const double x0 = ((double)u - c[2]) * ifx, y0 = ((double)v - c[3]) * ify;
    double x = x0, y = y0;
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
    return make_float2((float)x, (float)y);
}

__device__ __forceinline__ bool ro_inlier(const double* F, float4 m, float thr2) {
    const float x1 = m.x, y1 = m.y, x2 = m.z, y2 = m.w;
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    const float err = (float)(e1 < e2 ? e2 : e1);   // std::max(e1, e2)
    return err <= thr2;
}

// null vector of the 8x9 system by Gaussian elimination with complete pivoting (rows of A are in LDS, one system
// per thread, element (i, j) at A[(i * 9 + j) * 256]); false if a pivot is below 1e-12
__device__ bool ro_null_vector(double* A, double* f) {
    int col_of[9];
#pragma unroll
    for (int j = 0; j < 9; j++) col_of[j] = j;
#define RO_A(i, j) A[((i) * 9 + (j)) * 256]
    for (int k = 0; k < 8; k++) {
        int pi = k, pj = k;
        double best = -1.;
        for (int i = k; i < 8; i++)
            for (int j = k; j < 9; j++) {
                const double a = fabs(RO_A(i, j));
                if (a > best) { best = a; pi = i; pj = j; }   // first maximum in row-major order
            }
        if (!(best >= 1e-12)) return false;
        if (pi != k)
            for (int j = 0; j < 9; j++) { const double t = RO_A(k, j); RO_A(k, j) = RO_A(pi, j); RO_A(pi, j) = t; }
        if (pj != k) {
            for (int i = 0; i < 8; i++) { const double t = RO_A(i, k); RO_A(i, k) = RO_A(i, pj); RO_A(i, pj) = t; }
            int tc = 0, tk = 0;
#pragma unroll
            for (int j = 0; j < 9; j++) { if (j == k) tk = col_of[j]; if (j == pj) tc = col_of[j]; }
#pragma unroll
            for (int j = 0; j < 9; j++) { if (j == k) col_of[j] = tc; else if (j == pj) col_of[j] = tk; }
        }
        const double inv = 1. / RO_A(k, k);
        for (int i = k + 1; i < 8; i++) {
            const double m = RO_A(i, k) * inv;
            for (int j = k; j < 9; j++) RO_A(i, j) = RO_A(i, j) - m * RO_A(k, j);
        }
    }
    double y[9];
    y[8] = 1.;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        double acc = 0.;
#pragma unroll
        for (int j = k + 1; j < 9; j++) acc = acc + RO_A(k, j) * y[j];
        y[k] = -acc / RO_A(k, k);
    }
#pragma unroll
    for (int j = 0; j < 9; j++) {
#pragma unroll
        for (int q = 0; q < 9; q++)
            if (col_of[j] == q) f[q] = y[j];
    }
#undef RO_A
    return true;
}

// 8x9 doubles per thread do not fit LDS for 256 threads (147 KB), so the systems live in a global scratch laid out
// [element][job * 256 + thread]; elimination traffic stays in L2
__global__ void __launch_bounds__(256) reject_outliers_kernel(OutlierArgs a, double* __restrict__ sys_scratch) {
    __shared__ double best_F[9];
    __shared__ int red_count[256];
    __shared__ int winner;
    const int t = threadIdx.x;
    const LKJobDev* job = a.jobs + blockIdx.x;
    const int first = job->first, n = job->count;
    const double* c0 = job->calib;        // template image camera
    const double* c1 = job->calib + 8;    // target image camera

    // ---- normalized coordinates of both point sets ----
    for (int i = t; i < n; i += 256) {
        const float2 p = ro_undistort(a.prev_xy[2 * (first + i)], a.prev_xy[2 * (first + i) + 1], c0);
        const float2 q = ro_undistort(a.next_xy[2 * (first + i)], a.next_xy[2 * (first + i) + 1], c1);
        a.norm[first + i] = make_float4(p.x, p.y, q.x, q.y);
    }
    __syncthreads();   // also orders the global writes above before the reads below (same workgroup)

    const double f0 = c0[0] < c0[1] ? c0[1] : c0[0], f1 = c1[0] < c1[1] ? c1[1] : c1[0];
    const float thr = (float)(1. / (f0 < f1 ? f1 : f0)), thr2 = thr * thr;

    // ---- one hypothesis per thread ----
    int count = -1;
    double F[9];
    if (n >= 8 && t < a.hypotheses) {
        int idx[8];
        uint32_t draw = 0;
        for (int k = 0; k < 8; k++) {
            for (;;) {
                const uint32_t r = ro_mix32(a.seed ^ ro_mix32((uint32_t)t * 0x9E3779B9u + 0x85EBCA6Bu * draw));
                draw++;
                const int cand = (int)(r % (uint32_t)n);
                bool dup = false;
#pragma unroll
                for (int q = 0; q < 8; q++) dup |= (q < k) && idx[q] == cand;
                if (!dup) {
#pragma unroll
                    for (int q = 0; q < 8; q++) if (q == k) idx[q] = cand;
                    break;
                }
            }
        }
        double* A = sys_scratch + (size_t)blockIdx.x * 72 * 256 + t;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 m = a.norm[first + idx[k]];
            const double x1 = m.x, y1 = m.y, x2 = m.z, y2 = m.w;
            A[(k * 9 + 0) * 256] = x2 * x1; A[(k * 9 + 1) * 256] = x2 * y1; A[(k * 9 + 2) * 256] = x2;
            A[(k * 9 + 3) * 256] = y2 * x1; A[(k * 9 + 4) * 256] = y2 * y1; A[(k * 9 + 5) * 256] = y2;
            A[(k * 9 + 6) * 256] = x1;      A[(k * 9 + 7) * 256] = y1;      A[(k * 9 + 8) * 256] = 1.;
        }
        if (ro_null_vector(A, F)) {
            count = 0;
            for (int i = 0; i < n; i++) count += ro_inlier(F, a.norm[first + i], thr2) ? 1 : 0;
        }
    }

    // ---- most inliers wins, the lower hypothesis index on ties; at least 8 inliers ----
    red_count[t] = count;
    __syncthreads();
    if (t == 0) {
        int bh = -1, bc = 7;
        for (int h = 0; h < 256; h++)
            if (red_count[h] > bc) { bc = red_count[h]; bh = h; }
        winner = bh;
    }
    __syncthreads();
    if (t == winner) {
#pragma unroll
        for (int q = 0; q < 9; q++) best_F[q] = F[q];
    }
    __syncthreads();

    // ---- final status and the batch results, straight into the pinned host block ----
    const bool have_model = winner >= 0;
    for (int i = t; i < n; i += 256) {
        const int g = first + i;
        const bool keep = have_model && a.status[g] && ro_inlier(best_F, a.norm[g], thr2);
        a.out_xy[2 * g] = a.next_xy[2 * g];
        a.out_xy[2 * g + 1] = a.next_xy[2 * g + 1];
        a.out_err[g] = a.err[g];
        a.out_status[g] = keep ? 1 : 0;
    }
}

}  // namespace ftd
