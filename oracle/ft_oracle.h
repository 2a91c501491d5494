/*
 * ft_oracle.h — C interface of the CPU parity oracle (TEST INFRASTRUCTURE, not product code).
 *
 * PARITY UNPINNED: the arithmetic restated here lives in OpenCV 3 (cv::buildOpticalFlowPyramid,
 * cv::calcOpticalFlowPyrLK, cv::FAST), which the reference links but does not vendor, and the reference ships
 * no test, fixture or golden vector for it.  This file restates the published algorithm of the generic
 * (non-IPP, non-OpenCL) x86 code path; it is anchored on the reference's call sites only:
 *   src/track/TrackKLT.cpp:51-56   (pyramid), :306 and :402 (pyramidal LK)
 *   src/track/Grider_FAST.h:76-132 (per-cell FAST + top-k)
 *   src/propagator/TrackPropagator.h:107-153 and src/track/TrackMsckfVio.cpp:344-355 (gyro seed)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef FT_ORACLE_H
#define FT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag values follow OpenCV's OPTFLOW_* enum so the call sites read the same */
#define FTO_LK_USE_INITIAL_FLOW 4
#define FTO_LK_GET_MIN_EIGENVALS 8
#define FTO_LK_NO_ERROR 1024        /* leave err at 0 (the bounds test of the final position still applies to status) */
#define FTO_LK_FORWARD_BACKWARD 256 /* build-defined: see fto_lk_track */
#define FTO_LK_FLOAT_SUMS 2048       /* test switch: accumulate the normal-equation sums in float, in the order of OpenCV's SSE2
                                       code path, instead of exactly (bounds the one deliberate deviation of fto_lk_track) */
#define FTO_LK_REJECT_OUTLIERS 512  /* build-defined: see fto_reject_outliers (applied by the tracker backend) */

typedef struct fto_pyramid fto_pyramid;

typedef struct fto_lk_params {
    int win_w, win_h;   /* TrackKLT.h:122 -> 15x15 */
    int max_level;      /* TrackKLT.h:121 -> 3 */
    int max_iter;       /* TermCriteria COUNT, TrackKLT.cpp:305,:401 -> 15 */
    double eps;         /* TermCriteria EPS -> 0.01 (squared internally) */
    double min_eig;     /* calcOpticalFlowPyrLK default 1e-4 */
    int flags;          /* FTO_LK_* */
    float fb_thresh;    /* forward-backward distance bound in px (only with FTO_LK_FORWARD_BACKWARD) */
} fto_lk_params;

typedef struct fto_keypoint {
    float x, y;         /* pixel position (integers stored as float, like cv::KeyPoint::pt) */
    float response;     /* FAST score */
} fto_keypoint;

/* cv::pyrDown for 8-bit single channel, BORDER_REFLECT_101. dst is ((w+1)/2) x ((h+1)/2). */
void fto_pyr_down(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);

/* cv::buildOpticalFlowPyramid(img, pyr, win, max_level, false, REFLECT_101, CONSTANT, true); TrackKLT.cpp:51-56.
 * Returns a pyramid whose level count is (achieved max level + 1). */
fto_pyramid* fto_pyramid_build(const uint8_t* img, int w, int h, size_t stride, int win_w, int win_h, int max_level);
void fto_pyramid_free(fto_pyramid* p);
int fto_pyramid_levels(const fto_pyramid* p);
/* geometry + pointer to pixel (0,0) of a level; pixels at [-win, size+win) are valid (REFLECT_101 padding) */
int fto_pyramid_level(const fto_pyramid* p, int level, int* w, int* h, size_t* stride, const uint8_t** data);

/* calcSharrDeriv: dst holds w*h interleaved (Ix,Iy) int16 pairs, row stride = 2*w shorts. */
void fto_scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst);

/* cv::calcOpticalFlowPyrLK on two pyramids built without derivatives.
 * next_xy is in/out (initial guess when FTO_LK_USE_INITIAL_FLOW is set).
 * With FTO_LK_FORWARD_BACKWARD a second pass tracks next->prev seeded with the original point and clears status
 * where that pass failed or lands farther than fb_thresh from the original point.
 * iters_out (optional, n*(levels) ints, level-major per point: iters_out[i*nlev + level]) receives the number of
 * Newton iterations executed per level (-1 where the level was skipped).
 * n_threads <= 1 runs serially; otherwise features are split across std::threads. */
int fto_lk_track(const fto_pyramid* prev, const fto_pyramid* next, const float* prev_xy, float* next_xy,
                 uint8_t* status, float* err, int n, const fto_lk_params* prm, int n_threads, int* iters_out);

/* cv::FAST(img, kps, threshold, nonmax) (FAST-9/16). Writes up to cap keypoints in row-major order, returns the
 * total number found (may exceed cap). */
int fto_fast(const uint8_t* img, int w, int h, size_t stride, int threshold, int nonmax, fto_keypoint* out, int cap);

/* FAST score map used by the detector tests: score (0 where not a corner) for every pixel of the view. */
void fto_fast_score_map(const uint8_t* img, int w, int h, size_t stride, int threshold, uint8_t* score, size_t sc_stride);

/* Grider_FAST::perform_griding (Grider_FAST.h:76-132).
 * tie_mode 0: per-cell order = (response desc, y asc, x asc)  — the total order this build defines.
 * tie_mode 1: per-cell order = std::sort with the reference's comparator (libstdc++ introsort; unstable on ties).
 * Returns number of keypoints written (<= cap). */
int fto_grid_fast(const uint8_t* img, int w, int h, size_t stride, int num_features, int grid_x, int grid_y,
                  int threshold, int nonmax, int tie_mode, int n_threads, fto_keypoint* out, int cap);

/* Same with the per-cell keep count given directly; cell_counts (optional, one int per cell) receives the number of
 * corners FAST found in each cell before truncation. */
int fto_grid_fast_cells(const uint8_t* img, int w, int h, size_t stride, int per_cell, int grid_x, int grid_y, int threshold,
                        int nonmax, int tie_mode, int n_threads, fto_keypoint* out, int cap, int* cell_counts);

/* Mean-gyro rotation between two frames (TrackPropagator.h:107-153 with the window end fixed to use the
 * sample under the end iterator): R_p_c = Rodrigues(R_cam_imu_T * mean_gyro * dt)^T, row-major float[9].
 * imu_t: n timestamps; gyro: n*3; R_imu_to_cam: row-major 3x3 (the rotation block of T_cam_imu). */
int fto_integrate_gyro(const double* imu_t, const double* gyro, int n, double t_last, double t_cur,
                       const double* R_imu_to_cam, float* R_p_c);

/* p' = pi(K * R_p_c * K^-1 * [p;1]) in float32 (TrackMsckfVio.cpp:348-355). K = fx,fy,cx,cy. */
void fto_predict_flow(const float* K4, const float* R_p_c, const float* xy, float* xy_out, int n);

/* H = K * R * K^-1 as used by fto_predict_flow, row-major float[9]. */
void fto_homography(const float* K4, const float* R_p_c, float* H);

/* Outlier rejection after LK (src/track/TrackKLT.cpp:405-424: undistort_point on both point sets, then
 * cv::findFundamentalMat(FM_RANSAC, 1/max_focal, 0.999)).  OpenCV's RANSAC draws its samples from its own random
 * number generator, so its inlier set cannot be reproduced without OpenCV; this is the build's own, fully specified
 * variant with the same model, error measure and threshold:
 *   - both point sets are undistorted like TrackBase::undistort_point (src/track/TrackBase.h:289-332) and rounded to
 *     float, calib8 = fx, fy, cx, cy, k1, k2, p1, p2;
 *   - at most `hypotheses` models (0 = 1000, the iteration limit of cv::findFundamentalMat) are fitted in index order, model h from the 8 distinct pairs drawn by a
 *     counter-based hash of (seed, h, draw number).  Whenever a model beats the best so far, the number of models still
 *     to be tried drops to the smallest N with (1 - w^8)^N <= 0.001, w = that model's inlier ratio: the adaptive
 *     iteration count behind the confidence argument 0.999 (OpenCV computes it with logarithms; here it comes from
 *     repeated multiplication in a fixed order, so that every implementation gets the same N).  The null vector of the 8x9 epipolar system comes from Gauss-Jordan elimination with complete
 *     pivoting in double precision (ties: lowest row, then lowest column; a sample whose pivot falls below 1e-12 is skipped);
 *   - a pair is an inlier of a model if (float)max(d1^2*s1, d2^2*s2) <= thr^2, the symmetric epipolar distance of
 *     OpenCV's estimator, thr = 1 / max(fx, fy of both cameras);
 *   - the model with the most inliers wins, the lower index on ties; status[i] &= inlier[i]; no model with at least
 *     8 inliers, or fewer than 8 pairs: every status becomes 0.
 * All n pairs take part, tracked or not, as in the reference.  Returns the winning model index or -1. */
int fto_reject_outliers(const float* prev_xy, const float* next_xy, uint8_t* status, int n, const double* calib8_prev,
                        const double* calib8_next, int hypotheses, uint32_t seed, double* F_out);

/* Stereo matching geometry of TrackMsckfVio::stereoMatch (src/track/TrackMsckfVio.cpp:218-301), the parts around
 * its LK call.  Rt12 = R_cam0_cam1 (row-major 3x3) followed by t_cam0_cam1.
 * fto_stereo_seed: guess of every cam0 point in cam1 under the rotation alone, i.e.
 *   cv::undistortPoints(K0, D0, R_cam0_cam1) -> Point2f -> cv::projectPoints(K1, D1) (:238-249).
 * fto_stereo_check: status[i] = 0 for tracked points outside [0, w-1] x [0, h-1] (:259-268) and, if threshold > 0, for
 *   points whose distance to the epipolar line of E = [t]x R exceeds threshold * 4 / (fx0 + fy0 + fx1 + fy1) in
 *   normalized coordinates (:270-298). */
void fto_stereo_seed(const float* cam0_xy, int n, const double* calib8_cam0, const double* calib8_cam1, const double* Rt12,
                     float* cam1_xy);
void fto_stereo_check(const float* cam0_xy, const float* cam1_xy, uint8_t* status, int n, const double* calib8_cam0,
                      const double* calib8_cam1, const double* Rt12, int w, int h, double threshold);

/* 256-bit binary descriptors at keypoints, the step orb->compute() of TrackDescriptor::perform_detection_stereo
 * (src/track/TrackDescriptor.cpp:218-219).  ORB's sampling table is inside OpenCV and not available, so this is the
 * build's own BRIEF-style descriptor with ORB's geometry (parity unpinned): 5x5 binomial smoothing ((sum + 128) >> 8), test i
 * compares the smoothed values at two offsets in [-15, 15]^2, coordinate k of test i = mix32(0xB5297A4D + 4 i + k) mod 31
 * - 15 (mix32 as in fto_reject_outliers), bit i = S(p1) < S(p2) in byte i / 8 at bit i % 8, no rotation; a keypoint
 * closer than 17 px to the border gets valid = 0 and a zero descriptor.  Positions are truncated to whole pixels. */
int fto_compute_descriptors(const uint8_t* img, int w, int h, size_t stride, const float* xy, int n, uint8_t* desc, uint8_t* valid);

/* Descriptor matching of TrackDescriptor::robust_match before its RANSAC (src/track/TrackDescriptor.cpp:249-265,
 * :309-354) on 32-byte binary descriptors: 2 nearest neighbours by Hamming distance both ways (equal distances: the
 * lower index first), ratio test first / second > knn_ratio -> dropped, symmetry test.  out_qtd receives
 * (query, train, distance) int triples in query order; returns the number of matches (all counted, at most cap written). */
int fto_match_descriptors(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, int* out_qtd, int cap);

#ifdef __cplusplus
}
#endif
#endif
