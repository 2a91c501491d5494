/*
 * ft_device.h — C ABI of the MI355X (gfx950) visual front-end device layer.
 *
 * This is the boundary the reference would bind in place of its three OpenCV entry points and cv::FAST:
 *   cv::buildOpticalFlowPyramid   src/track/TrackKLT.cpp:51-56      -> ft_build_pyramid
 *   calcSharrDeriv (inside LK)    src/track/TrackKLT.cpp:306,:402   -> inside ft_lk_track; as images: ft_build_pyramid(with_gradients) / ft_build_gradients
 *   cv::calcOpticalFlowPyrLK      src/track/TrackKLT.cpp:306,:402   -> ft_lk_track
 *   cv::FAST + sort + top-k       src/track/Grider_FAST.h:94-130    -> ft_fast_grid
 *   K*R*K^-1 point warp           src/track/TrackMsckfVio.cpp:348-355 (commented sketch) -> ft_lk_job::H / ft_predict_flow
 *
 * Conventions
 *  - A context owns ALL device memory (a ring of `n_slots` pyramids + gradient pyramids for each of `n_streams`
 *    independent camera streams), the pinned staging buffers and one HIP queue.  Host pointers are borrowed for the
 *    duration of a call only, except the outputs of the *_async calls, which are written by the next ft_sync.
 *  - Every compute entry point is batched: it takes arrays of (stream, slot) pairs and covers all of them in one
 *    launch per kernel.  All images of one batched call must have the same size.
 *  - Calls on one context must be serialised by the caller (same contract as the reference's per-camera mutex,
 *    src/track/TrackBase.h:59-66).  Different contexts may be driven from different host threads.
 *  - All functions return 0 (FT_OK) or a negative FT_E_* code; nothing throws or exits.
 */
#ifndef FT_DEVICE_H
#define FT_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT_OK 0
#define FT_E_INVALID (-1)   /* bad argument */
#define FT_E_LIMIT (-2)     /* exceeds the limits the context was created with */
#define FT_E_HIP (-3)       /* a HIP runtime call failed; see ft_last_hip_error */
#define FT_E_NOMEM (-4)     /* allocation failed */
#define FT_E_STATE (-5)     /* slot has no frame / pyramid / gradients yet */
#define FT_E_NODEVICE (-6)  /* no usable gfx950 device */

#define FT_MAX_LEVELS 8     /* images per pyramid (max_level + 1) */
#define FT_MAX_WIN 32       /* largest LK window side */
#define FT_MAX_BATCH 256    /* images or LK jobs per batched call */

/* LK flags; the first two keep OpenCV's OPTFLOW_* values so call sites read the same */
#define FT_LK_USE_INITIAL_FLOW 4
#define FT_LK_GET_MIN_EIGENVALS 8
#define FT_LK_FORWARD_BACKWARD 256
/* After tracking, reject outliers per job with a RANSAC on the fundamental matrix of the undistorted point pairs:
 * the tail of TrackKLT::perform_matching, src/track/TrackKLT.cpp:405-424 (undistort_point + cv::findFundamentalMat
 * (FM_RANSAC, 1/max_focal, 0.999) + mask_klt & mask_ransac).  Model, error measure and threshold are OpenCV's; the
 * sampling is this library's own seeded scheme, because OpenCV's random sequence cannot be reproduced without it.
 * Hypotheses are tried in index order and, as in OpenCV, a better model lowers the number still to be tried (to the
 * smallest N with (1 - w^8)^N <= 0.001 for its inlier ratio w: confidence 0.999); ransac_hypotheses is the upper bound
 * (OpenCV's is 1000 iterations).
 * Where the search differs from OpenCV 3's (as its fundam.cpp is remembered; OpenCV is not available to check against):
 * OpenCV draws 7-point minimal samples (up to three models each from the cubic in the null space, every one rank 2),
 * lowers its iteration count with w^7, and for fewer than 15 points runs LMedS instead of RANSAC; here a sample is 8
 * points, the model is the null vector of the 8x9 system (no rank-2 step), the count falls with w^8, and RANSAC runs
 * from 8 points up.  The error measure, the threshold, the confidence and the iteration limit are OpenCV's.  What is
 * tested is therefore not the inlier set but its quality: precision and recall against known outliers
 * (tests/synth.py REJECTION_BARS).  Jobs need calib_prev / calib_next. */
#define FT_LK_REJECT_OUTLIERS 512
/* Do not compute the residual (err stays 0).  TrackKLT::perform_matching never reads the `error` vector it gets from
 * cv::calcOpticalFlowPyrLK (src/track/TrackKLT.cpp:399-402), so the tracker layer sets this; status is unaffected
 * (the bounds test of the final position, which can still clear it, is kept). */
#define FT_LK_NO_ERROR 1024

typedef struct ft_ctx ft_ctx;

typedef struct ft_limits {
    int max_width, max_height;   /* largest frame */
    int max_levels;              /* images per pyramid, 1..FT_MAX_LEVELS */
    int n_streams;               /* independent camera streams held by this context */
    int n_slots;                 /* pyramid ring depth per stream (>= 2 for frame-to-frame tracking) */
    int max_points;              /* total LK points per batched call */
    int max_grid_cells;          /* grid_x*grid_y upper bound for ft_fast_grid */
    int max_per_cell;            /* per-cell keep count upper bound for ft_fast_grid */
} ft_limits;

typedef struct ft_lk_params {
    int win_w, win_h;    /* reference: 15x15, src/track/TrackKLT.h:122 */
    int max_level;       /* reference: 3,    src/track/TrackKLT.h:121 */
    int max_iter;        /* reference: 15,   src/track/TrackKLT.cpp:305,:401 */
    double eps;          /* reference: 0.01 */
    double min_eig;      /* OpenCV default 1e-4 */
    int flags;           /* FT_LK_* */
    float fb_thresh;     /* px; only with FT_LK_FORWARD_BACKWARD */
    int ransac_hypotheses;    /* only with FT_LK_REJECT_OUTLIERS: models tried per job at most, 1..1024 (0 = 1000) */
    unsigned ransac_seed;     /* only with FT_LK_REJECT_OUTLIERS: seed of the sample sequence */
    double stereo_threshold;  /* only for jobs with stereo_Rt: largest distance to the epipolar line, in pixels
                                 (src/track/TrackMsckfVio.h:217 -> 3); <= 0 switches the check off (seed only) */
} ft_lk_params;

/* One LK job = one calcOpticalFlowPyrLK call of the reference (one image pair, n points). */
typedef struct ft_lk_job {
    int stream_prev, slot_prev;   /* template pyramid (its Scharr gradients are evaluated inside the kernel) */
    int stream_next, slot_next;   /* target pyramid */
    int n;                        /* number of points */
    const float* prev_xy;         /* n*2, level-0 pixels */
    float* next_xy;               /* n*2, in: initial guess (FT_LK_USE_INITIAL_FLOW and H == NULL), out: result */
    uint8_t* status;              /* n */
    float* err;                   /* n */
    const float* H;               /* optional row-major 3x3: the initial guess becomes pi(H*[prev;1]) (gyro seed) */
    int* iters;                   /* optional n*(max_level+1): Newton iterations per level, -1 = level skipped */
    const double* calib_prev;     /* with FT_LK_REJECT_OUTLIERS: fx fy cx cy k1 k2 p1 p2 of the template image's camera */
    const double* calib_next;     /* ... and of the target image's camera (the same pointer for temporal tracking) */
    const double* stereo_Rt;      /* optional, 12 doubles: R_cam0_cam1 (row-major 3x3) then t_cam0_cam1, prev = cam0, next = cam1.
                                     Makes the job a stereo match as TrackMsckfVio::stereoMatch does it
                                     (src/track/TrackMsckfVio.cpp:218-301): the initial guess of every point is its
                                     projection into cam1 under pure rotation (undistort with cam0, rotate, distort with
                                     cam1; :238-249; next_xy is not read), and after LK a point is dropped if it left the
                                     image (:259-268) or lies further than stereo_threshold pixels from its epipolar
                                     line under E = [t]x R (:270-298).  Needs calib_prev / calib_next. */
} ft_lk_job;

/* One set of point correspondences for outlier rejection alone. */
typedef struct ft_match_set {
    int n;
    const float* prev_xy;         /* n*2 pixels in the first image */
    const float* next_xy;         /* n*2 pixels in the second image */
    uint8_t* status;              /* n, in: 1 = still a candidate, out: candidate AND inlier of the winning model */
    const double* calib_prev;     /* fx fy cx cy k1 k2 p1 p2 of the first image's camera */
    const double* calib_next;     /* ... of the second image's camera */
} ft_match_set;

/* One accepted descriptor match (cv::DMatch: queryIdx, trainIdx, distance). */
typedef struct ft_match {
    int query, train;
    float distance;      /* Hamming distance, 0..256 */
} ft_match;

typedef struct ft_keypoint {
    float x, y;          /* full-image pixel position */
    float response;      /* FAST score */
} ft_keypoint;

/* Per-kernel device time accumulated while ft_profile(ctx, 1) is on (HIP events on the context's queue). */
typedef struct ft_kernel_time {
    double pyramid_ms;   int pyramid_launches;    /* fused pyrDown + Scharr level kernel */
    double lk_ms;        int lk_launches;
    double fast_ms;      int fast_launches;       /* score + nms/compact + top-k, per ft_fast_grid call */
    /* LK work done while ft_profile mode 2 is on: points tracked, (point, level) pairs whose template was evaluated,
     * and Newton iterations summed over them (from the kernel's own per-level counters) */
    long long lk_points, lk_level_visits, lk_iterations;
    double reject_ms;    int reject_launches;     /* outlier rejection (FT_LK_REJECT_OUTLIERS, ft_reject_outliers) */
} ft_kernel_time;

int ft_ctx_create(ft_ctx** out, int device, const ft_limits* limits);
void ft_ctx_destroy(ft_ctx* ctx);

/* Frame ingest.  ft_upload copies the host image into pinned staging (so the caller may reuse its buffer at
 * once, as after the clone() at src/track/TrackKLT.cpp:46-47) and enqueues the H2D copy into level 0.
 * ft_set_frame_device enqueues a device-to-device copy from a frame already resident in HBM (asynchronous: the
 * source must stay valid until the next ft_sync). */
int ft_upload(ft_ctx* ctx, int stream, int slot, const uint8_t* host, int w, int h, size_t stride);
int ft_set_frame_device(ft_ctx* ctx, int stream, int slot, const void* dev, int w, int h, size_t stride);
/* n device-resident frames of one size into n slots with a single copy launch */
int ft_set_frames_device(ft_ctx* ctx, int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride);

/* ft_set_frames_device followed by ft_build_pyramid(with_gradients = 0) in one pass over the frames: level 0 and
 * level 1 are written while each frame is read once (the clone() + buildOpticalFlowPyramid pair at
 * src/track/TrackKLT.cpp:46-56). Asynchronous; the frames must stay valid until the next ft_sync / ft_wait. */
int ft_set_frames_device_pyramid(ft_ctx* ctx, int n, const int* streams, const int* slots, const void* const* dev, int w, int h,
                                 size_t stride, int max_level, int win_w, int win_h, int* achieved);

/* Frames in host memory, for many streams at once and without a copy command: ft_stage_frame copies one frame into the
 * context's pinned staging buffer of (stream, slot) on the calling thread (it first waits until nothing on the device
 * reads that buffer any more; calls for different slots may run on different threads at the same time), and
 * ft_set_frames_staged_pyramid builds level 0 and the pyramid of the staged frames in one pass that reads the staging
 * over the bus.  Replaces, like ft_upload + ft_build_pyramid, img.clone() and cv::buildOpticalFlowPyramid
 * (src/track/TrackKLT.cpp:46-56). */
int ft_stage_frame(ft_ctx* ctx, int stream, int slot, const uint8_t* host, int w, int h, size_t stride);
int ft_set_frames_staged_pyramid(ft_ctx* ctx, int n, const int* streams, const int* slots, int w, int h, int max_level, int win_w,
                                 int win_h, int* achieved);

/* cv::buildOpticalFlowPyramid(img, pyr, win_size, pyr_levels) of src/track/TrackKLT.cpp:51-56 for n images at once:
 * levels 1..max_level by pyrDown and, if with_gradients, Scharr (Ix,Iy) of every level (calcSharrDeriv), one launch
 * per level.  achieved[i] (optional) receives the top level index after the early-stop rule of
 * buildOpticalFlowPyramid (stop when the next level would not exceed the window). Asynchronous. */
int ft_build_pyramid(ft_ctx* ctx, int n, const int* streams, const int* slots, int max_level, int win_w, int win_h,
                     int with_gradients, int* achieved);
/* Scharr gradients for pyramids that were built without them. Asynchronous. */
int ft_build_gradients(ft_ctx* ctx, int n, const int* streams, const int* slots);

/* cv::calcOpticalFlowPyrLK(pyr0, pyr1, pts0, pts1, status, err, win, maxLevel, term_crit, flags) of
 * src/track/TrackKLT.cpp:306 and :402, for n_jobs image pairs in one launch.  ft_lk_track returns with the outputs written;
 * ft_lk_track_async returns after enqueueing and the outputs are written by the next ft_sync. */
int ft_lk_track(ft_ctx* ctx, int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm);
int ft_lk_track_async(ft_ctx* ctx, int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm);

/* Outlier rejection on correspondences the caller already holds: undistort_point + cv::findFundamentalMat(FM_RANSAC,
 * 1/max_focal, 0.999) + "mask_klt && mask_ransac" of src/track/TrackKLT.cpp:405-424, for n_sets point sets in one
 * launch (all n points of a set enter the estimation, as in the reference; status only gates the output).  Same model
 * search as FT_LK_REJECT_OUTLIERS.  Fewer than 8 points, or no model with at least 8 inliers, clears the whole set.
 * hypotheses 1..1024 (0 = 1000).  Shares its staging with ft_lk_track: total points <= limits.max_points, and an
 * undelivered tracking batch is finished first. */
int ft_reject_outliers(ft_ctx* ctx, int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed);
int ft_reject_outliers_async(ft_ctx* ctx, int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed);

/* Grider_FAST::perform_griding (src/track/Grider_FAST.h:76-132) for n images: per-cell cv::FAST(roi, threshold, true) =
 * FAST-9/16 + 3x3 NMS, then the per-cell top `per_cell` by (response desc, y asc, x asc).
 * Image i writes at most `cap` keypoints at out + i*cap, cells in index order, and its count to n_out[i].
 * cell_counts (optional, n*cells) receives the number of NMS survivors per cell before truncation. */
int ft_fast_grid(ft_ctx* ctx, int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y,
                 int per_cell, ft_keypoint* out, int cap, int* n_out, int* cell_counts);
int ft_fast_grid_async(ft_ctx* ctx, int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y,
                       int per_cell, ft_keypoint* out, int cap, int* n_out, int* cell_counts);

/* 256-bit binary descriptors at n keypoints of level 0 of (stream, slot): the step orb->compute(img, pts, desc) of
 * TrackDescriptor::perform_detection_stereo (src/track/TrackDescriptor.cpp:218-219).  ORB's sampling table lives inside
 * OpenCV and is not available to this build, so the descriptor is the library's own BRIEF-style one with ORB's geometry
 * (parity unpinned): 5x5 binomial smoothing, 256 comparisons of smoothed intensities at pseudo-random offset pairs inside
 * the 31x31 patch (offset coordinate k of test i = mix32(0xB5297A4D + 4 i + k) mod 31 - 15 with the 32-bit mixer
 * x ^= x >> 16, x *= 0x7feb352d, x ^= x >> 15, x *= 0x846ca68b, x ^= x >> 16), bit i = S(p1) < S(p2) in byte i / 8 at
 * bit i % 8, no rotation.  xy: n*2 pixel positions (their integer parts are used); desc: n*32 bytes; valid[i] = 0 and a
 * zero descriptor for a keypoint closer than 17 pixels to the border (ORB drops those keypoints).  Synchronous;
 * n <= 65536. */
int ft_compute_descriptors(ft_ctx* ctx, int stream, int slot, const float* xy, int n, uint8_t* desc, uint8_t* valid);

/* Matching of two sets of 256-bit binary descriptors (ORB rows, 32 bytes each) as TrackDescriptor::robust_match does
 * it before its RANSAC (src/track/TrackDescriptor.cpp:249-265, :309-354): the two nearest neighbours of every
 * descriptor in the other set, both ways (cv::BFMatcher(NORM_HAMMING).knnMatch(…, 2); equal distances: the lower
 * index first), the ratio test distance[0] / distance[1] > knn_ratio -> dropped (also dropped: fewer than two
 * neighbours), and the symmetry test (query -> train kept only if train's own surviving match is query).  Accepted
 * matches are written in query order; *n_out is their number (at most cap are written).  The reference's RANSAC tail
 * (:267-304) is ft_reject_outliers on the matched points; descriptor extraction (:218-219) is ft_compute_descriptors.
 * Synchronous; n0, n1 <= 65536. */
int ft_match_descriptors(ft_ctx* ctx, const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio,
                         ft_match* out, int cap, int* n_out);

/* H = K*R*K^-1 (host, float32; K4 = fx,fy,cx,cy) and the device point warp p' = pi(H*[p;1]): the gyro-predicted
 * initial flow sketched at src/track/TrackMsckfVio.cpp:344-355, R_p_c from src/propagator/TrackPropagator.h:107-153. */
void ft_homography(const float* K4, const float* R_p_c, float* H);
int ft_predict_flow(ft_ctx* ctx, const float* H, const float* xy, float* xy_out, int n);

/* Parity/debug readback. Synchronous. */
int ft_level_info(ft_ctx* ctx, int stream, int slot, int level, int* w, int* h);
int ft_download_level(ft_ctx* ctx, int stream, int slot, int level, uint8_t* host, size_t stride);
int ft_download_grad(ft_ctx* ctx, int stream, int slot, int level, int16_t* host, size_t stride_shorts);

/* Wait for everything enqueued on the context and deliver pending *_async outputs. */
int ft_sync(ft_ctx* ctx);
/* Partial wait.  ft_fence marks the current end of the context's queue; ft_wait(token) blocks until the queue has
 * passed that mark and delivers the outputs of the *_async calls enqueued before it, while work enqueued after the
 * mark (e.g. the next frame's pyramid and detector) keeps running.  At most 16 fences may be outstanding; at most
 * two ft_fast_grid_async results and one ft_lk_track_async result may be undelivered at any time (a further call
 * first waits for the whole queue).  No reference counterpart: the reference's OpenCV calls are synchronous. */
int ft_fence(ft_ctx* ctx, uint64_t* token);
int ft_wait(ft_ctx* ctx, uint64_t token);
/* Second device queue for the work that does not depend on tracking results.  With on != 0, ft_upload,
 * ft_set_frame_device, ft_set_frames_device, ft_set_frames_device_pyramid, ft_set_frames_staged_pyramid, ft_build_pyramid and
 * ft_fast_grid_async put their work on a queue of their own, so that the next frame's pyramids and detector run
 * beside the current frame's tracking instead of behind it.  Everything else stays on the context's queue and waits
 * for that work before it starts; fences, ft_wait and ft_sync cover both queues.  The caller's part: a pyramid slot
 * these calls write must not be an operand of tracking work that is still queued (a ring of three slots per stream
 * does that for frame-to-frame tracking).  Worth it for one large stream alone on its device (1280x720 with 2 000
 * features: + 15 %); batches of small streams lose a few per cent, and the device takes four queues with work at a
 * time.  FT_FRONT_QUEUE=0 / 1 in the environment overrides the call.
 * No reference counterpart: the reference's OpenCV calls are synchronous. */
int ft_set_front_queue(ft_ctx* ctx, int on);
/* The pyramid calls that start from frames (ft_set_frames_device_pyramid, ft_set_frames_staged_pyramid) may hold back the
 * launches of the levels above 1: an ft_fast_grid_async call on the same images right behind them carries those steps
 * in its own launches (detector and level 1 -> 2 share one, ranking and level 2 -> 3 the next).  Any other call that puts
 * work on the queue or waits for it launches them first, so results never depend on it; ft_flush launches them now, for
 * a caller that will do host work before its next call (FT_FRONT_FUSE=0: nothing is ever held back). */
int ft_flush(ft_ctx* ctx);

/* Timing. ft_profile(ctx, mode): mode 1 brackets every kernel group with HIP events on the context's queue (adds a
 * little overhead; the kernels launched are exactly those of an unprofiled run); mode 2 counts the LK work instead
 * (points, level visits, Newton iterations: the tracking kernel then also writes its per-level iteration table, so its
 * time is not representative); 3 = both; 0 = off.  ft_kernel_times reads and optionally resets the totals.
 * ft_queue_time_* time a whole region. */
int ft_profile(ft_ctx* ctx, int enable);
int ft_kernel_times(ft_ctx* ctx, ft_kernel_time* out, int reset);
int ft_queue_time_begin(ft_ctx* ctx);
int ft_queue_time_end(ft_ctx* ctx, float* ms);

const char* ft_strerror(int code);
const char* ft_last_hip_error(ft_ctx* ctx);
int ft_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
