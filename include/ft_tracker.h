/*
 * ft_tracker.h — C interface of the host tracker layer (TrackKLT / TrackKLTBatch re-stated on the device C ABI),
 * for drivers that are not C++ (the Python test and benchmark harness binds this with ctypes).
 *
 * ftk_tracker mirrors one reference tracker object as the reference's driver uses it:
 *   new TrackKLT(num_pts, 0, fast_threshold, grid_x, grid_y, min_px_dist)   src/test_tracking.cpp:229
 *   set_calibration / feed_imu / feed_stereo                                src/test_tracking.cpp:231,:148,:207
 * ftk_batch runs n independent monocular streams through shared batched launches (no reference counterpart).
 */
#ifndef FT_TRACKER_H
#define FT_TRACKER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ftk_options {
    int num_pts;            /* src/FeatureTrackerOptions.h:36 -> 150 */
    int fast_threshold;     /* :39 -> 20 */
    int grid_x, grid_y;     /* :42,:45 -> 5,5 */
    int min_px_dist;        /* :48 -> 10 */
    int pyr_levels;         /* src/track/TrackKLT.h:121 -> 3 */
    int win_size;           /* src/track/TrackKLT.h:122 -> 15 */
    int lk_max_iter;        /* src/track/TrackKLT.cpp:305 -> 15 */
    double lk_eps;          /* -> 0.01 */
    double lk_min_eig;      /* -> 1e-4 */
    int use_gyro_seed;
    int forward_backward;
    float fb_thresh;
    int speculative_detect;
    int reject_outliers;    /* RANSAC on the fundamental matrix after LK, src/track/TrackKLT.cpp:405-424 -> on (see FT_LK_REJECT_OUTLIERS) */
    int ransac_hypotheses;  /* -> 1000 */
    int stereo_extrinsic_seed;  /* left -> right association seeded from the stereo extrinsics + known-E test,
                                   src/track/TrackMsckfVio.cpp:218-301 -> off ("same pixel", src/track/TrackKLT.cpp:290-295) */
    double stereo_threshold;    /* px, src/track/TrackMsckfVio.h:217 -> 3 */
    double drop_fraction;       /* load generator, no reference counterpart: forget this fraction of the active tracks at the
                                   start of every feed so that the detector must replenish on every frame -> 0 */
    int use_klt;                /* src/FeatureTrackerOptions.h:33 -> 1; 0 selects the descriptor tracker (TrackDescriptor, stereo only),
                                   src/test_tracking.cpp:229-230 */
    double knn_ratio;           /* descriptor tracker: ratio test of the two nearest neighbours, src/FeatureTrackerOptions.h:51 -> 0.85 */
    int descriptor_ransac_on_pairs;   /* descriptor tracker: 0 = the RANSAC step as the reference is written (it compares a point
                                   set with itself, src/track/TrackDescriptor.cpp:284-285, and rejects nothing); 1 = outlier
                                   rejection on the matched pairs -> 0 */
} ftk_options;

typedef struct ftk_tracker ftk_tracker;
typedef struct ftk_batch ftk_batch;

void ftk_default_options(ftk_options* o);

/* ---- one stereo-capable tracker (cameras 0 and 1) ---- */
int ftk_tracker_create(ftk_tracker** out, int device, const ftk_options* o, int max_width, int max_height);
void ftk_tracker_destroy(ftk_tracker* t);
/* calib: n_cams x 8 doubles (fx,fy,cx,cy,d1..d4); T_cam_imu: n_cams x 16 doubles (row-major 4x4) */
int ftk_tracker_set_calibration(ftk_tracker* t, int n_cams, const double* calib8, const double* T_cam_imu);
int ftk_tracker_feed_imu(ftk_tracker* t, double timestamp, const double* am3, const double* wm3);
int ftk_tracker_feed_stereo(ftk_tracker* t, double timestamp, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                            int cam_left, int cam_right);
int ftk_tracker_feed_monocular(ftk_tracker* t, double timestamp, const uint8_t* img, int w, int h, size_t stride, int cam);
/* current tracks of a camera: returns the count; fills up to cap entries */
int ftk_tracker_get_points(ftk_tracker* t, int cam, float* xy, uint64_t* ids, int cap);
int ftk_tracker_database_size(ftk_tracker* t);
/* observations of one feature in one camera: returns the count; fills up to cap (u,v,u_n,v_n) rows and timestamps */
int ftk_tracker_get_track(ftk_tracker* t, uint64_t id, int cam, float* uvn4, double* timestamps, int cap);

/* ---- n independent monocular streams, batched ----
 * The streams are split into `n_groups` groups (0 = choose: 3 from 32 streams on, 2 from 8 on, else 1).  A group has its own device
 * context and queue, a host driver thread and a feature-database writer thread, so groups overlap each other's device
 * work and host bookkeeping.  host_threads (0 = choose: two per group) is the total number of host threads; threads
 * beyond two per group help with the per-stream bookkeeping inside the groups. */
int ftk_batch_create(ftk_batch** out, int device, const ftk_options* o, int n_streams, int max_width, int max_height);
int ftk_batch_create_grouped(ftk_batch** out, int device, const ftk_options* o, int n_streams, int max_width, int max_height,
                             int n_groups, int host_threads);
void ftk_batch_destroy(ftk_batch* b);
int ftk_batch_set_calibration(ftk_batch* b, int stream, const double* calib8, const double* T_cam_imu);
int ftk_batch_feed_imu(ftk_batch* b, int stream, double timestamp, const double* am3, const double* wm3);
/* one frame per stream.  frames[i] is a host pointer (frames_on_device = 0) or a device pointer (= 1).
 * tracked/attempted (optional) receive the number of features with final status 1 / handed to LK. */
int ftk_batch_feed(ftk_batch* b, const double* timestamps, const void* const* frames, int frames_on_device, int w, int h,
                   size_t stride, long long* tracked, long long* attempted);
/* Pipelined feed for sustained throughput: posts this frame set to the groups and returns once the set posted two
 * calls earlier is complete, so every group always finds its next set waiting (it queues that set's device work while
 * it still tracks the current one).  tracked/attempted count the sets completed by this call; ftk_batch_flush
 * completes the rest.  Frames must stay valid until their set is complete (at most three calls, or a flush, later). */
int ftk_batch_feed_pipelined(ftk_batch* b, const double* timestamps, const void* const* frames, int frames_on_device, int w, int h,
                             size_t stride, long long* tracked, long long* attempted);
int ftk_batch_flush(ftk_batch* b, long long* tracked, long long* attempted);
int ftk_batch_get_points(ftk_batch* b, int stream, float* xy, uint64_t* ids, int cap);
/* number of features in a stream's database / observations of one feature (as ftk_tracker_get_track) */
int ftk_batch_database_size(ftk_batch* b, int stream);
int ftk_batch_get_track(ftk_batch* b, int stream, uint64_t id, float* uvn4, double* timestamps, int cap);
int ftk_batch_groups(ftk_batch* b);
/* the device context (an ft_ctx*) behind group g */
void* ftk_batch_group_ctx(ftk_batch* b, int group);
/* wall-clock seconds per stage accumulated over feeds (8 doubles: frame ingest, pyramid enqueue, detection, klt
 * prepare, klt enqueue, device wait, masks + feature DB update, total), cf. the stage timers at
 * src/track/TrackKLT.cpp:212-217 */
int ftk_batch_stage_times(ftk_batch* b, double* out8, long long* feeds, int reset);
/* the device context behind the batch (an ft_ctx*), for profiling switches and timers */
void* ftk_batch_device_ctx(ftk_batch* b);
const char* ftk_backend_name(void);

#ifdef __cplusplus
}
#endif
#endif
