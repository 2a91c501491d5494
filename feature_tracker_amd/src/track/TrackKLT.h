// KLT visual front-end (call surface of src/track/TrackKLT.h:40-127) driving the batched device kernels.
//
// Differences from the reference that follow from the device design, none of which change results:
//  * images live in a ring of two device pyramid slots per camera instead of host cv::Mat vectors;
//  * the two pyramids of a stereo pair, and the two temporal LK calls, are one batched launch each instead of
//    two host threads (src/track/TrackKLT.cpp:51-58, :91-98);
//  * the id merge uses a hash map instead of the O(N^2) search (src/track/TrackKLT.cpp:136-166);
//  * with speculative_detect the detector runs on every new frame (both frames of a stereo pair) right after its
//    pyramid, so that topping the previous frame up at the next call (src/track/TrackKLT.cpp:76-78) needs no extra
//    device round trip; the per-cell lists are ranked, so truncating them to the count actually needed gives what
//    the reference computes.
// The fundamental-matrix RANSAC after LK (src/track/TrackKLT.cpp:405-424) runs by default, as in the reference; it
// is a seeded RANSAC of this build's own definition (option reject_outliers; off: mask_out is the LK status).
#pragma once
#include <functional>
#include <vector>

#include "../FeatureTrackerOptions.h"
#include "Grider_FAST.h"
#include "TrackBase.h"

namespace feature_tracker {

// a frame to ingest: host pixels (copied through pinned staging) or a device-resident frame
struct FrameSource {
    const uint8_t* host = nullptr;
    const void* dev = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;
};

// ranked per-cell detector output for one image, as returned by one batched fast_grid call
struct Detection {
    bool valid = false;
    int per_cell = 0;                 // keep count the detector was launched with
    int cells = 0;
    const ft_keypoint* kps = nullptr; // cells concatenated in index order
    int n = 0;
    const int* cell_counts = nullptr; // NMS survivors per cell before truncation
};

// cache-line aligned: trackers of different streams are driven by different host threads
class alignas(64) TrackKLT : public TrackBase {
public:
    // reference constructor (src/track/TrackKLT.h:58-59) plus the backend and the first backend stream this tracker
    // uses: camera cam_id lives in backend stream stream_base + cam_id
    explicit TrackKLT(FrontendBackend* backend, int numfeats, int numaruco, int fast_threshold, int gridx, int gridy, int minpxdist,
                      int stream_base = 0)
        : TrackBase(numfeats, numaruco), be(backend), stream0(stream_base), threshold(fast_threshold), grid_x(gridx), grid_y(gridy),
          min_px_dist(minpxdist) {}

    TrackKLT(FrontendBackend* backend, const FeatureTrackerOptions& o, int numaruco = 0, int stream_base = 0)
        : TrackBase(o.num_pts, numaruco), be(backend), stream0(stream_base), threshold(o.fast_threshold), grid_x(o.grid_x),
          grid_y(o.grid_y), min_px_dist(o.min_px_dist), pyr_levels(o.pyr_levels), win_size(o.win_size), lk_max_iter(o.lk_max_iter),
          lk_eps(o.lk_eps), lk_min_eig(o.lk_min_eig), use_gyro_seed(o.use_gyro_seed), forward_backward(o.forward_backward),
          fb_thresh(o.fb_thresh), speculative_detect(o.speculative_detect), reject_outliers(o.reject_outliers),
          ransac_hypotheses(o.ransac_hypotheses), stereo_extrinsic_seed(o.stereo_extrinsic_seed), stereo_threshold(o.stereo_threshold),
          drop_fraction(o.drop_fraction) {}

    // Both feeds return FT_OK or the first error code of the device layer.  A failure while the frame is being
    // ingested (bad size, staging, pyramid launch) leaves the tracker as it was before the call.  A failure after
    // that (tracking launch, device wait) drops the tracks and the last pyramid of the cameras involved, so that the
    // next feed starts like a first frame; it is never reported as an ordinary "tracking lost" frame.
    int feed_stereo(double timestamp, const Image8& img_left, const Image8& img_right, size_t cam_id_left, size_t cam_id_right) override;

    // single-camera variant of feed_stereo (the fork's upstream has one; the benchmark streams are monocular)
    int feed_monocular(double timestamp, const Image8& img, size_t cam_id);

    ~TrackKLT() override {
        // the backend may still hold pointers into this object's detector buffers for a delivery that has not happened
        if (spec_pending && be) (void)be->sync();
    }

    // ---- staged monocular interface: lets a driver run many independent trackers through shared batched launches ----
    struct alignas(64) MonoPlan {
        size_t cam = 0;
        double timestamp = 0;
        int stream = 0, slot_cur = 0, slot_last = 0;
        int cols = 0, rows = 0;
        bool first = false;        // no previous points: detect on the current frame and return
        bool run_lk = false;
        bool reset = false;        // too few points: reference resets and re-detects next time
        std::vector<float> prev_xy, next_xy, err;
        std::vector<uint8_t> status;
        float H[9];
        bool use_H = false;
        double calib8[8];          // camera model handed to the outlier rejection
        // observations of this frame waiting to be written to the feature database (see mono_commit)
        std::vector<size_t> db_ids;
        std::vector<float> db_uvn;   // u, v, u_n, v_n per observation (u_n, v_n are filled in by mono_commit)
        bool db_pending = false;
    };
    // 1. choose the ring slot the new frame goes to (the caller ingests the frame into (plan.stream, plan.slot_cur))
    void mono_begin(double timestamp, int cols, int rows, size_t cam_id, MonoPlan& plan);
    // 2a. (non-speculative only) how many features the previous frame lacks; > 1 means it wants a detection on slot_last
    int mono_features_needed(const MonoPlan& plan);
    // 2b. top the previous frame up from `last_det` (may be invalid = nothing to add) and lay out the LK job
    void mono_prepare(MonoPlan& plan, const Detection& last_det, ft_lk_job& job_out);
    // 3. after the LK results are in: masks and state roll; cur_det = detection of the current frame (first-frame case /
    //    cache).  The frame's observations are left in the plan ...
    void mono_finish(MonoPlan& plan, const Detection& cur_det);
    // 4. ... and written to the feature database here (src/track/TrackKLT.cpp:187-198).  Nothing in the next frame's
    //    tracking depends on it, so a driver may run it while the device works on the next frame.
    // (part, parts): `parts` threads may commit one plan together, thread `part` writing the observations whose ids it owns
    // (FeatureDatabase::owner_of)
    void mono_commit(MonoPlan& plan, int part = 0, int parts = 1);
    // the ranked detection of the last frame, for a driver whose detector finishes after mono_finish (which then got
    // an invalid detection)
    void set_cached_detection(const Detection& det);

    // depth of the pyramid slot ring the monocular path cycles through (2 = previous + current; a driver that
    // ingests frame k+1 while frame k is still being tracked needs 3).  The backend must have that many slots.
    void set_slot_ring(int depth) { slot_ring = depth < 2 ? 2 : depth; }
    int slot_ring_depth() const { return slot_ring; }
    // slot the frame after the one described by `plan` will use
    int slot_after(const MonoPlan& plan) const { return (plan.slot_cur + 1) % slot_ring; }
    int next_slot(size_t cam_id) { return (cams[cam_id].slot_last + 1) % slot_ring; }

    ft_lk_params lk_params() const;
    int per_cell_max() const { return num_features / (grid_x * grid_y) + 1; }
    int grid_cols() const { return grid_x; }
    int grid_rows() const { return grid_y; }
    int fast_threshold() const { return threshold; }
    int max_level() const { return pyr_levels; }
    int window() const { return win_size; }
    bool speculative() const { return speculative_detect; }
    // detector output of the last frame cached by mono_finish (speculative mode)
    const std::vector<ft_keypoint>& cached_detection(int& per_cell, std::vector<int>& cell_counts) const {
        per_cell = cache_valid ? cache_per_cell : 0;
        cell_counts = cache_cell_counts;
        return cache_kps;
    }
    size_t tracked_last(size_t cam_id) const {
        auto it = pts_last.find(cam_id);
        return it == pts_last.end() ? 0 : it->second.size();
    }

protected:
    // src/track/TrackKLT.cpp:223-368
    int perform_detection_stereo(const DeviceImage& img0, const DeviceImage& img1, std::vector<KeyPoint>& pts0,
                                 std::vector<KeyPoint>& pts1, std::vector<size_t>& ids0, std::vector<size_t>& ids1,
                                 size_t cam_id_left = 0, size_t cam_id_right = 1);
    // monocular top-off: occupancy filter, then new points from a ranked detection
    void perform_detection_monocular(int cols, int rows, const Detection& det, std::vector<KeyPoint>& pts0, std::vector<size_t>& ids0);
    // src/track/TrackKLT.cpp:371-432 for one or two (img0 -> img1) pairs in one launch
    struct MatchJob {
        DeviceImage img0, img1;
        std::vector<KeyPoint>* kpts0;
        std::vector<KeyPoint>* kpts1;
        size_t id0, id1;
        std::vector<uint8_t>* mask_out;
        const float* H;
    };
    // after_launch (optional) runs between the tracking launch and the wait for its results
    int perform_matching(std::vector<MatchJob>& jobs, const std::function<void()>& after_launch = nullptr);

    // occupancy grid of (rows/min_px_dist+2) x (cols/min_px_dist+2) cells, first come wins (TrackKLT.cpp:230-263)
    struct Occupancy {
        int gw = 0, gh = 0, dist = 1;
        std::vector<uint8_t> cell;
        Occupancy(int cols, int rows, int min_px_dist) : gw(cols / min_px_dist + 2), gh(rows / min_px_dist + 2), dist(min_px_dist), cell((size_t)gw * gh, 0) {}
        // (int)(coord / min_px_dist): float division, truncation toward zero
        bool index(const Point2f& p, size_t& idx) const {
            const int cx = (int)(p.x / dist), cy = (int)(p.y / dist);
            if (cx < 0 || cy < 0 || cx >= gw || cy >= gh) return false;   // the reference would index out of bounds here
            idx = (size_t)cy * gw + cx;
            return true;
        }
        bool taken(const Point2f& p) const { size_t i; return index(p, i) ? cell[i] != 0 : true; }
        void take(const Point2f& p) { size_t i; if (index(p, i)) cell[i] = 1; }
        // taken() followed by take() with one index computation: true if the cell was free and is now marked
        bool try_take(const Point2f& p) {
            size_t i;
            if (!index(p, i) || cell[i]) return false;
            cell[i] = 1;
            return true;
        }
        void reset(int cols, int rows, int min_px_dist) {
            gw = cols / min_px_dist + 2; gh = rows / min_px_dist + 2; dist = min_px_dist;
            cell.assign((size_t)gw * gh, 0);
        }
        Occupancy() = default;
    };
    static void filter_by_occupancy(Occupancy& occ, std::vector<KeyPoint>& pts, std::vector<size_t>& ids);
    bool gyro_homography(size_t cam_id, double t_last, double t_cur, float* H);
    // fx fy cx cy k1 k2 p1 p2 of a camera; an uncalibrated camera counts as the identity model
    void calib8_of(size_t cam_id, double* out8) const;
    // R_cam0_cam1 (row-major) then t_cam0_cam1 from the two cameras' extrinsics (TrackMsckfVio.cpp:228-231); false
    // if one of them was never set
    bool stereo_Rt_of(size_t cam0, size_t cam1, double* out12) const;

    FrontendBackend* be;
    int stream0;

    // FAST grid detector parameters
    int threshold;
    int grid_x;
    int grid_y;
    int min_px_dist;

    // LK parameters (reference: pyr_levels = 3, win_size = 15x15)
    int pyr_levels = 3;
    int win_size = 15;
    int lk_max_iter = 15;
    double lk_eps = 0.01;
    double lk_min_eig = 1e-4;
    bool use_gyro_seed = false;
    bool forward_backward = false;
    float fb_thresh = 1.0f;
    bool speculative_detect = true;
    bool reject_outliers = true;
    int ransac_hypotheses = 1000;
    bool stereo_extrinsic_seed = false;
    double stereo_threshold = 3.0;
    double drop_fraction = 0.0;
    // forget drop_fraction of the tracks (same ids in every camera list handed in during one feed)
    void drop_tracks(std::vector<KeyPoint>& pts, std::vector<size_t>& ids) const;
    unsigned frames_fed = 0;   // seeds the outlier rejection's sample sequence: every frame draws other samples
    int slot_ring = 2;
    // per-tracker scratch reused every frame (no allocation in the steady state)
    Occupancy occ_scratch;

    // per camera: device slot holding the last pyramid (-1 = none), its size and timestamp
    struct CamState {
        int slot_last = -1;
        int cols = 0, rows = 0;
        double t_last = 0;
    };
    std::map<size_t, CamState> cams;

    // cached ranked detection of the last monocular frame
    std::vector<ft_keypoint> cache_kps;
    std::vector<int> cache_cell_counts;
    int cache_per_cell = 0;
    int cache_n = 0;
    bool cache_valid = false;

    // stereo path, speculative detection: the ranked detections of the newest left / right frames, queued right after
    // their pyramids and used when that pair is topped up at the next call (no detector round trip then)
    struct StereoDet {
        bool valid = false;
        int stream = -1, slot = -1, cols = 0, rows = 0, per_cell = 0, cells = 0, n = 0;
        std::vector<ft_keypoint> kps;
        std::vector<int> cell_counts;
    };
    StereoDet stereo_cache[2];            // promoted results: [0] left, [1] right
    std::vector<ft_keypoint> spec_kps;    // in flight: both images of one batched call
    std::vector<int> spec_n, spec_cells;
    bool spec_pending = false;
    int spec_per_cell = 0, spec_ncells = 0, spec_cap = 0, spec_cols = 0, spec_rows = 0, spec_stream[2] = {-1, -1}, spec_slot[2] = {-1, -1};
    int promote_stereo_detection();
    // Grider_FAST::perform_griding(img, ..., num_needed, ...) from the cached detection of that image, or by a call
    int grid_points(const DeviceImage& img, int num_needed, std::vector<KeyPoint>& out);
    // a device call failed after the frame's ingest: forget the tracks and last pyramids of these cameras and every
    // cached or pending detection; returns rc
    int abort_feed(int rc, size_t cam_a, size_t cam_b);

    // scratch for feed_monocular's own one-stream batch
    std::vector<ft_keypoint> det_buf;
    std::vector<int> det_cells;
};

}  // namespace feature_tracker
