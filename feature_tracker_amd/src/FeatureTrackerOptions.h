// Options of the visual front-end.  The first block keeps the reference's field names and defaults
// (src/FeatureTrackerOptions.h:28-51); the second block promotes values the reference hard-codes
// (src/track/TrackKLT.h:121-122, src/track/TrackKLT.cpp:305,:401) because other configurations vary them.
#pragma once
#include <cstdio>
#include <map>
#include <utility>

#include "types.h"

namespace feature_tracker {

struct FeatureTrackerOptions {
    int num_cameras = 2;

    // ---- tracker ----
    bool use_klt = true;        // KLT tracking; false: the descriptor tracker (TrackDescriptor)
    int num_pts = 150;          // points extracted and tracked in each image
    int fast_threshold = 20;    // FAST extraction threshold
    int grid_x = 5;             // detection grid columns
    int grid_y = 5;             // detection grid rows
    int min_px_dist = 10;       // minimum spacing between tracked features
    double knn_ratio = 0.85;    // descriptor matcher only

    // ---- promoted constants (defaults = the reference's hard-coded values) ----
    int pyr_levels = 3;         // LK maxLevel
    int win_size = 15;          // LK window side
    int lk_max_iter = 15;       // TermCriteria COUNT
    double lk_eps = 0.01;       // TermCriteria EPS
    double lk_min_eig = 1e-4;   // calcOpticalFlowPyrLK minEigThreshold
    bool use_gyro_seed = false; // seed temporal LK with the gyro-predicted flow
    bool forward_backward = false;
    float fb_thresh = 1.0f;     // px
    bool speculative_detect = true;   // run the detector on every new frame so replenishment needs no extra sync
    // outlier rejection after LK: on, as the reference always does it (src/track/TrackKLT.cpp:405-424).  The switch
    // exists because the sampling is this build's own and cannot reproduce OpenCV's inlier sets point for point:
    // off leaves mask_out = the LK status (and the forward-backward check)
    bool reject_outliers = true;
    int ransac_hypotheses = 1000;
    // left -> right association of new features as TrackMsckfVio::stereoMatch does it (src/track/TrackMsckfVio.cpp:218-301):
    // guess from the stereo extrinsics instead of "same pixel" (the TODO at src/track/TrackKLT.cpp:290-295), then the
    // known-essential-matrix test.  Needs both cameras' extrinsics.
    bool stereo_extrinsic_seed = false;
    double stereo_threshold = 3.0;   // px, src/track/TrackMsckfVio.h:217
    // Forget this fraction of the active tracks at the start of every feed (chosen by a hash of feature id and frame
    // count, so a run is repeatable), before the previous frame is topped up: forces detector replenishment on every
    // frame whatever the scene does.  A load generator for benchmarks and tests; the reference has nothing like it.
    double drop_fraction = 0.0;

    void print_trackers() const {
        printf("FEATURE TRACKING PARAMETERS:\n");
        printf("\t- num_pts: %d\n", num_pts);
        printf("\t- fast_threshold: %d\n\t- grid: %d x %d\n\t- min_px_dist: %d\n", fast_threshold, grid_x, grid_y, min_px_dist);
        printf("\t- lk: win %d, levels %d, term (%d, %g), gyro seed %d, forward-backward %d\n", win_size, pyr_levels,
               lk_max_iter, lk_eps, (int)use_gyro_seed, (int)forward_backward);
    }

    // ---- camera model ----
    std::map<size_t, Calib8> camera_intrinsics;          // fx, fy, cx, cy, d1..d4
    std::map<size_t, std::array<double, 7>> camera_extrinsics;   // q_ItoC (JPL xyzw), p_IinC
    std::map<size_t, Mat4d> camera_extrinsics_mat;       // T_cam_imu as a 4x4 matrix
    std::map<size_t, std::pair<int, int>> camera_wh;     // (width, height)

    void print_state() const {
        printf("STATE PARAMETERS:\n");
        for (const auto& kv : camera_intrinsics) {
            const auto wh = camera_wh.count(kv.first) ? camera_wh.at(kv.first) : std::make_pair(0, 0);
            printf("cam_%zu_wh: %d x %d\ncam_%zu_intrinsic: %g %g %g %g | %g %g %g %g\n", kv.first, wh.first, wh.second, kv.first,
                   kv.second[0], kv.second[1], kv.second[2], kv.second[3], kv.second[4], kv.second[5], kv.second[6], kv.second[7]);
        }
    }
};

}  // namespace feature_tracker
