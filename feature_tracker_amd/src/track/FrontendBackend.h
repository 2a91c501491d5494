// Abstract mirror of the device C ABI (include/ft_device.h).  The tracker classes talk to this interface; the
// product build binds it to the gfx950 library (HipBackend.h).  Tests bind a CPU checker behind the same
// interface to time and verify the identical orchestration code.
#pragma once
#include <cstddef>
#include <cstdint>

#include "ft_device.h"

namespace feature_tracker {

class FrontendBackend {
public:
    virtual ~FrontendBackend() = default;
    virtual int upload(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) = 0;
    virtual int set_frames_device(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride) = 0;
    // set_frames_device + build_pyramid(with_gradients = 0) as one pass over the frames
    virtual int set_frames_device_pyramid(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride,
                                          int max_level, int win_w, int win_h, int* achieved) = 0;
    // frames in host memory without a copy command: stage_frame copies one frame into the backend's staging (different
    // slots may be staged from different threads at once), set_frames_staged_pyramid turns the staged frames into pyramids
    virtual int stage_frame(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) = 0;
    virtual int set_frames_staged_pyramid(int n, const int* streams, const int* slots, int w, int h, int max_level, int win_w, int win_h,
                                          int* achieved) = 0;
    virtual int build_pyramid(int n, const int* streams, const int* slots, int max_level, int win_w, int win_h,
                              int with_gradients, int* achieved) = 0;
    virtual int lk_track_async(int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm) = 0;
    virtual int fast_grid_async(int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y,
                                int per_cell, ft_keypoint* out, int cap, int* n_out, int* cell_counts) = 0;
    // descriptor tracker: descriptors at keypoints of level 0, 2-NN Hamming matching with ratio and symmetry tests,
    // outlier rejection on point pairs the caller holds (all three complete before they return)
    virtual int compute_descriptors(int stream, int slot, const float* xy, int n, uint8_t* desc, uint8_t* valid) = 0;
    virtual int match_descriptors(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, ft_match* out, int cap, int* n_out) = 0;
    virtual int reject_outliers(int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed) = 0;
    virtual void homography(const float* K4, const float* R_p_c, float* H) = 0;
    virtual int sync() = 0;
    // partial wait: fence() marks the end of the queue, wait(token) returns once the queue has passed the mark and the
    // outputs of the async calls made before it are delivered; later work keeps running
    virtual int fence(uint64_t* token) = 0;
    virtual int wait(uint64_t token) = 0;
    // run ingest, pyramids and detector on a device queue of their own, beside the tracking work (a backend without
    // queues has nothing to do)
    virtual int set_front_queue(bool) { return FT_OK; }
    // launch what the backend holds back in the hope of a detector call (a backend that holds nothing back has nothing to do)
    virtual int flush() { return FT_OK; }
    virtual const char* name() const = 0;
};

}  // namespace feature_tracker
