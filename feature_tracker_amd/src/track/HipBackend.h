// FrontendBackend bound to the gfx950 device library through its C ABI.
#pragma once
#include <stdexcept>
#include <string>

#include "FrontendBackend.h"

namespace feature_tracker {

class HipBackend : public FrontendBackend {
public:
    HipBackend(int device, const ft_limits& limits) {
        const int rc = ft_ctx_create(&ctx_, device, &limits);
        if (rc != FT_OK) throw std::runtime_error(std::string("ft_ctx_create: ") + ft_strerror(rc));
    }
    ~HipBackend() override { ft_ctx_destroy(ctx_); }
    HipBackend(const HipBackend&) = delete;
    HipBackend& operator=(const HipBackend&) = delete;

    int upload(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) override {
        return ft_upload(ctx_, stream, slot, host, w, h, stride);
    }
    int set_frames_device(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride) override {
        return ft_set_frames_device(ctx_, n, streams, slots, dev, w, h, stride);
    }
    int set_frames_device_pyramid(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride,
                                  int max_level, int win_w, int win_h, int* achieved) override {
        return ft_set_frames_device_pyramid(ctx_, n, streams, slots, dev, w, h, stride, max_level, win_w, win_h, achieved);
    }
    int stage_frame(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) override {
        return ft_stage_frame(ctx_, stream, slot, host, w, h, stride);
    }
    int set_frames_staged_pyramid(int n, const int* streams, const int* slots, int w, int h, int max_level, int win_w, int win_h,
                                  int* achieved) override {
        return ft_set_frames_staged_pyramid(ctx_, n, streams, slots, w, h, max_level, win_w, win_h, achieved);
    }
    int build_pyramid(int n, const int* streams, const int* slots, int max_level, int win_w, int win_h, int with_gradients,
                      int* achieved) override {
        return ft_build_pyramid(ctx_, n, streams, slots, max_level, win_w, win_h, with_gradients, achieved);
    }
    int lk_track_async(int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm) override {
        return ft_lk_track_async(ctx_, n_jobs, jobs, prm);
    }
    int fast_grid_async(int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y, int per_cell,
                        ft_keypoint* out, int cap, int* n_out, int* cell_counts) override {
        return ft_fast_grid_async(ctx_, n, streams, slots, threshold, grid_x, grid_y, per_cell, out, cap, n_out, cell_counts);
    }
    int compute_descriptors(int stream, int slot, const float* xy, int n, uint8_t* desc, uint8_t* valid) override {
        return ft_compute_descriptors(ctx_, stream, slot, xy, n, desc, valid);
    }
    int match_descriptors(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, ft_match* out, int cap, int* n_out) override {
        return ft_match_descriptors(ctx_, desc0, n0, desc1, n1, knn_ratio, out, cap, n_out);
    }
    int reject_outliers(int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed) override {
        return ft_reject_outliers(ctx_, n_sets, sets, hypotheses, seed);
    }
    void homography(const float* K4, const float* R_p_c, float* H) override { ft_homography(K4, R_p_c, H); }
    int sync() override { return ft_sync(ctx_); }
    int fence(uint64_t* token) override { return ft_fence(ctx_, token); }
    int wait(uint64_t token) override { return ft_wait(ctx_, token); }
    int set_front_queue(bool on) override { return ft_set_front_queue(ctx_, on ? 1 : 0); }
    int flush() override { return ft_flush(ctx_); }
    const char* name() const override { return "hip-gfx950"; }
    ft_ctx* ctx() { return ctx_; }

private:
    ft_ctx* ctx_ = nullptr;
};

}  // namespace feature_tracker
