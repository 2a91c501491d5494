// CPU checker behind the tracker's backend interface (TEST INFRASTRUCTURE, PARITY UNPINNED — see ft_oracle.h).
//
// Runs the same host orchestration (TrackKLT / TrackKLTBatch) on the scalar restatement of the reference path, so
// tests can compare whole tracking runs and bench.py can time "the reference CPU path" on the box's host cores.
// Threading mirrors the reference: independent images in parallel (src/track/TrackKLT.cpp:51-58), LK parallel over
// features (cv::parallel_for_ inside calcOpticalFlowPyrLK), FAST parallel over grid cells (src/track/Grider_FAST.h:94).
// Never linked into the product libraries.
#include <cstring>
#include <thread>
#include <vector>

#include "ft_oracle.h"
#include "track/FrontendBackend.h"

using namespace feature_tracker;

namespace {

int g_threads = 1;

template <class F>
void parallel_for(int n, int threads, F f) {
    if (threads <= 1 || n <= 1) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    threads = std::min(threads, n);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=]() { for (int i = t; i < n; i += threads) f(i); });
    for (auto& t : th) t.join();
}

class OracleBackend : public FrontendBackend {
public:
    explicit OracleBackend(const ft_limits& lim) : lim_(lim), slots_((size_t)lim.n_streams * lim.n_slots) {}
    ~OracleBackend() override {
        for (auto& s : slots_) fto_pyramid_free(s.pyr);
    }

    int upload(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) override {
        Slot* S = at(stream, slot);
        if (!S || !host || w < 1 || h < 1) return FT_E_INVALID;
        if (w > lim_.max_width || h > lim_.max_height) return FT_E_LIMIT;
        S->img.resize((size_t)w * h);
        for (int y = 0; y < h; y++) memcpy(S->img.data() + (size_t)y * w, host + (size_t)y * stride, (size_t)w);
        S->w = w; S->h = h;
        fto_pyramid_free(S->pyr);
        S->pyr = nullptr;
        return FT_OK;
    }
    // the checker has no device: a "device" frame is a host pointer here
    int set_frames_device(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride) override {
        for (int i = 0; i < n; i++) {
            const int rc = upload(streams[i], slots[i], (const uint8_t*)dev[i], w, h, stride);
            if (rc != FT_OK) return rc;
        }
        return FT_OK;
    }
    int set_frames_device_pyramid(int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride,
                                  int max_level, int win_w, int win_h, int* achieved) override {
        const int rc = set_frames_device(n, streams, slots, dev, w, h, stride);
        return rc != FT_OK ? rc : build_pyramid(n, streams, slots, max_level, win_w, win_h, 0, achieved);
    }
    int stage_frame(int stream, int slot, const uint8_t* host, int w, int h, size_t stride) override {
        return upload(stream, slot, host, w, h, stride);   // slots are separate objects: safe from several threads
    }
    int set_frames_staged_pyramid(int n, const int* streams, const int* slots, int, int, int max_level, int win_w, int win_h,
                                  int* achieved) override {
        return build_pyramid(n, streams, slots, max_level, win_w, win_h, 0, achieved);
    }
    int build_pyramid(int n, const int* streams, const int* slots, int max_level, int win_w, int win_h, int, int* achieved) override {
        for (int i = 0; i < n; i++) { Slot* S = at(streams[i], slots[i]); if (!S || S->img.empty()) return FT_E_STATE; }
        parallel_for(n, g_threads, [&](int i) {
            Slot* S = at(streams[i], slots[i]);
            fto_pyramid_free(S->pyr);
            S->pyr = fto_pyramid_build(S->img.data(), S->w, S->h, (size_t)S->w, win_w, win_h, max_level);
            if (achieved) achieved[i] = fto_pyramid_levels(S->pyr) - 1;
        });
        return FT_OK;
    }
    int lk_track_async(int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm) override {
        fto_lk_params p;
        p.win_w = prm->win_w; p.win_h = prm->win_h; p.max_level = prm->max_level; p.max_iter = prm->max_iter;
        p.eps = prm->eps; p.min_eig = prm->min_eig; p.flags = prm->flags; p.fb_thresh = prm->fb_thresh;
        for (int j = 0; j < n_jobs; j++) {
            Slot* A = at(jobs[j].stream_prev, jobs[j].slot_prev);
            Slot* B = at(jobs[j].stream_next, jobs[j].slot_next);
            if (!A || !B || !A->pyr || !B->pyr) return FT_E_STATE;
        }
        // few jobs: threads inside each job (over features); many jobs: one job per thread
        const bool outer = n_jobs >= g_threads;
        parallel_for(n_jobs, outer ? g_threads : 1, [&](int j) {
            const ft_lk_job& J = jobs[j];
            if (J.n <= 0) return;
            fto_lk_params pj = p;
            const bool stereo = J.stereo_Rt && J.calib_prev && J.calib_next;
            if (stereo) {   // stereo seed from the extrinsic rotation
                fto_stereo_seed(J.prev_xy, J.n, J.calib_prev, J.calib_next, J.stereo_Rt, J.next_xy);
                pj.flags |= FTO_LK_USE_INITIAL_FLOW;
            } else if (J.H) {   // gyro seed: initial guess = pi(H * [prev;1])
                for (int i = 0; i < J.n; i++) {
                    const float x = J.prev_xy[2 * i], y = J.prev_xy[2 * i + 1];
                    const float p0 = J.H[0] * x + J.H[1] * y + J.H[2];
                    const float p1 = J.H[3] * x + J.H[4] * y + J.H[5];
                    const float p2 = J.H[6] * x + J.H[7] * y + J.H[8];
                    J.next_xy[2 * i] = p0 / p2; J.next_xy[2 * i + 1] = p1 / p2;
                }
                pj.flags |= FTO_LK_USE_INITIAL_FLOW;
            }
            fto_lk_track(at(J.stream_prev, J.slot_prev)->pyr, at(J.stream_next, J.slot_next)->pyr, J.prev_xy, J.next_xy, J.status,
                         J.err, J.n, &pj, outer ? 1 : g_threads, J.iters);
            if (stereo)
                fto_stereo_check(J.prev_xy, J.next_xy, J.status, J.n, J.calib_prev, J.calib_next, J.stereo_Rt, at(J.stream_prev, J.slot_prev)->w,
                                 at(J.stream_prev, J.slot_prev)->h, prm->stereo_threshold);
            if ((prm->flags & FT_LK_REJECT_OUTLIERS) && J.calib_prev && J.calib_next)
                fto_reject_outliers(J.prev_xy, J.next_xy, J.status, J.n, J.calib_prev, J.calib_next, prm->ransac_hypotheses, prm->ransac_seed, nullptr);
        });
        return FT_OK;
    }
    int fast_grid_async(int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y, int per_cell,
                        ft_keypoint* out, int cap, int* n_out, int* cell_counts) override {
        for (int i = 0; i < n; i++) { Slot* S = at(streams[i], slots[i]); if (!S || S->img.empty()) return FT_E_STATE; }
        const bool outer = n >= g_threads;
        std::vector<int> rcs((size_t)n, 0);
        parallel_for(n, outer ? g_threads : 1, [&](int i) {
            Slot* S = at(streams[i], slots[i]);
            const int size_x = S->w / grid_x, size_y = S->h / grid_y;
            const int cells = (size_x > 0 && size_y > 0) ? (S->w / size_x) * (S->h / size_y) : 0;
            static_assert(sizeof(ft_keypoint) == sizeof(fto_keypoint), "keypoint layouts must match");
            rcs[i] = fto_grid_fast_cells(S->img.data(), S->w, S->h, (size_t)S->w, per_cell, grid_x, grid_y, threshold, 1, 0,
                                         outer ? 1 : g_threads, reinterpret_cast<fto_keypoint*>(out + (size_t)i * cap), cap,
                                         cell_counts ? cell_counts + (size_t)i * cells : nullptr);
            if (n_out) n_out[i] = rcs[i] < 0 ? 0 : rcs[i];
        });
        for (int r : rcs) if (r < 0) return FT_E_INVALID;
        return FT_OK;
    }
    int compute_descriptors(int stream, int slot, const float* xy, int n, uint8_t* desc, uint8_t* valid) override {
        Slot* S = at(stream, slot);
        if (!S || n < 0) return FT_E_INVALID;
        if (S->img.empty()) return FT_E_STATE;
        return fto_compute_descriptors(S->img.data(), S->w, S->h, (size_t)S->w, xy, n, desc, valid) == 0 ? FT_OK : FT_E_INVALID;
    }
    int match_descriptors(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, ft_match* out, int cap, int* n_out) override {
        std::vector<int> qtd(3 * (size_t)std::max(cap, 1));
        const int n = fto_match_descriptors(desc0, n0, desc1, n1, knn_ratio, qtd.data(), cap);
        for (int i = 0; i < n && i < cap; i++) { out[i].query = qtd[3 * i]; out[i].train = qtd[3 * i + 1]; out[i].distance = (float)qtd[3 * i + 2]; }
        *n_out = n;
        return FT_OK;
    }
    int reject_outliers(int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed) override {
        for (int j = 0; j < n_sets; j++)
            if (sets[j].n > 0)
                fto_reject_outliers(sets[j].prev_xy, sets[j].next_xy, sets[j].status, sets[j].n, sets[j].calib_prev, sets[j].calib_next, hypotheses, seed, nullptr);
        return FT_OK;
    }
    void homography(const float* K4, const float* R, float* H) override { fto_homography(K4, R, H); }
    int sync() override { return FT_OK; }
    // every call of this backend completes before it returns
    int fence(uint64_t* token) override { *token = 1; return FT_OK; }
    int wait(uint64_t) override { return FT_OK; }
    const char* name() const override { return "cpu-oracle"; }

private:
    struct Slot {
        std::vector<uint8_t> img;
        int w = 0, h = 0;
        fto_pyramid* pyr = nullptr;
    };
    Slot* at(int stream, int slot) {
        if (stream < 0 || stream >= lim_.n_streams || slot < 0 || slot >= lim_.n_slots) return nullptr;
        return &slots_[(size_t)stream * lim_.n_slots + slot];
    }
    ft_limits lim_;
    std::vector<Slot> slots_;
};

}  // namespace

FrontendBackend* ftk_make_backend(int, const ft_limits& limits, void** device_ctx) {
    if (device_ctx) *device_ctx = nullptr;
    return new OracleBackend(limits);
}

const char* ftk_backend_label() { return "cpu-oracle"; }

extern "C" void fto_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
extern "C" int fto_get_threads(void) { return g_threads; }
