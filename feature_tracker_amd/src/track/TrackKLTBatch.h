// Drives many independent monocular TrackKLT trackers (one camera stream each) through shared batched launches:
// one pyramid launch per level, one detector call and one LK launch cover every stream of the batch, and the host
// waits for the device once per frame set.  Independent streams are the parallel axis of this front-end; there is
// no counterpart in the reference, which runs one stereo rig per process with a host thread per camera
// (src/track/TrackKLT.cpp:51-58, :91-98).
#pragma once
#include <memory>
#include <vector>

#include "TrackKLT.h"

namespace feature_tracker {

class TrackKLTBatch {
public:
    TrackKLTBatch(FrontendBackend* backend, const FeatureTrackerOptions& opt, int n_streams) : be(backend), options(opt) {
        for (int s = 0; s < n_streams; s++) trackers.emplace_back(new TrackKLT(backend, opt, 0, s));
        plans.resize(n_streams);
    }

    int size() const { return (int)trackers.size(); }
    TrackKLT& tracker(int s) { return *trackers[s]; }

    // one frame per stream, all of the same size; returns FT_OK or a device error code
    int feed(const double* timestamps, const FrameSource* frames) {
        const int S = size();
        if (S == 0) return FT_OK;
        TrackKLT& t0 = *trackers[0];
        const int gx = t0.grid_cols(), gy = t0.grid_rows(), kmax = t0.per_cell_max();
        const int cells = Grider_FAST::cells_for(frames[0].cols, frames[0].rows, gx, gy);
        streams.resize(S); cur_slots.resize(S);
        int rc;
        for (int s = 0; s < S; s++) {
            if ((rc = trackers[s]->mono_begin(timestamps[s], frames[s], 0, plans[s])) != FT_OK) return rc;
            streams[s] = plans[s].stream; cur_slots[s] = plans[s].slot_cur;
        }
        if ((rc = be->build_pyramid(S, streams.data(), cur_slots.data(), t0.max_level(), t0.window(), t0.window(), 1, nullptr)) != FT_OK) return rc;

        // detections of the previous frames: cached (speculative) or computed now for the streams that need them
        std::vector<Detection> last_det(S);
        if (t0.speculative()) {
            cached_cells.resize(S);
            for (int s = 0; s < S; s++) {
                int pc = 0;
                const std::vector<ft_keypoint>& k = trackers[s]->cached_detection(pc, cached_cells[s]);
                if (pc > 0 && !cached_cells[s].empty()) {
                    last_det[s].valid = true; last_det[s].per_cell = pc; last_det[s].cells = (int)cached_cells[s].size();
                    last_det[s].kps = k.data(); last_det[s].n = (int)k.size(); last_det[s].cell_counts = cached_cells[s].data();
                }
            }
        } else if (cells > 0) {
            std::vector<int> want, w_streams, w_slots;
            int per_cell = 0;
            for (int s = 0; s < S; s++) {
                const int need = trackers[s]->mono_features_needed(plans[s]);
                if (need > 1) { want.push_back(s); w_streams.push_back(plans[s].stream); w_slots.push_back(plans[s].slot_last); per_cell = std::max(per_cell, need / (gx * gy) + 1); }
            }
            if (!want.empty()) {
                const int cap = cells * per_cell;
                last_buf.resize((size_t)want.size() * cap); last_n.resize(want.size()); last_cells.resize((size_t)want.size() * cells);
                if ((rc = be->fast_grid_async((int)want.size(), w_streams.data(), w_slots.data(), t0.fast_threshold(), gx, gy, per_cell,
                                              last_buf.data(), cap, last_n.data(), last_cells.data())) != FT_OK) return rc;
                if ((rc = be->sync()) != FT_OK) return rc;
                for (size_t q = 0; q < want.size(); q++) {
                    Detection& d = last_det[want[q]];
                    d.valid = true; d.per_cell = per_cell; d.cells = cells;
                    d.kps = last_buf.data() + q * cap; d.n = last_n[q]; d.cell_counts = last_cells.data() + q * cells;
                }
            }
        }

        jobs.clear();
        bool any_first = false;
        for (int s = 0; s < S; s++) {
            ft_lk_job j;
            trackers[s]->mono_prepare(plans[s], last_det[s], j);
            if (plans[s].run_lk) jobs.push_back(j);
            any_first |= plans[s].first;
        }
        // detector on the new frames: always when speculative, otherwise only for streams that start from nothing
        std::vector<Detection> cur_det(S);
        if (cells > 0 && (t0.speculative() || any_first)) {
            std::vector<int> d_streams, d_slots, who;
            for (int s = 0; s < S; s++)
                if (t0.speculative() || plans[s].first) { who.push_back(s); d_streams.push_back(plans[s].stream); d_slots.push_back(plans[s].slot_cur); }
            const int cap = cells * kmax;
            det_buf.resize((size_t)who.size() * cap); det_n.resize(who.size()); det_cells.resize((size_t)who.size() * cells);
            if ((rc = be->fast_grid_async((int)who.size(), d_streams.data(), d_slots.data(), t0.fast_threshold(), gx, gy, kmax,
                                          det_buf.data(), cap, det_n.data(), det_cells.data())) != FT_OK) return rc;
            det_who = who;
        } else {
            det_who.clear();
        }
        if (!jobs.empty()) {
            const ft_lk_params prm = t0.lk_params();
            if ((rc = be->lk_track_async((int)jobs.size(), jobs.data(), &prm)) != FT_OK) return rc;
        }
        if ((rc = be->sync()) != FT_OK) return rc;
        const int cap = cells * kmax;
        for (size_t q = 0; q < det_who.size(); q++) {
            Detection& d = cur_det[det_who[q]];
            d.valid = true; d.per_cell = kmax; d.cells = cells;
            d.kps = det_buf.data() + q * cap; d.n = det_n[q]; d.cell_counts = det_cells.data() + q * cells;
        }
        tracked = 0;
        attempted = 0;
        for (int s = 0; s < S; s++) {
            const bool ran = plans[s].run_lk;
            if (ran) attempted += plans[s].status.size();
            trackers[s]->mono_finish(plans[s], cur_det[s]);
            if (ran) tracked += trackers[s]->tracked_last(0);
        }
        return FT_OK;
    }

    size_t tracked_features() const { return tracked; }     // features with final status 1 in the last feed
    size_t attempted_features() const { return attempted; } // features handed to LK in the last feed

private:
    FrontendBackend* be;
    FeatureTrackerOptions options;
    std::vector<std::unique_ptr<TrackKLT>> trackers;
    std::vector<TrackKLT::MonoPlan> plans;
    std::vector<int> streams, cur_slots, det_who;
    std::vector<ft_lk_job> jobs;
    std::vector<ft_keypoint> det_buf, last_buf;
    std::vector<int> det_n, det_cells, last_n, last_cells;
    std::vector<std::vector<int>> cached_cells;
    size_t tracked = 0, attempted = 0;
};

}  // namespace feature_tracker
