// Drives many independent monocular TrackKLT trackers (one camera stream each) through shared batched launches:
// one pyramid launch per level, one detector call and one LK launch cover every stream of the batch, and the host
// waits for the device once per frame set.  Independent streams are the parallel axis of this front-end; there is
// no counterpart in the reference, which runs one stereo rig per process with a host thread per camera
// (src/track/TrackKLT.cpp:51-58, :91-98).
#pragma once
#include <chrono>
#include <functional>
#include <memory>
#include <vector>

#include "../utils/WorkerPool.h"
#include "TrackKLT.h"

namespace feature_tracker {

class TrackKLTBatch {
public:
    // host_threads: threads for the per-stream bookkeeping, the calling thread included (0 = choose, at most 8)
    // slot_ring: pyramid slots per stream in the backend; from 3 on, stage() may ingest the next frame set while the
    // current one is still being tracked
    TrackKLTBatch(FrontendBackend* backend, const FeatureTrackerOptions& opt, int n_streams, int host_threads = 0, int slot_ring = 2)
        : be(backend), options(opt), ring(slot_ring < 2 ? 2 : slot_ring) {
        for (int s = 0; s < n_streams; s++) {
            trackers.emplace_back(new TrackKLT(backend, opt, 0, s));
            trackers.back()->set_slot_ring(ring);
        }
        plans.resize(n_streams);
        db_plans.resize(n_streams);
        if (host_threads <= 0) host_threads = default_host_threads(n_streams);
        own_pool.reset(new WorkerPool(host_threads - 1));
        pool = own_pool.get();
    }

    ~TrackKLTBatch() {
        // Undelivered asynchronous results (a set between begin() and complete(), a staged lookahead set, a late
        // detector) leave the backend holding pointers into plans[], det[] and last_buf: let it deliver them while
        // those arrays still exist.
        if (pending || staged || late_pending) (void)be->sync();
    }

    static int default_host_threads(int n_streams) {
        const int hw = (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(std::min(hw > 0 ? hw : 1, 8), n_streams / 4));
    }

    int size() const { return (int)trackers.size(); }
    TrackKLT& tracker(int s) { return *trackers[s]; }

    // one frame per stream, all of the same size; returns FT_OK or a device error code
    int feed(const double* timestamps, const FrameSource* frames) {
        int rc = begin(timestamps, frames);
        if (rc == FT_OK) rc = complete();
        flush_db();
        return rc;
    }
    // write the observations of the last completed frame set to the feature databases (begin() does this for the
    // previous set after it has queued the device work of the new one)
    void flush_db() {
        if (!db_pending) return;
        db_pending = false;
        commit_db(db_plans);
    }
    // hand the pending observations to the caller instead (e.g. to a writer thread); returns false if there are none
    bool take_db(std::vector<TrackKLT::MonoPlan>& out) {
        if (!db_pending) return false;
        db_pending = false;
        out.resize(db_plans.size());
        for (size_t s = 0; s < db_plans.size(); s++) {
            out[s].db_ids.swap(db_plans[s].db_ids); out[s].db_uvn.swap(db_plans[s].db_uvn);
            out[s].timestamp = db_plans[s].timestamp; out[s].cam = db_plans[s].cam; out[s].db_pending = db_plans[s].db_pending;
            db_plans[s].db_pending = false;
        }
        return true;
    }
    // apply observations taken with take_db; reads the (fixed) calibration and writes the feature databases, which carry their own locks, so it
    // may run on another thread than begin()/complete() as long as one stream's sets are applied in order
    // (part, parts): several threads can share one payload.  With at least as many streams as threads, thread `part`
    // applies the streams s with s % parts == part; with fewer (a few large streams), every thread applies its share of
    // every stream's observations (split by feature id, see FeatureDatabase::owner_of).
    void commit_db(std::vector<TrackKLT::MonoPlan>& payload, int part = 0, int parts = 1) {
        // a plain loop on purpose: the worker pool belongs to the thread that runs begin()/complete()
        if (parts == 1 || payload.size() >= (size_t)parts) {
            for (size_t s = (size_t)part; s < payload.size(); s += (size_t)parts) trackers[s]->mono_commit(payload[s]);
        } else {
            for (size_t s = 0; s < payload.size(); s++) trackers[s]->mono_commit(payload[s], part, parts);
        }
    }
    bool in_flight() const { return pending; }

    // Optional lookahead between begin() and complete() of the set in flight: enqueue everything of the NEXT frame set
    // that does not depend on the current set's tracking results (ingest, pyramids, speculative detector), into the
    // ring slots that set will use.  The device then has work queued while the host finishes the current set, and the
    // following begin() with the same timestamps and frames finds that part done.  Needs a slot ring of 3.
    bool can_stage() const { return ring >= 3 && pending && !staged && size() > 0 && trackers[0]->speculative(); }
    int stage(const double* timestamps, const FrameSource* frames) {
        if (!can_stage()) return FT_OK;
        const auto t_stage = std::chrono::steady_clock::now();
        const int S = size();
        staged_slots.resize(S);
        for (int s = 0; s < S; s++) staged_slots[s] = trackers[s]->slot_after(plans[s]);
        staged_det = det_cur ^ 1;
        const int rc = enqueue_front(timestamps, frames, staged_slots, det[staged_det], true, nullptr);
        if (rc != FT_OK) return rc;
        staged_ts.assign(timestamps, timestamps + S);
        staged_frames.assign(frames, frames + S);
        staged = true;
        // the staged set's ingest / pyramid / detector enqueue is booked under the pyramid stage
        stage_s[ST_PYRAMID] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count();
        return FT_OK;
    }

    // first half of feed(): host preparation and every device launch of the frame set; returns without waiting.
    // A driver that owns several batches on separate device contexts overlaps one batch's device work with another
    // batch's host bookkeeping by interleaving begin() and complete() calls.
    // late_detect (speculative detection only): queue the detector of the new frames AFTER their tracking launch.
    // complete() then waits for the tracking results alone, and the ranked detections are collected at the start of
    // the next begin(), where they are first needed (or in complete(), if a stream starts from nothing and needs
    // them at once).  For a caller that feeds one set at a time and waits for it, this takes the detector out of
    // the frame's latency; results are the same.
    int begin(const double* timestamps, const FrameSource* frames, bool late_detect = false) {
        const int S = size();
        if (S == 0) return FT_OK;
        using clk = std::chrono::steady_clock;
        auto mark = clk::now();
        const auto t_begin = mark;
        auto lap = [&](int stage) {
            const auto now = clk::now();
            stage_s[stage] += std::chrono::duration<double>(now - mark).count();
            mark = now;
        };
        TrackKLT& t0 = *trackers[0];
        const int gx = t0.grid_cols(), gy = t0.grid_rows(), kmax = t0.per_cell_max();
        const int cells = Grider_FAST::cells_for(frames[0].cols, frames[0].rows, gx, gy);
        streams.resize(S); cur_slots.resize(S);
        int rc;
        for (int s = 0; s < S; s++) {
            trackers[s]->mono_begin(timestamps[s], frames[s].cols, frames[s].rows, 0, plans[s]);
            streams[s] = plans[s].stream; cur_slots[s] = plans[s].slot_cur;
        }
        // was this set staged while the previous one was in flight?
        bool front_done = false;
        if (staged) {
            front_done = true;
            for (int s = 0; s < S && front_done; s++)
                front_done = staged_ts[s] == timestamps[s] && staged_slots[s] == cur_slots[s] && staged_frames[s].dev == frames[s].dev &&
                             staged_frames[s].host == frames[s].host && staged_frames[s].cols == frames[s].cols &&
                             staged_frames[s].rows == frames[s].rows && staged_frames[s].step == frames[s].step;
            staged = false;
            // a stale staging only wrote ring slots this set overwrites; its detector output lands in det[staged_det]
            det_cur = front_done ? staged_det : (staged_det ^ 1);
            if (late_pending && det_cur == late_idx && (rc = resolve_late()) != FT_OK) return rc;
        } else if (late_pending && det_cur == late_idx) {
            det_cur ^= 1;   // det[late_idx] still receives the previous set's detector output
        }
        const bool late = late_detect && t0.speculative() && !front_done && cells > 0;
        if (!front_done) {
            const std::function<void(int)> lap_fn = lap;
            if ((rc = enqueue_front(timestamps, frames, cur_slots, det[det_cur], t0.speculative() && !late, &lap_fn)) != FT_OK) return rc;
        } else {
            lap(ST_INGEST);
        }
        lap(ST_PYRAMID);
        // the previous set's late detector output: it ran while this set's frames were ingested
        if (late_pending && (rc = resolve_late()) != FT_OK) return rc;
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

        // streams without previous points detect on the new frame and return (src/track/TrackKLT.cpp:63-72)
        starts_fresh.assign(S, 0);
        bool any_first = false;
        for (int s = 0; s < S; s++) {
            starts_fresh[s] = plans[s].slot_last < 0 || trackers[s]->tracked_last(0) == 0;
            any_first |= starts_fresh[s] != 0;
        }
        // non-speculative: the detector runs on the new frame only for streams that start from nothing
        DetSet& D = det[det_cur];
        if (!t0.speculative()) {
            D.who.clear();
            if (cells > 0 && any_first) {
                std::vector<int> d_streams, d_slots;
                for (int s = 0; s < S; s++)
                    if (starts_fresh[s]) { D.who.push_back(s); d_streams.push_back(plans[s].stream); d_slots.push_back(plans[s].slot_cur); }
                const int cap = cells * kmax;
                D.buf.resize((size_t)D.who.size() * cap); D.n.resize(D.who.size()); D.cells.resize((size_t)D.who.size() * cells);
                if ((rc = be->fast_grid_async((int)D.who.size(), d_streams.data(), d_slots.data(), t0.fast_threshold(), gx, gy, kmax,
                                              D.buf.data(), cap, D.n.data(), D.cells.data())) != FT_OK) return rc;
            }
        }
        lap(ST_DETECT);
        jobs.clear();
        all_jobs.resize(S);
        pool->parallel_for(S, [&](int s) { trackers[s]->mono_prepare(plans[s], last_det[s], all_jobs[s]); });
        for (int s = 0; s < S; s++)
            if (plans[s].run_lk) jobs.push_back(all_jobs[s]);
        lap(ST_PREPARE);
        lap(ST_DETECT);
        if (!jobs.empty()) {
            const ft_lk_params prm = t0.lk_params();
            if ((rc = be->lk_track_async((int)jobs.size(), jobs.data(), &prm)) != FT_OK) return rc;
        }
        // the wait in complete() covers this set's detector and LK, not what stage() enqueues afterwards
        if ((rc = be->fence(&wait_token)) != FT_OK) return rc;
        lap(ST_LK_ENQUEUE);
        if (late) {
            if ((rc = enqueue_detect(frames, cur_slots, det[det_cur])) != FT_OK) return rc;
            if ((rc = be->fence(&det_token)) != FT_OK) return rc;
            late_pending = true; late_idx = det_cur; late_cells = cells; late_kmax = kmax;
            lap(ST_DETECT);
        }
        flush_db();   // the previous set's database writes overlap the device work queued above
        lap(ST_FINISH);
        stage_s[ST_TOTAL] += std::chrono::duration<double>(mark - t_begin).count();
        pend_cells = cells; pend_kmax = kmax;
        pending = true;
        return FT_OK;
    }

    // second half of feed(): wait for the device, then masks, feature database and state roll of every stream
    int complete() {
        if (!pending) return FT_OK;
        pending = false;
        const int S = size();
        using clk = std::chrono::steady_clock;
        auto mark = clk::now();
        const auto t_begin = mark;
        auto lap = [&](int stage) {
            const auto now = clk::now();
            stage_s[stage] += std::chrono::duration<double>(now - mark).count();
            mark = now;
        };
        int rc;
        const int cells = pend_cells, kmax = pend_kmax;
        if ((rc = be->wait(wait_token)) != FT_OK) return rc;
        lap(ST_WAIT);
        std::vector<Detection>& cur_det = cur_det_buf;
        cur_det.assign(S, Detection());
        if (late_pending) {   // this set's detector is still queued behind its tracking launch
            bool any_first = false;
            for (int s = 0; s < S; s++) any_first |= plans[s].first;
            if (any_first) {   // a stream that starts from nothing takes its points from the detection right now
                if ((rc = be->wait(det_token)) != FT_OK) return rc;
                late_pending = false;
                lap(ST_WAIT);
            }
        }
        if (!late_pending) fill_detections(det[det_cur], cells, kmax, cur_det);
        pool->parallel_for(S, [&](int s) { trackers[s]->mono_finish(plans[s], cur_det[s]); });
        for (int s = 0; s < S; s++)
            if (plans[s].db_pending) { db_pending = true; break; }
        for (int s = 0; s < S && db_pending; s++) {   // hand the payload to the commit plans right away
            db_plans[s].db_pending = plans[s].db_pending;
            if (!plans[s].db_pending) continue;
            db_plans[s].db_ids.swap(plans[s].db_ids); db_plans[s].db_uvn.swap(plans[s].db_uvn);
            db_plans[s].timestamp = plans[s].timestamp; db_plans[s].cam = plans[s].cam;
            plans[s].db_pending = false;
        }
        tracked = 0;
        attempted = 0;
        for (int s = 0; s < S; s++) {
            if (!plans[s].run_lk) continue;
            attempted += plans[s].status.size();
            tracked += trackers[s]->tracked_last(0);
        }
        lap(ST_FINISH);
        stage_s[ST_TOTAL] += std::chrono::duration<double>(mark - t_begin).count();
        n_feeds++;
        return FT_OK;
    }

    // wall-clock seconds per stage accumulated over feeds (the reference's stage timers, src/track/TrackKLT.cpp:212-217,
    // adapted to an asynchronous device: enqueue cost per stage, then one wait for everything)
    enum Stage { ST_INGEST = 0, ST_PYRAMID, ST_DETECT, ST_PREPARE, ST_LK_ENQUEUE, ST_WAIT, ST_FINISH, ST_TOTAL, ST_COUNT };
    static const char* stage_name(int i) {
        static const char* names[ST_COUNT] = {"frame ingest", "pyramid enqueue", "detection (host + enqueue)", "klt prepare (top-off, job layout)",
                                              "klt enqueue", "device wait", "masks + feature DB update", "total"};
        return names[i];
    }
    void stage_times(double* out, long long* feeds, bool reset) {
        for (int i = 0; i < ST_COUNT; i++) out[i] = stage_s[i];
        if (feeds) *feeds = n_feeds;
        if (reset) { for (double& v : stage_s) v = 0; n_feeds = 0; }
    }

    size_t tracked_features() const { return tracked; }     // features with final status 1 in the last feed
    size_t attempted_features() const { return attempted; } // features handed to LK in the last feed

private:
    // detector output of one frame set (two sets exist: the one in flight and the staged one)
    struct DetSet {
        std::vector<ft_keypoint> buf;
        std::vector<int> n, cells, who;
    };

    // ingest + pyramids + (speculative) detector of one frame set into the given ring slots
    int enqueue_front(const double* timestamps, const FrameSource* frames, const std::vector<int>& slots, DetSet& D, bool detect,
                      const std::function<void(int)>* lap) {
        (void)timestamps;
        const int S = size();
        TrackKLT& t0 = *trackers[0];
        int rc = FT_OK;
        front_streams.resize(S);
        bool all_dev = true, all_host = true;
        for (int s = 0; s < S; s++) {
            front_streams[s] = s;   // tracker s owns backend stream s (stream_base = s, camera 0)
            all_host &= frames[s].dev == nullptr && frames[s].host != nullptr && frames[s].cols == frames[0].cols && frames[s].rows == frames[0].rows;
            all_dev &= frames[s].dev != nullptr && frames[s].cols == frames[0].cols && frames[s].rows == frames[0].rows && frames[s].step == frames[0].step;
        }
        if (all_dev) {   // frames already in HBM: ingest and pyramids in one pass over the frames
            dev_ptrs.resize(S);
            for (int s = 0; s < S; s++) dev_ptrs[s] = frames[s].dev;
            if (lap) (*lap)(ST_INGEST);
            if ((rc = be->set_frames_device_pyramid(S, front_streams.data(), slots.data(), dev_ptrs.data(), frames[0].cols, frames[0].rows,
                                                    frames[0].step, t0.max_level(), t0.window(), t0.window(), nullptr)) != FT_OK) return rc;
        } else if (all_host) {   // frames in host memory: staged by this thread and its helpers, then one pass over the staging
            stage_rc.assign(S, FT_OK);
            pool->parallel_for(S, [&](int s) {
                stage_rc[s] = be->stage_frame(front_streams[s], slots[s], frames[s].host, frames[s].cols, frames[s].rows, frames[s].step);
            });
            for (int s = 0; s < S; s++) if (stage_rc[s] != FT_OK) return stage_rc[s];
            if (lap) (*lap)(ST_INGEST);
            if ((rc = be->set_frames_staged_pyramid(S, front_streams.data(), slots.data(), frames[0].cols, frames[0].rows, t0.max_level(),
                                                    t0.window(), t0.window(), nullptr)) != FT_OK) return rc;
        } else {
            for (int s = 0; s < S; s++) {
                const FrameSource& f = frames[s];
                if (f.dev) rc = be->set_frames_device(1, &front_streams[s], &slots[s], &f.dev, f.cols, f.rows, f.step);
                else rc = be->upload(front_streams[s], slots[s], f.host, f.cols, f.rows, f.step);
                if (rc != FT_OK) return rc;
            }
            if (lap) (*lap)(ST_INGEST);
            if ((rc = be->build_pyramid(S, front_streams.data(), slots.data(), t0.max_level(), t0.window(), t0.window(), 0, nullptr)) != FT_OK) return rc;
        }
        D.who.clear();
        if (detect) return enqueue_detect(frames, slots, D);
        return be->flush();   // no detector call follows: the pyramid's last launches must not wait for the host's next step
    }

    // the ranked per-cell detector of every stream's frame in the given ring slots, one batched call
    int enqueue_detect(const FrameSource* frames, const std::vector<int>& slots, DetSet& D) {
        const int S = size();
        TrackKLT& t0 = *trackers[0];
        const int gx = t0.grid_cols(), gy = t0.grid_rows(), kmax = t0.per_cell_max();
        const int cells = Grider_FAST::cells_for(frames[0].cols, frames[0].rows, gx, gy);
        D.who.clear();
        if (cells <= 0) return FT_OK;
        const int cap = cells * kmax;
        front_streams.resize(S);
        D.who.resize(S);
        for (int s = 0; s < S; s++) { front_streams[s] = s; D.who[s] = s; }
        D.buf.resize((size_t)S * cap); D.n.resize(S); D.cells.resize((size_t)S * cells);
        return be->fast_grid_async(S, front_streams.data(), slots.data(), t0.fast_threshold(), gx, gy, kmax, D.buf.data(), cap, D.n.data(),
                                   D.cells.data());
    }

    static void fill_detections(const DetSet& D, int cells, int kmax, std::vector<Detection>& out) {
        const int cap = cells * kmax;
        for (size_t q = 0; q < D.who.size(); q++) {
            Detection& d = out[D.who[q]];
            d.valid = true; d.per_cell = kmax; d.cells = cells;
            d.kps = D.buf.data() + q * cap; d.n = D.n[q]; d.cell_counts = D.cells.data() + q * cells;
        }
    }

    // collect a late detector output and hand it to the trackers as the cached detection of their last frame
    int resolve_late() {
        const int rc = be->wait(det_token);
        late_pending = false;
        if (rc != FT_OK) return rc;
        const int S = size();
        late_det_buf.assign(S, Detection());
        fill_detections(det[late_idx], late_cells, late_kmax, late_det_buf);
        for (int s = 0; s < S; s++) trackers[s]->set_cached_detection(late_det_buf[s]);
        return FT_OK;
    }

    FrontendBackend* be;
    FeatureTrackerOptions options;
    int ring = 2;
    DetSet det[2];
    int det_cur = 0, staged_det = 1;
    bool staged = false;
    std::vector<double> staged_ts;
    std::vector<FrameSource> staged_frames;
    std::vector<int> staged_slots, front_streams, stage_rc;
    uint64_t wait_token = 0, det_token = 0;
    bool late_pending = false;
    int late_idx = 0, late_cells = 0, late_kmax = 0;
    std::vector<Detection> late_det_buf;
    std::vector<std::unique_ptr<TrackKLT>> trackers;
    std::vector<TrackKLT::MonoPlan> plans, db_plans;
    std::vector<char> starts_fresh;
    bool db_pending = false;
    std::vector<int> streams, cur_slots;
    std::vector<const void*> dev_ptrs;
    std::vector<ft_lk_job> jobs, all_jobs;
    WorkerPool* pool;
    std::unique_ptr<WorkerPool> own_pool;
    std::vector<Detection> cur_det_buf;
    bool pending = false;
    int pend_cells = 0, pend_kmax = 0;
    std::vector<ft_keypoint> last_buf;
    std::vector<int> last_n, last_cells;
    std::vector<std::vector<int>> cached_cells;
    size_t tracked = 0, attempted = 0;
    double stage_s[ST_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long n_feeds = 0;
};

}  // namespace feature_tracker
