// C wrapper of the host tracker layer (include/ft_tracker.h).  The backend is created by ftk_make_backend(),
// which each library that compiles this file defines: the product library binds the gfx950 device layer.
#include "ft_tracker.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <new>

#include "FeatureTrackerOptions.h"
#include "track/TrackDescriptor.h"
#include "track/TrackKLT.h"
#include "track/TrackKLTBatch.h"

using namespace feature_tracker;

// defined once per library: which FrontendBackend the trackers of that library run on
FrontendBackend* ftk_make_backend(int device, const ft_limits& limits, void** device_ctx);
const char* ftk_backend_label();

namespace {

FeatureTrackerOptions to_options(const ftk_options& o) {
    FeatureTrackerOptions r;
    r.num_pts = o.num_pts; r.fast_threshold = o.fast_threshold; r.grid_x = o.grid_x; r.grid_y = o.grid_y;
    r.min_px_dist = o.min_px_dist; r.pyr_levels = o.pyr_levels; r.win_size = o.win_size; r.lk_max_iter = o.lk_max_iter;
    r.lk_eps = o.lk_eps; r.lk_min_eig = o.lk_min_eig; r.use_gyro_seed = o.use_gyro_seed != 0;
    r.forward_backward = o.forward_backward != 0; r.fb_thresh = o.fb_thresh; r.speculative_detect = o.speculative_detect != 0;
    r.reject_outliers = o.reject_outliers != 0; if (o.ransac_hypotheses > 0) r.ransac_hypotheses = o.ransac_hypotheses;
    r.stereo_extrinsic_seed = o.stereo_extrinsic_seed != 0; r.stereo_threshold = o.stereo_threshold;
    r.drop_fraction = o.drop_fraction;
    r.use_klt = o.use_klt != 0; if (o.knn_ratio > 0) r.knn_ratio = o.knn_ratio;
    return r;
}

ft_limits limits_for(const ftk_options& o, int n_streams, int w, int h, int n_slots = 2) {
    ft_limits L;
    L.max_width = w; L.max_height = h;
    L.max_levels = o.pyr_levels + 1;
    L.n_streams = n_streams;
    L.n_slots = n_slots;
    L.max_points = n_streams * (o.num_pts + o.grid_x * o.grid_y + 64);
    // a cell can be narrower than cols/grid_x suggests only in degenerate geometry; leave room for that
    L.max_grid_cells = (o.grid_x + 2) * (o.grid_y + 2);
    L.max_per_cell = o.num_pts / (o.grid_x * o.grid_y) + 1;
    return L;
}

bool valid_options(const ftk_options* o) {
    return o && o->num_pts > 0 && o->grid_x > 0 && o->grid_y > 0 && o->min_px_dist > 0 && o->pyr_levels >= 0 &&
           o->pyr_levels < FT_MAX_LEVELS && o->win_size >= 3 && o->win_size <= FT_MAX_WIN;
}

int get_points(TrackBase& t, int cam, float* xy, uint64_t* ids, int cap) {
    const std::vector<KeyPoint> p = t.get_last_points((size_t)cam);
    const std::vector<size_t> id = t.get_last_ids((size_t)cam);
    for (size_t i = 0; i < p.size() && (int)i < cap; i++) {
        if (xy) { xy[2 * i] = p[i].pt.x; xy[2 * i + 1] = p[i].pt.y; }
        if (ids) ids[i] = (uint64_t)id[i];
    }
    return (int)p.size();
}

void calib_maps(int n_cams, const double* calib8, const double* T, std::map<size_t, Calib8>& K, std::map<size_t, Mat4d>& E) {
    for (int c = 0; c < n_cams; c++) {
        Calib8 k;
        Mat4d e;
        for (int i = 0; i < 8; i++) k[i] = calib8[8 * c + i];
        for (int i = 0; i < 16; i++) e[i] = T ? T[16 * c + i] : ((i % 5 == 0) ? 1.0 : 0.0);
        K[(size_t)c] = k; E[(size_t)c] = e;
    }
}

}  // namespace

struct ftk_tracker {
    std::unique_ptr<FrontendBackend> be;
    std::unique_ptr<TrackBase> trk;   // TrackKLT, or TrackDescriptor when the options say use_klt = 0 (src/test_tracking.cpp:229-230)
    TrackKLT* klt = nullptr;          // == trk.get() for a KLT tracker
};

// One stream group = one device context + one TrackKLTBatch + one host driver thread.  Groups share nothing, so
// their host bookkeeping and their device queues run concurrently; frames are posted to a mailbox ring per group
// (MAIL_RING deep) so that a group always finds the next set waiting: it stages that set's device work while it
// tracks the current one, even when another group lags behind and delays the caller.
struct ftk_group {
    std::unique_ptr<FrontendBackend> be;
    std::unique_ptr<TrackKLTBatch> batch;
    void* device_ctx = nullptr;
    int first = 0, n = 0;

    struct Mail {
        std::vector<double> ts;
        std::vector<FrameSource> frames;
        long long tracked = 0, attempted = 0;
        int rc = FT_OK;
    };
    static constexpr int MAIL_RING = 8;          // power of two, >= pipe_depth() + 2
    // sets that may be in flight behind the one the caller collects (FT_PIPE_DEPTH overrides it, 1..6)
    static int pipe_depth() {
        static const int d = [] {
            const char* e = getenv("FT_PIPE_DEPTH");
            const int v = e ? atoi(e) : 3;   // 3: the 3840x2160 single stream holds its rate from region to region (2: -10 % in two of three)
            return v < 1 ? 1 : (v > MAIL_RING - 2 ? MAIL_RING - 2 : v);
        }();
        return d;
    }
    Mail mail[MAIL_RING];
    std::atomic<int> spin_us{20};   // how long a waiting side polls before it goes to sleep (microseconds)
    // The next frame set's ingest, pyramids and detector on a device queue of their own, beside this set's tracking:
    // asked for by the pipelined feed of a single large stream, applied by the driver thread, which owns the device
    // context.
    std::atomic<int> front_wanted{0};
    int front_now = 0;
    std::thread driver;

    // feature-database writers: nothing in tracking depends on the database, so its updates run on their own threads,
    // off the driver's critical path.  Writer w applies the streams s with s % N_WRITERS == w of every frame set, in
    // frame order, so one stream's observations are always applied in order by one thread.
    static constexpr int MAX_WRITERS = 4;
    // Writer threads of this group: two, or four for streams of a thousand features and more (their databases have four
    // shards, and two writers fall behind a 10 000-feature stream: 0.3 ms per frame's observations, wake-up included,
    // against a frame every 0.26 ms; the backlog then waits for the caller at the end of a burst).  FT_DB_WRITERS (1..4)
    // overrides it.  Set before start().
    int N_WRITERS = 2;
    void choose_writers(int features_per_stream) {
        const char* e = getenv("FT_DB_WRITERS");
        const int v = e ? atoi(e) : (features_per_stream >= 1024 ? MAX_WRITERS : 2);
        N_WRITERS = v < 1 ? 1 : (v > MAX_WRITERS ? MAX_WRITERS : v);
    }
    struct DbJob {
        std::vector<TrackKLT::MonoPlan> payload;
        std::atomic<int> remaining{0};
        int parts = 2;              // threads that share the payload (streams s with s % parts == p go to thread p)
    };
    std::thread writers[MAX_WRITERS];
    std::mutex wm;
    std::condition_variable wcv;
    std::deque<DbJob*> wq[MAX_WRITERS];
    std::deque<std::unique_ptr<DbJob>> wfree;
    std::vector<std::unique_ptr<DbJob>> wbusy;   // jobs in flight (owned here, referenced by the queues)
    long w_posted = 0;
    std::atomic<long> w_done{0};
    std::atomic<long> w_queued{0};   // jobs handed to the writers so far (what a writer polls before it goes to sleep)
    bool w_stop = false;

    // one thread's share of a job is written; the last one recycles the job
    void finish_part(DbJob* job) {
        if (job->remaining.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        {
            std::lock_guard<std::mutex> lk(wm);
            for (auto& b : wbusy)
                if (b.get() == job) { wfree.push_back(std::move(b)); std::swap(b, wbusy.back()); wbusy.pop_back(); break; }
        }
        w_done.fetch_add(1, std::memory_order_release);
    }
    void writer_run(int w) {
        long seen = 0;   // jobs this writer has taken
        for (;;) {
            // a short poll bridges back-to-back sets (same rule as the driver: the spin is long only while somebody waits)
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us.load(std::memory_order_relaxed));
            while (w_queued.load(std::memory_order_acquire) <= seen && std::chrono::steady_clock::now() < spin_until) __builtin_ia32_pause();
            DbJob* job;
            {
                std::unique_lock<std::mutex> lk(wm);
                wcv.wait(lk, [&] { return w_stop || !wq[w].empty(); });
                if (wq[w].empty()) return;
                job = wq[w].front();
                wq[w].pop_front();
            }
            seen++;
            batch->commit_db(job->payload, w, job->parts);
            finish_part(job);
        }
    }
    // Handing a set's observations to the writers has a cheap half (take_db: the payload vectors change hands) and a
    // dear one (queue locks, waking the writer threads).  The driver does the cheap half right after complete() and the
    // dear one after it has launched the next set's tracking, so that the wake-ups are not on the path from one
    // tracking result to the next tracking launch.
    std::unique_ptr<DbJob> taken;   // payload taken, not yet queued
    void take_db_job() {
        std::unique_ptr<DbJob> job;
        {
            std::lock_guard<std::mutex> lk(wm);
            if (!wfree.empty()) { job = std::move(wfree.front()); wfree.pop_front(); }
        }
        if (!job) job.reset(new DbJob());
        job->parts = N_WRITERS;
        if (!batch->take_db(job->payload)) {
            std::lock_guard<std::mutex> lk(wm);
            wfree.push_back(std::move(job));
            return;
        }
        {
            std::lock_guard<std::mutex> lk(wm);
            w_posted++;   // wait_db() counts a taken set as posted: it must not return before that set is written
        }
        taken = std::move(job);
    }
    void queue_db_job() {
        if (!taken) return;
        taken->remaining.store(taken->parts, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(wm);
            for (int w = 0; w < N_WRITERS; w++) wq[w].push_back(taken.get());
            wbusy.push_back(std::move(taken));
        }
        w_queued.fetch_add(1, std::memory_order_release);
        wcv.notify_all();
    }
    void post_db() { take_db_job(); queue_db_job(); }
    // The last set of a burst, whose database writes nothing is left to hide: the calling thread takes a share of them
    // next to the writers and returns when all are done.  Earlier jobs are completed first, because a stream's share
    // changes hands here (s % (N_WRITERS + 1) instead of s % N_WRITERS) and its sets must be applied in order.
    void write_db_together() {
        wait_db();
        take_db_job();
        if (!taken) return;
        DbJob* job = taken.get();
        job->parts = N_WRITERS + 1;
        queue_db_job();
        batch->commit_db(job->payload, N_WRITERS, N_WRITERS + 1);
        finish_part(job);
        wait_db();
    }
    void wait_db() {
        long target;
        {
            std::lock_guard<std::mutex> lk(wm);
            target = w_posted;
        }
        const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us.load(std::memory_order_relaxed));
        while (w_done.load(std::memory_order_acquire) < target && std::chrono::steady_clock::now() < spin_until) __builtin_ia32_pause();
        while (w_done.load(std::memory_order_acquire) < target) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    std::mutex m;
    std::condition_variable cv;
    std::atomic<long> posted{0}, done{0};
    std::atomic<int> sleeping{0};
    bool stop = false;

    void start() {
        for (int w = 0; w < N_WRITERS; w++) writers[w] = std::thread([this, w] { writer_run(w); });
        driver = std::thread([this] { run(); });
    }
    void shutdown() {
        if (tm_n && getenv("FT_GROUP_TIMING"))
            fprintf(stderr, "[group %d+%d] %ld sets: mail %.1f begin %.1f stage %.1f complete %.1f after %.1f us\n", first, n, tm_n, tm[0] / tm_n,
                    tm[1] / tm_n, tm[2] / tm_n, tm[3] / tm_n, tm[4] / tm_n);
        for (const TraceRow& r : trace)
            fprintf(stderr, "[trace %d] set %ld wait %.0f mail %.0f begun %.0f staged %.0f completed %.0f end %.0f\n", first, r.seq, r.t[0], r.t[1], r.t[2], r.t[3],
                    r.t[4], r.t[5]);
        if (driver.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m);
                stop = true;
            }
            cv.notify_all();
            driver.join();
        }
        {
            std::lock_guard<std::mutex> lk(wm);
            w_stop = true;
        }
        wcv.notify_all();
        for (auto& t : writers)
            if (t.joinable()) t.join();
    }
    // publish frame set `seq` as complete and wake a caller that sleeps on it
    void publish(long seq) {
        // sequentially consistent store/load pair: either the waiter sees `done` or this thread sees the waiter
        done.store(seq);
        if (waiters.load()) {
            std::lock_guard<std::mutex> lk(dm);
            dcv.notify_all();
        }
    }
    // FT_GROUP_TIMING=1: mean microseconds per frame set the driver spends waiting for mail / in begin / stage / complete /
    // after complete, printed when the group shuts down (a diagnostic for the pipelined feed)
    double tm[5] = {0, 0, 0, 0, 0};
    long tm_n = 0;
    // FT_GROUP_TRACE=1: the same instants per frame set on the monotonic clock (microseconds), printed at shutdown
    struct TraceRow { long seq; double t[6]; };
    std::vector<TraceRow> trace;
    void run() {
        const bool timing = getenv("FT_GROUP_TIMING") != nullptr;
        const bool tracing = getenv("FT_GROUP_TRACE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        long unpublished = 0;   // a completed set whose publication was deferred behind the next set's tracking launch
        for (;;) {
            const long next = (unpublished ? unpublished : done.load(std::memory_order_relaxed)) + 1;
            const auto t0 = now();
            // a short spin bridges back-to-back frames; after that sleep: a thread that spins for long steals issue
            // slots from whatever shares its core (the other group's driver, the HIP runtime's own threads)
            // (a caller that feeds one set at a time and waits for it asks for a long spin: wake-ups would dominate its latency)
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us.load(std::memory_order_relaxed));
            while (posted.load(std::memory_order_acquire) < next && std::chrono::steady_clock::now() < spin_until) __builtin_ia32_pause();
            if (posted.load(std::memory_order_acquire) < next) {
                std::unique_lock<std::mutex> lk(m);
                sleeping.store(1, std::memory_order_release);
                cv.wait(lk, [&] { return stop || posted.load(std::memory_order_acquire) >= next; });
                sleeping.store(0, std::memory_order_release);
                if (stop) return;   // (nothing is ever left unpublished here: deferral needs the next set in the mailbox)
            }
            Mail& M = mail[next & (MAIL_RING - 1)];
            if (front_wanted.load(std::memory_order_relaxed) != front_now) {
                front_now = front_wanted.load(std::memory_order_relaxed);
                (void)be->set_front_queue(front_now != 0);
            }
            // nothing else waiting: the caller feeds one set at a time, so keep the detector out of this set's latency
            const bool late_detect = posted.load(std::memory_order_acquire) == next;
            const auto t1 = now();
            M.rc = batch->begin(M.ts.data(), M.frames.data(), late_detect);
            const auto t2 = now();
            // the previous set's deferred chores, now that this set's tracking is queued on the device
            queue_db_job();
            if (unpublished) { publish(unpublished); unpublished = 0; }
            // lookahead: if the following set is already in the mailbox, queue its ingest, pyramids and detector now so
            // that the device stays busy while this set's results are processed
            if (M.rc == FT_OK && posted.load(std::memory_order_acquire) > next) {
                Mail& N = mail[(next + 1) & (MAIL_RING - 1)];
                M.rc = batch->stage(N.ts.data(), N.frames.data());
            }
            const auto t3 = now();
            if (M.rc == FT_OK) M.rc = batch->complete();
            const auto t4 = now();
            M.tracked = (long long)batch->tracked_features();
            M.attempted = (long long)batch->attempted_features();
            const bool more = posted.load(std::memory_order_acquire) > next;
            if (late_detect && !more && spin_us.load(std::memory_order_relaxed) > 100) {
                // one set at a time, and the caller waits for it (the synchronous feed and flush set the long spin): a
                // hundred or so observations are written faster here than a writer thread wakes up
                if (n > 4 || batch->tracker(0).get_num_features() >= 1024) {
                    write_db_together();   // thousands of observations: shared with the writer threads
                } else {
                    wait_db();             // (sets handed to the writers earlier come first)
                    batch->flush_db();
                }
                publish(next);
            } else if (more && M.rc == FT_OK) {
                // The next set is waiting: its tracking launch comes first.  This set's observations change hands now
                // (begin() would otherwise write them itself), the writers are woken and the set is published after
                // that launch.  The caller cannot be waiting for this set to post the next one: it already has.
                take_db_job();
                unpublished = next;
            } else {
                post_db();   // this set's observations go to the database on the writer threads
                publish(next);
            }
            if (timing) { tm[0] += us(t0, t1); tm[1] += us(t1, t2); tm[2] += us(t2, t3); tm[3] += us(t3, t4); tm[4] += us(t4, now()); tm_n++; }
            if (tracing && trace.size() < 100000) {
                auto abs_us = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(t.time_since_epoch()).count(); };
                trace.push_back(TraceRow{next, {abs_us(t0), abs_us(t1), abs_us(t2), abs_us(t3), abs_us(t4), abs_us(now())}});
            }
        }
    }
    // hand frame set `seq` (= posted + 1) to the driver; set seq - MAIL_RING + 1 and older must have been collected
    void post(const double* ts, const FrameSource* fr) {
        const long seq = posted.load(std::memory_order_relaxed) + 1;
        Mail& M = mail[seq & (MAIL_RING - 1)];
        M.ts.assign(ts, ts + n);
        M.frames.assign(fr, fr + n);
        {
            std::lock_guard<std::mutex> lk(m);
            posted.store(seq, std::memory_order_release);
        }
        if (sleeping.load(std::memory_order_acquire)) cv.notify_all();
    }
    // caller side: block until set `seq` is complete (short spin first, then sleep on the condition variable)
    std::mutex dm;
    std::condition_variable dcv;
    std::atomic<int> waiters{0};
    Mail& wait(long seq) {
        const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us.load(std::memory_order_relaxed));
        while (done.load(std::memory_order_acquire) < seq && std::chrono::steady_clock::now() < spin_until) __builtin_ia32_pause();
        if (done.load(std::memory_order_acquire) < seq) {
            std::unique_lock<std::mutex> lk(dm);
            waiters.fetch_add(1);
            dcv.wait(lk, [&] { return done.load() >= seq; });
            waiters.fetch_sub(1, std::memory_order_acq_rel);
        }
        return mail[seq & (MAIL_RING - 1)];
    }
};

struct ftk_batch {
    std::vector<std::unique_ptr<ftk_group>> groups;
    std::vector<FrameSource> frames;
    int n_streams = 0;
    long seq = 0;          // frame sets posted so far
    long collected = 0;    // frame sets whose results were handed to the caller

    ~ftk_batch() {
        for (auto& g : groups) g->shutdown();
    }
    ftk_group* group_of(int stream, int* local) {
        for (auto& g : groups)
            if (stream >= g->first && stream < g->first + g->n) { *local = stream - g->first; return g.get(); }
        return nullptr;
    }
    int fill_frames(const void* const* fr, int on_device, int w, int h, size_t stride) {
        for (int s = 0; s < n_streams; s++) {
            if (!fr[s]) return FT_E_INVALID;
            FrameSource& f = frames[s];
            f = FrameSource();
            if (on_device) f.dev = fr[s]; else f.host = (const uint8_t*)fr[s];
            f.cols = w; f.rows = h; f.step = stride;
        }
        return FT_OK;
    }
    void post_all(const double* timestamps) {
        seq++;
        for (auto& g : groups) g->post(timestamps + g->first, frames.data() + g->first);
    }
    // wait for every frame set up to `upto` and add up the counts of those not yet handed out
    int collect(long upto, long long& tracked, long long& attempted) {
        int rc = FT_OK;
        for (; collected < upto; collected++) {
            for (auto& g : groups) {
                ftk_group::Mail& M = g->wait(collected + 1);
                tracked += M.tracked;
                attempted += M.attempted;
                if (M.rc != FT_OK) rc = M.rc;
            }
        }
        return rc;
    }
    // trackers and databases may only be touched by the caller while their group is idle
    void quiesce() {
        long long a = 0, b = 0;
        collect(seq, a, b);
        for (auto& g : groups) g->wait_db();
    }
};

extern "C" {

void ftk_default_options(ftk_options* o) {
    if (!o) return;
    const FeatureTrackerOptions d;
    o->num_pts = d.num_pts; o->fast_threshold = d.fast_threshold; o->grid_x = d.grid_x; o->grid_y = d.grid_y;
    o->min_px_dist = d.min_px_dist; o->pyr_levels = d.pyr_levels; o->win_size = d.win_size; o->lk_max_iter = d.lk_max_iter;
    o->lk_eps = d.lk_eps; o->lk_min_eig = d.lk_min_eig; o->use_gyro_seed = d.use_gyro_seed; o->forward_backward = d.forward_backward;
    o->fb_thresh = d.fb_thresh; o->speculative_detect = d.speculative_detect;
    o->reject_outliers = d.reject_outliers; o->ransac_hypotheses = d.ransac_hypotheses;
    o->stereo_extrinsic_seed = d.stereo_extrinsic_seed; o->stereo_threshold = d.stereo_threshold;
    o->drop_fraction = d.drop_fraction;
    o->use_klt = d.use_klt; o->knn_ratio = d.knn_ratio; o->descriptor_ransac_on_pairs = 0;
}

const char* ftk_backend_name(void) { return ftk_backend_label(); }

int ftk_tracker_create(ftk_tracker** out, int device, const ftk_options* o, int max_width, int max_height) {
    if (!out || !valid_options(o) || max_width < 1 || max_height < 1) return FT_E_INVALID;
    *out = nullptr;
    try {
        std::unique_ptr<ftk_tracker> t(new ftk_tracker());
        void* dctx = nullptr;
        t->be.reset(ftk_make_backend(device, limits_for(*o, 2, max_width, max_height), &dctx));
        if (!t->be) return FT_E_NODEVICE;
        const FeatureTrackerOptions fo = to_options(*o);
        if (fo.use_klt) {
            t->klt = new TrackKLT(t->be.get(), fo, 0, 0);
            t->trk.reset(t->klt);
        } else {
            TrackDescriptor* d = new TrackDescriptor(t->be.get(), fo.num_pts, 0, fo.fast_threshold, fo.grid_x, fo.grid_y, fo.knn_ratio, 0);
            d->set_ransac_on_matched_pairs(o->descriptor_ransac_on_pairs != 0);
            t->trk.reset(d);
        }
        *out = t.release();
        return FT_OK;
    } catch (const std::bad_alloc&) {
        return FT_E_NOMEM;
    } catch (...) {
        return FT_E_HIP;
    }
}

void ftk_tracker_destroy(ftk_tracker* t) { delete t; }

int ftk_tracker_set_calibration(ftk_tracker* t, int n_cams, const double* calib8, const double* T_cam_imu) {
    if (!t || n_cams < 1 || !calib8) return FT_E_INVALID;
    std::map<size_t, Calib8> K;
    std::map<size_t, Mat4d> E;
    calib_maps(n_cams, calib8, T_cam_imu, K, E);
    t->trk->set_calibration(K, E);
    return FT_OK;
}

int ftk_tracker_feed_imu(ftk_tracker* t, double timestamp, const double* am3, const double* wm3) {
    if (!t || !am3 || !wm3) return FT_E_INVALID;
    t->trk->feed_imu(timestamp, Vec3d{{am3[0], am3[1], am3[2]}}, Vec3d{{wm3[0], wm3[1], wm3[2]}});
    return FT_OK;
}

int ftk_tracker_feed_stereo(ftk_tracker* t, double timestamp, const uint8_t* left, const uint8_t* right, int w, int h, size_t stride,
                            int cam_left, int cam_right) {
    if (!t || !left || !right || w < 1 || h < 1 || cam_left < 0 || cam_right < 0 || cam_left > 1 || cam_right > 1) return FT_E_INVALID;
    return t->trk->feed_stereo(timestamp, Image8(left, w, h, stride), Image8(right, w, h, stride), (size_t)cam_left, (size_t)cam_right);
}

int ftk_tracker_feed_monocular(ftk_tracker* t, double timestamp, const uint8_t* img, int w, int h, size_t stride, int cam) {
    if (!t || !t->klt || !img || w < 1 || h < 1 || cam < 0 || cam > 1) return FT_E_INVALID;   // the descriptor tracker is stereo only
    return t->klt->feed_monocular(timestamp, Image8(img, w, h, stride), (size_t)cam);
}

int ftk_tracker_get_points(ftk_tracker* t, int cam, float* xy, uint64_t* ids, int cap) {
    if (!t) return FT_E_INVALID;
    return get_points(*t->trk, cam, xy, ids, cap);
}

int ftk_tracker_database_size(ftk_tracker* t) { return t ? (int)t->trk->get_feature_database()->size() : FT_E_INVALID; }

static int get_track(TrackBase& trk, uint64_t id, int cam, float* uvn4, double* timestamps, int cap);

int ftk_tracker_get_track(ftk_tracker* t, uint64_t id, int cam, float* uvn4, double* timestamps, int cap) {
    if (!t) return FT_E_INVALID;
    return get_track(*t->trk, id, cam, uvn4, timestamps, cap);
}

static int get_track(TrackBase& trk, uint64_t id, int cam, float* uvn4, double* timestamps, int cap) {
    Feature* f = trk.get_feature_database()->get_feature((size_t)id);
    if (!f || !f->timestamps.count((size_t)cam)) return 0;
    const auto& ts = f->timestamps.at((size_t)cam);
    for (size_t i = 0; i < ts.size() && (int)i < cap; i++) {
        if (uvn4) {
            uvn4[4 * i] = f->uvs.at((size_t)cam)[i][0]; uvn4[4 * i + 1] = f->uvs.at((size_t)cam)[i][1];
            uvn4[4 * i + 2] = f->uvs_norm.at((size_t)cam)[i][0]; uvn4[4 * i + 3] = f->uvs_norm.at((size_t)cam)[i][1];
        }
        if (timestamps) timestamps[i] = ts[i];
    }
    return (int)ts.size();
}

int ftk_batch_create_grouped(ftk_batch** out, int device, const ftk_options* o, int n_streams, int max_width, int max_height,
                             int n_groups, int host_threads) {
    if (!out || !valid_options(o) || n_streams < 1 || n_streams > FT_MAX_BATCH || max_width < 1 || max_height < 1) return FT_E_INVALID;
    *out = nullptr;
    // groups overlap one group's host bookkeeping with the others' device work.  Measured on 64 streams: 3 groups
    // 63.5 M features/s, 2 groups 61.2 M, 4 groups 44.7 M (more submitting threads contend inside the HIP runtime, and
    // batches below about 20 streams run the kernels less efficiently).  32 streams: 3 groups 48.2-49.0 M, 2 groups
    // 47.0-47.6 M, 1 group 35 M; 16 streams: 2 groups 30.2 M, 3 groups 28.4-30.8 M, 1 group 24.5 M; 8 streams: 16-17 M
    // whatever the number of groups (a step is then one chain of dependent launches and host work, about 70 us).
    // A fourth group pays (+ 2.5 %) when every group's queue gets a hardware queue of its own: a process has
    // GPU_MAX_HW_QUEUES of them (4 unless the environment says otherwise), shared by all its streams, and with four
    // hardware queues a fourth group shares one with another group and the rate collapses (37 M against 57 M).
    if (n_groups <= 0) {
        const char* hq = getenv("GPU_MAX_HW_QUEUES");
        const int hw_queues = hq ? atoi(hq) : 4;
        // (with 8 hardware queues: 32 streams 50.2 M in four groups, 47-48 M in three; 16 streams 30.5 M in three, 29.8 M in two)
        n_groups = n_streams >= 32 ? (hw_queues >= 8 ? 4 : 3) : (n_streams >= 16 ? (hw_queues >= 8 ? 3 : 2) : (n_streams >= 8 ? 2 : 1));
    }
    n_groups = std::min(n_groups, n_streams);
    // host threads: one driver per group plus helpers for the per-stream bookkeeping inside each group
    if (host_threads <= 0) {
        const int hw = (int)std::thread::hardware_concurrency();
        host_threads = std::max(1, std::min(hw > 0 ? hw : 1, 2 * n_groups));
    }
    // per group: a driver and a database writer; further helpers only when the caller asks for more threads than that
    const int per_group = std::max(1, std::min(host_threads / n_groups - 1, std::max(1, (n_streams / n_groups) / 4)));
    try {
        std::unique_ptr<ftk_batch> b(new ftk_batch());
        b->n_streams = n_streams;
        b->frames.resize(n_streams);
        int first = 0;
        for (int g = 0; g < n_groups; g++) {
            b->groups.emplace_back(new ftk_group());
            ftk_group& G = *b->groups.back();
            G.first = first;
            G.n = n_streams / n_groups + (g < n_streams % n_groups ? 1 : 0);
            first += G.n;
            // three pyramid slots per stream: previous, current, and the next frame staged while the current is tracked
            G.be.reset(ftk_make_backend(device, limits_for(*o, G.n, max_width, max_height, 3), &G.device_ctx));
            if (!G.be) return FT_E_NODEVICE;
            G.batch.reset(new TrackKLTBatch(G.be.get(), to_options(*o), G.n, per_group, 3));
            G.choose_writers(o->num_pts);
        }
        for (auto& g : b->groups) g->start();
        *out = b.release();
        return FT_OK;
    } catch (const std::bad_alloc&) {
        return FT_E_NOMEM;
    } catch (...) {
        return FT_E_HIP;
    }
}

int ftk_batch_create(ftk_batch** out, int device, const ftk_options* o, int n_streams, int max_width, int max_height) {
    return ftk_batch_create_grouped(out, device, o, n_streams, max_width, max_height, 0, 0);
}

void ftk_batch_destroy(ftk_batch* b) { delete b; }

int ftk_batch_set_calibration(ftk_batch* b, int stream, const double* calib8, const double* T_cam_imu) {
    int local = 0;
    ftk_group* g = b ? b->group_of(stream, &local) : nullptr;
    if (!g || !calib8) return FT_E_INVALID;
    b->quiesce();
    std::map<size_t, Calib8> K;
    std::map<size_t, Mat4d> E;
    calib_maps(1, calib8, T_cam_imu, K, E);
    g->batch->tracker(local).set_calibration(K, E);
    return FT_OK;
}

int ftk_batch_feed_imu(ftk_batch* b, int stream, double timestamp, const double* am3, const double* wm3) {
    int local = 0;
    ftk_group* g = b ? b->group_of(stream, &local) : nullptr;
    if (!g || !am3 || !wm3) return FT_E_INVALID;
    // the propagator has its own lock (src/propagator/TrackPropagator.h:201): IMU samples may arrive while frames are tracked
    g->batch->tracker(local).feed_imu(timestamp, Vec3d{{am3[0], am3[1], am3[2]}}, Vec3d{{wm3[0], wm3[1], wm3[2]}});
    return FT_OK;
}

int ftk_batch_feed(ftk_batch* b, const double* timestamps, const void* const* frames, int frames_on_device, int w, int h,
                   size_t stride, long long* tracked, long long* attempted) {
    if (!b || !timestamps || !frames || w < 1 || h < 1) return FT_E_INVALID;
    long long tr = 0, at = 0;
    int rc = b->collect(b->seq, tr, at);       // a frame set left in flight by the pipelined variant belongs to an earlier call
    if (rc != FT_OK) return rc;
    if ((rc = b->fill_frames(frames, frames_on_device, w, h, stride)) != FT_OK) return rc;
    tr = at = 0;
    for (auto& g : b->groups) g->spin_us.store(1000, std::memory_order_relaxed);   // latency matters: poll, do not sleep
    for (auto& g : b->groups) g->front_wanted.store(0, std::memory_order_relaxed);
    b->post_all(timestamps);
    rc = b->collect(b->seq, tr, at);
    for (auto& g : b->groups) g->wait_db();    // the synchronous feed returns with the databases written
    if (tracked) *tracked = tr;
    if (attempted) *attempted = at;
    return rc;
}

int ftk_batch_feed_pipelined(ftk_batch* b, const double* timestamps, const void* const* frames, int frames_on_device, int w, int h,
                             size_t stride, long long* tracked, long long* attempted) {
    if (!b || !timestamps || !frames || w < 1 || h < 1) return FT_E_INVALID;
    long long tr = 0, at = 0;
    const long depth = ftk_group::pipe_depth();
    for (auto& g : b->groups) g->spin_us.store(20, std::memory_order_relaxed);     // throughput matters: leave the cores to the workers
    // one large stream alone on the device: its next frame's pyramids and detector run beside the tracking (1280x720 with
    // 2 000 features: 16.6 -> 19.1 M features/s; 3840x2160 with 10 000: no change, the host's part of the frame covers
    // them either way; small streams and batches of two groups lose 1-3 %)
    const int front = (b->groups.size() == 1 && b->n_streams == 1 && b->groups[0]->batch->tracker(0).get_num_features() >= 1024) ? 1 : 0;
    for (auto& g : b->groups) g->front_wanted.store(front, std::memory_order_relaxed);
    int rc = b->collect(b->seq - depth, tr, at);   // keeps the mailbox ring from wrapping onto a set still in use
    if (rc != FT_OK) return rc;
    if ((rc = b->fill_frames(frames, frames_on_device, w, h, stride)) != FT_OK) return rc;
    b->post_all(timestamps);
    rc = b->collect(b->seq - depth, tr, at);       // the newer sets keep the groups busy meanwhile
    if (tracked) *tracked = tr;
    if (attempted) *attempted = at;
    return rc;
}

int ftk_batch_flush(ftk_batch* b, long long* tracked, long long* attempted) {
    if (!b) return FT_E_INVALID;
    long long tr = 0, at = 0;
    // The caller waits for the tail of the pipeline and nothing is left to overlap it with: every waiting side polls
    // from here on (the sleeps and wake-ups of the throughput setting would add up to a few hundred microseconds over
    // the last two sets).  The next pipelined feed restores the short spin.
    for (auto& g : b->groups) g->spin_us.store(1000, std::memory_order_relaxed);
    const int rc = b->collect(b->seq, tr, at);
    for (auto& g : b->groups) g->wait_db();   // every observation is in its database when flush returns
    if (tracked) *tracked = tr;
    if (attempted) *attempted = at;
    return rc;
}

int ftk_batch_get_points(ftk_batch* b, int stream, float* xy, uint64_t* ids, int cap) {
    int local = 0;
    ftk_group* g = b ? b->group_of(stream, &local) : nullptr;
    if (!g) return FT_E_INVALID;
    b->quiesce();
    return get_points(g->batch->tracker(local), 0, xy, ids, cap);
}

int ftk_batch_stage_times(ftk_batch* b, double* out8, long long* feeds, int reset) {
    if (!b || !out8) return FT_E_INVALID;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    long long n = 0;
    b->quiesce();
    for (auto& g : b->groups) {
        double t[TrackKLTBatch::ST_COUNT];
        long long f = 0;
        g->batch->stage_times(t, &f, reset != 0);
        for (int i = 0; i < 8; i++) out8[i] += t[i];
        n = std::max(n, f);
    }
    if (feeds) *feeds = n;
    return FT_OK;
}

int ftk_batch_database_size(ftk_batch* b, int stream) {
    int local = 0;
    ftk_group* g = b ? b->group_of(stream, &local) : nullptr;
    if (!g) return FT_E_INVALID;
    b->quiesce();
    return (int)g->batch->tracker(local).get_feature_database()->size();
}

int ftk_batch_get_track(ftk_batch* b, int stream, uint64_t id, float* uvn4, double* timestamps, int cap) {
    int local = 0;
    ftk_group* g = b ? b->group_of(stream, &local) : nullptr;
    if (!g) return FT_E_INVALID;
    b->quiesce();
    return get_track(g->batch->tracker(local), id, 0, uvn4, timestamps, cap);
}

int ftk_batch_groups(ftk_batch* b) { return b ? (int)b->groups.size() : FT_E_INVALID; }

void* ftk_batch_group_ctx(ftk_batch* b, int group) {
    if (!b || group < 0 || group >= (int)b->groups.size()) return nullptr;
    return b->groups[group]->device_ctx;
}

void* ftk_batch_device_ctx(ftk_batch* b) { return ftk_batch_group_ctx(b, 0); }

}  // extern "C"
