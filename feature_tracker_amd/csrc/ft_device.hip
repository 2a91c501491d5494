// C ABI of the gfx950 visual front-end device layer (declared in include/ft_device.h).
//
// Owns the pyramid arena in HBM, the pinned staging buffers and one HIP queue per context; launches the
// hand-written kernels in ft_kernels_*.hpp.  No torch types, no exceptions across the boundary.
#include "ft_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ft_common.hpp"
#include "ft_kernels_brief.hpp"
#include "ft_kernels_fast.hpp"
#include "ft_kernels_front.hpp"
#include "ft_kernels_lk.hpp"
#include "ft_kernels_match.hpp"
#include "ft_kernels_outlier.hpp"
#include "ft_kernels_pyramid.hpp"
#include "ft_kernels_stereo.hpp"

using namespace ftd;

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int FT_RO_DEFAULT_HYP = 1000;   // OpenCV's iteration limit for findFundamentalMat(FM_RANSAC)
constexpr int FT_RO_ERROR_WORD = 16;   // word of the pinned completion-flag block that the rejection kernel sets when a barrier gave up

struct SlotState {
    int w = 0, h = 0;
    int levels = 0;        // images valid (0 = no frame)
    int grad_levels = 0;   // gradient images valid
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;   // 0 pyramid, 1 lk, 2 fast, 3 outlier rejection
};

struct PendingLK {
    bool active = false;
    uint64_t seq = 0;          // position in the queue (see ft_ctx::seq)
    std::vector<ft_lk_job> jobs;
    int total = 0;
    int nlev = 0;
    bool want_iters = false;
    bool count_work = false;   // profiling: fold the iteration counters into the context's totals
};

struct PendingFast {
    bool active = false;
    uint64_t seq = 0;
    int n = 0, cells = 0, per_cell = 0, cap = 0;
    ft_keypoint* out = nullptr;
    int* n_out = nullptr;
    int* cell_counts = nullptr;
};

}  // namespace

struct ft_ctx {
    int device = 0;
    ft_limits lim{};
    // One in-order queue for everything.  A second queue for the front work (ingest, pyramids, detector) next to the
    // tracking batches, ordered per pyramid slot with events, was tried twice.  With a higher priority for the
    // tracking queue, a high-priority queue that waits for an event of a normal-priority queue can starve that queue
    // and hang the device.  With equal priorities the kernels only slow each other down (they are bound by vector
    // issue, not by latency) and the event traffic costs host time: 0.160 -> 0.178 ms per step with two stream groups.
    // Small batches, whose kernels run at their latency floors, gain nothing either from 8 streams on (8 streams in
    // one group 74.6 -> 74.2 us per step, 16 streams 92.8 -> 101 us; 4 streams 68.5 -> 58.3 us): a step is then the
    // chain host preparation -> tracking -> outlier rejection -> host bookkeeping, and the next set's front work
    // already runs on the device while the host does its part of that chain.
    hipStream_t q_ = nullptr;            // use enq(ctx) to put work on it
    // Optional second queue for the work that does not depend on tracking results (frame ingest, pyramids, detector):
    // see ft_set_front_queue.  cur_ is the queue enq() hands out (q_ outside a FrontScope); front_dirty_ says that the
    // front queue holds work the main queue has not been told to wait for yet.
    hipStream_t front_q_ = nullptr;
    hipStream_t cur_ = nullptr;
    hipEvent_t front_ev_ = nullptr;
    bool front_enabled = false, front_dirty_ = false;
    int front_env = -1;                  // FT_FRONT_QUEUE: 0 never, 1 from creation on, unset: as ft_set_front_queue says
    // Completion flag of the outlier-rejection kernel: its last workgroup stores a sequence number to pinned host
    // memory, and a fence taken while that launch is the newest work on the queue is waited for by polling the flag
    // (about 5 us sooner than through an event).
    unsigned* done_count_dev = nullptr;  // device: workgroups of the launch that have finished
    unsigned* done_flag_host = nullptr;  // pinned
    unsigned* done_flag_dev = nullptr;   // its device address
    int* ro_gcnt_dev = nullptr;          // outlier rejection, pairs split over workgroups: inlier totals [pair][RO_MAX_HYP]
    unsigned* ro_gbar_dev = nullptr;     // ... and per pair {barrier arrivals, finished parts}; both zero between launches
    bool ro_split = true;                // FT_RO_SPLIT=0: one workgroup per pair whatever its size
    unsigned ro_spin_limit = RO_SPIN_LIMIT;   // FT_RO_SPIN_LIMIT: barrier polls before a part of a split pair gives up (tests set it to 1)
    bool use_done_flag = true;           // FT_DONE_FLAG=0: every fence records an event; the kernel's completion flag is not used
    uint32_t done_seq = 0;               // last sequence number handed to a launch
    uint32_t tail_flag_seq = 0;          // != 0: the newest work on the queue is the launch that stores this number
    ArenaDesc arena{};
    int lvl_w[FT_MAX_LEVELS]{}, lvl_h[FT_MAX_LEVELS]{};   // geometry of the largest frame
    std::vector<SlotState> slots;

    // frame upload staging: one pinned buffer per slot + completion event
    uint8_t* stage = nullptr;
    size_t stage_bytes_per_slot = 0;
    std::vector<hipEvent_t> stage_done;
    std::vector<char> stage_busy;
    uint8_t* stage_dev = nullptr;               // device-side address of `stage` (kernels read staged frames over the bus)
    // Staging in device memory (large BAR only; same layout as `stage`): ft_stage_frame writes a frame there itself,
    // through the bus window, and the pass that builds level 0 reads device memory instead of host memory over the bus.
    // For contexts of up to four streams and frames of up to 1 MiB, where the latency of a frame counts (single-stream
    // p50 79.2 -> 73.8 us); many frames per call are faster through pinned memory and one copy command (64 x 752x480:
    // 20.3 M features/s against 17.0 M with every frame written through the window by the caller's threads).
    // stage_in_bar[slot]: that slot's last staged frame lies there.  FT_STAGE_BAR=0: pinned staging only.
    uint8_t* stage_bar = nullptr;
    std::vector<char> stage_in_bar;
    uint8_t* ingest_dev = nullptr;              // device memory: landing area of a copy command that brings many staged frames at once
    static constexpr int STAGE_EV_RING = 4;
    hipEvent_t stage_batch_ev[STAGE_EV_RING] = {};   // recorded after a kernel that reads staged frames
    int stage_batch_next = 0;
    std::vector<int> stage_batch_idx;           // per slot: ring entry of the kernel still reading its staging, -1 = none
                                                // (a reused entry holds a later event of the same queue: still correct)

    // LK staging
    uint8_t* lk_in_host = nullptr;  uint8_t* lk_in_dev = nullptr;  size_t lk_in_bytes = 0;
    bool lk_in_bar = false;         // lk_in_dev is written by the host through the bus window (see ft_ctx_create) ...
    int lk_bar_max = 4096;          // ... for batches of up to this many points (FT_LK_BAR_MAX)
    uint8_t* lk_in_host_dev = nullptr;    // device-side address of lk_in_host
    uint8_t* lk_out_host = nullptr; uint8_t* lk_out_dev = nullptr; size_t lk_out_bytes = 0;
    uint8_t* lk_out_host_dev = nullptr;   // device-side address of lk_out_host
    float4* lk_norm_dev = nullptr;        // outlier rejection: normalized point pairs
    // descriptor matching, allocated on first use: both descriptor sets and both neighbour tables, device + pinned host
    uint8_t* match_dev = nullptr;  uint8_t* match_host = nullptr;  size_t match_cap = 0;   // descriptors per set
    PendingLK lk_pending;

    // FAST scratch
    uint32_t* cand = nullptr; size_t cand_entries_per_img = 0;
    int* cand_count = nullptr;          // [n_streams][max_grid_cells], FAST_COUNT_STRIDE ints apart
    uint8_t* fast_out_host_dev[2] = {nullptr, nullptr};   // device-side addresses of the pinned result buffers
    // detector outputs are double-buffered on the host side so that the detector of the next frame set can be
    // enqueued before the previous set's outputs have been delivered
    static constexpr int FAST_RING = 2;
    uint8_t* fast_out_host[FAST_RING] = {nullptr, nullptr}; size_t fast_out_bytes = 0;
    PendingFast fast_pending[FAST_RING];
    int fast_next = 0;

    // every *_async call takes the next sequence number; a fence remembers the number reached when it was recorded
    uint64_t seq = 0;
    static constexpr int FENCE_RING = 16;
    struct Fence { hipEvent_t ev = nullptr; uint64_t id = 0, seq = 0; uint32_t flag_seq = 0; bool recorded = false; };
    Fence fences[FENCE_RING];
    uint64_t fence_ids = 0;

    // small device scratch for ft_predict_flow
    float* pf_dev = nullptr; float* pf_host = nullptr; size_t pf_points = 0;

    // timing
    int profile = 0;   // bit 0: time kernel groups with events, bit 1: count LK work (iteration counters)
    // pyrDown steps of a frame set that are still to be launched (levels done_to .. top): ft_fast_grid_async on the same
    // images right after the pyramid call launches them together with the detector (ft_kernels_front.hpp); anything else
    // that is put on the queue, or waits for it, launches them first (enq, fences, syncs).  FT_FRONT_FUSE=0: never deferred.
    struct DeferredLevels {
        bool active = false, busy = false;
        ImgList L;
        int w = 0, h = 0, top = 0, done_to = 0;
    } deferred;
    bool front_fuse = true;
    int lk_direct_max = 1024; // points per tracking launch up to which the input block is read in place from pinned memory (FT_LK_DIRECT_MAX)
    bool lk_fixed_windows = true;   // kernels compiled for 21 x 21 and 15 x 15 windows (FT_LK_FIXED_WINDOWS=0: the general ones)
    bool lk_fused_fb = true;  // forward-backward tracking in one launch (FT_LK_FUSED_FB=0: two launches)
    bool lk_hold_single = true;   // one-stream contexts: 21 x 21 kernels that hold their pixel pairs (FT_LK_HOLD_SINGLE=0: the usual ones)
    int lk_pair_max = 2048;   // points per tracking launch up to which lk_pair_kernel runs (FT_LK_PAIR_MAX overrides it: 0 = never)
    std::vector<EventPair> ev_inflight;
    std::vector<EventPair> ev_pool;
    ft_kernel_time times{};
    hipEvent_t region_a = nullptr, region_b = nullptr;

    std::string hip_err;
    std::mutex err_mutex;   // ft_stage_frame may fail on several threads at once
};

namespace {

// the context's queue, for putting work on it: whatever is enqueued next is then the newest work on the queue
void flush_deferred(ft_ctx* c);
inline hipStream_t enq(ft_ctx* c) {
    if (c->deferred.active && !c->deferred.busy) flush_deferred(c);
    c->tail_flag_seq = 0;
    if (c->cur_ != c->q_) return c->cur_;
    if (c->front_dirty_) {   // everything put on the main queue from here on comes after the front queue's work so far
        (void)hipStreamWaitEvent(c->q_, c->front_ev_, 0);
        c->front_dirty_ = false;
    }
    return c->q_;
}

// Inside one of these, enq() hands out the front queue (if the context has one switched on).  The caller's contract:
// what the enclosed work writes (a pyramid ring slot, the detector's buffers) is not read by tracking work that is
// still queued.  The other direction is kept here: the main queue waits for the front queue's latest work before
// anything else is put on it.
struct FrontScope {
    ft_ctx* c;
    bool on;
    explicit FrontScope(ft_ctx* ctx) : c(ctx), on(ctx->front_enabled && ctx->front_q_ && ctx->cur_ == ctx->q_) {
        if (on) c->cur_ = c->front_q_;
    }
    ~FrontScope() {
        if (!on) return;
        (void)hipEventRecord(c->front_ev_, c->front_q_);
        c->front_dirty_ = true;
        c->cur_ = c->q_;
        c->tail_flag_seq = 0;
    }
    FrontScope(const FrontScope&) = delete;
    FrontScope& operator=(const FrontScope&) = delete;
};

#define FT_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            std::lock_guard<std::mutex> lk__((ctx)->err_mutex);                                    \
            (ctx)->hip_err = std::string(#call) + ": " + hipGetErrorString(e__);                   \
            return FT_E_HIP;                                                                       \
        }                                                                                          \
    } while (0)

int slot_index(const ft_ctx* c, int stream, int slot) {
    if (stream < 0 || stream >= c->lim.n_streams || slot < 0 || slot >= c->lim.n_slots) return -1;
    return stream * c->lim.n_slots + slot;
}

void prof_begin(ft_ctx* c, int kind, EventPair& ep) {
    if (!(c->profile & 1)) return;
    if (!c->ev_pool.empty()) { ep = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else { (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b); }
    ep.kind = kind;
    (void)hipEventRecord(ep.a, enq(c));
}
void prof_end(ft_ctx* c, EventPair& ep) {
    if (!(c->profile & 1) || !ep.a) return;
    (void)hipEventRecord(ep.b, enq(c));
    c->ev_inflight.push_back(ep);
}
void prof_collect(ft_ctx* c) {
    size_t kept = 0;
    for (auto& ep : c->ev_inflight) {
        if (hipEventQuery(ep.b) == hipErrorNotReady) { c->ev_inflight[kept++] = ep; continue; }   // still queued
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            if (ep.kind == 0) { c->times.pyramid_ms += ms; c->times.pyramid_launches++; }
            else if (ep.kind == 1) { c->times.lk_ms += ms; c->times.lk_launches++; }
            else if (ep.kind == 2) { c->times.fast_ms += ms; c->times.fast_launches++; }
            else { c->times.reject_ms += ms; c->times.reject_launches++; }
        }
        c->ev_pool.push_back(ep);
    }
    c->ev_inflight.resize(kept);
}

// number of the top level buildOpticalFlowPyramid would produce (early stop when the next level <= window)
int achieved_level(int w, int h, int win_w, int win_h, int max_level) {
    int level = 0;
    for (; level <= max_level; level++) {
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) return level;
    }
    return max_level;
}

void deliver_fast(ft_ctx* c, PendingFast& F, const uint8_t* host) {
    const size_t per_img = (size_t)F.cells * F.per_cell;
    const ft_keypoint* kp = reinterpret_cast<const ft_keypoint*>(host);
    const int* cnt = reinterpret_cast<const int*>(host + align_up(sizeof(ft_keypoint) * per_img * F.n, 16));
    const int* ccnt = cnt + (size_t)F.n * F.cells;
    for (int i = 0; i < F.n; i++) {
        int m = 0;
        for (int cell = 0; cell < F.cells; cell++) {
            const int k = cnt[(size_t)i * F.cells + cell];
            const ft_keypoint* src = kp + (size_t)i * per_img + (size_t)cell * F.per_cell;
            for (int r = 0; r < k && m < F.cap; r++) F.out[(size_t)i * F.cap + m++] = src[r];
        }
        if (F.n_out) F.n_out[i] = m;
        if (F.cell_counts) memcpy(F.cell_counts + (size_t)i * F.cells, ccnt + (size_t)i * F.cells, sizeof(int) * F.cells);
    }
    F.active = false;
}

// outputs of every *_async call with sequence number <= upto; the queue must have passed them
int deliver_pending(ft_ctx* c, uint64_t upto) {
    // detector results in enqueue order (the older ring entry first)
    for (int k = 0; k < ft_ctx::FAST_RING; k++) {
        const int i = (c->fast_next + k) % ft_ctx::FAST_RING;
        PendingFast& F = c->fast_pending[i];
        if (F.active && F.seq <= upto) deliver_fast(c, F, c->fast_out_host[i]);
    }
    if (c->lk_pending.active && c->lk_pending.seq <= upto) {
        PendingLK& P = c->lk_pending;
        if (c->done_flag_host[FT_RO_ERROR_WORD]) {
            // a part of a split pair gave up waiting for the others: the batch's masks are void.  The kernel leaves
            // its counters clean when all parts finish; they are cleared here all the same.
            c->done_flag_host[FT_RO_ERROR_WORD] = 0;
            (void)hipStreamSynchronize(c->q_);
            (void)hipMemset(c->ro_gcnt_dev, 0, sizeof(int) * FT_MAX_BATCH * RO_MAX_HYP);
            (void)hipMemset(c->ro_gbar_dev, 0, sizeof(unsigned) * 2 * FT_MAX_BATCH);
            P.active = false;
            std::lock_guard<std::mutex> lk(c->err_mutex);
            c->hip_err = "ransac_kernel: the workgroups of a split pair did not meet at their barrier";
            return FT_E_HIP;
        }
        const size_t N = (size_t)P.total;
        const float* xy = reinterpret_cast<const float*>(c->lk_out_host);
        const float* er = reinterpret_cast<const float*>(c->lk_out_host + 8 * N);
        const uint8_t* st = c->lk_out_host + 12 * N;
        const int* it = reinterpret_cast<const int*>(c->lk_out_host + align_up(13 * N, 16));
        size_t off = 0;
        for (const ft_lk_job& j : P.jobs) {
            const size_t n = (size_t)j.n;
            if (n) {
                if (j.next_xy) memcpy(j.next_xy, xy + 2 * off, 8 * n);   // absent for a rejection-only batch
                if (j.err) memcpy(j.err, er + off, 4 * n);
                memcpy(j.status, st + off, n);
                if (j.iters && P.want_iters) memcpy(j.iters, it + off * P.nlev, sizeof(int) * n * P.nlev);
            }
            off += n;
        }
        if (P.count_work) {
            c->times.lk_points += (long long)N;
            for (size_t i = 0; i < N * (size_t)P.nlev; i++)
                if (it[i] >= 0) { c->times.lk_level_visits++; c->times.lk_iterations += it[i]; }
        }
        P.active = false;
    }
    return FT_OK;
}

int sync_and_deliver(ft_ctx* c) {
    if (c->deferred.active) flush_deferred(c);
    if (c->front_q_) FT_HIP(c, hipStreamSynchronize(c->front_q_));   // (also when called from inside a FrontScope)
    FT_HIP(c, hipStreamSynchronize(c->q_));
    c->tail_flag_seq = 0;
    std::fill(c->stage_busy.begin(), c->stage_busy.end(), 0);
    std::fill(c->stage_batch_idx.begin(), c->stage_batch_idx.end(), -1);
    prof_collect(c);
    return deliver_pending(c, ~0ull);
}

// counts[j]: points of pair j.  Pairs of more than RO_CHUNK points are split into parts, one workgroup each, as long as
// the launch then has at most RO_MAX_WGS workgroups: the parts of a pair wait for each other inside the kernel, so all
// of them must be resident at once (256 threads and the hypothesis phase's registers: two workgroups per compute unit;
// RO_MAX_WGS leaves room for four such launches at a time).
void launch_reject(ft_ctx* c, OutlierArgs o, int n_jobs, const int* counts) {
    EventPair ep;
    prof_begin(c, 3, ep);
    const uint32_t seq = ++c->done_seq ? c->done_seq : ++c->done_seq;   // never 0
    o.done_count = c->done_count_dev; o.done_flag = c->use_done_flag ? c->done_flag_dev : nullptr; o.done_seq = seq;
    o.gcnt = c->ro_gcnt_dev; o.gbar = c->ro_gbar_dev; o.error_flag = c->done_flag_dev + FT_RO_ERROR_WORD; o.spin_limit = c->ro_spin_limit;
    int parts[FT_MAX_BATCH], wgs = 0, max_points = 0;
    for (int j = 0; j < n_jobs; j++) {
        parts[j] = c->ro_split ? std::min(std::max((counts[j] + RO_CHUNK - 1) / RO_CHUNK, 1), RO_MAX_PARTS) : 1;
        wgs += parts[j];
        max_points = std::max(max_points, counts[j]);
    }
    const bool split = wgs > n_jobs && wgs <= RO_MAX_WGS;
    wgs = 0;
    for (int j = 0; j < n_jobs; j++) {
        const int g = split ? parts[j] : 1;
        for (int k = 0; k < g; k++) o.wg_map[wgs++] = (uint32_t)j | ((uint32_t)k << 16) | ((uint32_t)g << 24);
    }
    // a split launch of up to 64 parts also takes 512 threads per part: the per-point phases of a part of 512 points are
    // half as long (cfg3 21.1 -> 19.0 us, cfg5 23.5 -> 21.8 us), and that many parts are resident beside a second launch
    // like it (FT_RO_SPLIT_THREADS=256: the narrow form)
    static const int split_threads = getenv("FT_RO_SPLIT_THREADS") ? atoi(getenv("FT_RO_SPLIT_THREADS")) : 512;
    const bool wide = (!split && max_points > 1024) || (split && split_threads == 512 && wgs <= 64);
    if (wide) hipLaunchKernelGGL(ransac_kernel<512>, dim3(wgs), dim3(512), 0, enq(c), o);   // 512: the most that leaves the hypothesis phase its 193 registers
    else hipLaunchKernelGGL(ransac_kernel<256>, dim3(wgs), dim3(256), 0, enq(c), o);
    prof_end(c, ep);
    if (!(c->profile & 1) && c->use_done_flag) c->tail_flag_seq = seq;   // (with timing on, an event record follows the launch)
}

int pick_seglen(int win_w, int win_h) {
    const int opts[3] = {4, 8, 16};
    for (int s : opts) {
        const int segs = (win_w + s - 1) / s;
        if (segs * win_h <= 64) return s;
    }
    return 0;
}

static int collect_images(ft_ctx* c, int n, const int* streams, const int* slots, ImgList& L, int& w, int& h) {
    if (n < 1 || n > FT_MAX_BATCH || !streams || !slots) return FT_E_INVALID;
    L.n = n;
    for (int i = 0; i < n; i++) {
        const int si = slot_index(c, streams[i], slots[i]);
        if (si < 0) return FT_E_INVALID;
        const SlotState& S = c->slots[si];
        if (S.levels < 1) return FT_E_STATE;
        if (i == 0) { w = S.w; h = S.h; }
        else if (S.w != w || S.h != h) return FT_E_INVALID;
        L.idx[i] = si;
    }
    return FT_OK;
}

// levels from_level..top of the images in L: pyrDown (level l -> l+1 for l < top) and/or gradients of every level
static int run_levels(ft_ctx* c, const ImgList& L, int w, int h, int top, bool down, bool grad, int from_level = 0, bool timed = true) {
    EventPair ep;
    if (timed) prof_begin(c, 0, ep);
    int lw = w, lh = h;
    const FramePtrs none{};
    for (int l = 0; l <= top; l++) {
        const int do_down = (down && l < top) ? 1 : 0;
        const int do_grad = grad ? 1 : 0;
        if (l >= from_level && (do_down || do_grad)) {
            const int tiles = ((lw + LV_TX - 1) / LV_TX) * ((lh + LV_TY - 1) / LV_TY);
            if (do_grad) hipLaunchKernelGGL((level_kernel<false, true>), dim3(xcd_grid(tiles * L.n)), dim3(256), 0, enq(c), c->arena, L, none, 0, l, lw, lh,
                                            (uint64_t)0, do_grad, do_down);
            else hipLaunchKernelGGL((level_kernel<false, false>), dim3(xcd_grid(tiles * L.n)), dim3(256), 0, enq(c), c->arena, L, none, 0, l, lw, lh,
                                    (uint64_t)0, 0, do_down);
        }
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    if (timed) prof_end(c, ep);
    FT_HIP(c, hipGetLastError());
    return FT_OK;
}

void flush_deferred(ft_ctx* c) {
    ft_ctx::DeferredLevels& D = c->deferred;
    if (!D.active || D.busy) return;
    D.busy = true;
    D.active = false;
    (void)run_levels(c, D.L, D.w, D.h, D.top, true, false, D.done_to, false);
    D.busy = false;
}

// copy a frame into the pinned staging buffer of a slot, tightly packed, once nothing on the device reads that buffer any more
static int stage_rows(ft_ctx* c, int si, const uint8_t* host, int w, int h, size_t stride, bool to_bar = false) {
    if (c->stage_busy[si]) {   // a copy command from this staging buffer may still be in flight
        FT_HIP(c, hipEventSynchronize(c->stage_done[si]));
        c->stage_busy[si] = 0;
    }
    if (c->stage_batch_idx[si] >= 0) {   // ... or a kernel that reads it
        FT_HIP(c, hipEventSynchronize(c->stage_batch_ev[c->stage_batch_idx[si]]));
        c->stage_batch_idx[si] = -1;
    }
    to_bar = to_bar && c->stage_bar != nullptr && (size_t)w * h <= ((size_t)1 << 20);
    uint8_t* st = (to_bar ? c->stage_bar : c->stage) + (size_t)si * c->stage_bytes_per_slot;
    if (stride == (size_t)w) memcpy(st, host, (size_t)w * h);
    else for (int y = 0; y < h; y++) memcpy(st + (size_t)y * w, host + (size_t)y * stride, (size_t)w);
    if (to_bar) __builtin_ia32_sfence();   // out of this thread's write-combining buffers before anybody launches the reader
    c->stage_in_bar[si] = to_bar ? 1 : 0;
    return FT_OK;
}

template <int SEGLEN, bool FUSED, int WIN>
static void launch_lk(ft_ctx* c, const LKArgs& a, int total) {
    // Up to one wavefront pair per wave slot pair of the device the two-wave form is the faster one (its chain of
    // dependent steps is a third shorter); beyond that the idle halves of the pairs cost more than the chain saves.
    // a context of one stream has the device to itself while it tracks: its 21 x 21 launches take the kernels that keep
    // their pixel pairs across iterations (122 registers instead of 112: nothing else wants the difference there)
    const bool hold = WIN == 21 && SEGLEN == 8 && c->lk_hold_single && c->lim.n_streams == 1 && a.win_h <= 21;
    if (total <= c->lk_pair_max) {
        if (a.win_h > 21) hipLaunchKernelGGL((lk_pair_kernel<SEGLEN, true, FUSED, 0>), dim3(xcd_grid(total)), dim3(128), 0, enq(c), a);
        else if (hold) hipLaunchKernelGGL((lk_pair_kernel<SEGLEN, false, FUSED, WIN, WIN == 21 && SEGLEN == 8>), dim3(xcd_grid(total)), dim3(128), 0, enq(c), a);
        else hipLaunchKernelGGL((lk_pair_kernel<SEGLEN, false, FUSED, WIN>), dim3(xcd_grid(total)), dim3(128), 0, enq(c), a);
        return;
    }
    if (a.win_h > 21) hipLaunchKernelGGL((lk_kernel<SEGLEN, true, FUSED, 0>), dim3(xcd_grid(total)), dim3(64), 0, enq(c), a);
    else if (hold) hipLaunchKernelGGL((lk_kernel<SEGLEN, false, FUSED, WIN, WIN == 21 && SEGLEN == 8>), dim3(xcd_grid(total)), dim3(64), 0, enq(c), a);
    else hipLaunchKernelGGL((lk_kernel<SEGLEN, false, FUSED, WIN>), dim3(xcd_grid(total)), dim3(64), 0, enq(c), a);
}
// the windows with kernels of their own: 21 x 21 (8-pixel segments) and the reference's 15 x 15 (4-pixel segments)
template <bool FUSED>
static void launch_lk_seglen(ft_ctx* c, const LKArgs& a, int total, int seglen) {
    const bool fixed = c->lk_fixed_windows && a.win_w == a.win_h;
    switch (seglen) {
        case 4:
            if (fixed && a.win_w == 15) launch_lk<4, FUSED, 15>(c, a, total);
            else launch_lk<4, FUSED, 0>(c, a, total);
            break;
        case 8:
            if (fixed && a.win_w == 21) launch_lk<8, FUSED, 21>(c, a, total);
            else launch_lk<8, FUSED, 0>(c, a, total);
            break;
        default: launch_lk<16, FUSED, 0>(c, a, total); break;
    }
}

}  // namespace

extern "C" {

const char* ft_strerror(int code) {
    switch (code) {
        case FT_OK: return "ok";
        case FT_E_INVALID: return "invalid argument";
        case FT_E_LIMIT: return "exceeds context limits";
        case FT_E_HIP: return "HIP runtime error";
        case FT_E_NOMEM: return "out of memory";
        case FT_E_STATE: return "slot has no frame, pyramid or gradients";
        case FT_E_NODEVICE: return "no usable GPU device";
        default: return "unknown error";
    }
}

const char* ft_last_hip_error(ft_ctx* ctx) { return ctx ? ctx->hip_err.c_str() : ""; }

int ft_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ft_ctx_create(ft_ctx** out, int device, const ft_limits* lim) {
    if (!out || !lim) return FT_E_INVALID;
    *out = nullptr;
    if (lim->max_width < 1 || lim->max_height < 1 || lim->max_levels < 1 || lim->max_levels > FT_MAX_LEVELS ||
        lim->n_streams < 1 || lim->n_streams > FT_MAX_BATCH || lim->n_slots < 1 || lim->max_points < 1 ||
        lim->max_grid_cells < 1 || lim->max_per_cell < 1)
        return FT_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return FT_E_NODEVICE;
    ft_ctx* c = new (std::nothrow) ft_ctx();
    if (!c) return FT_E_NOMEM;
    c->device = device;
    c->lim = *lim;
    if (const char* e = getenv("FT_LK_PAIR_MAX")) c->lk_pair_max = atoi(e);
    if (const char* e = getenv("FT_LK_DIRECT_MAX")) c->lk_direct_max = atoi(e);
    if (const char* e = getenv("FT_LK_HOLD_SINGLE")) c->lk_hold_single = atoi(e) != 0;
    if (const char* e = getenv("FT_FRONT_FUSE")) c->front_fuse = atoi(e) != 0;
    if (const char* e = getenv("FT_LK_FIXED_WINDOWS")) c->lk_fixed_windows = atoi(e) != 0;
    if (const char* e = getenv("FT_LK_FUSED_FB")) c->lk_fused_fb = atoi(e) != 0;
    auto fail = [&](int code) { ft_ctx_destroy(c); return code; };
    if (hipSetDevice(device) != hipSuccess) return fail(FT_E_NODEVICE);
    if (hipStreamCreateWithFlags(&c->q_, hipStreamNonBlocking) != hipSuccess) return fail(FT_E_HIP);
    c->cur_ = c->q_;
    if (const char* e = getenv("FT_FRONT_QUEUE")) c->front_env = atoi(e) != 0;

    // arena layout from the largest frame
    size_t off = 0;
    int w = lim->max_width, h = lim->max_height;
    for (int l = 0; l < lim->max_levels; l++) {
        c->lvl_w[l] = w; c->lvl_h[l] = h;
        LevelDesc& d = c->arena.lv[l];
        d.img_pitch = (uint32_t)align_up((size_t)w, 256);
        d.grad_pitch = (uint32_t)align_up((size_t)w * 4, 256);
        d.img_off = off; off += align_up((size_t)d.img_pitch * h, 256);
        d.grad_off = off; off += align_up((size_t)d.grad_pitch * h, 256);
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    c->arena.slot_bytes = align_up(off, 4096);
    const size_t n_slots_total = (size_t)lim->n_streams * lim->n_slots;
    c->slots.resize(n_slots_total);
    void* p = nullptr;
    if (hipMalloc(&p, c->arena.slot_bytes * n_slots_total) != hipSuccess) return fail(FT_E_NOMEM);
    c->arena.base = (uint8_t*)p;

    c->stage_bytes_per_slot = align_up((size_t)lim->max_width * lim->max_height, 4096);
    if (hipHostMalloc(&p, c->stage_bytes_per_slot * n_slots_total, hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
    c->stage = (uint8_t*)p;
    {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) return fail(FT_E_HIP);
        c->stage_dev = (uint8_t*)dp;
    }
    c->stage_in_bar.assign(n_slots_total, 0);
    {
        int large_bar = 0;
        (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device);
        const char* e = getenv("FT_STAGE_BAR");
        if (large_bar == 1 && lim->n_streams <= 4 && !(e && atoi(e) == 0)) {
            if (hipExtMallocWithFlags(&p, c->stage_bytes_per_slot * n_slots_total, hipDeviceMallocFinegrained) == hipSuccess) c->stage_bar = (uint8_t*)p;
            else (void)hipGetLastError();   // pinned staging only
        }
    }
    if (hipMalloc(&p, (size_t)lim->max_width * lim->max_height * lim->n_streams) != hipSuccess) return fail(FT_E_NOMEM);
    c->ingest_dev = (uint8_t*)p;
    c->stage_done.resize(n_slots_total, nullptr);
    c->stage_busy.assign(n_slots_total, 0);
    c->stage_batch_idx.assign(n_slots_total, -1);
    for (auto& e : c->stage_batch_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(FT_E_HIP);
    for (auto& e : c->stage_done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(FT_E_HIP);

    // LK staging
    const size_t N = (size_t)lim->max_points;
    c->lk_in_bytes = align_up(sizeof(LKJobDev) * FT_MAX_BATCH, 256) + align_up(2 * N, 256) + 2 * align_up(8 * N, 256) + 256;
    c->lk_out_bytes = align_up(13 * N, 16) + sizeof(int) * N * FT_MAX_LEVELS + 256;
    if (hipHostMalloc(&p, c->lk_in_bytes, hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
    c->lk_in_host = (uint8_t*)p;
    {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) return fail(FT_E_HIP);
        c->lk_in_host_dev = (uint8_t*)dp;
    }
    // The device copy of a tracking batch's input block.  Where the host reaches device memory through the bus window
    // (large BAR), the block is fine-grained device memory and the host writes it there itself (write-combined, 40 KB in
    // about 1.3 us): no kernel has to fetch it from pinned memory before the tracking launch.  FT_LK_BAR_INPUT=0: the
    // fetch kernel as before.
    {
        int large_bar = 0;
        (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device);
        const char* e = getenv("FT_LK_BAR_INPUT");
        c->lk_in_bar = large_bar == 1 && !(e && atoi(e) == 0);
        if (const char* m = getenv("FT_LK_BAR_MAX")) c->lk_bar_max = atoi(m);
    }
    if (c->lk_in_bar && hipExtMallocWithFlags(&p, c->lk_in_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        c->lk_in_bar = false;   // no such memory here: the fetch kernel as before
    }
    if (!c->lk_in_bar && hipMalloc(&p, c->lk_in_bytes) != hipSuccess) return fail(FT_E_NOMEM);
    c->lk_in_dev = (uint8_t*)p;
    if (hipHostMalloc(&p, c->lk_out_bytes, hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
    c->lk_out_host = (uint8_t*)p;
    {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) return fail(FT_E_HIP);
        c->lk_out_host_dev = (uint8_t*)dp;
    }
    if (hipMalloc(&p, c->lk_out_bytes) != hipSuccess) return fail(FT_E_NOMEM);
    c->lk_out_dev = (uint8_t*)p;
    if (hipMalloc(&p, sizeof(float4) * N) != hipSuccess) return fail(FT_E_NOMEM);
    c->lk_norm_dev = (float4*)p;
    if (hipMalloc(&p, 256) != hipSuccess) return fail(FT_E_NOMEM);
    c->done_count_dev = (unsigned*)p;
    if (hipMemset(c->done_count_dev, 0, 256) != hipSuccess) return fail(FT_E_HIP);
    if (hipMalloc(&p, sizeof(int) * FT_MAX_BATCH * RO_MAX_HYP) != hipSuccess) return fail(FT_E_NOMEM);
    c->ro_gcnt_dev = (int*)p;
    if (hipMemset(c->ro_gcnt_dev, 0, sizeof(int) * FT_MAX_BATCH * RO_MAX_HYP) != hipSuccess) return fail(FT_E_HIP);
    if (hipMalloc(&p, sizeof(unsigned) * 2 * FT_MAX_BATCH) != hipSuccess) return fail(FT_E_NOMEM);
    c->ro_gbar_dev = (unsigned*)p;
    if (hipMemset(c->ro_gbar_dev, 0, sizeof(unsigned) * 2 * FT_MAX_BATCH) != hipSuccess) return fail(FT_E_HIP);
    if (const char* e = getenv("FT_RO_SPLIT")) c->ro_split = atoi(e) != 0;
    if (const char* e = getenv("FT_DONE_FLAG")) c->use_done_flag = atoi(e) != 0;
    if (const char* e = getenv("FT_RO_SPIN_LIMIT")) c->ro_spin_limit = (unsigned)std::max(1, atoi(e));
    if (hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
    c->done_flag_host = (unsigned*)p;
    memset(c->done_flag_host, 0, 256);
    if (hipHostGetDevicePointer(&p, c->done_flag_host, 0) != hipSuccess) return fail(FT_E_HIP);
    c->done_flag_dev = (unsigned*)p;

    // FAST scratch
    c->cand_entries_per_img = ((size_t)lim->max_width + 2 * 64) * ((size_t)lim->max_height + 2 * 64) / 4 + (size_t)lim->max_grid_cells;
    if (hipMalloc(&p, sizeof(uint32_t) * c->cand_entries_per_img * lim->n_streams) != hipSuccess) return fail(FT_E_NOMEM);
    c->cand = (uint32_t*)p;
    const size_t cells_total = (size_t)lim->n_streams * lim->max_grid_cells;
    if (hipMalloc(&p, sizeof(int) * cells_total * FAST_COUNT_STRIDE) != hipSuccess) return fail(FT_E_NOMEM);
    c->cand_count = (int*)p;
    // the per-cell candidate counters start at zero and every top-k pass leaves them at zero again
    if (hipMemset(c->cand_count, 0, sizeof(int) * cells_total * FAST_COUNT_STRIDE) != hipSuccess) return fail(FT_E_HIP);
    c->fast_out_bytes = align_up(sizeof(ft_keypoint) * cells_total * lim->max_per_cell, 16) + sizeof(int) * cells_total * 2 + 256;
    for (int i = 0; i < ft_ctx::FAST_RING; i++) {
        if (hipHostMalloc(&p, c->fast_out_bytes, hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
        c->fast_out_host[i] = (uint8_t*)p;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) return fail(FT_E_HIP);
        c->fast_out_host_dev[i] = (uint8_t*)dp;
    }
    for (auto& f : c->fences)
        if (hipEventCreateWithFlags(&f.ev, hipEventDisableTiming) != hipSuccess) return fail(FT_E_HIP);

    c->pf_points = N;
    if (hipMalloc(&p, sizeof(float) * (4 * N + 16)) != hipSuccess) return fail(FT_E_NOMEM);
    c->pf_dev = (float*)p;
    if (hipHostMalloc(&p, sizeof(float) * (4 * N + 16), hipHostMallocDefault) != hipSuccess) return fail(FT_E_NOMEM);
    c->pf_host = (float*)p;

    if (hipEventCreate(&c->region_a) != hipSuccess || hipEventCreate(&c->region_b) != hipSuccess) return fail(FT_E_HIP);
    if (c->front_env == 1 && ft_set_front_queue(c, 1) != FT_OK) return fail(FT_E_HIP);
    *out = c;
    return FT_OK;
}

void ft_ctx_destroy(ft_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->front_q_) { (void)hipStreamSynchronize(c->front_q_); (void)hipStreamDestroy(c->front_q_); c->front_q_ = nullptr; }
    if (c->front_ev_) (void)hipEventDestroy(c->front_ev_);
    c->cur_ = c->q_; c->front_dirty_ = false;
    c->deferred.active = false;
    if (enq(c)) (void)hipStreamSynchronize(enq(c));
    for (auto& e : c->stage_done) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->stage_batch_ev) if (e) (void)hipEventDestroy(e);
    for (auto& ep : c->ev_inflight) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto& ep : c->ev_pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (c->region_a) (void)hipEventDestroy(c->region_a);
    if (c->region_b) (void)hipEventDestroy(c->region_b);
    if (c->arena.base) (void)hipFree(c->arena.base);
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->stage_bar) (void)hipFree(c->stage_bar);
    if (c->lk_in_host) (void)hipHostFree(c->lk_in_host);
    if (c->lk_in_dev) (void)hipFree(c->lk_in_dev);
    if (c->lk_out_host) (void)hipHostFree(c->lk_out_host);
    if (c->lk_out_dev) (void)hipFree(c->lk_out_dev);
    if (c->lk_norm_dev) (void)hipFree(c->lk_norm_dev);
    if (c->ingest_dev) (void)hipFree(c->ingest_dev);
    if (c->done_count_dev) (void)hipFree(c->done_count_dev);
    if (c->ro_gcnt_dev) (void)hipFree(c->ro_gcnt_dev);
    if (c->ro_gbar_dev) (void)hipFree(c->ro_gbar_dev);
    if (c->done_flag_host) (void)hipHostFree(c->done_flag_host);
    if (c->match_dev) (void)hipFree(c->match_dev);
    if (c->match_host) (void)hipHostFree(c->match_host);
    if (c->cand) (void)hipFree(c->cand);
    if (c->cand_count) (void)hipFree(c->cand_count);
    for (auto& h : c->fast_out_host) if (h) (void)hipHostFree(h);
    for (auto& f : c->fences) if (f.ev) (void)hipEventDestroy(f.ev);
    if (c->pf_dev) (void)hipFree(c->pf_dev);
    if (c->pf_host) (void)hipHostFree(c->pf_host);
    if (enq(c)) (void)hipStreamDestroy(enq(c));
    delete c;
}

int ft_upload(ft_ctx* c, int stream, int slot, const uint8_t* host, int w, int h, size_t stride) {
    if (!c || !host || w < 1 || h < 1 || stride < (size_t)w) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    if (si < 0) return FT_E_INVALID;
    if (w > c->lim.max_width || h > c->lim.max_height) return FT_E_LIMIT;
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    { const int rc = stage_rows(c, si, host, w, h, stride); if (rc) return rc; }
    uint8_t* st = c->stage + (size_t)si * c->stage_bytes_per_slot;
    uint8_t* dst = c->arena.base + (uint64_t)si * c->arena.slot_bytes + c->arena.lv[0].img_off;
    FT_HIP(c, hipMemcpy2DAsync(dst, c->arena.lv[0].img_pitch, st, (size_t)w, (size_t)w, (size_t)h, hipMemcpyHostToDevice, enq(c)));
    FT_HIP(c, hipEventRecord(c->stage_done[si], enq(c)));
    c->stage_busy[si] = 1;
    c->slots[si] = SlotState{w, h, 1, 0};
    return FT_OK;
}

int ft_set_frames_device(ft_ctx* c, int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride) {
    if (!c || n < 1 || n > FT_MAX_BATCH || !streams || !slots || !dev || w < 1 || h < 1 || stride < (size_t)w) return FT_E_INVALID;
    if (w > c->lim.max_width || h > c->lim.max_height) return FT_E_LIMIT;
    ImgList L;
    L.n = n;
    for (int i = 0; i < n; i++) {
        const int si = slot_index(c, streams[i], slots[i]);
        if (si < 0 || !dev[i]) return FT_E_INVALID;
        L.idx[i] = si;
    }
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    for (int first = 0; first < n; first += FRAME_PTRS) {
        const int m = std::min(FRAME_PTRS, n - first);
        FramePtrs F;
        for (int i = 0; i < m; i++) F.p[i] = static_cast<const uint8_t*>(dev[first + i]);
        dim3 grid(((w + 15) / 16 + 255) / 256, h, m);
        hipLaunchKernelGGL(ingest_kernel, grid, dim3(256), 0, enq(c), c->arena, L, F, first, w, h, (uint64_t)stride);
    }
    FT_HIP(c, hipGetLastError());
    for (int i = 0; i < n; i++) c->slots[L.idx[i]] = SlotState{w, h, 1, 0};
    return FT_OK;
}

int ft_set_frame_device(ft_ctx* c, int stream, int slot, const void* dev, int w, int h, size_t stride) {
    return ft_set_frames_device(c, 1, &stream, &slot, &dev, w, h, stride);
}

int ft_build_pyramid(ft_ctx* c, int n, const int* streams, const int* slots, int max_level, int win_w, int win_h,
                     int with_gradients, int* achieved) {
    if (!c || max_level < 0 || win_w < 3 || win_h < 3) return FT_E_INVALID;
    if (max_level + 1 > c->lim.max_levels) return FT_E_LIMIT;
    ImgList L;
    int w = 0, h = 0;
    int rc = collect_images(c, n, streams, slots, L, w, h);
    if (rc) return rc;
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    const int top = achieved_level(w, h, win_w, win_h, max_level);
    rc = run_levels(c, L, w, h, top, true, with_gradients != 0);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        SlotState& S = c->slots[L.idx[i]];
        S.levels = top + 1;
        S.grad_levels = with_gradients ? top + 1 : 0;
        if (achieved) achieved[i] = top;
    }
    return FT_OK;
}

// level 0 and the pyramid of n frames readable by the device at src[i] (row stride `stride`)
static int frames_to_pyramid(ft_ctx* c, const ImgList& L, int n, const uint8_t* const* src, int w, int h, size_t stride, int max_level,
                             int win_w, int win_h, int* achieved) {
    const int top = achieved_level(w, h, win_w, win_h, max_level);
    // level 0 and level 1 come out of one pass over the frames
    EventPair ep;
    prof_begin(c, 0, ep);
    const int tiles = ((w + LV_TX - 1) / LV_TX) * ((h + LV_TY - 1) / LV_TY);
    for (int first = 0; first < n; first += FRAME_PTRS) {
        const int m = std::min(FRAME_PTRS, n - first);
        FramePtrs F{};
        for (int i = 0; i < m; i++) F.p[i] = src[first + i];
        hipLaunchKernelGGL((level_kernel<true, false>), dim3(xcd_grid(tiles * m)), dim3(256), 0, enq(c), c->arena, L, F, first, 0, w, h,
                           (uint64_t)stride, 0, top > 0 ? 1 : 0);
    }
    FT_HIP(c, hipGetLastError());
    if (top > 1) {
        if (c->front_fuse && !(c->profile & 1) && !c->front_enabled && n <= FRAME_PTRS) {
            // the detector call that usually follows takes these steps into its launches; anything else launches them first
            ft_ctx::DeferredLevels& D = c->deferred;
            D.L = L; D.w = w; D.h = h; D.top = top; D.done_to = 1;
            D.active = true;
        } else {
            const int rc = run_levels(c, L, w, h, top, true, false, 1, false);
            if (rc) return rc;
        }
    }
    prof_end(c, ep);   // one timed group: ingest + every level
    for (int i = 0; i < n; i++) {
        c->slots[L.idx[i]] = SlotState{w, h, top + 1, 0};
        if (achieved) achieved[i] = top;
    }
    return FT_OK;
}

int ft_set_frames_device_pyramid(ft_ctx* c, int n, const int* streams, const int* slots, const void* const* dev, int w, int h, size_t stride,
                                 int max_level, int win_w, int win_h, int* achieved) {
    if (!c || n < 1 || n > FT_MAX_BATCH || !streams || !slots || !dev || w < 1 || h < 1 || stride < (size_t)w) return FT_E_INVALID;
    if (max_level < 0 || win_w < 3 || win_h < 3) return FT_E_INVALID;
    if (w > c->lim.max_width || h > c->lim.max_height || max_level + 1 > c->lim.max_levels) return FT_E_LIMIT;
    ImgList L;
    L.n = n;
    const uint8_t* src[FT_MAX_BATCH];
    for (int i = 0; i < n; i++) {
        const int si = slot_index(c, streams[i], slots[i]);
        if (si < 0 || !dev[i]) return FT_E_INVALID;
        L.idx[i] = si;
        src[i] = static_cast<const uint8_t*>(dev[i]);
    }
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    return frames_to_pyramid(c, L, n, src, w, h, stride, max_level, win_w, win_h, achieved);
}

int ft_stage_frame(ft_ctx* c, int stream, int slot, const uint8_t* host, int w, int h, size_t stride) {
    if (!c || !host || w < 1 || h < 1 || stride < (size_t)w) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    if (si < 0) return FT_E_INVALID;
    if (w > c->lim.max_width || h > c->lim.max_height) return FT_E_LIMIT;
    return stage_rows(c, si, host, w, h, stride, true);
}

int ft_set_frames_staged_pyramid(ft_ctx* c, int n, const int* streams, const int* slots, int w, int h, int max_level, int win_w, int win_h,
                                 int* achieved) {
    if (!c || n < 1 || n > FT_MAX_BATCH || !streams || !slots || w < 1 || h < 1) return FT_E_INVALID;
    if (max_level < 0 || win_w < 3 || win_h < 3) return FT_E_INVALID;
    if (w > c->lim.max_width || h > c->lim.max_height || max_level + 1 > c->lim.max_levels) return FT_E_LIMIT;
    ImgList L;
    L.n = n;
    const uint8_t* src[FT_MAX_BATCH];
    for (int i = 0; i < n; i++) {
        const int si = slot_index(c, streams[i], slots[i]);
        if (si < 0) return FT_E_INVALID;
        L.idx[i] = si;
        src[i] = c->stage_dev + (size_t)si * c->stage_bytes_per_slot;
    }
    bool all_in_bar = c->stage_bar != nullptr;
    for (int i = 0; i < n && all_in_bar; i++) all_in_bar = c->stage_in_bar[L.idx[i]] != 0;
    if (all_in_bar)
        for (int i = 0; i < n; i++) src[i] = c->stage_bar + (size_t)L.idx[i] * c->stage_bytes_per_slot;
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    int rc;
    if (!all_in_bar && w % 16 == 0 && (size_t)n * w * h >= ((size_t)2 << 20)) {
        // Many frames: the bus is the limit, so the staged bytes cross it in one linear pass into level 0 and the
        // pyramid is then built from device memory.  (A frame or two take the fused pass below: one launch less.)
        const int top = achieved_level(w, h, win_w, win_h, max_level);
        EventPair ep;
        prof_begin(c, 0, ep);
        const int n16 = w * h / 16;
        // Staging buffers at a constant distance (the same slot of consecutive streams: the usual call) come over with
        // one copy command, on the copy engine and at its rate whatever else occupies the compute units; then level 0
        // is written from device memory.  Otherwise the kernel below reads the staging over the bus itself.
        bool regular = n >= 2 && n <= c->lim.n_streams;
        const ptrdiff_t dist = n >= 2 ? src[1] - src[0] : 0;
        for (int i = 2; i < n && regular; i++) regular = src[i] - src[i - 1] == dist;
        regular = regular && dist >= (ptrdiff_t)w * h;
        if (regular) {
            const uint8_t* host0 = c->stage + (src[0] - c->stage_dev);
            FT_HIP(c, hipMemcpy2DAsync(c->ingest_dev, (size_t)w * h, host0, (size_t)dist, (size_t)w * h, (size_t)n, hipMemcpyHostToDevice, enq(c)));
            for (int i = 0; i < n; i++) src[i] = c->ingest_dev + (size_t)i * w * h;
        }
        for (int first = 0; first < n; first += FRAME_PTRS) {
            const int m = std::min(FRAME_PTRS, n - first);
            FramePtrs F{};
            for (int i = 0; i < m; i++) F.p[i] = src[first + i];
            hipLaunchKernelGGL(ingest_linear_kernel, dim3((n16 + 255) / 256, m), dim3(256), 0, enq(c), c->arena, L, F, first, w / 16, n16);
        }
        FT_HIP(c, hipGetLastError());
        rc = top > 0 ? run_levels(c, L, w, h, top, true, false, 0, false) : FT_OK;
        prof_end(c, ep);
        if (rc) return rc;
        for (int i = 0; i < n; i++) {
            c->slots[L.idx[i]] = SlotState{w, h, top + 1, 0};
            if (achieved) achieved[i] = top;
        }
    } else {
        rc = frames_to_pyramid(c, L, n, src, w, h, (size_t)w, max_level, win_w, win_h, achieved);
        if (rc) return rc;
    }
    // the staging buffers are free again once the pass that reads them has run
    const int e = c->stage_batch_next;
    c->stage_batch_next = (e + 1) % ft_ctx::STAGE_EV_RING;
    FT_HIP(c, hipEventRecord(c->stage_batch_ev[e], enq(c)));
    for (int i = 0; i < n; i++) c->stage_batch_idx[L.idx[i]] = e;
    return FT_OK;
}

int ft_build_gradients(ft_ctx* c, int n, const int* streams, const int* slots) {
    if (!c) return FT_E_INVALID;
    ImgList L;
    int w = 0, h = 0;
    int rc = collect_images(c, n, streams, slots, L, w, h);
    if (rc) return rc;
    int levels = c->slots[L.idx[0]].levels;
    for (int i = 1; i < n; i++) if (c->slots[L.idx[i]].levels != levels) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    rc = run_levels(c, L, w, h, levels - 1, false, true);
    if (rc) return rc;
    for (int i = 0; i < n; i++) c->slots[L.idx[i]].grad_levels = levels;
    return FT_OK;
}

int ft_lk_track_async(ft_ctx* c, int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm) {
    if (!c || !jobs || !prm || n_jobs < 1 || n_jobs > FT_MAX_BATCH) return FT_E_INVALID;
    if (prm->win_w < 3 || prm->win_h < 3 || prm->win_w > FT_MAX_WIN || prm->win_h > FT_MAX_WIN || prm->max_level < 0) return FT_E_INVALID;
    if (prm->max_level + 1 > FT_MAX_LEVELS) return FT_E_LIMIT;
    const int seglen = pick_seglen(prm->win_w, prm->win_h);
    if (!seglen) return FT_E_INVALID;
    const bool fb = (prm->flags & FT_LK_FORWARD_BACKWARD) != 0;
    const bool reject = (prm->flags & FT_LK_REJECT_OUTLIERS) != 0;
    if (reject && (prm->ransac_hypotheses < 0 || prm->ransac_hypotheses > RO_MAX_HYP)) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    if (c->lk_pending.active) {   // staging is single-buffered: finish the previous batch first
        int rc = sync_and_deliver(c);
        if (rc) return rc;
    }

    size_t total = 0;
    bool want_iters = false, stereo = false, need_guess = false;
    for (int j = 0; j < n_jobs; j++) {
        const ft_lk_job& J = jobs[j];
        if (J.n < 0) return FT_E_INVALID;
        if (J.n > 0 && (!J.prev_xy || !J.next_xy || !J.status || !J.err)) return FT_E_INVALID;
        if ((reject || J.stereo_Rt) && J.n > 0 && (!J.calib_prev || !J.calib_next)) return FT_E_INVALID;
        total += (size_t)J.n;
        if (J.iters) want_iters = true;
        if (J.stereo_Rt && J.n > 0) stereo = true;
        if (J.n > 0 && (!J.H || J.stereo_Rt)) need_guess = true;
    }
    const bool count_work = (c->profile & 2) != 0;
    if (count_work) want_iters = true;   // the counters come from the per-level iteration table
    if (total > (size_t)c->lim.max_points) return FT_E_LIMIT;
    const int nlev = prm->max_level + 1;

    // pack the input block
    const size_t N = total;
    uint8_t* hb = c->lk_in_host;
    // laid out for this batch's sizes, so that the fetch moves no unused bytes
    const size_t off_ptjob = align_up(sizeof(LKJobDev) * (size_t)n_jobs, 256);
    const size_t off_prev = off_ptjob + align_up(2 * N, 256);
    const size_t off_next = off_prev + align_up(8 * N, 256);
    LKJobDev* dj = reinterpret_cast<LKJobDev*>(hb);
    uint16_t* ptjob = reinterpret_cast<uint16_t*>(hb + off_ptjob);
    float* pxy = reinterpret_cast<float*>(hb + off_prev);
    float* nxy = reinterpret_cast<float*>(hb + off_next);
    int max_level_all = -1;
    size_t off = 0;
    for (int j = 0; j < n_jobs; j++) {
        const ft_lk_job& J = jobs[j];
        const int sp = slot_index(c, J.stream_prev, J.slot_prev), sn = slot_index(c, J.stream_next, J.slot_next);
        if (sp < 0 || sn < 0) return FT_E_INVALID;
        const SlotState& P = c->slots[sp];
        const SlotState& Q = c->slots[sn];
        if (P.levels < 1 || Q.levels < 1) return FT_E_STATE;
        if (P.w != Q.w || P.h != Q.h) return FT_E_INVALID;
        const int ml = std::min(prm->max_level, std::min(P.levels, Q.levels) - 1);
        if (max_level_all < 0) max_level_all = ml;
        else if (ml != max_level_all) return FT_E_INVALID;   // one batch, one pyramid depth
        dj[j].img_prev = sp; dj[j].img_next = sn; dj[j].w0 = P.w; dj[j].h0 = P.h;
        dj[j].use_H = (J.H && !J.stereo_Rt) ? 1 : 0;
        if (J.H) memcpy(dj[j].H, J.H, sizeof(float) * 9); else memset(dj[j].H, 0, sizeof(float) * 9);
        dj[j].first = (int)off; dj[j].count = J.n;
        if ((reject || J.stereo_Rt) && J.n > 0) { memcpy(dj[j].calib, J.calib_prev, sizeof(double) * 8); memcpy(dj[j].calib + 8, J.calib_next, sizeof(double) * 8); }
        else memset(dj[j].calib, 0, sizeof(dj[j].calib));
        dj[j].use_stereo = (J.stereo_Rt && J.n > 0) ? 1 : 0; dj[j].pad_ = 0;
        if (dj[j].use_stereo) memcpy(dj[j].stereo, J.stereo_Rt, sizeof(double) * 12); else memset(dj[j].stereo, 0, sizeof(dj[j].stereo));
        for (int i = 0; i < J.n; i++) ptjob[off + i] = (uint16_t)j;
        if (J.n) {
            memcpy(pxy + 2 * off, J.prev_xy, 8 * (size_t)J.n);
            if (need_guess) memcpy(nxy + 2 * off, J.next_xy, 8 * (size_t)J.n);
        }
        off += (size_t)J.n;
    }
    if (N == 0) return FT_OK;

    // Large batches: one transfer for the whole block (jobs .. end of next_xy in use), so that the tracking waves read
    // their inputs from device memory.  Small batches (a single stream's worth of points) read the pinned block in
    // place: the extra dependent launch would cost more than the few reads over the bus.  With the outlier rejection
    // behind the tracking, the tracking wavefronts leave device copies of the job records and the template points for
    // it (its workgroups reading them over the bus as well cost more than the launch saved: single-stream latency 77
    // -> 83 us, 8 streams 16.9 -> 16.3 M features/s).
    // Where the host writes the device copy itself, batches of up to lk_bar_max points take that way (64 streams in four
    // groups 75.0 -> 77.1 M features/s, single-stream p50 79.0 -> 75.8 us; a batch of 9 700 points is better off with the
    // fetch kernel: its 100 KB cost the host more than the launch, 39.1 -> 36.5 M).
    const bool bar_in = c->lk_in_bar && N <= (size_t)c->lk_bar_max;
    const bool direct_in = !bar_in && N <= (size_t)c->lk_direct_max && !stereo;
    const bool echo = direct_in && reject;
    const uint8_t* ib = direct_in ? c->lk_in_host_dev : c->lk_in_dev;
    // (every job seeded by its homography: the kernel never looks at the guesses, and they stay behind)
    const size_t in_bytes = (need_guess ? off_next : off_prev) + 8 * N;
    if (bar_in) {
        // nothing on the device reads the block any more (the previous batch was delivered above); the stores leave the
        // write-combining buffers before the launch's doorbell goes out
        memcpy(c->lk_in_dev, hb, in_bytes);
        __builtin_ia32_sfence();
    } else if (!direct_in) {
        const int n16 = (int)((in_bytes + 15) / 16);
        hipLaunchKernelGGL(fetch_block_kernel, dim3((n16 + 255) / 256), dim3(256), 0, enq(c), reinterpret_cast<const uint4*>(c->lk_in_host_dev),
                           reinterpret_cast<uint4*>(c->lk_in_dev), n16);
    }

    LKArgs a;
    a.A = c->arena;
    a.jobs = reinterpret_cast<const LKJobDev*>(ib);
    a.pt_job = reinterpret_cast<const uint16_t*>(ib + off_ptjob);
    a.prev_xy = reinterpret_cast<const float*>(ib + off_prev);
    a.guess_xy = reinterpret_cast<const float*>(ib + (need_guess ? off_next : off_prev));   // read by every wave, used or not
    // Without a backward pass (which reads the forward results on the device) and without iteration counters, lane 0
    // of every wave writes its 13 result bytes straight into the pinned host block: no copy command to enqueue.
    const bool direct_out = !fb && !want_iters && !reject && !stereo;
    uint8_t* ob = direct_out ? c->lk_out_host_dev : c->lk_out_dev;
    StereoArgs sa;
    if (stereo) {   // guesses of the stereo jobs from the extrinsic rotation, written over the fetched guess block
        sa.jobs = a.jobs; sa.pt_job = a.pt_job; sa.prev_xy = a.prev_xy;
        sa.next_xy = reinterpret_cast<float*>(c->lk_in_dev + off_next);
        sa.err = nullptr; sa.status = nullptr; sa.out_xy = nullptr; sa.out_err = nullptr; sa.out_status = nullptr;
        sa.total = (int)N; sa.emit = 0; sa.threshold = prm->stereo_threshold;
        hipLaunchKernelGGL(stereo_seed_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, enq(c), sa);
    }
    a.next_xy = reinterpret_cast<float*>(ob);
    a.err = reinterpret_cast<float*>(ob + 8 * N);
    a.status = ob + 12 * N;
    a.iters = want_iters ? reinterpret_cast<int*>(ob + align_up(13 * N, 16)) : nullptr;
    a.total = (int)N;
    a.nlev = nlev;
    a.win_w = prm->win_w; a.win_h = prm->win_h;
    a.segs = (prm->win_w + seglen - 1) / seglen;
    a.segs_inv = (65536 + a.segs - 1) / a.segs;
    a.max_level = max_level_all;
    a.max_iter = std::min(std::max(prm->max_iter, 0), 100);
    a.flags = prm->flags;
    a.backward = 0;
    a.echo_jobs = echo ? reinterpret_cast<LKJobDev*>(c->lk_in_dev) : nullptr;
    a.echo_prev_xy = echo ? reinterpret_cast<float*>(c->lk_in_dev + off_prev) : nullptr;
    a.n_jobs = n_jobs;
    a.fb_thresh2 = prm->fb_thresh * prm->fb_thresh;
    const double eps = std::min(std::max(prm->eps, 0.), 10.);
    a.epsilon = eps * eps;
    a.min_eig = prm->min_eig;
    if (want_iters) FT_HIP(c, hipMemsetAsync(a.iters, 0xff, sizeof(int) * N * nlev, enq(c)));

    EventPair ep;
    prof_begin(c, 1, ep);
    if (fb && c->lk_fused_fb) {
        // forward and backward pass of every feature in one launch, by the same wavefront(s)
        launch_lk_seglen<true>(c, a, (int)N, seglen);
    } else {
        for (int pass = 0; pass < (fb ? 2 : 1); pass++) {
            a.backward = pass;
            if (pass == 1) { a.iters = nullptr; a.echo_jobs = nullptr; a.echo_prev_xy = nullptr; }
            launch_lk_seglen<false>(c, a, (int)N, seglen);
        }
    }
    prof_end(c, ep);
    FT_HIP(c, hipGetLastError());

    if (stereo) {   // bounds + epipolar test of the stereo jobs; without a rejection pass after it, it also emits the results
        sa.next_xy = a.next_xy; sa.err = a.err; sa.status = a.status;
        sa.out_xy = reinterpret_cast<float*>(c->lk_out_host_dev);
        sa.out_err = reinterpret_cast<float*>(c->lk_out_host_dev + 8 * N);
        sa.out_status = c->lk_out_host_dev + 12 * N;
        sa.emit = reject ? 0 : 1;
        hipLaunchKernelGGL(stereo_check_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, enq(c), sa);
        FT_HIP(c, hipGetLastError());
        if (!reject && want_iters)
            FT_HIP(c, hipMemcpyAsync(c->lk_out_host + align_up(13 * N, 16), ob + align_up(13 * N, 16), sizeof(int) * N * nlev, hipMemcpyDeviceToHost, enq(c)));
    }
    if (reject) {
        // final mask on the device; the kernel also moves positions, residuals and status into the pinned host block
        OutlierArgs o;
        o.jobs = echo ? reinterpret_cast<const LKJobDev*>(c->lk_in_dev) : a.jobs;
        o.prev_xy = echo ? reinterpret_cast<const float*>(c->lk_in_dev + off_prev) : a.prev_xy; o.next_xy = a.next_xy; o.err = a.err; o.status = a.status;
        o.norm = c->lk_norm_dev;
        o.out_xy = reinterpret_cast<float*>(c->lk_out_host_dev);
        o.out_err = reinterpret_cast<float*>(c->lk_out_host_dev + 8 * N);
        o.out_status = c->lk_out_host_dev + 12 * N;
        o.hypotheses = prm->ransac_hypotheses > 0 ? prm->ransac_hypotheses : FT_RO_DEFAULT_HYP;
        o.seed = prm->ransac_seed;
        int counts[FT_MAX_BATCH];
        for (int j = 0; j < n_jobs; j++) counts[j] = jobs[j].n;
        launch_reject(c, o, n_jobs, counts);
        FT_HIP(c, hipGetLastError());
        if (want_iters)
            FT_HIP(c, hipMemcpyAsync(c->lk_out_host + align_up(13 * N, 16), ob + align_up(13 * N, 16), sizeof(int) * N * nlev, hipMemcpyDeviceToHost, enq(c)));
    } else if (!direct_out && !stereo) {
        const size_t out_bytes = align_up(13 * N, 16) + (want_iters ? sizeof(int) * N * nlev : 0);
        FT_HIP(c, hipMemcpyAsync(c->lk_out_host, ob, out_bytes, hipMemcpyDeviceToHost, enq(c)));
    }

    PendingLK& P = c->lk_pending;
    P.jobs.assign(jobs, jobs + n_jobs);
    P.total = (int)N; P.nlev = nlev; P.want_iters = want_iters; P.count_work = count_work; P.active = true;
    P.seq = ++c->seq;
    return FT_OK;
}

int ft_reject_outliers_async(ft_ctx* c, int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed) {
    if (!c || !sets || n_sets < 1 || n_sets > FT_MAX_BATCH || hypotheses < 0 || hypotheses > RO_MAX_HYP) return FT_E_INVALID;
    size_t N = 0;
    for (int j = 0; j < n_sets; j++) {
        const ft_match_set& S = sets[j];
        if (S.n < 0 || (S.n > 0 && (!S.prev_xy || !S.next_xy || !S.status || !S.calib_prev || !S.calib_next))) return FT_E_INVALID;
        N += (size_t)S.n;
    }
    if (N > (size_t)c->lim.max_points) return FT_E_LIMIT;
    if (N == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    if (c->lk_pending.active) {   // the tracking batches' staging blocks are reused
        int rc = sync_and_deliver(c);
        if (rc) return rc;
    }
    // same input block as a tracking batch; the per-point job index is not needed here, its place holds the status bytes
    uint8_t* hb = c->lk_in_host;
    const size_t off_status = align_up(sizeof(LKJobDev) * (size_t)n_sets, 256);
    const size_t off_prev = off_status + align_up(2 * N, 256);
    const size_t off_next = off_prev + align_up(8 * N, 256);
    LKJobDev* dj = reinterpret_cast<LKJobDev*>(hb);
    std::vector<ft_lk_job> deliver((size_t)n_sets);
    size_t off = 0;
    for (int j = 0; j < n_sets; j++) {
        const ft_match_set& S = sets[j];
        memset(&dj[j], 0, sizeof(LKJobDev));
        dj[j].first = (int)off; dj[j].count = S.n;
        memset(&deliver[j], 0, sizeof(ft_lk_job));
        deliver[j].n = S.n; deliver[j].status = S.status;
        if (S.n) {
            memcpy(dj[j].calib, S.calib_prev, sizeof(double) * 8);
            memcpy(dj[j].calib + 8, S.calib_next, sizeof(double) * 8);
            memcpy(hb + off_status + off, S.status, (size_t)S.n);
            memcpy(hb + off_prev + 8 * off, S.prev_xy, 8 * (size_t)S.n);
            memcpy(hb + off_next + 8 * off, S.next_xy, 8 * (size_t)S.n);
        }
        off += (size_t)S.n;
    }
    if (c->lk_in_bar && N <= (size_t)c->lk_bar_max) {
        memcpy(c->lk_in_dev, hb, off_next + 8 * N);
        __builtin_ia32_sfence();
    } else {
        const int n16 = (int)((off_next + 8 * N + 15) / 16);
        hipLaunchKernelGGL(fetch_block_kernel, dim3((n16 + 255) / 256), dim3(256), 0, enq(c), reinterpret_cast<const uint4*>(c->lk_in_host_dev),
                           reinterpret_cast<uint4*>(c->lk_in_dev), n16);
    }
    OutlierArgs o;
    o.jobs = reinterpret_cast<const LKJobDev*>(c->lk_in_dev);
    o.prev_xy = reinterpret_cast<const float*>(c->lk_in_dev + off_prev);
    o.next_xy = reinterpret_cast<const float*>(c->lk_in_dev + off_next);
    o.err = reinterpret_cast<const float*>(c->lk_out_dev + 8 * N);   // passed through by the kernel, not delivered
    o.status = c->lk_in_dev + off_status;
    o.norm = c->lk_norm_dev;
    o.out_xy = reinterpret_cast<float*>(c->lk_out_host_dev);
    o.out_err = reinterpret_cast<float*>(c->lk_out_host_dev + 8 * N);
    o.out_status = c->lk_out_host_dev + 12 * N;
    o.hypotheses = hypotheses > 0 ? hypotheses : FT_RO_DEFAULT_HYP;
    o.seed = seed;
    int counts[FT_MAX_BATCH];
    for (int j = 0; j < n_sets; j++) counts[j] = sets[j].n;
    launch_reject(c, o, n_sets, counts);
    FT_HIP(c, hipGetLastError());
    PendingLK& P = c->lk_pending;
    P.jobs = std::move(deliver);
    P.total = (int)N; P.nlev = 0; P.want_iters = false; P.count_work = false; P.active = true;
    P.seq = ++c->seq;
    return FT_OK;
}

int ft_reject_outliers(ft_ctx* c, int n_sets, const ft_match_set* sets, int hypotheses, unsigned seed) {
    int rc = ft_reject_outliers_async(c, n_sets, sets, hypotheses, seed);
    if (rc) return rc;
    return ft_sync(c);
}

int ft_lk_track(ft_ctx* c, int n_jobs, const ft_lk_job* jobs, const ft_lk_params* prm) {
    int rc = ft_lk_track_async(c, n_jobs, jobs, prm);
    if (rc) return rc;
    return ft_sync(c);
}

int ft_fast_grid_async(ft_ctx* c, int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y,
                       int per_cell, ft_keypoint* out, int cap, int* n_out, int* cell_counts) {
    if (!c || !out || cap < 0 || grid_x < 1 || grid_y < 1 || per_cell < 1) return FT_E_INVALID;
    ImgList L;
    int w = 0, h = 0;
    int rc = collect_images(c, n, streams, slots, L, w, h);
    if (rc) return rc;
    if (n > c->lim.n_streams) return FT_E_LIMIT;
    FastGrid G;
    G.w = w; G.h = h;
    G.size_x = w / grid_x; G.size_y = h / grid_y;                    // Grider_FAST.h:80-81
    if (G.size_x < 1 || G.size_y < 1 || G.size_x > 4096 || G.size_y > 4096) return FT_E_INVALID;
    G.ct_cols = w / G.size_x; G.ct_rows = h / G.size_y;              // Grider_FAST.h:91-92
    G.inv_x = (uint32_t)(((1ull << 32) + (uint64_t)G.size_x - 1) / (uint64_t)G.size_x);   // unused for a size of 1
    G.inv_y = (uint32_t)(((1ull << 32) + (uint64_t)G.size_y - 1) / (uint64_t)G.size_y);
    if (w > 65535 || h > 65535) return FT_E_LIMIT;                   // range of the reciprocal division
    const int cells = G.ct_cols * G.ct_rows;
    if (cells > c->lim.max_grid_cells || per_cell > c->lim.max_per_cell) return FT_E_LIMIT;
    G.threshold = std::min(std::max(threshold, 0), 255);
    G.per_cell = per_cell;
    // a strict 3x3 maximum leaves at most one survivor per 2x2 block of the cell
    const size_t need = (size_t)((G.size_x + 1) / 2) * ((G.size_y + 1) / 2);
    G.cand_cap = (int)std::min(need, c->cand_entries_per_img / (size_t)cells);
    FT_HIP(c, hipSetDevice(c->device));
    FrontScope front(c);
    const int ring = c->fast_next;
    if (c->fast_pending[ring].active) {   // both host buffers hold undelivered results: finish the queue first
        rc = sync_and_deliver(c);
        if (rc) return rc;
    }
    // the top-k kernel writes its (small) results straight into the pinned host buffer: no copy command to enqueue
    const size_t kp_bytes = sizeof(ft_keypoint) * (size_t)n * cells * per_cell;
    const size_t cnt_off = align_up(kp_bytes, 16);
    uint8_t* out_block = c->fast_out_host_dev[ring];
    ft_keypoint* kp_dev = reinterpret_cast<ft_keypoint*>(out_block);
    int* kept_dev = reinterpret_cast<int*>(out_block + cnt_off);
    int* surv_dev = kept_dev + (size_t)n * cells;
    const int tiles = fast_tiles_x(w) * fast_tiles_y(h);
    // pyrDown steps of these very images that are still to be launched ride along with the detector and the ranking
    ft_ctx::DeferredLevels& D = c->deferred;
    bool fuse = D.active && !D.busy && D.L.n == n && D.w == w && D.h == h && D.done_to == 1 && D.top >= 2;
    for (int i = 0; i < n && fuse; i++) fuse = D.L.idx[i] == L.idx[i];
    if (fuse) {
        D.active = false;
        D.busy = true;   // (enq() below must not look at the record again)
        const int w1 = (w + 1) / 2, h1 = (h + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2;
        const int lb1 = xcd_grid(((w1 + LV_TX - 1) / LV_TX) * ((h1 + LV_TY - 1) / LV_TY) * n);
        hipLaunchKernelGGL(front_detect_level_kernel, dim3(lb1 + xcd_grid((tiles * n + 3) / 4)), dim3(256), 0, enq(c), c->arena, L, G, c->cand, c->cand_count, 1, w1,
                           h1, lb1);
        int done_to = 2;
        if (D.top >= 3) {
            const int lb2 = xcd_grid(((w2 + LV_TX - 1) / LV_TX) * ((h2 + LV_TY - 1) / LV_TY) * n);
            hipLaunchKernelGGL(front_topk_level_kernel, dim3(lb2 + (n * cells + 3) / 4), dim3(256), 0, enq(c), G, c->cand, c->cand_count, kp_dev, kept_dev, surv_dev,
                               n * cells, c->arena, L, 2, w2, h2, lb2);
            done_to = 3;
        } else {
            hipLaunchKernelGGL(fast_topk_kernel, dim3((n * cells + 3) / 4), dim3(256), 0, enq(c), G, c->cand, c->cand_count, kp_dev, kept_dev, surv_dev, n * cells);
        }
        if (D.top > done_to) (void)run_levels(c, L, w, h, D.top, true, false, done_to, false);
        D.busy = false;
    } else {
        EventPair ep;
        prof_begin(c, 2, ep);
        hipLaunchKernelGGL(fast_detect_kernel, dim3(xcd_grid(tiles * n)), dim3(64), 0, enq(c), c->arena, L, G, c->cand, c->cand_count);
        hipLaunchKernelGGL(fast_topk_kernel, dim3((n * cells + 3) / 4), dim3(256), 0, enq(c), G, c->cand, c->cand_count, kp_dev, kept_dev, surv_dev, n * cells);
        prof_end(c, ep);
    }
    FT_HIP(c, hipGetLastError());
    // cand is indexed [image*cells + cell][cand_cap], the keypoints [image*cells + cell][per_cell]
    PendingFast& F = c->fast_pending[ring];
    F.active = true; F.seq = ++c->seq; F.n = n; F.cells = cells; F.per_cell = per_cell; F.cap = cap;
    F.out = out; F.n_out = n_out; F.cell_counts = cell_counts;
    c->fast_next = (ring + 1) % ft_ctx::FAST_RING;
    return FT_OK;
}

int ft_fast_grid(ft_ctx* c, int n, const int* streams, const int* slots, int threshold, int grid_x, int grid_y,
                 int per_cell, ft_keypoint* out, int cap, int* n_out, int* cell_counts) {
    int rc = ft_fast_grid_async(c, n, streams, slots, threshold, grid_x, grid_y, per_cell, out, cap, n_out, cell_counts);
    if (rc) return rc;
    return ft_sync(c);
}

void ft_homography(const float* K4, const float* R, float* H) {
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const float ifx = 1.f / fx, ify = 1.f / fy, ncx = -cx / fx, ncy = -cy / fy;
    float M[9];   // R * K^-1
    for (int r = 0; r < 3; r++) {
        M[3 * r] = R[3 * r] * ifx;
        M[3 * r + 1] = R[3 * r + 1] * ify;
        M[3 * r + 2] = R[3 * r] * ncx + R[3 * r + 1] * ncy + R[3 * r + 2];
    }
    for (int col = 0; col < 3; col++) {   // K * M
        H[col] = fx * M[col] + cx * M[6 + col];
        H[3 + col] = fy * M[3 + col] + cy * M[6 + col];
        H[6 + col] = M[6 + col];
    }
}

// descriptor staging (device + pinned host, 80 bytes per descriptor of capacity), grown on demand
static int ensure_match_cap(ft_ctx* c, size_t need) {
    if (need <= c->match_cap) return FT_OK;
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    if (c->match_dev) (void)hipFree(c->match_dev);
    if (c->match_host) (void)hipHostFree(c->match_host);
    c->match_dev = nullptr; c->match_host = nullptr; c->match_cap = 0;
    const size_t cap_n = align_up(need, 1024);
    void* p = nullptr;
    FT_HIP(c, hipMalloc(&p, cap_n * (32 + 8) * 2));
    c->match_dev = (uint8_t*)p;
    FT_HIP(c, hipHostMalloc(&p, cap_n * (32 + 8) * 2, hipHostMallocDefault));
    c->match_host = (uint8_t*)p;
    c->match_cap = cap_n;
    return FT_OK;
}

int ft_compute_descriptors(ft_ctx* c, int stream, int slot, const float* xy, int n, uint8_t* desc, uint8_t* valid) {
    if (!c || n < 0 || (n > 0 && (!xy || !desc || !valid))) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    if (si < 0) return FT_E_INVALID;
    if (c->slots[si].levels < 1) return FT_E_STATE;
    if (n > 65536) return FT_E_LIMIT;
    if (n == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    { const int rc = ensure_match_cap(c, (size_t)n); if (rc) return rc; }
    // layout of both blocks: positions (8 n), descriptors (32 n), valid flags (n)
    const size_t off_desc = align_up(8 * (size_t)n, 256), off_valid = off_desc + 32 * (size_t)n;
    memcpy(c->match_host, xy, 8 * (size_t)n);
    FT_HIP(c, hipMemcpyAsync(c->match_dev, c->match_host, 8 * (size_t)n, hipMemcpyHostToDevice, enq(c)));
    hipLaunchKernelGGL(brief_kernel, dim3(n), dim3(64), 0, enq(c), c->arena, si, c->slots[si].w, c->slots[si].h,
                       reinterpret_cast<const float*>(c->match_dev), n, c->match_dev + off_desc, c->match_dev + off_valid);
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(c->match_host + off_desc, c->match_dev + off_desc, 33 * (size_t)n, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    memcpy(desc, c->match_host + off_desc, 32 * (size_t)n);
    memcpy(valid, c->match_host + off_valid, (size_t)n);
    return FT_OK;
}

int ft_match_descriptors(ft_ctx* c, const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, float knn_ratio, ft_match* out, int cap, int* n_out) {
    if (!c || !n_out || n0 < 0 || n1 < 0 || cap < 0 || (cap > 0 && !out) || (n0 > 0 && !desc0) || (n1 > 0 && !desc1)) return FT_E_INVALID;
    if (n0 > 65536 || n1 > 65536) return FT_E_LIMIT;
    *n_out = 0;
    if (n0 == 0 || n1 == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    { const int rc = ensure_match_cap(c, (size_t)std::max(n0, n1)); if (rc) return rc; }
    // layout of both blocks: descriptors of set 0, of set 1, neighbour table of set 0, of set 1
    const size_t M = c->match_cap;
    const size_t off_d1 = 32 * M, off_b0 = 64 * M, off_b1 = 72 * M;
    memcpy(c->match_host, desc0, 32 * (size_t)n0);
    memcpy(c->match_host + off_d1, desc1, 32 * (size_t)n1);
    FT_HIP(c, hipMemcpyAsync(c->match_dev, c->match_host, 32 * (size_t)n0, hipMemcpyHostToDevice, enq(c)));
    FT_HIP(c, hipMemcpyAsync(c->match_dev + off_d1, c->match_host + off_d1, 32 * (size_t)n1, hipMemcpyHostToDevice, enq(c)));
    const uint4* d0 = reinterpret_cast<const uint4*>(c->match_dev);
    const uint4* d1 = reinterpret_cast<const uint4*>(c->match_dev + off_d1);
    hipLaunchKernelGGL(hamming_knn2_kernel, dim3((n0 + 3) / 4), dim3(256), 0, enq(c), d0, n0, d1, n1, reinterpret_cast<uint2*>(c->match_dev + off_b0));
    hipLaunchKernelGGL(hamming_knn2_kernel, dim3((n1 + 3) / 4), dim3(256), 0, enq(c), d1, n1, d0, n0, reinterpret_cast<uint2*>(c->match_dev + off_b1));
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(c->match_host + off_b0, c->match_dev + off_b0, 8 * (size_t)n0, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipMemcpyAsync(c->match_host + off_b1, c->match_dev + off_b1, 8 * (size_t)n1, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    // ratio and symmetry tests: O(n) on the host (TrackDescriptor.cpp:309-354)
    const uint2* b0 = reinterpret_cast<const uint2*>(c->match_host + off_b0);
    const uint2* b1 = reinterpret_cast<const uint2*>(c->match_host + off_b1);
    auto surviving = [knn_ratio](const uint2& b) -> int {
        if (b.y == MATCH_NONE) return -1;                                    // fewer than two neighbours
        const float first = (float)(b.x >> MATCH_IDX_BITS), second = (float)(b.y >> MATCH_IDX_BITS);
        if (first / second > knn_ratio) return -1;
        return (int)(b.x & ((1u << MATCH_IDX_BITS) - 1));
    };
    int n = 0;
    for (int i = 0; i < n0; i++) {
        const int j = surviving(b0[i]);
        if (j < 0 || surviving(b1[j]) != i) continue;
        if (n < cap) { out[n].query = i; out[n].train = j; out[n].distance = (float)(b0[i].x >> MATCH_IDX_BITS); }
        n++;
    }
    *n_out = n;
    return FT_OK;
}

int ft_predict_flow(ft_ctx* c, const float* H, const float* xy, float* xy_out, int n) {
    if (!c || !H || !xy || !xy_out || n < 0) return FT_E_INVALID;
    if ((size_t)n > c->pf_points) return FT_E_LIMIT;
    if (n == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    memcpy(c->pf_host, H, sizeof(float) * 9);
    memcpy(c->pf_host + 16, xy, sizeof(float) * 2 * (size_t)n);
    FT_HIP(c, hipMemcpyAsync(c->pf_dev, c->pf_host, sizeof(float) * (16 + 2 * (size_t)n), hipMemcpyHostToDevice, enq(c)));
    float* out_dev = c->pf_dev + 16 + 2 * c->pf_points;
    hipLaunchKernelGGL(predict_flow_kernel, dim3((n + 255) / 256), dim3(256), 0, enq(c), c->pf_dev, c->pf_dev + 16, out_dev, n);
    FT_HIP(c, hipGetLastError());
    float* out_host = c->pf_host + 16 + 2 * c->pf_points;
    FT_HIP(c, hipMemcpyAsync(out_host, out_dev, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    memcpy(xy_out, out_host, sizeof(float) * 2 * (size_t)n);
    return FT_OK;
}

int ft_level_info(ft_ctx* c, int stream, int slot, int level, int* w, int* h) {
    if (!c) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    if (si < 0 || level < 0) return FT_E_INVALID;
    const SlotState& S = c->slots[si];
    if (level >= S.levels) return FT_E_STATE;
    int lw = S.w, lh = S.h;
    for (int l = 0; l < level; l++) { lw = (lw + 1) / 2; lh = (lh + 1) / 2; }
    if (w) *w = lw;
    if (h) *h = lh;
    return FT_OK;
}

int ft_download_level(ft_ctx* c, int stream, int slot, int level, uint8_t* host, size_t stride) {
    int w = 0, h = 0;
    int rc = ft_level_info(c, stream, slot, level, &w, &h);
    if (rc) return rc;
    if (!host || stride < (size_t)w) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    FT_HIP(c, hipSetDevice(c->device));
    const uint8_t* src = c->arena.base + (uint64_t)si * c->arena.slot_bytes + c->arena.lv[level].img_off;
    FT_HIP(c, hipMemcpy2DAsync(host, stride, src, c->arena.lv[level].img_pitch, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    return FT_OK;
}

int ft_download_grad(ft_ctx* c, int stream, int slot, int level, int16_t* host, size_t stride_shorts) {
    int w = 0, h = 0;
    int rc = ft_level_info(c, stream, slot, level, &w, &h);
    if (rc) return rc;
    if (!host || stride_shorts < 2 * (size_t)w) return FT_E_INVALID;
    const int si = slot_index(c, stream, slot);
    if (level >= c->slots[si].grad_levels) return FT_E_STATE;
    FT_HIP(c, hipSetDevice(c->device));
    const uint8_t* src = c->arena.base + (uint64_t)si * c->arena.slot_bytes + c->arena.lv[level].grad_off;
    FT_HIP(c, hipMemcpy2DAsync(host, stride_shorts * 2, src, c->arena.lv[level].grad_pitch, (size_t)w * 4, (size_t)h, hipMemcpyDeviceToHost, enq(c)));
    FT_HIP(c, hipStreamSynchronize(enq(c)));
    return FT_OK;
}

int ft_sync(ft_ctx* c) {
    if (!c) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    return sync_and_deliver(c);
}

int ft_flush(ft_ctx* c) {
    if (!c) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    if (c->deferred.active) flush_deferred(c);
    FT_HIP(c, hipGetLastError());
    return FT_OK;
}

int ft_set_front_queue(ft_ctx* c, int on) {
    if (!c) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    if (c->front_env == 0) on = 0;
    if (on && !c->front_q_) {
        FT_HIP(c, hipStreamCreateWithFlags(&c->front_q_, hipStreamNonBlocking));
        FT_HIP(c, hipEventCreateWithFlags(&c->front_ev_, hipEventDisableTiming));
    }
    c->front_enabled = on != 0;
    return FT_OK;
}

int ft_fence(ft_ctx* c, uint64_t* token) {
    if (!c || !token) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    if (c->deferred.active) flush_deferred(c);
    if (c->front_dirty_) (void)enq(c);   // the fence covers the front queue's work as well
    const uint64_t id = ++c->fence_ids;
    ft_ctx::Fence& f = c->fences[id % ft_ctx::FENCE_RING];
    f.flag_seq = c->tail_flag_seq;   // the queue ends with a launch that reports its own completion: no event needed
    f.recorded = f.flag_seq == 0;
    if (f.recorded) FT_HIP(c, hipEventRecord(f.ev, c->q_));
    f.id = id;
    f.seq = c->seq;
    *token = id;
    return FT_OK;
}

int ft_wait(ft_ctx* c, uint64_t token) {
    if (!c) return FT_E_INVALID;
    ft_ctx::Fence& f = c->fences[token % ft_ctx::FENCE_RING];
    if (token == 0 || f.id != token) return FT_E_INVALID;   // never issued, or its ring entry was reused
    FT_HIP(c, hipSetDevice(c->device));
    bool done = false;
    if (f.flag_seq) {
        // poll the kernel's own completion flag; give up after a while and wait for the queue instead
        const volatile unsigned* flag = c->done_flag_host;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !done; spins++) {
            done = (int32_t)(*flag - f.flag_seq) >= 0;
            if (!done && (spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!done) {
        if (f.recorded) FT_HIP(c, hipEventSynchronize(f.ev));
        else FT_HIP(c, hipStreamSynchronize(c->q_));   // (the flag did not come: wait for the whole queue)
    }
    prof_collect(c);
    return deliver_pending(c, f.seq);
}

int ft_profile(ft_ctx* c, int enable) {
    if (!c) return FT_E_INVALID;
    if (enable < 0 || enable > 3) return FT_E_INVALID;
    c->profile = enable;
    return FT_OK;
}

int ft_kernel_times(ft_ctx* c, ft_kernel_time* out, int reset) {
    if (!c || !out) return FT_E_INVALID;
    int rc = ft_sync(c);
    if (rc) return rc;
    *out = c->times;
    if (reset) c->times = ft_kernel_time{};
    return FT_OK;
}

int ft_queue_time_begin(ft_ctx* c) {
    if (!c) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    FT_HIP(c, hipEventRecord(c->region_a, enq(c)));
    return FT_OK;
}

int ft_queue_time_end(ft_ctx* c, float* ms) {
    if (!c || !ms) return FT_E_INVALID;
    FT_HIP(c, hipSetDevice(c->device));
    FT_HIP(c, hipEventRecord(c->region_b, enq(c)));
    FT_HIP(c, hipEventSynchronize(c->region_b));
    FT_HIP(c, hipEventElapsedTime(ms, c->region_a, c->region_b));
    return FT_OK;
}

}  // extern "C"
