// Orchestration of the KLT front-end on top of the batched device kernels.  Follows the control flow of
// src/track/TrackKLT.cpp:27-432 (feed_stereo, perform_detection_stereo, perform_matching).
#include "TrackKLT.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace feature_tracker {

ft_lk_params TrackKLT::lk_params() const {
    ft_lk_params p;
    p.win_w = win_size; p.win_h = win_size;
    p.max_level = pyr_levels;
    p.max_iter = lk_max_iter;                // TermCriteria(COUNT + EPS, 15, 0.01), TrackKLT.cpp:305,:401
    p.eps = lk_eps;
    p.min_eig = lk_min_eig;
    // the residual is not computed: perform_matching never looks at it (TrackKLT.cpp:399-402)
    p.flags = FT_LK_USE_INITIAL_FLOW | FT_LK_NO_ERROR | (forward_backward ? FT_LK_FORWARD_BACKWARD : 0) |
              (reject_outliers ? FT_LK_REJECT_OUTLIERS : 0);
    p.fb_thresh = fb_thresh;
    p.ransac_hypotheses = ransac_hypotheses;     // TrackKLT.cpp:405-424
    p.ransac_seed = frames_fed;
    p.stereo_threshold = stereo_threshold;
    return p;
}

void TrackKLT::drop_tracks(std::vector<KeyPoint>& pts, std::vector<size_t>& ids) const {
    if (!(drop_fraction > 0.)) return;
    const uint64_t limit = drop_fraction >= 1. ? ~0ull : (uint64_t)(drop_fraction * 18446744073709551616.0);
    size_t o = 0;
    for (size_t i = 0; i < pts.size(); i++) {
        uint64_t z = (uint64_t)ids[i] * 0x9E3779B97F4A7C15ull + (uint64_t)frames_fed * 0xD1B54A32D192ED03ull;   // splitmix64 finalizer
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (z < limit) continue;
        if (o != i) { pts[o] = pts[i]; ids[o] = ids[i]; }
        o++;
    }
    pts.resize(o); ids.resize(o);
}

void TrackKLT::filter_by_occupancy(Occupancy& occ, std::vector<KeyPoint>& pts, std::vector<size_t>& ids) {
    size_t o = 0;
    for (size_t i = 0; i < pts.size(); i++) {
        if (!occ.try_take(pts[i].pt)) continue;   // too close to an earlier point: drop it (TrackKLT.cpp:238-242)
        if (o != i) { pts[o] = pts[i]; ids[o] = ids[i]; }
        o++;
    }
    pts.resize(o); ids.resize(o);
}

void TrackKLT::calib8_of(size_t cam_id, double* out8) const {
    const auto k = camera_k.find(cam_id);
    const auto d = camera_d.find(cam_id);
    const double ident[4] = {1., 1., 0., 0.};
    for (int i = 0; i < 4; i++) {
        out8[i] = k != camera_k.end() ? k->second[i] : ident[i];
        out8[4 + i] = d != camera_d.end() ? d->second[i] : 0.;
    }
}

bool TrackKLT::stereo_Rt_of(size_t cam0, size_t cam1, double* out) const {
    const auto r0 = R_cam_imu.find(cam0), r1 = R_cam_imu.find(cam1);
    const auto t0 = t_cam_imu.find(cam0), t1 = t_cam_imu.find(cam1);
    if (r0 == R_cam_imu.end() || r1 == R_cam_imu.end() || t0 == t_cam_imu.end() || t1 == t_cam_imu.end()) return false;
    const Mat3d& A = r0->second;
    const Mat3d& B = r1->second;
    // R_cam0_cam1 = R_cam_imu(0)^T R_cam_imu(1);  t_cam0_cam1 = R_cam_imu(1)^T (t_cam_imu(0) - t_cam_imu(1))
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) out[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
    const double d[3] = {t0->second[0] - t1->second[0], t0->second[1] - t1->second[1], t0->second[2] - t1->second[2]};
    for (int r = 0; r < 3; r++) out[9 + r] = B[r] * d[0] + B[3 + r] * d[1] + B[6 + r] * d[2];
    return true;
}

bool TrackKLT::gyro_homography(size_t cam_id, double t_last, double t_cur, float* H) {
    if (!use_gyro_seed || !camera_k.count(cam_id)) return false;
    Mat3f R0, R1;
    const int used = propagator->integrate_by_msckfVio(t_last, t_cur, R0, R1, false);
    if (used <= 0) return false;
    const auto& K = camera_k.at(cam_id);
    const float K4[4] = {(float)K[0], (float)K[1], (float)K[2], (float)K[3]};
    be->homography(K4, (cam_id == 1 ? R1 : R0).data(), H);   // H = K R K^-1 (TrackMsckfVio.cpp:348-349)
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// monocular path
// ---------------------------------------------------------------------------------------------------------------

void TrackKLT::mono_begin(double timestamp, int cols, int rows, size_t cam_id, MonoPlan& plan) {
    for (StereoDet& D : stereo_cache)   // a stereo-path detection of this camera's stream does not survive monocular feeds
        if (D.stream == stream0 + (int)cam_id) D.valid = false;
    CamState& cs = cams[cam_id];
    // reset the plan.  The tracking arrays keep their contents as well as their capacity: mono_prepare sets their sizes and
    // every element below those sizes is written before it is read (by this thread or by the delivery of the device's
    // results), while clearing them first made each resize fill a large stream's arrays with zeros again.  The database
    // arrays are cleared: they come back from a writer thread's core, and the zero fill of the resize that follows takes
    // their lines over in one streaming pass instead of one miss per line in mono_finish (10 000 points: 22 against 87 us)
    plan.first = plan.run_lk = plan.reset = plan.use_H = plan.db_pending = false;
    plan.db_ids.clear(); plan.db_uvn.clear();
    plan.cam = cam_id;
    frames_fed++;
    drop_tracks(pts_last[cam_id], ids_last[cam_id]);
    plan.timestamp = timestamp;
    plan.stream = stream0 + (int)cam_id;
    plan.slot_last = cs.slot_last;
    plan.slot_cur = (cs.slot_last + 1) % slot_ring;   // slot_last = -1 before the first frame
    plan.cols = cols; plan.rows = rows;
}

int TrackKLT::mono_features_needed(const MonoPlan& plan) {
    const CamState& cs = cams[plan.cam];
    auto& pts = pts_last[plan.cam];
    if (cs.slot_last < 0 || pts.empty()) return 0;   // first frame: detection happens on the current frame
    // the occupancy filter can only shrink the set, so count after applying it
    Occupancy occ(cs.cols, cs.rows, min_px_dist);
    filter_by_occupancy(occ, pts, ids_last[plan.cam]);
    return num_features - (int)pts.size();
}

void TrackKLT::perform_detection_monocular(int cols, int rows, const Detection& det, std::vector<KeyPoint>& pts0, std::vector<size_t>& ids0) {
    Occupancy& occ = occ_scratch;
    occ.reset(cols, rows, min_px_dist);
    filter_by_occupancy(occ, pts0, ids0);
    const int num_featsneeded = num_features - (int)pts0.size();
    if (num_featsneeded > 1 && det.valid) {                      // "> 1", as TrackKLT.cpp:271
        // first num_needed/(grid_x*grid_y)+1 entries of every cell of the ranked detection (Grider_FAST.h:88,:118-130)
        const int k = std::min(num_featsneeded / (grid_x * grid_y) + 1, det.per_cell);
        int off = 0;
        for (int c = 0; c < det.cells; c++) {
            const int kept = std::min(det.cell_counts[c], det.per_cell);
            for (int r = 0; r < kept && r < k && off + r < det.n; r++) {
                const KeyPoint kpt = Grider_FAST::to_keypoint(det.kps[off + r]);
                if (!occ.try_take(kpt.pt)) continue;             // TrackKLT.cpp:280-288
                pts0.push_back(kpt);
                ids0.push_back(++currid);
            }
            off += kept;
        }
    }
}

void TrackKLT::mono_prepare(MonoPlan& plan, const Detection& last_det, ft_lk_job& job) {
    CamState& cs = cams[plan.cam];
    auto& pts = pts_last[plan.cam];
    auto& ids = ids_last[plan.cam];
    memset(&job, 0, sizeof(job));
    if (cs.slot_last < 0 || pts.empty()) {   // TrackKLT.cpp:63-72
        plan.first = true;
        return;
    }
    // top the previous frame up so that a constant number of features is tracked (TrackKLT.cpp:76-78)
    perform_detection_monocular(cs.cols, cs.rows, last_det, pts, ids);
    if (pts.size() < 10) {                   // not enough points: all-zero mask (TrackKLT.cpp:392-396)
        plan.reset = true;
        return;
    }
    const size_t n = pts.size();
    plan.prev_xy.resize(2 * n); plan.next_xy.resize(2 * n); plan.err.resize(n); plan.status.resize(n);
    for (size_t i = 0; i < n; i++) {
        plan.prev_xy[2 * i] = pts[i].pt.x; plan.prev_xy[2 * i + 1] = pts[i].pt.y;
    }
    plan.next_xy = plan.prev_xy;             // initial guess = previous position (TrackKLT.cpp:87-88)
    plan.use_H = gyro_homography(plan.cam, cs.t_last, plan.timestamp, plan.H);
    plan.run_lk = true;
    job.stream_prev = plan.stream; job.slot_prev = cs.slot_last;
    job.stream_next = plan.stream; job.slot_next = plan.slot_cur;
    job.n = (int)n;
    job.prev_xy = plan.prev_xy.data(); job.next_xy = plan.next_xy.data();
    job.status = plan.status.data(); job.err = plan.err.data();
    job.H = plan.use_H ? plan.H : nullptr;
    job.iters = nullptr;
    if (reject_outliers) {
        calib8_of(plan.cam, plan.calib8);
        job.calib_prev = plan.calib8; job.calib_next = plan.calib8;
    }
}

void TrackKLT::mono_finish(MonoPlan& plan, const Detection& cur_det) {
    CamState& cs = cams[plan.cam];
    auto& pts = pts_last[plan.cam];
    auto& ids = ids_last[plan.cam];
    if (plan.first) {
        pts.clear(); ids.clear();
        perform_detection_monocular(plan.cols, plan.rows, cur_det, pts, ids);
    } else if (plan.reset || !plan.run_lk) {
        pts.clear(); ids.clear();            // lost tracking: re-detect on the next frame (TrackKLT.cpp:118-129)
    } else {
        // keep the successfully tracked points, compacting in place, and note their raw pixel positions for the
        // database; the undistortion of those positions happens in mono_commit, off the tracking path
        const bool to_db = camera_k.count(plan.cam) != 0;    // TrackKLT.cpp:187-192
        if (to_db) { plan.db_uvn.resize(4 * pts.size()); plan.db_ids.resize(pts.size()); }
        size_t o = 0;
        for (size_t i = 0; i < pts.size(); i++) {
            const float x = plan.next_xy[2 * i], y = plan.next_xy[2 * i + 1];
            if (x < 0 || y < 0) continue;    // TrackKLT.cpp:138
            if (!plan.status[i]) continue;
            if (o != i) { pts[o] = pts[i]; ids[o] = ids[i]; }
            pts[o].pt = Point2f(x, y);
            if (to_db) { plan.db_uvn[4 * o] = x; plan.db_uvn[4 * o + 1] = y; plan.db_ids[o] = ids[o]; }
            o++;
        }
        pts.resize(o); ids.resize(o);
        if (to_db) {
            plan.db_uvn.resize(4 * o); plan.db_ids.resize(o);
            plan.db_pending = true;
        }
    }
    // keep the ranked detection of this frame for the next call
    set_cached_detection(cur_det);
    cs.slot_last = plan.slot_cur;
    cs.cols = plan.cols; cs.rows = plan.rows;
    cs.t_last = plan.timestamp;
}

void TrackKLT::set_cached_detection(const Detection& det) {
    cache_valid = det.valid;
    if (det.valid) {
        cache_kps.assign(det.kps, det.kps + det.n);
        cache_cell_counts.assign(det.cell_counts, det.cell_counts + det.cells);
        cache_per_cell = det.per_cell;
        cache_n = det.n;
    }
}

void TrackKLT::mono_commit(MonoPlan& plan, int part, int parts) {
    if (!plan.db_pending) return;
    if (parts == 1) plan.db_pending = false;   // (threads that share a plan leave it alone: it is set anew for every frame)
    // raw pixel -> normalized coordinates (TrackKLT.cpp:187-192); calibration is read-only while frames are fed.
    // Every thread converts the observations it is going to write.
    const std::array<double, 4>& K = camera_k.at(plan.cam);
    const std::array<double, 4>& Dc = camera_d.at(plan.cam);
    const size_t n = plan.db_ids.size();
    if (Dc[0] == 0. && Dc[1] == 0. && Dc[2] == 0. && Dc[3] == 0.) {   // no distortion: the same arithmetic without the iteration
        const double ifx = 1. / K[0], ify = 1. / K[1];
        for (size_t i = 0; i < n; i++) {
            if (parts > 1 && database->owner_of(plan.db_ids[i], parts) != part) continue;
            plan.db_uvn[4 * i + 2] = (float)(((double)plan.db_uvn[4 * i] - K[2]) * ifx);
            plan.db_uvn[4 * i + 3] = (float)(((double)plan.db_uvn[4 * i + 1] - K[3]) * ify);
        }
    } else {
        for (size_t i = 0; i < n; i++) {
            if (parts > 1 && database->owner_of(plan.db_ids[i], parts) != part) continue;
            const Point2f npt = undistort_point_brown(Point2f(plan.db_uvn[4 * i], plan.db_uvn[4 * i + 1]), K, Dc);
            plan.db_uvn[4 * i + 2] = npt.x; plan.db_uvn[4 * i + 3] = npt.y;
        }
    }
    database->update_features(n, plan.db_ids.data(), plan.timestamp, plan.cam, plan.db_uvn.data(), part, parts);
}

int TrackKLT::abort_feed(int rc, size_t cam_a, size_t cam_b) {
    (void)be->sync();   // nothing may be left waiting to be written into this object's buffers
    spec_pending = false;
    for (StereoDet& D : stereo_cache) D.valid = false;
    cache_valid = false;
    const size_t cam_ids[2] = {cam_a, cam_b};
    for (size_t cam : cam_ids) {
        pts_last[cam].clear(); ids_last[cam].clear();
        cams[cam].slot_last = -1;
    }
    return rc;
}

int TrackKLT::feed_monocular(double timestamp, const Image8& img, size_t cam_id) {
    std::unique_ptr<std::lock_guard<std::mutex>> lck;
    if (cam_id < mtx_feeds.size()) lck.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam_id]));
    int rc;
    // through the pinned staging buffer; level 0 and the pyramid come out of one pass over it (no copy command).
    // The slot written is not the one holding the last pyramid, so a failure here leaves the tracker untouched.
    const int stream_cur = stream0 + (int)cam_id, slot_cur = next_slot(cam_id);
    if ((rc = be->stage_frame(stream_cur, slot_cur, img.data, img.cols, img.rows, img.step)) != FT_OK) return rc;
    if ((rc = be->set_frames_staged_pyramid(1, &stream_cur, &slot_cur, img.cols, img.rows, pyr_levels, win_size, win_size, nullptr)) != FT_OK) return rc;
    if ((rc = be->flush()) != FT_OK) return rc;   // no detector call rides on this pyramid: launch all of it now
    MonoPlan plan;
    mono_begin(timestamp, img.cols, img.rows, cam_id, plan);
    const ft_lk_params prm = lk_params();

    const int cells = Grider_FAST::cells_for(img.cols, img.rows, grid_x, grid_y);
    const int kmax = per_cell_max();
    Detection last_det, cur_det;
    std::vector<ft_keypoint> last_buf;
    std::vector<int> last_cells;
    int last_n = 0, cur_n = 0;
    const bool have_last = cams[cam_id].slot_last >= 0 && !pts_last[cam_id].empty();
    if (speculative_detect) {
        if (have_last && cache_valid) {
            last_det.valid = true; last_det.per_cell = cache_per_cell; last_det.cells = (int)cache_cell_counts.size();
            last_det.kps = cache_kps.data(); last_det.n = cache_n; last_det.cell_counts = cache_cell_counts.data();
        }
    } else if (have_last) {
        const int need = mono_features_needed(plan);
        if (need > 1 && cells > 0) {         // detector on the previous frame, only when needed (TrackKLT.cpp:271-275)
            const int per_cell = need / (grid_x * grid_y) + 1;
            last_buf.resize((size_t)cells * per_cell); last_cells.resize(cells);
            if ((rc = be->fast_grid_async(1, &plan.stream, &plan.slot_last, threshold, grid_x, grid_y, per_cell, last_buf.data(),
                                          (int)last_buf.size(), &last_n, last_cells.data())) != FT_OK || (rc = be->sync()) != FT_OK)
                return abort_feed(rc, cam_id, cam_id);
            last_det.valid = true; last_det.per_cell = per_cell; last_det.cells = cells;
            last_det.kps = last_buf.data(); last_det.n = last_n; last_det.cell_counts = last_cells.data();
        }
    }
    ft_lk_job job;
    mono_prepare(plan, last_det, job);
    bool want_cur_det = (speculative_detect || plan.first) && cells > 0;
    if (want_cur_det) {
        det_buf.resize((size_t)cells * kmax); det_cells.resize(cells);
        if ((rc = be->fast_grid_async(1, &plan.stream, &plan.slot_cur, threshold, grid_x, grid_y, kmax, det_buf.data(),
                                      (int)det_buf.size(), &cur_n, det_cells.data())) != FT_OK)
            return abort_feed(rc, cam_id, cam_id);
    }
    if (plan.run_lk && (rc = be->lk_track_async(1, &job, &prm)) != FT_OK) return abort_feed(rc, cam_id, cam_id);
    if ((rc = be->sync()) != FT_OK) return abort_feed(rc, cam_id, cam_id);
    if (want_cur_det) {
        cur_det.valid = true; cur_det.per_cell = kmax; cur_det.cells = cells;
        cur_det.kps = det_buf.data(); cur_det.n = cur_n; cur_det.cell_counts = det_cells.data();
    }
    mono_finish(plan, cur_det);
    mono_commit(plan);
    return FT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// stereo path
// ---------------------------------------------------------------------------------------------------------------

int TrackKLT::perform_matching(std::vector<MatchJob>& jobs, const std::function<void()>& after_launch) {
    const ft_lk_params prm = lk_params();
    struct Buf { std::vector<float> p0, p1, err; std::vector<uint8_t> st; double c0[8], c1[8]; };
    std::vector<Buf> bufs(jobs.size());
    std::vector<ft_lk_job> lk;
    std::vector<size_t> lk_owner;
    for (size_t j = 0; j < jobs.size(); j++) {
        MatchJob& M = jobs[j];
        M.mask_out->clear();
        const size_t n = M.kpts0->size();
        if (n == 0 || M.kpts1->size() != n) continue;          // TrackKLT.cpp:377-381: mask stays empty
        if (n < 10) { M.mask_out->assign(n, 0); continue; }    // TrackKLT.cpp:392-396
        Buf& B = bufs[j];
        B.p0.resize(2 * n); B.p1.resize(2 * n); B.err.resize(n); B.st.resize(n);
        for (size_t i = 0; i < n; i++) {
            B.p0[2 * i] = (*M.kpts0)[i].pt.x; B.p0[2 * i + 1] = (*M.kpts0)[i].pt.y;
            B.p1[2 * i] = (*M.kpts1)[i].pt.x; B.p1[2 * i + 1] = (*M.kpts1)[i].pt.y;
        }
        ft_lk_job J;
        memset(&J, 0, sizeof(J));
        J.stream_prev = M.img0.stream; J.slot_prev = M.img0.slot;
        J.stream_next = M.img1.stream; J.slot_next = M.img1.slot;
        J.n = (int)n; J.prev_xy = B.p0.data(); J.next_xy = B.p1.data(); J.status = B.st.data(); J.err = B.err.data();
        J.H = M.H;
        if (reject_outliers) {
            calib8_of(M.id0, B.c0); calib8_of(M.id1, B.c1);
            J.calib_prev = B.c0; J.calib_next = B.c1;
        }
        lk.push_back(J);
        lk_owner.push_back(j);
    }
    if (lk.empty()) return FT_OK;
    int rc;
    if ((rc = be->lk_track_async((int)lk.size(), lk.data(), &prm)) != FT_OK) return rc;
    // wait for the tracking results only: what the caller queues behind them keeps running after the wait
    uint64_t token = 0;
    const bool fenced = be->fence(&token) == FT_OK;
    if (after_launch) after_launch();
    if ((rc = fenced ? be->wait(token) : be->sync()) != FT_OK) return rc;
    for (size_t q = 0; q < lk.size(); q++) {
        MatchJob& M = jobs[lk_owner[q]];
        Buf& B = bufs[lk_owner[q]];
        const size_t n = M.kpts0->size();
        // with reject_outliers the status already carries the RANSAC verdict (TrackKLT.cpp:405-424 -> the device's
        // rejection kernel); otherwise it is the LK status alone
        M.mask_out->assign(B.st.begin(), B.st.end());
        for (size_t i = 0; i < n; i++) (*M.kpts1)[i].pt = Point2f(B.p1[2 * i], B.p1[2 * i + 1]);   // TrackKLT.cpp:427-430
    }
    return FT_OK;
}

int TrackKLT::promote_stereo_detection() {
    if (!spec_pending) return FT_OK;
    spec_pending = false;
    const int rc = be->sync();   // delivers the detector output if an earlier wait has not done so already
    const bool ok = rc == FT_OK;
    for (int i = 0; i < 2; i++) {
        StereoDet& D = stereo_cache[i];
        D.valid = ok;
        if (!ok) continue;
        D.stream = spec_stream[i]; D.slot = spec_slot[i]; D.cols = spec_cols; D.rows = spec_rows; D.per_cell = spec_per_cell; D.cells = spec_ncells; D.n = spec_n[i];
        D.kps.assign(spec_kps.begin() + (size_t)i * spec_cap, spec_kps.begin() + (size_t)i * spec_cap + spec_n[i]);
        D.cell_counts.assign(spec_cells.begin() + (size_t)i * spec_ncells, spec_cells.begin() + (size_t)(i + 1) * spec_ncells);
    }
    return rc;
}

int TrackKLT::grid_points(const DeviceImage& img, int num_needed, std::vector<KeyPoint>& out) {
    for (const StereoDet& D : stereo_cache) {
        if (!D.valid || D.stream != img.stream || D.slot != img.slot || D.cols != img.cols || D.rows != img.rows) continue;
        // first num_needed/(grid_x*grid_y)+1 entries of every cell of the ranked lists, cells in index order: what
        // Grider_FAST::perform_griding returns for this count (Grider_FAST.h:88,:118-130)
        const int k = num_needed / (grid_x * grid_y) + 1;
        if (k > D.per_cell) break;
        int off = 0;
        for (int c = 0; c < D.cells; c++) {
            const int kept = std::min(D.cell_counts[c], D.per_cell);
            for (int r = 0; r < kept && r < k && off + r < D.n; r++) out.push_back(Grider_FAST::to_keypoint(D.kps[off + r]));
            off += kept;
        }
        return FT_OK;
    }
    if (Grider_FAST::cells_for(img.cols, img.rows, grid_x, grid_y) < 1) return FT_OK;   // image smaller than the grid: nothing to find
    return Grider_FAST::perform_griding(img, out, num_needed, grid_x, grid_y, threshold, true);
}

int TrackKLT::perform_detection_stereo(const DeviceImage& img0, const DeviceImage& img1, std::vector<KeyPoint>& pts0,
                                       std::vector<KeyPoint>& pts1, std::vector<size_t>& ids0, std::vector<size_t>& ids1,
                                       size_t cam_id_left, size_t cam_id_right) {
    int rc;
    Occupancy grid0(img0.cols, img0.rows, min_px_dist), grid1(img1.cols, img1.rows, min_px_dist);
    filter_by_occupancy(grid0, pts0, ids0);
    filter_by_occupancy(grid1, pts1, ids1);

    // LEFT: extract, then try to find the new points in the right image (TrackKLT.cpp:266-341)
    const int num_featsneeded_0 = num_features - (int)pts0.size();
    if (num_featsneeded_0 > 1) {
        std::vector<KeyPoint> pts0_ext;
        if ((rc = grid_points(img0, num_featsneeded_0, pts0_ext)) != FT_OK) return rc;
        std::vector<KeyPoint> kpts0_new;
        for (const KeyPoint& kpt : pts0_ext) {
            if (grid0.taken(kpt.pt)) continue;
            kpts0_new.push_back(kpt);
            grid0.take(kpt.pt);
        }
        if (!kpts0_new.empty()) {
            // same-pixel initial guess in the right image (TrackKLT.cpp:290-295)
            const size_t n = kpts0_new.size();
            std::vector<float> p0(2 * n), p1(2 * n), err(n);
            std::vector<uint8_t> mask(n);
            for (size_t i = 0; i < n; i++) { p0[2 * i] = kpts0_new[i].pt.x; p0[2 * i + 1] = kpts0_new[i].pt.y; }
            p1 = p0;
            ft_lk_job J;
            memset(&J, 0, sizeof(J));
            J.stream_prev = img0.stream; J.slot_prev = img0.slot; J.stream_next = img1.stream; J.slot_next = img1.slot;
            J.n = (int)n; J.prev_xy = p0.data(); J.next_xy = p1.data(); J.status = mask.data(); J.err = err.data();
            ft_lk_params prm = lk_params();
            prm.flags = FT_LK_USE_INITIAL_FLOW | FT_LK_NO_ERROR;   // the stereo association has no backward pass in the reference
            // optional: guess from the stereo extrinsics and known-E test (TrackMsckfVio.cpp:218-301) instead of "same pixel"
            double c0[8], c1[8], Rt[12];
            if (stereo_extrinsic_seed && stereo_Rt_of(cam_id_left, cam_id_right, Rt)) {
                calib8_of(cam_id_left, c0); calib8_of(cam_id_right, c1);
                J.calib_prev = c0; J.calib_next = c1; J.stereo_Rt = Rt;
            }
            if ((rc = be->lk_track_async(1, &J, &prm)) != FT_OK || (rc = be->sync()) != FT_OK) return rc;
            for (size_t i = 0; i < n; i++) {
                const Point2f a(p0[2 * i], p0[2 * i + 1]), b(p1[2 * i], p1[2 * i + 1]);
                const bool oob0 = a.x < 0 || a.y < 0 || a.x >= img0.cols || a.y >= img0.rows;
                const bool oob1 = b.x < 0 || b.y < 0 || b.x >= img1.cols || b.y >= img1.rows;
                if (!oob0 && mask[i] == 1 && !oob1 && !grid1.taken(b)) {   // stereo feature (TrackKLT.cpp:318-328)
                    KeyPoint k0 = kpts0_new[i], k1 = kpts0_new[i];
                    k0.pt = a; k1.pt = b;
                    pts0.push_back(k0); pts1.push_back(k1);
                    const size_t id = ++currid;
                    ids0.push_back(id); ids1.push_back(id);
                } else if (!oob0) {                                        // left-only feature (TrackKLT.cpp:329-337)
                    KeyPoint k0 = kpts0_new[i];
                    k0.pt = a;
                    pts0.push_back(k0);
                    ids0.push_back(++currid);
                }
            }
        }
    }

    // RIGHT: extract right-only features (TrackKLT.cpp:343-365)
    const int num_featsneeded_1 = num_features - (int)pts1.size();
    if (num_featsneeded_1 > 1) {
        std::vector<KeyPoint> pts1_ext;
        if ((rc = grid_points(img1, num_featsneeded_1, pts1_ext)) != FT_OK) return rc;
        for (const KeyPoint& kpt : pts1_ext) {
            if (grid1.taken(kpt.pt)) continue;
            pts1.push_back(kpt);
            ids1.push_back(++currid);
            grid1.take(kpt.pt);
        }
    }
    return FT_OK;
}

int TrackKLT::feed_stereo(double timestamp, const Image8& img_leftin, const Image8& img_rightin, size_t cam_id_left, size_t cam_id_right) {
    std::unique_ptr<std::lock_guard<std::mutex>> lck1, lck2;
    if (cam_id_left < mtx_feeds.size()) lck1.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam_id_left]));
    if (cam_id_right < mtx_feeds.size() && cam_id_right != cam_id_left) lck2.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam_id_right]));

    CamState& cl = cams[cam_id_left];
    CamState& cr = cams[cam_id_right];
    const int sl = stream0 + (int)cam_id_left, sr = stream0 + (int)cam_id_right;
    const int cur_l = cl.slot_last == 0 ? 1 : 0, cur_r = cr.slot_last == 0 ? 1 : 0;

    // ingest both frames and build both pyramids in one batched launch per level (TrackKLT.cpp:45-58).  The slots
    // written are not the ones holding the last pair, so a failure in this part leaves the tracker as it was.
    const int streams[2] = {sl, sr}, slots[2] = {cur_l, cur_r};
    int rc;
    if (img_leftin.cols == img_rightin.cols && img_leftin.rows == img_rightin.rows && sl != sr) {
        // both frames through the pinned staging buffers; level 0 and the pyramids come out of one pass over them
        if ((rc = be->stage_frame(sl, cur_l, img_leftin.data, img_leftin.cols, img_leftin.rows, img_leftin.step)) != FT_OK) return rc;
        if ((rc = be->stage_frame(sr, cur_r, img_rightin.data, img_rightin.cols, img_rightin.rows, img_rightin.step)) != FT_OK) return rc;
        // the previous pair's detector ran while the frames were copied
        if ((rc = promote_stereo_detection()) != FT_OK) return abort_feed(rc, cam_id_left, cam_id_right);
        if ((rc = be->set_frames_staged_pyramid(2, streams, slots, img_leftin.cols, img_leftin.rows, pyr_levels, win_size, win_size, nullptr)) != FT_OK) return rc;
        if ((rc = be->flush()) != FT_OK) return rc;   // (the detector of this pair is queued behind its tracking)
    } else {
        if ((rc = promote_stereo_detection()) != FT_OK) return abort_feed(rc, cam_id_left, cam_id_right);
        if ((rc = be->upload(sl, cur_l, img_leftin.data, img_leftin.cols, img_leftin.rows, img_leftin.step)) != FT_OK) return rc;
        if ((rc = be->upload(sr, cur_r, img_rightin.data, img_rightin.cols, img_rightin.rows, img_rightin.step)) != FT_OK) return rc;
        if ((rc = be->build_pyramid(1, streams, slots, pyr_levels, win_size, win_size, 0, nullptr)) != FT_OK) return rc;
        if ((rc = be->build_pyramid(1, streams + 1, slots + 1, pyr_levels, win_size, win_size, 0, nullptr)) != FT_OK) return rc;
    }
    frames_fed++;   // seeds the outlier rejection's sample sequence
    drop_tracks(pts_last[cam_id_left], ids_last[cam_id_left]);
    if (cam_id_right != cam_id_left) drop_tracks(pts_last[cam_id_right], ids_last[cam_id_right]);
    const DeviceImage cur_left{be, sl, cur_l, img_leftin.cols, img_leftin.rows};
    const DeviceImage cur_right{be, sr, cur_r, img_rightin.cols, img_rightin.rows};
    // speculative detection: the ranked detector of both new frames, whose output is what topping this pair up at the
    // next call needs (the reference detects then, TrackKLT.cpp:76-78).  It is queued behind this call's last tracking
    // launch and collected at the next call, so it adds nothing to this call's latency.
    const int spec_cells_now = Grider_FAST::cells_for(img_leftin.cols, img_leftin.rows, grid_x, grid_y);
    const bool can_spec = speculative_detect && spec_cells_now > 0 && sl != sr && img_leftin.cols == img_rightin.cols &&
                          img_leftin.rows == img_rightin.rows;
    auto enqueue_spec = [&]() {
        if (!can_spec || spec_pending) return;
        spec_per_cell = per_cell_max(); spec_ncells = spec_cells_now; spec_cap = spec_cells_now * spec_per_cell;
        spec_kps.resize((size_t)2 * spec_cap); spec_n.assign(2, 0); spec_cells.assign((size_t)2 * spec_cells_now, 0);
        spec_stream[0] = sl; spec_stream[1] = sr; spec_slot[0] = cur_l; spec_slot[1] = cur_r;
        spec_cols = img_leftin.cols; spec_rows = img_leftin.rows;
        spec_pending = be->fast_grid_async(2, streams, slots, threshold, grid_x, grid_y, spec_per_cell, spec_kps.data(), spec_cap,
                                           spec_n.data(), spec_cells.data()) == FT_OK;
    };
    // the detector of this pair is queued at most once per call: a pair that was detected and promoted within this
    // call (first call, tracking lost) has its ranked lists in the cache already
    bool detected_now = false;
    auto roll = [&]() {
        cl.slot_last = cur_l; cl.cols = img_leftin.cols; cl.rows = img_leftin.rows; cl.t_last = timestamp;
        cr.slot_last = cur_r; cr.cols = img_rightin.cols; cr.rows = img_rightin.rows; cr.t_last = timestamp;
        if (!detected_now) enqueue_spec();   // (already queued behind the tracking launch on the usual path)
    };
    auto& ptsL = pts_last[cam_id_left];
    auto& ptsR = pts_last[cam_id_right];
    auto& idsL = ids_last[cam_id_left];
    auto& idsR = ids_last[cam_id_right];

    // first call, or tracking was lost: just extract (TrackKLT.cpp:63-72)
    if (ptsL.empty() || ptsR.empty() || cl.slot_last < 0 || cr.slot_last < 0) {
        enqueue_spec();
        detected_now = spec_pending;
        // this pair's own detection is needed at once
        if ((rc = promote_stereo_detection()) != FT_OK) return abort_feed(rc, cam_id_left, cam_id_right);
        if ((rc = perform_detection_stereo(cur_left, cur_right, ptsL, ptsR, idsL, idsR, cam_id_left, cam_id_right)) != FT_OK)
            return abort_feed(rc, cam_id_left, cam_id_right);
        roll();
        return FT_OK;
    }

    // top the previous pair up (TrackKLT.cpp:76-78)
    const DeviceImage last_left{be, sl, cl.slot_last, cl.cols, cl.rows};
    const DeviceImage last_right{be, sr, cr.slot_last, cr.cols, cr.rows};
    if ((rc = perform_detection_stereo(last_left, last_right, ptsL, ptsR, idsL, idsR, cam_id_left, cam_id_right)) != FT_OK)
        return abort_feed(rc, cam_id_left, cam_id_right);

    // temporal tracking of both cameras in one launch (TrackKLT.cpp:86-98)
    std::vector<uint8_t> mask_ll, mask_rr;
    std::vector<KeyPoint> pts_left_new = ptsL, pts_right_new = ptsR;
    float Hl[9], Hr[9];
    Mat3f R0, R1;
    bool seeded = false;
    if (use_gyro_seed && camera_k.count(cam_id_left) && camera_k.count(cam_id_right) &&
        propagator->integrate_by_msckfVio(cl.t_last, timestamp, R0, R1, false) > 0) {
        const auto& Kl = camera_k.at(cam_id_left);
        const auto& Kr = camera_k.at(cam_id_right);
        const float K4l[4] = {(float)Kl[0], (float)Kl[1], (float)Kl[2], (float)Kl[3]};
        const float K4r[4] = {(float)Kr[0], (float)Kr[1], (float)Kr[2], (float)Kr[3]};
        be->homography(K4l, R0.data(), Hl);
        be->homography(K4r, R1.data(), Hr);
        seeded = true;
    }
    std::vector<MatchJob> jobs(2);
    jobs[0] = MatchJob{last_left, cur_left, &ptsL, &pts_left_new, cam_id_left, cam_id_left, &mask_ll, seeded ? Hl : nullptr};
    jobs[1] = MatchJob{last_right, cur_right, &ptsR, &pts_right_new, cam_id_right, cam_id_right, &mask_rr, seeded ? Hr : nullptr};
    if ((rc = perform_matching(jobs, enqueue_spec)) != FT_OK) return abort_feed(rc, cam_id_left, cam_id_right);

    // not enough points: reset and re-detect next time (TrackKLT.cpp:118-129)
    if (mask_ll.empty() || mask_rr.empty()) {
        roll();
        ptsL.clear(); ptsR.clear(); idsL.clear(); idsR.clear();
        printf("[KLT-EXTRACTOR]: Failed to get enough points to do RANSAC, resetting.....\n");
        return FT_OK;
    }

    // merge: stereo tracks need both temporal tracks to succeed (TrackKLT.cpp:132-181)
    std::vector<KeyPoint> good_left, good_right;
    std::vector<size_t> good_ids_left, good_ids_right;
    std::unordered_map<size_t, size_t> right_index;
    for (size_t n = idsR.size(); n-- > 0;) right_index[idsR[n]] = n;   // first occurrence wins, as the linear search does
    std::unordered_map<size_t, char> right_added;
    for (size_t i = 0; i < pts_left_new.size(); i++) {
        if (pts_left_new[i].pt.x < 0 || pts_left_new[i].pt.y < 0) continue;
        auto it = right_index.find(idsL[i]);
        const bool found_right = it != right_index.end();
        const size_t index_right = found_right ? it->second : 0;
        if (mask_ll[i] && found_right && mask_rr[index_right]) {
            if (pts_right_new[index_right].pt.x < 0 || pts_right_new[index_right].pt.y < 0) continue;
            good_left.push_back(pts_left_new[i]);
            good_right.push_back(pts_right_new[index_right]);
            good_ids_left.push_back(idsL[i]);
            good_ids_right.push_back(idsR[index_right]);
            right_added[idsR[index_right]] = 1;
        } else if (mask_ll[i]) {
            good_left.push_back(pts_left_new[i]);
            good_ids_left.push_back(idsL[i]);
        }
    }
    for (size_t i = 0; i < pts_right_new.size(); i++) {
        if (pts_right_new[i].pt.x < 0 || pts_right_new[i].pt.y < 0) continue;
        if (mask_rr[i] && !right_added.count(idsR[i])) {
            good_right.push_back(pts_right_new[i]);
            good_ids_right.push_back(idsR[i]);
            right_added[idsR[i]] = 1;
        }
    }

    // feature database (TrackKLT.cpp:187-198)
    if (camera_k.count(cam_id_left) && camera_k.count(cam_id_right)) {
        for (size_t i = 0; i < good_left.size(); i++) {
            const Point2f n = undistort_point(good_left[i].pt, cam_id_left);
            database->update_feature(good_ids_left[i], timestamp, cam_id_left, good_left[i].pt.x, good_left[i].pt.y, n.x, n.y);
        }
        for (size_t i = 0; i < good_right.size(); i++) {
            const Point2f n = undistort_point(good_right[i].pt, cam_id_right);
            database->update_feature(good_ids_right[i], timestamp, cam_id_right, good_right[i].pt.x, good_right[i].pt.y, n.x, n.y);
        }
    }

    // move forward in time (TrackKLT.cpp:201-208)
    roll();
    ptsL.swap(good_left); ptsR.swap(good_right);
    idsL.swap(good_ids_left); idsR.swap(good_ids_right);
    return FT_OK;
}

}  // namespace feature_tracker
