// Descriptor-based stereo tracker: orchestration after src/track/TrackDescriptor.cpp:27-306 on the batched device calls.
#include "TrackDescriptor.h"

#include <algorithm>
#include <cstring>

namespace feature_tracker {

int TrackDescriptor::extract(const DeviceImage& img, std::vector<KeyPoint>& pts, Desc& desc) {
    std::vector<KeyPoint> ext;
    int rc = FT_OK;
    if (Grider_FAST::cells_for(img.cols, img.rows, grid_x, grid_y) >= 1 &&
        (rc = Grider_FAST::perform_griding(img, ext, num_features, grid_x, grid_y, threshold, true)) != FT_OK)   // :205-208
        return rc;
    const size_t n = ext.size();
    std::vector<float> xy(2 * n);
    for (size_t i = 0; i < n; i++) { xy[2 * i] = ext[i].pt.x; xy[2 * i + 1] = ext[i].pt.y; }
    Desc all(32 * n);
    std::vector<uint8_t> valid(n);
    if (n && (rc = be->compute_descriptors(img.stream, img.slot, xy.data(), (int)n, all.data(), valid.data())) != FT_OK) return rc;   // :218-219
    pts.clear(); desc.clear();
    for (size_t i = 0; i < n; i++) {
        if (!valid[i]) continue;
        pts.push_back(ext[i]);
        desc.insert(desc.end(), all.begin() + 32 * i, all.begin() + 32 * (i + 1));
    }
    return FT_OK;
}

int TrackDescriptor::robust_match(const std::vector<KeyPoint>& pts0, const std::vector<KeyPoint>& pts1, const Desc& desc0, const Desc& desc1,
                                  size_t id0, size_t id1, std::vector<ft_match>& matches) {
    matches.clear();
    const int n0 = (int)pts0.size(), n1 = (int)pts1.size();
    std::vector<ft_match> good((size_t)std::max(n0, 1));
    int n_good = 0;
    // knnMatch both ways, ratio test, symmetry test (:254-265)
    const int rc = be->match_descriptors(desc0.data(), n0, desc1.data(), n1, (float)knn_ratio, good.data(), (int)good.size(), &n_good);
    if (rc != FT_OK) return rc;
    good.resize((size_t)std::min(n_good, (int)good.size()));
    if (good.size() < 10) return FT_OK;   // not enough points for RANSAC: no matches at all (:276-277)
    if (!ransac_pairs) {                  // :280-304 as written rejects nothing (identical or affinely related sets): see the header
        matches = good;
        return FT_OK;
    }
    std::vector<float> a(2 * good.size()), b(2 * good.size());
    std::vector<uint8_t> st(good.size(), 1);
    for (size_t i = 0; i < good.size(); i++) {
        a[2 * i] = pts0[(size_t)good[i].query].pt.x; a[2 * i + 1] = pts0[(size_t)good[i].query].pt.y;
        b[2 * i] = pts1[(size_t)good[i].train].pt.x; b[2 * i + 1] = pts1[(size_t)good[i].train].pt.y;
    }
    double c0[8], c1[8];
    const double ident[8] = {1., 1., 0., 0., 0., 0., 0., 0.};
    for (int i = 0; i < 8; i++) {
        c0[i] = camera_k.count(id0) ? (i < 4 ? camera_k.at(id0)[i] : camera_d.at(id0)[i - 4]) : ident[i];
        c1[i] = camera_k.count(id1) ? (i < 4 ? camera_k.at(id1)[i] : camera_d.at(id1)[i - 4]) : ident[i];
    }
    ft_match_set set;
    set.n = (int)good.size(); set.prev_xy = a.data(); set.next_xy = b.data(); set.status = st.data(); set.calib_prev = c0; set.calib_next = c1;
    const int rc2 = be->reject_outliers(1, &set, 0, pairs_fed);
    if (rc2 != FT_OK) return rc2;
    for (size_t i = 0; i < good.size(); i++)
        if (st[i] == 1) matches.push_back(good[i]);
    return FT_OK;
}

int TrackDescriptor::perform_detection_stereo(const DeviceImage& img0, const DeviceImage& img1, std::vector<KeyPoint>& pts0,
                                              std::vector<KeyPoint>& pts1, Desc& desc0, Desc& desc1, size_t cam_id0, size_t cam_id1,
                                              std::vector<size_t>& ids0, std::vector<size_t>& ids1) {
    std::vector<KeyPoint> pts0_ext, pts1_ext;
    Desc desc0_ext, desc1_ext;
    int rc;
    if ((rc = extract(img0, pts0_ext, desc0_ext)) != FT_OK) return rc;
    if ((rc = extract(img1, pts1_ext, desc1_ext)) != FT_OK) return rc;
    std::vector<ft_match> matches;
    if ((rc = robust_match(pts0_ext, pts1_ext, desc0_ext, desc1_ext, cam_id0, cam_id1, matches)) != FT_OK) return rc;   // :226-227
    for (const ft_match& m : matches) {   // :230-243
        pts0.push_back(pts0_ext[(size_t)m.query]);
        pts1.push_back(pts1_ext[(size_t)m.train]);
        desc0.insert(desc0.end(), desc0_ext.begin() + 32 * (size_t)m.query, desc0_ext.begin() + 32 * (size_t)(m.query + 1));
        desc1.insert(desc1.end(), desc1_ext.begin() + 32 * (size_t)m.train, desc1_ext.begin() + 32 * (size_t)(m.train + 1));
        const size_t temp = ++currid;
        ids0.push_back(temp);
        ids1.push_back(temp);
    }
    return FT_OK;
}

int TrackDescriptor::feed_stereo(double timestamp, const Image8& img_leftin, const Image8& img_rightin, size_t cam_id_left, size_t cam_id_right) {
    std::unique_ptr<std::lock_guard<std::mutex>> lck1, lck2;
    if (cam_id_left < mtx_feeds.size()) lck1.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam_id_left]));
    if (cam_id_right < mtx_feeds.size() && cam_id_right != cam_id_left) lck2.reset(new std::lock_guard<std::mutex>(mtx_feeds[cam_id_right]));

    // the images go to slot 0 of the two cameras' streams (the tracker keeps descriptors, not images, between frames)
    const int sl = stream0 + (int)cam_id_left, sr = stream0 + (int)cam_id_right;
    int rc;
    if ((rc = be->upload(sl, 0, img_leftin.data, img_leftin.cols, img_leftin.rows, img_leftin.step)) != FT_OK) return rc;
    if ((rc = be->upload(sr, 0, img_rightin.data, img_rightin.cols, img_rightin.rows, img_rightin.step)) != FT_OK) return rc;
    const DeviceImage img_left{be, sl, 0, img_leftin.cols, img_leftin.rows}, img_right{be, sr, 0, img_rightin.cols, img_rightin.rows};
    pairs_fed++;

    // everything new is built on the side and swapped in at the end, so that an error leaves the tracker as it was
    std::vector<KeyPoint> pts_left_new, pts_right_new;
    Desc desc_left_new, desc_right_new;
    std::vector<size_t> ids_left_new, ids_right_new;
    const size_t currid_before = currid;
    if ((rc = perform_detection_stereo(img_left, img_right, pts_left_new, pts_right_new, desc_left_new, desc_right_new, cam_id_left,
                                       cam_id_right, ids_left_new, ids_right_new)) != FT_OK) {
        currid = currid_before;
        return rc;
    }
    auto& ptsL = pts_last[cam_id_left];
    auto& ptsR = pts_last[cam_id_right];
    auto& idsL = ids_last[cam_id_left];
    auto& idsR = ids_last[cam_id_right];

    // first frame, or tracking was lost: the detection is the new state (:43-52)
    if (ptsL.empty() || ptsR.empty()) {
        ptsL.swap(pts_left_new); ptsR.swap(pts_right_new);
        idsL.swap(ids_left_new); idsR.swap(ids_right_new);
        desc_last[cam_id_left].swap(desc_left_new); desc_last[cam_id_right].swap(desc_right_new);
        return FT_OK;
    }

    // temporal matching, last -> new, per camera (:70-81)
    std::vector<ft_match> matches_ll, matches_rr;
    if ((rc = robust_match(ptsL, pts_left_new, desc_last[cam_id_left], desc_left_new, cam_id_left, cam_id_left, matches_ll)) != FT_OK ||
        (rc = robust_match(ptsR, pts_right_new, desc_last[cam_id_right], desc_right_new, cam_id_right, cam_id_right, matches_rr)) != FT_OK) {
        currid = currid_before;
        return rc;
    }

    // a new stereo pair whose left and right points both match the same old feature takes that feature's id (:96-143);
    // the reference scans all matches for every new point and keeps the last hit
    std::vector<int> old_of_left(pts_left_new.size(), -1), old_of_right(pts_right_new.size(), -1);
    for (const ft_match& m : matches_ll) old_of_left[(size_t)m.train] = m.query;
    for (const ft_match& m : matches_rr) old_of_right[(size_t)m.train] = m.query;
    std::vector<size_t> good_ids_left(pts_left_new.size()), good_ids_right(pts_left_new.size());
    for (size_t i = 0; i < pts_left_new.size(); i++) {
        const int idll = old_of_left[i], idrr = old_of_right[i];
        if (idll != -1 && idrr != -1 && idsL[(size_t)idll] == idsR[(size_t)idrr]) {
            good_ids_left[i] = idsL[(size_t)idll];
            good_ids_right[i] = idsR[(size_t)idrr];
        } else {
            good_ids_left[i] = ids_left_new[i];
            good_ids_right[i] = ids_left_new[i];
        }
    }

    // feature database (:152-166)
    if (camera_k.count(cam_id_left) && camera_k.count(cam_id_right)) {
        for (size_t i = 0; i < pts_left_new.size(); i++) {
            const Point2f nl = undistort_point(pts_left_new[i].pt, cam_id_left), nr = undistort_point(pts_right_new[i].pt, cam_id_right);
            database->update_feature(good_ids_left[i], timestamp, cam_id_left, pts_left_new[i].pt.x, pts_left_new[i].pt.y, nl.x, nl.y);
            database->update_feature(good_ids_left[i], timestamp, cam_id_right, pts_right_new[i].pt.x, pts_right_new[i].pt.y, nr.x, nr.y);
        }
    }

    // move forward in time (:175-184)
    ptsL.swap(pts_left_new); ptsR.swap(pts_right_new);
    idsL.swap(good_ids_left); idsR.swap(good_ids_right);
    desc_last[cam_id_left].swap(desc_left_new); desc_last[cam_id_right].swap(desc_right_new);
    return FT_OK;
}

}  // namespace feature_tracker
