// Error behaviour of TrackKLT::feed_monocular / feed_stereo when a backend call fails: the code reaches the caller
// (it is not turned into a "tracking lost" frame) and the tracker's state stays consistent.  The backend is the CPU
// checker behind a wrapper that fails one chosen call.
//
//  * a failure while the frame is ingested (stage_frame, pyramid) leaves points, ids and the last frame as they were:
//    feeding the same frame again afterwards gives what an undisturbed run gives;
//  * a failure of the tracking launch or of the wait drops the tracks, and the next feed starts like a first frame.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "FeatureTrackerOptions.h"
#include "track/TrackKLT.h"

using namespace feature_tracker;

FrontendBackend* ftk_make_backend(int device, const ft_limits& limits, void** device_ctx);

extern "C" {
void fts_canvas(uint8_t* out, int cw, int ch, uint32_t seed, int n_waves, int n_shapes, int threads);
void fts_frame(const uint8_t* canvas, int cw, int ch, uint8_t* out, int w, int h, size_t ostride, double ccx, double ccy, double theta,
               double sigma, uint32_t noise_seed, int threads);
}

static int checks = 0;
#define CHECK(c)                                                                 \
    do {                                                                         \
        checks++;                                                                \
        if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } \
    } while (0)

namespace {

enum Call { NONE, STAGE, PYRAMID, LK, SYNC, FAST };

class FailingBackend : public FrontendBackend {
public:
    explicit FailingBackend(FrontendBackend* inner) : in(inner) {}
    Call fail = NONE;
    int code = FT_E_HIP;
    int fired = 0;
    int trip(Call c) { if (fail == c) { fired++; return code; } return FT_OK; }

    int upload(int s, int l, const uint8_t* h, int w, int hh, size_t st) override { int r = trip(STAGE); return r ? r : in->upload(s, l, h, w, hh, st); }
    int set_frames_device(int n, const int* s, const int* l, const void* const* d, int w, int h, size_t st) override { return in->set_frames_device(n, s, l, d, w, h, st); }
    int set_frames_device_pyramid(int n, const int* s, const int* l, const void* const* d, int w, int h, size_t st, int ml, int ww, int wh, int* a) override {
        return in->set_frames_device_pyramid(n, s, l, d, w, h, st, ml, ww, wh, a);
    }
    int stage_frame(int s, int l, const uint8_t* h, int w, int hh, size_t st) override { int r = trip(STAGE); return r ? r : in->stage_frame(s, l, h, w, hh, st); }
    int set_frames_staged_pyramid(int n, const int* s, const int* l, int w, int h, int ml, int ww, int wh, int* a) override {
        int r = trip(PYRAMID);
        return r ? r : in->set_frames_staged_pyramid(n, s, l, w, h, ml, ww, wh, a);
    }
    int build_pyramid(int n, const int* s, const int* l, int ml, int ww, int wh, int g, int* a) override { int r = trip(PYRAMID); return r ? r : in->build_pyramid(n, s, l, ml, ww, wh, g, a); }
    int lk_track_async(int n, const ft_lk_job* j, const ft_lk_params* p) override { int r = trip(LK); return r ? r : in->lk_track_async(n, j, p); }
    int fast_grid_async(int n, const int* s, const int* l, int t, int gx, int gy, int pc, ft_keypoint* o, int cap, int* no, int* cc) override {
        int r = trip(FAST);
        return r ? r : in->fast_grid_async(n, s, l, t, gx, gy, pc, o, cap, no, cc);
    }
    int compute_descriptors(int s, int l, const float* xy, int n, uint8_t* d, uint8_t* v) override { return in->compute_descriptors(s, l, xy, n, d, v); }
    int match_descriptors(const uint8_t* d0, int n0, const uint8_t* d1, int n1, float r, ft_match* o, int cap, int* no) override {
        return in->match_descriptors(d0, n0, d1, n1, r, o, cap, no);
    }
    int reject_outliers(int n, const ft_match_set* s, int h, unsigned seed) override { return in->reject_outliers(n, s, h, seed); }
    void homography(const float* K4, const float* R, float* H) override { in->homography(K4, R, H); }
    int sync() override { int r = trip(SYNC); return r ? r : in->sync(); }
    int fence(uint64_t* t) override { return in->fence(t); }
    int wait(uint64_t t) override { int r = trip(SYNC); return r ? r : in->wait(t); }
    const char* name() const override { return "failing"; }

private:
    std::unique_ptr<FrontendBackend> in;
};

struct Rig {
    std::unique_ptr<FailingBackend> be;
    std::unique_ptr<TrackKLT> trk;
};

Rig make_rig(int W, int H) {
    FeatureTrackerOptions o;
    o.num_pts = 80; o.grid_x = 4; o.grid_y = 3; o.min_px_dist = 8; o.win_size = 15; o.pyr_levels = 2;
    ft_limits L;
    L.max_width = W; L.max_height = H; L.max_levels = 3; L.n_streams = 2; L.n_slots = 2; L.max_points = 2 * (80 + 12 + 64);
    L.max_grid_cells = 36; L.max_per_cell = 80 / 12 + 1;
    Rig r;
    r.be.reset(new FailingBackend(ftk_make_backend(0, L, nullptr)));
    r.trk.reset(new TrackKLT(r.be.get(), o, 0, 0));
    std::map<size_t, Calib8> K;
    std::map<size_t, Mat4d> E;
    for (size_t c = 0; c < 2; c++) {
        K[c] = {{150., 150., W / 2., H / 2., 0., 0., 0., 0.}};
        Mat4d e{};
        for (int i = 0; i < 4; i++) e[5 * i] = 1.;
        E[c] = e;
    }
    r.trk->set_calibration(K, E);
    return r;
}

struct Snapshot {
    std::vector<size_t> ids[2];
    std::vector<KeyPoint> pts[2];
    bool operator==(const Snapshot& o) const {
        for (int c = 0; c < 2; c++) {
            if (ids[c] != o.ids[c] || pts[c].size() != o.pts[c].size()) return false;
            for (size_t i = 0; i < pts[c].size(); i++)
                if (pts[c][i].pt.x != o.pts[c][i].pt.x || pts[c][i].pt.y != o.pts[c][i].pt.y) return false;
        }
        return true;
    }
};

Snapshot snap(TrackKLT& t) {
    Snapshot s;
    for (int c = 0; c < 2; c++) { s.ids[c] = t.get_last_ids((size_t)c); s.pts[c] = t.get_last_points((size_t)c); }
    return s;
}

}  // namespace

int main() {
    const int W = 240, H = 160, F = 8;
    std::vector<std::vector<uint8_t>> left((size_t)F, std::vector<uint8_t>((size_t)W * H)), right = left;
    {
        std::vector<uint8_t> canvas((size_t)2 * W * 2 * H);
        fts_canvas(canvas.data(), 2 * W, 2 * H, 77, 32, 200, 2);
        for (int f = 0; f < F; f++) {
            fts_frame(canvas.data(), 2 * W, 2 * H, left[(size_t)f].data(), W, H, W, W + 1.7 * f, H + 0.9 * f, 0.003 * f, 1.0, 10 + f, 2);
            fts_frame(canvas.data(), 2 * W, 2 * H, right[(size_t)f].data(), W, H, W, W + 1.7 * f + 3.0, H + 0.9 * f, 0.003 * f, 1.0, 50 + f, 2);
        }
    }
    auto L = [&](int f) { return Image8(left[(size_t)f].data(), W, H, (size_t)W); };
    auto R = [&](int f) { return Image8(right[(size_t)f].data(), W, H, (size_t)W); };

    // ---- monocular: undisturbed run ----
    std::vector<Snapshot> good;
    {
        Rig r = make_rig(W, H);
        for (int f = 0; f < F; f++) { CHECK(r.trk->feed_monocular(0.05 * f, L(f), 0) == FT_OK); good.push_back(snap(*r.trk)); }
        CHECK(good.back().ids[0].size() > 20);
    }
    // ---- monocular: ingest failures change nothing ----
    for (Call c : {STAGE, PYRAMID}) {
        Rig r = make_rig(W, H);
        for (int f = 0; f < 4; f++) CHECK(r.trk->feed_monocular(0.05 * f, L(f), 0) == FT_OK);
        const Snapshot before = snap(*r.trk);
        r.be->fail = c; r.be->code = c == STAGE ? FT_E_LIMIT : FT_E_HIP;
        CHECK(r.trk->feed_monocular(0.05 * 4, L(4), 0) == r.be->code);
        CHECK(r.be->fired == 1);
        CHECK(snap(*r.trk) == before);
        r.be->fail = NONE;
        for (int f = 4; f < F; f++) { CHECK(r.trk->feed_monocular(0.05 * f, L(f), 0) == FT_OK); CHECK(snap(*r.trk) == good[(size_t)f]); }
    }
    // ---- monocular: tracking launch / wait failures reach the caller and restart the tracker ----
    for (Call c : {LK, SYNC, FAST}) {
        Rig r = make_rig(W, H);
        for (int f = 0; f < 4; f++) CHECK(r.trk->feed_monocular(0.05 * f, L(f), 0) == FT_OK);
        r.be->fail = c; r.be->code = FT_E_HIP;
        CHECK(r.trk->feed_monocular(0.05 * 4, L(4), 0) == FT_E_HIP);
        CHECK(r.be->fired >= 1);
        CHECK(r.trk->get_last_points(0).empty() && r.trk->get_last_ids(0).empty());
        r.be->fail = NONE;
        CHECK(r.trk->feed_monocular(0.05 * 5, L(5), 0) == FT_OK);            // detects like a first frame
        const Snapshot s5 = snap(*r.trk);
        CHECK(s5.ids[0].size() > 20);
        CHECK(r.trk->feed_monocular(0.05 * 6, L(6), 0) == FT_OK);            // and tracks from there
        const Snapshot s6 = snap(*r.trk);
        size_t kept = 0;
        for (size_t id : s6.ids[0]) for (size_t id5 : s5.ids[0]) kept += id == id5;
        CHECK(kept > 20);
    }

    // ---- stereo ----
    std::vector<Snapshot> good_st;
    {
        Rig r = make_rig(W, H);
        for (int f = 0; f < F; f++) { CHECK(r.trk->feed_stereo(0.05 * f, L(f), R(f), 0, 1) == FT_OK); good_st.push_back(snap(*r.trk)); }
        CHECK(good_st.back().ids[0].size() > 20 && good_st.back().ids[1].size() > 20);
    }
    for (Call c : {STAGE, PYRAMID}) {
        Rig r = make_rig(W, H);
        for (int f = 0; f < 4; f++) CHECK(r.trk->feed_stereo(0.05 * f, L(f), R(f), 0, 1) == FT_OK);
        const Snapshot before = snap(*r.trk);
        r.be->fail = c; r.be->code = FT_E_LIMIT;
        CHECK(r.trk->feed_stereo(0.05 * 4, L(4), R(4), 0, 1) == FT_E_LIMIT);
        CHECK(snap(*r.trk) == before);
        r.be->fail = NONE;
        for (int f = 4; f < F; f++) { CHECK(r.trk->feed_stereo(0.05 * f, L(f), R(f), 0, 1) == FT_OK); CHECK(snap(*r.trk) == good_st[(size_t)f]); }
    }
    for (Call c : {LK, SYNC}) {
        Rig r = make_rig(W, H);
        for (int f = 0; f < 4; f++) CHECK(r.trk->feed_stereo(0.05 * f, L(f), R(f), 0, 1) == FT_OK);
        r.be->fail = c; r.be->code = FT_E_HIP;
        CHECK(r.trk->feed_stereo(0.05 * 4, L(4), R(4), 0, 1) == FT_E_HIP);
        for (int cam = 0; cam < 2; cam++) CHECK(r.trk->get_last_points((size_t)cam).empty() && r.trk->get_last_ids((size_t)cam).empty());
        r.be->fail = NONE;
        CHECK(r.trk->feed_stereo(0.05 * 5, L(5), R(5), 0, 1) == FT_OK);
        CHECK(r.trk->get_last_points(0).size() > 20 && r.trk->get_last_points(1).size() > 20);
        CHECK(r.trk->feed_stereo(0.05 * 6, L(6), R(6), 0, 1) == FT_OK);
        CHECK(r.trk->get_last_points(0).size() > 20);
    }
    std::printf("test_feed_errors: %d checks passed\n", checks);
    return 0;
}
