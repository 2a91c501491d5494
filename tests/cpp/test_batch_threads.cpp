// Host threading of the batched tracker (group driver threads, mailboxes, database writer threads, pipelined feed)
// exercised from a plain C++ program so that ThreadSanitizer can watch it.  Linked with the CPU checker's backend
// (oracle/oracle_backend.cpp) instead of the device library: the host layer is the same code either way.
//
// Checks: a grouped, pipelined run equals a plain single-group run frame for frame (counts, ids, positions), and
// the sanitizer reports no data race.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ft_tracker.h"

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

struct Run {
    std::vector<long long> tracked;
    std::vector<std::vector<float>> xy;           // per stream, after the last frame
    std::vector<std::vector<uint64_t>> ids;
    std::vector<int> db;
    std::vector<std::vector<float>> tracks;       // per stream: every observation of every live feature, in id order
};

int main() {
    const int S = 8, W = 188, H = 120, F = 12;
    // frames of S streams: a drifting, slowly rotating view of a textured canvas
    std::vector<std::vector<uint8_t>> frames((size_t)S * F, std::vector<uint8_t>((size_t)W * H));
    for (int s = 0; s < S; s++) {
        std::vector<uint8_t> canvas((size_t)2 * W * 2 * H);
        fts_canvas(canvas.data(), 2 * W, 2 * H, 100 + s, 32, 120, 2);
        for (int f = 0; f < F; f++)
            fts_frame(canvas.data(), 2 * W, 2 * H, frames[(size_t)s * F + f].data(), W, H, W, W + 2.3 * f, H + 1.1 * f, 0.004 * f, 1.5, 7 * s + f, 2);
    }
    const double calib8[8] = {120., 120., W / 2., H / 2., 0., 0., 0., 0.};

    // n_streams <= S streams; num_pts >= 1024 makes the trackers split their feature databases by id, and with fewer
    // streams than writer threads the writers then share every frame's observations of a stream
    auto run = [&](int groups, bool pipelined, int n_streams = 8, int num_pts = 60) {
        const int S = n_streams;
        ftk_options o;
        ftk_default_options(&o);
        o.num_pts = num_pts; o.grid_x = 3; o.grid_y = 2; o.min_px_dist = num_pts > 1000 ? 2 : 6; o.win_size = 9; o.pyr_levels = 2;
        ftk_batch* b = nullptr;
        CHECK(ftk_batch_create_grouped(&b, 0, &o, S, W, H, groups, 0) == 0);
        for (int s = 0; s < S; s++) CHECK(ftk_batch_set_calibration(b, s, calib8, nullptr) == 0);
        Run r;
        long long total = 0;
        for (int f = 0; f < F; f++) {
            std::vector<double> ts((size_t)S, 0.05 * f);
            std::vector<const void*> ptr((size_t)S);
            for (int s = 0; s < S; s++) ptr[(size_t)s] = frames[(size_t)s * F + f].data();
            long long tr = 0, at = 0;
            if (pipelined) CHECK(ftk_batch_feed_pipelined(b, ts.data(), ptr.data(), 0, W, H, W, &tr, &at) == 0);
            else CHECK(ftk_batch_feed(b, ts.data(), ptr.data(), 0, W, H, W, &tr, &at) == 0);
            total += tr;
            if (!pipelined) r.tracked.push_back(tr);
        }
        if (pipelined) { long long tr = 0, at = 0; CHECK(ftk_batch_flush(b, &tr, &at) == 0); total += tr; }
        r.tracked.push_back(total);
        for (int s = 0; s < S; s++) {
            std::vector<float> xy(2 * 2048);
            std::vector<uint64_t> ids(2048);
            const int n = ftk_batch_get_points(b, s, xy.data(), ids.data(), 2048);
            CHECK(n >= 0 && n <= 2048);
            xy.resize(2 * (size_t)n); ids.resize((size_t)n);
            r.xy.push_back(xy); r.ids.push_back(ids);
            r.db.push_back(ftk_batch_database_size(b, s));
            std::vector<float> tr;
            for (uint64_t id : ids) {
                float uvn[4 * 64];
                double ts[64];
                const int m = ftk_batch_get_track(b, s, id, uvn, ts, 64);
                CHECK(m >= 0 && m <= 64);
                tr.push_back((float)m);
                for (int k = 0; k < m; k++) { tr.insert(tr.end(), uvn + 4 * k, uvn + 4 * k + 4); tr.push_back((float)ts[k]); }
            }
            r.tracks.push_back(tr);
        }
        ftk_batch_destroy(b);
        return r;
    };

    const Run plain = run(1, false);
    const Run piped = run(2, true);
    const Run piped4 = run(4, true);
    const Run piped1 = run(1, true);   // eight streams in one group: the last set's database writes are shared with the writers
    long long plain_total = 0;
    for (size_t i = 0; i + 1 < plain.tracked.size(); i++) plain_total += plain.tracked[i];
    CHECK(plain_total > 200);
    for (const Run* r : {&piped, &piped4, &piped1}) {
        CHECK(r->tracked.back() == plain_total);
        for (int s = 0; s < S; s++) {
            CHECK(r->ids[(size_t)s] == plain.ids[(size_t)s]);
            CHECK(r->xy[(size_t)s] == plain.xy[(size_t)s]);
            CHECK(r->db[(size_t)s] == plain.db[(size_t)s]);
            CHECK(r->tracks[(size_t)s] == plain.tracks[(size_t)s]);
        }
    }
    // one and two large streams in one group: databases split by id, every frame's observations written by the two
    // writer threads (and the driver, for the last set) together
    for (int n_streams : {1, 2}) {
        const Run big_plain = run(1, false, n_streams, 1200);
        const Run big_piped = run(1, true, n_streams, 1200);
        long long big_total = 0;
        for (size_t i = 0; i + 1 < big_plain.tracked.size(); i++) big_total += big_plain.tracked[i];
        CHECK(big_total > 1000 && big_piped.tracked.back() == big_total);
        for (int s = 0; s < n_streams; s++) {
            CHECK(big_piped.ids[(size_t)s] == big_plain.ids[(size_t)s] && big_piped.xy[(size_t)s] == big_plain.xy[(size_t)s]);
            CHECK(big_piped.db[(size_t)s] == big_plain.db[(size_t)s] && big_piped.db[(size_t)s] > 100);
            CHECK(big_piped.tracks[(size_t)s] == big_plain.tracks[(size_t)s]);
        }
    }
    std::printf("%d checks passed, %lld features tracked\n", checks, plain_total);
    return 0;
}
