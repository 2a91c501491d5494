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

    auto run = [&](int groups, bool pipelined) {
        ftk_options o;
        ftk_default_options(&o);
        o.num_pts = 60; o.grid_x = 3; o.grid_y = 2; o.min_px_dist = 6; o.win_size = 9; o.pyr_levels = 2;
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
            std::vector<float> xy(2 * 256);
            std::vector<uint64_t> ids(256);
            const int n = ftk_batch_get_points(b, s, xy.data(), ids.data(), 256);
            CHECK(n >= 0);
            xy.resize(2 * (size_t)n); ids.resize((size_t)n);
            r.xy.push_back(xy); r.ids.push_back(ids);
            r.db.push_back(ftk_batch_database_size(b, s));
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
        }
    }
    std::printf("%d checks passed, %lld features tracked\n", checks, plain_total);
    return 0;
}
