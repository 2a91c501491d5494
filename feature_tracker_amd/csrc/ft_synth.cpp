// Seeded synthetic camera sequences for the benchmark and the end-to-end tests (host code, no GPU).
//
// A stream is a textured canvas (sum of random-phase sinusoids with periods of 8..128 px, plus filled rectangles
// and discs that give FAST corners) viewed through a window that translates and rotates in the image plane from
// frame to frame; every frame is a bilinear resampling of the canvas plus seeded Gaussian noise.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

template <class F>
void rows_parallel(int rows, int threads, F f) {
    threads = std::max(1, std::min(threads, rows));
    if (threads == 1) { f(0, rows); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(f, (int)((int64_t)rows * t / threads), (int)((int64_t)rows * (t + 1) / threads));
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// canvas of cw x ch pixels for stream `seed`
void fts_canvas(uint8_t* out, int cw, int ch, uint32_t seed, int n_waves, int n_shapes, int threads) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    struct Wave { float kx, ky, ph, amp; };
    std::vector<Wave> waves((size_t)n_waves);
    for (auto& w : waves) {
        const double period = 8.0 + 120.0 * U(rng), ang = 2 * M_PI * U(rng);
        w.kx = (float)(2 * M_PI * std::cos(ang) / period); w.ky = (float)(2 * M_PI * std::sin(ang) / period);
        w.ph = (float)(2 * M_PI * U(rng)); w.amp = (float)(0.3 + 0.7 * U(rng));
    }
    std::vector<float> acc((size_t)cw * ch);
    const float norm = 40.f / std::sqrt((float)std::max(1, n_waves) / 2.f);
    rows_parallel(ch, threads, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++) {
            float* r = acc.data() + (size_t)y * cw;
            for (int x = 0; x < cw; x++) r[x] = 0.f;
            for (const Wave& w : waves) {
                const float base = w.ky * y + w.ph;
                for (int x = 0; x < cw; x++) r[x] += w.amp * std::sin(w.kx * x + base);
            }
            for (int x = 0; x < cw; x++) r[x] = 128.f + norm * r[x];
        }
    });
    for (int s = 0; s < n_shapes; s++) {
        const double cx = U(rng) * cw, cy = U(rng) * ch, sz = 4 + 20 * U(rng), val = -70 + 140 * U(rng);
        const bool rect = U(rng) < 0.5;
        const double ang = U(rng) * M_PI / 2, asp = 0.4 + 0.6 * U(rng);
        const int x0 = std::max(0, (int)(cx - 1.5 * sz - 2)), x1 = std::min(cw - 1, (int)(cx + 1.5 * sz + 2));
        const int y0 = std::max(0, (int)(cy - 1.5 * sz - 2)), y1 = std::min(ch - 1, (int)(cy + 1.5 * sz + 2));
        const double ca = std::cos(ang), sa = std::sin(ang);
        for (int y = y0; y <= y1; y++)
            for (int x = x0; x <= x1; x++) {
                const double dx = x - cx, dy = y - cy;
                bool in;
                if (rect) { const double u = dx * ca + dy * sa, v = -dx * sa + dy * ca; in = std::fabs(u) < sz && std::fabs(v) < sz * asp; }
                else in = dx * dx + dy * dy < sz * sz;
                if (in) acc[(size_t)y * cw + x] += (float)val;
            }
    }
    for (size_t i = 0; i < acc.size(); i++) out[i] = (uint8_t)std::min(255.f, std::max(0.f, std::nearbyint(acc[i])));
}

// Canvas for the feature-dense configurations (thousands of features per frame): the same sinusoid background at a
// lower amplitude, and instead of large shapes a jittered lattice of small Gaussian blobs of either polarity.  A blob
// (sigma 1.15..1.25 px, amplitude 95..101) has one strict intensity extremum, so FAST + 3x3 NMS answers with a single
// strong corner per blob; blobs are `spacing` +- `jitter` pixels apart, so with spacing - 2*jitter above the tracker's
// minimum feature distance no two of them fall into one cell of its occupancy grid.  The blobs are about equally
// strong on purpose: their FAST scores then differ by sub-pixel phase and noise only, the per-cell ranking changes
// from frame to frame, and the "best need/cells + 1 per cell" rule of the detector keeps finding untracked blobs
// (with a fixed ranking it would return the already tracked ones again and replenishment would stall).
void fts_canvas_blobs(uint8_t* out, int cw, int ch, uint32_t seed, int n_waves, double spacing, double jitter, int threads) {
    std::mt19937 rng(seed ^ 0x9e3779b9u);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    struct Wave { float kx, ky, ph, amp; };
    std::vector<Wave> waves((size_t)n_waves);
    for (auto& w : waves) {
        const double period = 16.0 + 112.0 * U(rng), ang = 2 * M_PI * U(rng);
        w.kx = (float)(2 * M_PI * std::cos(ang) / period); w.ky = (float)(2 * M_PI * std::sin(ang) / period);
        w.ph = (float)(2 * M_PI * U(rng)); w.amp = (float)(0.3 + 0.7 * U(rng));
    }
    std::vector<float> acc((size_t)cw * ch);
    const float norm = 14.f / std::sqrt((float)std::max(1, n_waves) / 2.f);
    rows_parallel(ch, threads, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++) {
            float* r = acc.data() + (size_t)y * cw;
            for (int x = 0; x < cw; x++) r[x] = 0.f;
            for (const Wave& w : waves) {
                const float base = w.ky * y + w.ph;
                for (int x = 0; x < cw; x++) r[x] += w.amp * std::sin(w.kx * x + base);
            }
            for (int x = 0; x < cw; x++) r[x] = 128.f + norm * r[x];
        }
    });
    const int nx = std::max(1, (int)(cw / spacing)), ny = std::max(1, (int)(ch / spacing));
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            const double cx = (i + 0.5) * spacing + (2 * U(rng) - 1) * jitter, cy = (j + 0.5) * spacing + (2 * U(rng) - 1) * jitter;
            const double sigma = 1.15 + 0.1 * U(rng), amp = (95 + 6 * U(rng)) * (U(rng) < 0.5 ? -1.0 : 1.0);
            const double inv = 1.0 / (2 * sigma * sigma);
            const int x0 = std::max(0, (int)std::floor(cx - 5)), x1 = std::min(cw - 1, (int)std::ceil(cx + 5));
            const int y0 = std::max(0, (int)std::floor(cy - 5)), y1 = std::min(ch - 1, (int)std::ceil(cy + 5));
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    const double d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
                    acc[(size_t)y * cw + x] += (float)(amp * std::exp(-d2 * inv));
                }
        }
    for (size_t i = 0; i < acc.size(); i++) out[i] = (uint8_t)std::min(255.f, std::max(0.f, std::nearbyint(acc[i])));
}

// frame of w x h pixels: pixel (x, y) looks at canvas position c + R(theta) * ((x, y) - (w-1, h-1)/2)
void fts_frame(const uint8_t* canvas, int cw, int ch, uint8_t* out, int w, int h, size_t ostride, double ccx, double ccy, double theta,
               double sigma, uint32_t noise_seed, int threads) {
    const double ca = std::cos(theta), sa = std::sin(theta);
    const double hx = (w - 1) * 0.5, hy = (h - 1) * 0.5;
    rows_parallel(h, threads, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++) {
            std::mt19937 rng(noise_seed * 2654435761u + (uint32_t)y);   // per-row seed: the frame does not depend on `threads`
            std::normal_distribution<float> N(0.f, (float)sigma);
            uint8_t* o = out + (size_t)y * ostride;
            for (int x = 0; x < w; x++) {
                const double dx = x - hx, dy = y - hy;
                double sx = ccx + ca * dx - sa * dy, sy = ccy + sa * dx + ca * dy;
                sx = std::min((double)cw - 1.001, std::max(0.0, sx)); sy = std::min((double)ch - 1.001, std::max(0.0, sy));
                const int ix = (int)sx, iy = (int)sy;
                const float fx = (float)(sx - ix), fy = (float)(sy - iy);
                const uint8_t* p = canvas + (size_t)iy * cw + ix;
                float v = (1 - fy) * ((1 - fx) * p[0] + fx * p[1]) + fy * ((1 - fx) * p[cw] + fx * p[cw + 1]);
                if (sigma > 0) v += N(rng);
                o[x] = (uint8_t)std::min(255.f, std::max(0.f, std::nearbyint(v)));
            }
        }
    });
}

}  // extern "C"
