// Outlier rejection after LK for gfx950: undistortion of both point sets and a seeded RANSAC on the fundamental
// matrix.
//
// Replaces the tail of TrackKLT::perform_matching (reference: src/track/TrackKLT.cpp:405-424): undistort_point on
// both point sets (src/track/TrackBase.h:289-332), cv::findFundamentalMat(FM_RANSAC, 1/max_focal, 0.999) and the
// final mask = mask_klt & mask_ransac.  OpenCV's sample sequence cannot be reproduced without OpenCV, so the sampling
// is the build's own (a counter-based hash); model, error measure and threshold are OpenCV's.  The arithmetic is
// double precision in a fixed order, so the result equals the CPU checker's bit for bit.
//
// One launch per batch of image pairs (ransac_kernel).  A pair of up to RO_CHUNK points has one workgroup; a larger one
// has ceil(n / RO_CHUNK) of them (its "parts"), each owning a contiguous chunk of the points: every part builds the same
// hypotheses (the arithmetic is deterministic, so nothing has to be handed over), counts the inliers among its own
// points, adds its counts to the pair's counters in device memory, and after a barrier among the pair's parts every part
// reads the same totals and takes the same decisions.  Hypotheses are tried in index order in
// rounds of 32.  A hypothesis is solved by eight lanes, one per equation of the 8x9 system, each holding its row in
// registers; pivot search and the pivot row travel by lane shuffles, so the elimination touches no memory.  The inlier
// counts of a round's 32 models are shared by all 256 threads (32 models x 8 slices of the point set, points read from
// LDS), eight models per counting pass.  After every pass the first wavefront walks the pass's models in index order exactly
// as a sequential loop would (lane k holds model k's count and the hypothesis count it would leave): a model with
// more inliers than the best so far becomes the best and lowers the number of hypotheses still to be tried to the
// smallest N with (1 - w^8)^N <= 0.001, w = its inlier ratio (the adaptive iteration count that the confidence argument
// 0.999 of cv::findFundamentalMat stands for; OpenCV's own count falls with w^7, because its minimal sample has seven
// points, and it switches to LMedS below 15 points: the search here is RANSAC on 8-point samples throughout); models past
// that number are ignored.  The same workgroup then writes the
// final mask and the batch results (positions, residuals, status) into the pinned host block.
#pragma once
#include "ft_kernels_lk.hpp"

namespace ftd {

struct OutlierArgs {
    const LKJobDev* jobs;
    const float* prev_xy;      // device
    const float* next_xy;      // device, LK results
    const float* err;          // device
    const uint8_t* status;     // device, LK status
    float4* norm;              // device scratch: x1 y1 x2 y2 normalized, one per point
    float* out_xy;             // pinned host
    float* out_err;            // pinned host
    uint8_t* out_status;       // pinned host
    int hypotheses;            // <= RO_MAX_HYP
    uint32_t seed;
    // completion flag (optional): the last workgroup to finish stores done_seq to done_flag (pinned host memory)
    unsigned* done_count;      // device, zero between launches
    unsigned* done_flag;
    unsigned done_seq;
    // pairs with more than one workgroup: inlier totals [pair][RO_MAX_HYP], and per pair {barrier arrivals, finished
    // parts} (device, zero between launches); error word in pinned host memory (set if a barrier gave up)
    int* gcnt;
    unsigned* gbar;
    unsigned* error_flag;
    unsigned spin_limit;       // barrier polls before a part gives up
    // workgroup -> pair | part << 16 | parts << 24
    uint32_t wg_map[FT_MAX_BATCH];
};

__device__ __forceinline__ uint32_t ro_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// TrackBase::undistort_point: remove K, five fixed-point iterations of the inverse radial-tangential model
__device__ __forceinline__ void ro_undistort_d(float u, float v, const double* c, double& x, double& y) {
    const double ifx = 1. / c[0], ify = 1. / c[1];
    const double x0 = ((double)u - c[2]) * ifx, y0 = ((double)v - c[3]) * ify;
    x = x0; y = y0;
    if (c[4] != 0. || c[5] != 0. || c[6] != 0. || c[7] != 0.) {
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1. + (c[5] * r2 + c[4]) * r2);
            const double dx = 2. * c[6] * x * y + c[7] * (r2 + 2. * x * x);
            const double dy = c[6] * (r2 + 2. * y * y) + 2. * c[7] * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    }
}

__device__ __forceinline__ float2 ro_undistort(float u, float v, const double* c) {
    double x, y;
    ro_undistort_d(u, v, c, x, y);
    return make_float2((float)x, (float)y);
}

__device__ __forceinline__ bool ro_inlier(const double* F, float4 m, float thr2) {
    const float x1 = m.x, y1 = m.y, x2 = m.z, y2 = m.w;
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    const float err = (float)(e1 < e2 ? e2 : e1);   // std::max(e1, e2)
    return err <= thr2;
}

constexpr int RO_HYP_PER_WG = 32;     // hypotheses per round (8 lanes each)
constexpr int RO_SUB = 8;             // models whose inliers are counted together (a power of two)
constexpr int RO_NORM_LDS = 1024;     // normalized points kept in LDS; the rest is read from the global copy
constexpr int RO_MAX_HYP = 1024;      // upper bound of OutlierArgs::hypotheses
constexpr int RO_CHUNK = 512;         // points per workgroup of a pair that is split into parts
constexpr int RO_MAX_PARTS = 32;      // parts per pair at most (larger pairs get larger chunks)
constexpr int RO_MAX_WGS = 128;      // workgroups per launch when pairs are split.  The parts of a pair wait for each other, so
                                     // every workgroup of such a launch must become resident while others spin: the device holds 512
                                     // of them (two per compute unit), which leaves room for the launches of four contexts at once
constexpr unsigned RO_SPIN_LIMIT = 1u << 20;   // barrier polls before a part gives up (a fraction of a second); OutlierArgs::spin_limit

// Barrier among the `parts` workgroups of one pair, called by one thread per workgroup after a workgroup barrier.  The
// k-th barrier of the launch is passed when parts * (k + 1) arrivals have been counted.  All parts of a launch are
// resident at once (the host never launches more split workgroups than the device holds), so every arrival comes;
// the poll count is bounded all the same, and a part that gives up reports it instead of hanging the queue.
// Everything the parts share (the inlier totals and this counter) is read and written with device-scope atomic
// operations only, which are performed at the memory side, past the per-XCD caches; the callers' additions have
// returned their old values before the arrival is counted.  So no cache has to be written back or invalidated here
// (release / acquire on the counter cost about 2 us per barrier, a tenth of the kernel).
__device__ inline bool ro_pair_barrier(unsigned* bar, unsigned target, unsigned spin_limit) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 0; spins < spin_limit; spins++) {
        if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// Hypotheses to try before a sample of eight inliers has been drawn with probability 0.999, given the best model so far
// has `inliers` of n points: the smallest N with (1 - w^8)^N <= 0.001, at most max_h, where the power is the product
// p_N = (...((q * q) * q)...) of plain multiplications in that order (no pow, no log), so the CPU checker gets the same
// number.  A poor model (few inliers) makes that a walk over max_h dependent multiplications by one thread while its
// workgroup waits, so the walk is skipped whenever its outcome is certain: p_N differs from the exact power by less
// than N * 2^-52 relative, and so does the power by repeated squaring below; where that one clears the bound by a
// billionth on either side of N, the walk would stop at the same N.  In the band (and only there) the walk runs.
__device__ inline double ro_pow_int(double q, int e) {
    double r = 1., b = q;
    while (e) {
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}
__device__ inline int ro_hypotheses_needed(int inliers, int n, int max_h) {
    const double w = (double)inliers / (double)n;
    const double w2 = w * w, w4 = w2 * w2, w8 = w4 * w4;
    const double q = 1. - w8;
    if (max_h <= 1) return max_h < 0 ? 0 : max_h;
    if (q <= 0.001) return 1;                     // p_1 = q
    {
        const double hi = 0.001 * (1. + 1e-9), lo = 0.001 * (1. - 1e-9);
        // about ln(0.001) / ln(q); single precision is enough for a first guess
        const float fw8 = (float)w8;
        const float lq = fw8 < 0.01f ? -fw8 * (1.f + 0.5f * fw8) : __logf((float)q);
        const float est = -6.9077553f / lq;
        int N0 = est >= (float)(max_h - 1) ? max_h - 1 : ((int)est < 1 ? 1 : (int)est);
        double p = ro_pow_int(q, N0);
        for (int tries = 0; tries < 6; tries++) {
            if (p > hi) {                          // the walk goes past N0
                if (N0 >= max_h - 1) return max_h;
                const double pn = p * q;
                if (pn < lo) return N0 + 1;
                if (!(pn > hi)) break;
                N0++; p = pn;
            } else if (p < lo) {                   // the walk stops at N0 or before
                if (N0 == 1) return 1;
                const double pp = ro_pow_int(q, N0 - 1);
                if (pp > hi) return N0;
                if (!(pp < lo)) break;
                N0--; p = pp;
            } else {
                break;
            }
        }
    }
    double p = 1.;
    int N = 0;
    while (N < max_h) {
        p = p * q;
        N++;
        if (p <= 0.001) break;
    }
    return N;
}

// T threads per workgroup: 256 (the usual job of a few hundred points: the latency of the dependent phases dominates)
// or 512 (jobs of thousands of points, where the per-point phases dominate).  Hypotheses are always built by the first
// 256 threads; everything per point is shared by all T.  The result does not depend on T.
template <int T>
__global__ void __launch_bounds__(T) ransac_kernel(OutlierArgs a) {
    __shared__ float4 nl[RO_NORM_LDS];
    __shared__ double Fs[RO_HYP_PER_WG][9];
    __shared__ double best_F[9];
    __shared__ int valid[RO_HYP_PER_WG];
    __shared__ int cnt[RO_HYP_PER_WG];
    __shared__ int s_stop, s_best, s_have, s_abort;
    const int t = threadIdx.x;
    const uint32_t wm = a.wg_map[blockIdx.x];
    const int pair = (int)(wm & 0xffffu), part = (int)((wm >> 16) & 0xffu), parts = (int)(wm >> 24);
    const LKJobDev* job = a.jobs + pair;
    const int first = job->first, n = job->count;
    // this workgroup's points [lo, hi) of the pair
    const int chunk = (n + parts - 1) / parts;
    const int lo = part * chunk, hi = min(n, lo + chunk), own = max(hi - lo, 0);
    int* const gcnt = a.gcnt + (size_t)pair * RO_MAX_HYP;
    unsigned* const gbar = a.gbar + 2 * (size_t)pair;
    unsigned barriers = 0;   // passed so far (same in every part: they take the same decisions)
    const double* c0 = job->calib;        // template image camera
    const double* c1 = job->calib + 8;    // target image camera
    const double f0 = c0[0] < c0[1] ? c0[1] : c0[0], f1 = c1[0] < c1[1] ? c1[1] : c1[0];
    const float thr = (float)(1. / (f0 < f1 ? f1 : f0)), thr2 = thr * thr;

    // ---- normalized coordinates of both point sets (own points; index relative to lo) ----
    for (int i = t; i < own; i += T) {
        const int g = first + lo + i;
        const float2 p = ro_undistort(a.prev_xy[2 * g], a.prev_xy[2 * g + 1], c0);
        const float2 q = ro_undistort(a.next_xy[2 * g], a.next_xy[2 * g + 1], c1);
        const float4 m = make_float4(p.x, p.y, q.x, q.y);
        if (i < RO_NORM_LDS) nl[i] = m;
        else a.norm[g] = m;
    }
    if (t == 0) { s_stop = n < 8 ? 0 : a.hypotheses; s_best = 7; s_have = 0; s_abort = 0; }   // a model needs at least 8 inliers
    __syncthreads();   // also orders this workgroup's global writes above before its reads below

    for (int h0 = 0; h0 < s_stop; h0 += RO_HYP_PER_WG) {   // s_stop only changes between the barriers at the loop's end
        if (t < RO_HYP_PER_WG) cnt[t] = 0;
        // ---- one hypothesis per eight lanes, lane r holds equation r ----
        if (T == 256 || t < 256) {   // whole wavefronts either way
            const int r = t & 7, hl = t >> 3, h = h0 + hl;
            const int gbase = (t & 63) & ~7;
            int mine = 0;
            {
                int idx[8];
                uint32_t draw = 0;
                const uint32_t n_inv = 0xffffffffu / (uint32_t)n;   // (n >= 8 here)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    for (;;) {
                        const uint32_t rnd = ro_mix32(a.seed ^ ro_mix32((uint32_t)h * 0x9E3779B9u + 0x85EBCA6Bu * draw));
                        draw++;
                        // rnd % n by the reciprocal: the quotient estimate is short by two at most
                        uint32_t rem = rnd - __umulhi(rnd, n_inv) * (uint32_t)n;
                        rem = rem >= (uint32_t)n ? rem - (uint32_t)n : rem;
                        rem = rem >= (uint32_t)n ? rem - (uint32_t)n : rem;
                        const int cand = (int)rem;
                        bool dup = false;
#pragma unroll
                        for (int q = 0; q < 8; q++) dup |= (q < k) && idx[q] == cand;
                        if (!dup) {
#pragma unroll
                            for (int q = 0; q < 8; q++) idx[q] = (q == k) ? cand : idx[q];
                            break;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) mine = (r == k) ? idx[k] : mine;
            }
            // the sample may belong to another part: normalized here from the raw positions (the same arithmetic as
            // above, so the same value); a pair with one workgroup has it in its own copy
            float4 m;
            if (parts == 1) m = mine < RO_NORM_LDS ? nl[mine] : a.norm[first + mine];
            else {
                const float2 p = ro_undistort(a.prev_xy[2 * (first + mine)], a.prev_xy[2 * (first + mine) + 1], c0);
                const float2 q = ro_undistort(a.next_xy[2 * (first + mine)], a.next_xy[2 * (first + mine) + 1], c1);
                m = make_float4(p.x, p.y, q.x, q.y);
            }
            const double x1 = m.x, y1 = m.y, x2 = m.z, y2 = m.w;
            double A[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.};
            uint32_t colmask = 0;
            bool row_used = false, ok = true;
            int my_pc = 0;
            // Everything below is written without short-circuit operators, nested conditionals or lane-dependent
            // branches: with them the compiler builds trees of divergent branches around the double-precision
            // comparisons and selects, and the eight steps take 25 000 cycles instead of a third of that.
#pragma unroll
            for (int step = 0; step < 8; step++) {
                // largest remaining entry: first in this row, then across the eight rows (ties: lowest row, lowest column)
                double bm = -1.;
                int brc = r << 4;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const double v = fabs(A[j]);
                    const bool take = (int)(!row_used) & (int)(!((colmask >> j) & 1u)) & (int)(v > bm);
                    bm = take ? v : bm;
                    brc = take ? ((r << 4) | j) : brc;
                }
                // partner lanes r ^ 1, r ^ 2, then 7 - r (the other quad, whose four lanes agree by then): lane
                // permutations inside a row of 16, no LDS round trip
#define RO_PIVOT_ROUND(ctrl)                                                                                               \
                {                                                                                                          \
                    const uint64_t bb = __builtin_bit_cast(uint64_t, bm);                                                   \
                    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bb, ctrl, 0xF, 0xF, false);          \
                    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bb >> 32), ctrl, 0xF, 0xF, false);  \
                    const int orc = __builtin_amdgcn_update_dpp(0, brc, ctrl, 0xF, 0xF, false);                            \
                    const double om = __builtin_bit_cast(double, ((uint64_t)ohi << 32) | olo);                              \
                    const bool take = (int)(om > bm) | ((int)(om == bm) & (int)((orc >> 4) < (brc >> 4)));                 \
                    bm = take ? om : bm;                                                                                   \
                    brc = take ? orc : brc;                                                                                \
                }
                RO_PIVOT_ROUND(0xB1)    // quad_perm [1,0,3,2]
                RO_PIVOT_ROUND(0x4E)    // quad_perm [2,3,0,1]
                RO_PIVOT_ROUND(0x141)   // row_half_mirror
#undef RO_PIVOT_ROUND
                ok = (int)ok & (int)(bm >= 1e-12);   // same for the eight lanes; they run on (the exchanges need all of them)
                const int pr = brc >> 4, pc = brc & 15;
                double P[9];
#pragma unroll
                for (int j = 0; j < 9; j++) P[j] = __shfl(A[j], gbase | pr);
                // entry pc of the pivot row and of this row, picked with bit masks
                uint64_t ppc = 0, apc = 0;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const uint64_t m = j == pc ? ~0ull : 0ull;
                    ppc |= __builtin_bit_cast(uint64_t, P[j]) & m;
                    apc |= __builtin_bit_cast(uint64_t, A[j]) & m;
                }
                const double inv = 1. / __builtin_bit_cast(double, ppc);
                const double mult = __builtin_bit_cast(double, apc) * inv;
                colmask |= 1u << pc;
                const bool is_pivot_row = r == pr;
                row_used = (int)row_used | (int)is_pivot_row;
                my_pc = is_pivot_row ? pc : my_pc;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const double upd = A[j] - mult * P[j];
                    const bool keep = (int)is_pivot_row | (int)((colmask >> j) & 1u);
                    A[j] = keep ? A[j] : upd;
                }
            }
            const int free_col = __builtin_ctz(~colmask & 0x1ffu);
            uint64_t afree = 0, apiv = 0;
#pragma unroll
            for (int j = 0; j < 9; j++) {
                afree |= __builtin_bit_cast(uint64_t, A[j]) & (j == free_col ? ~0ull : 0ull);
                apiv |= __builtin_bit_cast(uint64_t, A[j]) & (j == my_pc ? ~0ull : 0ull);
            }
            const double y = -__builtin_bit_cast(double, afree) / __builtin_bit_cast(double, apiv);
            Fs[hl][my_pc] = y;
            if (r == 0) {
                Fs[hl][free_col] = 1.;
                valid[hl] = (ok && h < a.hypotheses) ? 1 : 0;
            }
        }
        __syncthreads();

        // ---- inlier counts and the walk over the models in index order, RO_SUB models at a time: the search usually
        //      ends within the first few hypotheses, and the models past that point are never counted ----
        for (int q0 = 0; q0 < RO_HYP_PER_WG && h0 + q0 < s_stop; q0 += RO_SUB) {   // s_stop: see above
            {
                const int hh = q0 + (t & (RO_SUB - 1)), slice = t / RO_SUB;
                if (valid[hh] && h0 + hh < s_stop) {
                    double F[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) F[q] = Fs[hh][q];
                    int c = 0;
                    for (int i = slice; i < own; i += T / RO_SUB) c += ro_inlier(F, i < RO_NORM_LDS ? nl[i] : a.norm[first + lo + i], thr2) ? 1 : 0;
                    if (c) atomicAdd(&cnt[hh], c);
                }
            }
            __syncthreads();
            if (parts > 1) {
                // own counts to the pair's totals, barrier among the parts, totals back (every part reads the same)
                if (t < RO_SUB && cnt[q0 + t]) {
                    const int before = atomicAdd(&gcnt[h0 + q0 + t], cnt[q0 + t]);
                    asm volatile("" ::"v"(before));   // the addition has been performed once its old value is here
                }
                __syncthreads();
                if (t == 0) {
                    barriers++;
                    if (!ro_pair_barrier(gbar, (unsigned)parts * barriers, a.spin_limit)) s_abort = 1;
                }
            }
            // As a sequential loop over the hypotheses would see them, by the first wavefront: lane k holds model k's
            // count and the number of hypotheses it would leave if it became the best (worked out side by side: that
            // number is a function of the count alone); the walk itself then runs on registers.
            if (t < 64) {
                const int k_own = t & (RO_SUB - 1);
                const int c_own = parts > 1 ? __hip_atomic_load(&gcnt[h0 + q0 + k_own], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cnt[q0 + k_own];
                const int v_own = valid[q0 + k_own];
                int stop = s_stop, best = s_best, win = -1;
                const int need_own = (v_own && c_own > best) ? ro_hypotheses_needed(c_own, n, a.hypotheses) : a.hypotheses;
#pragma unroll
                for (int k = 0; k < RO_SUB; k++) {
                    const int ck = __builtin_amdgcn_readlane(c_own, k), vk = __builtin_amdgcn_readlane(v_own, k), nk = __builtin_amdgcn_readlane(need_own, k);
                    if (h0 + q0 + k < stop && vk && ck > best) {
                        best = ck; win = k;
                        stop = nk < stop ? nk : stop;
                    }
                }
                if (win >= 0 && t < 9) best_F[t] = Fs[q0 + win][t];
                if (t == 0) {
                    if (win >= 0) s_have = 1;
                    s_stop = s_abort ? 0 : stop; s_best = best;
                }
            }
            __syncthreads();
        }
    }

    // ---- final mask and the batch results into the pinned host block ----
    const bool have_model = s_have != 0 && !s_abort;
    if (s_abort && t == 0) *reinterpret_cast<volatile unsigned*>(a.error_flag) = 1u;
    if (parts > 1) {
        // the last part of the pair to get here leaves the pair's counters at zero for the next launch (every part has
        // passed its last barrier and read its last totals by then)
        __shared__ unsigned s_last;
        if (t == 0) s_last = __hip_atomic_fetch_add(&gbar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)parts - 1;
        __syncthreads();
        if (s_last) {   // (atomic stores, like every other access to these words)
            for (int i = t; i < min(RO_MAX_HYP, a.hypotheses + RO_HYP_PER_WG); i += T) __hip_atomic_store(&gcnt[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 2) __hip_atomic_store(&gbar[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int i = t; i < own; i += T) {
        const int g = first + lo + i;
        const bool keep = have_model && a.status[g] && ro_inlier(best_F, i < RO_NORM_LDS ? nl[i] : a.norm[g], thr2);
        a.out_xy[2 * g] = a.next_xy[2 * g];
        a.out_xy[2 * g + 1] = a.next_xy[2 * g + 1];
        a.out_err[g] = a.err[g];
        a.out_status[g] = keep ? 1 : 0;
    }
    if (a.done_flag) {
        // The results above went to pinned host memory, which no cache of the device holds: once a store is
        // acknowledged (vmcnt) it has been performed at the host.  The flag protocol rests on that property of the
        // hardware, not on the language's memory model: the counter below is a relaxed atomic, and only the last
        // workgroup issues a fence.  Release / acquire on the counter make every workgroup write back its L2 (measured:
        // + 3.3 us per launch of 64 workgroups; a system-scope fence in every thread: + 12 us).  FT_DONE_FLAG=0 turns
        // the flag off (every wait then goes through an event), for telling a stale result from another fault.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __syncthreads();
        if (t == 0) {
            const unsigned finished = atomicAdd(a.done_count, 1u);
            if (finished == gridDim.x - 1) {   // every other workgroup's stores have been acknowledged
                *a.done_count = 0;
                __threadfence_system();
                *reinterpret_cast<volatile unsigned*>(a.done_flag) = a.done_seq;
            }
        }
    }
}

}  // namespace ftd
