// FAST-9/16 corner detection with per-cell 3x3 non-max suppression and per-cell top-k, for gfx950.
//
// Replaces, per grid cell, cv::FAST(img(roi), pts, threshold, true) + std::sort + truncation
// (reference: src/track/Grider_FAST.h:106-123).  A cell is a cv::Mat ROI in the reference, so no corner can lie
// in the 3-pixel band inside each cell border and NMS never sees a neighbouring cell's score: the score map is
// therefore zero outside the cell insets, and a single image-wide 3x3 strict-maximum test equals per-cell NMS.
//
//   fast_detect_kernel : u8 image -> per-cell candidate lists (score + position keys, unordered); one wavefront per
//                        tile, pixels and scores in LDS, one global atomic per NMS survivor.  The kernel is
//                        bound by vector-instruction issue, so the pre-test handles four pixels per lane from
//                        dword LDS reads and the exact strength runs only on compacted candidates.
//   fast_topk_kernel   : one wavefront per cell extracts the first `per_cell` candidates in the total order
//                        (score desc, y asc, x asc).
#pragma once
#include <cstddef>

#include "ft_common.hpp"

namespace ftd {

// Every per-cell candidate counter has a 128-byte line of its own: read-modify-write operations on one line are
// serialised by the memory system, and with packed counters a few dozen lines took every atomic of a launch.
constexpr int FAST_COUNT_STRIDE = 32;   // ints

struct FastGrid {
    int w, h;
    int size_x, size_y;     // cell size   (Grider_FAST.h:80-81)
    int ct_cols, ct_rows;   // cell counts (Grider_FAST.h:91-92)
    int threshold;
    int per_cell;
    int cand_cap;           // candidate slots per cell
    uint32_t inv_x, inv_y;  // ceil(2^32 / size): g / size == umulhi(g, inv) for g < 65536 and 2 <= size <= 4096
};

// coordinate / cell size without the integer-division sequence
__device__ __forceinline__ int fast_cell_of(int g, int size, uint32_t inv) { return size == 1 ? g : (int)__umulhi((uint32_t)g, inv); }

typedef unsigned short fd_v2u __attribute__((ext_vector_type(2)));

// Compass pre-test for two pixels at once on packed 16-bit lanes: a 9-arc holds one pixel of every opposite pair, so a
// corner needs (N or S) and (E or W) beyond the threshold on the same side.  Returns a non-zero half per passing pixel.
__device__ __forceinline__ uint32_t fast_compass2(uint32_t c, uint32_t n, uint32_t s, uint32_t w, uint32_t e, uint32_t thr2) {
    const fd_v2u C = __builtin_bit_cast(fd_v2u, c), N = __builtin_bit_cast(fd_v2u, n), S = __builtin_bit_cast(fd_v2u, s);
    const fd_v2u W = __builtin_bit_cast(fd_v2u, w), E = __builtin_bit_cast(fd_v2u, e), T = __builtin_bit_cast(fd_v2u, thr2);
    const fd_v2u lo = __builtin_elementwise_sub_sat(C, T), hi = C + T;
    // darker:   (lo > N or lo > S) and (lo > E or lo > W)  <=>  lo > max(min(N, S), min(E, W))
    // brighter: (N > hi or S > hi) and (E > hi or W > hi)  <=>  min(max(N, S), max(E, W)) > hi
    const fd_v2u dark = __builtin_elementwise_sub_sat(lo, __builtin_elementwise_max(__builtin_elementwise_min(N, S), __builtin_elementwise_min(E, W)));
    const fd_v2u bright = __builtin_elementwise_sub_sat(__builtin_elementwise_min(__builtin_elementwise_max(N, S), __builtin_elementwise_max(E, W)), hi);
    return __builtin_bit_cast(uint32_t, dark) | __builtin_bit_cast(uint32_t, bright);
}

// strength = max over the 16 arcs of 9 contiguous circle pixels of min(d) (darker arc) or min(-d) (brighter arc),
// d[k] = v - p[k].  The pixel is a corner iff strength > threshold; its OpenCV score is strength - 1.
// Sliding 9-wide min/max as two rounds of 3-input min/max (v_min3 / v_max3).
__device__ inline int fast_strength(const int (&d)[16]) {
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        mx3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    }
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
        const int mx9 = max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]);
        best = max(max(best, mn9), -mx9);
    }
    return best;
}

// candidate key: score in the top byte, then inverted cell-local y, then inverted cell-local x, so that a larger
// key is earlier in (score desc, y asc, x asc)
__device__ inline uint32_t fast_key(int score, int ly, int lx) {
    return ((uint32_t)score << 24) | ((uint32_t)(4095 - ly) << 12) | (uint32_t)(4095 - lx);
}

// Detector tile geometry.  One wavefront owns one tile and shares nothing with other wavefronts (no workgroup barrier:
// a wavefront that waits for its pixels or for LDS is covered by the seven others of its SIMD, not held up by the
// slowest wavefront of a workgroup).  A tile reports the corners of a 60x14 output region whose first column is 2 mod 4
// (OX = 60*bx + 2: no corner can lie in image columns 0..2, so nothing is lost).  Scores are computed for the 64x16
// "container" around it, CX0 = OX - 2 (a multiple of 4) .. CX0 + 63, rows OY - 1 .. OY + 14: exactly 16 dwords x 16 rows
// = four 4-pixel tasks per lane, and the 1-pixel apron the 3x3 suppression needs.  The pixel tile adds the radius of
// the circle and is staged from column CX0 - 8 in 8-byte groups.  (Measured on 64 images of 752x480: tiles of 14 and
// of 30 rows take the same time within 2 %, the taller tile stages 1.7 instead of 2.1 bytes per output pixel; on the 16
// images of one stream group's launch the time is one wavefront's chain of dependent steps, and the lower tile has half
// of it: 64 streams in four groups + 1 %, 32 streams + 2.5 %, 8 streams + 4 %.)
constexpr int FD_ROUNDS = 4;                     // 4 container rows per round of the 64 lanes
constexpr int FD_OW = 60, FD_OH = 4 * FD_ROUNDS - 2;    // output region
constexpr int FD_CW = 64, FD_CH = 4 * FD_ROUNDS;        // score container
constexpr int FD_PPITCH = 80;                    // pixel tile row: columns CX0-8 .. CX0+71
constexpr int FD_PROWS = FD_CH + 6;              // rows OY-4 .. OY+17
constexpr int FD_GROUPS = FD_PPITCH / 8;         // 8-byte groups per pixel row
constexpr int FD_LIST = FD_CW * FD_CH;           // candidate list capacity
constexpr int FD_ENTRIES_AT = FD_LIST - FD_LIST / 4;   // one task entry per four pixels, kept in the list's last quarter

struct __attribute__((packed)) FdU2Unaligned { uint32_t a, b; };

__host__ __device__ inline int fast_tiles_x(int w) { return (max(w - 2, 1) + FD_OW - 1) / FD_OW; }
__host__ __device__ inline int fast_tiles_y(int h) { return (h + FD_OH - 1) / FD_OH; }

// LDS operations of one wavefront complete in order; this only pins the compiler's ordering between the phases
__device__ __forceinline__ void fd_phase_fence() {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 3 within quads of the row scan, then the row broadcasts)
__device__ __forceinline__ int fd_wave_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2, 3
    return v;
}

// Fused detector: the pixel tile is staged in LDS once; a cheap compass test on four pixels per lane compacts the
// candidates; the exact corner strength is evaluated only for those (dense lanes); strict 3x3 maxima of the LDS score
// tile go straight to the per-cell candidate lists.  No score map in HBM.
// LDS of one detector wavefront (the wavefronts of a workgroup share nothing)
struct __attribute__((aligned(16))) FastTileLds {
    uint8_t pix[FD_PROWS * FD_PPITCH];
    uint8_t sc[FD_CH * FD_CW];
    uint16_t list[FD_LIST];
    uint8_t col_ok[FD_CW];   // column lies in the inset of some cell
    uint8_t row_ok[FD_CH];   // row lies in the inset of some cell
};
static_assert(offsetof(FastTileLds, sc) % 16 == 0 && offsetof(FastTileLds, list) % 16 == 0 && offsetof(FastTileLds, col_ok) % 4 == 0, "LDS alignment of the detector tile");

// one tile (work item `item` of the list (image, tile row, tile column), images outermost) on one wavefront
__device__ __forceinline__ void fast_detect_tile(const ArenaDesc& A, const ImgList& L, const FastGrid& G, uint32_t* __restrict__ cand,
                                                 int* __restrict__ cand_count, int item, int lane, FastTileLds& T) {
    uint8_t* const pix = T.pix;
    uint8_t* const sc = T.sc;
    uint16_t* const list = T.list;
    uint8_t* const col_ok = T.col_ok;
    uint8_t* const row_ok = T.row_ok;
    const int tiles_x = fast_tiles_x(G.w), tiles_y = fast_tiles_y(G.h);
    const int img_i = item / (tiles_x * tiles_y), tile_i = item - img_i * (tiles_x * tiles_y);
    const int slot = L.idx[img_i];
    const int CX0 = (tile_i % tiles_x) * FD_OW, OY = (tile_i / tiles_x) * FD_OH;
    const uint8_t* __restrict__ img = level_img(A, slot, 0);
    const uint32_t pitch = A.lv[0].img_pitch;

    // ---- stage pixels: 22 rows x 10 groups of 8 bytes = 220 loads, four per lane, all issued before the first is used.
    //      Rows are clamped into the image and the group left of the image is clamped to column 0: only insets are
    //      ever evaluated, and an inset pixel's circle lies inside the image.  No byte outside [0, w) of a row is
    //      read: a group that straddles the row end is loaded from the last 8 bytes of the row and shifted into
    //      place, a group past the row end is not loaded at all (the image may be a caller's buffer with stride w). ----
    {
        constexpr int SR = (FD_PROWS * FD_GROUPS + 63) / 64;
        uint2 v[SR];
        // most tiles: every staged byte is an image byte (wave-uniform), so no clamps and no row-end cases
        const bool interior = CX0 >= 8 && CX0 - 8 + FD_PPITCH <= G.w && OY >= 4 && OY - 4 + FD_PROWS <= G.h;
        if (interior) {
            const uint32_t origin = (uint32_t)(OY - 4) * pitch + (uint32_t)(CX0 - 8);
#pragma unroll
            for (int k = 0; k < SR; k++) {
                const int g = min(lane + 64 * k, FD_PROWS * FD_GROUPS - 1);
                const int row = g / FD_GROUPS, c8 = g - row * FD_GROUPS;
                const FdU2Unaligned u = *reinterpret_cast<const FdU2Unaligned*>(img + (origin + (uint32_t)row * pitch + 8u * (uint32_t)c8));
                v[k] = make_uint2(u.a, u.b);
            }
        } else
#pragma unroll
        for (int k = 0; k < SR; k++) {
            const int g = min(lane + 64 * k, FD_PROWS * FD_GROUPS - 1);
            const int row = g / FD_GROUPS, c8 = g - row * FD_GROUPS;
            const int gx = max(CX0 - 8 + 8 * c8, 0);
            // 0: whole group inside the row, 1: straddles the row end (w >= 8), 2: nothing to load, 3: row shorter than a group
            const int mode = gx + 8 <= G.w ? 0 : (gx >= G.w ? 2 : (G.w >= 8 ? 1 : 3));
            const int ld_x = mode == 1 ? G.w - 8 : gx;
            const uint32_t sh = mode == 1 ? 8u * (uint32_t)(gx - ld_x) : 0u;
            const int gy = min(max(OY - 4 + row, 0), G.h - 1);
            const uint8_t* src = img + ((uint32_t)gy * pitch + (uint32_t)ld_x);
            if (mode <= 1) {
                const FdU2Unaligned u = *reinterpret_cast<const FdU2Unaligned*>(src);
                const uint64_t q = (((uint64_t)u.b << 32) | u.a) >> sh;
                v[k] = make_uint2((uint32_t)q, (uint32_t)(q >> 32));
            } else if (mode == 2) {
                v[k] = make_uint2(0u, 0u);
            } else {
                uint64_t q = 0;
                for (int b = 0; b < G.w - gx; b++) q |= (uint64_t)src[b] << (8 * b);
                v[k] = make_uint2((uint32_t)q, (uint32_t)(q >> 32));
            }
        }
#pragma unroll
        for (int k = 0; k < SR; k++) {
            const int g = lane + 64 * k;
            if (g < FD_PROWS * FD_GROUPS) reinterpret_cast<uint2*>(pix)[g] = v[k];   // FD_PPITCH = 8 * FD_GROUPS: groups are contiguous
        }
    }
#pragma unroll
    for (int k = 0; k < FD_CH * FD_CW / 512; k++) reinterpret_cast<uint2*>(sc)[lane + 64 * k] = make_uint2(0u, 0u);   // the whole score tile
    // cell membership is separable: one division per column and per row of the container instead of two per pixel
    {
        const int gx = CX0 + lane;
        const int cxi = fast_cell_of(gx, G.size_x, G.inv_x), lx = gx - cxi * G.size_x;
        col_ok[lane] = (gx < G.w && cxi < G.ct_cols && lx >= 3 && lx < G.size_x - 3) ? 1 : 0;
        if (lane < FD_CH) {
            const int gy = OY - 1 + lane;
            const int cyi = gy >= 0 ? fast_cell_of(gy, G.size_y, G.inv_y) : 0, ly = gy - cyi * G.size_y;
            row_ok[lane] = (gy >= 0 && gy < G.h && cyi < G.ct_rows && ly >= 3 && ly < G.size_y - 3) ? 1 : 0;
        }
    }
    fd_phase_fence();

    // ---- compass test, four pixels per task as two packed pairs ----
    const int thr = G.threshold;
    const uint32_t thr2 = (uint32_t)thr * 0x00010001u;
    const int d = lane & 15;                                                 // container dword column of this lane
    const uint32_t colbytes = *reinterpret_cast<const uint32_t*>(&col_ok[4 * d]);   // one byte per column
    const uint32_t colbits = (colbytes & 1u) | ((colbytes >> 7) & 2u) | ((colbytes >> 14) & 4u) | ((colbytes >> 21) & 8u);
    int n_ent = 0;                                                           // uniform: task entries so far
#pragma unroll
    for (int round = 0; round < FD_ROUNDS; round++) {
        const int r = (lane >> 4) + 4 * round;                               // container row
        uint32_t hits = 0;
        if (colbits != 0 && row_ok[r]) {
            const uint8_t* p = &pix[(r + 3) * FD_PPITCH + 8 + 4 * d];
            const uint32_t C = *reinterpret_cast<const uint32_t*>(p);
            const uint32_t Wd = *reinterpret_cast<const uint32_t*>(p - 4), Ed = *reinterpret_cast<const uint32_t*>(p + 4);
            const uint32_t N = *reinterpret_cast<const uint32_t*>(p - 3 * FD_PPITCH), S = *reinterpret_cast<const uint32_t*>(p + 3 * FD_PPITCH);
            // bytes -> 16-bit lanes (v_perm_b32: selector bytes 0-3 take from the second operand, 4-7 from the first, 0x0c = 0)
            const uint32_t m01 = fast_compass2(__builtin_amdgcn_perm(0u, C, 0x0c010c00u), __builtin_amdgcn_perm(0u, N, 0x0c010c00u),
                                               __builtin_amdgcn_perm(0u, S, 0x0c010c00u), __builtin_amdgcn_perm(C, Wd, 0x0c020c01u),
                                               __builtin_amdgcn_perm(Ed, C, 0x0c040c03u), thr2);      // pixels x, x+1: W = x-3, x-2; E = x+3, x+4
            const uint32_t m23 = fast_compass2(__builtin_amdgcn_perm(0u, C, 0x0c030c02u), __builtin_amdgcn_perm(0u, N, 0x0c030c02u),
                                               __builtin_amdgcn_perm(0u, S, 0x0c030c02u), __builtin_amdgcn_perm(C, Wd, 0x0c040c03u),
                                               __builtin_amdgcn_perm(Ed, C, 0x0c060c05u), thr2);      // pixels x+2, x+3: W = x-1, x; E = x+5, x+6
            // non-zero halves -> bits 0 and 16 (packed minimum with 1), then the four bits side by side
            // (as an assembly statement: the compiler rewrites min(x, 1) into four compares, four selects and two permutes)
            uint32_t h01, h23;
            asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(h01) : "v"(m01));
            asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(h23) : "v"(m23));
            const uint32_t h = h01 | (h23 << 2);                             // bits 0, 2, 16, 18
            hits = (h | (h >> 15)) & colbits;                                // bits 0..3: pixels x .. x+3
        }
        // Compaction without atomics (the list belongs to this wavefront), first by task: one entry (row, dword column,
        // hit bits) per lane that has a hit, ranked with one ballot.  The entries live in the last quarter of the list.
        const uint64_t b = __ballot(hits != 0u);
        if (b != 0ull) {
            if (hits != 0u)
                list[FD_ENTRIES_AT + n_ent + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] =
                    (uint16_t)((r << 8) | (d << 4) | (int)hits);
            n_ent += __builtin_popcountll(b);
        }
    }
    fd_phase_fence();

    // ---- entries -> one list element per candidate pixel (dense lanes for the strength pass).  A batch of 64 entries
    //      writes at most 256 elements from 256 * batch on, which never reaches an entry that is still to be read
    //      (the entries of batch j start at FD_ENTRIES_AT + 64 j = 768 + 64 j, and there are four batches at most). ----
    int n_list = 0;                                                          // uniform: elements in the candidate list
    for (int base = 0; base < n_ent; base += 64) {
        const int e = base + lane < n_ent ? (int)list[FD_ENTRIES_AT + base + lane] : 0;
        const int h = e & 15;
        const int cnt = __builtin_popcount((unsigned)h);
        const int incl = fd_wave_scan(cnt);                                  // inclusive prefix sum over the lanes
        int at = n_list + incl - cnt;
        const int key = ((e >> 8) << 6) | (((e >> 4) & 15) << 2);            // row, first pixel column
        if (h & 1) list[at++] = (uint16_t)key;
        if (h & 2) list[at++] = (uint16_t)(key | 1);
        if (h & 4) list[at++] = (uint16_t)(key | 2);
        if (h & 8) list[at] = (uint16_t)(key | 3);
        n_list += __builtin_amdgcn_readlane(incl, 63);
    }
    fd_phase_fence();

    // ---- exact strength for the compacted candidates; the corners among them are compacted in place (entry n_corner +
    //      rank never lies beyond the entries this pass has already read) ----
    int n_corner = 0;                                                        // uniform
    for (int base = 0; base < n_list; base += 64) {
        const int c = base + lane;
        int i = 0, s = 0;
        if (c < n_list) {
            i = list[c];
            const int sj = i >> 6, si = i & 63;
            const uint8_t* p = &pix[(sj + 3) * FD_PPITCH + 8 + si];
            const int v = p[0];
            int dd[16];
            dd[0] = v - p[3 * FD_PPITCH];      dd[1] = v - p[3 * FD_PPITCH + 1];   dd[2] = v - p[2 * FD_PPITCH + 2];   dd[3] = v - p[FD_PPITCH + 3];
            dd[4] = v - p[3];                  dd[5] = v - p[-FD_PPITCH + 3];      dd[6] = v - p[-2 * FD_PPITCH + 2];  dd[7] = v - p[-3 * FD_PPITCH + 1];
            dd[8] = v - p[-3 * FD_PPITCH];     dd[9] = v - p[-3 * FD_PPITCH - 1];  dd[10] = v - p[-2 * FD_PPITCH - 2]; dd[11] = v - p[-FD_PPITCH - 3];
            dd[12] = v - p[-3];                dd[13] = v - p[FD_PPITCH - 3];      dd[14] = v - p[2 * FD_PPITCH - 2];  dd[15] = v - p[3 * FD_PPITCH - 1];
            s = fast_strength(dd);
        }
        const bool corner = s > thr;
        const uint64_t b = __ballot(corner);
        if (corner) {
            sc[i] = (uint8_t)(s - 1);
            list[n_corner + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)i;
        }
        n_corner += __builtin_popcountll(b);
    }
    fd_phase_fence();

    // ---- strict 3x3 maxima of the output region (container rows 1..14, columns 2..61) -> per-cell candidate lists.
    //      Only the listed corners have a score; the eight neighbours are read unconditionally so that the LDS reads of
    //      one entry are in flight together instead of forming a chain of dependent, short-circuited tests. ----
    const int cells = G.ct_cols * G.ct_rows;
    for (int c = lane; c < n_corner; c += 64) {
        const int i = list[c];
        const int r = i >> 6, col = i & 63;
        const int s = sc[i];
        if (s == 0 || r < 1 || r > FD_OH || col < 2 || col > FD_OW + 1) continue;
        const uint8_t* q = &sc[i];
        const int n0 = q[-FD_CW - 1], n1 = q[-FD_CW], n2 = q[-FD_CW + 1], n3 = q[-1], n4 = q[1], n5 = q[FD_CW - 1], n6 = q[FD_CW], n7 = q[FD_CW + 1];
        const int top = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
        if (s > top) {
            const int gx = CX0 + col, gy = OY - 1 + r;
            const int cxi = fast_cell_of(gx, G.size_x, G.inv_x), cyi = fast_cell_of(gy, G.size_y, G.inv_y);
            const int gcell = img_i * cells + cyi * G.ct_cols + cxi;
            const int at = atomicAdd(&cand_count[(size_t)gcell * FAST_COUNT_STRIDE], 1);
            if (at < G.cand_cap) cand[(uint64_t)gcell * G.cand_cap + at] = fast_key(s, gy - cyi * G.size_y, gx - cxi * G.size_x);
        }
    }
}

__global__ void __launch_bounds__(64) fast_detect_kernel(ArenaDesc A, ImgList L, FastGrid G, uint32_t* __restrict__ cand, int* __restrict__ cand_count) {
    __shared__ FastTileLds T;
    const int total = fast_tiles_x(G.w) * fast_tiles_y(G.h) * L.n;
    const int item = xcd_item(blockIdx.x, total);
    if (item >= total) return;
    fast_detect_tile(A, L, G, cand, cand_count, item, threadIdx.x, T);
}

// wave-wide maximum (result uniform in all lanes): four DPP stages inside the rows of 16 lanes, two across rows
__device__ __forceinline__ uint32_t fast_wave_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast15 into rows 1,3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast31 into rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One wavefront per (image, cell), four cells per workgroup: repeated "largest key below the previous one" with the
// first 256 candidates of the cell in registers (a cell rarely has more than a few dozen NMS survivors).  Also
// reports the number of NMS survivors of the cell and leaves its candidate counter at zero for the next detector pass.
constexpr int FT_TOPK_REGS = 4;
__device__ __forceinline__ void fast_topk_cell(const FastGrid& G, const uint32_t* __restrict__ cand, int* __restrict__ cand_count,
                                               ft_keypoint* __restrict__ out, int* __restrict__ out_count, int* __restrict__ surv_count, int gcell, int lane) {
    const int cell = gcell % (G.ct_cols * G.ct_rows);
    const int survivors = cand_count[(size_t)gcell * FAST_COUNT_STRIDE];
    const int n = min(survivors, G.cand_cap);
    const uint32_t* c = cand + (uint64_t)gcell * G.cand_cap;
    const int keep = min(n, G.per_cell);
    const int cell_x = (cell % G.ct_cols) * G.size_x, cell_y = (cell / G.ct_cols) * G.size_y;
    uint32_t key[FT_TOPK_REGS];
#pragma unroll
    for (int k = 0; k < FT_TOPK_REGS; k++) key[k] = (lane + 64 * k < n) ? c[lane + 64 * k] : 0u;
    uint32_t bound = 0xffffffffu;
    for (int r = 0; r < keep; r++) {
        uint32_t best = 0;
#pragma unroll
        for (int k = 0; k < FT_TOPK_REGS; k++) best = max(best, key[k] < bound ? key[k] : 0u);
        for (int i = lane + 64 * FT_TOPK_REGS; i < n; i += 64) {
            const uint32_t k = c[i];
            if (k < bound && k > best) best = k;
        }
        best = fast_wave_max(best);
        if (lane == 0) {
            ft_keypoint kp;
            kp.x = (float)(cell_x + 4095 - (int)(best & 0xfff));
            kp.y = (float)(cell_y + 4095 - (int)((best >> 12) & 0xfff));
            kp.response = (float)(best >> 24);
            out[(uint64_t)gcell * G.per_cell + r] = kp;
        }
        bound = best;
    }
    if (lane == 0) {   // after the wave's read of the counter above (program order within the wave)
        out_count[gcell] = keep;
        surv_count[gcell] = survivors;
        cand_count[(size_t)gcell * FAST_COUNT_STRIDE] = 0;
    }
}
__global__ void __launch_bounds__(256) fast_topk_kernel(FastGrid G, const uint32_t* __restrict__ cand, int* __restrict__ cand_count,
                                                       ft_keypoint* __restrict__ out, int* __restrict__ out_count, int* __restrict__ surv_count,
                                                       int total_cells) {
    const int gcell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gcell >= total_cells) return;          // uniform for the wave
    fast_topk_cell(G, cand, cand_count, out, out_count, surv_count, gcell, threadIdx.x & 63);
}

}  // namespace ftd
