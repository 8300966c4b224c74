// A2-A7 -- grid FAST-10 extractor with ORB orientation + rotated BRIEF.
// Replaces FeatureDetector::Detect / ComputeAngleAndDescriptor (src/Algorithm/FeatureDetector.cpp:345-444,
// 467-594) including the libfast calls at :366-381.  Bit-exact with oracle/{fast,orb}.c.
//
// Design (not a translation of the CPU loop):
//  k_fast_select  one workgroup per 64x32 pixel tile of one level of one frame.  The tile (+halo 5)
//                 is staged once in LDS with 4-byte coalesced loads; every later access is LDS.
//                 1) 16-bit brighter/darker ring masks -> FAST-10 test (10 contiguous bits by
//                    shift-AND), corners appended to an LDS work list (dense work for step 2);
//                 2) per listed corner the score in closed form (max over arcs of the arc minimum,
//                    equal to libfast's bisection result) into an LDS score tile;
//                 3) 3x3 non-max suppression on the score tile, InFrame border test, Shi-Tomasi
//                    from LDS, and the per-cell winner by 64-bit atomicMax on
//                    (ordered(score) << 32 | ~visit_index): the reference's sequential
//                    "strictly greater replaces, levels 0->2, raster order" rule is an argmax with
//                    earliest-visit tie-break, so no ordered compaction of corners is needed.
//                    A NaN Shi-Tomasi score (sqrt of a slightly negative discriminant) is kept
//                    order-exact through a second key (atomicMin of visit_index<<1|isnan).
//  k_compact      one workgroup per frame: block scan over the grid cells -> keypoint SoA in cell order
//                 (the order Detect pushes features to frame->_features).
//  k_describe     one wavefront per keypoint: 39x39 neighbourhood in LDS, integer moments by
//                 wave reduction, cv::fastAtan2 polynomial, 256 tests = 4 ballots of 64 lanes.
#include "ygz_internal.h"
#include <cstring>
#include <stdlib.h>
#include <stdio.h>
#include "../../include/ygz_orb_pattern.h"

#define FT_W    64
#ifndef FT_H
#define FT_H    32
#endif
#ifndef FT_NT
#define FT_NT   128                // threads per tile (two wavefronts: 0.496 ms per 512 VGA frames; 256 threads 0.520, 64 x 16 tiles of 128 threads 0.510, 64 x 8 of 64 0.637, 64 x 64 of 256 0.521)
#endif
#define FT_NW   (FT_NT / 64)
static_assert(2 * (FT_H + 2) <= FT_NT && (FT_H + 2 + FT_NW - 1) / FT_NW < 31, "the two ring columns are one thread each; a lane's row flags are bits of one dword");
#define FT_LW   80                 // staged columns: x in [x0-8, x0+72)
#define FT_LH   (FT_H + 10)        // staged rows:    y in [y0-5, y0+FT_H+5)
#define FT_X0   8                  // tile column of x0
#define FT_Y0   5                  // tile row of y0
#define R1_W    (FT_W + 2)         // corner-test region = interior + 1
#define R1_H    (FT_H + 2)
#define R1_LW   68

static __device__ const int c_circ_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static __device__ const int c_circ_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

struct FastArgs {
    const uint8_t *img;            // level base (slot 0)
    int w, h, level;
    int thr, tie;
    int img_cols, img_rows;        // level-0 size (Frame::_color)
    int cell, grid_cols, cells;
    uint32_t cell_magic;           // ceil(2^32 / cell) (0: cell == 1): v / cell == umulhi(v, cell_magic) for v < 2^16
    uint32_t *cell_first;
    unsigned long long *cell_best;
    const uint8_t *occupied;
    uint8_t *dbg_score, *dbg_nms;  // may be null
    int slot_begin, n_slots;
    long long *dbg_cyc;            // YGZ_FAST_TIMERS: [level][8] phase cycles
};

__device__ __forceinline__ bool ring10(uint32_t m)
{
    m |= m << 16;
    const uint32_t a = m & (m >> 1);
    const uint32_t b = a & (a >> 2);
    const uint32_t c = b & (b >> 4);
    return (c & (a >> 8) & 0xFFFFu) != 0;
}

static __device__ __forceinline__ void fast_select_body(const FastArgs &A, const int bx_, const int by_, const int so_)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[FT_LH][FT_LW];
    __shared__ __attribute__((aligned(16))) uint8_t sc[R1_H][R1_LW];      // 0 = no corner, else score+1
    __shared__ uint16_t list[R1_W * R1_H], cand[R1_W * R1_H];
    __shared__ int n_list, n_cand;
    __shared__ uint8_t occ_l[16][32];                                      // the occupied flags of the grid cells under the tile

    const int slot = A.slot_begin + so_;
    const size_t npix = (size_t)A.w * A.h;
    const uint8_t *img = A.img + (size_t)slot * npix;
    const int x0 = bx_ * FT_W, y0 = by_ * FT_H;
    const int tid = threadIdx.x;
    // a corner outside Frame::InFrame(px, 20, L) can never be selected (step 3): tiles entirely outside that rectangle are
    // skipped unless the caller asked for the full corner / NMS maps
    if (!A.dbg_score && !A.dbg_nms) {
        const int sc_ = 1 << A.level;
        if (x0 + FT_W <= 20 * sc_ || y0 + FT_H <= 20 * sc_ || x0 >= (A.img_cols - 20) * sc_ || y0 >= (A.img_rows - 20) * sc_) return;
    }

#ifdef YGZ_FAST_TIMERS
    long long tq = clock64();
#define FS_PHASE(k) do { if (tid == 0 && A.dbg_cyc && so_ == 0) { const long long tn = clock64(); A.dbg_cyc[8 * (by_ * 32 + bx_) + k] = tn - tq; tq = tn; } } while (0)
#else
#define FS_PHASE(k) do { } while (0)
#endif
#ifdef YGZ_FAST_TIMERS
#define FS_COUNT(n) do { if (A.dbg_cyc && so_ == 0) { A.dbg_cyc[8 * (by_ * 32 + bx_) + 4] = (n); A.dbg_cyc[8 * (by_ * 32 + bx_) + 5] = 1; } } while (0)
#else
#define FS_COUNT(n) do { } while (0)
#endif
    if (tid == 0) { n_list = 0; n_cand = 0; }
    for (int i = tid; i < R1_H * R1_LW / 4; i += FT_NT) reinterpret_cast<uint32_t *>(&sc[0][0])[i] = 0u;
    // grid cell of a level pixel coordinate: (v * 2^level) / cell (FeatureDetector.cpp:383-386) without the integer division
    const int scale = 1 << A.level;
    auto cell_of = [&](const int v) { return A.cell_magic ? (int)__umulhi((uint32_t)(v * scale), A.cell_magic) : v * scale; };
    // the occupied flags step 3 needs are requested now, with the tile (a global load behind the NMS was a second memory latency per tile)
    const int gx0 = cell_of(x0), gy0 = cell_of(y0);
    const bool occ_fits = cell_of(x0 + FT_W - 1) - gx0 < 32 && cell_of(y0 + FT_H - 1) - gy0 < 16;
    constexpr int FT_RP = FT_NT / 32, OCC_PS = 16 / FT_RP;                  // rows per pass of the (row, 32 columns) thread mapping
    uint8_t occ_v[OCC_PS];
#pragma unroll
    for (int ps = 0; ps < OCC_PS; ++ps) occ_v[ps] = 1;
    if (occ_fits) {
#pragma unroll
        for (int ps = 0; ps < OCC_PS; ++ps) {
            const int k = (gy0 + (tid >> 5) + FT_RP * ps) * A.grid_cols + gx0 + (tid & 31);
            if (k >= 0 && k < A.cells) occ_v[ps] = A.occupied[(size_t)slot * A.cells + k];
        }
    }
    const bool aligned = (A.w & 3) == 0;
    if (aligned && x0 - FT_X0 >= 0 && x0 - FT_X0 + FT_LW <= A.w && y0 - FT_Y0 >= 0 && y0 - FT_Y0 + FT_LH <= A.h) {
        // the staged rectangle lies inside the level (every tile but those along the level's edges): thread = (dword column of 20, row),
        // passes of FT_RP rows -- no bounds per item, no index division; all loads go out before the first LDS store
        const int c = tid & 31, r = tid >> 5;
        constexpr int ST_PS = (FT_LH + FT_RP - 1) / FT_RP;
        if (c < FT_LW / 4) {
            const uint8_t *p = img + (size_t)(y0 - FT_Y0 + r) * A.w + (x0 - FT_X0 + 4 * c);
            uint32_t tv[ST_PS];
#pragma unroll
            for (int ps = 0; ps < ST_PS; ++ps) { tv[ps] = 0u; if (FT_RP * (ps + 1) <= FT_LH || r + FT_RP * ps < FT_LH) tv[ps] = *reinterpret_cast<const uint32_t *>(p + (size_t)(FT_RP * ps) * A.w); }
#pragma unroll
            for (int ps = 0; ps < ST_PS; ++ps) if (FT_RP * (ps + 1) <= FT_LH || r + FT_RP * ps < FT_LH) *reinterpret_cast<uint32_t *>(&tile[r + FT_RP * ps][4 * c]) = tv[ps];
        }
    } else {
    // all of a thread's tile loads go out before the first LDS store (rolled, the loop was a chain of four dependent memory latencies)
    constexpr int FT_TOT = FT_LH * (FT_LW / 4), FT_LN = (FT_TOT + FT_NT - 1) / FT_NT;
    uint32_t tv[FT_LN];
#pragma unroll
    for (int k4 = 0; k4 < FT_LN; ++k4) {
        const int i = tid + FT_NT * k4;
        const int r = i / (FT_LW / 4), c4 = (i % (FT_LW / 4)) * 4;
        const int y = y0 - FT_Y0 + r, x = x0 - FT_X0 + c4;
        uint32_t v = 0;
        if (i < FT_TOT && y >= 0 && y < A.h) {
            const uint8_t *row = img + (size_t)y * A.w;
            if (aligned && x >= 0 && x + 3 < A.w) v = *reinterpret_cast<const uint32_t *>(row + x);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (x + k >= 0 && x + k < A.w) v |= (uint32_t)row[x + k] << (8 * k);
            }
        }
        tv[k4] = v;
    }
#pragma unroll
    for (int k4 = 0; k4 < FT_LN; ++k4) {
        const int i = tid + FT_NT * k4;
        if (i < FT_TOT) *reinterpret_cast<uint32_t *>(&tile[i / (FT_LW / 4)][(i % (FT_LW / 4)) * 4]) = tv[k4];
    }
    }
#pragma unroll
    for (int ps = 0; ps < OCC_PS; ++ps) occ_l[(tid >> 5) + FT_RP * ps][tid & 31] = occ_v[ps];
    __syncthreads();
    FS_PHASE(0);

    // ---- 1) FAST-10 segment test on the interior + 1 ring, in two passes so that the expensive part runs on dense lanes:
    //      1a every pixel: the 4 compass pixels.  A 10-arc of the 16-ring always contains two ADJACENT compass pixels, so a
    //         corner needs an adjacent pair that is brighter than p + t (or darker than p - t); survivors go to an LDS list.
    //      1b list entries: full 16-pixel ring -> bright / dark masks -> 10 contiguous bits.
    // (a pixel of the region is named (ry << 7) | rx in the lists.)  1a: wavefront = row, lane = interior column, the two ring columns in a
    // pass of their own: the row test is scalar, the column test loop-invariant, every LDS address the same register + a constant
    {
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto compass = [&](const int ry, const int rx) {
            const uint8_t *c = &tile[ry + FT_Y0 - 1][rx + FT_X0 - 1];
            const int p = *c, hi = p + A.thr, lo = p - A.thr;
            const int v0 = c[3 * FT_LW], v4 = c[3], v8 = c[-3 * FT_LW], v12 = c[-3];
            const bool b0 = v0 > hi, b4 = v4 > hi, b8 = v8 > hi, b12 = v12 > hi;
            const bool d0 = v0 < lo, d4 = v4 < lo, d8 = v8 < lo, d12 = v12 < lo;
            return (((b0 | b8) & (b4 | b12)) | ((d0 | d8) & (d4 | d12))) != 0;        // some adjacent compass pair agrees
        };
        const int xl = x0 + lane;                                              // rx = 1 + lane
        const bool xok = xl >= 3 && xl < A.w - 3;
        // every row's flag first (the LDS reads of the nine rows are independent: one round trip), then ONE append per wavefront
        constexpr int FT_CI = (R1_H + FT_NW - 1) / FT_NW;                   // rows per wavefront
        uint32_t cm = 0u;
#pragma unroll
        for (int it = 0; it < FT_CI; ++it) {
            const int ry = wv + FT_NW * it, y = y0 - 1 + ry;
            const bool t = compass(min(ry, R1_H - 1), 1 + lane);
            if (ry < R1_H && y >= 3 && y < A.h - 3 && xok && t) cm |= 1u << it;
        }
        const int rry = min(tid >> 1, R1_H - 1), rrx = (tid & 1) ? R1_W - 1 : 0;
        {
            const int x = x0 - 1 + rrx, y = y0 - 1 + rry;
            const bool t = compass(rry, rrx);
            if (tid < 2 * R1_H && x >= 3 && y >= 3 && x < A.w - 3 && y < A.h - 3 && t) cm |= 1u << FT_CI;
        }
        const int cnt = __popc(cm);
        int incl = cnt;
#define FS_SCAN(ctrl, rmask) incl += __builtin_amdgcn_update_dpp(0, incl, (ctrl), (rmask), 0xF, false)
        FS_SCAN(0x111, 0xF); FS_SCAN(0x112, 0xF); FS_SCAN(0x114, 0xF); FS_SCAN(0x118, 0xF); FS_SCAN(0x142, 0xA); FS_SCAN(0x143, 0xC);
#undef FS_SCAN
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(&n_cand, total);
        int pos = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
#pragma unroll
        for (int it = 0; it < FT_CI; ++it) if ((cm >> it) & 1u) cand[pos++] = (uint16_t)(((wv + FT_NW * it) << 7) | (1 + lane));
        if ((cm >> FT_CI) & 1u) cand[pos] = (uint16_t)((rry << 7) | rrx);
    }
    __syncthreads();
    const int nc = n_cand;
    for (int ci = tid; ci < nc; ci += FT_NT) {
        const int i = cand[ci];
        const int ry = i >> 7, rx = i & 127;
        const uint8_t *c = &tile[ry + FT_Y0 - 1][rx + FT_X0 - 1];
        const int p = *c, hi = p + A.thr, lo = p - A.thr;
        // the sign bit of (hi - v) / (v - lo) shifted in from the right: two instructions per ring pixel and polarity.  The masks come out
        // mirrored (ring pixel 0 in bit 15), which a test for ten contiguous bits on a circle does not see
        uint32_t bright = 0, dark = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int v = c[c_circ_dy[k] * FT_LW + c_circ_dx[k]];
            bright = __builtin_amdgcn_alignbit(bright, (uint32_t)(hi - v), 31);
            dark = __builtin_amdgcn_alignbit(dark, (uint32_t)(v - lo), 31);
        }
        if (ring10(bright) || ring10(dark)) {
            const int pos = atomicAdd(&n_list, 1);
            list[pos] = (uint16_t)i;
        }
    }
    __syncthreads();
    FS_PHASE(1);
    const int n = n_list;

    // ---- 2) score in closed form: max over 10-arcs of min(v-p) (bright) / min(p-v) (dark), minus 1
    for (int li = tid; li < n; li += FT_NT) {
        const int i = list[li];
        const int ry = i >> 7, rx = i & 127;
        const uint8_t *c = &tile[ry + FT_Y0 - 1][rx + FT_X0 - 1];
        const int p = *c;
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = (int)c[c_circ_dy[k] * FT_LW + c_circ_dx[k]] - p;
        int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
        for (int k = 0; k < 16; ++k) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
        int best = -1000;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int mn10 = min(min(mn4[k], mn4[(k + 4) & 15]), mn2[(k + 8) & 15]);
            const int mx10 = max(max(mx4[k], mx4[(k + 4) & 15]), mx2[(k + 8) & 15]);
            best = max(best, max(mn10, -mx10));
        }
        sc[ry][rx] = (uint8_t)(best - 1 + 1);          // score = best-1 in [thr,254]; stored +1
    }
    __syncthreads();
    FS_PHASE(2);

    // ---- 3) NMS + border + Shi-Tomasi + per-cell winner, in two passes so that the 64-pixel Shi-Tomasi window runs on dense
    //         lanes: 3a (lane = corner) keeps the corners that survive NMS, the InFrame test and the occupied-cell test in an LDS
    //         list; 3b gives every survivor 4 lanes (2 window rows each), sums in exact int32 and reduces inside the quad.
    if (tid == 0) n_cand = 0;                                   // the candidate list of step 1 is free: reuse it for the survivors
    __syncthreads();
    for (int li = tid; li < n; li += FT_NT) {
        const int i = list[li];
        const int ry = i >> 7, rx = i & 127;
        if (rx < 1 || rx > FT_W || ry < 1 || ry > FT_H) continue;        // ring pixels belong to neighbours
        const int x = x0 - 1 + rx, y = y0 - 1 + ry;
        const int s = sc[ry][rx];
        if (A.dbg_score) A.dbg_score[(size_t)slot * npix + (size_t)y * A.w + x] = (uint8_t)s;
        bool keep = true;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0) continue;
                const int o = sc[ry + dy][rx + dx];
                keep = keep && !(A.tie ? (o >= s) : (o > s));
            }
        if (!keep) continue;
        if (A.dbg_nms) A.dbg_nms[(size_t)slot * npix + (size_t)y * A.w + x] = 1;
        // Frame::InFrame(px,20,L) with level coords divided by 2^L again (Basic/Frame.h:67-71)
        if (!(x >= 20 * scale && x < (A.img_cols - 20) * scale && y >= 20 * scale && y < (A.img_rows - 20) * scale)) continue;
        const int gy = cell_of(y), gx = cell_of(x);
        const int k = gy * A.grid_cols + gx;
        if (k < 0 || k >= A.cells) continue;
        if (occ_fits ? occ_l[gy - gy0][gx - gx0] : A.occupied[(size_t)slot * A.cells + k]) continue;
        cand[atomicAdd(&n_cand, 1)] = (uint16_t)i;
    }
    __syncthreads();
    const int n_sel = n_cand;
    for (int t = tid; t < 4 * n_sel; t += FT_NT) {
        const int i = cand[t >> 2], part = t & 3;
        const int ry = i >> 7, rx = i & 127;
        const int x = x0 - 1 + rx, y = y0 - 1 + ry;
        // FeatureDetector::ShiTomasiScore (:467-507) on the LDS tile.  Every partial sum is an integer below 2^24 (|dx| <= 255,
        // 64 terms), so the reference's float accumulation is exact in any order: accumulate in int32, convert once
        const bool inside = !(x - 4 < 1 || x + 4 >= A.w - 1 || y - 4 < 1 || y + 4 >= A.h - 1);
        int iXX = 0, iYY = 0, iXY = 0;
        if (inside) {
            const uint8_t *c = &tile[ry + FT_Y0 - 1][rx + FT_X0 - 1];
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int xx = -4; xx < 4; ++xx) {
                    const uint8_t *q = c + (2 * part - 4 + yy) * FT_LW + xx;
                    const int dx = (int)q[1] - (int)q[-1];
                    const int dy = (int)q[FT_LW] - (int)q[-FT_LW];
                    iXX += __mul24(dx, dx); iYY += __mul24(dy, dy); iXY += __mul24(dx, dy);
                }
        }
#define FS_QUAD_SUM(v) { v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }
        FS_QUAD_SUM(iXX) FS_QUAD_SUM(iYY) FS_QUAD_SUM(iXY)
#undef FS_QUAD_SUM
        if (part != 0) continue;
        float score = 0.0f;
        if (inside) {
            float dXX = (float)iXX, dYY = (float)iYY, dXY = (float)iXY;
            dXX = __fmul_rn(dXX, 1.0f / 128.0f); dYY = __fmul_rn(dYY, 1.0f / 128.0f); dXY = __fmul_rn(dXY, 1.0f / 128.0f);
            const float tr = __fadd_rn(dXX, dYY);
            const float disc = __fsub_rn(__fmul_rn(tr, tr),
                                         __fmul_rn(4.0f, __fsub_rn(__fmul_rn(dXX, dYY), __fmul_rn(dXY, dXY))));
            score = __fmul_rn(0.5f, __fsub_rn(tr, ygz_sqrtf_cr(disc)));
        }
        const int gy = cell_of(y), gx = cell_of(x);
        const int k = gy * A.grid_cols + gx;
        const uint32_t visit = ((uint32_t)A.level << 28) | ((uint32_t)y << 14) | (uint32_t)x;
        const bool isnan_ = score != score;
        atomicMin(&A.cell_first[(size_t)slot * A.cells + k], (visit << 1) | (isnan_ ? 1u : 0u));
        if (!isnan_) {
            const unsigned long long key = ((unsigned long long)ygz_f2ord(score) << 32) | (unsigned long long)(~visit);
            atomicMax(&A.cell_best[(size_t)slot * A.cells + k], key);
        }
    }
    FS_PHASE(3);
    if (tid == 0) FS_COUNT(n);
}

// ---------------------------------------------------------------------------------------------
// All pyramid levels of the extractor in ONE launch: block x enumerates the tiles of level 0, then level 1, ... of a frame (the
// levels only meet in the per-cell atomicMax, which does not care about order); three launches cost two launch gaps and two tails.
struct FastArgsAll { FastArgs lv[YGZ_MAX_LEVELS]; int n_levels; int tiles_x[YGZ_MAX_LEVELS]; int tile_end[YGZ_MAX_LEVELS]; };

__global__ __launch_bounds__(FT_NT) void k_fast_select(FastArgsAll AA)
{
    int t_, y_, so_;
    if (!ygz_xcd_remap3(AA.lv[0].n_slots, t_, y_, so_)) return;      // frame pinned to one XCD's L2 (block-uniform)
    int L = 0;
    while (L + 1 < AA.n_levels && t_ >= AA.tile_end[L]) ++L;
    const int tl = t_ - (L ? AA.tile_end[L - 1] : 0);
    const int by_ = tl / AA.tiles_x[L], bx_ = tl - by_ * AA.tiles_x[L];
    fast_select_body(AA.lv[L], bx_, by_, so_);
}

__global__ __launch_bounds__(1024) void k_compact(const uint32_t *__restrict__ cell_first,
                                                  const unsigned long long *__restrict__ cell_best, int cells,
                                                  double *__restrict__ kp_px, int32_t *__restrict__ kp_level,
                                                  float *__restrict__ kp_score, int32_t *__restrict__ n_kp, int slot_begin)
{
    __shared__ int wave_sum[16];
    const int slot = slot_begin + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (cells + 1023) / 1024;
    const int k0 = tid * per;
    const uint32_t *cf = cell_first + (size_t)slot * cells;
    int cnt = 0;
    for (int j = 0; j < per; ++j) { const int k = k0 + j; if (k < cells && cf[k] != 0xFFFFFFFFu) ++cnt; }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wave_sum[w];
    int pos = base + incl - cnt;
    for (int j = 0; j < per; ++j) {
        const int k = k0 + j;
        if (k >= cells) break;
        const uint32_t f = cf[k];
        if (f == 0xFFFFFFFFu) continue;
        uint32_t visit; float score;
        if (f & 1u) { visit = f >> 1; score = __uint_as_float(0x7FC00000u); }
        else {
            const unsigned long long b = cell_best[(size_t)slot * cells + k];
            visit = ~(uint32_t)(b & 0xFFFFFFFFull);
            score = ygz_ord2f((uint32_t)(b >> 32));
        }
        const int L = (int)(visit >> 28), y = (int)((visit >> 14) & 0x3FFFu), x = (int)(visit & 0x3FFFu);
        const size_t o = (size_t)slot * cells + pos;
        kp_px[2 * o] = (double)(x << L); kp_px[2 * o + 1] = (double)(y << L);
        kp_level[o] = L; kp_score[o] = score;
        ++pos;
    }
    if (tid == 1023) n_kp[slot] = base + incl;
}

// ---------------------------------------------------------------------------------------------
#define DP_R   19
#define DP_W   39
#define DP_N   (DP_W * DP_W)

// canonical ORB half-widths umax[|v|] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 } (packed as nibbles in k_describe)

struct DescArgs {
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int n_levels, cells;
    const double *kp_px; const int32_t *kp_level; const int32_t *n_kp;
    float *kp_angle; uint32_t *kp_desc;
    int slot_begin, n_slots;
    int given_angle;               // !=0: kp_angle is an input (FeatureDetector::ComputeDescriptor, :591-594)
};

// cv::fastAtan2 [OpenCV 3.x polynomial]; no FMA contraction
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sin and cos of |x| < 8 (an angle in [0, 2 pi)) in double, ~1e-16 absolute: quadrant reduction with a two-part pi/2 and the
// fdlibm kernel polynomials.  The results are rounded to float by the caller (oracle: (float)cos((double)angle)); the library
// sincos carries the huge-argument reduction and costs ~4x the instructions.
__device__ __forceinline__ void ygz_sincos_small(double x, double *sn, double *cs)
{
    const double kq = rint(x * 0.63661977236758134308);                 // 2 / pi
    double r = fma(-kq, 1.57079632679489655800e+00, x);                 // pi/2 high part (exact product by fma)
    r = fma(-kq, 6.12323399573676603587e-17, r);                        // pi/2 low part
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08); ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04); ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09); pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05); pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kq & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

#define DP_P   40          // LDS row pitch of the patch (bytes): rows start dword-aligned
#define DP_KPW 8           // keypoints per wavefront (0.334 / 0.276 / 0.260 / 0.258 ms per 512 VGA frames with 1 / 4 / 8 / 16)

// IC_Angle as byte dot products (round 5).  lane = (row v = lane / 2 - 15, half hh of the row): the 16 bytes of the half row are columns
// u = -15 .. 0 (hh = 0) or 1 .. 16 (hh = 1).  Table entry of the lane: four dwords |u| of the columns inside the circle (umax[|v|], 0 elsewhere) and
// four dwords of ones for the same columns: m10 = +- sum |u| I = v_dot4_u32_u8 x 4, the row sum for m01 likewise.  Lanes 62, 63: zero.
struct DescIcTab { uint32_t v[64][8]; };
constexpr DescIcTab desc_make_ic()
{
    DescIcTab t{};
    const int umax[16] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 };
    for (int l = 0; l < 62; ++l) {
        const int v = (l >> 1) - 15, hh = l & 1, av = v < 0 ? -v : v, um = umax[av];
        for (int k = 0; k < 16; ++k) {
            const int au = hh ? 1 + k : 15 - k;
            if (au <= um && au <= 15) { t.v[l][k >> 2] |= (uint32_t)au << (8 * (k & 3)); t.v[l][4 + (k >> 2)] |= 1u << (8 * (k & 3)); }
        }
    }
    return t;
}
static __device__ const DescIcTab c_desc_ic __attribute__((aligned(16))) = desc_make_ic();
// the sampling pattern as floats (x0, y0, x1, y1 per bit): one 16-byte load per lane and bit instead of a dword and four byte -> float conversions
struct DescPatF { float v[1024]; };
constexpr DescPatF desc_make_patf()
{
    const signed char p[1024] = { YGZ_ORB_PATTERN_VALUES };
    DescPatF t{};
    for (int i = 0; i < 1024; ++i) t.v[i] = (float)p[i];
    return t;
}
static __device__ const DescPatF c_desc_patf __attribute__((aligned(16))) = desc_make_patf();

// KPW keypoints per wavefront, one after the other (the grid covers every grid cell of every frame and two thirds of them hold no keypoint:
// fewer, longer workgroups leave fewer empty ones to dispatch)
template <int KPW>
__global__ __launch_bounds__(256) void k_describe(DescArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t patch_all[4][DP_W * DP_P / 4 + 6];
    int bx, so;
    if (!ygz_xcd_remap(A.n_slots, bx, so)) return;
    const int slot = A.slot_begin + so;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nk = A.n_kp[slot];
    uint32_t *patch32 = patch_all[wv];
    const uint8_t *patch = reinterpret_cast<const uint8_t *>(patch32);
    // what a lane needs for every keypoint: its entry of the moment table, its four pattern pairs
    const uint4 ic_w = reinterpret_cast<const uint4 *>(c_desc_ic.v[lane])[0], ic_1 = reinterpret_cast<const uint4 *>(c_desc_ic.v[lane])[1];
    const float4 *patf = reinterpret_cast<const float4 *>(c_desc_patf.v) + lane;
    const float4 pt0 = patf[0], pt1 = patf[64], pt2 = patf[128], pt3 = patf[192];
    for (int it = 0; it < KPW; ++it) {
        const int kp = (bx * KPW + it) * 4 + wv;
        if (kp >= nk) return;                       // wave-uniform (kp grows with it)
        const size_t o = (size_t)slot * A.cells + kp;
        const int L = A.kp_level[o];
        const int w = A.w[L], h = A.h[L];
        const uint8_t *img = A.lvl[L] + (size_t)slot * w * h;
        // cvRound(pixel / 2^L): round half to even (FeatureDetector.cpp:514,547); the division by a power of two as the exact product with 2^-L
        const double inv_sc = __hiloint2double((1023 - L) << 20, 0);
        const int cx = __builtin_amdgcn_readfirstlane((int)rint(A.kp_px[2 * o] * inv_sc)), cy = __builtin_amdgcn_readfirstlane((int)rint(A.kp_px[2 * o + 1] * inv_sc));
        const int n = w * h;
        // 39 x 39 neighbourhood, linear addressing as center[dy*step+dx] (reads outside the level buffer are 0), rows of 40 bytes in LDS
        const int base = (cy - DP_R) * w + (cx - DP_R);
        if (base >= 0 && base + (DP_W - 1) * w + DP_P <= n) {
            // the whole window lies inside the level (all but the keypoints next to the first / last rows): lane = (row of eight, 8-byte piece of
            // five), five passes of eight rows, no per-item bounds, no index division
            const int rr = lane >> 3, j2 = lane & 7;
            if (j2 < 5) {
                const uint8_t *p = img + base + rr * w + 8 * j2;
                uint32_t lo[5], hi[5];
#pragma unroll
                for (int ps = 0; ps < 5; ++ps) { lo[ps] = 0u; hi[ps] = 0u; if (ps < 4 || rr < 7) ygz_load8(p + (size_t)(8 * ps) * w, lo[ps], hi[ps]); }
                uint2 *q = reinterpret_cast<uint2 *>(patch32 + rr * (DP_P / 4) + 2 * j2);
#pragma unroll
                for (int ps = 0; ps < 5; ++ps) if (ps < 4 || rr < 7) q[ps * 8 * (DP_P / 8)] = make_uint2(lo[ps], hi[ps]);
            }
        } else {
            // 39 rows x 10 dwords, each ONE unaligned dword load where it lies inside the buffer, byte by byte with zeros outside where it does not
            constexpr int DP_ITEMS = DP_W * 10, DP_LN = (DP_ITEMS + 63) / 64;
#pragma unroll 1
            for (int k = 0; k < DP_LN; ++k) {
                const int item = lane + 64 * k;
                if (item >= DP_ITEMS) continue;
                const int r = item / 10, j = item - 10 * r;
                const int idx0 = (cy + r - DP_R) * w + (cx - DP_R) + 4 * j;
                uint32_t v = 0u;
                if (idx0 >= 0 && idx0 + 4 <= n) v = *(ygz_gptr32u)reinterpret_cast<uintptr_t>(img + idx0);
                else for (int kk = 0; kk < 4; ++kk) { const int idx = idx0 + kk; if (idx >= 0 && idx < n) v |= (uint32_t)img[idx] << (8 * kk); }
                patch32[item] = v;                      // item == r * (DP_P / 4) + j
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        // IC_Angle (:509-537): integer moments over the circular patch, lane = (row v, half of the row): 4 LDS dwords against the lane's table entry
        int m10, m01;
        {
            const int v = (lane >> 1) - 15, hh = lane & 1;
            const uint32_t *rowp = patch32 + (v + DP_R) * (DP_P / 4) + 1 + 4 * hh;        // column 4 (u = -15) or 20 (u = 1)
            const uint32_t d0 = rowp[0], d1 = rowp[1], d2 = rowp[2], d3 = rowp[3];
            uint32_t sw = __builtin_amdgcn_udot4(d0, ic_w.x, 0u, false), s1 = __builtin_amdgcn_udot4(d0, ic_1.x, 0u, false);
            sw = __builtin_amdgcn_udot4(d1, ic_w.y, sw, false); s1 = __builtin_amdgcn_udot4(d1, ic_1.y, s1, false);
            sw = __builtin_amdgcn_udot4(d2, ic_w.z, sw, false); s1 = __builtin_amdgcn_udot4(d2, ic_1.z, s1, false);
            sw = __builtin_amdgcn_udot4(d3, ic_w.w, sw, false); s1 = __builtin_amdgcn_udot4(d3, ic_1.w, s1, false);
            m10 = hh ? (int)sw : -(int)sw;
            m01 = v * (int)s1;
        }
        m10 = ygz_wave_sum_i(m10); m01 = ygz_wave_sum_i(m01);
        const float angle = A.given_angle ? A.kp_angle[o] : fast_atan2_deg((float)m01, (float)m10);
        // ComputeOrbDescriptor (:539-578)
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float ang = __fmul_rn(angle, factorPI);
        double sn, cs;
        ygz_sincos_small((double)ang, &sn, &cs);
        const float a = (float)cs, b = (float)sn;
        unsigned long long bits[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 pt = j == 0 ? pt0 : j == 1 ? pt1 : j == 2 ? pt2 : pt3;
            const int dx0 = __float2int_rn(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)));
            const int dy0 = __float2int_rn(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)));
            const int dx1 = __float2int_rn(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)));
            const int dy1 = __float2int_rn(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)));
            const int t0 = patch[(dy0 + DP_R) * DP_P + dx0 + DP_R];
            const int t1 = patch[(dy1 + DP_R) * DP_P + dx1 + DP_R];
            bits[j] = __ballot(t0 < t1);
        }
        if (lane == 0) {
            A.kp_angle[o] = angle;
            uint4 *d = reinterpret_cast<uint4 *>(A.kp_desc + 8 * o);
            d[0] = make_uint4((uint32_t)bits[0], (uint32_t)(bits[0] >> 32), (uint32_t)bits[1], (uint32_t)(bits[1] >> 32));
            d[1] = make_uint4((uint32_t)bits[2], (uint32_t)(bits[2] >> 32), (uint32_t)bits[3], (uint32_t)(bits[3] >> 32));
        }
        __builtin_amdgcn_wave_barrier();            // the next keypoint's rows overwrite the patch: after every lane has read its bytes
    }
}

// the per-cell selection state of the slots about to be extracted, in one launch (three fill launches cost 18 us of the step's serial head)
__global__ __launch_bounds__(256) void k_detect_clear(uint32_t *__restrict__ cell_first, unsigned long long *__restrict__ cell_best,
                                                      uint8_t *__restrict__ occupied /* or null: the caller supplied the mask */, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    cell_first[i] = 0xFFFFFFFFu; cell_best[i] = 0ull;
    if (occupied) occupied[i] = 0;
}

static int detect_clear(ygz_hip_ctx *ctx, int slot_begin, int n_slots, bool clear_occupied)
{
    const size_t Cn = (size_t)ctx->cells, n = (size_t)n_slots * Cn, o = (size_t)slot_begin * Cn;
    hipLaunchKernelGGL(k_detect_clear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cell_first + o,
                       reinterpret_cast<unsigned long long *>(ctx->cell_best) + o, clear_occupied ? ctx->occupied + o : nullptr, n);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_launch_detect(ygz_hip_ctx *ctx, int slot_begin, int n_slots)
{
    FastArgsAll AA;
    AA.n_levels = ctx->prm.pyramid_levels;
    int tiles = 0;
    for (int L = 0; L < ctx->prm.pyramid_levels; ++L) {
        const size_t npix = (size_t)ctx->lw[L] * ctx->lh[L];
        if (ctx->prm.debug_maps) {
            YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->dbg_score[L] + (size_t)slot_begin * npix, 0, (size_t)n_slots * npix, ctx->stream));
            YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->dbg_nms[L] + (size_t)slot_begin * npix, 0, (size_t)n_slots * npix, ctx->stream));
        }
        FastArgs &A = AA.lv[L];
        A.img = ctx->lvl[L]; A.w = ctx->lw[L]; A.h = ctx->lh[L]; A.level = L;
        A.thr = ctx->prm.fast_threshold; A.tie = ctx->prm.nms_tie_suppress;
        A.img_cols = ctx->lw[0]; A.img_rows = ctx->lh[0];
        A.cell = ctx->prm.cell_size; A.grid_cols = ctx->grid_cols; A.cells = ctx->cells;
        A.cell_magic = ctx->prm.cell_size >= 2 ? (uint32_t)((((unsigned long long)1 << 32) + (unsigned)ctx->prm.cell_size - 1) / (unsigned)ctx->prm.cell_size) : 0u;
        A.cell_first = ctx->cell_first; A.cell_best = ctx->cell_best; A.occupied = ctx->occupied;
        A.dbg_score = ctx->prm.debug_maps ? ctx->dbg_score[L] : nullptr;
        A.dbg_nms = ctx->prm.debug_maps ? ctx->dbg_nms[L] : nullptr;
        A.slot_begin = slot_begin; A.n_slots = n_slots;
        A.dbg_cyc = nullptr;
#ifdef YGZ_FAST_TIMERS
        { void *dd = nullptr; if (getenv("YGZ_FAST_DEBUG") && ygz_scratch(ctx, SCR_GEN_0 + 1, 3 * 65536, &dd) == YGZ_OK) { if (L == 0) (void)hipMemsetAsync(dd, 0, 3 * 65536, ctx->stream); A.dbg_cyc = (long long *)dd + 8192 * L; } }
#endif
        AA.tiles_x[L] = ygz_div_up(A.w, FT_W);
        tiles += AA.tiles_x[L] * ygz_div_up(A.h, FT_H);
        AA.tile_end[L] = tiles;
    }
    for (int L = ctx->prm.pyramid_levels; L < YGZ_MAX_LEVELS; ++L) { AA.lv[L] = AA.lv[0]; AA.tiles_x[L] = 1; AA.tile_end[L] = tiles; }
    YGZ_LAUNCH(ctx, KID_FAST_SELECT, k_fast_select, dim3(tiles, 1, ygz_round_up8(n_slots)), dim3(FT_NT), AA);
    YGZ_LAUNCH(ctx, KID_COMPACT, k_compact, dim3(n_slots), dim3(1024), ctx->cell_first, ctx->cell_best, ctx->cells,
                       ctx->kp_px, ctx->kp_level, ctx->kp_score, ctx->n_kp, slot_begin);
    YGZ_HIPCHK(ctx, hipGetLastError());
#ifdef YGZ_FAST_TIMERS
    if (getenv("YGZ_FAST_DEBUG")) {
        void *dd = nullptr; static long long h[3 * 8192];
        if (ygz_scratch(ctx, SCR_GEN_0 + 1, 3 * 65536, &dd) == YGZ_OK) {
            (void)hipMemcpyAsync(h, dd, sizeof(h), hipMemcpyDeviceToHost, ctx->stream); (void)hipStreamSynchronize(ctx->stream);
            for (int L = 0; L < 3; ++L) {
                double acc[6] = { 0, 0, 0, 0, 0, 0 };
                for (int b = 0; b < 1024; ++b) for (int k = 0; k < 6; ++k) acc[k] += (double)h[8192 * L + 8 * b + k];
                const double nb = acc[5] > 0 ? acc[5] : 1;
                fprintf(stderr, "[fast-debug] level %d (slot 0): blocks %.0f corners/block %.1f; cycles/block load %.0f test %.0f score %.0f nms+select %.0f\n", L, acc[5], acc[4] / nb,
                        acc[0] / nb, acc[1] / nb, acc[2] / nb, acc[3] / nb);
            }
        }
    }
#endif
    if (ctx->describe_aside) {
        // descriptors feed the matcher only: its side stream runs them while the main stream goes on to the track sets and LK
        YgzAuxScope aux(ctx, YGZ_AUX_MATCH);
        return ygz_launch_describe(ctx, slot_begin, n_slots);
    }
    return ygz_launch_describe(ctx, slot_begin, n_slots);
}

static int launch_describe(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int given_angle)
{
    DescArgs D;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { D.lvl[L] = ctx->lvl[L]; D.w[L] = ctx->lw[L]; D.h[L] = ctx->lh[L]; }
    D.n_levels = ctx->prm.pyramid_levels; D.cells = ctx->cells;
    D.kp_px = ctx->kp_px; D.kp_level = ctx->kp_level; D.n_kp = ctx->n_kp;
    D.kp_angle = ctx->kp_angle; D.kp_desc = ctx->kp_desc; D.slot_begin = slot_begin; D.n_slots = n_slots; D.given_angle = given_angle;
    // batches: DP_KPW keypoints per wavefront; a few frames (the single-frame calls): one each, the launch is latency, not throughput
    if (n_slots >= 16) YGZ_LAUNCH(ctx, KID_DESCRIBE, k_describe<DP_KPW>, dim3(ygz_div_up(ctx->cells, 4 * DP_KPW), ygz_round_up8(n_slots)), dim3(256), D);
    else YGZ_LAUNCH(ctx, KID_DESCRIBE, k_describe<1>, dim3(ygz_div_up(ctx->cells, 4), ygz_round_up8(n_slots)), dim3(256), D);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_launch_describe(ygz_hip_ctx *ctx, int slot_begin, int n_slots) { return launch_describe(ctx, slot_begin, n_slots, 0); }

extern "C" {

int ygz_hip_detect(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *occupied)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx, 1u << YGZ_AUX_BA); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    for (int s = slot_begin; s < slot_begin + n_slots; ++s) if (!ctx->pyr_valid[s]) return YGZ_E_STATE;
    const size_t Cn = (size_t)ctx->cells;
    if (occupied) {                                           // through the page-locked arena: the caller's array may be pageable and short-lived
        void *st = ygz_stage(ctx, (size_t)n_slots * Cn);
        if (!st) return YGZ_E_HIP;
        memcpy(st, occupied, (size_t)n_slots * Cn);
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->occupied + (size_t)slot_begin * Cn, st, (size_t)n_slots * Cn, hipMemcpyHostToDevice, ctx->stream));
    }
    { const int rc = detect_clear(ctx, slot_begin, n_slots, occupied == nullptr); if (rc != YGZ_OK) return rc; }
    return ygz_launch_detect(ctx, slot_begin, n_slots);
}

int ygz_hip_keypoint_count(ygz_hip_ctx *ctx, int slot, int *n)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !n || slot < 0 || slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(n, ctx->n_kp + slot, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_get_keypoints(ygz_hip_ctx *ctx, int slot, ygz_kpt_soa *out, int capacity, int *n_out)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !out || !n_out || slot < 0 || slot >= ctx->prm.max_frames || capacity < 0) return YGZ_E_INVALID;
    // the count and `rows` rows of every requested field into ONE page-locked slice, one wait (the count is not known on the host yet, so
    // min(capacity, cells) rows are fetched: 184 KB for a whole VGA grid -- microseconds of link time against a wait + a staged pageable copy per field)
    const size_t o = (size_t)slot * ctx->cells, rows = (size_t)(capacity < ctx->cells ? capacity : ctx->cells);
    const size_t b_px = out->px ? rows * 16 : 0, b_lv = out->level ? rows * 4 : 0, b_sc = out->score ? rows * 4 : 0, b_an = out->angle ? rows * 4 : 0,
                 b_de = out->desc ? rows * 32 : 0;
    // (one gather kernel + ONE copy back: six copies on the stream before)
    YgzPack pk;
    int rc = ygz_pack_begin(ctx, &pk, 64 + b_px + b_lv + b_sc + b_an + b_de, SCR_GEN_0 + 7);
    if (rc != YGZ_OK) return rc;
    uint8_t *st = (uint8_t *)ygz_pack_add(&pk, ctx->n_kp + slot, 4);
    uint8_t *h_px = b_px ? (uint8_t *)ygz_pack_add(&pk, ctx->kp_px + 2 * o, b_px) : nullptr, *h_lv = b_lv ? (uint8_t *)ygz_pack_add(&pk, ctx->kp_level + o, b_lv) : nullptr;
    uint8_t *h_sc = b_sc ? (uint8_t *)ygz_pack_add(&pk, ctx->kp_score + o, b_sc) : nullptr, *h_an = b_an ? (uint8_t *)ygz_pack_add(&pk, ctx->kp_angle + o, b_an) : nullptr;
    uint8_t *h_de = b_de ? (uint8_t *)ygz_pack_add(&pk, ctx->kp_desc + 8 * o, b_de) : nullptr;
    if (!st || (b_px && !h_px) || (b_lv && !h_lv) || (b_sc && !h_sc) || (b_an && !h_an) || (b_de && !h_de)) return YGZ_E_CAPACITY;
    if ((rc = ygz_pack_fetch(ctx, &pk)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n = *reinterpret_cast<const int32_t *>(st);
    *n_out = n;
    if (n > capacity) return YGZ_E_CAPACITY;
    const size_t N = (size_t)n;
    if (out->px) memcpy(out->px, h_px, N * 16);
    if (out->level) memcpy(out->level, h_lv, N * 4);
    if (out->score) memcpy(out->score, h_sc, N * 4);
    if (out->angle) memcpy(out->angle, h_an, N * 4);
    if (out->desc) memcpy(out->desc, h_de, N * 32);
    return YGZ_OK;
}

static int describe_impl(ygz_hip_ctx *ctx, int slot, const double *px, const int32_t *level, const float *angle, int n)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || slot < 0 || slot >= ctx->prm.max_frames || n < 0 || (n > 0 && (!px || !level))) return YGZ_E_INVALID;
    if (n > ctx->cells) return YGZ_E_CAPACITY;
    if (!ctx->pyr_valid[slot]) return YGZ_E_STATE;
    for (int i = 0; i < n; ++i) if (level[i] < 0 || level[i] >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    const size_t o = (size_t)slot * ctx->cells;
    if (n > 0) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_px + 2 * o, px, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_level + o, level, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        if (angle) YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_angle + o, angle, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->n_kp + slot, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // &n is a stack variable
    return launch_describe(ctx, slot, 1, angle ? 1 : 0);
}

int ygz_hip_describe(ygz_hip_ctx *ctx, int slot, const double *px, const int32_t *level, int n)
{
    YgzDeviceGuard dg_(ctx);
    return describe_impl(ctx, slot, px, level, nullptr, n);
}

int ygz_hip_describe_given_angle(ygz_hip_ctx *ctx, int slot, const double *px, const int32_t *level, const float *angle, int n)
{
    YgzDeviceGuard dg_(ctx);
    if (n > 0 && !angle) return YGZ_E_INVALID;
    return describe_impl(ctx, slot, px, level, angle, n);
}

int ygz_hip_get_fast_maps(ygz_hip_ctx *ctx, int slot, int level, uint8_t *score, uint8_t *nms)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || slot < 0 || slot >= ctx->prm.max_frames || level < 0 || level >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    if (!ctx->prm.debug_maps) return YGZ_E_STATE;
    const size_t npix = (size_t)ctx->lw[level] * ctx->lh[level];
    if (score) YGZ_HIPCHK(ctx, hipMemcpyAsync(score, ctx->dbg_score[level] + slot * npix, npix, hipMemcpyDeviceToHost, ctx->stream));
    if (nms) YGZ_HIPCHK(ctx, hipMemcpyAsync(nms, ctx->dbg_nms[level] + slot * npix, npix, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
