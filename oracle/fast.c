/* ORACLE (test infrastructure) -- FAST-10 detect / score / 3x3 non-max suppression.
 *
 * [frozen spec] The arithmetic lives in uzh-rpg/fast (find_package(fast), reference
 * CMakeLists.txt:46, unpinned, not vendored, not in this image).  Restated from the
 * published algorithm (E. Rosten, "Machine learning for high-speed corner detection"):
 *   - 16-pixel Bresenham circle of radius 3;
 *   - a pixel p is a corner at threshold b iff >= 10 CONTIGUOUS circle pixels are all
 *     > p+b or all < p-b (strict compares);
 *   - detection scans x in [3,w-3), y in [3,h-3) in raster order (the SSE2 and plain
 *     detectors of the library return the same set);
 *   - the score is the largest b for which the pixel is still a corner, found by
 *     bisection on [threshold,255] exactly as fast_corner_score_10 does;
 *   - non-max suppression over the 8-neighbourhood of the raster-ordered corner list.
 * Call sites in the reference: src/Algorithm/FeatureDetector.cpp:366-381.
 * PARITY UNPINNED against the real library (no golden vectors exist anywhere).
 */
#include "ygz_oracle.h"
#include <stdlib.h>

static const int CIRC_DX[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int CIRC_DY[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int is_corner_10(const uint8_t *p, int stride, int b)
{
    const int cb = *p + b, c_b = *p - b;
    unsigned bright = 0, dark = 0;
    for (int i = 0; i < 16; ++i) {
        const int v = p[CIRC_DY[i] * stride + CIRC_DX[i]];
        if (v > cb) bright |= 1u << i;
        if (v < c_b) dark |= 1u << i;
    }
    /* any run of >= 10 set bits on the 16-cycle */
    for (int pass = 0; pass < 2; ++pass) {
        unsigned m = pass ? dark : bright;
        m |= m << 16;
        for (int s = 0; s < 16; ++s)
            if (((m >> s) & 0x3FFu) == 0x3FFu) return 1;
    }
    return 0;
}

int yo_fast10_detect(const uint8_t *img, int w, int h, int stride, int thr,
                     int16_t *xy, int max)
{
    int n = 0;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x)
            if (is_corner_10(img + (size_t)y * stride + x, stride, thr)) {
                if (n < max) { xy[2 * n] = (int16_t)x; xy[2 * n + 1] = (int16_t)y; }
                ++n;
            }
    return n;
}

/* fast_corner_score_10: bisection, b in [thr,255] */
void yo_fast10_score(const uint8_t *img, int stride, const int16_t *xy, int n, int thr,
                     int *scores)
{
    for (int i = 0; i < n; ++i) {
        const uint8_t *p = img + (size_t)xy[2 * i + 1] * stride + xy[2 * i];
        int bmin = thr, bmax = 255, b = (bmax + bmin) / 2;
        for (;;) {
            if (is_corner_10(p, stride, b)) bmin = b; else bmax = b;
            if (bmin == bmax - 1 || bmin == bmax) break;
            b = (bmin + bmax) / 2;
        }
        scores[i] = bmin;
    }
}

/* Closed form of the bisection result for a pixel that is a corner at some b >= 0:
 * max over 10-arcs of min(v_i - p) - 1 (bright) or min(p - v_i) - 1 (dark).
 * Returns -1 if the pixel is not a corner at any b >= 0. */
int yo_fast10_score_closed_form(const uint8_t *p, int stride)
{
    int d[32];
    for (int i = 0; i < 16; ++i) d[i] = d[i + 16] = (int)p[CIRC_DY[i] * stride + CIRC_DX[i]] - (int)*p;
    int best = -1000;
    for (int s = 0; s < 16; ++s) {
        int mn = 1000, mx = -1000;
        for (int k = 0; k < 10; ++k) { if (d[s + k] < mn) mn = d[s + k]; if (d[s + k] > mx) mx = d[s + k]; }
        if (mn > best) best = mn;          /* all brighter by at least mn */
        if (-mx > best) best = -mx;        /* all darker by at least -mx */
    }
    return best - 1 < -1 ? -1 : best - 1;
}

/* fast_nonmax_3x3 on a raster-ordered list (row_start index like the library). */
int yo_fast_nonmax_3x3(const int16_t *xy, const int *scores, int n, int tie_suppress,
                       int *nm_idx)
{
    if (n < 1) return 0;
    const int last_row = xy[2 * (n - 1) + 1];
    int *row_start = (int *)malloc(sizeof(int) * (size_t)(last_row + 2));
    for (int r = 0; r <= last_row + 1; ++r) row_start[r] = -1;
    int prev_row = -1;
    for (int i = 0; i < n; ++i)
        if (xy[2 * i + 1] != prev_row) { row_start[xy[2 * i + 1]] = i; prev_row = xy[2 * i + 1]; }

    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const int x = xy[2 * i], y = xy[2 * i + 1], s = scores[i];
        int suppressed = 0;
#define YO_BEATS(o) (tie_suppress ? ((o) >= s) : ((o) > s))
        if (i > 0 && xy[2 * (i - 1) + 1] == y && xy[2 * (i - 1)] == x - 1 && YO_BEATS(scores[i - 1])) suppressed = 1;
        if (!suppressed && i < n - 1 && xy[2 * (i + 1) + 1] == y && xy[2 * (i + 1)] == x + 1 && YO_BEATS(scores[i + 1])) suppressed = 1;
        for (int dr = -1; dr <= 1 && !suppressed; dr += 2) {
            const int r = y + dr;
            if (r < 0 || r > last_row || row_start[r] < 0) continue;
            for (int j = row_start[r]; j < n && xy[2 * j + 1] == r && xy[2 * j] <= x + 1; ++j)
                if (xy[2 * j] >= x - 1 && YO_BEATS(scores[j])) { suppressed = 1; break; }
        }
#undef YO_BEATS
        if (!suppressed) nm_idx[cnt++] = i;
    }
    free(row_start);
    return cnt;
}

int yo_detect_level_corners(const uint8_t *img, int w, int h, int thr, int tie_suppress,
                            int16_t *xy, int *scores, int max)
{
    int16_t *all = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)w * h);
    const int n = yo_fast10_detect(img, w, h, w, thr, all, w * h);
    int *sc = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *nm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    yo_fast10_score(img, w, all, n, thr, sc);
    const int m = yo_fast_nonmax_3x3(all, sc, n, tie_suppress, nm);
    for (int i = 0; i < m && i < max; ++i) {
        xy[2 * i] = all[2 * nm[i]]; xy[2 * i + 1] = all[2 * nm[i] + 1]; scores[i] = sc[nm[i]];
    }
    free(all); free(sc); free(nm);
    return m;
}
