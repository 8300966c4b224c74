/* ORACLE (test infrastructure) -- pyramidal Lucas-Kanade tracker.
 *
 * [frozen spec] cv::calcOpticalFlowPyrLK as called from src/Algorithm/Tracker.cpp:92-98:
 *   winSize 21x21, maxLevel 4, TermCriteria(COUNT+EPS, 30, 0.001),
 *   OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4 (OpenCV default).
 * OpenCV (>=3.1, unpinned, reference CMakeLists.txt:34) is not vendored and not in this
 * image; this file restates the GENERIC C++ path of modules/video/src/lkpyramid.cpp
 * (CV_SSE2/CV_NEON blocks off): buildOpticalFlowPyramid (pyrDown + REFLECT_101 border of
 * winSize), calcSharrDeriv (3/10/3 Scharr, int16, reflect-101 inside the image, zero
 * outside), LKTrackerInvoker (14-bit fixed-point bilinear weights, patch stored as
 * int16 <<5, float accumulators summed in raster order, 2x2 solve, termination on
 * |delta|^2 <= eps^2 or the 0.01 oscillation rule, err = sum|diff|/(32*win*win)).
 * PARITY UNPINNED against the real library.  The SIMD builds of OpenCV accumulate the
 * same integer products in a different float order; that difference is below the 1e-5
 * tolerance the tests use for tracks.
 * See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

void yo_klt_params_default(yo_klt_params *p)
{
    p->win = 21; p->max_level = 4; p->max_iter = 30; p->eps = 0.001;
    p->min_eig_threshold = 1e-4; p->use_initial_flow = 1;
}

static inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

/* analysis tap (tools/klt_iter_stats.py): when set, [n][YO_MAX_LEVELS] mismatch evaluations per point and level */
int *yo_klt_iter_log = NULL;

typedef struct { int w, h; uint8_t *img; int16_t *deriv; /* [h][w][2] */ } klt_level;

static inline int I_at(const klt_level *L, int x, int y)
{   /* image with its REFLECT_101 border (copyMakeBorder in buildOpticalFlowPyramid) */
    return L->img[(size_t)refl101(y, L->h) * L->w + refl101(x, L->w)];
}

static inline int D_at(const klt_level *L, int x, int y, int c)
{   /* derivative with BORDER_CONSTANT(0) border */
    if (x < 0 || y < 0 || x >= L->w || y >= L->h) return 0;
    return L->deriv[((size_t)y * L->w + x) * 2 + c];
}

/* calcSharrDeriv (scalar path) */
static void scharr_deriv(klt_level *L)
{
    const int rows = L->h, cols = L->w;
    L->deriv = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)rows * cols);
    int16_t *trow0 = (int16_t *)malloc(sizeof(int16_t) * (size_t)(cols + 2));
    int16_t *trow1 = (int16_t *)malloc(sizeof(int16_t) * (size_t)(cols + 2));
    int16_t *t0 = trow0 + 1, *t1 = trow1 + 1;
    for (int y = 0; y < rows; ++y) {
        const uint8_t *s0 = L->img + (size_t)(y > 0 ? y - 1 : rows > 1 ? 1 : 0) * cols;
        const uint8_t *s1 = L->img + (size_t)y * cols;
        const uint8_t *s2 = L->img + (size_t)(y < rows - 1 ? y + 1 : rows > 1 ? rows - 2 : 0) * cols;
        for (int x = 0; x < cols; ++x) {
            t0[x] = (int16_t)((s0[x] + s2[x]) * 3 + s1[x] * 10);
            t1[x] = (int16_t)(s2[x] - s0[x]);
        }
        const int x0 = cols > 1 ? 1 : 0, x1 = cols > 1 ? cols - 2 : 0;
        t0[-1] = t0[x0]; t0[cols] = t0[x1];
        t1[-1] = t1[x0]; t1[cols] = t1[x1];
        int16_t *d = L->deriv + (size_t)y * cols * 2;
        for (int x = 0; x < cols; ++x) {
            d[2 * x] = (int16_t)(t0[x + 1] - t0[x - 1]);
            d[2 * x + 1] = (int16_t)((t1[x + 1] + t1[x - 1]) * 3 + t1[x] * 10);
        }
    }
    free(trow0); free(trow1);
}

/* buildOpticalFlowPyramid: returns the effective maxLevel */
static int build_pyr(const uint8_t *img, int w, int h, int win, int max_level, klt_level *lv, int want_deriv)
{
    lv[0].w = w; lv[0].h = h;
    lv[0].img = (uint8_t *)malloc((size_t)w * h);
    memcpy(lv[0].img, img, (size_t)w * h);
    int sw = w, sh = h, level = 0;
    for (level = 0; level <= max_level; ++level) {
        if (level != 0) {
            lv[level].w = sw; lv[level].h = sh;
            lv[level].img = (uint8_t *)malloc((size_t)sw * sh);
            yo_pyr_down(lv[level - 1].img, lv[level - 1].w, lv[level - 1].h, lv[level].img);
        }
        lv[level].deriv = NULL;
        if (want_deriv) scharr_deriv(&lv[level]);
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= win || sh <= win) return level;
    }
    return max_level;
}

static inline int cv_floor(float v) { return (int)floorf(v); }
static inline int cv_round_f(float v) { return (int)lrint((double)v); }
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

void yo_klt_track(const uint8_t *prev, const uint8_t *next, int w, int h,
                  const float *prev_pts, float *next_pts, int n,
                  const yo_klt_params *prm, uint8_t *status, float *err)
{
    const int win = prm->win, W_BITS = 14, W_BITS1 = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    klt_level pl[YO_MAX_LEVELS], nl[YO_MAX_LEVELS];
    int max_level = prm->max_level < YO_MAX_LEVELS - 1 ? prm->max_level : YO_MAX_LEVELS - 1;
    max_level = build_pyr(prev, w, h, win, max_level, pl, 1);
    max_level = build_pyr(next, w, h, win, max_level, nl, 0);
    int max_count = prm->max_iter < 0 ? 0 : prm->max_iter > 100 ? 100 : prm->max_iter;
    double epsilon = prm->eps < 0 ? 0 : prm->eps > 10 ? 10 : prm->eps;
    epsilon *= epsilon;
    const float min_eig_thr = (float)prm->min_eig_threshold;
    const float half = (win - 1) * 0.5f;
    int16_t *IWin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win);
    int16_t *dIWin = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)win * win);
    for (int i = 0; i < n; ++i) { status[i] = 1; if (err) err[i] = 0; }
    if (!prm->use_initial_flow) memcpy(next_pts, prev_pts, sizeof(float) * 2 * (size_t)n);

    for (int level = max_level; level >= 0; --level) {
        const klt_level *I = &pl[level], *J = &nl[level];
        for (int p = 0; p < n; ++p) {
            if (yo_klt_iter_log) yo_klt_iter_log[(size_t)p * YO_MAX_LEVELS + level] = 0;
            const float s = (float)(1. / (1 << level));
            float prevx = prev_pts[2 * p] * s, prevy = prev_pts[2 * p + 1] * s;
            float nx, ny;
            if (level == max_level) {
                if (prm->use_initial_flow) { nx = next_pts[2 * p] * s; ny = next_pts[2 * p + 1] * s; }
                else { nx = prevx; ny = prevy; }
            } else { nx = next_pts[2 * p] * 2.f; ny = next_pts[2 * p + 1] * 2.f; }
            next_pts[2 * p] = nx; next_pts[2 * p + 1] = ny;
            prevx -= half; prevy -= half;
            const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) { status[p] = 0; if (err) err[p] = 0; }
                continue;
            }
            float a = prevx - ipx, b = prevy - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win; ++y)
                for (int x = 0; x < win; ++x) {
                    const int X = ipx + x, Y = ipy + y;
                    const int ival = DESCALE(I_at(I, X, Y) * iw00 + I_at(I, X + 1, Y) * iw01 +
                                             I_at(I, X, Y + 1) * iw10 + I_at(I, X + 1, Y + 1) * iw11, W_BITS1 - 5);
                    const int ixval = DESCALE(D_at(I, X, Y, 0) * iw00 + D_at(I, X + 1, Y, 0) * iw01 +
                                              D_at(I, X, Y + 1, 0) * iw10 + D_at(I, X + 1, Y + 1, 0) * iw11, W_BITS1);
                    const int iyval = DESCALE(D_at(I, X, Y, 1) * iw00 + D_at(I, X + 1, Y, 1) * iw01 +
                                              D_at(I, X, Y + 1, 1) * iw10 + D_at(I, X + 1, Y + 1, 1) * iw11, W_BITS1);
                    IWin[y * win + x] = (int16_t)ival;
                    dIWin[2 * (y * win + x)] = (int16_t)ixval;
                    dIWin[2 * (y * win + x) + 1] = (int16_t)iyval;
                    iA11 += (float)(ixval * ixval);
                    iA12 += (float)(ixval * iyval);
                    iA22 += (float)(iyval * iyval);
                }
            const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
            if (minEig < min_eig_thr || D < FLT_EPSILON) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_count; ++j) {
                const int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                if (yo_klt_iter_log) yo_klt_iter_log[(size_t)p * YO_MAX_LEVELS + level] = j + 1;
                a = nx - inx; b = ny - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        const int X = inx + x, Y = iny + y;
                        const int diff = DESCALE(I_at(J, X, Y) * iw00 + I_at(J, X + 1, Y) * iw01 +
                                                 I_at(J, X, Y + 1) * iw10 + I_at(J, X + 1, Y + 1) * iw11, W_BITS1 - 5)
                                         - IWin[y * win + x];
                        ib1 += (float)(diff * dIWin[2 * (y * win + x)]);
                        ib2 += (float)(diff * dIWin[2 * (y * win + x) + 1]);
                    }
                const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx; ny += dy;
                next_pts[2 * p] = nx + half; next_pts[2 * p + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                    next_pts[2 * p] -= dx * 0.5f; next_pts[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[p] && err && level == 0) {
                const float px_ = next_pts[2 * p] - half, py_ = next_pts[2 * p + 1] - half;
                const int inx = cv_floor(px_), iny = cv_floor(py_);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) { status[p] = 0; continue; }
                const float aa = px_ - inx, bb = py_ - iny;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float errval = 0.f;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        const int X = inx + x, Y = iny + y;
                        const int diff = DESCALE(I_at(J, X, Y) * iw00 + I_at(J, X + 1, Y) * iw01 +
                                                 I_at(J, X, Y + 1) * iw10 + I_at(J, X + 1, Y + 1) * iw11, W_BITS1 - 5)
                                         - IWin[y * win + x];
                        errval += fabsf((float)diff);
                    }
                err[p] = errval * 1.f / (32 * win * win);
            }
        }
    }
    for (int l = 0; l <= max_level; ++l) { free(pl[l].img); free(pl[l].deriv); free(nl[l].img); }
    /* levels above the effective max_level of the second build (if it shrank) */
    free(IWin); free(dIWin);
}
