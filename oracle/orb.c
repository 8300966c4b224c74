/* ORACLE (test infrastructure) -- grid FAST extractor + ORB orientation/descriptor.
 * Restates src/Algorithm/FeatureDetector.cpp:299-596.  See ygz_oracle.h for the rules.
 *
 * Defined behaviour where the reference has UB / depends on absent libraries:
 *  D1 _umax (FeatureDetector.cpp:304-322) reads out of bounds in the reference; the
 *     oracle uses the canonical ORB table {15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3}.
 *  D2 InFrame(px,20,L) divides level-L coordinates by 2^L again (Basic/Frame.h:67-71),
 *     so level>=1 features may sit near the right/bottom border and the 31x31 patch
 *     can leave the level image.  The reference then reads center[dy*step+dx] linearly
 *     (wrapping into neighbouring rows).  The oracle keeps exactly that linear
 *     addressing and defines a read outside the level buffer [0,w*h) as 0.
 *  D3 cos/sin of the float angle (FeatureDetector.cpp:544): evaluated in double and
 *     rounded to float (the reference resolves to std::cos(float); libm-version
 *     dependent in the last ulp).
 *  D4 _cell_size is uninitialised before LoadParams(); the oracle takes it from params.
 */
#include "ygz_oracle.h"
#include "../include/ygz_orb_pattern.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

static const int UMAX[16] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 };

void yo_detect_params_default(yo_detect_params *p)
{
    p->image_width = 640; p->image_height = 480;   /* default.yaml:15-16 */
    p->cell_size = 10;                             /* default.yaml:50 */
    p->detection_threshold = 15.0;                 /* default.yaml:51 */
    p->pyramid_levels = 3;                         /* Basic/Frame.h:23, default.yaml:39 */
    p->nms_tie_suppress = 0;
}

/* FeatureDetector::ShiTomasiScore -- FeatureDetector.cpp:467-507 */
float yo_shi_tomasi(const uint8_t *img, int w, int h, int stride, int u, int v)
{
    float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
    const int halfbox_size = 4, box_size = 8, box_area = 64;
    const int x_min = u - halfbox_size, x_max = u + halfbox_size;
    const int y_min = v - halfbox_size, y_max = v + halfbox_size;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1)
        return 0.0f;
    for (int y = y_min; y < y_max; ++y) {
        const uint8_t *l = img + stride * y + x_min - 1, *r = img + stride * y + x_min + 1;
        const uint8_t *t = img + stride * (y - 1) + x_min, *b = img + stride * (y + 1) + x_min;
        for (int x = 0; x < box_size; ++x, ++l, ++r, ++t, ++b) {
            const float dx = (float)(*r - *l), dy = (float)(*b - *t);
            dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
    }
    dXX = (float)(dXX / (2.0 * box_area));
    dYY = (float)(dYY / (2.0 * box_area));
    dXY = (float)(dXY / (2.0 * box_area));
    /* float expression with std::sqrt(float) (using namespace std, Common.h:17), then 0.5* in double */
    const float tr = dXX + dYY;
    const float disc = tr * tr - 4 * (dXX * dYY - dXY * dXY);
    return (float)(0.5 * (tr - sqrtf(disc)));
}

/* cv::fastAtan2 [frozen spec: OpenCV 3.x mathfuncs, 7th-order odd polynomial] */
float yo_fast_atan2(float y, float x)
{
    static const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static inline int cv_round(double v) { return (int)lrint(v); }   /* cvRound: round-half-even */

static inline int px_linear(const uint8_t *img, int w, int h, int cx, int cy, int dx, int dy)
{
    const long idx = (long)(cy + dy) * w + (cx + dx);     /* D2: linear addressing */
    return (idx < 0 || idx >= (long)w * h) ? 0 : img[idx];
}

/* FeatureDetector::IC_Angle -- FeatureDetector.cpp:509-537 */
float yo_ic_angle(const uint8_t *img, int w, int h, double ptx, double pty)
{
    int m_01 = 0, m_10 = 0;
    const int cx = cv_round(ptx), cy = cv_round(pty);
    for (int u = -15; u <= 15; ++u) m_10 += u * px_linear(img, w, h, cx, cy, u, 0);
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0;
        const int d = UMAX[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = px_linear(img, w, h, cx, cy, u, v), vm = px_linear(img, w, h, cx, cy, u, -v);
            v_sum += vp - vm;
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return yo_fast_atan2((float)m_01, (float)m_10);
}

/* FeatureDetector::ComputeOrbDescriptor -- FeatureDetector.cpp:539-578 */
void yo_orb_descriptor(const uint8_t *img, int w, int h, double px, double py, int level,
                       float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);   /* D3 */
    const int scale = 1 << level;
    const int cx = cv_round(px / scale), cy = cv_round(py / scale);
    const signed char *pat = YGZ_ORB_PATTERN;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k, pat += 4) {
            const float x0 = (float)pat[0], y0 = (float)pat[1], x1 = (float)pat[2], y1 = (float)pat[3];
            const int t0 = px_linear(img, w, h, cx, cy, cv_round(x0 * a - y0 * b), cv_round(x0 * b + y0 * a));
            const int t1 = px_linear(img, w, h, cx, cy, cv_round(x1 * a - y1 * b), cv_round(x1 * b + y1 * a));
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* Frame::InFrame(pixel, boarder, level) -- Basic/Frame.h:67-71 (level coords divided again, D2) */
static int in_frame_level(int x, int y, int border, int level, int cols, int rows)
{
    const double s = (double)(1 << level);
    return x / s >= border && x / s < cols - border && y / s >= border && y / s < rows - border;
}

/* FeatureDetector::Detect -- FeatureDetector.cpp:345-444 */
int yo_detect(const yo_pyramid *pyr, const yo_detect_params *prm, const uint8_t *occupied,
              yo_keypoint *out)
{
    const int cell = prm->cell_size;
    const int grid_rows = (int)ceil((double)prm->image_height / cell);
    const int grid_cols = (int)ceil((double)prm->image_width / cell);
    const int ncell = grid_rows * grid_cols;
    const int thr = (int)(short)prm->detection_threshold;    /* double -> short at the call, :368 */
    uint8_t *has = (uint8_t *)calloc((size_t)ncell, 1);
    yo_keypoint *best = (yo_keypoint *)calloc((size_t)ncell, sizeof(yo_keypoint));

    for (int L = 0; L < pyr->levels && L < prm->pyramid_levels; ++L) {
        const int scale = 1 << L, w = pyr->w[L], h = pyr->h[L];
        const uint8_t *img = pyr->img[L];
        int16_t *xy = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)w * h);
        const int n = yo_fast10_detect(img, w, h, w, thr, xy, w * h);
        int *sc = (int *)malloc(sizeof(int) * (size_t)(n + 1));
        int *nm = (int *)malloc(sizeof(int) * (size_t)(n + 1));
        yo_fast10_score(img, w, xy, n, thr, sc);
        const int m = yo_fast_nonmax_3x3(xy, sc, n, prm->nms_tie_suppress, nm);
        for (int i = 0; i < m; ++i) {
            const int x = xy[2 * nm[i]], y = xy[2 * nm[i] + 1];
            if (!in_frame_level(x, y, 20, L, prm->image_width, prm->image_height)) continue;
            const int gy = (y * scale) / cell, gx = (x * scale) / cell;
            const int k = gy * grid_cols + gx;
            if (k < 0 || k >= ncell) continue;     /* ":394 k > size()" guard; also keeps memory safe */
            if (occupied && occupied[k]) continue;
            const float score = yo_shi_tomasi(img, w, h, w, x, y);
            if (has[k] && !(score > best[k].score)) continue;      /* strict >, first visited wins ties */
            has[k] = 1;
            best[k].px = (double)(x * scale); best[k].py = (double)(y * scale);
            best[k].level = L; best[k].score = score;
        }
        free(xy); free(sc); free(nm);
    }
    int cnt = 0;
    for (int k = 0; k < ncell; ++k) {
        if (!has[k]) continue;
        yo_keypoint kp = best[k];
        const int L = kp.level;
        kp.angle = yo_ic_angle(pyr->img[L], pyr->w[L], pyr->h[L], kp.px / (1 << L), kp.py / (1 << L));
        yo_orb_descriptor(pyr->img[L], pyr->w[L], pyr->h[L], kp.px, kp.py, L, kp.angle, kp.desc);
        out[cnt++] = kp;
    }
    free(has); free(best);
    return cnt;
}

/* FeatureDetector::ComputeAngleAndDescriptor -- FeatureDetector.cpp:580-588 */
void yo_describe(const yo_pyramid *pyr, yo_keypoint *kps, int n)
{
    for (int i = 0; i < n; ++i) {
        const int L = kps[i].level;
        kps[i].angle = yo_ic_angle(pyr->img[L], pyr->w[L], pyr->h[L], kps[i].px / (1 << L), kps[i].py / (1 << L));
        yo_orb_descriptor(pyr->img[L], pyr->w[L], pyr->h[L], kps[i].px, kps[i].py, L, kps[i].angle, kps[i].desc);
    }
}
