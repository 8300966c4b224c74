/* ORACLE (test infrastructure) -- image pyramid.  See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <stdlib.h>
#include <string.h>

/* cv::cvtColor(_color, gray, CV_BGR2GRAY) -- Frame.cpp:27.
 * [frozen spec] OpenCV 3.1 RGB2Gray<uchar>: 14-bit fixed point,
 * B2Y=1868 G2Y=9617 R2Y=4899, rounding constant 1<<13. */
void yo_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = bgr + (size_t)y * stride;
        uint8_t *d = gray + (size_t)y * w;
        for (int x = 0; x < w; ++x, s += 3)
            d[x] = (uint8_t)((1868 * s[0] + 9617 * s[1] + 4899 * s[2] + 8192) >> 14);
    }
}

static inline int reflect101(int i, int n)
{
    /* cv::borderInterpolate(i, n, BORDER_REFLECT_101) */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* cv::pyrDown(src, dst) -- Frame.cpp:38.
 * [frozen spec] 8-bit path of OpenCV pyrDown_: separable [1 4 6 4 1], integer
 * accumulation, dst = (sum + 128) >> 8, BORDER_REFLECT_101 (the default),
 * dst size = ((w+1)/2, (h+1)/2). */
void yo_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int dy = 0; dy < dh; ++dy) {
        for (int k = 0; k < 5; ++k) {
            const int sy = reflect101(2 * dy - 2 + k, h);
            const uint8_t *s = src + (size_t)sy * w;
            int *r = rows + (size_t)k * dw;
            for (int dx = 0; dx < dw; ++dx) {
                const int x0 = reflect101(2 * dx - 2, w), x1 = reflect101(2 * dx - 1, w);
                const int x2 = reflect101(2 * dx, w), x3 = reflect101(2 * dx + 1, w);
                const int x4 = reflect101(2 * dx + 2, w);
                r[dx] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
            }
        }
        uint8_t *d = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx) {
            const int v = rows[dx] + 4 * rows[dw + dx] + 6 * rows[2 * dw + dx]
                        + 4 * rows[3 * dw + dx] + rows[4 * dw + dx];
            d[dx] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* Frame::InitFrame / CreateImagePyramid -- Frame.cpp:22-40 (level 0 = gray image) */
void yo_pyramid_build(yo_pyramid *p, const uint8_t *gray, int w, int h, int levels)
{
    memset(p, 0, sizeof(*p));
    p->levels = levels;
    p->w[0] = w; p->h[0] = h;
    p->img[0] = (uint8_t *)malloc((size_t)w * h);
    memcpy(p->img[0], gray, (size_t)w * h);
    for (int l = 1; l < levels; ++l) {
        p->w[l] = (p->w[l - 1] + 1) / 2;
        p->h[l] = (p->h[l - 1] + 1) / 2;
        p->img[l] = (uint8_t *)malloc((size_t)p->w[l] * p->h[l]);
        yo_pyr_down(p->img[l - 1], p->w[l - 1], p->h[l - 1], p->img[l]);
    }
}

void yo_pyramid_free(yo_pyramid *p)
{
    for (int l = 0; l < p->levels; ++l) { free(p->img[l]); p->img[l] = NULL; }
    p->levels = 0;
}
