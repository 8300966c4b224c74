/* ORACLE (test infrastructure) -- the map-building loops that follow the matcher.  See ygz_oracle.h. */
#include "ygz_oracle.h"
#include <math.h>
#include <string.h>

static void px2cam1(const yo_camera *c, const double px[2], double out[3])
{   /* Camera::Pixel2Camera(px) with depth = 1, Basic/Camera.h:56-62 */
    out[0] = (px[0] - c->cx) * 1.0 / c->fx; out[1] = (px[1] - c->cy) * 1.0 / c->fy; out[2] = 1.0;
}

/* The triangulation loop of LocalMapping::CreateNewMapPoints (src/Module/LocalMapping.cpp:416-495), first branch (:425-493: neither
 * feature has a map point yet), for n matched feature pairs (i1 in the current keyframe = frame 1, i2 in the neighbour = frame 2):
 *   parallax test cos >= 0.9998 (:433-435), DepthFromTriangulation(T12.inverse(), pt1, pt2) (:439-442), fea1->_depth = depth1 and
 *   Matcher::FindDirectProjection(current_kf, f2, fea1, px_curr = fea2->_pixel, level) (:445-450), fea2->_pixel = px_curr and a second
 *   triangulation (:453-457), reprojection error of pt1 * depth1 in frame 2 against 5.991 px (:460-466), then the map point
 *   Camera2World(pt1 * depth1, T1) (:478).
 * code [n]: 0 = map point created, 1 parallel rays, 2 first triangulation rejected, 3 direct projection failed, 4 second
 * triangulation rejected, 5 reprojection error.  px2 [n][2] in/out: the reference overwrites fea2->_pixel as soon as the direct
 * projection succeeds (codes 0, 4, 5).  depth1 / depth2 / pos_world are written for code 0.  Returns the number of new points. */
int yo_create_map_points(const yo_camera *cam, const yo_pyramid *pyr1, const yo_se3 *T1, const yo_pyramid *pyr2, const yo_se3 *T2, int n,
                         const double *px1, const int32_t *level1, double *px2, int32_t *code, double *depth1, double *depth2,
                         double *pos_world, int32_t *search_level)
{
    yo_se3 T2i, T1i, T12, T21;
    yo_se3_inv(T2, &T2i); yo_se3_mul(T1, &T2i, &T12);          /* SE3 T12 = _current_kf->_TCW * f2->_TCW.inverse()  (:402) */
    yo_se3_inv(&T12, &T21);                                     /* T12.inverse() */
    yo_se3_inv(T1, &T1i);
    int created = 0;
    for (int i = 0; i < n; ++i) {
        double pt1[3], pt2[3], d1 = 0, d2 = 0;
        code[i] = 0; search_level[i] = 0;
        px2cam1(cam, px1 + 2 * (size_t)i, pt1); px2cam1(cam, px2 + 2 * (size_t)i, pt2);
        const double dot = pt1[0] * pt2[0] + pt1[1] * pt2[1] + pt1[2] * pt2[2];
        const double n1 = sqrt(pt1[0] * pt1[0] + pt1[1] * pt1[1] + pt1[2] * pt1[2]), n2 = sqrt(pt2[0] * pt2[0] + pt2[1] * pt2[1] + pt2[2] * pt2[2]);
        const double cos_para_rays = dot / (n1 * n2);
        if (cos_para_rays >= 0.9998) { code[i] = 1; continue; }
        int ret = yo_depth_from_triangulation(&T21, pt1, pt2, 1e-5, &d1, &d2);
        if (!ret || d1 < 0 || d2 < 0) { code[i] = 2; continue; }
        double px_curr[2] = { px2[2 * (size_t)i], px2[2 * (size_t)i + 1] };
        int level = 0;
        ret = yo_find_direct_projection(cam, pyr1, T1, pyr2, T2, px1 + 2 * (size_t)i, d1, level1[i], px_curr, &level);
        search_level[i] = level;
        if (!ret) { code[i] = 3; continue; }
        px2[2 * (size_t)i] = px_curr[0]; px2[2 * (size_t)i + 1] = px_curr[1];
        px2cam1(cam, px_curr, pt2);
        ret = yo_depth_from_triangulation(&T21, pt1, pt2, 1e-5, &d1, &d2);
        if (!ret || d1 < 0 || d2 < 0) { code[i] = 4; continue; }
        const double ptt[3] = { pt1[0] * d1, pt1[1] * d1, pt1[2] * d1 };
        double pc[3];
        yo_se3_act(&T21, ptt, pc);
        const double rx = ((double)cam->fx * pc[0] / pc[2] + (double)cam->cx) - px_curr[0], ry = ((double)cam->fy * pc[1] / pc[2] + (double)cam->cy) - px_curr[1];
        if (sqrt(rx * rx + ry * ry) > 5.991) { code[i] = 5; continue; }
        depth1[i] = d1; depth2[i] = d2;
        yo_se3_act(&T1i, ptt, pos_world + 3 * (size_t)i);       /* Camera2World(pt1 * depth1, _current_kf->_TCW) */
        ++created;
    }
    return created;
}
