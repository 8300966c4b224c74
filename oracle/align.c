/* ORACLE (test infrastructure) -- 8x8 inverse-compositional patch alignment and the
 * direct-projection search around it.  Restates src/Algorithm/CVUtils.cpp:186-318,
 * include/ygz/Algorithm/CVUtils.h:59-71, src/Algorithm/Matcher.cpp:385-466,
 * include/ygz/Algorithm/Matcher.h:123-134, include/ygz/Basic/Camera.h:41-69.
 * [frozen spec of Eigen] Matrix3f::inverse() = cofactor formula (compute_inverse
 * size 3), 3-term sums evaluated left to right; Matrix2d::inverse() = adjugate/det.
 * See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <string.h>

static float cof3(const float m[9], int i, int j)
{   /* Eigen cofactor_3x3<i,j> on a row-major copy */
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

static void inv3f(const float m[9], float r[9])
{
    const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const float invdet = 1.0f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
    r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

/* cvutils::Align2D -- CVUtils.cpp:186-318 (scalar path; NEON dispatch :194-197 is dead on x86) */
int yo_align2d(const uint8_t *cur, int w, int h, int stride,
               const uint8_t *ref_patch_with_border, const uint8_t *ref_patch,
               int n_iter, double *pu, double *pv, float *chi2_out, int *iters_out)
{
    const int halfpatch_size_ = 4, patch_size_ = 8;
    int converged = 0;
    float ref_patch_dx[64], ref_patch_dy[64];
    float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int ref_step = patch_size_ + 2;
    for (int y = 0, k = 0; y < patch_size_; ++y) {
        const uint8_t *it = ref_patch_with_border + (y + 1) * ref_step + 1;
        for (int x = 0; x < patch_size_; ++x, ++it, ++k) {
            float J[3];
            J[0] = (float)(0.5 * (it[1] - it[-1]));
            J[1] = (float)(0.5 * (it[ref_step] - it[-ref_step]));
            J[2] = 1;
            ref_patch_dx[k] = J[0]; ref_patch_dy[k] = J[1];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[3 * a + b] += J[a] * J[b];
        }
    }
    float Hinv[9];
    inv3f(H, Hinv);
    float mean_diff = 0;
    float u = (float)*pu, v = (float)*pv;
    const float min_update_squared = (float)(0.03 * 0.03);
    float update[3] = { 0, 0, 0 };
    float chi2 = 0;
    int iter = 0;
    for (; iter < n_iter; ++iter) {
        chi2 = 0;
        if (isnan(u) || isnan(v)) break;     /* :249 is unreachable on x86: (int)floor(NaN)=INT_MIN breaks at :246 first */
        const int u_r = (int)floorf(u), v_r = (int)floorf(v);
        if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= w - halfpatch_size_ || v_r >= h - halfpatch_size_)
            break;
        const float subpix_x = u - u_r, subpix_y = v - v_r;
        const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
        const float wTR = (float)(subpix_x * (1.0 - subpix_y));
        const float wBL = (float)((1.0 - subpix_x) * subpix_y);
        const float wBR = subpix_x * subpix_y;
        float Jres[3] = { 0, 0, 0 };
        for (int y = 0, k = 0; y < patch_size_; ++y) {
            const uint8_t *it = cur + (v_r + y - halfpatch_size_) * stride + u_r - halfpatch_size_;
            for (int x = 0; x < patch_size_; ++x, ++it, ++k) {
                const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[stride] + wBR * it[stride + 1];
                const float res = search_pixel - ref_patch[k] + mean_diff;
                Jres[0] -= res * ref_patch_dx[k];
                Jres[1] -= res * ref_patch_dy[k];
                Jres[2] -= res;
                chi2 += res * res;
            }
        }
        for (int a = 0; a < 3; ++a)
            update[a] = (Hinv[3 * a] * Jres[0] + Hinv[3 * a + 1] * Jres[1]) + Hinv[3 * a + 2] * Jres[2];
        u += update[0]; v += update[1]; mean_diff += update[2];
        if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = 1; break; }
    }
    *pu = u; *pv = v;
    if (chi2_out) *chi2_out = chi2;
    if (iters_out) *iters_out = iter;
    return converged && chi2 < 20000;
}

/* Basic/Camera.h:53-62 */
static void pixel2camera(const yo_camera *c, const double px[2], double depth, double out[3])
{
    out[0] = (px[0] - c->cx) * depth / c->fx;
    out[1] = (px[1] - c->cy) * depth / c->fy;
    out[2] = depth;
}

static void camera2pixel(const yo_camera *c, const double p[3], double out[2])
{   /* Camera.h:46-51 */
    out[0] = c->fx * p[0] / p[2] + c->cx;
    out[1] = c->fy * p[1] / p[2] + c->cy;
}

/* Matcher::GetWarpAffineMatrix -- Matcher.cpp:420-436 (frame mixing restated as written) */
void yo_warp_affine_matrix(const yo_camera *cam, const yo_se3 *T_ref_w,
                           const double px_ref[2], const double pt_ref[3], int level,
                           const yo_se3 *TCR, double A[4])
{
    yo_se3 Tinv; double pw[3], pdu[3], pdv[3], q[3], pc[2], pu[2], pv[2];
    yo_se3_inv(T_ref_w, &Tinv);
    yo_se3_act(&Tinv, pt_ref, pw);                                   /* Camera2World */
    const double pxu[2] = { px_ref[0] + 4.0 * (1 << level), px_ref[1] + 0.0 * (1 << level) };
    const double pxv[2] = { px_ref[0] + 0.0 * (1 << level), px_ref[1] + 4.0 * (1 << level) };
    pixel2camera(cam, pxu, pt_ref[2], pdu);
    pixel2camera(cam, pxv, pt_ref[2], pdv);
    yo_se3_act(TCR, pw, q);  camera2pixel(cam, q, pc);               /* World2Pixel(.., TCR) */
    yo_se3_act(TCR, pdu, q); camera2pixel(cam, q, pu);
    yo_se3_act(TCR, pdv, q); camera2pixel(cam, q, pv);
    A[0] = (pu[0] - pc[0]) / 4; A[2] = (pu[1] - pc[1]) / 4;          /* col 0 */
    A[1] = (pv[0] - pc[0]) / 4; A[3] = (pv[1] - pc[1]) / 4;          /* col 1 */
}

/* Matcher::GetBestSearchLevel -- Matcher.h:123-134 */
int yo_best_search_level(const double A[4], int max_level)
{
    int search_level = 0;
    double D = A[0] * A[3] - A[2] * A[1];
    while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
    return search_level;
}

/* cvutils::GetBilateralInterpUchar -- CVUtils.h:59-71 */
static uint8_t interp_uchar(double x, double y, const uint8_t *img, int step)
{
    const double xx = x - floor(x), yy = y - floor(y);
    const uint8_t *d = img + (int)y * step + (int)x;
    return (uint8_t)((1 - xx) * (1 - yy) * d[0] + xx * (1 - yy) * d[1] + (1 - xx) * yy * d[step] + xx * yy * d[step + 1]);
}

/* Matcher::WarpAffine -- Matcher.cpp:438-466 */
void yo_warp_affine(const double A[4], const uint8_t *img_ref, int w, int h,
                    const double px_ref[2], int level_ref, int search_level,
                    int half_patch_size, uint8_t *patch)
{
    const int patch_size = half_patch_size * 2;
    const double det = A[0] * A[3] - A[2] * A[1];
    const double invdet = 1.0 / det;
    const double R[4] = { A[3] * invdet, -A[1] * invdet, -A[2] * invdet, A[0] * invdet };   /* ARC */
    const double rx = px_ref[0] / (1 << level_ref), ry = px_ref[1] / (1 << level_ref);
    for (int y = 0; y < patch_size; ++y)
        for (int x = 0; x < patch_size; ++x, ++patch) {
            double ppx = x - half_patch_size, ppy = y - half_patch_size;
            ppx *= (1 << search_level); ppy *= (1 << search_level);
            const double qx = (R[0] * ppx + R[1] * ppy) + rx, qy = (R[2] * ppx + R[3] * ppy) + ry;
            if (qx < 0 || qy < 0 || qx >= w - 1 || qy >= h - 1) *patch = 0;
            else *patch = interp_uchar(qx, qy, img_ref, w);
        }
}

/* Matcher::FindDirectProjection(Frame*,Frame*,Feature*,...) -- Matcher.cpp:385-417.
 * The MapPoint overload (:356-383) differs only in how the depth is obtained. */
static int fdp_body(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                    const yo_pyramid *cur, const yo_se3 *T_cur,
                    const double px_ref[2], double depth_ref, int level_ref,
                    double px_cur[2], int *search_level_out)
{
    double pt_ref[3], A[4];
    yo_se3 Tri, TCR;
    pixel2camera(cam, px_ref, depth_ref, pt_ref);
    yo_se3_inv(T_ref, &Tri);
    yo_se3_mul(T_cur, &Tri, &TCR);
    yo_warp_affine_matrix(cam, T_ref, px_ref, pt_ref, level_ref, &TCR, A);
    const int search_level = yo_best_search_level(A, cur->levels - 1);
    uint8_t pwb[100], patch[64];
    yo_warp_affine(A, ref->img[level_ref], ref->w[level_ref], ref->h[level_ref], px_ref, level_ref,
                   search_level, 5, pwb);
    for (int y = 1; y < 9; ++y) memcpy(patch + (y - 1) * 8, pwb + y * 10 + 1, 8);
    double u = px_cur[0] / (1 << search_level), v = px_cur[1] / (1 << search_level);
    const int ok = yo_align2d(cur->img[search_level], cur->w[search_level], cur->h[search_level],
                              cur->w[search_level], pwb, patch, 10, &u, &v, NULL, NULL);
    px_cur[0] = u * (1 << search_level); px_cur[1] = v * (1 << search_level);
    if (search_level_out) *search_level_out = search_level;
    /* Frame::InFrame(px, 10) -- Basic/Frame.h:54-58 (level-0 size) */
    if (!(px_cur[0] >= 10 && px_cur[0] < cur->w[0] - 10 && px_cur[1] >= 10 && px_cur[1] < cur->h[0] - 10))
        return 0;
    return ok;
}

int yo_find_direct_projection(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                              const yo_pyramid *cur, const yo_se3 *T_cur,
                              const double px_ref[2], double depth_ref, int level_ref,
                              double px_cur[2], int *search_level_out)
{
    if (depth_ref < 0) return 0;                                   /* Matcher.cpp:388-392 */
    return fdp_body(cam, ref, T_ref, cur, T_cur, px_ref, depth_ref, level_ref, px_cur, search_level_out);
}

/* the same for n features of one reference frame (the loop of LocalMapping::ProjectMapPoints / the bench's CPU leg, kept in C
 * so that the timing is the algorithm's and not the binding's) */
int yo_find_direct_projection_n(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                                const yo_pyramid *cur, const yo_se3 *T_cur, int n,
                                const double *px_ref, const double *depth_ref, const int32_t *level_ref,
                                double *px_cur, int32_t *search_level, uint8_t *ok)
{
    int good = 0;
    for (int i = 0; i < n; ++i) {
        int sl = 0;
        ok[i] = (uint8_t)yo_find_direct_projection(cam, ref, T_ref, cur, T_cur, px_ref + 2 * (size_t)i, depth_ref[i], level_ref[i],
                                                   px_cur + 2 * (size_t)i, &sl);
        search_level[i] = sl;
        good += ok[i];
    }
    return good;
}

/* Matcher::FindDirectProjection(Frame*,Frame*,MapPoint*,...) -- Matcher.cpp:356-383: the depth is the z of the map point in
 * the reference keyframe, World2Camera(mp->_pos_world, ref->_TCW)[2] (:362), and is NOT tested for its sign;
 * px_ref / level_ref are those of the map point's observation in that keyframe (:360-361). */
int yo_find_direct_projection_mp(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                                 const yo_pyramid *cur, const yo_se3 *T_cur, const double pos_world[3],
                                 const double px_ref[2], int level_ref, double px_cur[2], int *search_level_out)
{
    double pc[3];
    yo_se3_act(T_ref, pos_world, pc);
    return fdp_body(cam, ref, T_ref, cur, T_cur, px_ref, pc[2], level_ref, px_cur, search_level_out);
}

/* LocalMapping::FindCandidates + ProjectMapPoints -- src/Module/LocalMapping.cpp:47-120.
 * FindCandidates (:47-79): every non-bad local map point is projected into the current frame; behind the camera or outside
 * InFrame(px,20) -> not in view; otherwise each of its observations in a local keyframe becomes a candidate carrying the
 * projected pixel.  ProjectMapPoints (:81-120): candidates are visited in order; a map point that already matched is skipped;
 * FindDirectProjection (MapPoint overload) refines the projection; the first success of a point wins.
 * The reference visits candidates in std::map<Feature*,...> order, i.e. by heap address; here the order is the caller's
 * (cand_* arrays), which is the only reproducible statement of it.
 * Inputs: K local keyframes (pyramid + pose), P points (pos_world [P][3], point_bad [P]), C candidates
 * (cand_point, cand_kf, cand_px_ref [C][2], cand_level).  Outputs per point: in_view, px_proj [P][2] (valid when in view),
 * match_cand (candidate index or -1), px_match [P][2], match_level.  Returns the number of matched points. */
int yo_track_local_map(const yo_camera *cam, const yo_pyramid *kf_pyr, const yo_se3 *kf_T, int K,
                       const yo_pyramid *cur, const yo_se3 *T_cur,
                       const double *pos_world, const uint8_t *point_bad, int P,
                       const int32_t *cand_point, const int32_t *cand_kf, const double *cand_px_ref, const int32_t *cand_level, int C,
                       uint8_t *in_view, double *px_proj, int32_t *match_cand, double *px_match, int32_t *match_level)
{
    int matched = 0;
    for (int p = 0; p < P; ++p) {
        in_view[p] = 0; match_cand[p] = -1; match_level[p] = 0;
        px_proj[2 * p] = px_proj[2 * p + 1] = px_match[2 * p] = px_match[2 * p + 1] = 0.0;
        if (point_bad && point_bad[p]) continue;
        double pc[3], px[2];
        yo_se3_act(T_cur, pos_world + 3 * (size_t)p, pc);               /* World2Camera, Camera.h:41-43 */
        camera2pixel(cam, pc, px);
        px_proj[2 * p] = px[0]; px_proj[2 * p + 1] = px[1];
        if (pc[2] < 0 || !(px[0] >= 20 && px[0] < cur->w[0] - 20 && px[1] >= 20 && px[1] < cur->h[0] - 20)) continue;   /* :59 */
        in_view[p] = 1;
    }
    for (int c = 0; c < C; ++c) {
        const int p = cand_point[c], kf = cand_kf[c];
        if (p < 0 || p >= P || kf < 0 || kf >= K || !in_view[p] || match_cand[p] >= 0) continue;
        double px[2] = { px_proj[2 * p], px_proj[2 * p + 1] };
        int level = 0;
        if (yo_find_direct_projection_mp(cam, &kf_pyr[kf], &kf_T[kf], cur, T_cur, pos_world + 3 * (size_t)p,
                                         cand_px_ref + 2 * (size_t)c, cand_level[c], px, &level)) {
            match_cand[p] = c; px_match[2 * p] = px[0]; px_match[2 * p + 1] = px[1]; match_level[p] = level; ++matched;
        }
    }
    return matched;
}

/* The hand-over between VisualOdometry::TrackRefFrame and LocalMapping::TrackLocalMap for the features of ONE reference frame
 * (src/Module/VisualOdometry.cpp:293, src/Module/LocalMapping.cpp:47-79): the map point of feature i is its back-projection
 * Camera2World(Pixel2Camera(px, depth), T_ref) (Camera.h:45-62,70-72 -- what the reference stores in MapPoint::_pos_world when the
 * point is created); FindCandidates projects it with the current pose and drops it when it lies behind the camera or outside
 * InFrame(px, 20).  Features without depth (<= 0) have no map point.  Returns the number of candidates. */
int yo_track_candidates(const yo_camera *cam, const yo_se3 *T_ref, const yo_se3 *T_cur, const double *px_ref, const double *depth,
                        int n, int w, int h, double *pos_world, double *px_pred, uint8_t *cand)
{
    yo_se3 Tri;
    yo_se3_inv(T_ref, &Tri);
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        cand[i] = 0;
        px_pred[2 * i] = px_ref[2 * i]; px_pred[2 * i + 1] = px_ref[2 * i + 1];
        pos_world[3 * i] = pos_world[3 * i + 1] = pos_world[3 * i + 2] = 0.0;
        if (!(depth[i] > 0)) continue;
        double pr[3], pc[3], px[2];
        pixel2camera(cam, px_ref + 2 * (size_t)i, depth[i], pr);
        yo_se3_act(&Tri, pr, pos_world + 3 * (size_t)i);
        yo_se3_act(T_cur, pos_world + 3 * (size_t)i, pc);
        camera2pixel(cam, pc, px);
        px_pred[2 * i] = px[0]; px_pred[2 * i + 1] = px[1];
        if (pc[2] < 0 || !(px[0] >= 20 && px[0] < w - 20 && px[1] >= 20 && px[1] < h - 20)) continue;
        cand[i] = 1; ++cnt;
    }
    return cnt;
}

/* cvutils::DepthFromTriangulation -- include/ygz/Algorithm/CVUtils.h:18-38, with Eigen's evaluation order
 * [frozen spec of Eigen: 2x2 inverse = adjugate * (1/det); (-inv * A^T) is formed before it multiplies t].
 * Returns 1 and |depth| of the ray in the reference / the search frame, or 0 when det(A^T A) < determinant_th. */
int yo_depth_from_triangulation(const yo_se3 *T_search_ref, const double f_ref[3], const double f_cur[3],
                                double determinant_th, double *depth1, double *depth2)
{
    double R[9], a0[3];
    yo_quat_to_R(T_search_ref->q, R);
    for (int i = 0; i < 3; ++i) a0[i] = R[3 * i] * f_ref[0] + R[3 * i + 1] * f_ref[1] + R[3 * i + 2] * f_ref[2];
    const double a1[3] = { -f_cur[0], -f_cur[1], -f_cur[2] };
    const double m00 = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2], m01 = a0[0] * a1[0] + a0[1] * a1[1] + a0[2] * a1[2];
    const double m10 = a1[0] * a0[0] + a1[1] * a0[1] + a1[2] * a0[2], m11 = a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2];
    const double det = m00 * m11 - m10 * m01;
    if (det < determinant_th) return 0;
    const double invdet = 1.0 / det;
    const double i00 = -(m11 * invdet), i01 = -(-m01 * invdet), i10 = -(-m10 * invdet), i11 = -(m00 * invdet);     /* -AtA.inverse() */
    double M[6];                                                                                                   /* (-inv) * A^T : 2x3 */
    for (int c = 0; c < 3; ++c) { M[c] = i00 * a0[c] + i01 * a1[c]; M[3 + c] = i10 * a0[c] + i11 * a1[c]; }
    const double *t = T_search_ref->t;
    *depth1 = fabs(M[0] * t[0] + M[1] * t[1] + M[2] * t[2]);
    *depth2 = fabs(M[3] * t[0] + M[4] * t[1] + M[5] * t[2]);
    return 1;
}
