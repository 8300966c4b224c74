/* ORACLE (test infrastructure) -- the map-building loops that follow the matcher.  See ygz_oracle.h. */
#include "ygz_oracle.h"
#include "../include/ygz_exp.h"
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

/* ------------------------------------------------------------------------------------------------------------------------------
 * The SVO depth filter of the LEGACY tree: DepthFilter::UpdateSeeds / UpdateSeed / ComputeTau (src/optimizer.cpp:537-735) and the
 * epipolar search it calls, utils::FindEpipolarMatchDirect with utils::GetWarpAffineMatrix, WarpAffine, ZMSSD<4> and the legacy
 * utils::Align2D (src/utils.cpp:37-98,102-281,330-661, include/ygz/utils.h:185-196,288-465).
 *
 * The legacy tree does not compile (it includes ygz/frame.h, ygz/camera.h, ygz/memory.h, which no longer exist); what those
 * headers defined is taken from their live successors and marked [live]:
 *   Frame::InFrame(px) and InFrame(px, border, level)  -> include/ygz/Basic/Frame.h:54-71 (border 10; the 3-argument form divides by
 *                                                          2^level and compares with the LEVEL-0 size, reproduced);
 *   PinholeCamera::focal()                              -> Basic/Camera.h:24,92: float (fx + fy) / 2;
 *   boost::math::pdf(normal_distribution<float>)        -> [frozen spec of boost 1.58 normal.hpp] e = x - mean; e *= -e; e /= 2 sd^2;
 *                                                          exp(e) / (sd * sqrt(2 pi)), all in float.
 * Defined here where the reference is undefined: the ZMSSD patch of a candidate must lie inside the level image (the legacy loop
 * would read outside it for levels > 0); matched_px is left untouched on the paths that never assign it. */
static void proj2d(const double v[3], double o[2]) { o[0] = v[0] / v[2]; o[1] = v[1] / v[2]; }       /* utils.h:26-29 */
static void cam2px_unit(const yo_camera *c, const double uv[2], double px[2])
{   /* Camera2Pixel(Vector3d(u, v, 1)) */
    px[0] = (double)c->fx * uv[0] / 1.0 + (double)c->cx; px[1] = (double)c->fy * uv[1] / 1.0 + (double)c->cy;
}

/* legacy utils::Align2D, src/utils.cpp:102-281, convergence_condition = false */
static int align2d_legacy(const uint8_t *cur, int w, int h, const uint8_t *pwb, const uint8_t *ref_patch, int n_iter, double *pu, double *pv)
{
    int converged = 0;
    double first_u = *pu, first_v = *pv;
    float dxs[64], dys[64], H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int y = 0, k = 0; y < 8; ++y) {
        const uint8_t *it = pwb + (y + 1) * 10 + 1;
        for (int x = 0; x < 8; ++x, ++it, ++k) {
            float J[3];
            J[0] = (float)(0.5 * (it[1] - it[-1])); J[1] = (float)(0.5 * (it[10] - it[-10])); J[2] = 1;
            dxs[k] = J[0]; dys[k] = J[1];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[3 * a + b] += J[a] * J[b];
        }
    }
    float Hinv[9];
    {   /* Matrix3f::inverse(): cofactors, as oracle/align.c */
        #define CF(i, j) (H[3 * (((i) + 1) % 3) + (((j) + 1) % 3)] * H[3 * (((i) + 2) % 3) + (((j) + 2) % 3)] - H[3 * (((i) + 1) % 3) + (((j) + 2) % 3)] * H[3 * (((i) + 2) % 3) + (((j) + 1) % 3)])
        const float c0 = CF(0, 0), c1 = CF(1, 0), c2 = CF(2, 0);
        const float det = (c0 * H[0] + c1 * H[3]) + c2 * H[6], invdet = 1.0f / det;
        Hinv[0] = c0 * invdet; Hinv[1] = c1 * invdet; Hinv[2] = c2 * invdet;
        Hinv[3] = CF(0, 1) * invdet; Hinv[4] = CF(1, 1) * invdet; Hinv[5] = CF(2, 1) * invdet;
        Hinv[6] = CF(0, 2) * invdet; Hinv[7] = CF(1, 2) * invdet; Hinv[8] = CF(2, 2) * invdet;
        #undef CF
    }
    float mean_diff = 0, u = (float)*pu, v = (float)*pv;
    const float min_update_squared = 0.001f;
    float update[3] = { 0, 0, 0 }, last_chi2 = 0;
    int n_chi2 = 0, error_increased = 0;
    for (int iter = 0; iter < n_iter; ++iter) {
        float chi2 = 0;
        if (isnan(u) || isnan(v)) return 0;                 /* (int)floor(NaN) breaks the loop first on x86; either way the call fails */
        const int u_r = (int)floorf(u), v_r = (int)floorf(v);
        if (u_r < 4 || v_r < 4 || u_r >= w - 4 || v_r >= h - 4) break;
        const float sx = u - u_r, sy = v - v_r;
        const float wTL = (float)((1.0 - sx) * (1.0 - sy)), wTR = (float)(sx * (1.0 - sy)), wBL = (float)((1.0 - sx) * sy), wBR = sx * sy;
        float Jres[3] = { 0, 0, 0 };
        for (int y = 0, k = 0; y < 8; ++y) {
            const uint8_t *it = cur + (v_r + y - 4) * w + u_r - 4;
            for (int x = 0; x < 8; ++x, ++it, ++k) {
                const float sp = wTL * it[0] + wTR * it[1] + wBL * it[w] + wBR * it[w + 1];
                const float res = sp - ref_patch[k] + mean_diff;
                Jres[0] -= res * dxs[k]; Jres[1] -= res * dys[k]; Jres[2] -= res;
                chi2 += res * res;
            }
        }
        for (int a = 0; a < 3; ++a) update[a] = (Hinv[3 * a] * Jres[0] + Hinv[3 * a + 1] * Jres[1]) + Hinv[3 * a + 2] * Jres[2];
        u += update[0]; v += update[1]; mean_diff += update[2];
        if (iter > 0 && chi2 > last_chi2) { error_increased = 1; break; }          /* :218-222 */
        last_chi2 = chi2; ++n_chi2;
        if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { first_u = u; first_v = v; converged = 1; break; }
    }
    *pu = u; *pv = v;
    if (converged) return 1;
    if (n_chi2 == 0) return 0;
    if (error_increased) {
        if (last_chi2 < 15000) { *pu = first_u; *pv = first_v; return 1; }
        return 0;
    }
    return last_chi2 < 15000;
}

/* utils::FindEpipolarMatchDirect, src/utils.cpp:330-661 */
static int find_epipolar_match(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref, const yo_pyramid *cur, const yo_se3 *T_cur,
                               const double px_ref[2], int octave, double d_estimate, double d_min, double d_max, double *depth,
                               double matched_px[2], int *search_level_out)
{
    yo_se3 Tri, T_cur_ref;
    yo_se3_inv(T_ref, &Tri); yo_se3_mul(T_cur, &Tri, &T_cur_ref);
    int zmssd_best = 2000 * 64;
    double uv_best[2] = { 0, 0 }, pt_ref[3], t[3], q[3], A[2], B[2];
    px2cam1(cam, px_ref, pt_ref);
    for (int k = 0; k < 3; ++k) t[k] = pt_ref[k] * d_min;
    yo_se3_act(&T_cur_ref, t, q); proj2d(q, A);
    for (int k = 0; k < 3; ++k) t[k] = pt_ref[k] * d_max;
    yo_se3_act(&T_cur_ref, t, q); proj2d(q, B);
    const double ep_dir[2] = { A[0] - B[0], A[1] - B[1] };
    /* utils::GetWarpAffineMatrix (:37-64) with pt_ref * d_estimate: Camera2World with ref->_T_c_w, World2Pixel with curr->_T_c_w */
    double Am[4];
    {
        double p3[3], pw[3], pdu[3], pdv[3], pc[2], pu[2], pv[2], c3[3];
        for (int k = 0; k < 3; ++k) p3[k] = pt_ref[k] * d_estimate;
        yo_se3_act(&Tri, p3, pw);
        const double s = (double)(1 << octave);
        const double pxu[2] = { px_ref[0] + 4.0 * s, px_ref[1] + 0.0 * s }, pxv[2] = { px_ref[0] + 0.0 * s, px_ref[1] + 4.0 * s };
        double cu[3] = { (pxu[0] - cam->cx) * p3[2] / cam->fx, (pxu[1] - cam->cy) * p3[2] / cam->fy, p3[2] };      /* Pixel2World(px, T, depth) */
        double cv_[3] = { (pxv[0] - cam->cx) * p3[2] / cam->fx, (pxv[1] - cam->cy) * p3[2] / cam->fy, p3[2] };
        yo_se3_act(&Tri, cu, pdu); yo_se3_act(&Tri, cv_, pdv);
        yo_se3_act(T_cur, pw, c3);  pc[0] = (double)cam->fx * c3[0] / c3[2] + (double)cam->cx; pc[1] = (double)cam->fy * c3[1] / c3[2] + (double)cam->cy;
        yo_se3_act(T_cur, pdu, c3); pu[0] = (double)cam->fx * c3[0] / c3[2] + (double)cam->cx; pu[1] = (double)cam->fy * c3[1] / c3[2] + (double)cam->cy;
        yo_se3_act(T_cur, pdv, c3); pv[0] = (double)cam->fx * c3[0] / c3[2] + (double)cam->cx; pv[1] = (double)cam->fy * c3[1] / c3[2] + (double)cam->cy;
        Am[0] = (pu[0] - pc[0]) / 4; Am[2] = (pu[1] - pc[1]) / 4; Am[1] = (pv[0] - pc[0]) / 4; Am[3] = (pv[1] - pc[1]) / 4;
    }
    const int search_level = yo_best_search_level(Am, 2);                                  /* GetBestSearchLevel(A_cur_ref, 2) */
    if (search_level_out) *search_level_out = search_level;
    double px_A[2], px_B[2];
    cam2px_unit(cam, A, px_A); cam2px_unit(cam, B, px_B);
    const double dxl = px_A[0] - px_B[0], dyl = px_A[1] - px_B[1];
    const double epi_length = sqrt(dxl * dxl + dyl * dyl) / (1 << search_level);
    uint8_t pwb[100], patch[64];
    yo_warp_affine(Am, ref->img[octave], ref->w[octave], ref->h[octave], px_ref, octave, search_level, 5, pwb);
    for (int y = 1; y < 9; ++y) memcpy(patch + (y - 1) * 8, pwb + y * 10 + 1, 8);
    const int cw = cur->w[search_level], ch = cur->h[search_level];
    const uint8_t *cimg = cur->img[search_level];
    double px_cur[2];
    int have_px = 0;
    if (epi_length < 2.0) {                                                                 /* :454-507 */
        px_cur[0] = (px_A[0] + px_B[0]) / 2.0; px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
        double su = px_cur[0] / (1 << search_level), sv = px_cur[1] / (1 << search_level);
        const int res = align2d_legacy(cimg, cw, ch, pwb, patch, 10, &su, &sv);
        if (res) {
            px_cur[0] = su * (1 << search_level); px_cur[1] = sv * (1 << search_level);
            double fc[3], d2;
            px2cam1(cam, px_cur, fc);
            matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
            return yo_depth_from_triangulation(&T_cur_ref, pt_ref, fc, 1e-5, depth, &d2);
        }
        matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
        return 0;
    }
    size_t n_steps = (size_t)(epi_length / 0.7);
    const double step[2] = { ep_dir[0] / n_steps, ep_dir[1] / n_steps };
    if (n_steps > 1000) return 0;
    /* PatchScore patch_score(patch): ZMSSD<4>, utils.h:288-465 */
    int sumA = 0, sumAA = 0;
    for (int r = 0; r < 64; ++r) { sumA += patch[r]; sumAA += patch[r] * patch[r]; }
    double uv[2] = { B[0] - step[0], B[1] - step[1] };
    int last_x = 0, last_y = 0;
    ++n_steps;
    for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
        double px[2];
        cam2px_unit(cam, uv, px);
        const int pxi_x = (int)(px[0] / (1 << search_level) + 0.5), pxi_y = (int)(px[1] / (1 << search_level) + 0.5);
        if (pxi_x == last_x && pxi_y == last_y) continue;
        last_x = pxi_x; last_y = pxi_y;
        {   /* [live] Frame::InFrame(pxi.cast<double>(), 8, search_level), Basic/Frame.h:67-71 */
            const double xx = (double)pxi_x / (1 << search_level), yy = (double)pxi_y / (1 << search_level);
            if (!(xx >= 8 && xx < cur->w[0] - 8 && yy >= 8 && yy < cur->h[0] - 8)) continue;
        }
        if (pxi_x - 4 < 0 || pxi_y - 4 < 0 || pxi_x + 4 > cw || pxi_y + 4 > ch) continue;    /* defined here: the patch must exist */
        const uint8_t *cp = cimg + (pxi_y - 4) * cw + (pxi_x - 4);
        int sumB = 0, sumBB = 0, sumAB = 0;
        for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
            const int c = cp[y * cw + x];
            sumB += c; sumBB += c * c; sumAB += c * patch[8 * y + x];
        }
        const int zmssd = sumAA - 2 * sumAB + sumBB - (sumA * sumA - 2 * sumA * sumB + sumB * sumB) / 64;
        if (zmssd < zmssd_best) { zmssd_best = zmssd; uv_best[0] = uv[0]; uv_best[1] = uv[1]; }
    }
    if (zmssd_best < 2000 * 64) {
        cam2px_unit(cam, uv_best, px_cur); have_px = 1;
        double su = px_cur[0] / (1 << search_level), sv = px_cur[1] / (1 << search_level);
        const int res = align2d_legacy(cimg, cw, ch, pwb, patch, 10, &su, &sv);
        if (res) {
            px_cur[0] = su * (1 << search_level); px_cur[1] = sv * (1 << search_level);
            double fc[3], d2;
            px2cam1(cam, px_cur, fc);
            matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
            return yo_depth_from_triangulation(&T_cur_ref, pt_ref, fc, 1e-5, depth, &d2);
        }
    }
    if (have_px) { matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1]; }
    return 0;
}

/* DepthFilter::ComputeTau, src/optimizer.cpp:711-726 */
static double compute_tau(const yo_se3 *T_ref_cur, const double f[3], double z, double px_error_angle)
{
    const double *t = T_ref_cur->t;
    const double a[3] = { f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2] };
    const double t_norm = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), a_norm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double alpha = acos((f[0] * t[0] + f[1] * t[1] + f[2] * t[2]) / t_norm);
    const double beta = acos((a[0] * -t[0] + a[1] * -t[1] + a[2] * -t[2]) / (t_norm * a_norm));
    const double beta_plus = beta + px_error_angle;
    const double gamma_plus = M_PI - alpha - beta_plus;
    const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
    return z_plus - z;
}

/* expf [frozen spec of libm, like sqrtf]: the double exponential of include/ygz_exp.h (plain IEEE operations, the same sequence on the
 * host and on the device) rounded once to float.  glibc's own expf is within 0.502 ulp of it; the device's native expf (1-2 ulp) is not,
 * and one ulp of the pdf is amplified ~100 x by the cancellations of the Beta update below -- so both sides use this form. */
static int g_exp_libm = 0;             /* test hook: 1 = glibc's expf (what the unmodified reference calls), to measure what the shared form changes */
static float yo_expf_cr(float x) { return g_exp_libm ? expf(x) : (float)ygz_exp_nonpos((double)x); }
void yo_set_exp_libm(int on) { g_exp_libm = on; }
float yo_expf_shared(float x) { return (float)ygz_exp_nonpos((double)x); }
/* one DepthFilter::UpdateSeed step on caller-held parameters (the test of the exponential's influence on a, b, mu, sigma2) */
static void update_seed(float x, float tau2, float *a, float *b, float *mu, float z_range, float *sigma2);
void yo_update_seed(float x, float tau2, float *a, float *b, float *mu, float z_range, float *sigma2) { update_seed(x, tau2, a, b, mu, z_range, sigma2); }

/* DepthFilter::UpdateSeed, src/optimizer.cpp:683-708 (all float) */
static void update_seed(float x, float tau2, float *a, float *b, float *mu, float z_range, float *sigma2)
{
    const float norm_scale = sqrtf(*sigma2 + tau2);
    if (isnan(norm_scale)) return;
    float e_ = x - *mu; e_ *= -e_; e_ /= 2 * norm_scale * norm_scale;
    const float pdf = yo_expf_cr(e_) / (norm_scale * sqrtf(2 * 3.14159265358979323846f));
    const float s2 = (float)(1. / (1. / *sigma2 + 1. / tau2));
    const float m = s2 * (*mu / *sigma2 + x / tau2);
    float C1 = *a / (*a + *b) * pdf;
    float C2 = (float)(*b / (*a + *b) * 1. / z_range);
    const float normalization_constant = C1 + C2;
    C1 /= normalization_constant; C2 /= normalization_constant;
    const float f = (float)(C1 * (*a + 1.) / (*a + *b + 1.) + C2 * *a / (*a + *b + 1.));
    const float e = (float)(C1 * (*a + 1.) * (*a + 2.) / ((*a + *b + 1.) * (*a + *b + 2.))
                            + C2 * *a * (*a + 1.0f) / ((*a + *b + 1.0f) * (*a + *b + 2.0f)));
    const float mu_new = C1 * m + C2 * *mu;
    *sigma2 = C1 * (s2 + m * m) + C2 * (*sigma2 + *mu * *mu) - mu_new * mu_new;
    *mu = mu_new;
    *a = (e - f) / (f - e / f);
    *b = *a * (1.0f - f) / f;
}

/* DepthFilter::UpdateSeeds for n seeds against ONE new frame.  Seed i lives in reference frame frame_idx[i] (pyramid refs[..], pose
 * T_refs[..]) whose id for the age test is frame_id[i]; kp [n][2] = cv::KeyPoint::pt (float), octave [n]; a, b, mu, z_range, sigma2 [n]
 * float, updated in place.  state [n]: 0 updated and kept, 1 behind the camera, 2 projects outside the frame, 3 no epipolar match
 * (1-3: kept unchanged), 4 erased: too old, 5 erased: converged (pos_world [n][3] = the new map point, as written at :628), 6 erased:
 * NaN.  z [n] = the depth FindEpipolarMatchDirect returned, matched_px [n][2].  Returns the number of updated seeds. */
int yo_depth_filter_update(const yo_camera *cam, const yo_pyramid *refs, const yo_se3 *T_refs, const int32_t *frame_idx,
                           const uint64_t *frame_id, int batch_counter, int max_n_kfs, double convergence_sigma2_thresh,
                           const yo_pyramid *cur, const yo_se3 *T_cur, int n, const float *kp, const int32_t *octave,
                           float *a, float *b, float *mu, const float *z_range, float *sigma2,
                           int32_t *state, double *z_out, double *matched_px, double *pos_world)
{
    const double focal_length = (double)((cam->fx + cam->fy) / 2);          /* [live] Camera.h:24 */
    const double px_noise = 1.0;
    const double px_error_angle = atan(px_noise / (2.0 * focal_length)) * 2.0;
    yo_se3 T_cur_inv;
    yo_se3_inv(T_cur, &T_cur_inv);
    int n_updates = 0;
    for (int i = 0; i < n; ++i) {
        z_out[i] = 0; matched_px[2 * i] = matched_px[2 * i + 1] = 0;
        if (((uint64_t)(int64_t)batch_counter - frame_id[i]) > (uint64_t)(int64_t)max_n_kfs) { state[i] = 4; continue; }   /* int - unsigned long: unsigned arithmetic */
        const yo_pyramid *ref = &refs[frame_idx[i]];
        const yo_se3 *T_ref = &T_refs[frame_idx[i]];
        yo_se3 T_ref_cur, T_ref_cur_inv;
        yo_se3_mul(T_ref, &T_cur_inv, &T_ref_cur); yo_se3_inv(&T_ref_cur, &T_ref_cur_inv);
        const double px_ref[2] = { (double)kp[2 * i], (double)kp[2 * i + 1] };
        double pt_ref[3], sc[3], xyz_f[3];
        px2cam1(cam, px_ref, pt_ref);
        for (int k = 0; k < 3; ++k) sc[k] = 1.0 / mu[i] * pt_ref[k];
        yo_se3_act(&T_ref_cur_inv, sc, xyz_f);
        if (xyz_f[2] < 0.0) { state[i] = 1; continue; }
        {   /* [live] frame->InFrame(Camera2Pixel(xyz_f)), border 10 */
            const double u = (double)cam->fx * xyz_f[0] / xyz_f[2] + (double)cam->cx, v = (double)cam->fy * xyz_f[1] / xyz_f[2] + (double)cam->cy;
            if (!(u >= 10 && u < cur->w[0] - 10 && v >= 10 && v < cur->h[0] - 10)) { state[i] = 2; continue; }
        }
        const float z_inv_min = mu[i] + sqrtf(sigma2[i]);
        const float z_inv_max = fmaxf(mu[i] - sqrtf(sigma2[i]), 0.00000001f);
        double z = 0;
        int sl = 0;
        if (!find_epipolar_match(cam, ref, T_ref, cur, T_cur, px_ref, octave[i], 0.9 / mu[i], 1.1 / z_inv_min, 1.0 / z_inv_max, &z,
                                 matched_px + 2 * (size_t)i, &sl)) { state[i] = 3; continue; }
        z_out[i] = z;
        const double tau = compute_tau(&T_ref_cur, pt_ref, z, px_error_angle);
        const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, z - tau) - 1.0 / (z + tau));
        update_seed((float)(1. / z), (float)(tau_inverse * tau_inverse), &a[i], &b[i], &mu[i], z_range[i], &sigma2[i]);
        ++n_updates;
        if (sqrtf(sigma2[i]) < z_range[i] / convergence_sigma2_thresh) {
            double p[3];
            for (int k = 0; k < 3; ++k) p[k] = pt_ref[k] * (1.0 / mu[i]);
            yo_se3_act(&T_cur_inv, p, pos_world + 3 * (size_t)i);          /* frame->_T_c_w.inverse() * (pt_ref / mu), :628 as written */
            state[i] = 5;
        } else if (isnan(z_inv_min)) state[i] = 6;
        else state[i] = 0;
    }
    return n_updates;
}
