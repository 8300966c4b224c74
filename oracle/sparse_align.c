/* ORACLE (test infrastructure) -- SVO-style sparse image alignment.
 * Restates src/Algorithm/SparseImageAlign.cpp:21-238, the Gauss-Newton driver
 * include/ygz/Algorithm/NLSSolver_impl.hpp:15-89 and the Levenberg-Marquardt driver :91-212
 * (+ reset() :283-293, the defaults of NLSSolver.h:75-95), with the
 * projection Jacobian cvutils::JacobXYZ2Cam (include/ygz/Algorithm/CVUtils.h:77-99).
 * [frozen spec of Eigen] H_.ldlt().solve(Jres_) = pivoted LDLT with pseudo-inverse of D.
 * Defined behaviour: ref_patch_cache_ is uninitialised memory in the reference
 * (SparseImageAlign.cpp:33); the oracle zero-fills it.  visible_fts_ is NOT reset
 * between levels (TODO at :35) and that is reproduced.
 * See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_HALF 2
#define PATCH_SIZE 4
#define PATCH_AREA 16

int yo_ldlt6_solve(const double Hin[36], const double b[6], double x[6])
{
    enum { N = 6 };
    double m[36]; int tr[N];
    memcpy(m, Hin, sizeof(m));
    /* Eigen ldlt_inplace<Lower>::unblocked (lower triangle of a symmetric matrix) */
    for (int k = 0; k < N; ++k) {
        int piv = k; double big = fabs(m[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (fabs(m[i * N + i]) > big) { big = fabs(m[i * N + i]); piv = i; }
        tr[k] = piv;
        if (piv != k) {
            /* symmetric row/column swap restricted to the lower triangle */
            for (int j = 0; j < k; ++j) { double t = m[k * N + j]; m[k * N + j] = m[piv * N + j]; m[piv * N + j] = t; }
            for (int i = piv + 1; i < N; ++i) { double t = m[i * N + k]; m[i * N + k] = m[i * N + piv]; m[i * N + piv] = t; }
            { double t = m[k * N + k]; m[k * N + k] = m[piv * N + piv]; m[piv * N + piv] = t; }
            for (int i = k + 1; i < piv; ++i) { double t = m[i * N + k]; m[i * N + k] = m[piv * N + i]; m[piv * N + i] = t; }
        }
        if (k > 0) {
            double temp[N];
            for (int j = 0; j < k; ++j) temp[j] = m[j * N + j] * m[k * N + j];
            double s = 0; for (int j = 0; j < k; ++j) s += m[k * N + j] * temp[j];
            m[k * N + k] -= s;
            for (int i = k + 1; i < N; ++i) {
                double t = 0; for (int j = 0; j < k; ++j) t += m[i * N + j] * temp[j];
                m[i * N + k] -= t;
            }
        }
        const double akk = m[k * N + k];
        if (k == 0 && !(fabs(akk) > 0)) { for (int j = 1; j < N; ++j) tr[j] = j; break; }
        if (fabs(akk) > 0) for (int i = k + 1; i < N; ++i) m[i * N + k] /= akk;
    }
    /* solve: P b, L^-1, D^+ , L^-T, P^-1 */
    double y[N];
    memcpy(y, b, sizeof(y));
    for (int k = 0; k < N; ++k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) y[i] -= m[i * N + j] * y[j];
    double dmax = 0; for (int i = 0; i < N; ++i) if (fabs(m[i * N + i]) > dmax) dmax = fabs(m[i * N + i]);
    double tol = dmax * 2.220446049250313e-16;
    if (tol < 1.0 / 1.7976931348623157e308) tol = 1.0 / 1.7976931348623157e308;
    for (int i = 0; i < N; ++i) y[i] = (fabs(m[i * N + i]) > tol) ? y[i] / m[i * N + i] : 0.0;
    for (int i = N - 1; i >= 0; --i) for (int j = i + 1; j < N; ++j) y[i] -= m[j * N + i] * y[j];
    for (int k = N - 1; k >= 0; --k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    memcpy(x, y, sizeof(y));
    return !isnan(x[0]);       /* SparseImgAlign::solve :225-231 */
}

/* cvutils::JacobXYZ2Cam -- CVUtils.h:77-99 (translation first, sign included) */
static void jacob_xyz2cam(const double xyz[3], double J[12])
{
    const double x = xyz[0], y = xyz[1], z_inv = 1. / xyz[2], z_inv_2 = z_inv * z_inv;
    J[0] = -z_inv; J[1] = 0.0; J[2] = x * z_inv_2; J[3] = y * J[2]; J[4] = -(1.0 + x * J[2]); J[5] = y * z_inv;
    J[6] = 0.0; J[7] = -z_inv; J[8] = y * z_inv_2; J[9] = 1.0 + y * J[8]; J[10] = -J[3]; J[11] = -x * z_inv;
}

typedef struct {
    const yo_camera *cam; const yo_pyramid *ref, *cur;
    const double *px, *depth; const uint8_t *has_mp; int n;
    float *patch_cache;      /* [n][16] */
    double *jac_cache;       /* [n*16][6] */
    uint8_t *visible;        /* [n] */
    int have_cache, level;
    double H[36], Jres[6], x[6];
    size_t n_meas;
} sa_state;

/* SparseImgAlign::precomputeReferencePatches -- SparseImageAlign.cpp:59-122 */
static void precompute_reference_patches(sa_state *s)
{
    const int border = PATCH_HALF + 1, L = s->level;
    const uint8_t *ref_img = s->ref->img[L];
    const int cols = s->ref->w[L], rows = s->ref->h[L], stride = cols;
    const float scale = 1.0f / (1 << L);
    const double focal_length = (double)((s->cam->fx + s->cam->fy) / 2);     /* Camera.h:25, float */
    for (int i = 0; i < s->n; ++i) {
        const float u_ref = (float)(s->px[2 * i] * scale), v_ref = (float)(s->px[2 * i + 1] * scale);
        const int u_ref_i = (int)floorf(u_ref), v_ref_i = (int)floorf(v_ref);
        if (!s->has_mp[i] || u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i + border >= cols || v_ref_i + border >= rows)
            continue;
        s->visible[i] = 1;
        const double xyz_ref[3] = { (s->px[2 * i] - s->cam->cx) * s->depth[i] / s->cam->fx,
                                    (s->px[2 * i + 1] - s->cam->cy) * s->depth[i] / s->cam->fy, s->depth[i] };
        double fj[12];
        jacob_xyz2cam(xyz_ref, fj);
        const float su = u_ref - u_ref_i, sv = v_ref - v_ref_i;
        const float w_tl = (float)((1.0 - su) * (1.0 - sv)), w_tr = (float)(su * (1.0 - sv));
        const float w_bl = (float)((1.0 - su) * sv), w_br = su * sv;
        float *cache = s->patch_cache + PATCH_AREA * (size_t)i;
        int pc = 0;
        for (int y = 0; y < PATCH_SIZE; ++y) {
            const uint8_t *p = ref_img + (v_ref_i + y - PATCH_HALF) * stride + (u_ref_i - PATCH_HALF);
            for (int x = 0; x < PATCH_SIZE; ++x, ++p, ++pc) {
                cache[pc] = w_tl * p[0] + w_tr * p[1] + w_bl * p[stride] + w_br * p[stride + 1];
                const float dx = 0.5f * ((w_tl * p[1] + w_tr * p[2] + w_bl * p[stride + 1] + w_br * p[stride + 2])
                                         - (w_tl * p[-1] + w_tr * p[0] + w_bl * p[stride - 1] + w_br * p[stride]));
                const float dy = 0.5f * ((w_tl * p[stride] + w_tr * p[1 + stride] + w_bl * p[stride * 2] + w_br * p[stride * 2 + 1])
                                         - (w_tl * p[-stride] + w_tr * p[1 - stride] + w_bl * p[0] + w_br * p[1]));
                double *jc = s->jac_cache + 6 * ((size_t)i * PATCH_AREA + pc);
                const double f = focal_length / (1 << L);
                for (int k = 0; k < 6; ++k) jc[k] = (dx * fj[k] + dy * fj[6 + k]) * f;
            }
        }
    }
    s->have_cache = 1;
}

/* SparseImgAlign::computeResiduals -- SparseImageAlign.cpp:124-223 (unit weights) */
static double compute_residuals(sa_state *s, const yo_se3 *T_cur_from_ref, int linearize)
{
    const int L = s->level, border = PATCH_HALF + 1;
    const uint8_t *cur_img = s->cur->img[L];
    const int cols = s->cur->w[L], rows = s->cur->h[L], stride = cols;
    if (!s->have_cache) precompute_reference_patches(s);
    const float scale = 1.0f / (1 << L);
    float chi2 = 0.0f;
    for (int i = 0; i < s->n; ++i) {
        if (!s->visible[i]) continue;
        const double xyz_ref[3] = { (s->px[2 * i] - s->cam->cx) * s->depth[i] / s->cam->fx,
                                    (s->px[2 * i + 1] - s->cam->cy) * s->depth[i] / s->cam->fy, s->depth[i] };
        double xyz_cur[3];
        yo_se3_act(T_cur_from_ref, xyz_ref, xyz_cur);
        const double pu = s->cam->fx * xyz_cur[0] / xyz_cur[2] + s->cam->cx;
        const double pv = s->cam->fy * xyz_cur[1] / xyz_cur[2] + s->cam->cy;
        const float u_cur = (float)pu * scale, v_cur = (float)pv * scale;
        const int u_cur_i = (int)floorf(u_cur), v_cur_i = (int)floorf(v_cur);
        if (!(u_cur == u_cur) || !(v_cur == v_cur)) continue;   /* NaN: (int)floorf(NaN)=INT_MIN fails the test below on x86 */
        if (u_cur_i < 0 || v_cur_i < 0 || u_cur_i - border < 0 || v_cur_i - border < 0 || u_cur_i + border >= cols || v_cur_i + border >= rows)
            continue;
        const float su = u_cur - u_cur_i, sv = v_cur - v_cur_i;
        const float w_tl = (float)((1.0 - su) * (1.0 - sv)), w_tr = (float)(su * (1.0 - sv));
        const float w_bl = (float)((1.0 - su) * sv), w_br = su * sv;
        const float *cache = s->patch_cache + PATCH_AREA * (size_t)i;
        int pc = 0;
        for (int y = 0; y < PATCH_SIZE; ++y) {
            const uint8_t *p = cur_img + (v_cur_i + y - PATCH_HALF) * stride + (u_cur_i - PATCH_HALF);
            for (int x = 0; x < PATCH_SIZE; ++x, ++pc, ++p) {
                const float intensity_cur = w_tl * p[0] + w_tr * p[1] + w_bl * p[stride] + w_br * p[stride + 1];
                const float res = intensity_cur - cache[pc];
                const float weight = 1.0f;
                chi2 += res * res * weight;
                s->n_meas++;
                if (linearize) {
                    const double *J = s->jac_cache + 6 * ((size_t)i * PATCH_AREA + pc);
                    for (int a = 0; a < 6; ++a) {
                        for (int b = 0; b < 6; ++b) s->H[6 * a + b] += J[a] * J[b] * weight;
                        s->Jres[a] -= J[a] * res * weight;
                    }
                }
            }
        }
    }
    return (double)(chi2 / (float)s->n_meas);     /* float / size_t -> float */
}

static double norm_max6(const double x[6])
{
    double mx = -1;
    for (int i = 0; i < 6; ++i) if (fabs(x[i]) > mx) mx = fabs(x[i]);
    return mx;
}

size_t yo_sparse_align(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref_w,
                       const yo_pyramid *cur, yo_se3 *T_cur_w,
                       const double *px, const double *depth, const uint8_t *has_mappoint,
                       int n, int max_level, int min_level, int n_iter,
                       yo_sparse_align_stats *stats)
{
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n <= 0) return 0;                                   /* :25-29 */
    sa_state s; memset(&s, 0, sizeof(s));
    s.cam = cam; s.ref = ref; s.cur = cur; s.px = px; s.depth = depth; s.has_mp = has_mappoint; s.n = n;
    s.patch_cache = (float *)calloc((size_t)n * PATCH_AREA, sizeof(float));
    s.jac_cache = (double *)calloc((size_t)n * PATCH_AREA * 6, sizeof(double));
    s.visible = (uint8_t *)calloc((size_t)n, 1);
    /* NLLSSolver::reset -- NLSSolver_impl.hpp:283-293 */
    double chi2_ = 1e10; int stop_ = 0;
    const double eps_ = 0.000001;                           /* SparseImageAlign.cpp:18 */
    yo_se3 Tri, T;                                          /* T_cur_from_ref :37 */
    yo_se3_inv(T_ref_w, &Tri);
    yo_se3_mul(T_cur_w, &Tri, &T);
    for (int level = max_level; level >= min_level; --level) {
        s.level = level;
        memset(s.jac_cache, 0, sizeof(double) * 6 * PATCH_AREA * (size_t)n);   /* :42 */
        s.have_cache = 0;
        /* NLLSSolver::optimizeGaussNewton -- NLSSolver_impl.hpp:15-89 */
        yo_se3 old_model = T;
        int it = 0;
        for (; it < n_iter; ++it) {
            memset(s.H, 0, sizeof(s.H)); memset(s.Jres, 0, sizeof(s.Jres));
            s.n_meas = 0;
            const double new_chi2 = compute_residuals(&s, &T, 1);
            if (stats) { stats->n_iter_total++; stats->chi2_last = new_chi2; stats->n_meas_last = (int)s.n_meas; }
            if (!yo_ldlt6_solve(s.H, s.Jres, s.x)) stop_ = 1;
            if ((it > 0 && new_chi2 > chi2_) || stop_) { T = old_model; break; }
            /* update: T_new = T_old * exp(-x) -- SparseImageAlign.cpp:233-238 */
            double mx[6]; for (int k = 0; k < 6; ++k) mx[k] = -s.x[k];
            yo_se3 E, Tn;
            yo_se3_exp(mx, &E);
            yo_se3_mul(&T, &E, &Tn);
            old_model = T; T = Tn;
            chi2_ = new_chi2;
            if (norm_max6(s.x) <= eps_) { ++it; break; }
        }
        if (stats && level < YO_MAX_LEVELS) stats->iters_per_level[level] = it;
    }
    yo_se3 out;
    yo_se3_mul(&T, T_ref_w, &out);                          /* :48 */
    *T_cur_w = out;
    const size_t ret = s.n_meas / PATCH_AREA;
    free(s.patch_cache); free(s.jac_cache); free(s.visible);
    return ret;
}

/* SparseImgAlign::run with method_ = LevenbergMarquardt: NLLSSolver::optimizeLevenbergMarquardt, NLSSolver_impl.hpp:91-212, per level.
 * Reproduced as written: run() sets mu_ = 0.1 before every level (SparseImageAlign.cpp:41), so the "mu_ < 0" initialisation never runs; nu_,
 * stop_ and n_meas_ are NOT reset between levels -- the first computeResiduals of a level (:101) adds its measurements to the count the last
 * evaluation of the level before left, so chi2_ of a finer level starts as its float sum over BOTH counts; every trial linearises at the model
 * again (H_, Jres_ zeroed, :133-139), damps H_ += diag(H_) mu_ (:142), solves with Eigen's ldlt, tries T exp(-x); rho_ = chi2_ - new_chi2 > 0
 * accepts (mu_ *= max(1/3, min(1 - (2 rho_ - 1)^3, 2/3)), nu_ = 2, stop_ = |x|_max <= eps_), else mu_ *= nu_, nu_ *= 2 and after
 * n_trials_max_ = 5 failures stop_.  No weights (use_weights_ false), no prior. */
size_t yo_sparse_align_lm(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref_w,
                          const yo_pyramid *cur, yo_se3 *T_cur_w,
                          const double *px, const double *depth, const uint8_t *has_mappoint,
                          int n, int max_level, int min_level, int n_iter,
                          yo_sparse_align_stats *stats)
{
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n <= 0) return 0;
    sa_state s; memset(&s, 0, sizeof(s));
    s.cam = cam; s.ref = ref; s.cur = cur; s.px = px; s.depth = depth; s.has_mp = has_mappoint; s.n = n;
    s.patch_cache = (float *)calloc((size_t)n * PATCH_AREA, sizeof(float));
    s.jac_cache = (double *)calloc((size_t)n * PATCH_AREA * 6, sizeof(double));
    s.visible = (uint8_t *)calloc((size_t)n, 1);
    /* reset(): chi2_ = 1e10, mu_ = mu_init_ (0.01f), nu_ = nu_init_ (2), n_meas_ = 0, stop_ = false */
    double chi2_ = 1e10, mu_ = (double)0.01f, nu_ = 2.0, rho_ = 0;
    int stop_ = 0;
    const double eps_ = 0.000001;
    const int n_trials_max_ = 5;
    yo_se3 Tri, T;
    yo_se3_inv(T_ref_w, &Tri);
    yo_se3_mul(T_cur_w, &Tri, &T);
    for (int level = max_level; level >= min_level; --level) {
        s.level = level;
        mu_ = 0.1;                                             /* SparseImageAlign.cpp:41 */
        memset(s.jac_cache, 0, sizeof(double) * 6 * PATCH_AREA * (size_t)n);
        s.have_cache = 0;
        chi2_ = compute_residuals(&s, &T, 1);                  /* :101 -- n_meas_ (and H_, Jres_) carried over, see above */
        if (mu_ < 0) {                                         /* :113-120 (never true here) */
            double mx = 0;
            for (int j = 0; j < 6; ++j) if (fabs(s.H[7 * j]) > mx) mx = fabs(s.H[7 * j]);
            mu_ = 1e-4 * mx;
        }
        int it = 0;
        for (; it < n_iter; ++it) {
            rho_ = 0;
            int n_trials_ = 0;
            do {
                yo_se3 new_model = T;
                double new_chi2 = -1;
                memset(s.H, 0, sizeof(s.H)); memset(s.Jres, 0, sizeof(s.Jres));
                s.n_meas = 0;
                compute_residuals(&s, &T, 1);
                for (int j = 0; j < 6; ++j) s.H[7 * j] += s.H[7 * j] * mu_;
                if (yo_ldlt6_solve(s.H, s.Jres, s.x)) {
                    double mx[6]; for (int k = 0; k < 6; ++k) mx[k] = -s.x[k];
                    yo_se3 E;
                    yo_se3_exp(mx, &E);
                    yo_se3_mul(&T, &E, &new_model);
                    s.n_meas = 0;
                    new_chi2 = compute_residuals(&s, &new_model, 0);
                    rho_ = chi2_ - new_chi2;
                } else rho_ = -1;
                if (stats) { stats->n_iter_total++; stats->chi2_last = new_chi2; stats->n_meas_last = (int)s.n_meas; }
                if (rho_ > 0) {
                    T = new_model;
                    chi2_ = new_chi2;
                    stop_ = norm_max6(s.x) <= eps_;
                    const double c = 1. - pow(2 * rho_ - 1, 3);
                    const double f = c < 2. / 3. ? c : 2. / 3.;
                    mu_ *= (1. / 3. > f ? 1. / 3. : f);
                    nu_ = 2.;
                } else {
                    mu_ *= nu_;
                    nu_ *= 2.;
                    ++n_trials_;
                    if (n_trials_ >= n_trials_max_) stop_ = 1;
                }
            } while (!(rho_ > 0 || stop_));
            if (stop_) break;
        }
        if (stats && level < YO_MAX_LEVELS) stats->iters_per_level[level] = it;
    }
    yo_se3 out;
    yo_se3_mul(&T, T_ref_w, &out);
    *T_cur_w = out;
    const size_t ret = s.n_meas / PATCH_AREA;
    free(s.patch_cache); free(s.jac_cache); free(s.visible);
    return ret;
}

double yo_sparse_align_linearize(const yo_camera *cam, const yo_pyramid *ref,
                                 const yo_pyramid *cur, const yo_se3 *T_cur_ref,
                                 const double *px, const double *depth, const uint8_t *has_mappoint,
                                 int n, int level, uint8_t *visible, double H[36], double Jres[6],
                                 int *n_meas)
{
    sa_state s; memset(&s, 0, sizeof(s));
    s.cam = cam; s.ref = ref; s.cur = cur; s.px = px; s.depth = depth; s.has_mp = has_mappoint; s.n = n;
    s.patch_cache = (float *)calloc((size_t)n * PATCH_AREA, sizeof(float));
    s.jac_cache = (double *)calloc((size_t)n * PATCH_AREA * 6, sizeof(double));
    s.visible = visible; s.level = level;
    const double c = compute_residuals(&s, T_cur_ref, 1);
    memcpy(H, s.H, sizeof(s.H)); memcpy(Jres, s.Jres, sizeof(s.Jres));
    if (n_meas) *n_meas = (int)s.n_meas;
    free(s.patch_cache); free(s.jac_cache);
    return c;
}
