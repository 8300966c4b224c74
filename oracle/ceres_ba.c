/* ORACLE (test infrastructure) -- the ceres-side bundle-adjustment rows and the two
 * non-linear least-squares drivers the reference runs around its edge stacks.
 *
 *  - the three auto-differentiated functors, evaluated the way ceres evaluates them: with forward-mode dual numbers
 *    ("Jets") through the very expressions of include/ygz/Ceres/CeresReprojectionError.h:33-69,
 *    CeresReprojectionErrorPoseOnly.h:27-58 and CeresReprojectionErrorPointOnly.h:45-77.  A pose is [t(3); angle-axis(3)]
 *    (BA.cpp:96-99,190-193), the update is plain addition, the observation is in normalised image coordinates.
 *  - ceres::AngleAxisRotatePoint [frozen spec of ceres-solver include/ceres/rotation.h -- NOT in /root/reference:
 *    theta^2 > DBL_EPSILON: p cos + (w x p) sin + w (w.p)(1-cos), else p + aa x p].
 *  - ceres::HuberLoss + Corrector [frozen spec of ceres-solver loss_function.cc / corrector.cc: for Huber rho'' <= 0, so
 *    residual and Jacobian are both scaled by sqrt(rho')].
 *  - ceres::Solve with the options the reference leaves at their defaults (BA.cpp:219-226,372-375): trust-region
 *    Levenberg-Marquardt [frozen spec of ceres-solver 1.13 trust_region_minimizer.cc + levenberg_marquardt_strategy.cc +
 *    trust_region_step_evaluator.cc; unpinned: the reference names no ceres version].  The linear solve is an exact Schur
 *    elimination of the points followed by a dense Cholesky (what DENSE_SCHUR does; every ceres linear solver computes
 *    the same LM step up to rounding).  TwoViewBACeres asks for DOGLEG (BA.cpp:59); it is run with the same LM
 *    strategy here -- recorded divergence.
 *  - g2o's OptimizationAlgorithmLevenberg as ba::LocalBAG2O drives it (BA.cpp:390-395,501-502) [frozen spec of g2o
 *    optimization_algorithm_levenberg.cpp; unpinned], around yo_ba_linearize.
 *
 * parity unpinned: neither ceres nor g2o exist in this image and the reference has no asserting test for these paths;
 * the checks available are mathematical (tests/test_oracle_golden.py: Jets against central differences, zero-noise
 * fixtures of test/test_local_ba.cpp:9-37 converge to the ground truth).  See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ Jets (9 duals) */
#define NJ 9
typedef struct { double a; double v[NJ]; } jet;

static jet j_const(double a) { jet r; r.a = a; memset(r.v, 0, sizeof(r.v)); return r; }
static jet j_var(double a, int k) { jet r = j_const(a); if (k >= 0) r.v[k] = 1.0; return r; }
static jet j_add(jet x, jet y) { jet r; r.a = x.a + y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
static jet j_sub(jet x, jet y) { jet r; r.a = x.a - y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
static jet j_mul(jet x, jet y) { jet r; r.a = x.a * y.a; for (int i = 0; i < NJ; ++i) r.v[i] = y.a * x.v[i] + x.a * y.v[i]; return r; }
static jet j_div(jet x, jet y)
{   /* ceres jet.h operator/: a/b, (da - a/b db)/b */
    jet r; const double inv = 1.0 / y.a, q = x.a * inv;
    r.a = q; for (int i = 0; i < NJ; ++i) r.v[i] = (x.v[i] - q * y.v[i]) * inv; return r;
}
static jet j_sqrt(jet x) { jet r; r.a = sqrt(x.a); const double s = 1.0 / (2.0 * r.a); for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] * s; return r; }
static jet j_cos(jet x) { jet r; r.a = cos(x.a); const double s = -sin(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = s * x.v[i]; return r; }
static jet j_sin(jet x) { jet r; r.a = sin(x.a); const double c = cos(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = c * x.v[i]; return r; }
static jet j_scale(double s, jet x) { jet r; r.a = s * x.a; for (int i = 0; i < NJ; ++i) r.v[i] = s * x.v[i]; return r; }

/* ceres::AngleAxisRotatePoint<T> [frozen spec, ceres/rotation.h] */
static void j_angle_axis_rotate(const jet aa[3], const jet pt[3], jet out[3])
{
    const jet theta2 = j_add(j_add(j_mul(aa[0], aa[0]), j_mul(aa[1], aa[1])), j_mul(aa[2], aa[2]));
    if (theta2.a > DBL_EPSILON) {
        const jet theta = j_sqrt(theta2), costheta = j_cos(theta), sintheta = j_sin(theta);
        const jet theta_inverse = j_div(j_const(1.0), theta);
        const jet w[3] = { j_mul(aa[0], theta_inverse), j_mul(aa[1], theta_inverse), j_mul(aa[2], theta_inverse) };
        const jet wxp[3] = { j_sub(j_mul(w[1], pt[2]), j_mul(w[2], pt[1])),
                             j_sub(j_mul(w[2], pt[0]), j_mul(w[0], pt[2])),
                             j_sub(j_mul(w[0], pt[1]), j_mul(w[1], pt[0])) };
        const jet tmp = j_mul(j_add(j_add(j_mul(w[0], pt[0]), j_mul(w[1], pt[1])), j_mul(w[2], pt[2])),
                              j_sub(j_const(1.0), costheta));
        for (int i = 0; i < 3; ++i)
            out[i] = j_add(j_add(j_mul(pt[i], costheta), j_mul(wxp[i], sintheta)), j_mul(w[i], tmp));
    } else {
        const jet wxp[3] = { j_sub(j_mul(aa[1], pt[2]), j_mul(aa[2], pt[1])),
                             j_sub(j_mul(aa[2], pt[0]), j_mul(aa[0], pt[2])),
                             j_sub(j_mul(aa[0], pt[1]), j_mul(aa[1], pt[0])) };
        for (int i = 0; i < 3; ++i) out[i] = j_add(pt[i], wxp[i]);
    }
}

void yo_ceres_rotate_point(const double aa[3], const double p[3], double out[3])
{
    jet a[3], q[3], o[3];
    for (int i = 0; i < 3; ++i) { a[i] = j_const(aa[i]); q[i] = j_const(p[i]); }
    j_angle_axis_rotate(a, q, o);
    for (int i = 0; i < 3; ++i) out[i] = o[i].a;
}

/* CeresReprojectionError::operator() (CeresReprojectionError.h:33-69), weight 1 (SetWeight has no caller).
 * r [2]; Jpose [2][6] = d r / d [t; aa]; Jpt [2][3] = d r / d p_w; *p_z = depth in the camera.  The PoseOnly / PointOnly
 * functors are the same expression with one block held constant (their Jacobian is the matching sub-block). */
void yo_ceres_edge(const double pose[6], const double pt[3], const double obs_n[2],
                   double r[2], double Jpose[12], double Jpt[6], double *p_z)
{
    jet P[6], X[3], rot[3], p[3];
    for (int i = 0; i < 6; ++i) P[i] = j_var(pose[i], i);
    for (int i = 0; i < 3; ++i) X[i] = j_var(pt[i], 6 + i);
    for (int i = 0; i < 3; ++i) rot[i] = P[i + 3];
    j_angle_axis_rotate(rot, X, p);
    p[0] = j_add(p[0], P[0]); p[1] = j_add(p[1], P[1]); p[2] = j_add(p[2], P[2]);
    const jet r0 = j_scale(1.0, j_sub(j_const(obs_n[0]), j_div(p[0], p[2])));
    const jet r1 = j_scale(1.0, j_sub(j_const(obs_n[1]), j_div(p[1], p[2])));
    r[0] = r0.a; r[1] = r1.a;
    for (int i = 0; i < 6; ++i) { Jpose[i] = r0.v[i]; Jpose[6 + i] = r1.v[i]; }
    for (int i = 0; i < 3; ++i) { Jpt[i] = r0.v[6 + i]; Jpt[3 + i] = r1.v[6 + i]; }
    if (p_z) *p_z = p[2].a;
}

/* ------------------------------------------------------------------------------------------------ linearisation */
static int edge_on(const yo_ceres_problem *pb, int e) { return !pb->edge_enable || pb->edge_enable[e]; }
static int pose_free(const yo_ceres_problem *pb, int k) { return !(pb->pose_fixed && pb->pose_fixed[k]); }
static int point_free(const yo_ceres_problem *pb, int l) { return !(pb->point_fixed && pb->point_fixed[l]); }

/* Evaluate the program at (poses, points).  cost = 1/2 sum rho(|r|^2).  Blocks in the layout of yo_ba_linearize
 * (H = J^T J, b = -J^T r, loss-corrected; zero for constant blocks).  Jx_out [E][12] / Jp_out [E][6] / rc_out [E][2] are the
 * corrected per-edge Jacobians and residuals.  Returns 0, or -1 when a functor reports failure (PoseOnly behind camera). */
int yo_ceres_linearize(const yo_ceres_problem *pb, const double *poses, const double *points, double *cost,
                       double *Hpp, double *bp, double *Hll, double *bl, double *Hpl,
                       double *Jx_out, double *Jp_out, double *rc_out)
{
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges;
    if (Hpp) memset(Hpp, 0, sizeof(double) * 36 * (size_t)K);
    if (bp) memset(bp, 0, sizeof(double) * 6 * (size_t)K);
    if (Hll) memset(Hll, 0, sizeof(double) * 9 * (size_t)P);
    if (bl) memset(bl, 0, sizeof(double) * 3 * (size_t)P);
    if (Hpl) memset(Hpl, 0, sizeof(double) * 18 * (size_t)E);
    if (Jx_out) memset(Jx_out, 0, sizeof(double) * 12 * (size_t)E);
    if (Jp_out) memset(Jp_out, 0, sizeof(double) * 6 * (size_t)E);
    if (rc_out) memset(rc_out, 0, sizeof(double) * 2 * (size_t)E);
    double total = 0;
    for (int e = 0; e < E; ++e) {
        if (!edge_on(pb, e)) continue;                       /* _enable == false: residual 0, Jacobian 0 */
        const int ip = pb->edge_pose[e], il = pb->edge_point[e];
        double r[2], Jx[12], Jp[6], z;
        yo_ceres_edge(poses + 6 * (size_t)ip, points + 3 * (size_t)il, pb->obs_n + 2 * (size_t)e, r, Jx, Jp, &z);
        if (pb->fail_behind_camera && z < 0) return -1;      /* CeresReprojectionErrorPoseOnly.h:48-51 */
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho0 = s, rho1 = 1.0;
        const double a = pb->edge_huber ? pb->edge_huber[e] : 0.0;
        if (a > 0 && s > a * a) {                            /* HuberLoss::Evaluate */
            const double rr = sqrt(s);
            rho0 = 2 * a * rr - a * a;
            rho1 = a / rr; if (rho1 < DBL_MIN) rho1 = DBL_MIN;
        }
        total += 0.5 * rho0;
        const double sq = sqrt(rho1);                        /* Corrector: rho'' <= 0 branch */
        for (int i = 0; i < 12; ++i) Jx[i] *= sq;
        for (int i = 0; i < 6; ++i) Jp[i] *= sq;
        r[0] *= sq; r[1] *= sq;
        const int pf = pose_free(pb, ip), lf = point_free(pb, il);
        if (!pf) memset(Jx, 0, sizeof(Jx));
        if (!lf) memset(Jp, 0, sizeof(Jp));
        if (Jx_out) memcpy(Jx_out + 12 * (size_t)e, Jx, sizeof(Jx));
        if (Jp_out) memcpy(Jp_out + 6 * (size_t)e, Jp, sizeof(Jp));
        if (rc_out) { rc_out[2 * (size_t)e] = r[0]; rc_out[2 * (size_t)e + 1] = r[1]; }
        if (lf && Hll) {
            double *hl = Hll + 9 * (size_t)il, *gl = bl + 3 * (size_t)il;
            for (int x = 0; x < 3; ++x) {
                for (int y = 0; y < 3; ++y) hl[3 * x + y] += Jp[x] * Jp[y] + Jp[3 + x] * Jp[3 + y];
                gl[x] += -(Jp[x] * r[0] + Jp[3 + x] * r[1]);
            }
        }
        if (pf && Hpp) {
            double *hp = Hpp + 36 * (size_t)ip, *gp = bp + 6 * (size_t)ip;
            for (int x = 0; x < 6; ++x) {
                for (int y = 0; y < 6; ++y) hp[6 * x + y] += Jx[x] * Jx[y] + Jx[6 + x] * Jx[6 + y];
                gp[x] += -(Jx[x] * r[0] + Jx[6 + x] * r[1]);
                if (Hpl && lf) for (int y = 0; y < 3; ++y)
                    Hpl[18 * (size_t)e + 3 * x + y] = Jx[x] * Jp[y] + Jx[6 + x] * Jp[3 + y];
            }
        }
    }
    if (cost) *cost = total;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ block solver */
static int inv3(const double *m, double *r)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (!(fabs(det) > 0) || !isfinite(det)) return 0;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    r[3] = c1 * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    r[6] = c2 * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 1;
}

static int chol_solve(double *A, double *b, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    return 1;
}

/* Solve [Hpp + diag(dp), Hpl; Hpl^T, Hll + diag(dl)] [xp; xl] = [bp; bl] with the points eliminated first.
 * pose_free / point_free: 0 = constant block (x = 0).  dp [K][6], dl [P][3] are the damping terms added to the diagonal.
 * Returns 1, or 0 when a pivot block is not positive definite. */
int yo_ba_schur_solve(int K, int P, int E, const int32_t *edge_pose, const int32_t *edge_point,
                      const uint8_t *pose_free_, const uint8_t *point_free_,
                      const double *Hpp, const double *Hll, const double *Hpl, const double *bp, const double *bl,
                      const double *dp, const double *dl, double *xp, double *xl)
{
    int *fidx = (int *)malloc(sizeof(int) * (size_t)(K > 0 ? K : 1)), Kf = 0;
    for (int k = 0; k < K; ++k) fidx[k] = pose_free_[k] ? Kf++ : -1;
    const int n = 6 * Kf;
    int *off = (int *)calloc((size_t)P + 2, sizeof(int)), *lst = (int *)malloc(sizeof(int) * (size_t)(E > 0 ? E : 1));
    for (int e = 0; e < E; ++e) off[edge_point[e] + 1]++;
    for (int l = 0; l < P; ++l) off[l + 1] += off[l];
    { int *c = (int *)malloc(sizeof(int) * (size_t)(P + 1)); memcpy(c, off, sizeof(int) * (size_t)(P + 1));
      for (int e = 0; e < E; ++e) lst[c[edge_point[e]]++] = e;
      free(c); }
    double *S = (double *)calloc((size_t)(n > 0 ? n : 1) * (size_t)(n > 0 ? n : 1), sizeof(double));
    double *bs = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    double *Dinv = (double *)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double));
    int ok = 1;
    memset(xp, 0, sizeof(double) * 6 * (size_t)K);
    memset(xl, 0, sizeof(double) * 3 * (size_t)P);
    for (int k = 0; k < K; ++k) if (fidx[k] >= 0) {
        const int a = fidx[k];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) S[(size_t)(6 * a + r) * n + 6 * a + c] = Hpp[(size_t)k * 36 + 6 * r + c];
            S[(size_t)(6 * a + r) * n + 6 * a + r] += dp[(size_t)k * 6 + r];
            bs[6 * a + r] = bp[(size_t)k * 6 + r];
        }
    }
    for (int l = 0; l < P && ok; ++l) {
        if (!point_free_[l]) continue;
        double D[9]; memcpy(D, Hll + (size_t)l * 9, sizeof(D));
        D[0] += dl[(size_t)l * 3]; D[4] += dl[(size_t)l * 3 + 1]; D[8] += dl[(size_t)l * 3 + 2];
        double *Di = Dinv + (size_t)l * 9;
        if (!inv3(D, Di)) { ok = 0; break; }
        for (int ci = off[l]; ci < off[l + 1]; ++ci) {
            const int ei = lst[ci], a = fidx[edge_pose[ei]];
            if (a < 0) continue;
            const double *Bi = Hpl + (size_t)ei * 18;
            double BD[18];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c)
                BD[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
            for (int r = 0; r < 6; ++r)
                bs[6 * a + r] -= BD[3 * r] * bl[(size_t)l * 3] + BD[3 * r + 1] * bl[(size_t)l * 3 + 1] + BD[3 * r + 2] * bl[(size_t)l * 3 + 2];
            for (int cj = off[l]; cj < off[l + 1]; ++cj) {
                const int ej = lst[cj], b2 = fidx[edge_pose[ej]];
                if (b2 < 0) continue;
                const double *Bj = Hpl + (size_t)ej * 18;
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
                    S[(size_t)(6 * a + r) * n + 6 * b2 + c] -= BD[3 * r] * Bj[3 * c] + BD[3 * r + 1] * Bj[3 * c + 1] + BD[3 * r + 2] * Bj[3 * c + 2];
            }
        }
    }
    if (ok && n > 0) ok = chol_solve(S, bs, n);
    if (ok) {
        for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int r = 0; r < 6; ++r) xp[(size_t)k * 6 + r] = bs[6 * fidx[k] + r];
        for (int l = 0; l < P; ++l) {
            if (!point_free_[l]) continue;
            double r3[3] = { bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2] };
            for (int ci = off[l]; ci < off[l + 1]; ++ci) {
                const int ei = lst[ci], k = edge_pose[ei];
                if (fidx[k] < 0) continue;
                const double *Bi = Hpl + (size_t)ei * 18;
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) r3[c] -= Bi[3 * r + c] * xp[(size_t)k * 6 + r];
            }
            const double *Di = Dinv + (size_t)l * 9;
            for (int c = 0; c < 3; ++c) xl[(size_t)l * 3 + c] = Di[3 * c] * r3[0] + Di[3 * c + 1] * r3[1] + Di[3 * c + 2] * r3[2];
        }
    }
    free(fidx); free(off); free(lst); free(S); free(bs); free(Dinv);
    return ok;
}

/* ------------------------------------------------------------------------------------------------ ceres::Solve */
void yo_ceres_default_options(yo_ceres_options *o)
{   /* ceres Solver::Options defaults (solver.h) */
    o->max_num_iterations = 50;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1; o->max_num_consecutive_invalid_steps = 5;
    o->trust_region_strategy = YO_CERES_LEVENBERG_MARQUARDT;
}

/* opt->trust_region_strategy = YO_CERES_DOGLEG: DoglegStrategy with TRADITIONAL_DOGLEG [frozen spec of ceres-solver 1.13
 * internal/ceres/dogleg_strategy.cc, restated from its published algorithm; parity unpinned like the rest of this file].  In the coordinates
 * scaled by diagonal_ = sqrt(clamp(|J col|^2, min_lm_diagonal, max_lm_diagonal)): gradient_ = D^-1 J^T r, the Cauchy point -alpha_ gradient_ with
 * alpha_ = |gradient_|^2 / |J D^-1 gradient_|^2, the Gauss-Newton step from (J^T J + mu_ D^2) y = J^T r (mu_ from 1e-8, x 10 while the factorisation
 * fails, up to 1), and the step = Gauss-Newton if it lies inside the radius, else the scaled gradient direction if even the Cauchy point lies outside,
 * else the point of the segment between them on the boundary; StepAccepted: radius x 0.5 below a step quality of 0.25, max(radius, 3 |step|) above
 * 0.75, mu_ = max(1e-8, mu_ / 5); StepRejected: radius x 0.5 and the same two vectors interpolated again (reuse_); StepIsInvalid: mu_ x 10. */
int yo_ceres_solve(yo_ceres_problem *pb, const yo_ceres_options *opt, yo_ceres_summary *sum)
{
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges;
    const size_t nK = (size_t)(K > 0 ? K : 1), nP = (size_t)(P > 0 ? P : 1), nE = (size_t)(E > 0 ? E : 1);
    double *Hpp = (double *)malloc(8 * 36 * nK), *bp = (double *)malloc(8 * 6 * nK), *Hll = (double *)malloc(8 * 9 * nP),
           *bl = (double *)malloc(8 * 3 * nP), *Hpl = (double *)malloc(8 * 18 * nE), *Jx = (double *)malloc(8 * 12 * nE),
           *Jp = (double *)malloc(8 * 6 * nE), *rc = (double *)malloc(8 * 2 * nE);
    double *sHpp = (double *)malloc(8 * 36 * nK), *sbp = (double *)malloc(8 * 6 * nK), *sHll = (double *)malloc(8 * 9 * nP),
           *sbl = (double *)malloc(8 * 3 * nP), *sHpl = (double *)malloc(8 * 18 * nE);
    double *scp = (double *)malloc(8 * 6 * nK), *scl = (double *)malloc(8 * 3 * nP), *dp = (double *)malloc(8 * 6 * nK),
           *dl = (double *)malloc(8 * 3 * nP), *xp = (double *)malloc(8 * 6 * nK), *xl = (double *)malloc(8 * 3 * nP);
    double *cposes = (double *)malloc(8 * 6 * nK), *cpoints = (double *)malloc(8 * 3 * nP);
    const int dogleg = opt->trust_region_strategy == YO_CERES_DOGLEG;
    double *dgp = (double *)calloc(6 * nK, 8), *dgl = (double *)calloc(3 * nP, 8), *grp = (double *)calloc(6 * nK, 8), *grl = (double *)calloc(3 * nP, 8),
           *gnp = (double *)calloc(6 * nK, 8), *gnl = (double *)calloc(3 * nP, 8);     /* diagonal_, gradient_, gauss_newton_step_ (pose / point parts) */
    double dl_mu = 1e-8, dl_alpha = 0, dl_step_norm = 0;
    int dl_reuse = 0, dl_gn_ok = 0;
    uint8_t *pfree = (uint8_t *)malloc(nK), *lfree = (uint8_t *)malloc(nP);
    for (int k = 0; k < K; ++k) pfree[k] = (uint8_t)pose_free(pb, k);
    for (int l = 0; l < P; ++l) lfree[l] = (uint8_t)point_free(pb, l);
    yo_ceres_summary S; memset(&S, 0, sizeof(S));
    double x_cost = 0, radius = opt->initial_trust_region_radius, decrease_factor = 2.0, x_norm = 0, gmax = 0;
    int invalid_run = 0, term = YO_CERES_NO_CONVERGENCE;

#define X_NORM_GRAD()                                                                                         \
    do { double s2 = 0; gmax = 0;                                                                              \
         for (int k = 0; k < K; ++k) if (pfree[k]) for (int d = 0; d < 6; ++d) {                              \
             const double v = pb->poses[(size_t)k * 6 + d]; s2 += v * v;                                       \
             const double g = fabs(bp[(size_t)k * 6 + d]); if (g > gmax) gmax = g; }                           \
         for (int l = 0; l < P; ++l) if (lfree[l]) for (int d = 0; d < 3; ++d) {                              \
             const double v = pb->points[(size_t)l * 3 + d]; s2 += v * v;                                      \
             const double g = fabs(bl[(size_t)l * 3 + d]); if (g > gmax) gmax = g; }                           \
         x_norm = sqrt(s2); } while (0)

    /* IterationZero */
    if (yo_ceres_linearize(pb, pb->poses, pb->points, &x_cost, Hpp, bp, Hll, bl, Hpl, Jx, Jp, rc) != 0) { term = YO_CERES_FAILURE; goto done; }
    S.initial_cost = x_cost;
    for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d)   /* EstimateScale: 1 / (1 + sqrt(|J col|^2)), frozen at iteration 0 */
        scp[(size_t)k * 6 + d] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(Hpp[(size_t)k * 36 + 7 * d])) : 1.0;
    for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d)
        scl[(size_t)l * 3 + d] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(Hll[(size_t)l * 9 + 4 * d])) : 1.0;
    X_NORM_GRAD();

    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (S.iterations >= opt->max_num_iterations) { term = YO_CERES_NO_CONVERGENCE; break; }
        if (gmax <= opt->gradient_tolerance) { term = YO_CERES_GRADIENT_TOLERANCE; break; }
        if (radius <= opt->min_trust_region_radius) { term = YO_CERES_MIN_RADIUS; break; }
        ++S.iterations;
        /* the column-scaled system (the minimizer scales the Jacobian's columns before it hands it to the strategy) */
        int valid = 1;
        if (!dogleg || !dl_reuse) {
            for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) {
                for (int c = 0; c < 6; ++c) sHpp[(size_t)k * 36 + 6 * r + c] = Hpp[(size_t)k * 36 + 6 * r + c] * scp[(size_t)k * 6 + r] * scp[(size_t)k * 6 + c];
                sbp[(size_t)k * 6 + r] = bp[(size_t)k * 6 + r] * scp[(size_t)k * 6 + r];
            }
            for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) sHll[(size_t)l * 9 + 3 * r + c] = Hll[(size_t)l * 9 + 3 * r + c] * scl[(size_t)l * 3 + r] * scl[(size_t)l * 3 + c];
                sbl[(size_t)l * 3 + r] = bl[(size_t)l * 3 + r] * scl[(size_t)l * 3 + r];
            }
            for (int e = 0; e < E; ++e) {
                const int ip = pb->edge_pose[e], il = pb->edge_point[e];
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c)
                    sHpl[(size_t)e * 18 + 3 * r + c] = Hpl[(size_t)e * 18 + 3 * r + c] * scp[(size_t)ip * 6 + r] * scl[(size_t)il * 3 + c];
            }
        }
        if (!dogleg) {
            /* LevenbergMarquardtStrategy::ComputeStep */
            for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) {
                double dg = sHpp[(size_t)k * 36 + 7 * r];
                dg = dg < opt->min_lm_diagonal ? opt->min_lm_diagonal : (dg > opt->max_lm_diagonal ? opt->max_lm_diagonal : dg);
                dp[(size_t)k * 6 + r] = dg / radius;
            }
            for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) {
                double dg = sHll[(size_t)l * 9 + 4 * r];
                dg = dg < opt->min_lm_diagonal ? opt->min_lm_diagonal : (dg > opt->max_lm_diagonal ? opt->max_lm_diagonal : dg);
                dl[(size_t)l * 3 + r] = dg / radius;
            }
            valid = yo_ba_schur_solve(K, P, E, pb->edge_pose, pb->edge_point, pfree, lfree, sHpp, sHll, sHpl, sbp, sbl, dp, dl, xp, xl);
        } else {
            /* DoglegStrategy::ComputeStep */
            if (!dl_reuse) {
                dl_reuse = 1;
                double g2 = 0;
                for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) {
                    double dg = sHpp[(size_t)k * 36 + 7 * r];
                    dg = dg < opt->min_lm_diagonal ? opt->min_lm_diagonal : (dg > opt->max_lm_diagonal ? opt->max_lm_diagonal : dg);
                    dgp[(size_t)k * 6 + r] = sqrt(dg);
                    grp[(size_t)k * 6 + r] = pfree[k] ? -sbp[(size_t)k * 6 + r] / dgp[(size_t)k * 6 + r] : 0.0;       /* J^T r = -b */
                    g2 += grp[(size_t)k * 6 + r] * grp[(size_t)k * 6 + r];
                }
                for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) {
                    double dg = sHll[(size_t)l * 9 + 4 * r];
                    dg = dg < opt->min_lm_diagonal ? opt->min_lm_diagonal : (dg > opt->max_lm_diagonal ? opt->max_lm_diagonal : dg);
                    dgl[(size_t)l * 3 + r] = sqrt(dg);
                    grl[(size_t)l * 3 + r] = lfree[l] ? -sbl[(size_t)l * 3 + r] / dgl[(size_t)l * 3 + r] : 0.0;
                    g2 += grl[(size_t)l * 3 + r] * grl[(size_t)l * 3 + r];
                }
                /* ComputeCauchyPoint: Jg = J_scaled (D^-1 gradient_) through the per-edge Jacobians (J_scaled = J diag(sc)) */
                double jg2 = 0;
                for (int e = 0; e < E; ++e) {
                    const int ip = pb->edge_pose[e], il = pb->edge_point[e];
                    const double *jx = Jx + 12 * (size_t)e, *jp = Jp + 6 * (size_t)e;
                    for (int a = 0; a < 2; ++a) {
                        double m = 0;
                        for (int c = 0; c < 6; ++c) m += jx[6 * a + c] * (scp[(size_t)ip * 6 + c] * grp[(size_t)ip * 6 + c] / dgp[(size_t)ip * 6 + c]);
                        for (int c = 0; c < 3; ++c) m += jp[3 * a + c] * (scl[(size_t)il * 3 + c] * grl[(size_t)il * 3 + c] / dgl[(size_t)il * 3 + c]);
                        jg2 += m * m;
                    }
                }
                dl_alpha = g2 / jg2;
                /* ComputeGaussNewtonStep */
                dl_gn_ok = 0;
                while (dl_mu < 1.0) {
                    for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) dp[(size_t)k * 6 + r] = dgp[(size_t)k * 6 + r] * dgp[(size_t)k * 6 + r] * dl_mu;
                    for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) dl[(size_t)l * 3 + r] = dgl[(size_t)l * 3 + r] * dgl[(size_t)l * 3 + r] * dl_mu;
                    int ok = yo_ba_schur_solve(K, P, E, pb->edge_pose, pb->edge_point, pfree, lfree, sHpp, sHll, sHpl, sbp, sbl, dp, dl, xp, xl);
                    if (ok) {
                        for (int k = 0; k < K && ok; ++k) for (int d = 0; d < 6; ++d) if (!isfinite(xp[(size_t)k * 6 + d])) ok = 0;
                        for (int l = 0; l < P && ok; ++l) for (int d = 0; d < 3; ++d) if (!isfinite(xl[(size_t)l * 3 + d])) ok = 0;
                    }
                    if (!ok) { dl_mu *= 10.0; continue; }
                    dl_gn_ok = 1;
                    break;
                }
                if (dl_gn_ok) {      /* the scaled Gauss-Newton step D (-(J^T J)^-1 g): the solve above already carries the sign (b = -g) */
                    for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) gnp[(size_t)k * 6 + d] = pfree[k] ? xp[(size_t)k * 6 + d] * dgp[(size_t)k * 6 + d] : 0.0;
                    for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) gnl[(size_t)l * 3 + d] = lfree[l] ? xl[(size_t)l * 3 + d] * dgl[(size_t)l * 3 + d] : 0.0;
                }
            }
            valid = dl_gn_ok;
            if (valid) {
                /* ComputeTraditionalDoglegStep */
                double g2 = 0, n2 = 0, gdn = 0;
                for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) { const double g = grp[(size_t)k * 6 + d], nn = gnp[(size_t)k * 6 + d]; g2 += g * g; n2 += nn * nn; gdn += g * nn; }
                for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) { const double g = grl[(size_t)l * 3 + d], nn = gnl[(size_t)l * 3 + d]; g2 += g * g; n2 += nn * nn; gdn += g * nn; }
                const double gradient_norm = sqrt(g2), gauss_newton_norm = sqrt(n2);
                double cg, cn;                                   /* step (scaled) = cg * gradient_ + cn * gauss_newton_step_ */
                if (gauss_newton_norm <= radius) { cg = 0.0; cn = 1.0; dl_step_norm = gauss_newton_norm; }
                else if (gradient_norm * dl_alpha >= radius) { cg = -(radius / gradient_norm); cn = 0.0; dl_step_norm = radius; }
                else {
                    const double b_dot_a = -dl_alpha * gdn, a_squared_norm = (dl_alpha * gradient_norm) * (dl_alpha * gradient_norm);
                    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gauss_newton_norm * gauss_newton_norm;
                    const double c = b_dot_a - a_squared_norm;
                    const double d = sqrt(c * c + b_minus_a_squared_norm * (radius * radius - a_squared_norm));
                    const double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
                    cg = -dl_alpha * (1.0 - beta); cn = beta;
                    double s2 = 0;
                    for (int k = 0; k < K; ++k) for (int dd = 0; dd < 6; ++dd) { const double v = cg * grp[(size_t)k * 6 + dd] + cn * gnp[(size_t)k * 6 + dd]; s2 += v * v; }
                    for (int l = 0; l < P; ++l) for (int dd = 0; dd < 3; ++dd) { const double v = cg * grl[(size_t)l * 3 + dd] + cn * gnl[(size_t)l * 3 + dd]; s2 += v * v; }
                    dl_step_norm = sqrt(s2);
                }
                for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d)
                    xp[(size_t)k * 6 + d] = pfree[k] ? (cg * grp[(size_t)k * 6 + d] + cn * gnp[(size_t)k * 6 + d]) / dgp[(size_t)k * 6 + d] : 0.0;
                for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d)
                    xl[(size_t)l * 3 + d] = lfree[l] ? (cg * grl[(size_t)l * 3 + d] + cn * gnl[(size_t)l * 3 + d]) / dgl[(size_t)l * 3 + d] : 0.0;
            }
        }
        double model_cost_change = 0;
        if (valid) {
            for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) { if (!isfinite(xp[(size_t)k * 6 + d])) valid = 0; xp[(size_t)k * 6 + d] *= scp[(size_t)k * 6 + d]; }
            for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) { if (!isfinite(xl[(size_t)l * 3 + d])) valid = 0; xl[(size_t)l * 3 + d] *= scl[(size_t)l * 3 + d]; }
        }
        if (valid) {      /* model_cost_change = -model_residuals . (residuals + model_residuals / 2), model_residuals = J delta */
            for (int e = 0; e < E; ++e) {
                const double *jx = Jx + 12 * (size_t)e, *jp = Jp + 6 * (size_t)e;
                const double *dx = xp + 6 * (size_t)pb->edge_pose[e], *dq = xl + 3 * (size_t)pb->edge_point[e];
                for (int a = 0; a < 2; ++a) {
                    double m = 0;
                    for (int c = 0; c < 6; ++c) m += jx[6 * a + c] * dx[c];
                    for (int c = 0; c < 3; ++c) m += jp[3 * a + c] * dq[c];
                    model_cost_change -= m * (rc[2 * (size_t)e + a] + m / 2);
                }
            }
            if (!(model_cost_change > 0)) valid = 0;
        }
        if (!valid) {     /* HandleInvalidStep */
            if (++invalid_run >= opt->max_num_consecutive_invalid_steps) { term = YO_CERES_FAILURE; break; }
            if (dogleg) { dl_mu *= 10.0; dl_reuse = 0; }                 /* DoglegStrategy::StepIsInvalid */
            else radius *= 0.5;
            ++S.unsuccessful_steps;
            continue;
        }
        invalid_run = 0;
        double step2 = 0;
        for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) {
            const double dv = pfree[k] ? xp[(size_t)k * 6 + d] : 0.0;
            cposes[(size_t)k * 6 + d] = pb->poses[(size_t)k * 6 + d] + dv; step2 += dv * dv; }
        for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) {
            const double dv = lfree[l] ? xl[(size_t)l * 3 + d] : 0.0;
            cpoints[(size_t)l * 3 + d] = pb->points[(size_t)l * 3 + d] + dv; step2 += dv * dv; }
        double cand_cost = DBL_MAX;                         /* evaluation failure = a step of very high cost */
        if (yo_ceres_linearize(pb, cposes, cpoints, &cand_cost, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != 0) cand_cost = DBL_MAX;
        /* ParameterToleranceReached */
        if (sqrt(step2) <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = YO_CERES_PARAMETER_TOLERANCE; break; }
        /* FunctionToleranceReached */
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= opt->function_tolerance * x_cost) { term = YO_CERES_FUNCTION_TOLERANCE; break; }
        const double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > opt->min_relative_decrease) {      /* HandleSuccessfulStep */
            memcpy(pb->poses, cposes, 8 * 6 * (size_t)K); memcpy(pb->points, cpoints, 8 * 3 * (size_t)P);
            if (yo_ceres_linearize(pb, pb->poses, pb->points, &x_cost, Hpp, bp, Hll, bl, Hpl, Jx, Jp, rc) != 0) { term = YO_CERES_FAILURE; break; }
            X_NORM_GRAD();
            if (dogleg) {                                   /* DoglegStrategy::StepAccepted */
                if (relative_decrease < 0.25) radius *= 0.5;
                if (relative_decrease > 0.75) { const double r3 = 3.0 * dl_step_norm; if (r3 > radius) radius = r3; }
                if (radius > opt->max_trust_region_radius) radius = opt->max_trust_region_radius;
                dl_mu = 2.0 * dl_mu / 10.0; if (dl_mu < 1e-8) dl_mu = 1e-8;
                dl_reuse = 0;
            } else {
                double t = 2.0 * relative_decrease - 1.0;   /* LevenbergMarquardtStrategy::StepAccepted */
                t = 1.0 - t * t * t;
                radius = radius / (t > 1.0 / 3.0 ? t : 1.0 / 3.0);
                if (radius > opt->max_trust_region_radius) radius = opt->max_trust_region_radius;
                decrease_factor = 2.0;
            }
            ++S.successful_steps;
        } else {                                                   /* HandleUnsuccessfulStep -> StepRejected */
            if (dogleg) { radius *= 0.5; dl_reuse = 1; }
            else { radius = radius / decrease_factor; decrease_factor *= 2.0; }
            ++S.unsuccessful_steps;
        }
    }
done:
    S.termination = term; S.final_cost = x_cost; S.final_radius = radius;
    if (sum) *sum = S;
    free(Hpp); free(bp); free(Hll); free(bl); free(Hpl); free(Jx); free(Jp); free(rc);
    free(sHpp); free(sbp); free(sHll); free(sbl); free(sHpl); free(scp); free(scl); free(dp); free(dl); free(xp); free(xl);
    free(cposes); free(cpoints); free(pfree); free(lfree);
    free(dgp); free(dgl); free(grp); free(grl); free(gnp); free(gnl);
    return term == YO_CERES_FAILURE ? -1 : 0;
#undef X_NORM_GRAD
}

/* ------------------------------------------------------------------------------------------------ g2o LM (B4) */
/* OptimizationAlgorithmLevenberg::solve as optimizer.optimize(n) drives it (BA.cpp:501-502) [frozen spec of g2o]:
 * lambda_0 = 1e-5 max diag(H); trial: solve (H + lambda I) x = b, update, rho = (chi - chi_new) / (x.(lambda x + b) + 1e-3);
 * rho > 0: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), nu = 2; else restore, lambda *= nu, nu *= 2; at most 10 trials;
 * the outer loop stops when 10 trials failed or rho == 0.  poses are [omega; t] (G2oTypes.h:88). */
int yo_g2o_lm(const yo_ba_problem *pb0, double *poses, double *points, int max_iterations, yo_lm_stats *stats)
{
    const int K = pb0->n_poses, P = pb0->n_points, E = pb0->n_edges;
    const size_t nK = (size_t)(K > 0 ? K : 1), nP = (size_t)(P > 0 ? P : 1), nE = (size_t)(E > 0 ? E : 1);
    yo_ba_problem pb = *pb0; pb.poses = poses; pb.points = points;
    double *Hpp = (double *)malloc(8 * 36 * nK), *bp = (double *)malloc(8 * 6 * nK), *Hll = (double *)malloc(8 * 9 * nP),
           *bl = (double *)malloc(8 * 3 * nP), *Hpl = (double *)malloc(8 * 18 * nE);
    double *dp = (double *)malloc(8 * 6 * nK), *dl = (double *)malloc(8 * 3 * nP), *xp = (double *)malloc(8 * 6 * nK), *xl = (double *)malloc(8 * 3 * nP);
    double *bposes = (double *)malloc(8 * 6 * nK), *bpoints = (double *)malloc(8 * 3 * nP);
    double *t1 = (double *)malloc(8 * 36 * nK), *t2 = (double *)malloc(8 * 6 * nK), *t3 = (double *)malloc(8 * 9 * nP), *t4 = (double *)malloc(8 * 3 * nP);
    uint8_t *pfree = (uint8_t *)malloc(nK), *lfree = (uint8_t *)malloc(nP);
    for (int k = 0; k < K; ++k) pfree[k] = !(pb0->pose_fixed && pb0->pose_fixed[k]);
    memset(lfree, 1, nP);
    yo_lm_stats st; memset(&st, 0, sizeof(st));
    double lambda = 0, ni = 2, currentChi = 0;
    for (int it = 0; it < max_iterations; ++it) {
        currentChi = yo_ba_linearize(&pb, Hpp, bp, Hll, bl, Hpl, NULL, NULL);
        if (it == 0) {
            st.chi2_initial = currentChi;
            double mx = 0;
            for (int k = 0; k < K; ++k) if (pfree[k]) for (int d = 0; d < 6; ++d) mx = fmax(mx, fabs(Hpp[(size_t)k * 36 + 7 * d]));
            for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) mx = fmax(mx, fabs(Hll[(size_t)l * 9 + 4 * d]));
            lambda = 1e-5 * mx; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(bposes, poses, 8 * 6 * (size_t)K); memcpy(bpoints, points, 8 * 3 * (size_t)P);
            for (size_t i = 0; i < 6 * (size_t)K; ++i) dp[i] = lambda;
            for (size_t i = 0; i < 3 * (size_t)P; ++i) dl[i] = lambda;
            const int ok = yo_ba_schur_solve(K, P, E, pb.edge_pose, pb.edge_point, pfree, lfree, Hpp, Hll, Hpl, bp, bl, dp, dl, xp, xl);
            double tempChi = DBL_MAX;
            if (ok) {
                for (int k = 0; k < K; ++k) if (pfree[k]) yo_ba_pose_oplus(poses + 6 * (size_t)k, xp + 6 * (size_t)k);
                for (size_t i = 0; i < 3 * (size_t)P; ++i) points[i] += xl[i];
                tempChi = yo_ba_linearize(&pb, t1, t2, t3, t4, NULL, NULL, NULL);
            }
            rho = currentChi - tempChi;
            double scale = 0;
            if (ok) {
                for (int k = 0; k < K; ++k) if (pfree[k]) for (int d = 0; d < 6; ++d) { const double x = xp[(size_t)k * 6 + d]; scale += x * (lambda * x + bp[(size_t)k * 6 + d]); }
                for (size_t i = 0; i < 3 * (size_t)P; ++i) scale += xl[i] * (lambda * xl[i] + bl[i]);
            }
            scale += 1e-3;
            rho /= scale;
            ++st.lm_trials;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                if (alpha > 2. / 3.) alpha = 2. / 3.;
                lambda *= (alpha > 1. / 3. ? alpha : 1. / 3.);
                ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(poses, bposes, 8 * 6 * (size_t)K); memcpy(points, bpoints, 8 * 3 * (size_t)P);
                if (!isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++st.iterations;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
    }
    st.chi2_final = currentChi; st.lambda_final = lambda;
    if (stats) *stats = st;
    free(Hpp); free(bp); free(Hll); free(bl); free(Hpl); free(dp); free(dl); free(xp); free(xl); free(bposes); free(bpoints);
    free(t1); free(t2); free(t3); free(t4); free(pfree); free(lfree);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ B7 */
/* ba::OptimizeCurrentPoseOnly (BA.cpp:188-264) on plain arrays.  pose_io = [t; log(so3)] of current->_TCW on entry and the
 * value the function leaves in _TCW on exit.  Reproduced as written: every round restarts from the ENTRY pose
 * (pose = pose_backup, :229), the inlier test of round `it` runs with the _TCW of round it-1 (the pose is committed only
 * after the test, :254), a round with fewer than 10 inliers leaves _TCW at the previous round's value (:252-253).
 * bad_out [n] (Feature::_bad), depth_out [n] (Feature::_depth, written for inliers only), returns the last inlier count. */
int yo_optimize_current_pose_only(const yo_camera *cam, double pose_io[6], int n, const double *px /*[n][2]*/,
                                  const double *pw /*[n][3]*/, uint8_t *bad_out, double *depth_out, int *rounds_run)
{
    const float chi2Mono = 5.991f;
    double pose_backup[6], pose[6], tcw[6];
    memcpy(pose_backup, pose_io, sizeof(pose_backup)); memcpy(tcw, pose_io, sizeof(tcw));
    double *obs = (double *)malloc(16 * (size_t)(n > 0 ? n : 1));
    int32_t *ep = (int32_t *)calloc((size_t)(n > 0 ? n : 1), 4), *el = (int32_t *)malloc(4 * (size_t)(n > 0 ? n : 1));
    uint8_t *enable = (uint8_t *)malloc((size_t)(n > 0 ? n : 1)), *pfix = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    double *pts = (double *)malloc(24 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {                           /* Pixel2Camera2D, Camera.h:64-69 */
        obs[2 * i] = (px[2 * i] - (double)cam->cx) / (double)cam->fx; obs[2 * i + 1] = (px[2 * i + 1] - (double)cam->cy) / (double)cam->fy;
        el[i] = i; enable[i] = 1; pfix[i] = 1; bad_out[i] = 0;
        pts[3 * i] = pw[3 * i]; pts[3 * i + 1] = pw[3 * i + 1]; pts[3 * i + 2] = pw[3 * i + 2];
    }
    yo_ceres_problem pb; memset(&pb, 0, sizeof(pb));
    pb.n_poses = 1; pb.n_points = n; pb.n_edges = n; pb.poses = pose; pb.points = pts; pb.point_fixed = pfix;
    pb.edge_pose = ep; pb.edge_point = el; pb.obs_n = obs; pb.edge_enable = enable; pb.fail_behind_camera = 1;
    yo_ceres_options opt; yo_ceres_default_options(&opt);
    int cntInlier = 0, it = 0;
    for (it = 0; it < 4; ++it) {
        memcpy(pose, pose_backup, sizeof(pose));
        yo_ceres_solve(&pb, &opt, NULL);
        cntInlier = 0;
        yo_se3 T; double th_;                               /* _TCW = SE3(SO3::exp(aa), t), BA.cpp:254 */
        yo_so3_exp(tcw + 3, T.q, &th_); T.t[0] = tcw[0]; T.t[1] = tcw[1]; T.t[2] = tcw[2];
        for (int i = 0; i < n; ++i) {                       /* World2Pixel / World2Camera with current->_TCW, Camera.h:41-52,73-75 */
            double pc[3];
            yo_se3_act(&T, pw + 3 * i, pc);
            const double u = (double)cam->fx * pc[0] / pc[2] + (double)cam->cx, v = (double)cam->fy * pc[1] / pc[2] + (double)cam->cy;
            const double dx = u - px[2 * i], dy = v - px[2 * i + 1], error2 = dx * dx + dy * dy;
            if (error2 > chi2Mono) { bad_out[i] = 1; enable[i] = 0; }
            else { depth_out[i] = pc[2]; bad_out[i] = 0; ++cntInlier; enable[i] = 1; }
        }
        if (cntInlier < 10) { ++it; break; }
        memcpy(tcw, pose, sizeof(tcw));
    }
    if (rounds_run) *rounds_run = it;
    memcpy(pose_io, tcw, sizeof(tcw));
    free(obs); free(ep); free(el); free(enable); free(pfix); free(pts);
    return cntInlier;
}

/* ---- the other ceres-based entry points of src/Algorithm/BA.cpp on plain arrays (B7) --------------------------------------
 * Each function builds the residual blocks in the order the reference adds them, solves with yo_ceres_solve (default options =
 * trust-region Levenberg-Marquardt) and applies the reference's write-back.  Poses are [t; log(so3)] 6-vectors. */
static void se3_to_taa(const yo_se3 *T, double p[6])
{
    double th;
    p[0] = T->t[0]; p[1] = T->t[1]; p[2] = T->t[2];
    yo_so3_log(T->q, p + 3, &th);                         /* pose.head<3>() = translation, pose.tail<3>() = so3().log() */
}
static void taa_to_se3(const double p[6], yo_se3 *T)
{
    double th;
    yo_so3_exp(p + 3, T->q, &th); T->t[0] = p[0]; T->t[1] = p[1]; T->t[2] = p[2];   /* SE3(SO3::exp(tail), head) */
}
static void world2pixel(const yo_camera *cam, const yo_se3 *T, const double pw[3], double px[2], double *z)
{   /* Camera::World2Pixel / World2Camera, Camera.h:41-52,73-75 */
    double pc[3];
    yo_se3_act(T, pw, pc);
    px[0] = (double)cam->fx * pc[0] / pc[2] + (double)cam->cx; px[1] = (double)cam->fy * pc[1] / pc[2] + (double)cam->cy;
    if (z) *z = pc[2];
}
static void pixel2camera2d(const yo_camera *cam, const double px[2], double out[2])
{   /* Camera.h:64-69 */
    out[0] = (px[0] - (double)cam->cx) / (double)cam->fx; out[1] = (px[1] - (double)cam->cy) / (double)cam->fy;
}

/* ba::TwoViewBACeres -- BA.cpp:11-89.  ref is constant (CeresReprojectionErrorPointOnly), curr is optimised together with the
 * points; outlier points restart from (0,0,1) and their two residual blocks get HuberLoss(0.1).  options.trust_region_strategy_type =
 * DOGLEG (:59-60): yo_ceres_solve's DoglegStrategy (round 6; rounds 3-5 solved this problem with the Levenberg-Marquardt strategy -- both stop at a
 * stationary point of the same cost, tests/test_oracle_ceres.py holds them against each other and against scipy's dogbox solver).
 * inlier [n] in/out, pts_ref [n][3] in/out. */
int yo_two_view_ba_ceres(const yo_camera *cam, const yo_se3 *ref, yo_se3 *curr, int n, const double *px_ref, const double *px_curr,
                         uint8_t *inlier, double *pts_ref, yo_ceres_summary *sum)
{
    const size_t nz = (size_t)(n > 0 ? n : 1);
    double poses[12];
    se3_to_taa(ref, poses); se3_to_taa(curr, poses + 6);
    const uint8_t pose_fixed[2] = { 1, 0 };
    int32_t *ep = (int32_t *)malloc(8 * nz), *el = (int32_t *)malloc(8 * nz);
    double *obs = (double *)malloc(32 * nz), *hub = (double *)malloc(16 * nz);
    for (int i = 0; i < n; ++i) {
        if (!inlier[i]) { pts_ref[3 * i] = 0; pts_ref[3 * i + 1] = 0; pts_ref[3 * i + 2] = 1; }      /* :36-38 */
        const double a = inlier[i] ? 0.0 : 0.1;
        ep[2 * i] = 0; el[2 * i] = i; pixel2camera2d(cam, px_ref + 2 * i, obs + 4 * i); hub[2 * i] = a;            /* :41-45 */
        ep[2 * i + 1] = 1; el[2 * i + 1] = i; pixel2camera2d(cam, px_curr + 2 * i, obs + 4 * i + 2); hub[2 * i + 1] = a;   /* :48-55 */
    }
    yo_ceres_problem pb; memset(&pb, 0, sizeof(pb));
    pb.n_poses = 2; pb.n_points = n; pb.n_edges = 2 * n; pb.poses = poses; pb.pose_fixed = pose_fixed; pb.points = pts_ref;
    pb.edge_pose = ep; pb.edge_point = el; pb.obs_n = obs; pb.edge_huber = hub;
    yo_ceres_options opt; yo_ceres_default_options(&opt);
    opt.trust_region_strategy = YO_CERES_DOGLEG;                                        /* :60 */
    if (n > 0) yo_ceres_solve(&pb, &opt, sum);
    taa_to_se3(poses + 6, curr);                                                        /* :65 */
    const double ch2 = 5.991;
    int n_in = 0;
    for (int i = 0; i < n; ++i) {                                                        /* :68-84 */
        double p1[2], p2[2], d1, d2;
        world2pixel(cam, ref, pts_ref + 3 * i, p1, &d1); world2pixel(cam, curr, pts_ref + 3 * i, p2, &d2);
        const double e1x = px_ref[2 * i] - p1[0], e1y = px_ref[2 * i + 1] - p1[1], e2x = px_curr[2 * i] - p2[0], e2y = px_curr[2 * i + 1] - p2[1];
        if (e1x * e1x + e1y * e1y > ch2 || e2x * e2x + e2y * e2y > ch2) inlier[i] = 0;
        else if (d1 < 0 || d2 < 0) inlier[i] = 0;
        else { inlier[i] = 1; ++n_in; }
    }
    free(ep); free(el); free(obs); free(hub);
    return n_in;
}

/* the residual blocks shared by OptimizeCurrent / OptimizeCurrentPointOnly: per feature one block to the current frame, then one
 * PointOnly block per keyframe observation of ITS map point (so a map point shared by two features brings its keyframe
 * observations twice, as the reference's loops do).  Keyframes are constant poses 1..K, the current frame is pose 0. */
static int current_frame_problem(const yo_camera *cam, int n, const double *px, const int32_t *feat_point, const uint8_t *skip,
                                 const int32_t *obs_off, const int32_t *obs_kf, const double *obs_px, double huber_a,
                                 int32_t **ep_out, int32_t **el_out, double **obs_out, double **hub_out)
{
    int E = 0;
    for (int i = 0; i < n; ++i) if (!(skip && skip[i])) E += 1 + (obs_off[feat_point[i] + 1] - obs_off[feat_point[i]]);
    const size_t Ez = (size_t)(E > 0 ? E : 1);
    int32_t *ep = (int32_t *)malloc(4 * Ez), *el = (int32_t *)malloc(4 * Ez);
    double *obs = (double *)malloc(16 * Ez), *hub = (double *)malloc(8 * Ez);
    int e = 0;
    for (int i = 0; i < n; ++i) {
        if (skip && skip[i]) continue;
        const int l = feat_point[i];
        ep[e] = 0; el[e] = l; pixel2camera2d(cam, px + 2 * i, obs + 2 * e); hub[e] = huber_a; ++e;
        for (int o = obs_off[l]; o < obs_off[l + 1]; ++o) {
            ep[e] = 1 + obs_kf[o]; el[e] = l; pixel2camera2d(cam, obs_px + 2 * o, obs + 2 * e); hub[e] = huber_a; ++e;
        }
    }
    *ep_out = ep; *el_out = el; *obs_out = obs; *hub_out = hub;
    return E;
}

/* ba::OptimizeCurrent -- BA.cpp:91-186: the current pose and the map points of its features, every observing keyframe constant,
 * HuberLoss(0.1) on every block; then Feature::_bad where the reprojection error^2 exceeds 4 * 5.991 (float), else _depth.
 * feat_point [n] = map point of each feature (index into points [P][3]); keyframe observations in CSR form per map point. */
int yo_optimize_current(const yo_camera *cam, yo_se3 *T_cur, int n, const double *px, const int32_t *feat_point, int P, double *points,
                        int K, const yo_se3 *kf_T, const int32_t *obs_off, const int32_t *obs_kf, const double *obs_px,
                        uint8_t *bad, double *depth, yo_ceres_summary *sum)
{
    const float chi2Mono = 5.991 * 4;
    double *poses = (double *)malloc(48 * (size_t)(K + 1));
    uint8_t *pfix = (uint8_t *)malloc((size_t)(K + 1));
    se3_to_taa(T_cur, poses); pfix[0] = 0;
    for (int k = 0; k < K; ++k) { se3_to_taa(&kf_T[k], poses + 6 * (k + 1)); pfix[k + 1] = 1; }
    int32_t *ep, *el; double *obs, *hub;
    const int E = current_frame_problem(cam, n, px, feat_point, NULL, obs_off, obs_kf, obs_px, 0.1, &ep, &el, &obs, &hub);
    yo_ceres_problem pb; memset(&pb, 0, sizeof(pb));
    pb.n_poses = K + 1; pb.n_points = P; pb.n_edges = E; pb.poses = poses; pb.pose_fixed = pfix; pb.points = points;
    pb.edge_pose = ep; pb.edge_point = el; pb.obs_n = obs; pb.edge_huber = hub;
    yo_ceres_options opt; yo_ceres_default_options(&opt);
    if (E > 0) yo_ceres_solve(&pb, &opt, sum);
    taa_to_se3(poses, T_cur);                                                            /* :144-146 */
    int cntInlier = 0;
    for (int i = 0; i < n; ++i) {                                                        /* :148-158 */
        double p[2], z;
        world2pixel(cam, T_cur, points + 3 * (size_t)feat_point[i], p, &z);
        const double dx = p[0] - px[2 * i], dy = p[1] - px[2 * i + 1], error2 = dx * dx + dy * dy;
        if (error2 > chi2Mono) bad[i] = 1;
        else { depth[i] = z; ++cntInlier; }
    }
    free(poses); free(pfix); free(ep); free(el); free(obs); free(hub);
    return cntInlier;
}

/* ba::OptimizeCurrentPointOnly -- BA.cpp:266-322: only the map points move; features that are bad or have no map point are
 * skipped (feat_point < 0 = no map point); no loss function. */
int yo_optimize_current_point_only(const yo_camera *cam, const yo_se3 *T_cur, int n, const double *px, const int32_t *feat_point,
                                   const uint8_t *feat_bad, int P, double *points, int K, const yo_se3 *kf_T, const int32_t *obs_off,
                                   const int32_t *obs_kf, const double *obs_px, yo_ceres_summary *sum)
{
    double *poses = (double *)malloc(48 * (size_t)(K + 1));
    uint8_t *pfix = (uint8_t *)malloc((size_t)(K + 1)), *skip = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    se3_to_taa(T_cur, poses); pfix[0] = 1;
    for (int k = 0; k < K; ++k) { se3_to_taa(&kf_T[k], poses + 6 * (k + 1)); pfix[k + 1] = 1; }
    for (int i = 0; i < n; ++i) skip[i] = (uint8_t)((feat_bad && feat_bad[i]) || feat_point[i] < 0);    /* :272-273 */
    int32_t *ep, *el; double *obs, *hub;
    const int E = current_frame_problem(cam, n, px, feat_point, skip, obs_off, obs_kf, obs_px, 0.0, &ep, &el, &obs, &hub);
    yo_ceres_problem pb; memset(&pb, 0, sizeof(pb));
    pb.n_poses = K + 1; pb.n_points = P; pb.n_edges = E; pb.poses = poses; pb.pose_fixed = pfix; pb.points = points;
    pb.edge_pose = ep; pb.edge_point = el; pb.obs_n = obs;
    yo_ceres_options opt; yo_ceres_default_options(&opt);
    int rc = 0;
    if (E > 0) rc = yo_ceres_solve(&pb, &opt, sum);
    free(poses); free(pfix); free(skip); free(ep); free(el); free(obs); free(hub);
    return rc;
}
