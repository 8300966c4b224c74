/* ORACLE (test infrastructure) -- SO3/SE3 as in thirdparty/Sophus/sophus/{so3,se3}.cpp
 * (non-template Sophus, unit quaternion + translation), with the Eigen quaternion
 * kernels it calls restated [frozen spec of Eigen 3 Quaternion: product, normalize,
 * _transformVector, toRotationMatrix].  See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>

#define SMALL_EPS 1e-10     /* sophus/so3.h:35 */

static void quat_normalize(double q[4])
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void quat_mul(const double a[4], const double b[4], double c[4])
{   /* Eigen quat_product (scalar path); storage x,y,z,w */
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by + ay * bw + az * bx - ax * bz;
    c[2] = aw * bz + az * bw + ax * by - ay * bx;
}

static void quat_rotate(const double q[4], const double v[3], double out[3])
{   /* Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv */
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = { q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0] };
    out[0] = v[0] + q[3] * uv[0] + c[0];
    out[1] = v[1] + q[3] * uv[1] + c[1];
    out[2] = v[2] + q[3] * uv[2] + c[2];
}

void yo_quat_to_R(const double q[4], double R[9])
{   /* Eigen QuaternionBase::toRotationMatrix */
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* SO3::expAndTheta -- so3.cpp:178-202 (+ SO3(Quaterniond) ctor normalisation :43-47) */
void yo_so3_exp(const double w[3], double q[4], double *theta_out)
{
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double half_theta = 0.5 * theta;
    double imag_factor;
    const double real_factor = cos(half_theta);
    if (theta < SMALL_EPS) {
        const double theta_sq = theta * theta, theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
        imag_factor = sin(half_theta) / theta;
    }
    q[3] = real_factor; q[0] = imag_factor * w[0]; q[1] = imag_factor * w[1]; q[2] = imag_factor * w[2];
    quat_normalize(q);
    if (theta_out) *theta_out = theta;
}

/* SO3::logAndTheta -- so3.cpp:127-169.  NB the |w|<eps branch (:150-160) is
 * immediately overwritten by :161 in the reference; restated as written. */
void yo_so3_log(const double q[4], double out[3], double *theta_out)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double w = q[3], squared_w = w * w;
    double two_atan_nbyw_by_n;
    if (n < SMALL_EPS)
        two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
    else
        two_atan_nbyw_by_n = 2 * atan(n / w) / n;
    if (theta_out) *theta_out = two_atan_nbyw_by_n * n;
    out[0] = two_atan_nbyw_by_n * q[0]; out[1] = two_atan_nbyw_by_n * q[1]; out[2] = two_atan_nbyw_by_n * q[2];
}

void yo_se3_identity(yo_se3 *T)
{
    T->q[0] = T->q[1] = T->q[2] = 0; T->q[3] = 1; T->t[0] = T->t[1] = T->t[2] = 0;
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static void hat(const double v[3], double O[9])
{   /* SO3::hat -- so3.cpp:204-212 */
    O[0] = 0;     O[1] = -v[2]; O[2] = v[1];
    O[3] = v[2];  O[4] = 0;     O[5] = -v[0];
    O[6] = -v[1]; O[7] = v[0];  O[8] = 0;
}

/* SE3::exp -- se3.cpp:170-196.  update = [upsilon(3); omega(3)] */
void yo_se3_exp(const double u[6], yo_se3 *T)
{
    double theta, Om[9], Om2[9], V[9];
    yo_so3_exp(u + 3, T->q, &theta);
    hat(u + 3, Om);
    mat3_mul(Om, Om, Om2);
    if (theta < SMALL_EPS) {
        yo_quat_to_R(T->q, V);
    } else {
        const double theta_sq = theta * theta;
        const double a = (1 - cos(theta)) / theta_sq, b = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
    }
    for (int i = 0; i < 3; ++i) T->t[i] = V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2];
}

/* SE3::log -- se3.cpp:198-220 */
void yo_se3_log(const yo_se3 *T, double out[6])
{
    double theta, Om[9], Om2[9], Vi[9];
    yo_so3_log(T->q, out + 3, &theta);
    hat(out + 3, Om);
    mat3_mul(Om, Om, Om2);
    if (theta < SMALL_EPS) {
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
    for (int i = 0; i < 3; ++i) out[i] = Vi[3 * i] * T->t[0] + Vi[3 * i + 1] * T->t[1] + Vi[3 * i + 2] * T->t[2];
}

/* SE3::operator* -- se3.cpp:59-66; SO3::operator*= -- so3.cpp:73-78 (normalises) */
void yo_se3_mul(const yo_se3 *A, const yo_se3 *B, yo_se3 *C)
{
    double r[3], q[4];
    quat_rotate(A->q, B->t, r);
    quat_mul(A->q, B->q, q);
    quat_normalize(q);
    C->t[0] = A->t[0] + r[0]; C->t[1] = A->t[1] + r[1]; C->t[2] = A->t[2] + r[2];
    C->q[0] = q[0]; C->q[1] = q[1]; C->q[2] = q[2]; C->q[3] = q[3];
}

/* SE3::inverse -- se3.cpp:77-84 (SO3::inverse -> SO3(conjugate) ctor normalises) */
void yo_se3_inv(const yo_se3 *A, yo_se3 *B)
{
    double q[4] = { -A->q[0], -A->q[1], -A->q[2], A->q[3] };
    quat_normalize(q);
    const double nt[3] = { A->t[0] * -1., A->t[1] * -1., A->t[2] * -1. };
    double r[3];
    quat_rotate(q, nt, r);
    B->q[0] = q[0]; B->q[1] = q[1]; B->q[2] = q[2]; B->q[3] = q[3];
    B->t[0] = r[0]; B->t[1] = r[1]; B->t[2] = r[2];
}

/* SE3::operator*(Vector3d) -- se3.cpp:92-96 */
void yo_se3_act(const yo_se3 *T, const double p[3], double out[3])
{
    double r[3];
    quat_rotate(T->q, p, r);
    out[0] = r[0] + T->t[0]; out[1] = r[1] + T->t[1]; out[2] = r[2] + T->t[2];
}

void yo_camera_default(yo_camera *c)
{   /* config/default.yaml:32-35, stored as float (Basic/Camera.h:107) */
    c->fx = 520.9f; c->fy = 521.0f; c->cx = 325.1f; c->cy = 249.7f;
}
