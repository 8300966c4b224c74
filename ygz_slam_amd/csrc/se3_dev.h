// Device-side SO3/SE3 in the representation of thirdparty/Sophus (non-template): unit quaternion
// (x,y,z,w) + translation, double precision.  Same operation order as Sophus/Eigen
// (sophus/so3.cpp:80-202, sophus/se3.cpp:59-220; Eigen quaternion product / _transformVector /
// toRotationMatrix), so results agree with oracle/se3.c to the last bits of the libm calls.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
#include <math.h>
#ifndef __host__
#define __host__
#define __device__
#endif
#endif

#define YGZ_SMALL_EPS 1e-10

struct Se3 { double q[4]; double t[3]; };

__host__ __device__ inline void quat_normalize_d(double q[4])
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

__host__ __device__ inline void quat_mul_d(const double a[4], const double b[4], double c[4])
{
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by + ay * bw + az * bx - ax * bz;
    c[2] = aw * bz + az * bw + ax * by - ay * bx;
}

__host__ __device__ inline void quat_rotate_d(const double q[4], const double v[3], double out[3])
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c0 = q[1] * uv[2] - q[2] * uv[1], c1 = q[2] * uv[0] - q[0] * uv[2], c2 = q[0] * uv[1] - q[1] * uv[0];
    out[0] = v[0] + q[3] * uv[0] + c0;
    out[1] = v[1] + q[3] * uv[1] + c1;
    out[2] = v[2] + q[3] * uv[2] + c2;
}

__host__ __device__ inline void quat_to_R_d(const double q[4], double R[9])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// SO3::expAndTheta (so3.cpp:178-202) + SO3(Quaterniond) normalisation (:43-47)
__host__ __device__ inline void so3_exp_d(const double w[3], double q[4], double *theta_out)
{
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double half_theta = 0.5 * theta;
    double imag_factor;
    const double real_factor = cos(half_theta);
    if (theta < YGZ_SMALL_EPS) {
        const double theta_sq = theta * theta, theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
        imag_factor = sin(half_theta) / theta;
    }
    q[3] = real_factor; q[0] = imag_factor * w[0]; q[1] = imag_factor * w[1]; q[2] = imag_factor * w[2];
    quat_normalize_d(q);
    *theta_out = theta;
}

// SO3::logAndTheta (so3.cpp:127-169), restated as written (the |w|<eps branch is overwritten)
__host__ __device__ inline void so3_log_d(const double q[4], double out[3], double *theta_out)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double w = q[3], squared_w = w * w;
    double f;
    if (n < YGZ_SMALL_EPS) f = 2. / w - 2. * (n * n) / (w * squared_w);
    else f = 2 * atan(n / w) / n;
    *theta_out = f * n;
    out[0] = f * q[0]; out[1] = f * q[1]; out[2] = f * q[2];
}

__host__ __device__ inline void mat3_mul_d(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

__host__ __device__ inline void hat_d(const double v[3], double O[9])
{
    O[0] = 0;     O[1] = -v[2]; O[2] = v[1];
    O[3] = v[2];  O[4] = 0;     O[5] = -v[0];
    O[6] = -v[1]; O[7] = v[0];  O[8] = 0;
}

// SE3::exp (se3.cpp:170-196); u = [upsilon; omega]
__host__ __device__ inline void se3_exp_d(const double u[6], Se3 *T)
{
    double theta, Om[9], Om2[9], V[9];
    so3_exp_d(u + 3, T->q, &theta);
    hat_d(u + 3, Om);
    mat3_mul_d(Om, Om, Om2);
    if (theta < YGZ_SMALL_EPS) {
        quat_to_R_d(T->q, V);
    } else {
        const double theta_sq = theta * theta;
        const double a = (1 - cos(theta)) / theta_sq, b = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
    }
    for (int i = 0; i < 3; ++i) T->t[i] = V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2];
}

// SE3::log (se3.cpp:198-220)
__host__ __device__ inline void se3_log_d(const Se3 *T, double out[6])
{
    double theta, Om[9], Om2[9], Vi[9];
    so3_log_d(T->q, out + 3, &theta);
    hat_d(out + 3, Om);
    mat3_mul_d(Om, Om, Om2);
    if (theta < YGZ_SMALL_EPS) {
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
    for (int i = 0; i < 3; ++i) out[i] = Vi[3 * i] * T->t[0] + Vi[3 * i + 1] * T->t[1] + Vi[3 * i + 2] * T->t[2];
}

// SE3::operator* (se3.cpp:59-66); SO3::operator*= normalises (so3.cpp:73-78)
__host__ __device__ inline void se3_mul_d(const Se3 *A, const Se3 *B, Se3 *C)
{
    double r[3], q[4];
    quat_rotate_d(A->q, B->t, r);
    quat_mul_d(A->q, B->q, q);
    quat_normalize_d(q);
    C->t[0] = A->t[0] + r[0]; C->t[1] = A->t[1] + r[1]; C->t[2] = A->t[2] + r[2];
    C->q[0] = q[0]; C->q[1] = q[1]; C->q[2] = q[2]; C->q[3] = q[3];
}

// SE3::inverse (se3.cpp:77-84)
__host__ __device__ inline void se3_inv_d(const Se3 *A, Se3 *B)
{
    double q[4] = { -A->q[0], -A->q[1], -A->q[2], A->q[3] };
    quat_normalize_d(q);
    const double nt[3] = { A->t[0] * -1., A->t[1] * -1., A->t[2] * -1. };
    double r[3];
    quat_rotate_d(q, nt, r);
    B->q[0] = q[0]; B->q[1] = q[1]; B->q[2] = q[2]; B->q[3] = q[3];
    B->t[0] = r[0]; B->t[1] = r[1]; B->t[2] = r[2];
}

// SE3::operator*(Vector3d) (se3.cpp:92-96)
__host__ __device__ inline void se3_act_d(const Se3 *T, const double p[3], double out[3])
{
    double r[3];
    quat_rotate_d(T->q, p, r);
    out[0] = r[0] + T->t[0]; out[1] = r[1] + T->t[1]; out[2] = r[2] + T->t[2];
}
