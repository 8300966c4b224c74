// 6x6 pivoted LDL^T solve with pseudo-inverse of D -- the arithmetic of Eigen's
// H_.ldlt().solve(Jres_) that SparseImgAlign::solve uses (src/Algorithm/SparseImageAlign.cpp:225-231).
// Same operation order as oracle/sparse_align.c:yo_ldlt6_solve.
#pragma once
#include <hip/hip_runtime.h>

// workspace variant: m[36], y[6], temp[6], tr[6] provided by the caller (LDS on the device)
__host__ __device__ inline bool ldlt6_solve_ws(const double Hin[36], const double b[6], double x[6],
                                               double *m, double *y, double *temp, int *tr)
{
    const int N = 6;
    for (int i = 0; i < 36; ++i) m[i] = Hin[i];
    for (int k = 0; k < N; ++k) {
        int piv = k; double big = fabs(m[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (fabs(m[i * N + i]) > big) { big = fabs(m[i * N + i]); piv = i; }
        tr[k] = piv;
        if (piv != k) {
            for (int j = 0; j < k; ++j) { double t = m[k * N + j]; m[k * N + j] = m[piv * N + j]; m[piv * N + j] = t; }
            for (int i = piv + 1; i < N; ++i) { double t = m[i * N + k]; m[i * N + k] = m[i * N + piv]; m[i * N + piv] = t; }
            { double t = m[k * N + k]; m[k * N + k] = m[piv * N + piv]; m[piv * N + piv] = t; }
            for (int i = k + 1; i < piv; ++i) { double t = m[i * N + k]; m[i * N + k] = m[piv * N + i]; m[piv * N + i] = t; }
        }
        if (k > 0) {
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
    for (int i = 0; i < N; ++i) y[i] = b[i];
    for (int k = 0; k < N; ++k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) y[i] -= m[i * N + j] * y[j];
    double dmax = 0; for (int i = 0; i < N; ++i) if (fabs(m[i * N + i]) > dmax) dmax = fabs(m[i * N + i]);
    double tol = dmax * 2.220446049250313e-16;
    if (tol < 1.0 / 1.7976931348623157e308) tol = 1.0 / 1.7976931348623157e308;
    for (int i = 0; i < N; ++i) y[i] = (fabs(m[i * N + i]) > tol) ? y[i] / m[i * N + i] : 0.0;
    for (int i = N - 1; i >= 0; --i) for (int j = i + 1; j < N; ++j) y[i] -= m[j * N + i] * y[j];
    for (int k = N - 1; k >= 0; --k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < N; ++i) x[i] = y[i];
    return !(x[0] != x[0]);
}

__host__ __device__ inline bool ldlt6_solve_d(const double Hin[36], const double b[6], double x[6])
{
    double m[36], y[6], temp[6]; int tr[6];
    return ldlt6_solve_ws(Hin, b, x, m, y, temp, tr);
}
