/* exp(x) for x <= 0 in IEEE double arithmetic with nothing but +, -, *, / and one ldexp -- the SAME sequence of correctly rounded
 * operations wherever it is compiled (gcc and hipcc, both with -ffp-contract=off), so a float rounded from it is bit-identical on the
 * host and on the device.  Used where a float exponential feeds a cancellation (DepthFilter::UpdateSeed, src/optimizer.cpp:683-708:
 * one ulp of the Gaussian pdf moves the Beta parameters a, b by ~1e-5): libm's expf and the device's differ in the last ulp.
 * [frozen spec of libm's expf: the result is the double below rounded once to float -- the correctly rounded float exponential unless
 * exp(x) lies within ~2e-16 relative of a rounding boundary.]
 * Method: x = k ln2 + r, |r| <= ln2 / 2 (Cody-Waite split of ln2), exp(r) by its Taylor series to r^13 / 13! (remainder < 5e-18
 * relative), Horner form; exp(x) = 2^k exp(r). */
#ifndef YGZ_EXP_H_
#define YGZ_EXP_H_
#include <math.h>
#if defined(__HIPCC__)
#define YGZ_EXP_FN __host__ __device__ static inline
#else
#define YGZ_EXP_FN static inline
#endif
YGZ_EXP_FN double ygz_exp_nonpos(double x)
{
    if (!(x <= 0.0)) return x != x ? x : 1.0;           /* NaN stays NaN; the callers pass -(d*d)/(2 s^2) <= 0 */
    if (x < -745.0) return 0.0;
    const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double kf = floor(x * inv_ln2 + 0.5);
    const double r = (x - kf * ln2_hi) - kf * ln2_lo;
    double p = 1.0 / 6227020800.0;                       /* 1 / 13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return ldexp(p, (int)kf);
}
#endif
