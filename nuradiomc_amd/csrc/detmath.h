// detmath.h -- exp / log built only from IEEE-754 binary64 +, -, *, / and integer bit operations.
//
// The ray-solution finder the reference uses is numerically chaotic at the 1e-7 level: its
// MINPACK iteration on (delta y)^2 stops at an iteration count that flips with the last bit of exp / log, and that
// decides C0 to ~1e-7 and, rarely, whether a root is reported at all.  A vendor libm differs from the host libm in
// those last bits, so results would differ between GPU and CPU.  These two functions follow the classic
// table-free algorithms of Sun's fdlibm (e_exp.c: k ln2 + r reduction and a degree-5 Remez rational; e_log.c:
// s = f / (2 + f) series with the Lg1..Lg7 minimax coefficients), whose every step is a correctly rounded basic
// operation -- so any IEEE machine that does not fuse multiply-adds (-ffp-contract=off) produces the same bits.
// Accuracy is < 1 ulp, the same class as the reference's numpy / libm.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

__device__ inline double det_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    double hi = 0., lo = 0.;
    int k = 0;
    double ax = fabs(x);
    if (ax > 0.34657359027997264) {  // 0.5 ln2
        if (ax < 1.0397207708399179) {  // 1.5 ln2
            if (x > 0) { hi = x - ln2HI; lo = ln2LO; k = 1; }
            else       { hi = x + ln2HI; lo = -ln2LO; k = -1; }
        } else {
            k = (int)(invln2 * x + (x > 0 ? 0.5 : -0.5));
            double t = k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {  // 2^-28
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    long long bits = __double_as_longlong(y);
    if (k >= -1021) {
        bits += (long long)k << 52;
        return __longlong_as_double(bits);
    }
    bits += (long long)(k + 1000) << 52;
    return __longlong_as_double(bits) * 9.33263618503218878990e-302;
}

__device__ inline double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    long long bits = __double_as_longlong(x);
    if (bits < 0x0010000000000000LL) {  // subnormal
        x *= 1.80143985094819840000e+16;
        k -= 54;
        bits = __double_as_longlong(x);
    }
    int hx = (int)(bits >> 32);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int i = (hx + 0x95f64) & 0x100000;
    bits = (bits & 0x00000000ffffffffLL) | ((long long)(hx | (i ^ 0x3ff00000)) << 32);  // x or x / 2 in [sqrt2/2, sqrt2)
    x = __longlong_as_double(bits);
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
        if (f == 0.) {
            if (k == 0) return 0.;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

}  // namespace nrhip
