/* oracle/detmath_c.h -- TEST INFRASTRUCTURE.  exp / log from correctly rounded IEEE operations only (+, -, *, /,
 * explicit fma; exp = k ln2 + r reduction and a degree-13 Taylor polynomial, log = Sun fdlibm's published e_log.c),
 * so that the oracle's ray tracer produces the same bits on any IEEE-754 host or device compiled without implicit
 * multiply-add contraction.  Accuracy <= 1 ulp (checked against libm in
 * tests/test_oracle_golden.py).  See DESIGN.md section 2 for why last bits matter on this path. */
#ifndef ORC_DETMATH_C_H
#define ORC_DETMATH_C_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int64_t orc_bits(double x) { int64_t b; memcpy(&b, &x, 8); return b; }
static inline double orc_from_bits(int64_t b) { double x; memcpy(&x, &b, 8); return x; }

/* the "fma" clone inlines the hardware instruction, the default clone calls libm's (exactly rounded) fma: same bits */
__attribute__((target_clones("fma", "default")))
double orc_exp(double x)
{
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00;
    double kd, r, p;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    kd = rint(x * invln2);
    r = __builtin_fma(-kd, ln2HI, x);
    r = __builtin_fma(-kd, ln2LO, r);
    p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 0.0001984126984126984);
    p = __builtin_fma(p, r, 0.001388888888888889);
    p = __builtin_fma(p, r, 0.008333333333333333);
    p = __builtin_fma(p, r, 0.041666666666666664);
    p = __builtin_fma(p, r, 0.16666666666666666);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

static double orc_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    static const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    int k = 0, hx, i, j;
    int64_t bits;
    double f, dk, s, z, w, t1, t2, R, hfsq;
    if (x != x) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    if (x == INFINITY) return x;
    bits = orc_bits(x);
    if (bits < 0x0010000000000000LL) {
        x *= 1.80143985094819840000e+16;
        k -= 54;
        bits = orc_bits(x);
    }
    hx = (int)(bits >> 32);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    bits = (bits & 0x00000000ffffffffLL) | ((int64_t)(hx | (i ^ 0x3ff00000)) << 32);
    x = orc_from_bits(bits);
    k += (i >> 20);
    f = x - 1.0;
    dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.) {
            if (k == 0) return 0.;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}
#endif
