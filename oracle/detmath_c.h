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

/* exp for the attenuation integrand (round 6): reduction to |r| <= ln2 / 128 with a 64-entry table of 2^(j / 64) (correctly
 * rounded doubles, the same literals as nuradiomc_amd/csrc/detmath.h) and the degree-5 Taylor polynomial of exp(r) in Horner form
 * (truncation 3.4e-17): x = (64 k + j) ln2 / 64 + r, exp(x) = 2^k (T[j] p(r)).  Every step a correctly rounded IEEE operation in a
 * fixed order -- same bits on host and device.  Accuracy <= 2 ulp (the table entry, the polynomial and their product each round
 * once): the integrand is integrated to epsrel = 1e-2.  7 FP64 operations fewer per value than orc_exp. */
static const double orc_exp_tab64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
__attribute__((target_clones("fma", "default")))
double orc_exp_tab(double x)
{
    static const double ln2HI64 = 0x1.62e42fee00000p-7, ln2LO64 = 0x1.a39ef35793c76p-39, inv64 = 0x1.71547652b82fep+6;
    double kd, r, p;
    int ki;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    kd = rint(x * inv64);
    r = __builtin_fma(-kd, ln2HI64, x);
    r = __builtin_fma(-kd, ln2LO64, r);
    ki = (int)kd;
    p = 0x1.1111111111111p-7;                          /* 1/120 */
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);     /* 1/24 */
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);     /* 1/6 */
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(orc_exp_tab64[ki & 63] * p, ki >> 6);
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
