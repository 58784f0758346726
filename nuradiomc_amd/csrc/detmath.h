// detmath.h -- exp / log built only from IEEE-754 binary64 +, -, *, / and integer bit operations.
//
// The ray-solution finder the reference uses is numerically chaotic at the 1e-7 level: its
// MINPACK iteration on (delta y)^2 stops at an iteration count that flips with the last bit of exp / log, and that
// decides C0 to ~1e-7 and, rarely, whether a root is reported at all.  A vendor libm differs from the host libm in
// those last bits, so results would differ between GPU and CPU.  These two functions are table-free and every step
// is a correctly rounded IEEE operation (+, -, *, /, and EXPLICIT fused multiply-adds, which are exactly rounded on
// any machine; the compiler is kept from fusing anything else with -ffp-contract=off), so any IEEE host and the device produce
// the same bits.  det_exp: k ln2 + r reduction and a Taylor polynomial (division-free: 16 FMAs); det_log: fdlibm's
// e_log.c (s = f / (2 + f) series with the Lg1..Lg7 minimax coefficients).  Accuracy is <= 1 ulp, the same class as
// the reference's numpy / libm.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

__device__ inline double det_exp(double x)
{
    // k = nearest integer to x / ln2, r = x - k ln2 (two exact-product FMAs), exp(r) by its degree-13 Taylor
    // polynomial in Horner form with FMAs (|r| <= 0.3466: truncation 4e-18), result scaled by 2^k (ldexp).
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double kd = rint(x * invln2);
    double r = __builtin_fma(-kd, ln2HI, x);
    r = __builtin_fma(-kd, ln2LO, r);
    double p = 1.6059043836821613e-10;               // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
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

// det_exp for arguments known to lie in [-700, 700] (no NaN / overflow / underflow branches): the same operations in the same
// order, hence the same bits.  The Horner steps are spelled as three-address v_fma_f64 -- left to itself the compiler emits a
// register copy plus v_fmac_f64 per step (the coefficient registers stay live), i.e. 13 extra VALU instructions per call.
__device__ __forceinline__ double det_fma_asm(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the same with the addend in a scalar register pair (one constant-bus operand per VOP3 instruction on gfx9): the polynomial
// coefficients then live in SGPRs for the whole kernel instead of being copied into VGPRs before every use
__device__ __forceinline__ double det_fma_asm_sc(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ double det_exp_inrange(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double kd = rint(x * invln2);
    double r = __builtin_fma(-kd, ln2HI, x);
    r = __builtin_fma(-kd, ln2LO, r);
    double p = det_fma_asm(1.6059043836821613e-10, r, 2.08767569878681e-09);
    p = det_fma_asm_sc(p, r, 2.505210838544172e-08);
    p = det_fma_asm_sc(p, r, 2.755731922398589e-07);
    p = det_fma_asm_sc(p, r, 2.7557319223985893e-06);
    p = det_fma_asm_sc(p, r, 2.48015873015873e-05);
    p = det_fma_asm_sc(p, r, 0.0001984126984126984);
    p = det_fma_asm_sc(p, r, 0.001388888888888889);
    p = det_fma_asm_sc(p, r, 0.008333333333333333);
    p = det_fma_asm_sc(p, r, 0.041666666666666664);
    p = det_fma_asm_sc(p, r, 0.16666666666666666);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

// two independent det_exp_inrange evaluations with their Horner chains interleaved instruction by instruction (each chain is
// 16 dependent FP64 operations; a wave that alternates between two chains does not wait for its own results).  Same operations
// per value, hence the same bits as det_exp_inrange.  The ten steps with coefficients in scalar registers are ONE asm block: the
// compiler pads every asm statement with a wait state, and it would copy the coefficients into vector registers otherwise.
__device__ __forceinline__ void det_exp_inrange2(double xa, double xb, double& ea, double& eb)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double ka = rint(xa * invln2), kb = rint(xb * invln2);
    double ra = __builtin_fma(-ka, ln2HI, xa), rb = __builtin_fma(-kb, ln2HI, xb);
    ra = __builtin_fma(-ka, ln2LO, ra);
    rb = __builtin_fma(-kb, ln2LO, rb);
    double pa, pb;
    asm("v_fma_f64 %0, %4, %2, %5\n\t"
        "v_fma_f64 %1, %4, %3, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %1, %1, %3, %6\n\t"
        "v_fma_f64 %0, %0, %2, %7\n\t"
        "v_fma_f64 %1, %1, %3, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\t"
        "v_fma_f64 %1, %1, %3, %8\n\t"
        "v_fma_f64 %0, %0, %2, %9\n\t"
        "v_fma_f64 %1, %1, %3, %9\n\t"
        "v_fma_f64 %0, %0, %2, %10\n\t"
        "v_fma_f64 %1, %1, %3, %10\n\t"
        "v_fma_f64 %0, %0, %2, %11\n\t"
        "v_fma_f64 %1, %1, %3, %11\n\t"
        "v_fma_f64 %0, %0, %2, %12\n\t"
        "v_fma_f64 %1, %1, %3, %12\n\t"
        "v_fma_f64 %0, %0, %2, %13\n\t"
        "v_fma_f64 %1, %1, %3, %13\n\t"
        "v_fma_f64 %0, %0, %2, %14\n\t"
        "v_fma_f64 %1, %1, %3, %14"
        : "=&v"(pa), "=&v"(pb)
        : "v"(ra), "v"(rb), "v"(1.6059043836821613e-10), "v"(2.08767569878681e-09), "s"(2.505210838544172e-08),
          "s"(2.755731922398589e-07), "s"(2.7557319223985893e-06), "s"(2.48015873015873e-05), "s"(0.0001984126984126984),
          "s"(0.001388888888888889), "s"(0.008333333333333333), "s"(0.041666666666666664), "s"(0.16666666666666666));
    pa = __builtin_fma(pa, ra, 0.5);
    pb = __builtin_fma(pb, rb, 0.5);
    pa = __builtin_fma(pa, ra, 1.0);
    pb = __builtin_fma(pb, rb, 1.0);
    pa = __builtin_fma(pa, ra, 1.0);
    pb = __builtin_fma(pb, rb, 1.0);
    ea = ldexp(pa, (int)ka);
    eb = ldexp(pb, (int)kb);
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
