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

// ---- exp of the attenuation integrand (round 6) -------------------------------------------------------------------------------------
// 19 of the 28 vector instructions per quadrature node were det_exp_inrange's.  Table-reduced form: x = (64 k + j) ln2 / 64 + r,
// |r| <= ln2 / 128, exp(x) = 2^k (T[j] p(r)) with T[j] = 2^(j / 64) correctly rounded and p the degree-5 Taylor polynomial of
// exp(r) (truncation 3.4e-17): 12 FP64 operations instead of 19, one table read (LDS in the kernels that evaluate millions of nodes).
// Still only correctly rounded IEEE operations in a fixed order: the CPU checker (its exp of the same construction) gives the same
// bits.  <= 2 ulp; the integrand is integrated to epsrel = 1e-2.
static __device__ const double det_exp_tab64[64] = {
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
#define DET_LN2HI64 0x1.62e42fee00000p-7
#define DET_LN2LO64 0x1.a39ef35793c76p-39
#define DET_INV64 0x1.71547652b82fep+6
// general form (NaN / overflow / underflow branches), table in global memory
__device__ inline double det_exp_tab(double x)
{
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double kd = rint(x * DET_INV64);
    double r = __builtin_fma(-kd, DET_LN2HI64, x);
    r = __builtin_fma(-kd, DET_LN2LO64, r);
    const int ki = (int)kd;
    double p = 0x1.1111111111111p-7;
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(det_exp_tab64[ki & 63] * p, ki >> 6);
}
// arguments known to lie in [-700, 700]; tab: a copy of det_exp_tab64 (LDS).  Same operations, same order, same bits.
__device__ __forceinline__ double det_exp_tab_inrange(double x, const double* tab)
{
    const double kd = rint(x * DET_INV64);
    double r = __builtin_fma(-kd, DET_LN2HI64, x);
    r = __builtin_fma(-kd, DET_LN2LO64, r);
    const int ki = (int)kd;
    const double t = tab[ki & 63];
    double p = det_fma_asm(0x1.1111111111111p-7, r, 0x1.5555555555555p-5);
    p = det_fma_asm_sc(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(t * p, ki >> 6);
}
// two values with their chains interleaved (see det_exp_inrange2)
__device__ __forceinline__ void det_exp_tab_inrange2(double xa, double xb, const double* tab, double& ea, double& eb)
{
    const double ka = rint(xa * DET_INV64), kb = rint(xb * DET_INV64);
    double ra = __builtin_fma(-ka, DET_LN2HI64, xa), rb = __builtin_fma(-kb, DET_LN2HI64, xb);
    const int ia = (int)ka, ib = (int)kb;
    const double ta = tab[ia & 63], tb = tab[ib & 63];
    ra = __builtin_fma(-ka, DET_LN2LO64, ra);
    rb = __builtin_fma(-kb, DET_LN2LO64, rb);
    double pa, pb;
    asm("v_fma_f64 %0, %4, %2, %5\n\t"
        "v_fma_f64 %1, %4, %3, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %1, %1, %3, %6"
        : "=&v"(pa), "=&v"(pb)
        : "v"(ra), "v"(rb), "v"(0x1.1111111111111p-7), "v"(0x1.5555555555555p-5), "s"(0x1.5555555555555p-3));
    pa = __builtin_fma(pa, ra, 0.5);
    pb = __builtin_fma(pb, rb, 0.5);
    pa = __builtin_fma(pa, ra, 1.0);
    pb = __builtin_fma(pb, rb, 1.0);
    pa = __builtin_fma(pa, ra, 1.0);
    pb = __builtin_fma(pb, rb, 1.0);
    ea = ldexp(ta * pa, ia >> 6);
    eb = ldexp(tb * pb, ib >> 6);
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
