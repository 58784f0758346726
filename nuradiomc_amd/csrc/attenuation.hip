// attenuation.hip -- exp(-int_path ds / L_att(z, f)) per (ray, coarse frequency) on MI355X (gfx950).
//
// One LANE per (ray, frequency) item, item = ray * n_freq + i_freq, so the 25..37 frequencies of one
// ray sit in adjacent lanes: they share the ray geometry (same loads, broadcast from L1) and mostly
// the same control flow.  Each lane runs the adaptive Gauss-Kronrod (21-point) quadrature with the
// same subdivision / epsilon-extrapolation decisions as QUADPACK's QAGS (no break point) or QAGP (one
// break point at the turning depth) at epsabs 1.49e-8, epsrel 1e-2, limit 50 -- what
// scipy.integrate.quad does at analyticraytracing.py:1067-1072 -- because with epsrel = 1e-2 the
// reference's result is NOT the converged integral (it can be off by 1e-3), and parity at 1e-6 needs
// the same estimate, not a better one.  About 40 % of the items bisect at least once.
//
// FP64 VALU bound: 21..462 integrand evaluations (2 exp, 2 sqrt, 3 div each) per item against 8 B
// written; the per-lane interval lists (2.2 KB) live in scratch.
#include "ray_device.h"
#include "nrhip_internal.h"
#include <cstdlib>

namespace nrhip {

// attenuation length L(z, f) [m]; model ids as NuRadioMC/utilities/attenuation.py:14
struct AttLane {
    int model;
    double f;   // frequency [GHz]
    double w;   // ln f  (SP1)
};
// GL3 depth table: [3][n] depth (positive), slope, offset (attenuation.py:16-34); kept out of AttLane so that the
// quadrature kernels of the other models carry no extra registers
struct Gl3Tab {
    const double* t;
    int n;
};

// linear interpolation of one GL3 table column like scipy.interpolate.interp1d(bounds_error=False, fill_value=(first, last))
__device__ inline double gl3_interp(double x, const double* __restrict__ d, const double* __restrict__ fp, int n)
{
    if (x < d[0]) return fp[0];
    if (x > d[n - 1]) return fp[n - 1];
    int lo = 0, hi = n;  // searchsorted(side='left'): first index with d[i] >= x
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (d[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    double slope = (fp[idx] - fp[idx - 1]) / (d[idx] - d[idx - 1]);
    return slope * (x - d[idx - 1]) + fp[idx - 1];
}

__device__ inline double attenuation_length(double z, const AttLane& a, Gl3Tab gl3 = Gl3Tab{nullptr, 0})
{
    double L;
    if (a.model == 1) {  // SP1 (attenuation.py:130-142, :168-192)
        double z2 = fabs(z);
        double t = 1.83415e-09 * (z2 * z2 * z2) + (-1.59061e-08 * (z2 * z2)) + 0.00267687 * z2 + (-51.0696);
        const double w0 = -9.210340371976182;  // ln 1e-4
        const double w2 = 1.1505720275988207;  // ln 3.16
        double b0 = -6.74890 + t * (0.026709 - t * 0.000884);
        double b1 = -6.22121 - t * (0.070927 + t * 0.001773);
        double b2 = -4.09468 - t * (0.002213 + t * 0.000332);
        double aa, bb;
        if (a.f < 1.) {
            aa = (b1 * w0 - b0 * 0.0) / (w0 - 0.0);
            bb = (b1 - b0) / (0.0 - w0);
        } else {
            aa = (b2 * 0.0 - b1 * w2) / (0.0 - w2);
            bb = (b2 - b1) / (w2 - 0.0);
        }
        L = 1. / det_exp(aa + bb * a.w);
    } else if (a.model == 2) {  // GL1 (attenuation.py:99-128, :194-196), 75 MHz length clamped at 100 m
        const double fit[6] = {1.16052586e+03, 6.87257150e-02, -9.82378264e-05,
                               -3.50628312e-07, -2.21040482e-10, -3.63912864e-14};
        double att = 0, zp = 1;
        for (int p = 0; p < 6; p++) { att += fit[p] * zp; zp *= z; }
        if (att < 100.) att = 100.;
        L = att - 0.55 * (a.f / 1e-3 - 75);
    } else if (a.model == 5) {  // GL3 (:206-221): L = slope(depth) f + offset(depth)
        if (!gl3.t || gl3.n < 2) return NAN;
        L = gl3_interp(-z, gl3.t, gl3.t + gl3.n, gl3.n) * a.f + gl3_interp(-z, gl3.t, gl3.t + 2 * gl3.n, gl3.n);
    } else if (a.model == 4) {  // GL2 (:198-204)
        const double fit[6] = {1.20547286e+00, 1.58815679e-05, -2.58901767e-07,
                               -5.16435542e-10, -2.89124473e-13, -4.58987344e-17};
        double bulk = 852.0 + (-0.54 / 1e-3) * a.f;
        double poly = 0;
        for (int p = 5; p >= 0; p--) poly = poly * z + fit[p];
        L = bulk * poly;
    } else {  // MB1 (:224-244)
        const double R = 0.82, d_ice = 576.;
        L = 460. - 180. * a.f;
        L *= 1. / (1 + L / (2 * d_ice) * det_log(R));
        double d = -z * 420. / d_ice;
        double LL = (1250. * 0.08886 * det_exp(-0.048827 * (225.6746 - 86.517596 * (det_log(848.870 - (d)) / 2.302585092994046))));
        L *= LL / 231.21;
    }
    if (L < 1.) L = 1.;
    if (z > 0) L = INFINITY;
    return L;
}

// SP1: exponent x of L = 1 / exp(x), x = a + b ln f with (a, b) of the frequency branch (attenuation.py:168-192)
__device__ inline void sp1_coefficients(double z, double p[4])
{
    double z2 = fabs(z);
    double t_ = 1.83415e-09 * (z2 * z2 * z2) + (-1.59061e-08 * (z2 * z2)) + 0.00267687 * z2 + (-51.0696);
    const double w0 = -9.210340371976182, w2 = 1.1505720275988207;
    double b0 = -6.74890 + t_ * (0.026709 - t_ * 0.000884);
    double b1 = -6.22121 - t_ * (0.070927 + t_ * 0.001773);
    double b2 = -4.09468 - t_ * (0.002213 + t_ * 0.000332);
    p[0] = (b1 * w0 - b0 * 0.0) / (w0 - 0.0);
    p[1] = (b1 - b0) / (0.0 - w0);
    p[2] = (b2 * 0.0 - b1 * w2) / (0.0 - w2);
    p[3] = (b2 - b1) / (w2 - 0.0);
}

// ds / L(z, f).  SP1: L = max(1 / exp(x), 1), infinite above the surface, hence ds / L = ds * min(exp(x), 1) --
// the same number without the two divisions (the CPU checker's integrand is written the same way).
__device__ inline double sp1_ds_over_length(double ds, double z, double aa, double bb, double w)
{
    double e = det_exp_tab(aa + bb * w);   // (the integrand's own exp: detmath.h)
    if (e > 1.) e = 1.;
    if (z > 0) e = 0.;
    return ds * e;
}

__device__ inline double ds_over_length(double ds, double z, const AttLane& a)
{
    if (a.model == 100) return ds;  // path length only (the turning-point segment of the segment-sum integration)
    if (a.model == 1) {
        double p[4];
        sp1_coefficients(z, p);
        return sp1_ds_over_length(ds, z, (a.f < 1.) ? p[0] : p[2], (a.f < 1.) ? p[1] : p[3], a.w);
    }
    return ds / attenuation_length(z, a);
}

struct AttItem {
    double C0, z_turn;
    AttLane lane;
};

// dt(t) = ds(t) / L(z(t), f)  (analyticraytracing.py:986-988, :513-517)
__device__ inline double integrand(double t, const AttItem& it, const IceConst& m)
{
    double z = (t > it.z_turn) ? 2 * it.z_turn - t : t;
    double nz = n_of_z(z, m);
    double q = (it.C0 * it.C0) * (nz * nz);
    double yd = (q > 1) ? 1 / sqrt(q - 1) : INFINITY;
    double ds = sqrt(yd * yd + 1);
    return ds_over_length(ds, z, it.lane);
}

struct GK { double result, abserr, resabs, resasc; };

// 21-point Gauss-Kronrod rule, accumulation order of QUADPACK's DQK21
__device__ __noinline__ GK gk21(double a, double b, const AttItem& it, const IceConst& m)
{
    const double XGK[11] = {
        0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
        0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
        0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
        0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
        0.294392862701460198131126603103866, 0.148874338981631210884826001129720, 0.};
    const double WGK[11] = {
        0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
        0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
        0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
        0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
        0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
        0.149445554002916905664936468389821};
    const double WG[5] = {
        0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
        0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
        0.295524224714752870173815619188769};
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308;
    double fv1[10], fv2[10];
    double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double resg = 0.;
    double fc = integrand(centr, it, m);
    double resk = WGK[10] * fc;
    double resabs = fabs(resk);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int jtw = 2 * j + 1;
        double absc = hlgth * XGK[jtw];
        double f1 = integrand(centr - absc, it, m), f2 = integrand(centr + absc, it, m);
        fv1[jtw] = f1; fv2[jtw] = f2;
        double fsum = f1 + f2;
        resg += WG[j] * fsum;
        resk += WGK[jtw] * fsum;
        resabs += WGK[jtw] * (fabs(f1) + fabs(f2));
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int jtwm1 = 2 * j;
        double absc = hlgth * XGK[jtwm1];
        double f1 = integrand(centr - absc, it, m), f2 = integrand(centr + absc, it, m);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        double fsum = f1 + f2;
        resk += WGK[jtwm1] * fsum;
        resabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
    }
    double reskh = resk * 0.5;
    double resasc = WGK[10] * fabs(fc - reskh);
#pragma unroll
    for (int j = 0; j < 10; j++) resasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    GK o;
    o.result = resk * hlgth;
    o.resabs = resabs * dhlgth;
    o.resasc = resasc * dhlgth;
    o.abserr = fabs((resk - resg) * hlgth);
    if (o.resasc != 0. && o.abserr != 0.) {
        double r = 200. * o.abserr / o.resasc;
        o.abserr = o.resasc * fmin(1., r * sqrt(r));  // r ** 1.5
    }
    if (o.resabs > uflow / (50. * epmach)) o.abserr = fmax((epmach * 50.) * o.resabs, o.abserr);
    return o;
}

#define QLIM 50

// keep the error list ordered (QUADPACK DQPSRT); 1-based indices
__device__ inline void sort_errors(int last, int& maxerr, double& ermax, const double* elist, int* iord, int& nrmax)
{
    const int limit = QLIM;
    if (last <= 2) {
        iord[1] = 1;
        iord[2] = 2;
    } else {
        double errmax = elist[maxerr];
        if (nrmax != 1) {
            int ido = nrmax - 1;
            for (int i = 1; i <= ido; i++) {
                int isucc = iord[nrmax - 1];
                if (errmax <= elist[isucc]) break;
                iord[nrmax] = isucc;
                nrmax--;
            }
        }
        int jupbn = last;
        if (last > (limit / 2 + 2)) jupbn = limit + 3 - last;
        double errmin = elist[last];
        int jbnd = jupbn - 1;
        int ibeg = nrmax + 1;
        int i = ibeg;
        bool found = false;
        for (; i <= jbnd; i++) {
            int isucc = iord[i];
            if (errmax >= elist[isucc]) { found = true; break; }
            iord[i - 1] = isucc;
        }
        if (!found) {
            iord[jbnd] = maxerr;
            iord[jupbn] = last;
        } else {
            iord[i - 1] = maxerr;
            int k = jbnd;
            bool placed = false;
            for (int j = i; j <= jbnd; j++) {
                int isucc = iord[k];
                if (errmin < elist[isucc]) {
                    iord[k + 1] = last;
                    placed = true;
                    break;
                }
                iord[k + 1] = isucc;
                k--;
            }
            if (!placed) iord[i] = last;
        }
    }
    maxerr = iord[nrmax];
    ermax = elist[maxerr];
}

// Wynn epsilon algorithm (QUADPACK DQELG); epstab 1-based [1..52], res3la 1-based
__device__ inline void epsilon_extrap(int& n, double* epstab, double& result, double& abserr, double* res3la, int& nres)
{
    const double epmach = 2.220446049250313e-16, oflow = 1.7976931348623157e+308;
    nres++;
    abserr = oflow;
    result = epstab[n];
    if (n >= 3) {
        const int limexp = 50;
        epstab[n + 2] = epstab[n];
        int newelm = (n - 1) / 2;
        epstab[n] = oflow;
        int num = n, k1 = n;
        bool early = false;
        for (int i = 1; i <= newelm; i++) {
            int k2 = k1 - 1, k3 = k1 - 2;
            double res = epstab[k1 + 2];
            double e0 = epstab[k3], e1 = epstab[k2], e2 = res;
            double e1abs = fabs(e1);
            double delta2 = e2 - e1, err2 = fabs(delta2), tol2 = fmax(fabs(e2), e1abs) * epmach;
            double delta3 = e1 - e0, err3 = fabs(delta3), tol3 = fmax(e1abs, fabs(e0)) * epmach;
            if (!(err2 > tol2 || err3 > tol3)) {  // converged to machine accuracy
                result = res;
                abserr = fmax(err2 + err3, 5. * epmach * fabs(result));
                early = true;
                break;
            }
            double e3 = epstab[k1];
            epstab[k1] = e1;
            double delta1 = e1 - e3, err1 = fabs(delta1), tol1 = fmax(e1abs, fabs(e3)) * epmach;
            if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) { n = i + i - 1; break; }
            double ss = 1. / delta1 + 1. / delta2 - 1. / delta3;
            double epsinf = fabs(ss * e1);
            if (!(epsinf > 1e-4)) { n = i + i - 1; break; }
            res = e1 + 1. / ss;
            epstab[k1] = res;
            k1 -= 2;
            double error = err2 + fabs(res - e2) + err3;
            if (error > abserr) continue;
            abserr = error;
            result = res;
        }
        if (early) return;  // (abserr already floored)
        if (n == limexp) n = 2 * (limexp / 2) - 1;
        int ib = ((num / 2) * 2 == num) ? 2 : 1;
        int ie = newelm + 1;
        for (int i = 1; i <= ie; i++) {
            epstab[ib] = epstab[ib + 2];
            ib += 2;
        }
        if (num != n) {
            int indx = num - n + 1;
            for (int i = 1; i <= n; i++) epstab[i] = epstab[indx++];
        }
        if (nres < 4) {
            res3la[nres] = result;
            abserr = oflow;
        } else {
            abserr = fabs(result - res3la[3]) + fabs(result - res3la[2]) + fabs(result - res3la[1]);
            res3la[1] = res3la[2];
            res3la[2] = res3la[3];
            res3la[3] = result;
        }
    }
    abserr = fmax(abserr, 5. * epmach * fabs(result));
}

// ---- evaluation policies -------------------------------------------------------------------------------
// The adaptive driver below asks for Gauss-Kronrod estimates on one or two intervals at a time.  Two ways to
// provide them:
//   LaneEval   every lane evaluates its own 21 (42) integrand values;
//   GroupEval  the G lanes holding the frequencies of ONE ray cooperate: the subdivision pattern of a ray is
//              (almost always) the same for all its frequencies, and ds(z), the ice temperature polynomial etc. do
//              not depend on the frequency.  When every busy lane of the group asks for the same interval, lane n
//              computes the frequency-independent part of node n once and the group shares it by wave shuffles;
//              each lane then finishes its own 21-term sums (same values, same order -> same bits).  When lanes of a
//              group disagree on the interval (rare), each falls back to LaneEval.
struct NodeShared {
    double ds;     // path element ds/dz at the node
    double z;      // un-mirrored depth
    double p[4];   // model dependent, frequency independent (SP1: a, b of both frequency branches)
};

__device__ inline NodeShared node_shared(double t, const AttItem& it, const IceConst& m)
{
    NodeShared n;
    double z = (t > it.z_turn) ? 2 * it.z_turn - t : t;
    double nz = n_of_z(z, m);
    double q = (it.C0 * it.C0) * (nz * nz);
    double yd = (q > 1) ? 1 / sqrt(q - 1) : INFINITY;
    n.ds = sqrt(yd * yd + 1);
    n.z = z;
    n.p[0] = n.p[1] = n.p[2] = n.p[3] = 0.;
    if (it.lane.model == 1) sp1_coefficients(z, n.p);  // frequency independent: (a, b) of both branches
    return n;
}

// integrand value of one lane (frequency) from the shared node data: ds / L(z, f)
__device__ inline double node_finish(double ds, double z, double p0, double p1, double p2, double p3, const AttLane& a)
{
    if (a.model == 1) return sp1_ds_over_length(ds, z, (a.f < 1.) ? p0 : p2, (a.f < 1.) ? p1 : p3, a.w);
    return ds / attenuation_length(z, a);
}

// abscissa of node n (0 = centre, then QUADPACK's evaluation order: Gauss nodes first, then the Kronrod-only ones)
__device__ inline double gk_node(int n, double a, double b)
{
    const double XGK[11] = {
        0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
        0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
        0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
        0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
        0.294392862701460198131126603103866, 0.148874338981631210884826001129720, 0.};
    double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a);
    if (n == 0) return centr;
    int j = (n - 1) >> 1;                       // 0..9
    int idx = (j < 5) ? 2 * j + 1 : 2 * (j - 5);  // xgk index
    double absc = hlgth * XGK[idx];
    return ((n - 1) & 1) ? centr + absc : centr - absc;
}

// finish a 21-point rule from per-node integrand values f(n) (n as in gk_node); identical accumulation to gk21()
#ifndef ATT_BLOCK
#define ATT_BLOCK 128  // threads per block of the cooperative quadrature kernel
#endif
#ifndef ATT_WAVES
#define ATT_WAVES 2
#endif
#ifndef GK_UNROLL
#define GK_UNROLL 1
#endif
template <class F>
__device__ inline GK gk21_from_nodes(double a, double b, F&& fval, double* __restrict__ fbuf)
{
    const double WGK[11] = {
        0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
        0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
        0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
        0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
        0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
        0.149445554002916905664936468389821};
    const double WG[5] = {
        0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
        0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
        0.295524224714752870173815619188769};
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308;
    // the 20 node values needed again for resasc live in a lane-private LDS column (stride 256), not in registers
    double hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double resg = 0.;
    double fc = fval(0);
    double resk = WGK[10] * fc;
    double resabs = fabs(resk);
#pragma unroll GK_UNROLL
    for (int j = 0; j < 5; j++) {
        const int jtw = 2 * j + 1;
        double f1 = fval(1 + 2 * j), f2 = fval(2 + 2 * j);
        fbuf[jtw * ATT_BLOCK] = f1; fbuf[(10 + jtw) * ATT_BLOCK] = f2;
        double fsum = f1 + f2;
        resg += WG[j] * fsum;
        resk += WGK[jtw] * fsum;
        resabs += WGK[jtw] * (fabs(f1) + fabs(f2));
    }
#pragma unroll GK_UNROLL
    for (int j = 0; j < 5; j++) {
        const int jtwm1 = 2 * j;
        double f1 = fval(11 + 2 * j), f2 = fval(12 + 2 * j);
        fbuf[jtwm1 * ATT_BLOCK] = f1; fbuf[(10 + jtwm1) * ATT_BLOCK] = f2;
        double fsum = f1 + f2;
        resk += WGK[jtwm1] * fsum;
        resabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
    }
    double reskh = resk * 0.5;
    double resasc = WGK[10] * fabs(fc - reskh);
#pragma unroll GK_UNROLL
    for (int j = 0; j < 10; j++) resasc += WGK[j] * (fabs(fbuf[j * ATT_BLOCK] - reskh) + fabs(fbuf[(10 + j) * ATT_BLOCK] - reskh));
    GK o;
    o.result = resk * hlgth;
    o.resabs = resabs * dhlgth;
    o.resasc = resasc * dhlgth;
    o.abserr = fabs((resk - resg) * hlgth);
    if (o.resasc != 0. && o.abserr != 0.) {
        double r = 200. * o.abserr / o.resasc;
        o.abserr = o.resasc * fmin(1., r * sqrt(r));  // r ** 1.5
    }
    if (o.resabs > uflow / (50. * epmach)) o.abserr = fmax((epmach * 50.) * o.resabs, o.abserr);
    return o;
}

struct LaneEval {
#ifdef NRHIP_ATT_TIMING
    mutable unsigned long long at_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    __device__ inline bool any(bool busy) const { return busy; }
    __device__ inline void pair(bool want, bool two, double a1, double b1, double a2, double b2, const AttItem& it,
                                const IceConst& m, GK& g1, GK& g2) const
    {
        if (!want) return;
        g1 = gk21(a1, b1, it, m);
        if (two) g2 = gk21(a2, b2, it, m);
    }
};

#ifdef NRHIP_ATT_TIMING   // debug builds: shader clocks per phase (wave 0 of every block), read by nrhip_debug_att_clocks
__device__ unsigned long long g_att_clk[12];
#define AT_MARK(t) unsigned long long t = __builtin_amdgcn_s_memtime()
#define AT_ADD(e, i, t0) do { (e).at_acc[i] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#else
#define AT_MARK(t)
#define AT_ADD(e, i, t0)
#endif

template <int G, int MODEL = 0>
struct GroupEval {
    int lane, gl, gb;               // lane in wave, lane in group, first lane of the group
    unsigned long long gmask;       // lanes of this group
    NodeShared* nodes;              // LDS: 2 x 21 node records of this group
    double* fbuf;                   // LDS: this lane's column for the node values of one 21-point rule
#ifdef NRHIP_ATT_TIMING
    mutable unsigned long long at_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    __device__ inline void init(NodeShared* lds, double* lds_f)
    {
        fbuf = lds_f + threadIdx.x;
        lane = threadIdx.x & 63;
        gl = lane & (G - 1);
        gb = lane - gl;
        gmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << gb);
        nodes = lds + (threadIdx.x / G) * 42;
    }
    __device__ inline bool any(bool busy) const { return (__ballot(busy) & gmask) != 0ULL; }
    __device__ __forceinline__ void pair(bool want, bool two, double a1, double b1, double a2, double b2,
                                         const AttItem& it, const IceConst& m, GK& g1, GK& g2) const
    {
        unsigned long long wm = __ballot(want) & gmask;
        if (wm == 0ULL) return;
        int leader = __ffsll((long long)wm) - 1;
        double la1 = __shfl(a1, leader), lb1 = __shfl(b1, leader), la2 = __shfl(a2, leader), lb2 = __shfl(b2, leader);
        two = (__shfl((int)two, leader) != 0);  // group-uniform (lanes without work may carry another value)
        const double lC0 = __shfl(it.C0, leader), lzt = __shfl(it.z_turn, leader);  // the group's ray
        bool same = !want || (a1 == la1 && b1 == lb1 && (!two || (a2 == la2 && b2 == lb2)));
        bool allsame = ((__ballot(same) & gmask) == gmask);
        if (allsame) {
            // lane gl evaluates node gl of both intervals with the leader's (== everybody's) ray and interval and
            // leaves the record in LDS; the group is part of one wave, whose LDS operations execute in order
            AT_MARK(t_nodes);
            if (gl < 21) {
                AttItem li = it;
                li.C0 = lC0;
                li.z_turn = lzt;
                NodeShared n1 = node_shared(gk_node(gl, la1, lb1), li, m);
                // SP1: the integrand is ds * (z > 0 ? 0 : min(exp(x), 1)); the depth test is a property of the node, so it is
                // folded into ds here (ds * 0 for nodes above the surface: the same product, NaN for an infinite ds included)
                if (MODEL == 1 && n1.z > 0) n1.ds = n1.ds * 0.;
                nodes[gl] = n1;
                if (two) {
                    NodeShared n2 = node_shared(gk_node(gl, la2, lb2), li, m);
                    if (MODEL == 1 && n2.z > 0) n2.ds = n2.ds * 0.;
                    nodes[21 + gl] = n2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            AT_ADD(*this, 0, t_nodes);
            AT_MARK(t_fin);
            if (MODEL == 1) {
                // straight-line evaluation: x = a(z) + b(z) ln f stays within a few tens for any physical ray, so the range /
                // NaN branches of det_exp are checked once per rule (wave-uniform) instead of per node; a wave in which any
                // argument leaves [-700, 700] (or is NaN) repeats the rule with the general code below
                bool ok = true;
                GK f1, f2;
                f1.result = f1.abserr = f1.resabs = f1.resasc = 0.;
                f2 = f1;
                if (want) {
                    const int sel = !(it.lane.f < 1.) ? 2 : 0;
                    const double w = it.lane.w;
                    f1 = gk21_from_nodes(a1, b1, [&](int n) {
                        const NodeShared& s = nodes[n];
                        const double x = s.p[sel] + s.p[sel + 1] * w;
                        ok = ok && (fabs(x) <= 700.);
                        return s.ds * fmin(det_exp_tab_inrange(x, det_exp_tab64), 1.);   // (overflow rays only: the table from L1)
                    }, fbuf);
                    __builtin_amdgcn_sched_barrier(0);
                    if (two)
                        f2 = gk21_from_nodes(a2, b2, [&](int n) {
                            const NodeShared& s = nodes[21 + n];
                            const double x = s.p[sel] + s.p[sel + 1] * w;
                            ok = ok && (fabs(x) <= 700.);
                            return s.ds * fmin(det_exp_tab_inrange(x, det_exp_tab64), 1.);   // (overflow rays only: the table from L1)
                        }, fbuf);
                }
                if (__ballot(!ok) != 0ULL && want) {   // never on physical rays: the per-lane general code (same bits)
                    f1 = gk21(a1, b1, it, m);
                    if (two) f2 = gk21(a2, b2, it, m);
                }
                if (want) { g1 = f1; if (two) g2 = f2; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                AT_ADD(*this, 1, t_fin);
                return;
            }
            if (MODEL != 1 && want) {
                // every lane reads the 21 records (same address in all lanes: LDS broadcast) and finishes its own sums
                const int sel = (it.lane.model == 1 && !(it.lane.f < 1.)) ? 2 : 0;
                g1 = gk21_from_nodes(a1, b1, [&](int n) {
                    const NodeShared& s = nodes[n];
                    return node_finish(s.ds, s.z, s.p[sel], s.p[sel + 1], s.p[sel], s.p[sel + 1], it.lane);
                }, fbuf);
                __builtin_amdgcn_sched_barrier(0);  // finish interval 1 before interval 2: halves the live values
                if (two)
                    g2 = gk21_from_nodes(a2, b2, [&](int n) {
                        const NodeShared& s = nodes[21 + n];
                        return node_finish(s.ds, s.z, s.p[sel], s.p[sel + 1], s.p[sel], s.p[sel + 1], it.lane);
                    }, fbuf);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (want) {
            g1 = gk21(a1, b1, it, m);
            if (two) g2 = gk21(a2, b2, it, m);
        }
    }
};

// adaptive integration over [a, b] with optional interior break point (QUADPACK QAGS / QAGP decisions); every lane of
// an evaluation group must call this together (lanes without work pass valid = false).  Returns the integral estimate.
template <class EV, bool QAGP0 = false>
__device__ __forceinline__ double quad_gk21(bool valid, double a, double b, bool with_point, double point, const AttItem& it,
                                   const IceConst& m, int* neval_out, const EV& ev, bool qagp_no_point = false)
{
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308, oflow = 1.7976931348623157e+308;
    const double epsabs = 1.49e-8, epsrel = 1e-2;
    const int limit = QLIM;
    double alist[QLIM + 2], blist[QLIM + 2], rlist[QLIM + 2], elist[QLIM + 2];
    int iord[QLIM + 2];
    unsigned char level[QLIM + 2];
    double rlist2[53], res3la[4];
    double result = 0., abserr = 0., resabs = 0., errsum = 0., errbnd = 0., errmax = 0., area = 0., dres;
    double erlarg = 0., ertest = 0., correc = 0., small = 0., reseps = 0., abseps = 0.;
    int ier = 0, ierro = 0, iroff1 = 0, iroff2 = 0, iroff3 = 0, ksgn = 1, ktmin = 0, last = 0, maxerr = 1, neval = 0,
        nres = 0, nrmax = 1, numrl2 = 1, levmax = 1, levcur = 0;
    bool extrap = false, noext = false;
    double sign = 1.;
    // qagp_no_point: DQAGPE on one interval (what scipy runs when the requested break point lies outside (a, b))
    // (QAGP0 is a template flag so that the kernels which never need this mode compile exactly as before)
    const bool qagp = with_point || (QAGP0 && qagp_no_point);
    const int nint = (QAGP0 && !with_point) ? 1 : 2;
    bool busy = false;
    int exit_code = 0;  // 1: sum the list (label 190 / 115); 2: final-result logic (label 170 / 100)
    // ---- first estimate(s): (a, b) or (a, point), (point, b) ------------------------------------------------
    GK g1, g2;
    g1.result = g1.abserr = g1.resabs = g1.resasc = 0.;
    g2 = g1;
    {
        double lo = fmin(a, b), hi = fmax(a, b);
        double a1 = qagp ? lo : a, b1 = with_point ? point : (qagp ? hi : b), a2 = point, b2 = hi;
        ev.pair(valid, with_point, a1, b1, a2, b2, it, m, g1, g2);
        AT_MARK(t_init);
        if (valid && qagp) {
            if (a > b) sign = -1.;
            const GK gg[3] = {g1, g1, g2};
            const double aa[3] = {0., a1, a2}, bb[3] = {0., b1, b2};
            bool nd[3] = {false, false, false};
            for (int i = 1; i <= nint; i++) {
                const GK& g = gg[i];
                abserr += g.abserr;
                result += g.result;
                nd[i] = (g.abserr == g.resasc && g.abserr != 0.);
                resabs += g.resabs;
                level[i] = 0;
                elist[i] = g.abserr;
                alist[i] = aa[i];
                blist[i] = bb[i];
                rlist[i] = g.result;
                iord[i] = i;
            }
            for (int i = 1; i <= nint; i++) {
                if (nd[i]) elist[i] = abserr;
                errsum += elist[i];
            }
            last = nint;
            neval = 21 * nint;
            dres = fabs(result);
            errbnd = fmax(epsabs, epsrel * dres);
            if (abserr <= 100. * epmach * resabs && abserr > errbnd) ier = 2;
            if (nint == 2 && !(elist[iord[1]] > elist[iord[2]])) { int t = iord[1]; iord[1] = iord[2]; iord[2] = t; }
            if (!(ier != 0 || abserr <= errbnd)) {
                rlist2[1] = result;
                maxerr = iord[1];
                errmax = elist[maxerr];
                area = result;
                nrmax = 1;
                numrl2 = 1;
                erlarg = errsum;
                ertest = errbnd;
                abserr = oflow;
                ksgn = (dres >= (1. - 50. * epmach) * resabs) ? 1 : -1;
                last = nint + 1;
                busy = true;
            }
        } else if (valid) {
            result = g1.result;
            abserr = g1.abserr;
            double defabs = g1.resabs;
            dres = fabs(result);
            errbnd = fmax(epsabs, epsrel * dres);
            last = 1;
            alist[1] = a; blist[1] = b; rlist[1] = result; elist[1] = abserr; iord[1] = 1;
            if (abserr <= 100. * epmach * defabs && abserr > errbnd) ier = 2;
            if (ier != 0 || (abserr <= errbnd && abserr != g1.resasc) || abserr == 0.) {
                neval = 21;
            } else {
                rlist2[1] = result;
                errmax = abserr;
                maxerr = 1;
                area = result;
                errsum = abserr;
                abserr = oflow;
                nrmax = 1;
                numrl2 = 2;
                ksgn = (dres >= (1. - 50. * epmach) * defabs) ? 1 : -1;
                resabs = defabs;
                last = 2;
                busy = true;
            }
        }
        AT_ADD(ev, 7, t_init);
    }
    const bool entered_loop = busy;
    // The end points of the two intervals the last bisection made stay in registers: the next interval to bisect is nearly always
    // one of them, and the lists live in scratch (HBM latency on a dependent path).  ia / ib: their list slots (0: none).
    int ia = 0, ib = 0;
    double ca_lo = 0., ca_hi = 0., cb_lo = 0., cb_hi = 0.;
    auto end_points = [&](int idx, double& lo, double& hi) {
        if (idx == ia) { lo = ca_lo; hi = ca_hi; }
        else if (idx == ib) { lo = cb_lo; hi = cb_hi; }
        else { lo = alist[idx]; hi = blist[idx]; }
    };
    // ---- main loop: bisect the interval with the largest error estimate --------------------------------------
    while (ev.any(busy)) {
        double a1 = 0., b1 = 0., a2 = 0., b2 = 0., erlast = 0.;
        AT_MARK(t_top);
        if (busy) {
            if (qagp) levcur = level[maxerr] + 1;
            double lo_, hi_;
            end_points(maxerr, lo_, hi_);
            a1 = lo_;
            b1 = 0.5 * (lo_ + hi_);
            a2 = b1;
            b2 = hi_;
            erlast = errmax;
        }
        AT_ADD(ev, 3, t_top);
        ev.pair(busy, true, a1, b1, a2, b2, it, m, g1, g2);
        AT_MARK(t_book);
        if (busy) {
            do {
                neval += 42;
                double area12 = g1.result + g2.result;
                double erro12 = g1.abserr + g2.abserr;
                errsum = errsum + erro12 - errmax;
                area = area + area12 - rlist[maxerr];
                if (g1.resasc != g1.abserr && g2.resasc != g2.abserr) {
                    if (fabs(rlist[maxerr] - area12) <= 1e-5 * fabs(area12) && erro12 >= 0.99 * errmax) {
                        if (extrap) iroff2++;
                        else iroff1++;
                    }
                    if (last > 10 && erro12 > errmax) iroff3++;
                }
                if (qagp) {
                    level[maxerr] = (unsigned char)levcur;
                    level[last] = (unsigned char)levcur;
                }
                rlist[maxerr] = g1.result;
                rlist[last] = g2.result;
                errbnd = fmax(epsabs, epsrel * fabs(area));
                if (iroff1 + iroff2 >= 10 || iroff3 >= 20) ier = 2;
                if (iroff2 >= 5) ierro = 3;
                if (last == limit) ier = 1;
                if (fmax(fabs(a1), fabs(b2)) <= (1. + 100. * epmach) * (fabs(a2) + 1000. * uflow)) ier = 4;
                ia = maxerr;
                ib = last;
                if (g2.abserr > g1.abserr) {
                    alist[maxerr] = a2;
                    blist[maxerr] = b2;   // (unchanged; written so that slot and registers hold the same pair)
                    alist[last] = a1;
                    blist[last] = b1;
                    rlist[maxerr] = g2.result;
                    rlist[last] = g1.result;
                    elist[maxerr] = g2.abserr;
                    elist[last] = g1.abserr;
                    ca_lo = a2; ca_hi = b2; cb_lo = a1; cb_hi = b1;
                } else {
                    alist[last] = a2;
                    blist[maxerr] = b1;
                    blist[last] = b2;
                    elist[maxerr] = g1.abserr;
                    elist[last] = g2.abserr;
                    ca_lo = a1; ca_hi = b1; cb_lo = a2; cb_hi = b2;
                }
                AT_ADD(ev, 4, t_book);
                AT_MARK(t_sort);
                sort_errors(last, maxerr, errmax, elist, iord, nrmax);
                AT_ADD(ev, 5, t_sort);
                if (errsum <= errbnd) { exit_code = 1; busy = false; break; }
                if (ier != 0) { exit_code = 2; busy = false; break; }
                if (!qagp && last == 2) {
                    small = fabs(b - a) * 0.375;
                    erlarg = errsum;
                    ertest = errbnd;
                    rlist2[2] = area;
                    break;  // continue
                }
                if (noext) break;  // continue
                erlarg -= erlast;
                if (qagp) { if (levcur + 1 <= levmax) erlarg += erro12; }
                else      { if (fabs(b1 - a1) > small) erlarg += erro12; }
                if (!extrap) {
                    double lo_ = 0., hi_ = 0.;
                    if (!qagp) end_points(maxerr, lo_, hi_);
                    bool is_smallest = qagp ? !(level[maxerr] + 1 <= levmax) : !(fabs(hi_ - lo_) > small);
                    if (!is_smallest) break;  // continue
                    extrap = true;
                    nrmax = 2;
                }
                if (!(ierro == 3 || erlarg <= ertest)) {
                    int jupbnd = last;
                    if (last > (2 + limit / 2)) jupbnd = limit + 3 - last;
                    bool cont = false;
                    for (int k = nrmax; k <= jupbnd; k++) {
                        maxerr = iord[nrmax];
                        errmax = elist[maxerr];
                        double lo_ = 0., hi_ = 0.;
                        if (!qagp) end_points(maxerr, lo_, hi_);
                        bool big = qagp ? (level[maxerr] + 1 <= levmax) : (fabs(hi_ - lo_) > small);
                        if (big) { cont = true; break; }
                        nrmax++;
                    }
                    if (cont) break;  // continue
                }
                numrl2++;
                rlist2[numrl2] = area;
                bool skip_eps = qagp && numrl2 <= 2;
                if (!skip_eps) {
                    epsilon_extrap(numrl2, rlist2, reseps, abseps, res3la, nres);
                    ktmin++;
                    if (ktmin > 5 && abserr < 1e-3 * errsum) ier = 5;
                    if (abseps < abserr) {
                        ktmin = 0;
                        abserr = abseps;
                        result = reseps;
                        correc = erlarg;
                        ertest = fmax(epsabs, epsrel * fabs(reseps));
                        if (qagp ? (abserr < ertest) : (abserr <= ertest)) { exit_code = 2; busy = false; break; }
                    }
                    if (numrl2 == 1) noext = true;
                    if (qagp ? (ier >= 5) : (ier == 5)) { exit_code = 2; busy = false; break; }
                }
                maxerr = iord[1];
                errmax = elist[maxerr];
                nrmax = 1;
                extrap = false;
                if (qagp) levmax++;
                else small *= 0.5;
                erlarg = errsum;
            } while (0);
            if (busy) {
                last++;
                if (last > limit) busy = false;  // cannot happen (ier = 1 at last == limit), kept as a guard
            }
        }
        AT_ADD(ev, 6, t_book);   // everything after the rules of this bisection (includes 4 and 5)
    }
    AT_MARK(t_fin2);
    if (valid && entered_loop) {
        if (last > limit) last = limit;
        bool sum_list = (exit_code == 1);
        if (!sum_list) {  // label 170 / 100
            if (abserr == oflow) sum_list = true;
            else {
                bool to_div_test = true;
                if (ier + ierro != 0) {
                    if (ierro == 3) abserr += correc;
                    if (ier == 0) ier = 3;
                    if (result != 0. && area != 0.) {
                        if (abserr / fabs(result) > errsum / fabs(area)) { sum_list = true; to_div_test = false; }
                    } else {
                        if (abserr > errsum) { sum_list = true; to_div_test = false; }
                        else if (area == 0.) to_div_test = false;
                    }
                }
                if (to_div_test) {
                    if (!(ksgn == -1 && fmax(fabs(result), fabs(area)) <= resabs * 0.01)) {
                        if (0.01 > (result / area) || (result / area) > 100. || errsum > fabs(area)) ier = 6;
                    }
                }
            }
        }
        if (sum_list) {
            // (the list lives in scratch: eight loads are issued before the first addition waits for one; same order of additions)
            result = 0.;
            for (int k0 = 1; k0 <= last; k0 += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = rlist[min(k0 + j, QLIM + 1)];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (k0 + j <= last) result += v[j];
            }
            abserr = errsum;
        }
        if (!qagp) neval = 42 * last - 21;
    }
    if (qagp) result *= sign;
    if (neval_out) *neval_out = neval;
    AT_ADD(ev, 8, t_fin2);
    return result;
}

// zint: per ray {z_start, z_stop_mirrored, z_turn}; one lane per (ray, frequency)
__global__ void __launch_bounds__(256, 2)
attenuation_kernel(long n_rays, const double* __restrict__ C0, const double* __restrict__ zint, int n_freq,
                   const double* __restrict__ freqs, int model, IceConst m, double* __restrict__ att,
                   int* __restrict__ neval, const int* __restrict__ ray_index)
{
    long n_items = n_rays * n_freq;
    LaneEval ev;
    for (long it0 = blockIdx.x * (long)blockDim.x + threadIdx.x; it0 < n_items;
         it0 += (long)gridDim.x * blockDim.x) {
        long ray = it0 / n_freq;
        int jf = (int)(it0 - ray * n_freq);
        if (ray_index) ray = ray_index[ray];  // optional indirection: only the listed rays are integrated
        const long item = ray * n_freq + jf;
        AttItem it;
        it.C0 = C0[ray];
        double z1 = zint[3 * ray], z2m = zint[3 * ray + 1];
        it.z_turn = zint[3 * ray + 2];
        it.lane.model = model;
        it.lane.f = freqs[jf];
        it.lane.w = det_log(it.lane.f);
        if (isnan(it.C0)) {
            att[item] = NAN;
            if (neval) neval[item] = 0;
            continue;
        }
        bool with_point = (z1 < it.z_turn && it.z_turn < z2m);
        int ne;
        double integral = quad_gk21(true, z1, z2m, with_point, it.z_turn, it, m, &ne, ev);
        att[item] = det_exp(-1 * integral);
        if (neval) neval[item] = ne;
    }
}

// G lanes per ray (G = 32 for n_freq <= 32, else 64): the cooperative evaluation of GroupEval
template <int G, int MODEL>
__global__ void __launch_bounds__(ATT_BLOCK, ATT_WAVES)
attenuation_group_kernel(long n_rays, const double* __restrict__ C0, const double* __restrict__ zint, int n_freq,
                         const double* __restrict__ freqs, int model, IceConst m, double* __restrict__ att,
                         int* __restrict__ neval, const int* __restrict__ ray_index,
                         unsigned long long* __restrict__ eval_counter, const int* __restrict__ n_rays_dev = nullptr)
{
    __shared__ NodeShared sh_nodes[(ATT_BLOCK / G) * 42];
    if (n_rays_dev) n_rays = *n_rays_dev;   // the overflow list of attenuation_dense_kernel (length known on the device only)
    __shared__ double sh_f[20 * ATT_BLOCK];
    GroupEval<G, MODEL> ev;
    ev.init(sh_nodes, sh_f);
    model = MODEL;  // compile-time: the branches on the ice model fold away
    unsigned long long my_evals = 0;
    AT_MARK(t_all);
    const long groups_per_block = blockDim.x / G;
    const long n_iter = (n_rays + (long)gridDim.x * groups_per_block - 1) / ((long)gridDim.x * groups_per_block);
    // the lane's frequency (and its logarithm) is the same for every ray; the parameters of the NEXT ray are requested while the
    // current one is integrated (they come from HBM: one exposed latency per ray otherwise)
    const int jf = threadIdx.x & (G - 1);
    const double lane_f = (jf < n_freq) ? freqs[jf] : 1.;
    const double lane_w = det_log(lane_f);
    struct RayPar { bool ok; long ray; double C0, z1, z2m, zt; };
    auto fetch = [&](long iter) -> RayPar {
        RayPar r{false, 0, NAN, 0., 0., 0.};
        if (iter >= n_iter) return r;
        const long g = (iter * gridDim.x + blockIdx.x) * groups_per_block + threadIdx.x / G;
        r.ok = g < n_rays;
        if (r.ok) {
            r.ray = ray_index ? ray_index[g] : g;
            r.C0 = C0[r.ray];
            r.z1 = zint[3 * r.ray];
            r.z2m = zint[3 * r.ray + 1];
            r.zt = zint[3 * r.ray + 2];
        }
        return r;
    };
    RayPar nxt = fetch(0);
    for (long iter = 0; iter < n_iter; iter++) {  // uniform trip count: every lane takes part in the shuffles
        const RayPar cur = nxt;
        nxt = fetch(iter + 1);
        const bool ray_ok = cur.ok;
        const long ray = cur.ray;
        AttItem it;
        it.C0 = cur.C0;
        const double z1 = cur.z1, z2m = cur.z2m;
        it.z_turn = cur.zt;
        it.lane.model = model;
        it.lane.f = lane_f;
        it.lane.w = lane_w;
        bool valid = ray_ok && jf < n_freq && !isnan(it.C0);
        bool with_point = (z1 < it.z_turn && it.z_turn < z2m);
        int ne = 0;
        AT_MARK(t_q);
        double integral = quad_gk21(valid, z1, z2m, with_point, it.z_turn, it, m, &ne, ev);
        AT_ADD(ev, 9, t_q);
        AT_MARK(t_st);
        if (ray_ok && jf < n_freq) {
            const long item = ray * n_freq + jf;
            att[item] = valid ? det_exp(-1 * integral) : NAN;
            if (neval) neval[item] = ne;
            my_evals += (unsigned long long)ne;
        }
        AT_ADD(ev, 10, t_st);
    }
#ifdef NRHIP_ATT_TIMING
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < 12; i++)
            if (i != 2) atomicAdd(&g_att_clk[i], ev.at_acc[i]);
        atomicAdd(&g_att_clk[2], __builtin_amdgcn_s_memtime() - t_all);
    }
#endif
    if (eval_counter) {  // integrand evaluations QUADPACK would count (for the FP64 rate reported by bench.py)
        for (int off = 32; off > 0; off >>= 1) my_evals += __shfl_xor(my_evals, off);
        if ((threadIdx.x & 63) == 0 && my_evals) atomicAdd(eval_counter, my_evals);   // (an empty overflow pass must not queue 4096 atomics on one word)
    }
}

#include "attenuation_dense.h"

// ---------------------------------------------------------------------------------------------------------
// The reference's speed-optimised path integral for the models in speedup_attenuation_models (GL3),
// analyticraytracing.py:998-1064: the path is cut into ~10 m depth segments (np.linspace), each contributes
// ds(mid) / L(z(mid), f) * width; the segment around the turning depth (+- 10 m, where ds diverges) is replaced by
// quad(ds) / L(z_turn, f) -- one QUADPACK run per ray, not per frequency.  32 lanes per ray: lane j evaluates the
// frequency-independent part of segment c * 32 + j, the lanes (= frequencies) then add the 32 terms in order.
// ---------------------------------------------------------------------------------------------------------
__device__ inline int n_steps_of(double a, double b, double dx)
{
    int n = (int)floor(fabs(a - b) / dx);
    return n < 3 ? 3 : n;
}
// point i of np.linspace(a, b, n) ([a] if a == b, n = 1)
__device__ inline double linspace_at(double a, double b, int n, int i)
{
    if (n == 1) return a;
    if (i == n - 1) return b;
    return i * ((b - a) / (n - 1)) + a;
}

__global__ void __launch_bounds__(256)
attenuation_segments_kernel(long n_rays, const double* __restrict__ C0, const double* __restrict__ zint, int n_freq,
                            const double* __restrict__ freqs, int model, IceConst m, double* __restrict__ att,
                            int* __restrict__ neval, const int* __restrict__ ray_index, const double* __restrict__ gl3,
                            int gl3_n)
{
    __shared__ double sh[8][32][3];  // per group: ds, width and (un-mirrored) depth of 32 segments
    const int G = 32, grp = threadIdx.x / G, jf = threadIdx.x & (G - 1);
    const long groups_per_block = blockDim.x / G;
    const long n_iter = (n_rays + (long)gridDim.x * groups_per_block - 1) / ((long)gridDim.x * groups_per_block);
    LaneEval ev;
    for (long iter = 0; iter < n_iter; iter++) {
        long g = (iter * gridDim.x + blockIdx.x) * groups_per_block + grp;
        const bool ray_ok = g < n_rays;
        long ray = 0;
        if (ray_ok) ray = ray_index ? ray_index[g] : g;
        const double c0 = ray_ok ? C0[ray] : NAN;
        const double z1 = ray_ok ? zint[3 * ray] : 0., z2m = ray_ok ? zint[3 * ray + 1] : 0., zt = ray_ok ? zint[3 * ray + 2] : 0.;
        const bool valid = ray_ok && !isnan(c0);
        AttLane lane;
        lane.model = model;
        lane.f = (jf < n_freq) ? freqs[jf] : 1.;
        lane.w = 0.;
        const Gl3Tab tab{gl3, gl3_n};
        const double dx = 10., window = 20.;
        const bool fallback = (z1 - window / 2 < zt && zt < z2m + window / 2);
        // step list: linspace(z1, w0) ++ linspace(w1, z2m) (fallback) or linspace(z1, z2m)
        const double w0 = fmax(z1, zt - window / 2), w1 = fmin(zt + window / 2, z2m);
        const double a0 = z1, b0 = fallback ? w0 : z2m, a1 = w1, b1 = z2m;
        const int n0 = (a0 == b0) ? 1 : n_steps_of(a0, b0, dx);
        const int n1 = fallback ? ((a1 == b1) ? 1 : n_steps_of(a1, b1, dx)) : 0;
        const int n = n0 + n1;
        auto step_at = [&](int i) { return i < n0 ? linspace_at(a0, b0, n0, i) : linspace_at(a1, b1, n1, i - n0); };
        int idx = -2;
        double integrand = 0.;
        int ne_quad = 0;
        if (valid && fallback) {
            int cnt = 0;  // np.digitize(z_turn, steps) - 1
            for (int i = 0; i < n; i++) cnt += (step_at(i) <= zt) ? 1 : 0;
            idx = cnt - 1;
            if (idx == n - 1) idx -= 1;
            else if (idx == -1) idx = 0;
            const double lo = step_at(idx), hi = step_at(idx + 1);
            const bool inside = (fmin(lo, hi) < zt && zt < fmax(lo, hi));
            AttItem it;
            it.C0 = c0;
            it.z_turn = zt;
            it.lane = lane;
            it.lane.model = 100;  // ds only
            integrand = quad_gk21<LaneEval, true>(true, lo, hi, inside, zt, it, m, &ne_quad, ev, !inside);
        }
        double sum = 0.;
        const int n_seg = valid ? n - 1 : 0;
        for (int c0i = 0; c0i < n_seg; c0i += G) {  // (trip count differs between the groups of a block: no block barriers)
            const int i = c0i + jf;
            double dsdx = 0., zu = 0.;
            if (i < n_seg) {
                const double s0 = step_at(i), dxa = step_at(i + 1) - s0, mid = s0 + dxa / 2;
                zu = (mid > zt) ? 2 * zt - mid : mid;
                const double nz = n_of_z(zu, m);
                const double q = (c0 * c0) * (nz * nz);
                const double yd = (q > 1) ? 1 / sqrt(q - 1) : INFINITY;
                dsdx = sqrt(yd * yd + 1);
                sh[grp][jf][1] = dxa;
                sh[grp][jf][2] = zu;
            }
            sh[grp][jf][0] = dsdx;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = 0; k < G && c0i + k < n_seg; k++) {
                const double ds_k = sh[grp][k][0], dxa_k = sh[grp][k][1], z_k = sh[grp][k][2];
                double term;
                if (c0i + k == idx) term = integrand / attenuation_length(zt, lane, tab);
                else term = ds_k / attenuation_length(z_k, lane, tab) * dxa_k;
                sum += term;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (ray_ok && jf < n_freq) {
            const long item = ray * n_freq + jf;
            att[item] = valid ? det_exp(-1 * sum) : NAN;
            if (neval) neval[item] = valid ? n - 1 + ne_quad : 0;
        }
    }
}

// per-ray integration limits from (pair geometry, C0): {z1, get_z_mirrored(...)[1], z_turn}
__global__ void __launch_bounds__(256)
ray_limits_kernel(long n_rays, const double* __restrict__ x1, const double* __restrict__ x2,
                  const double* __restrict__ C0, IceConst m, double* __restrict__ zint)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    double A[3] = {x1[3 * i], x1[3 * i + 1], x1[3 * i + 2]};
    double B[3] = {x2[3 * i], x2[3 * i + 1], x2[3 * i + 2]};
    if (B[2] < A[2]) {
        for (int d = 0; d < 3; d++) { double t = A[d]; A[d] = B[d]; B[d] = t; }
    }
    double dX[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
    double rho = sqrt(dX[0] * dX[0] + dX[1] * dX[1]);
    double cph = 1., sph = 0.;
    if (rho > 0) {
        cph = dX[0] / rho;
        sph = -(dX[1] / rho);
    }
    Pair2D p;
    p.y1 = A[0];
    p.z1 = A[2];
    p.y2 = (cph * dX[0] + (-sph) * dX[1] + 0 * dX[2]) + A[0];
    p.z2 = (0 * dX[0] + 0 * dX[1] + 1 * dX[2]) + A[2];
    p.g1 = gamma_of_z(p.z1, m);
    p.g2 = gamma_of_z(p.z2, m);
    double c0 = C0[i];
    if (isnan(c0)) {
        zint[3 * i] = zint[3 * i + 1] = zint[3 * i + 2] = NAN;
        return;
    }
    C0State st = make_c0(c0, m);
    double C1 = C1_of(st, p, m);
    zint[3 * i] = p.z1;
    zint[3 * i + 1] = z_mirrored(p.y2, p.z2, st, C1, p);
    zint[3 * i + 2] = st.z_turn;
}

__global__ void attenuation_length_kernel(long n, const double* __restrict__ z, const double* __restrict__ f,
                                          int model, double* __restrict__ L, const double* __restrict__ gl3, int gl3_n)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    AttLane a;
    a.model = model;
    a.f = f[i];
    a.w = det_log(a.f);
    L[i] = attenuation_length(z[i], a, Gl3Tab{gl3, gl3_n});
}

void launch_attenuation_length(hipStream_t stream, long n, const double* z, const double* f, int model, double* L,
                               const double* gl3, int gl3_n)
{
    if (n <= 0) return;
    int block = 256;
    long grid = (n + block - 1) / block;
    hipLaunchKernelGGL(attenuation_length_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n, z, f, model, L, gl3, gl3_n);
}

void launch_ray_limits(hipStream_t stream, long n_rays, const double* x1, const double* x2, const double* C0,
                       const IceConst& m, double* zint)
{
    if (n_rays <= 0) return;
    int block = 256;
    long grid = (n_rays + block - 1) / block;
    hipLaunchKernelGGL(ray_limits_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_rays, x1, x2, C0, m, zint);
}

void launch_attenuation_items(hipStream_t stream, long n_rays, const double* C0, const double* zint, int n_freq,
                              const double* freqs, int model, const IceConst& m, double* att, int* neval,
                              const int* ray_index, unsigned long long* eval_counter, const double* gl3, int gl3_n,
                              int* overflow)
{
    long n_items = n_rays * n_freq;
    if (n_items <= 0) return;
    int block = 256;
    if (model == 5) {  // speedup_attenuation_models = ["GL3"]: segment sums (n_freq <= 32 checked by the caller)
        long grid = (n_rays + 7) / 8;
        if (grid > 256L * 64) grid = 256L * 64;
        hipLaunchKernelGGL(attenuation_segments_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_rays, C0, zint, n_freq,
                           freqs, model, m, att, neval, ray_index, gl3, gl3_n);
        return;
    }
    if (overflow && model != 5 && n_freq <= 32 && !getenv("NRHIP_ATT_LEGACY") && !getenv("NRHIP_ATT_LANES")) {
        // dense packing + LDS lists (attenuation_dense.h); overflow[0]: number of rays it left over, overflow[1..]: those rays,
        // integrated by the general kernel afterwards
        const DenseMap map = make_dense_map(n_freq);
        const long n_pairs = (n_rays + map.rays_per_pair - 1) / map.rays_per_pair;
        long grid = 2 * n_pairs;
        if (grid > 256L * 128) grid = 256L * 128;
        (void)hipMemsetAsync(overflow, 0, sizeof(int), stream);
#define NRHIP_ATTD_LAUNCH(MM)                                                                                          \
    hipLaunchKernelGGL((attenuation_dense_kernel<MM>), dim3((unsigned)grid), dim3(64), 0, stream, n_rays, C0, zint, n_freq, \
                       freqs, m, att, neval, ray_index, eval_counter, map, overflow, overflow + 1)
        switch (model) {
            case 1: NRHIP_ATTD_LAUNCH(1); break;
            case 2: NRHIP_ATTD_LAUNCH(2); break;
            case 4: NRHIP_ATTD_LAUNCH(4); break;
            default: NRHIP_ATTD_LAUNCH(3); break;
        }
        long grid2 = (n_rays + ATT_BLOCK / 32 - 1) / (ATT_BLOCK / 32);
        if (grid2 > 512) grid2 = 512;   // the overflow list is empty or short (its length is known on the device only)
#define NRHIP_ATTO_LAUNCH(MM)                                                                                              \
    hipLaunchKernelGGL((attenuation_group_kernel<32, MM>), dim3((unsigned)grid2), dim3(ATT_BLOCK), 0, stream, 2 * n_rays, C0, zint, \
                       n_freq, freqs, model, m, att, neval, overflow + 1, eval_counter, overflow)
        switch (model) {
            case 1: NRHIP_ATTO_LAUNCH(1); break;
            case 2: NRHIP_ATTO_LAUNCH(2); break;
            case 4: NRHIP_ATTO_LAUNCH(4); break;
            default: NRHIP_ATTO_LAUNCH(3); break;
        }
        return;
    }
    if (n_freq <= 64 && !getenv("NRHIP_ATT_LANES")) {
        block = ATT_BLOCK;
        int G = (n_freq <= 32) ? 32 : 64;
        long grid = (n_rays + block / G - 1) / (block / G);
        if (grid > 256L * 64) grid = 256L * 64;
#define NRHIP_ATT_LAUNCH(GG, MM)                                                                                   \
    hipLaunchKernelGGL((attenuation_group_kernel<GG, MM>), dim3((unsigned)grid), dim3(block), 0, stream, n_rays, C0, \
                       zint, n_freq, freqs, model, m, att, neval, ray_index, eval_counter)
#define NRHIP_ATT_MODELS(GG)                                                              \
    switch (model) {                                                                      \
        case 1: NRHIP_ATT_LAUNCH(GG, 1); break;                                           \
        case 2: NRHIP_ATT_LAUNCH(GG, 2); break;                                           \
        case 4: NRHIP_ATT_LAUNCH(GG, 4); break;                                           \
        default: NRHIP_ATT_LAUNCH(GG, 3); break;                                          \
    }
        if (G == 32) { NRHIP_ATT_MODELS(32) } else { NRHIP_ATT_MODELS(64) }
        return;
    }
    long grid = (n_items + block - 1) / block;
    if (grid > 256L * 64) grid = 256L * 64;
    hipLaunchKernelGGL(attenuation_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_rays, C0, zint, n_freq,
                       freqs, model, m, att, neval, ray_index);
}

}  // namespace nrhip

#ifdef NRHIP_ATT_TIMING
extern "C" int nrhip_debug_att_clocks(unsigned long long* out12, int reset)
{
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(nrhip::g_att_clk), 12 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[12] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nrhip::g_att_clk), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
