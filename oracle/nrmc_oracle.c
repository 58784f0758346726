/*
 * oracle/nrmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's analytic ray tracer
 * (NuRadioMC/SignalProp/analyticraytracing.py, pure-Python path, use_cpp=False) and of the
 * attenuation-length models (NuRadioMC/utilities/attenuation.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this
 * file; the product path (nuradiomc_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *   - the reference's own golden vectors (tests/golden/ref_C0_SP.npz from
 *     NuRadioMC/test/SignalProp/reference_C0.pkl; ref_single_events_1e18.npz from
 *     NuRadioMC/test/SingleEvents/1e18_output_reference.hdf5), and
 *   - outputs of the reference itself run in the build container
 *     (tests/golden/raytrace_{A,B,C}.npz, generator tests/golden/gen/gen_raytrace.py).
 *
 * The reference delegates three numerical kernels to SciPy (unpinned "*" in pyproject.toml; the
 * build container has SciPy 1.15.3).  Their published algorithms are restated here:
 *   - scipy.optimize.root(method='hybr')  = MINPACK HYBRD (More', Garbow, Hillstrom 1980), n = 1
 *     (call site analyticraytracing.py:1479)
 *   - scipy.optimize.brentq               = Brent 1973 as implemented in scipy/optimize/Zeros/brentq.c
 *     (call sites :1504, :1526)
 *   - scipy.integrate.quad                = QUADPACK DQAGSE / DQAGPE with DQK21, DQELG, DQPSRT
 *     (Piessens et al. 1983) (call site :1071)
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "detmath_c.h"

typedef struct { double n_ice, delta_n, z_0; } ice_t;

#define SPEED_OF_LIGHT 0.299792458 /* m/ns, analyticraytracing.py:56 */

/* ------------------------------------------------------------------------------------------
 * 2-D analytic ray path helpers (analyticraytracing.py:99-370).
 *
 * Arithmetic conventions (so that this file is bit-reproducible on any IEEE-754 machine):
 * exp / log are orc_exp / orc_log (detmath_c.h); x ** 0.5 is sqrt(x), x ** -0.5 is 1 / sqrt(x), x ** -2 is
 * 1 / (x * x); sin / cos of arctan(.) are taken algebraically.  Each differs from the reference's libm call by
 * at most ~1 ulp -- far below the reference's own 1e-7 first-root noise -- and the golden-vector tests pin it.
 * ---------------------------------------------------------------------------------------- */
static double n_of_z(double z, const ice_t *m) { return m->n_ice - m->delta_n * orc_exp(z / m->z_0); } /* :358 */
static double get_gamma(double z, const ice_t *m) { return m->delta_n * orc_exp(z / m->z_0); }        /* :127 */
static double C0_from_log(double logC0, const ice_t *m) { return orc_exp(logC0) + 1. / m->n_ice; }    /* :99 */

/* :105-125 */
static double get_y(double gamma, double C0, double C1, const ice_t *m)
{
    double b = 2 * m->n_ice;
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double root = fabs(gamma * gamma - gamma * b + c);
    double logargument = gamma / (2 * sqrt(c) * sqrt(root) - b * gamma + 2 * c);
    double pref = m->z_0 / sqrt(m->n_ice * m->n_ice * C0 * C0 - 1);
    return pref * orc_log(logargument) + C1;
}

/* :133-158 */
static void get_turning_point(double c, const ice_t *m, double *gamma_turn, double *z_turn)
{
    double b = 2 * m->n_ice;
    double gamma2 = b * 0.5 - sqrt(0.25 * b * b - c);
    double z2 = orc_log(gamma2 / m->delta_n) * m->z_0;
    if (z2 > 0) {
        z2 = 0;
        gamma2 = m->delta_n; /* get_gamma(0) = delta_n * exp(0) */
    }
    *gamma_turn = gamma2;
    *z_turn = z2;
}

/* :160-184 */
static double get_y_with_z_mirror(double z, double C0, const ice_t *m, double C1)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double y_turn = get_y(gamma_turn, C0, C1, m);
    if (z < z_turn)
        return get_y(get_gamma(z, m), C0, C1, m);
    return 2 * y_turn - get_y(get_gamma(2 * z_turn - z, m), C0, C1, m);
}

static double get_C1(const double x1[2], double C0, const ice_t *m) /* :487 */
{
    return x1[0] - get_y_with_z_mirror(x1[1], C0, m, 0.0);
}

/* :186-202 */
static double get_y_turn(double C0, const double x1[2], const ice_t *m)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double C1 = x1[0] - get_y_with_z_mirror(x1[1], C0, m, 0.0);
    return get_y(gamma_turn, C0, C1, m);
}

/* :281-291: where the ray (C0, C1) comes back down to the reflective layer at depth z_refl */
static void get_reflection_point(double C0, double C1, const ice_t *m, double z_refl, double out[2])
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    out[1] = z_refl;
    out[0] = get_y_with_z_mirror(-z_refl + 2 * z_turn, C0, m, C1);
}

/* :204-272.  reflection = number of reflections off the bottom layer at z_refl, reflection_case 1: the ray starts
 * upwards, 2: downwards (the start point is moved to the left, to where an upward ray passes through x1 going down).
 * The start point is a LOCAL COPY here, as in the C++ twin (analytic_raytracing.cpp:405-470, which the reference's own
 * golden table reference_C0_MooresBay.pkl was written with).  The Python text shifts x1[0] in place (:226-229) on the
 * one array scipy passes to every objective evaluation, so there the shift accumulates from call to call and the
 * reflection_case = 2 roots are lost (tests/golden/gen/gen_mooresbay.py measures that: a handful found of 1996). */
static double get_delta_y_refl(double C0, const double x1_in[2], const double x2[2], const ice_t *m, int reflection,
                               int reflection_case, double z_refl)
{
    if (C0 < 1. / m->n_ice || C0 > INFINITY)
        return -INFINITY;
    double x1[2] = { x1_in[0], x1_in[1] };
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    if (reflection > 0 && reflection_case == 2) {
        double y_turn = get_y_turn(C0, x1, m);
        double dy = y_turn - x1[0];
        x1[0] = x1[0] - 2.0 * dy;
    }
    for (int i = 0; i < reflection; i++) {
        double C1 = x1[0] - get_y_with_z_mirror(x1[1], C0, m, 0.0);
        get_reflection_point(C0, C1, m, z_refl, x1);
    }
    double C1 = x1[0] - get_y_with_z_mirror(x1[1], C0, m, 0.0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double y_turn = get_y(gamma_turn, C0, C1, m);
    if (z_turn < x2[1]) {
        double dz = z_turn - x2[1], dy = y_turn - x2[0];
        double diff = sqrt(dz * dz + dy * dy) + 10 * fabs(dz);
        return -diff;
    }
    if (y_turn > x2[0]) {
        double y2_fit = get_y(get_gamma(x2[1], m), C0, C1, m);
        return x2[0] - y2_fit;
    } else {
        double y2_raw = get_y(get_gamma(x2[1], m), C0, C1, m);
        double y2_fit = 2 * y_turn - y2_raw;
        return -1 * (x2[0] - y2_fit);
    }
}

static double get_delta_y(double C0, const double x1[2], const double x2[2], const ice_t *m)
{
    return get_delta_y_refl(C0, x1, x2, m, 0, 1, 0.);
}

typedef struct { const double *x1, *x2; const ice_t *m; long nfev; int reflection, reflection_case; double z_refl; } obj_t;

static double obj_delta_y(double logC0, void *p) /* :1357 */
{
    obj_t *o = (obj_t *)p;
    o->nfev++;
    return get_delta_y_refl(C0_from_log(logC0, o->m), o->x1, o->x2, o->m, o->reflection, o->reflection_case, o->z_refl);
}
static double obj_delta_y_square(double logC0, void *p) /* :274 */
{
    double d = obj_delta_y(logC0, p);
    return d * d;
}

/* ------------------------------------------------------------------------------------------
 * MINPACK HYBRD restated for n = 1, as driven by scipy.optimize.root(method='hybr', tol=xtol):
 * maxfev = 200*(n+1), ml = mu = n-1, epsfcn = machine eps, factor = 100, mode = 1.
 * For n = 1:  QR of the 1x1 Jacobian a gives Q = -1, R = -a (Householder sign convention of QRFAC),
 * the dogleg step is the Newton step clipped to the trust region, R1UPDT is r += u*v and the
 * Givens sweeps of R1UPDT / R1MPYQ are empty.
 * ---------------------------------------------------------------------------------------- */
typedef double (*fun1_t)(double, void *);

int orc_hybrd1(fun1_t fcn, void *p, double *x_io, double *fvec_out, double xtol, int *nfev_out)
{
    const double epsmch = DBL_EPSILON;
    const double p1 = .1, p5 = .5, p001 = .001, p0001 = 1e-4, factor = 100.;
    const int maxfev = 400;
    double x = *x_io;
    int info = 0, nfev = 0;
    double fvec = fcn(x, p);
    nfev = 1;
    double fnorm = fabs(fvec);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double diag = 0, delta = 0, xnorm = 0;
    double r = 0, qtf = 0, fjac = 0;
    for (;;) { /* outer loop: (re)compute the Jacobian by forward difference (FDJAC1) */
        int jeval = 1;
        double eps = sqrt(epsmch); /* epsfcn = epsmch */
        double h = eps * fabs(x);
        if (h == 0.) h = eps;
        double wa1f = fcn(x + h, p);
        nfev += 1;
        double a = (wa1f - fvec) / h;
        /* QRFAC on 1x1: acnorm = |a|; rdiag = -a; householder vector element = 2 (if a != 0) */
        double acnorm = fabs(a);
        double hh = (a != 0.) ? 2. : 0.; /* fjac(1,1) after qrfac: a/ajnorm + 1 */
        double rdiag = -a;              /* -ajnorm with ajnorm carrying the sign of a */
        if (a == 0.) rdiag = -0.0;
        if (iter == 1) {
            diag = acnorm;
            if (acnorm == 0.) diag = 1.;
            xnorm = fabs(diag * x);
            delta = factor * xnorm;
            if (delta == 0.) delta = factor;
        }
        /* qtf = Q^T fvec */
        double wa4 = fvec;
        if (hh != 0.) {
            double sum = hh * wa4;
            double temp = -sum / hh;
            wa4 += hh * temp;
        }
        qtf = wa4;
        r = rdiag;
        /* QFORM: Q = I - v v^T / v1 -> 1 - 2*2/2 = -1 ; if a == 0 the column is zero -> Q = 1 */
        fjac = (hh != 0.) ? -1. : 1.;
        if (acnorm > diag) diag = acnorm; /* mode 1 rescale */
        for (;;) { /* inner loop */
            /* DOGLEG (n = 1) */
            double temp = r;
            if (temp == 0.) {
                temp = epsmch * fabs(r); /* epsmch * max column abs */
                if (temp == 0.) temp = epsmch;
            }
            double xs = qtf / temp; /* Gauss-Newton direction */
            double qnorm = fabs(diag * xs);
            if (qnorm > delta) {
                double g = r * qtf / diag; /* scaled gradient */
                double gnorm = fabs(g);
                double sgnorm = 0., alpha = delta / qnorm;
                double w1 = g;
                if (gnorm != 0.) {
                    w1 = (g / gnorm) / diag;
                    double w2 = r * w1;
                    double t = fabs(w2);
                    sgnorm = (gnorm / t) / t;
                    alpha = 0.;
                    if (sgnorm < delta) {
                        double bnorm = fabs(qtf);
                        double tt = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
                        double dq = delta / qnorm, sd = sgnorm / delta;
                        tt = tt - dq * (sd * sd) + sqrt((tt - dq) * (tt - dq) + (1. - dq * dq) * (1. - sd * sd));
                        alpha = (dq * (1. - sd * sd)) / tt;
                    }
                }
                double t2 = (1. - alpha) * fmin(sgnorm, delta);
                xs = t2 * w1 + alpha * xs;
            }
            double wa1 = -xs;         /* step p */
            double wa2 = x + wa1;     /* trial point */
            double wa3 = diag * wa1;
            double pnorm = fabs(wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
            double f_new = fcn(wa2, p);
            nfev += 1;
            double fnorm1 = fabs(f_new);
            double actred = -1.;
            if (fnorm1 < fnorm) actred = 1. - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            /* predicted reduction: |r*p + qtf| */
            double w3 = qtf + r * wa1;
            double tnorm = fabs(w3);
            double prered = 0.;
            if (tnorm < fnorm) prered = 1. - (tnorm / fnorm) * (tnorm / fnorm);
            double ratio = 0.;
            if (prered > 0.) ratio = actred / prered;
            if (ratio < p1) {
                ncsuc = 0;
                ncfail++;
                delta = p5 * delta;
            } else {
                ncfail = 0;
                ncsuc++;
                if (ratio >= p5 || ncsuc > 1) delta = fmax(delta, pnorm / p5);
                if (fabs(ratio - 1.) <= p1) delta = pnorm / p5;
            }
            if (ratio >= p0001) { /* successful iteration */
                x = wa2;
                fvec = f_new;
                xnorm = fabs(diag * x);
                fnorm = fnorm1;
                iter++;
            }
            nslow1++;
            if (actred >= p001) nslow1 = 0;
            if (jeval) nslow2++;
            if (actred >= p1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0.) info = 1;
            if (info != 0) goto done;
            if (nfev >= maxfev) info = 2;
            if (p1 * fmax(p1 * delta, pnorm) <= epsmch * xnorm) info = 3;
            if (nslow2 == 5) info = 4;
            if (nslow1 == 10) info = 5;
            if (info != 0) goto done;
            if (ncfail == 2) break; /* recompute Jacobian */
            /* rank-one (Broyden) update of R, and of qtf */
            double sum = fjac * f_new;
            double v = (sum - w3) / pnorm;
            double u = diag * ((diag * wa1) / pnorm);
            if (ratio >= p0001) qtf = sum;
            r = r + u * v; /* R1UPDT, n = 1 */
            jeval = 0;
        }
    }
done:
    *x_io = x;
    *fvec_out = fvec;
    if (nfev_out) *nfev_out = nfev;
    return info;
}

/* scipy/optimize/Zeros/brentq.c restated; xtol = 2e-12, rtol = 4*eps, maxiter = 100 (scipy defaults) */
int orc_brentq(fun1_t f, void *p, double xa, double xb, double *root)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    const int maxiter = 100;
    double xpre = xa, xcur = xb, xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis;
    double delta, stry, dpre, dblk;
    fpre = f(xpre, p);
    fcur = f(xcur, p);
    if (fpre == 0) { *root = xpre; return 0; }
    if (fcur == 0) { *root = xcur; return 0; }
    if (signbit(fpre) == signbit(fcur)) { *root = NAN; return -1; }
    for (int i = 0; i < maxiter; i++) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) { *root = xcur; return 0; }
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
                spre = scur; scur = stry;
            } else {
                spre = sbis; scur = sbis;
            }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur, p);
    }
    *root = xcur;
    return -2;
}

/* np.sign semantics incl. NaN (np.sign(nan) = nan, and nan != anything is True) */
static int sign_differs(double a, double b)
{
    if (isnan(a) || isnan(b)) return 1;
    double sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
    return sa != sb;
}

/* :1365-1398 */
static int determine_solution_type(const double x1[2], const double x2[2], double C0, const ice_t *m)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double C1 = get_C1(x1, C0, m);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double y_turn = get_y(gamma_turn, C0, C1, m);
    if (x2[0] < y_turn) return 1;     /* direct */
    if (z_turn == 0) return 3;        /* reflected */
    return 2;                         /* refracted */
}

/* ------------------------------------------------------------------------------------------
 * The solution finder without the hybr stage (round 5; DESIGN section 2, "the true solution set").
 *
 * delta_y(log C0) of :204-272 is min(u, v) wherever the ray's turning point lies above the receiver:
 *     u = x2.y - y(z2)              the receiver's offset from the ray on its way UP to the turning point,
 *     v = (2 y_turn - y(z2)) - x2.y the same on its way DOWN (mirrored branch, or after the reflection at the surface);
 * u rises monotonically with C0 (a steeper launch reaches the receiver's depth earlier), v rises to one maximum -- the
 * farthest point any ray reaches at that depth -- and falls; both verified on 1e6 random pairs in three ice models
 * (tools/root_shapes.py).  So the solutions are: the root of u (the direct ray) and the root of v beyond its maximum
 * when the ray that turns AT the receiver's depth overshoots the receiver (v > 0 there), else the two roots of v either
 * side of its maximum if that is positive, else none.  The reference looks for the same roots with scipy.optimize.root
 * on (delta_y)^2 from log C0 = -1 -- about 37 evaluations creeping onto a double root, stopped 1e-7 away from it and
 * kept by a coin flip (:1479-1483) -- and two Brent searches either side of where that stopped; here every root comes
 * out of a bracket, to Brent's 2e-12, in a third of the evaluations, and none is lost.
 *
 * Searches run in t = sqrt(log C0 - x_lo), x_lo = log(1 / n(z2) - 1 / n_ice) the launch parameter of the ray that turns at
 * the receiver's depth: u and v start like sqrt(log C0 - x_lo) there.  Used where the exponential profile is resolved at
 * the receiver (z2 >= -10 z_0: every detector); deeper receivers -- n(z2) = n_ice to 1e-5, x_lo ill-conditioned -- keep
 * the reference's procedure below.
 * ---------------------------------------------------------------------------------------- */
#define ORC_T_START 3e-5       /* t of the lower end of every search: log C0 = x_lo + 9e-10 */
#define ORC_SHALLOW 4.5399929762484854e-05   /* exp(-10) */
typedef struct { const double *x1, *x2; const ice_t *m; long nfev; double x_lo, g1, g2; } uv_t;

static void uv_at(double t, uv_t *o, double *u, double *v)
{
    const ice_t *m = o->m;
    o->nfev++;
    double C0 = C0_from_log(o->x_lo + t * t, m);
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double b = 2 * m->n_ice;
    double gamma_turn = b * 0.5 - sqrt(0.25 * b * b - c);
    if (gamma_turn > m->delta_n) gamma_turn = m->delta_n;   /* turning point above the surface: reflection at z = 0 (:147-151) */
    double y_turn0 = get_y(gamma_turn, C0, 0.0, m);
    double C1 = o->x1[0] - get_y(o->g1, C0, 0.0, m);         /* the start point lies below the turning point */
    double y_turn = y_turn0 + C1;
    double y2 = get_y(o->g2, C0, 0.0, m) + C1;
    *u = o->x2[0] - y2;
    *v = -1 * (o->x2[0] - (2 * y_turn - y2));
}

/* scipy's brentq (xtol 2e-12, rtol 4 eps) on one component of (u, v) as a function of t; end values given */
static double brent_uv(uv_t *o, int comp, double xa, double xb, double fa, double fb)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0., fpre = fa, fcur = fb, fblk = 0., spre = 0., scur = 0.;
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    for (int i = 0; i < 100; i++) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2;
        double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur);
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        double u, v;
        uv_at(xcur, o, &u, &v);
        fcur = comp ? v : u;
    }
    return xcur;
}

/* roots as log C0 (unsorted); *kind: 1 the ray turning at the receiver's depth overshoots it (direct + one more), 2 it falls
 * short and so does every other ray (none), 3 it falls short, others do not (two roots either side of the farthest ray) */
static int find_solutions_bracketed(const double x1[2], const double x2[2], const ice_t *m, double *logC0, long *nfev, int *kind)
{
    uv_t o = { x1, x2, m, 0, 0., get_gamma(x1[1], m), get_gamma(x2[1], m) };
    int n = 0;
    o.x_lo = orc_log(1. / (m->n_ice - o.g2) - 1. / m->n_ice);
    const double ta = ORC_T_START, tm = sqrt(2. - o.x_lo), tt = sqrt(100. - o.x_lo);   /* log C0 = x_lo + 9e-10, 2, 100 */
    double r[2];
    double ua, va, um, vm, ut = 0., vt = 0.;
    uv_at(ta, &o, &ua, &va);
    uv_at(tm, &o, &um, &vm);
    if (va > 0) {
        *kind = 1;
        int have_t = 0;
        if (ua < 0) {   /* the direct ray: u rises through zero once */
            if (um > 0) r[n++] = brent_uv(&o, 0, ta, tm, ua, um);
            else {
                uv_at(tt, &o, &ut, &vt);
                have_t = 1;
                if (ut > 0) r[n++] = brent_uv(&o, 0, tm, tt, um, ut);
            }
        }
        if (vm < 0) r[n++] = brent_uv(&o, 1, ta, tm, va, vm);   /* v falls through zero once beyond its maximum */
        else {
            if (!have_t) uv_at(tt, &o, &ut, &vt);
            if (vt < 0) r[n++] = brent_uv(&o, 1, tm, tt, vm, vt);
        }
    } else {
        /* v <= 0 at the lower end: is its maximum positive?  Brent's minimiser (golden section + parabolic steps) on -v over
         * (a, b); a and b are always evaluated points (fa, fb = v there, <= 0 so far); it stops at the first v > 0 */
        *kind = 2;
        const double CG = 0.3819660112501051;
        double a = ta, fa = va, b = tm, fb = vm;
        double x, fx;
        if (vm > va) {   /* still rising at log C0 = 2: the maximum may lie beyond */
            uv_at(tt, &o, &ut, &vt);
            b = tt; fb = vt;
            x = tm; fx = -vm;
        } else {
            double uu, q;
            x = a + CG * (b - a);
            uv_at(x, &o, &uu, &q);
            fx = -q;
        }
        double w = x, vv = x, fw = fx, fv = fx, d = 0., e = 0.;
        int found = (fx < 0);
        for (int it = 0; it < 60 && !found; it++) {
            double xm = 0.5 * (a + b), tol1 = 1e-6 * fabs(x) + 1e-7, tol2 = 2. * tol1;
            if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
            int golden = 1;
            if (fabs(e) > tol1) {
                double rr = (x - w) * (fx - fv), q = (x - vv) * (fx - fw), p = (x - vv) * q - (x - w) * rr;
                q = 2. * (q - rr);
                if (q > 0.) p = -p;
                q = fabs(q);
                double etemp = e;
                e = d;
                if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
                    d = p / q;
                    double xn = x + d;
                    if (xn - a < tol2 || b - xn < tol2) d = (xm - x >= 0) ? tol1 : -tol1;
                    golden = 0;
                }
            }
            if (golden) {
                e = (x >= xm) ? a - x : b - x;
                d = CG * e;
            }
            double xu = (fabs(d) >= tol1) ? x + d : x + ((d >= 0) ? tol1 : -tol1);
            double uu, q;
            uv_at(xu, &o, &uu, &q);
            double fu = -q;
            if (fu < 0) {   /* v > 0: inside the interval of solutions, a < xu < b */
                if (xu < x) { b = x; fb = -fx; } else { a = x; fa = -fx; }
                x = xu; fx = fu;
                found = 1;
                break;
            }
            if (fu <= fx) {
                if (xu >= x) { a = x; fa = -fx; } else { b = x; fb = -fx; }
                vv = w; fv = fw; w = x; fw = fx; x = xu; fx = fu;
            } else {
                if (xu < x) { a = xu; fa = -fu; } else { b = xu; fb = -fu; }
                if (fu <= fw || w == x) { vv = w; fv = fw; w = xu; fw = fu; }
                else if (fu <= fv || vv == x || vv == w) { vv = xu; fv = fu; }
            }
        }
        if (found) {
            *kind = 3;
            r[n++] = brent_uv(&o, 1, a, x, fa, -fx);   /* (an end with v == 0 exactly is returned at once) */
            r[n++] = brent_uv(&o, 1, x, b, -fx, fb);
        }
    }
    for (int i = 0; i < n; i++) logC0[i] = o.x_lo + r[i] * r[i];
    if (nfev) *nfev = o.nfev;
    return n;
}

/* ray_tracing_2D.find_solutions, Python branch (:1433-1547), receiver in ice; `reflection` bottom reflections with the
 * ray starting upwards (reflection_case 1) or downwards (2).  Returns number of solutions (<= 3) sorted by C0; hybr
 * diagnostics optional. */
/* The finder of the checker follows the product's (include/nrhip.h, nrhip_ctx_set_ray_finder):
 *   orc_reference_procedure == 0  the true solution set: the bracketed finder; where that does not apply (deep receivers, end
 *                                 points above each other, calls with a reflective layer) hybr + two Brent searches with the first
 *                                 root also accepted by a sign change either side of the hybr iterate (plain call only);
 *   orc_reference_procedure != 0  the reference to the letter (:1476-1547): hybr, its acceptance test (delta y)^2 < 1e-7 alone,
 *                                 two Brent searches -- for every pair and every call.                                            */
int orc_reference_procedure = 0;
static int orc_force_procedure = 0;   /* hybr + Brent also where the bracketed finder would apply (the calls of a reflective layer) */
void orc_set_reference_procedure(int on) { orc_reference_procedure = on; }

int orc_find_solutions_2d_refl(const double x1[2], const double x2[2], const double ice[3], int reflection,
                               int reflection_case, double z_refl, double *C0s, double *C1s, int *types, double *hybr_x,
                               double *hybr_fun, int *nfev)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    obj_t o = { x1, x2, &m, 0, reflection, reflection_case, z_refl };
    int n = 0;
    double logC0[3];
    if (x2[1] > 0) return 0; /* ice->air special case (:1437-1460) not restated */
    /* (end points exactly above each other: the solutions are the vertical rays, log C0 -> infinity -- the reference's procedure
     * reports them at its search limits; the brackets of the finder above have no sign change to find) */
    if (reflection == 0 && get_gamma(x2[1], &m) >= ORC_SHALLOW * m.delta_n && x2[0] > x1[0] && !orc_reference_procedure && !orc_force_procedure) {
        int kind;
        n = find_solutions_bracketed(x1, x2, &m, logC0, &o.nfev, &kind);
        if (hybr_x) *hybr_x = NAN;
        if (hybr_fun) *hybr_fun = NAN;
        goto have_roots;
    }
    double xr = -1.;
    double fun;
    int nf;
    orc_hybrd1(obj_delta_y_square, &o, &xr, &fun, 1e-6, &nf);
    if (hybr_x) *hybr_x = xr;
    if (hybr_fun) *hybr_fun = fun;
    const double d_hi = obj_delta_y(xr + 0.0001, &o), d_lo = obj_delta_y(xr - 0.0001, &o);
    if (fun < 1e-7) logC0[n++] = xr;
    else if (!orc_reference_procedure && reflection == 0 && d_lo != 0 && d_hi != 0 && !isnan(d_lo) && !isnan(d_hi) && signbit(d_lo) != signbit(d_hi)) {
        /* THE TRUE SOLUTION SET (round 5; DESIGN section 2, tools/true_roots.py).  The reference keeps its first root only
         * if (delta_y)^2 < 1e-7 where hybr stopped (:1483) -- about 1e-7 off a double root, where that number is 2e-8 ... 3e-6:
         * a coin flip on the last bits of exp / log, and the two Brent searches leave the 2e-4 around the iterate out.
         * delta_y is continuous in log C0 (it is a - |s|: a the horizontal half-width of the ray's arc at the receiver's
         * depth, s the receiver's offset from the turning point; the "turning point below the receiver" branch :247-253
         * joins it continuously where a = 0), so opposite signs either side of the iterate ARE a root between them, and it
         * is taken from there -- never a false root, and every root the reference can report is reported. */
        double rt;
        if (orc_brentq(obj_delta_y, &o, xr - 0.0001, xr + 0.0001, &rt) != -1) logC0[n++] = rt;
    }
    {
        double a = xr + 0.0001, b = 100.;
        double da = d_hi, db = obj_delta_y(b, &o);
        if (sign_differs(da, db)) {
            double rt;
            if (orc_brentq(obj_delta_y, &o, a, b, &rt) != -1) logC0[n++] = rt; /* -1: scipy raises ValueError */
        }
    }
    {
        double a = -100., b = xr - 0.0001;
        double da = obj_delta_y(a, &o), db = d_lo;
        if (sign_differs(da, db)) {
            double rt;
            if (orc_brentq(obj_delta_y, &o, a, b, &rt) != -1) logC0[n++] = rt; /* -1: scipy raises ValueError */
        }
    }
have_roots:
    for (int i = 0; i < n; i++) {
        C0s[i] = C0_from_log(logC0[i], &m);
        types[i] = determine_solution_type(x1, x2, C0s[i], &m);
        C1s[i] = get_C1(x1, C0s[i], &m);
    }
    /* sorted(results, key=('reflection','C0')) -- stable insertion sort */
    for (int i = 1; i < n; i++)
        for (int j = i; j > 0 && C0s[j] < C0s[j - 1]; j--) {
            double t = C0s[j]; C0s[j] = C0s[j - 1]; C0s[j - 1] = t;
            t = C1s[j]; C1s[j] = C1s[j - 1]; C1s[j - 1] = t;
            int k = types[j]; types[j] = types[j - 1]; types[j - 1] = k;
        }
    if (nfev) *nfev = (int)o.nfev;
    return n;
}

int orc_find_solutions_2d(const double x1[2], const double x2[2], const double ice[3],
                          double *C0s, double *C1s, int *types, double *hybr_x, double *hybr_fun, int *nfev)
{
    return orc_find_solutions_2d_refl(x1, x2, ice, 0, 1, 0., C0s, C1s, types, hybr_x, hybr_fun, nfev);
}

/* :496-511 */
static double get_z_mirrored(const double x1[2], const double x2[2], double C0, const ice_t *m)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double C1 = get_C1(x1, C0, m);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double y_turn = get_y(gamma_turn, C0, C1, m);
    double zstop = x2[1];
    if (y_turn < x2[0]) zstop = x1[1] + fabs(z_turn - x1[1]) + fabs(z_turn - x2[1]);
    return zstop;
}

/* :293-304 */
static double get_z_unmirrored(double z, double C0, const ice_t *m)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    if (z > z_turn) return 2 * z_turn - z;
    return z;
}

/* :306-355 (in_air = False) */
static double get_y_diff(double z_raw, double C0, const ice_t *m)
{
    double z = get_z_unmirrored(z_raw, C0, m);
    double n_z = n_of_z(z, m);
    double res;
    if ((C0 * C0) * (n_z * n_z) > 1) res = 1 / sqrt((C0 * C0) * (n_z * n_z) - 1); /* C_0**2 * n_z**2 */
    else res = INFINITY;
    if (z != z_raw) res *= -1;
    return res;
}

/* :1161-1199: (sin, cos) of the angle to the +z axis of the ray at x, taken algebraically from dy/dz */
static void get_angle_sincos(const double x[2], const double x_start[2], double C0, const ice_t *m, double *sn, double *cs)
{
    double z = get_z_mirrored(x_start, x, C0, m);
    double dy = get_y_diff(z, C0, m); /* signed: negative on the mirrored branch */
    if (isinf(dy)) {
        *sn = 1.;
        *cs = 0.;
        return;
    }
    double a = fabs(dy);
    double h = sqrt(1 + a * a);
    *sn = a / h;                     /* angle = arctan(dy), + pi if negative: sin >= 0 */
    *cs = (dy < 0 ? -1. : 1.) / h;
}

/* :1201-1237, reflection = 0: returns NaN for "None" */
static double get_reflection_angle(const double x1[2], const double x2[2], double C0, const ice_t *m)
{
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    double C1 = get_C1(x1, C0, m);
    double y_turn = get_y(gamma_turn, C0, C1, m); /* get_y_turn :186 */
    if (z_turn >= 0 && y_turn > x1[0] && y_turn < x2[0]) {
        double xs[2] = { y_turn, 0. }, sn, cs;
        get_angle_sincos(xs, x1, C0, m, &sn, &cs);
        return atan2(sn, cs);
    }
    return NAN;
}

/* :602-690 and :692-783, reflection = 0, receiver in ice */
static void path_length_and_time(const double x1[2], const double x2[2], double C0, const ice_t *m,
                                 double *D, double *T)
{
    double z1 = x1[1], z2 = x2[1];
    int solution_type = determine_solution_type(x1, x2, C0, m);
    double sin_launch, cos_launch;
    get_angle_sincos(x1, x1, C0, m, &sin_launch, &cos_launch);
    double n_ice = m->n_ice, z_0 = m->z_0;
    double n1 = n_of_z(z1, m);
    double beta = n1 * sin_launch;
    double alpha = n_ice * n_ice - beta * beta;
    double zz[3] = { z1, z2, 0. };
    double s[3], ct[3];
    if (solution_type == 2) {
        double g, zt;
        get_turning_point(n_ice * n_ice - 1. / (C0 * C0), m, &g, &zt);
        zz[2] = zt;
    }
    for (int i = 0; i < 3; i++) {
        double nz = n_of_z(zz[i], m);
        double gamma = fmax(0., nz * nz - beta * beta);
        double l1 = sqrt(alpha * gamma) + n_ice * nz - beta * beta;
        double l2 = sqrt(gamma) + nz;
        double ll1 = orc_log(l1), ll2 = orc_log(l2), sa = sqrt(alpha);
        s[i] = n_ice / sa * (zz[i] - z_0 * ll1) + z_0 * ll2;
        ct[i] = z_0 * (sqrt(gamma) - n_ice * n_ice / sa * ll1 + n_ice * ll2) + n_ice * n_ice * zz[i] / sa;
    }
    if (solution_type == 1) {
        *D = s[1] - s[0];
        *T = (ct[1] - ct[0]) / SPEED_OF_LIGHT;
    } else {
        *D = 2 * s[2] - s[0] - s[1];
        *T = (2 * ct[2] - ct[0] - ct[1]) / SPEED_OF_LIGHT;
    }
}

/* get_path_segments (:1091-1159): one segment per stretch between two bottom reflections.
 * Segment = (start, stop, C1); first_start = the start point as given (x1_orig of the reference's segment tuple). */
#define ORC_MAX_REFL 4
typedef struct { double x1[2], x2[2], C1; } seg_t;

static int path_segments(const double x1_in[2], const double x2_in[2], double C0, int reflection, int reflection_case,
                         double z_refl, const ice_t *m, seg_t *segs)
{
    double x1[2] = { x1_in[0], x1_in[1] };
    if (reflection == 0) {
        segs[0].x1[0] = x1[0]; segs[0].x1[1] = x1[1];
        segs[0].x2[0] = x2_in[0]; segs[0].x2[1] = x2_in[1];
        segs[0].C1 = get_C1(x1, C0, m);
        return 1;
    }
    if (reflection_case == 2) {
        double y_turn = get_y_turn(C0, x1, m);
        double dy = y_turn - x1[0];
        x1[0] = x1[0] - 2 * dy;
    }
    int n = 0;
    for (int i = 0; i < reflection + 1; i++) {
        double C1 = get_C1(x1, C0, m);
        double x2[2];
        get_reflection_point(C0, C1, m, z_refl, x2);
        int stop = 0;
        if (x2[0] > x2_in[0]) {
            stop = 1;
            x2[0] = x2_in[0]; x2[1] = x2_in[1];
        }
        segs[n].x1[0] = x1[0]; segs[n].x1[1] = x1[1];
        segs[n].x2[0] = x2[0]; segs[n].x2[1] = x2[1];
        segs[n].C1 = C1;
        n++;
        if (stop) break;
        x1[0] = x2[0]; x1[1] = x2[1];
    }
    return n;
}

/* the end points the reference integrates a segment between: a first segment that starts downwards is mirrored
 * (:629-636, :720-727, :943-950): from (y of the original start, z of the segment's end) up to (y of the end, z of the start) */
static void segment_end_points(const seg_t *sg, int iS, int reflection_case, const double x1_orig[2], double a[2], double b[2])
{
    if (iS == 0 && reflection_case == 2) {
        a[0] = x1_orig[0]; a[1] = sg->x2[1];
        b[0] = sg->x2[0];  b[1] = x1_orig[1];
    } else {
        a[0] = sg->x1[0]; a[1] = sg->x1[1];
        b[0] = sg->x2[0]; b[1] = sg->x2[1];
    }
}

/* get_path_length_analytic / get_travel_time_analytic with bottom reflections: the sum over the segments */
static void path_length_and_time_refl(const double x1[2], const double x2[2], double C0, int reflection, int reflection_case,
                                      double z_refl, const ice_t *m, double *D, double *T)
{
    seg_t segs[ORC_MAX_REFL + 1];
    int n = path_segments(x1, x2, C0, reflection, reflection_case, z_refl, m, segs);
    double d = 0, t = 0;
    for (int i = 0; i < n; i++) {
        double a[2], b[2], di, ti;
        segment_end_points(&segs[i], i, reflection_case, x1, a, b);
        path_length_and_time(a, b, C0, m, &di, &ti);
        d += di;
        t += ti * SPEED_OF_LIGHT;   /* the reference sums c t and divides once (:783) */
    }
    *D = d;
    *T = t / SPEED_OF_LIGHT;
}

/* get_reflection_angle (:1201-1237) per segment: NaN where the segment has no reflection at the surface; returns
 * the number of segments */
static int reflection_angles(const double x1[2], const double x2[2], double C0, int reflection, int reflection_case,
                             double z_refl, const ice_t *m, double *out)
{
    seg_t segs[ORC_MAX_REFL + 1];
    int n = path_segments(x1, x2, C0, reflection, reflection_case, z_refl, m, segs);
    double c = m->n_ice * m->n_ice - 1. / (C0 * C0);
    double gamma_turn, z_turn;
    get_turning_point(c, m, &gamma_turn, &z_turn);
    for (int i = 0; i < n; i++) {
        double y_turn = get_y_turn(C0, segs[i].x1, m);
        out[i] = NAN;
        if (z_turn >= 0 && y_turn > x1[0] && y_turn < x2[0]) {
            double xs[2] = { y_turn, 0. }, sn, cs;
            get_angle_sincos(xs, segs[i].x1, C0, m, &sn, &cs);
            out[i] = atan2(sn, cs);
        }
    }
    return n;
}

/* get_angle (:1161-1193) with bottom reflections: the start of the LAST segment of the path x_start -> x takes the
 * place of x_start */
static void get_angle_sincos_refl(const double x[2], const double x_start[2], double C0, int reflection, int reflection_case,
                                  double z_refl, const ice_t *m, double *sn, double *cs)
{
    seg_t segs[ORC_MAX_REFL + 1];
    int n = path_segments(x_start, x, C0, reflection, reflection_case, z_refl, m, segs);
    get_angle_sincos(x, segs[n - 1].x1, C0, m, sn, cs);
}

/* ------------------------------------------------------------------------------------------
 * attenuation length (NuRadioMC/utilities/attenuation.py:145-262), scalar branch
 * model ints follow attenuation.py:14  {"SP1": 1, "GL1": 2, "MB1": 3, "GL2": 4, "GL3": 5}
 * ---------------------------------------------------------------------------------------- */
double orc_attenuation_length(double z, double frequency, int model);

/* GL3 depth table (NuRadioMC/utilities/data/GL3_params.csv: depth [m, positive], slope, offset), linear interpolation like
 * scipy.interpolate.interp1d(bounds_error=False, fill_value=(first, last)) (attenuation.py:16-34) */
static int gl3_n = 0;
static double gl3_d[1024], gl3_s[1024], gl3_o[1024];
void orc_set_gl3_table(int n, const double *depth, const double *slope, const double *offset)
{
    gl3_n = n > 1024 ? 1024 : n;
    for (int i = 0; i < gl3_n; i++) { gl3_d[i] = depth[i]; gl3_s[i] = slope[i]; gl3_o[i] = offset[i]; }
}
static double gl3_interp(double x, const double *fp)
{
    if (x < gl3_d[0]) return fp[0];
    if (x > gl3_d[gl3_n - 1]) return fp[gl3_n - 1];
    int lo = 0, hi = gl3_n; /* searchsorted(side='left'): first index with d[i] >= x */
    while (lo < hi) { int mid = (lo + hi) / 2; if (gl3_d[mid] < x) lo = mid + 1; else hi = mid; }
    int idx = lo < 1 ? 1 : (lo > gl3_n - 1 ? gl3_n - 1 : lo);
    double slope = (fp[idx] - fp[idx - 1]) / (gl3_d[idx] - gl3_d[idx - 1]);
    return slope * (x - gl3_d[idx - 1]) + fp[idx - 1];
}

/* SP1 :168-192: the attenuation length is 1 / exp(a + b ln f); this returns the exponent a + b ln f */
static double sp1_exponent(double z, double frequency)
{
    double z2 = fabs(z);
    double t = 1.83415e-09 * (z2 * z2 * z2) + (-1.59061e-08 * (z2 * z2)) + 0.00267687 * z2 + (-51.0696);
    double w0 = -9.210340371976182, w1 = 0.0, w2 = 1.1505720275988207; /* ln 1e-4, ln 3.16 */
    double w = orc_log(frequency);
    double b0 = -6.74890 + t * (0.026709 - t * 0.000884);
    double b1 = -6.22121 - t * (0.070927 + t * 0.001773);
    double b2 = -4.09468 - t * (0.002213 + t * 0.000332);
    double a, bb;
    if (frequency < 1.) {
        a = (b1 * w0 - b0 * w1) / (w0 - w1);
        bb = (b1 - b0) / (w1 - w0);
    } else {
        a = (b2 * w1 - b1 * w2) / (w1 - w2);
        bb = (b2 - b1) / (w2 - w1);
    }
    return a + bb * w;
}

/* ds / L(z, f) of the path integrand.  For SP1, L = max(1 / exp(x), 1) (and infinity above the surface), so
 * ds / L = ds * min(exp(x), 1): the same number without the two divisions (the device kernel does the same). */
static double ds_over_length(double ds, double z, double frequency, int model)
{
    if (model == 1) {
        double e = orc_exp_tab(sp1_exponent(z, frequency));   /* (the integrand's own exp, detmath_c.h) */
        if (e > 1.) e = 1.;
        if (z > 0) e = 0.;
        return ds * e;
    }
    return ds / orc_attenuation_length(z, frequency, model);
}

double orc_attenuation_length(double z, double frequency, int model)
{
    double L;
    if (model == 1) { /* SP1 :168-192 */
        double z2 = fabs(z);
        double t = 1.83415e-09 * (z2 * z2 * z2) + (-1.59061e-08 * (z2 * z2)) + 0.00267687 * z2 + (-51.0696);
        double f0 = 0.0001, f2 = 3.16;
        double w0 = -9.210340371976182, w1 = 0.0, w2 = 1.1505720275988207; /* ln 1e-4, ln 3.16 */
        double w = orc_log(frequency);
        (void)f0; (void)f2;
        double b0 = -6.74890 + t * (0.026709 - t * 0.000884);
        double b1 = -6.22121 - t * (0.070927 + t * 0.001773);
        double b2 = -4.09468 - t * (0.002213 + t * 0.000332);
        double a, bb;
        if (frequency < 1.) {
            a = (b1 * w0 - b0 * w1) / (w0 - w1);
            bb = (b1 - b0) / (w1 - w0);
        } else {
            a = (b2 * w1 - b1 * w2) / (w1 - w2);
            bb = (b2 - b1) / (w2 - w1);
        }
        L = 1. / orc_exp(a + bb * w);
    } else if (model == 2) { /* GL1 :99-128, :194-196 (Python clamps the 75 MHz length at 100 m) */
        static const double fit[6] = { 1.16052586e+03, 6.87257150e-02, -9.82378264e-05,
                                       -3.50628312e-07, -2.21040482e-10, -3.63912864e-14 };
        double att = 0, zp = 1;
        for (int p = 0; p < 6; p++) { att += fit[p] * zp; zp *= z; }
        if (att < 100.) att = 100.;
        L = att - 0.55 * (frequency / 1e-3 - 75);
    } else if (model == 4) { /* GL2 :198-204 */
        static const double fit[6] = { 1.20547286e+00, 1.58815679e-05, -2.58901767e-07,
                                       -5.16435542e-10, -2.89124473e-13, -4.58987344e-17 };
        double bulk = 852.0 + (-0.54 / 1e-3) * frequency;
        double poly = 0; /* np.poly1d(flip(fit))(z): Horner from highest power */
        for (int p = 5; p >= 0; p--) poly = poly * z + fit[p];
        L = bulk * poly;
    } else if (model == 3) { /* MB1 :224-244 */
        double R = 0.82, d_ice = 576.;
        L = 460. - 180. * frequency;
        L *= 1. / (1 + L / (2 * d_ice) * orc_log(R));
        double d = -z * 420. / d_ice;
        double LL = (1250. * 0.08886 * orc_exp(-0.048827 * (225.6746 - 86.517596 * (orc_log(848.870 - (d)) / 2.302585092994046))));
        L *= LL / 231.21;
    } else if (model == 5) { /* GL3 :206-221: L = slope(depth) f + offset(depth), depth table set by orc_set_gl3_table */
        if (gl3_n < 2) return NAN;
        L = gl3_interp(-z, gl3_s) * frequency + gl3_interp(-z, gl3_o);
    } else {
        return NAN;
    }
    if (L < 1.) L = 1.;
    if (z > 0) L = INFINITY;
    return L;
}

/* ------------------------------------------------------------------------------------------
 * QUADPACK restatement (DQK21, DQPSRT, DQELG, DQAGSE, DQAGPE)
 * ---------------------------------------------------------------------------------------- */
typedef double (*integrand_t)(double, void *);

static const double XGK[11] = {
    0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
    0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
    0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
    0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
    0.294392862701460198131126603103866, 0.148874338981631210884826001129720,
    0.000000000000000000000000000000000 };
static const double WGK[11] = {
    0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
    0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
    0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
    0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
    0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
    0.149445554002916905664936468389821 };
static const double WG[5] = {
    0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
    0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
    0.295524224714752870173815619188769 };

static void dqk21(integrand_t f, void *p, double a, double b, double *result, double *abserr,
                  double *resabs, double *resasc)
{
    const double epmach = DBL_EPSILON, uflow = DBL_MIN;
    double fv1[10], fv2[10];
    double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double resg = 0.;
    double fc = f(centr, p);
    double resk = WGK[10] * fc;
    *resabs = fabs(resk);
    for (int j = 0; j < 5; j++) {
        int jtw = 2 * j + 1;
        double absc = hlgth * XGK[jtw];
        double fval1 = f(centr - absc, p), fval2 = f(centr + absc, p);
        fv1[jtw] = fval1; fv2[jtw] = fval2;
        double fsum = fval1 + fval2;
        resg += WG[j] * fsum;
        resk += WGK[jtw] * fsum;
        *resabs += WGK[jtw] * (fabs(fval1) + fabs(fval2));
    }
    for (int j = 0; j < 5; j++) {
        int jtwm1 = 2 * j;
        double absc = hlgth * XGK[jtwm1];
        double fval1 = f(centr - absc, p), fval2 = f(centr + absc, p);
        fv1[jtwm1] = fval1; fv2[jtwm1] = fval2;
        double fsum = fval1 + fval2;
        resk += WGK[jtwm1] * fsum;
        *resabs += WGK[jtwm1] * (fabs(fval1) + fabs(fval2));
    }
    double reskh = resk * 0.5;
    *resasc = WGK[10] * fabs(fc - reskh);
    for (int j = 0; j < 10; j++) *resasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    *result = resk * hlgth;
    *resabs *= dhlgth;
    *resasc *= dhlgth;
    *abserr = fabs((resk - resg) * hlgth);
    if (*resasc != 0. && *abserr != 0.) {
        double r = 200. * *abserr / *resasc;
        *abserr = *resasc * fmin(1., r * sqrt(r)); /* r ** 1.5 */
    }
    if (*resabs > uflow / (50. * epmach)) *abserr = fmax((epmach * 50.) * *resabs, *abserr);
}

/* 1-based index arithmetic kept as in the Fortran; arrays are [0..limit] with slot 0 unused */
static void dqpsrt(int limit, int last, int *maxerr, double *ermax, const double *elist, int *iord, int *nrmax)
{
    double errmax, errmin;
    int i, ibeg, ido, isucc, j, jbnd, jupbn, k;
    if (last <= 2) {
        iord[1] = 1;
        iord[2] = 2;
        goto L90;
    }
    errmax = elist[*maxerr];
    if (*nrmax != 1) {
        ido = *nrmax - 1;
        for (i = 1; i <= ido; i++) {
            isucc = iord[*nrmax - 1];
            if (errmax <= elist[isucc]) break;
            iord[*nrmax] = isucc;
            (*nrmax)--;
        }
    }
    jupbn = last;
    if (last > (limit / 2 + 2)) jupbn = limit + 3 - last;
    errmin = elist[last];
    jbnd = jupbn - 1;
    ibeg = *nrmax + 1;
    if (ibeg <= jbnd) {
        for (i = ibeg; i <= jbnd; i++) {
            isucc = iord[i];
            if (errmax >= elist[isucc]) goto L60;
            iord[i - 1] = isucc;
        }
    }
    iord[jbnd] = *maxerr;
    iord[jupbn] = last;
    goto L90;
L60:
    iord[i - 1] = *maxerr;
    k = jbnd;
    for (j = i; j <= jbnd; j++) {
        isucc = iord[k];
        if (errmin < elist[isucc]) {
            iord[k + 1] = last;
            goto L90;
        }
        iord[k + 1] = isucc;
        k--;
    }
    iord[i] = last;
L90:
    *maxerr = iord[*nrmax];
    *ermax = elist[*maxerr];
}

/* epstab is 1-based [1..52], res3la 1-based [1..3] */
static void dqelg(int *n, double *epstab, double *result, double *abserr, double *res3la, int *nres)
{
    const double epmach = DBL_EPSILON, oflow = DBL_MAX;
    int i, ib, ib2, ie, indx, k1, k2, k3, limexp, newelm, num;
    double delta1, delta2, delta3, epsinf, error, err1, err2, err3, e0, e1, e1abs, e2, e3, res, ss, tol1, tol2, tol3;
    (*nres)++;
    *abserr = oflow;
    *result = epstab[*n];
    if (*n < 3) goto L100;
    limexp = 50;
    epstab[*n + 2] = epstab[*n];
    newelm = (*n - 1) / 2;
    epstab[*n] = oflow;
    num = *n;
    k1 = *n;
    for (i = 1; i <= newelm; i++) {
        k2 = k1 - 1;
        k3 = k1 - 2;
        res = epstab[k1 + 2];
        e0 = epstab[k3];
        e1 = epstab[k2];
        e2 = res;
        e1abs = fabs(e1);
        delta2 = e2 - e1;
        err2 = fabs(delta2);
        tol2 = fmax(fabs(e2), e1abs) * epmach;
        delta3 = e1 - e0;
        err3 = fabs(delta3);
        tol3 = fmax(e1abs, fabs(e0)) * epmach;
        if (!(err2 > tol2 || err3 > tol3)) {
            *result = res;
            *abserr = err2 + err3;
            *abserr = fmax(*abserr, 5. * epmach * fabs(*result));
            goto L100;
        }
        e3 = epstab[k1];
        epstab[k1] = e1;
        delta1 = e1 - e3;
        err1 = fabs(delta1);
        tol1 = fmax(e1abs, fabs(e3)) * epmach;
        if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) {
            *n = i + i - 1;
            break;
        }
        ss = 1. / delta1 + 1. / delta2 - 1. / delta3;
        epsinf = fabs(ss * e1);
        if (!(epsinf > 1e-4)) {
            *n = i + i - 1;
            break;
        }
        res = e1 + 1. / ss;
        epstab[k1] = res;
        k1 -= 2;
        error = err2 + fabs(res - e2) + err3;
        if (error > *abserr) continue;
        *abserr = error;
        *result = res;
    }
    if (*n == limexp) *n = 2 * (limexp / 2) - 1;
    ib = 1;
    if ((num / 2) * 2 == num) ib = 2;
    ie = newelm + 1;
    for (i = 1; i <= ie; i++) {
        ib2 = ib + 2;
        epstab[ib] = epstab[ib2];
        ib = ib2;
    }
    if (num != *n) {
        indx = num - *n + 1;
        for (i = 1; i <= *n; i++) {
            epstab[i] = epstab[indx];
            indx++;
        }
    }
    if (*nres < 4) {
        res3la[*nres] = *result;
        *abserr = oflow;
        goto L100;
    }
    *abserr = fabs(*result - res3la[3]) + fabs(*result - res3la[2]) + fabs(*result - res3la[1]);
    res3la[1] = res3la[2];
    res3la[2] = res3la[3];
    res3la[3] = *result;
L100:
    *abserr = fmax(*abserr, 5. * epmach * fabs(*result));
}

#define QLIMIT 50
/* DQAGSE when npts == 0, DQAGPE when npts == 1 (one interior break point) or npts == 2 (flag: DQAGPE, no interior
 * point), as scipy.integrate.quad
 * dispatches (points=None -> _qagse ; points=[p] -> _qagpe).  limit = 50, epsabs = 1.49e-8. */
int orc_quad(integrand_t f, void *p, double a, double b, int npts, double point, double epsabs,
             double epsrel, double *result_out, double *abserr_out, int *neval_out, int *last_out)
{
    const double epmach = DBL_EPSILON, uflow = DBL_MIN, oflow = DBL_MAX;
    const int limit = QLIMIT;
    double alist[QLIMIT + 2], blist[QLIMIT + 2], rlist[QLIMIT + 2], elist[QLIMIT + 2];
    int iord[QLIMIT + 2], level[QLIMIT + 2], ndin[4];
    double rlist2[53], res3la[4];
    double abseps = 0, area, area1, area12, area2, a1, a2, b1, b2, correc = 0, defabs, defab1, defab2, dres,
           erlarg = 0, erlast, errbnd, errmax, error1, error2, erro12, errsum, ertest = 0, resabs, reseps = 0, result,
           abserr, small = 0, resa;
    int ier = 0, ierro = 0, iroff1 = 0, iroff2 = 0, iroff3 = 0, k, ksgn, ktmin = 0, last, maxerr, neval = 0,
        nres = 0, nrmax, numrl2, extrap = 0, noext = 0, levmax = 1, levcur = 0, id, jupbnd;
    double sign = 1.;
    int qagp = (npts > 0);
    if (npts == 2) npts = 0; /* DQAGPE without an interior break point: what scipy runs when points=[p] lies outside (a, b) */
    int npts2 = npts + 2, nint = npts + 1;
    result = 0.; abserr = 0.;
    if (qagp) {
        double pts[3];
        if (a > b) sign = -1.;
        pts[0] = fmin(a, b); pts[1] = point; pts[2] = fmax(a, b);
        if (npts == 0) pts[1] = pts[2];
        resabs = 0.;
        a1 = pts[0];
        for (int i = 1; i <= nint; i++) {
            b1 = pts[i];
            dqk21(f, p, a1, b1, &area1, &error1, &defabs, &resa);
            abserr += error1;
            result += area1;
            ndin[i] = 0;
            if (error1 == resa && error1 != 0.) ndin[i] = 1;
            resabs += defabs;
            level[i] = 0;
            elist[i] = error1;
            alist[i] = a1;
            blist[i] = b1;
            rlist[i] = area1;
            iord[i] = i;
            a1 = b1;
        }
        errsum = 0.;
        for (int i = 1; i <= nint; i++) {
            if (ndin[i] == 1) elist[i] = abserr;
            errsum += elist[i];
        }
        last = nint;
        neval = 21 * nint;
        dres = fabs(result);
        errbnd = fmax(epsabs, epsrel * dres);
        if (abserr <= 100. * epmach * resabs && abserr > errbnd) ier = 2;
        if (nint != 1) {
            for (int i = 1; i <= npts; i++) {
                int jlow = i + 1, ind1 = iord[i], ind2, kk = i;
                for (int j = jlow; j <= nint; j++) {
                    ind2 = iord[j];
                    if (elist[ind1] > elist[ind2]) continue;
                    ind1 = ind2;
                    kk = j;
                }
                if (ind1 != iord[i]) {
                    iord[kk] = iord[i];
                    iord[i] = ind1;
                }
            }
            if (limit < npts2) ier = 1;
        }
        if (ier != 0 || abserr <= errbnd) goto L210;
        rlist2[1] = result;
        maxerr = iord[1];
        errmax = elist[maxerr];
        area = result;
        nrmax = 1;
        nres = 0;
        numrl2 = 1;
        ktmin = 0;
        extrap = 0;
        noext = 0;
        erlarg = errsum;
        ertest = errbnd;
        levmax = 1;
        abserr = oflow;
        ksgn = -1;
        if (dres >= (1. - 50. * epmach) * resabs) ksgn = 1;
        last = npts2;
    } else {
        alist[1] = a; blist[1] = b; rlist[1] = 0.; elist[1] = 0.;
        dqk21(f, p, a, b, &result, &abserr, &defabs, &resabs);
        dres = fabs(result);
        errbnd = fmax(epsabs, epsrel * dres);
        last = 1;
        rlist[1] = result;
        elist[1] = abserr;
        iord[1] = 1;
        if (abserr <= 100. * epmach * defabs && abserr > errbnd) ier = 2;
        if (limit == 1) ier = 1;
        if (ier != 0 || (abserr <= errbnd && abserr != resabs) || abserr == 0.) goto L140;
        rlist2[1] = result;
        errmax = abserr;
        maxerr = 1;
        area = result;
        errsum = abserr;
        abserr = oflow;
        nrmax = 1;
        nres = 0;
        numrl2 = 2;
        ktmin = 0;
        extrap = 0;
        noext = 0;
        ksgn = -1;
        if (dres >= (1. - 50. * epmach) * defabs) ksgn = 1;
        resabs = defabs; /* used in the divergence test below (qagse tests against defabs) */
        last = 2;
    }
    /* main loop */
    for (; last <= limit; last++) {
        if (qagp) levcur = level[maxerr] + 1;
        a1 = alist[maxerr];
        b1 = 0.5 * (alist[maxerr] + blist[maxerr]);
        a2 = b1;
        b2 = blist[maxerr];
        erlast = errmax;
        dqk21(f, p, a1, b1, &area1, &error1, &resa, &defab1);
        dqk21(f, p, a2, b2, &area2, &error2, &resa, &defab2);
        neval += 42;
        area12 = area1 + area2;
        erro12 = error1 + error2;
        errsum = errsum + erro12 - errmax;
        area = area + area12 - rlist[maxerr];
        if (defab1 != error1 && defab2 != error2) {
            if (fabs(rlist[maxerr] - area12) <= 1e-5 * fabs(area12) && erro12 >= 0.99 * errmax) {
                if (extrap) iroff2++;
                else iroff1++;
            }
            if (last > 10 && erro12 > errmax) iroff3++;
        }
        if (qagp) {
            level[maxerr] = levcur;
            level[last] = levcur;
        }
        rlist[maxerr] = area1;
        rlist[last] = area2;
        errbnd = fmax(epsabs, epsrel * fabs(area));
        if (iroff1 + iroff2 >= 10 || iroff3 >= 20) ier = 2;
        if (iroff2 >= 5) ierro = 3;
        if (last == limit) ier = 1;
        if (fmax(fabs(a1), fabs(b2)) <= (1. + 100. * epmach) * (fabs(a2) + 1000. * uflow)) ier = 4;
        if (error2 > error1) {
            alist[maxerr] = a2;
            alist[last] = a1;
            blist[last] = b1;
            rlist[maxerr] = area2;
            rlist[last] = area1;
            elist[maxerr] = error2;
            elist[last] = error1;
        } else {
            alist[last] = a2;
            blist[maxerr] = b1;
            blist[last] = b2;
            elist[maxerr] = error1;
            elist[last] = error2;
        }
        dqpsrt(limit, last, &maxerr, &errmax, elist, iord, &nrmax);
        if (errsum <= errbnd) goto L190;
        if (ier != 0) goto L170;
        if (!qagp && last == 2) {
            small = fabs(b - a) * 0.375;
            erlarg = errsum;
            ertest = errbnd;
            rlist2[2] = area;
            continue;
        }
        if (noext) continue;
        erlarg -= erlast;
        if (qagp) {
            if (levcur + 1 <= levmax) erlarg += erro12;
        } else {
            if (fabs(b1 - a1) > small) erlarg += erro12;
        }
        if (!extrap) {
            if (qagp) {
                if (level[maxerr] + 1 <= levmax) continue;
            } else {
                if (fabs(blist[maxerr] - alist[maxerr]) > small) continue;
            }
            extrap = 1;
            nrmax = 2;
        }
        if (!(ierro == 3 || erlarg <= ertest)) {
            id = nrmax;
            jupbnd = last;
            if (last > (2 + limit / 2)) jupbnd = limit + 3 - last;
            int cont = 0;
            for (k = id; k <= jupbnd; k++) {
                maxerr = iord[nrmax];
                errmax = elist[maxerr];
                if (qagp) {
                    if (level[maxerr] + 1 <= levmax) { cont = 1; break; }
                } else {
                    if (fabs(blist[maxerr] - alist[maxerr]) > small) { cont = 1; break; }
                }
                nrmax++;
            }
            if (cont) continue;
        }
        /* perform extrapolation */
        numrl2++;
        rlist2[numrl2] = area;
        if (qagp && numrl2 <= 2) goto L155;
        dqelg(&numrl2, rlist2, &reseps, &abseps, res3la, &nres);
        ktmin++;
        if (ktmin > 5 && abserr < 1e-3 * errsum) ier = 5;
        if (abseps < abserr) {
            ktmin = 0;
            abserr = abseps;
            result = reseps;
            correc = erlarg;
            ertest = fmax(epsabs, epsrel * fabs(reseps));
            if (qagp) { if (abserr < ertest) goto L170; }
            else      { if (abserr <= ertest) goto L170; }
        }
        if (numrl2 == 1) noext = 1;
        if (qagp) { if (ier >= 5) goto L170; }
        else      { if (ier == 5) goto L170; }
    L155:
        maxerr = iord[1];
        errmax = elist[maxerr];
        nrmax = 1;
        extrap = 0;
        if (qagp) levmax++;
        else small *= 0.5;
        erlarg = errsum;
    }
    last = limit; /* loop fell through (Fortran DO leaves last = limit + 1; results use 1..limit) */
L170:
    if (last > limit) last = limit;
    if (abserr == oflow) goto L190;
    if (ier + ierro != 0) {
        if (ierro == 3) abserr += correc;
        if (ier == 0) ier = 3;
        if (result != 0. && area != 0.) {
            if (abserr / fabs(result) > errsum / fabs(area)) goto L190;
        } else {
            if (abserr > errsum) goto L190;
            if (area == 0.) goto L210;
        }
    }
    /* test on divergence */
    if (ksgn == -1 && fmax(fabs(result), fabs(area)) <= resabs * 0.01) goto L210;
    if (0.01 > (result / area) || (result / area) > 100. || errsum > fabs(area)) ier = 6;
    goto L210;
L190:
    if (last > limit) last = limit;
    result = 0.;
    for (k = 1; k <= last; k++) result += rlist[k];
    abserr = errsum;
L210:
    if (ier > 2) ier--;
    if (qagp) result *= sign;
    else neval = 42 * last - 21;
    goto L999;
L140:
    neval = 42 * last - 21;
L999:
    *result_out = result;
    if (abserr_out) *abserr_out = abserr;
    if (neval_out) *neval_out = neval;
    if (last_out) *last_out = last;
    return ier;
}

/* ------------------------------------------------------------------------------------------
 * attenuation along the path (:933-1089, Python branch, non-"optimized" models), reflection = 0
 * ---------------------------------------------------------------------------------------- */
typedef struct { double C0, f; const ice_t *m; int model; long nev; } att_t;

static double att_integrand(double t, void *p) /* dt() :986-988 with ds() :513-517 */
{
    att_t *a = (att_t *)p;
    a->nev++;
    double z = get_z_unmirrored(t, a->C0, a->m);
    double yd = get_y_diff(t, a->C0, a->m);
    double ds = sqrt(yd * yd + 1);
    return ds_over_length(ds, z, a->f, a->model);
}

static double ds_only(double t, void *p) /* ds() :513-517 */
{
    att_t *a = (att_t *)p;
    a->nev++;
    double yd = get_y_diff(t, a->C0, a->m);
    return sqrt(yd * yd + 1);
}

/* np.linspace(a, b, max(int(|a - b| // dx), 3)) or [a] (get_equidistant_steps :56-76); returns the number of points */
static int equidistant_steps(double a, double b, double dx, double *out)
{
    if (a == b) { out[0] = a; return 1; }
    int n = (int)floor(fabs(a - b) / dx);
    if (n < 3) n = 3;
    double step = (b - a) / (n - 1);
    for (int i = 0; i < n; i++) out[i] = i * step + a;
    out[n - 1] = b;
    return n;
}

/* the speed-optimised path integral of models in speedup_attenuation_models (GL3), analyticraytracing.py:998-1064 */
static void attenuation_segments_2d(const double x1[2], double x2m, double z_turn, double C0, const ice_t *m, int model,
                                    int n_freq, const double *freqs, double *att, int *neval)
{
    static double steps[4096];
    const double dx = 10., window = 20.;
    int fallback = (x1[1] - window / 2 < z_turn && z_turn < x2m + window / 2), n;
    if (fallback) {
        double w0 = fmax(x1[1], z_turn - window / 2), w1 = fmin(z_turn + window / 2, x2m);
        n = equidistant_steps(x1[1], w0, dx, steps);
        n += equidistant_steps(w1, x2m, dx, steps + n);
    } else {
        n = equidistant_steps(x1[1], x2m, dx, steps);
    }
    int idx = -2;
    double integrand = 0.;
    int ne_quad = 0;
    if (fallback) {
        int cnt = 0; /* np.digitize(z_turn, steps) - 1 */
        for (int i = 0; i < n; i++) if (steps[i] <= z_turn) cnt++;
        idx = cnt - 1;
        if (idx == n - 1) idx -= 1;
        else if (idx == -1) idx = 0;
        att_t a = { C0, 0., m, model, 0 };
        double err; int la;
        double lo = steps[idx], hi = steps[idx + 1];
        int inside = (fmin(lo, hi) < z_turn && z_turn < fmax(lo, hi));
        orc_quad(ds_only, &a, lo, hi, inside ? 1 : 2, z_turn, 1.49e-8, 1e-2, &integrand, &err, &ne_quad, &la);
    }
    for (int k = 0; k < n_freq; k++) {
        double sum = 0.;
        for (int i = 0; i + 1 < n; i++) {
            double dxa = steps[i + 1] - steps[i], mid = steps[i] + dxa / 2, term;
            if (i == idx) {
                term = integrand / orc_attenuation_length(z_turn, freqs[k], model);
            } else {
                double z = get_z_unmirrored(mid, C0, m);
                double yd = get_y_diff(mid, C0, m);
                term = sqrt(yd * yd + 1) / orc_attenuation_length(z, freqs[k], model) * dxa;
            }
            sum += term;
        }
        att[k] = orc_exp(-1 * sum);
        if (neval) neval[k] = n - 1 + ne_quad;
    }
}

/* x1, x2: 2-D points; freqs: the sparse frequency vector (already chosen); att: exp(-integral) */
void orc_attenuation_2d(const double x1[2], const double x2[2], double C0, const double ice[3], int model,
                        int n_freq, const double *freqs, double *att, int *neval)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    double x2m = get_z_mirrored(x1, x2, C0, &m);
    double g, z_turn;
    get_turning_point(m.n_ice * m.n_ice - 1. / (C0 * C0), &m, &g, &z_turn);
    if (model == 5) { /* speedup_attenuation_models = ["GL3"] (:63) */
        attenuation_segments_2d(x1, x2m, z_turn, C0, &m, model, n_freq, freqs, att, neval);
        return;
    }
    int npts = (x1[1] < z_turn && z_turn < x2m) ? 1 : 0;
    for (int i = 0; i < n_freq; i++) {
        att_t a = { C0, freqs[i], &m, model, 0 };
        double res, err;
        int ne, la;
        orc_quad(att_integrand, &a, x1[1], x2m, npts, z_turn, 1.49e-8, 1e-2, &res, &err, &ne, &la);
        att[i] = orc_exp(-1 * res);
        if (neval) neval[i] = ne;
    }
}

/* ------------------------------------------------------------------------------------------
 * 3-D wrapper (class ray_tracing, :2057-2090, :2118-2130, :2560-2624, :2650-2742)
 * Fixed stride MAXS = 2 solutions per pair (2 + 4*n_reflections with n_reflections = 0).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double X1[3], X2[3], R[9], x1[2], x2[2]; int swap; } geom_t;

static void set_start_and_end_point(const double x1[3], const double x2[3], geom_t *g)
{
    g->swap = 0;
    memcpy(g->X1, x1, 24);
    memcpy(g->X2, x2, 24);
    if (g->X2[2] < g->X1[2]) {
        g->swap = 1;
        memcpy(g->X2, x1, 24);
        memcpy(g->X1, x2, 24);
    }
    double dX[3] = { g->X2[0] - g->X1[0], g->X2[1] - g->X1[1], g->X2[2] - g->X1[2] };
    /* rotation by dPhi = -arctan2(dy, dx): cos / sin taken algebraically */
    double rho = sqrt(dX[0] * dX[0] + dX[1] * dX[1]);
    double c = 1., s = 0.;
    if (rho > 0) {
        c = dX[0] / rho;
        s = -(dX[1] / rho);
    }
    double R[9] = { c, -s, 0, s, c, 0, 0, 0, 1 };
    memcpy(g->R, R, sizeof R);
    double X2r[3];
    for (int i = 0; i < 3; i++) X2r[i] = R[3 * i] * dX[0] + R[3 * i + 1] * dX[1] + R[3 * i + 2] * dX[2] + g->X1[i];
    g->x1[0] = g->X1[0]; g->x1[1] = g->X1[2];
    g->x2[0] = X2r[0];   g->x2[1] = X2r[2];
}

static void rotate_back(const geom_t *g, const double v2d[3], double out[3]) /* np.dot(R.T, v) */
{
    for (int i = 0; i < 3; i++) out[i] = g->R[i] * v2d[0] + g->R[3 + i] * v2d[1] + g->R[6 + i] * v2d[2];
}

#define MAXS 2

/* batch over pairs; arrays are [n][MAXS](...) with NaN / 0 padding like the reference's HDF5 tables */
void orc_raytrace_batch(long n, const double *x1, const double *x2, const double ice[3],
                        int *n_sol, int *type, double *C0, double *C1, double *D, double *T,
                        double *launch, double *receive, double *refl_angle, double *hybr_x, double *hybr_fun)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    for (long i = 0; i < n; i++) {
        geom_t g;
        set_start_and_end_point(x1 + 3 * i, x2 + 3 * i, &g);
        double c0[3], c1[3], hx, hf;
        int ty[3];
        int ns = orc_find_solutions_2d(g.x1, g.x2, ice, c0, c1, ty, &hx, &hf, NULL);
        if (hybr_x) hybr_x[i] = hx;
        if (hybr_fun) hybr_fun[i] = hf;
        if (ns > MAXS) ns = 0; /* :2127-2130 */
        n_sol[i] = ns;
        for (int s = 0; s < MAXS; s++) {
            long k = i * MAXS + s;
            type[k] = 0;
            C0[k] = C1[k] = D[k] = T[k] = refl_angle[k] = NAN;
            for (int d = 0; d < 3; d++) launch[3 * k + d] = receive[3 * k + d] = NAN;
            if (s >= ns) continue;
            type[k] = ty[s];
            C0[k] = c0[s];
            C1[k] = c1[s];
            path_length_and_time(g.x1, g.x2, c0[s], &m, &D[k], &T[k]);
            /* launch angle la; receive angle ra = pi - angle at x2: sin ra = s2, cos ra = -c2 */
            double sL, cL, s2, c2;
            get_angle_sincos(g.x1, g.x1, c0[s], &m, &sL, &cL);
            get_angle_sincos(g.x2, g.x1, c0[s], &m, &s2, &c2);
            double lv[3] = { sL, 0, cL }, rv[3] = { -s2, 0, -c2 };
            if (g.swap) {
                lv[0] = -s2; lv[2] = -c2;
                rv[0] = sL;  rv[2] = cL;
            }
            rotate_back(&g, lv, launch + 3 * k);
            rotate_back(&g, rv, receive + 3 * k);
            refl_angle[k] = get_reflection_angle(g.x1, g.x2, c0[s], &m);
        }
    }
}

/* ---- bottom reflections (ice shelf, medium.reflection: Moore's Bay) ---------------------------------------------
 * ray_tracing.find_solutions (:2118-2130): the plain call, then (reflection i, case 1), (i, case 2) for i = 1..n_reflections;
 * more than 2 + 4 n_reflections solutions -> none.  Arrays are [n][stride], stride = 2 + 4 n_reflections.
 * given != 0: the solution records (n_sol, type, C0, C1, reflection, reflection_case) are inputs (set_solution :2092). */
void orc_raytrace_batch_refl(long n, const double *x1, const double *x2, const double ice[3], int n_reflections,
                             double z_refl, int given, int *n_sol, int *type, double *C0, double *C1, int *reflection,
                             int *reflection_case, double *D, double *T, double *launch, double *receive,
                             double *refl_angle, int *n_surface, int *n_segments, int *surface_mask)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    const int stride = 2 + 4 * n_reflections;
    for (long i = 0; i < n; i++) {
        geom_t g;
        set_start_and_end_point(x1 + 3 * i, x2 + 3 * i, &g);
        if (!given) {
            double c0[64], c1[64];
            int ty[64], rf[64], rc[64], ns = 0;
            for (int r = 0; r <= n_reflections; r++)
                for (int cs = 1; cs <= (r == 0 ? 1 : 2); cs++) {
                    /* with a reflective layer every call -- the plain one included -- is the reference's procedure (find_refl_kernel) */
                    const int keep = orc_force_procedure;
                    orc_force_procedure = 1;
                    int k = orc_find_solutions_2d_refl(g.x1, g.x2, ice, r, cs, z_refl, c0 + ns, c1 + ns, ty + ns, NULL, NULL, NULL);
                    orc_force_procedure = keep;
                    for (int j = 0; j < k; j++) { rf[ns + j] = r; rc[ns + j] = cs; }
                    ns += k;
                }
            if (ns > stride) ns = 0; /* :2127-2130 */
            n_sol[i] = ns;
            for (int s = 0; s < stride; s++) {
                long k = i * stride + s;
                type[k] = reflection[k] = reflection_case[k] = 0;
                C0[k] = C1[k] = NAN;
                if (s < ns) { type[k] = ty[s]; C0[k] = c0[s]; C1[k] = c1[s]; reflection[k] = rf[s]; reflection_case[k] = rc[s]; }
            }
        }
        for (int s = 0; s < stride; s++) {
            long k = i * stride + s;
            D[k] = T[k] = refl_angle[k] = NAN;
            n_surface[k] = n_segments[k] = surface_mask[k] = 0;
            for (int d = 0; d < 3; d++) launch[3 * k + d] = receive[3 * k + d] = NAN;
            if (s >= n_sol[i]) continue;
            const int rf = reflection[k], rc = reflection_case[k];
            path_length_and_time_refl(g.x1, g.x2, C0[k], rf, rc, z_refl, &m, &D[k], &T[k]);
            double sL, cL, s2, c2;
            get_angle_sincos_refl(g.x1, g.x1, C0[k], rf, rc, z_refl, &m, &sL, &cL);
            get_angle_sincos_refl(g.x2, g.x1, C0[k], rf, rc, z_refl, &m, &s2, &c2);
            double lv[3] = { sL, 0, cL }, rv[3] = { -s2, 0, -c2 };
            if (g.swap) {
                lv[0] = -s2; lv[2] = -c2;
                rv[0] = sL;  rv[2] = cL;
            }
            rotate_back(&g, lv, launch + 3 * k);
            rotate_back(&g, rv, receive + 3 * k);
            double ra[ORC_MAX_REFL + 1];
            int nseg = reflection_angles(g.x1, g.x2, C0[k], rf, rc, z_refl, &m, ra);
            for (int j = 0; j < nseg; j++)
                if (!isnan(ra[j])) { refl_angle[k] = ra[j]; n_surface[k]++; surface_mask[k] |= 1 << j; } /* the same angle in every segment that has one */
            n_segments[k] = nseg;
        }
    }
}

/* get_attenuation_along_path (:933-1089) with bottom reflections: the product over the path segments */
void orc_attenuation_batch_refl(long n, const double *x1, const double *x2, const double *C0, const int *reflection,
                                const int *reflection_case, const double ice[3], double z_refl, int model, int n_freq,
                                const double *freqs, double *att)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n_freq);
    for (long i = 0; i < n; i++) {
        geom_t g;
        set_start_and_end_point(x1 + 3 * i, x2 + 3 * i, &g);
        for (int k = 0; k < n_freq; k++) att[i * n_freq + k] = isnan(C0[i]) ? NAN : 1.;
        if (isnan(C0[i])) continue;
        seg_t segs[ORC_MAX_REFL + 1];
        int ns = path_segments(g.x1, g.x2, C0[i], reflection[i], reflection_case[i], z_refl, &m, segs);
        for (int j = 0; j < ns; j++) {
            double a[2], b[2];
            segment_end_points(&segs[j], j, reflection_case[i], g.x1, a, b);
            orc_attenuation_2d(a, b, C0[i], ice, model, n_freq, freqs, tmp, NULL);
            for (int k = 0; k < n_freq; k++) att[i * n_freq + k] *= tmp[k];
        }
    }
    free(tmp);
}

/* attenuation for a batch of rays given 3-D end points and C0 (get_attenuation :2744) */
void orc_attenuation_batch(long n, const double *x1, const double *x2, const double *C0, const double ice[3],
                           int model, int n_freq, const double *freqs, double *att, int *neval)
{
    for (long i = 0; i < n; i++) {
        geom_t g;
        set_start_and_end_point(x1 + 3 * i, x2 + 3 * i, &g);
        if (isnan(C0[i])) {
            for (int k = 0; k < n_freq; k++) att[i * n_freq + k] = NAN;
            continue;
        }
        orc_attenuation_2d(g.x1, g.x2, C0[i], ice, model, n_freq, freqs, att + i * n_freq,
                           neval ? neval + i * n_freq : NULL);
    }
}

/* test hooks */
double orc_delta_y(double logC0, const double x1[2], const double x2[2], const double ice[3])
{
    ice_t m = { ice[0], ice[1], ice[2] };
    return get_delta_y(C0_from_log(logC0, &m), x1, x2, &m);
}

/* u, v of find_solutions_bracketed on a grid of log C0 (tools/root_shapes.py checks the monotony of u and the single maximum of v) */
void orc_uv_grid(int n, const double *logC0, const double x1[2], const double x2[2], const double ice[3], double *u, double *v)
{
    ice_t m = { ice[0], ice[1], ice[2] };
    uv_t o = { x1, x2, &m, 0, 0., get_gamma(x1[1], &m), get_gamma(x2[1], &m) };
    for (int i = 0; i < n; i++) {
        if (logC0[i] < 0 && 0) continue;
        o.x_lo = logC0[i];   /* t = 0: the point itself */
        uv_at(0., &o, u + i, v + i);
    }
}

/* both finders on 2-D pairs (tests): n_sol and sorted C0 [n][3], objective evaluations per pair */
void orc_find_solutions_2d_batch(long n, const double *x1, const double *x2, const double ice[3], int reference_procedure,
                                 int *n_sol, double *C0, int *nfev)
{
    const int keep = orc_reference_procedure;
    orc_reference_procedure = reference_procedure;
    for (long i = 0; i < n; i++) {
        double c[3], c1[3];
        int t[3];
        int k = orc_find_solutions_2d(x1 + 2 * i, x2 + 2 * i, ice, c, c1, t, NULL, NULL, nfev + i);
        n_sol[i] = k;
        for (int j = 0; j < 3; j++) C0[3 * i + j] = j < k ? c[j] : NAN;
    }
    orc_reference_procedure = keep;
}

/* detmath_c.h for the accuracy test (tests/test_oracle_golden.py::test_detmath_accuracy) */
double orc_log_value(double x) { return orc_log(x); }
