// root_device.h -- the two scalar root finders of the reference's solution finder, restated for one unknown:
// MINPACK HYBRD as driven by scipy.optimize.root(method='hybr') and Brent's method as in scipy's brentq.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

// ---- MINPACK HYBRD, n = 1 (scipy.optimize.root(method='hybr', tol=1e-6): factor 100, mode 1, --------
// ---- maxfev 400, forward-difference Jacobian with eps = sqrt(machine eps), Broyden updates) ---------
// For one unknown the QR factor of the Jacobian a is Q = -1, R = -a, the dogleg step is the Newton
// step clipped to the trust radius, and the rank-one update is r += u v.
template <class F>
__device__ inline double hybrd1(F&& fcn, double x, double xtol, double* f_out)
{
    const double epsmch = 2.220446049250313e-16;
    const double p1 = .1, p5 = .5, p001 = .001, p0001 = 1e-4;
    const int maxfev = 400;
    double fvec = fcn(x);
    int nfev = 1;
    double fnorm = fabs(fvec);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double diag = 0, delta = 0, xnorm = 0;
    const double eps = 1.4901161193847656e-08;  // sqrt(epsmch)
    bool done = false;
    while (!done) {
        bool jeval = true;
        double h = eps * fabs(x);
        if (h == 0.) h = eps;
        double a = (fcn(x + h) - fvec) / h;
        nfev++;
        double acnorm = fabs(a);
        bool nz = (a != 0.);
        if (iter == 1) {
            diag = nz ? acnorm : 1.;
            xnorm = fabs(diag * x);
            delta = 100. * xnorm;
            if (delta == 0.) delta = 100.;
        }
        // Q^T f : Householder reflection with v = 2 maps f -> f + 2 * (-(2 f) / 2) = -f
        double qtf = fvec;
        if (nz) qtf += 2. * (-(2. * fvec) / 2.);
        double r = -a;
        double q = nz ? -1. : 1.;
        if (acnorm > diag) diag = acnorm;
        for (;;) {
            double temp = r;
            if (temp == 0.) {
                temp = epsmch * fabs(r);
                if (temp == 0.) temp = epsmch;
            }
            double xs = qtf / temp;
            double qnorm = fabs(diag * xs);
            if (qnorm > delta) {  // clip to the trust region (dogleg, n = 1)
                double g = r * qtf / diag;
                double gnorm = fabs(g);
                double sgnorm = 0., alpha = delta / qnorm, w1 = g;
                if (gnorm != 0.) {
                    w1 = (g / gnorm) / diag;
                    double t = fabs(r * w1);
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
                xs = (1. - alpha) * fmin(sgnorm, delta) * w1 + alpha * xs;
            }
            double step = -xs;
            double xt = x + step;
            double pnorm = fabs(diag * step);
            if (iter == 1) delta = fmin(delta, pnorm);
            double f_new = fcn(xt);
            nfev++;
            double fnorm1 = fabs(f_new);
            double actred = -1.;
            if (fnorm1 < fnorm) actred = 1. - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            double w3 = qtf + r * step;
            double tnorm = fabs(w3);
            double prered = 0.;
            if (tnorm < fnorm) prered = 1. - (tnorm / fnorm) * (tnorm / fnorm);
            double ratio = (prered > 0.) ? actred / prered : 0.;
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
            if (ratio >= p0001) {
                x = xt;
                fvec = f_new;
                xnorm = fabs(diag * x);
                fnorm = fnorm1;
                iter++;
            }
            nslow1++;
            if (actred >= p001) nslow1 = 0;
            if (jeval) nslow2++;
            if (actred >= p1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0.) { done = true; break; }
            if (nfev >= maxfev || p1 * fmax(p1 * delta, pnorm) <= epsmch * xnorm || nslow2 == 5 || nslow1 == 10) {
                done = true;
                break;
            }
            if (ncfail == 2) break;  // re-evaluate the Jacobian
            double sum = q * f_new;
            double v = (sum - w3) / pnorm;
            double u = diag * ((diag * step) / pnorm);
            if (ratio >= p0001) qtf = sum;
            r = r + u * v;
            jeval = false;
        }
    }
    *f_out = fvec;
    return x;
}

// ---- Brent's method as in scipy/optimize/Zeros/brentq.c (xtol 2e-12, rtol 4 eps, 100 iterations) -----
// fa, fb are the already-evaluated end point values (the reference evaluates them for its sign test).
template <class F>
__device__ inline double brentq(F&& f, double xa, double xb, double fa, double fb)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0., fpre = fa, fcur = fb, fblk = 0., spre = 0., scur = 0.;
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    for (int i = 0; i < 100; i++) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
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
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
                spre = scur;
                scur = stry;
            } else {
                spre = sbis;
                scur = sbis;
            }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur);
    }
    return xcur;
}

// may brentq(a, b) be called with these end values?  The reference asks np.sign(fa) != np.sign(fb) (np_sign_differs) and scipy's
// brentq then raises unless fa * fb <= 0 -- an end value of exactly zero (of either sign) is returned as the root.  A pair whose
// end points lie exactly above each other has delta_y(log C0 = 100) = -0.0: the vertical ray.
__device__ inline bool np_sign_differs(double a, double b);
__device__ inline bool brent_bracket_ok(double fa, double fb)
{
    return np_sign_differs(fa, fb) && (fa == 0 || fb == 0 || signbit(fa) != signbit(fb));
}

__device__ inline bool np_sign_differs(double a, double b)
{
    if (isnan(a) || isnan(b)) return true;  // np.sign(nan) != anything
    int sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
    return sa != sb;
}

}  // namespace nrhip
