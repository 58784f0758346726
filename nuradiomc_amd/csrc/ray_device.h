// ray_device.h -- device-side analytic ray geometry in n(z) = n_ice - delta_n * exp(z / z_0).
//
// MI355X (gfx950) product code.  One lane owns one (vertex, channel) pair or one (ray, frequency)
// item; everything that depends only on the ice model lives in IceConst (kernel argument, SGPRs),
// everything that depends only on the launch parameter C0 lives in C0State (computed once per
// objective evaluation and reused by the three y(z) evaluations of that objective).
//
// Mirrors the behaviour of NuRadioMC/SignalProp/analyticraytracing.py (Python path):
//   get_y :105, get_turning_point :133, get_y_with_z_mirror :160, get_delta_y :204,
//   get_z_unmirrored :293, get_y_diff :306, determine_solution_type :1365, get_z_mirrored :496,
//   get_angle :1161, get_reflection_angle :1201, get_path_length_analytic :602,
//   get_travel_time_analytic :692.
#pragma once
#include <hip/hip_runtime.h>
#include "detmath.h"

namespace nrhip {

struct IceConst {
    double n_ice, delta_n, z_0;
    double n2;      // n_ice^2
    double b;       // 2 n_ice
    double qb2;     // 0.25 b^2
    double inv_n;   // 1 / n_ice
    double inv_z0;  // unused in parity-critical expressions (z / z_0 is kept as a division)
};

__host__ __device__ inline IceConst make_ice(double n_ice, double delta_n, double z_0)
{
    IceConst m;
    m.n_ice = n_ice; m.delta_n = delta_n; m.z_0 = z_0;
    m.n2 = n_ice * n_ice;
    m.b = 2 * n_ice;
    m.qb2 = 0.25 * m.b * m.b;
    m.inv_n = 1. / n_ice;
    m.inv_z0 = 1. / z_0;
    return m;
}

// exp / log on this path are the bit-reproducible ones of detmath.h (see there why)
__device__ inline double gamma_of_z(double z, const IceConst& m) { return m.delta_n * det_exp(z / m.z_0); }
__device__ inline double n_of_z(double z, const IceConst& m) { return m.n_ice - m.delta_n * det_exp(z / m.z_0); }

// Everything the path function needs for one value of C0.
struct C0State {
    double C0;
    double c;        // n_ice^2 - C0^-2
    double two_sc;   // 2 sqrt(c)
    double two_c;    // 2 c
    double pref;     // z_0 (n_ice^2 C0^2 - 1)^-1/2
    double g_turn;   // gamma at the turning point
    double z_turn;   // depth of the turning point (clamped to the surface)
    double y_turn0;  // y(gamma_turn) with C1 = 0
};

// y(gamma) with C1 = 0  (analyticraytracing.py:105-125)
__device__ inline double y_of_gamma(double g, const C0State& s, const IceConst& m)
{
    double root = fabs(g * g - g * m.b + s.c);
    double logarg = g / (s.two_sc * sqrt(root) - m.b * g + s.two_c);
    return s.pref * det_log(logarg);
}

__device__ inline C0State make_c0(double C0, const IceConst& m)
{
    C0State s;
    s.C0 = C0;
    s.c = m.n2 - 1. / (C0 * C0);
    s.two_sc = 2 * sqrt(s.c);
    s.two_c = 2 * s.c;
    s.pref = m.z_0 / sqrt(m.n2 * C0 * C0 - 1);
    double g2 = m.b * 0.5 - sqrt(m.qb2 - s.c);
    double z2 = det_log(g2 / m.delta_n) * m.z_0;
    if (z2 > 0) {  // a surface reflection is a turning point at z = 0
        z2 = 0;
        g2 = m.delta_n;
    }
    s.g_turn = g2;
    s.z_turn = z2;
    s.y_turn0 = y_of_gamma(g2, s, m);
    return s;
}

// get_y_with_z_mirror(z, C0, C1 = 0) (:160-184); g_of_z = delta_n exp(z / z_0) precomputed by the caller
__device__ inline double y_mirror0(double z, double g_of_z, const C0State& s, const IceConst& m)
{
    if (z < s.z_turn) return y_of_gamma(g_of_z, s, m);
    return 2 * s.y_turn0 - y_of_gamma(gamma_of_z(2 * s.z_turn - z, m), s, m);
}

struct Pair2D {
    double y1, z1, y2, z2;  // 2-D start / stop (y = horizontal, z = depth), stop is the higher point
    double g1, g2;          // gamma(z1), gamma(z2)
};

// signed miss distance at x2 of the ray launched from x1 with parameter C0 (:204-272, reflection = 0)
__device__ __noinline__ double delta_y(double logC0, const Pair2D& p, const IceConst& m)
{
    double C0 = det_exp(logC0) + m.inv_n;
    if (C0 < m.inv_n) return -INFINITY;
    C0State s = make_c0(C0, m);
    double C1 = p.y1 - y_mirror0(p.z1, p.g1, s, m);
    double y_turn = s.y_turn0 + C1;
    if (s.z_turn < p.z2) {  // turning point below the receiver: smooth penalty (:247-253)
        double dz = s.z_turn - p.z2, dy = y_turn - p.y2;
        return -(sqrt(dz * dz + dy * dy) + 10 * fabs(dz));
    }
    double y2 = y_of_gamma(p.g2, s, m) + C1;
    if (y_turn > p.y2) return p.y2 - y2;    // direct branch
    return -1 * (p.y2 - (2 * y_turn - y2));  // mirrored branch
}

// the same objective with the pair's six numbers in LDS columns (sp[k * stride], k = y1, z1, y2, z2, g1, g2): the root finders call it
// ~1e2 times per pair, and a Pair2D passed by reference to a non-inlined function lives in scratch -- 48 B re-read per call, 24 GB
// of HBM traffic per 5e6 pairs (the scratch of all resident waves exceeds the L2)
__device__ __noinline__ double delta_y_lds(double logC0, const double* __restrict__ sp, int stride, const IceConst& m)
{
    Pair2D p;
    p.y1 = sp[0]; p.z1 = sp[stride]; p.y2 = sp[2 * stride]; p.z2 = sp[3 * stride]; p.g1 = sp[4 * stride]; p.g2 = sp[5 * stride];
    double C0 = det_exp(logC0) + m.inv_n;
    if (C0 < m.inv_n) return -INFINITY;
    C0State s = make_c0(C0, m);
    double C1 = p.y1 - y_mirror0(p.z1, p.g1, s, m);
    double y_turn = s.y_turn0 + C1;
    if (s.z_turn < p.z2) {  // turning point below the receiver: smooth penalty (:247-253)
        double dz = s.z_turn - p.z2, dy = y_turn - p.y2;
        return -(sqrt(dz * dz + dy * dy) + 10 * fabs(dz));
    }
    double y2 = y_of_gamma(p.g2, s, m) + C1;
    if (y_turn > p.y2) return p.y2 - y2;    // direct branch
    return -1 * (p.y2 - (2 * y_turn - y2));  // mirrored branch
}

__device__ inline double C1_of(const C0State& s, const Pair2D& p, const IceConst& m)
{
    return p.y1 - y_mirror0(p.z1, p.g1, s, m);
}

// 1 direct / 2 refracted / 3 reflected (:1365-1398)
__device__ inline int solution_type(const C0State& s, double C1, const Pair2D& p)
{
    double y_turn = s.y_turn0 + C1;
    if (p.y2 < y_turn) return 1;
    if (s.z_turn == 0) return 3;
    return 2;
}

// |dy/dz| at depth z on the un-mirrored branch (:306-355, eq. C.12); +inf at / beyond the turning point
__device__ inline double abs_dydz(double z, double C0, const IceConst& m)
{
    double nz = n_of_z(z, m);
    double q = (C0 * C0) * (nz * nz);  // C_0**2 * n_z**2
    return (q > 1) ? 1 / sqrt(q - 1) : INFINITY;
}

// get_z_mirrored(x1, x, C0)[1] (:496-511) for stop point (y, z)
__device__ inline double z_mirrored(double y, double z, const C0State& s, double C1, const Pair2D& p)
{
    double y_turn = s.y_turn0 + C1;
    if (y_turn < y) return p.z1 + fabs(s.z_turn - p.z1) + fabs(s.z_turn - z);
    return z;
}

// get_angle(x, x1, C0) (:1161-1193) as (sin, cos) of the angle to the +z axis of the ray at (y, z): the reference's
// arctan(dy/dz) (+ pi if negative) followed by sin / cos is evaluated algebraically (sqrt and division only), so that
// launch / receive vectors and beta = n sin(theta) are bit-reproducible.
__device__ inline void ray_sincos(double y, double z, const C0State& s, double C1, const Pair2D& p, const IceConst& m,
                                  double* sn, double* cs)
{
    double zm = z_mirrored(y, z, s, C1, p);
    double zu = (zm > s.z_turn) ? 2 * s.z_turn - zm : zm;
    double dy = abs_dydz(zu, s.C0, m);
    bool neg = (zu != zm);
    if (isinf(dy)) {
        *sn = 1.;
        *cs = 0.;
        return;
    }
    double h = sqrt(1 + dy * dy);
    *sn = dy / h;
    *cs = (neg ? -1. : 1.) / h;
}

// closed-form path length [m] and travel time [ns] (:602-783, Bouma thesis), receiver in ice
__device__ inline void path_length_time(const C0State& s, double C1, int type, double sin_launch,
                                        const Pair2D& p, const IceConst& m, double* D, double* T)
{
    const double c_light = 0.299792458;
    double n1 = n_of_z(p.z1, m);
    double beta = n1 * sin_launch;
    double beta2 = beta * beta;
    double alpha = m.n2 - beta2;
    double sa = sqrt(alpha);
    double zz[3] = {p.z1, p.z2, (type == 2) ? s.z_turn : 0.};
    double sv[3], cv[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double nz = n_of_z(zz[i], m);
        double gam = fmax(0., nz * nz - beta2);
        double sg = sqrt(gam);
        double l1 = sqrt(alpha * gam) + m.n_ice * nz - beta2;
        double l2 = sg + nz;
        double ll1 = det_log(l1), ll2 = det_log(l2);
        sv[i] = m.n_ice / sa * (zz[i] - m.z_0 * ll1) + m.z_0 * ll2;
        cv[i] = m.z_0 * (sg - m.n2 / sa * ll1 + m.n_ice * ll2) + m.n2 * zz[i] / sa;
    }
    if (type == 1) {
        *D = sv[1] - sv[0];
        *T = (cv[1] - cv[0]) / c_light;
    } else {
        *D = 2 * sv[2] - sv[0] - sv[1];
        *T = (2 * cv[2] - cv[0] - cv[1]) / c_light;
    }
}

}  // namespace nrhip
