// birefringence.hip -- birefringent pulse propagation along the analytic ray path on MI355X (gfx950).
//
// Reference: NuRadioMC/SignalProp/analyticraytracing.py get_pulse_propagation_birefringence (:2369-2445): the path is cut
// into acc = int(D / m) points (get_path :1239-1291, :2148-2163); per step the three principal indices
// n(z) + spline_i(-z) - 1.78 (medium_base.py:378-420), the two effective indices (:2165-2207), the two eigen-polarisations
// in the on-sky basis of the step direction (:2243-2367) and E <- R^T diag(1, shift by t_1 - t_0) R E on the
// (eTheta, ePhi) spectra, R = their (theta, phi) components.
//
// Two kernels.  `bire_steps_kernel`: one LANE per path step -- closed-form path points i and i + 1, de Boor evaluation of
// the three depth splines, effective indices, polarisation vectors -> (a, b, c, d, t_1 - t_0) per step, 40 B.
// `bire_propagate_kernel`: a LANE owns nine frequency bins k = lane + j T, the block walks the ray's steps in order with the
// step records staged through LDS in tiles (every lane needs every step: broadcast reads).  The phase of a step is linear in the
// bin number, so per step a lane takes one small-angle sine / cosine (polynomial) and reaches its other bins by complex
// multiplication with exp(i T theta), which the staging threads compute once per step: 24 FP64 operations per (step, bin), of
// them 20 the two real 2 x 2 by complex-vector products.  The chain of non-commuting 2 x 2 factors is inherently sequential per
// bin, so the parallel axes are bins x rays.  FP64 VALU bound: a ray of 2000 steps and 2049 bins is 1e8 fused multiply-adds
// against 80 kB of step records and 131 kB of spectra.
#include <hip/hip_runtime.h>
#include "ray_device.h"
#include "birefringence.h"

namespace nrhip {

// sum_i c_i B_{i,3}(x): FITPACK splev's recursion (fpbspl); outside the knot range the end pieces continue (ext = 0)
__device__ inline double bire_spline(const double* __restrict__ t, const double* __restrict__ c, int n, double x)
{
    const int k = 3;
    int lo = 0, hi = n;  // last knot index with t[l] <= x (searchsorted right - 1)
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (t[mid] <= x) lo = mid + 1; else hi = mid;
    }
    int l = lo - 1;
    l = l < k ? k : (l > n - k - 2 ? n - k - 2 : l);
    double h[4] = {1., 0., 0., 0.}, hh[3];
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.;
        for (int i = 0; i < j; i++) {
            const int li = l + 1 + i, lj = li - j;
            if (t[li] == t[lj]) {
                h[i + 1] = 0.;
                continue;
            }
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] += f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
    double s = 0.;
    for (int i = 0; i <= k; i++) s += c[l - k + i] * h[i];
    return s;
}

// hp.cartesian_to_spherical + on_sky_birefringence (:2339-2367): theta / phi components of p seen along `dir`
__device__ inline void bire_on_sky(const double dir[3], const double p[3], double* p_theta, double* p_phi)
{
    // theta = acos(dir_z / r), phi = atan2(dir_y, dir_x) of the reference enter only through their sines and cosines, which are
    // ratios of the components (no inverse trigonometry + sincos: ~4 x fewer instructions, equal to rounding)
    const double rho2 = dir[0] * dir[0] + dir[1] * dir[1];
    const double r = sqrt(rho2 + dir[2] * dir[2]), rho = sqrt(rho2);
    double ct = 1., st = 0., cp = signbit(dir[0]) ? -1. : 1., sp = 0.;
    if (r != 0.) { ct = dir[2] / r; st = rho / r; }
    if (rho != 0.) { cp = dir[0] / rho; sp = dir[1] / rho; }
    *p_theta = ct * cp * p[0] + ct * sp * p[1] + (-st) * p[2];
    *p_phi = (-sp) * p[0] + cp * p[1] + 0 * p[2];
}

__device__ inline void bire_simple(double n, const double dir[3], double nx, double ny, double nz, double p[3])
{
    p[0] = dir[0] / (n * n - nx * nx);
    p[1] = dir[1] / (n * n - ny * ny);
    p[2] = dir[2] / (n * n - nz * nz);
    const double nrm = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    p[0] /= nrm; p[1] /= nrm; p[2] /= nrm;
}

// path point j of acc (get_path): 3-D position incl. the rotation to the ice-flow frame
__device__ inline void bire_path_point(int j, int acc, double zstart, double zstop, const C0State& s, double C1, const IceConst& m,
                                       double x1y, double x1z, double X1x, double X1y, double cph, double sph, double ca,
                                       double sa, double out[3])
{
    const double step = (zstop - zstart) / (acc - 1);
    const double z = (j == acc - 1) ? zstop : j * step + zstart;  // np.linspace
    const double y_turn = s.y_turn0 + C1;
    double y, zs;
    if (z < s.z_turn) {
        y = y_of_gamma(gamma_of_z(z, m), s, m) + C1;
        zs = z;
    } else {
        y = 2 * y_turn - (y_of_gamma(gamma_of_z(2 * s.z_turn - z, m), s, m) + C1);
        zs = 2 * s.z_turn - z;
    }
    const double dPx = y - x1y, dPz = zs - x1z;           // path_2d - (X1[0], 0, X1[2])
    const double px = cph * dPx + X1x, py = -sph * dPx + X1y;  // R^T dP + X1
    out[0] = ca * px - sa * py;
    out[1] = sa * px + ca * py;
    out[2] = dPz + x1z;
}

// The three depth splines as polynomial pieces in LDS: per knot interval l (the piece FITPACK's splev evaluates for
// t_l <= x < t_l+1, the end pieces beyond the knot range) the cubic through four equally spaced de Boor values, kept in Newton
// form about t_l: 6 doubles (t_l, h, f0, d1, d2, d3), p(u) = f0 + u (d1 + (u - h) (d2 + (u - 2 h) d3)), u = x - t_l.
// Equal to the recursion to a few ulp of the coefficients' size; ~10 x fewer operations per evaluation than the recursion with
// its six divisions, and the knot search runs on LDS.
struct BireSplines { double knot[BIRE_MAX_KNOTS]; double piece[BIRE_MAX_KNOTS][6]; };

__global__ void bire_pieces_kernel(BireBatch b)
{
    double* out = b.spline_pieces;
    const int nk = b.n_knots[0] + b.n_knots[1] + b.n_knots[2];
    for (int q = threadIdx.x; q < nk; q += blockDim.x) {
        int s = 0, base = 0;
        while (s < 2 && q >= base + b.n_knots[s]) { base += b.n_knots[s]; s++; }
        const double* t = b.knots + base;
        const double* c = b.coeffs + base;
        const int n = b.n_knots[s], l = q - base;
        double rec[6] = {t[l], 0., 0., 0., 0., 0.};
        if (l >= 3 && l <= n - 5 && t[l + 1] > t[l]) {
            const double h = (t[l + 1] - t[l]) / 3.;
            double f[4];
            for (int m4 = 0; m4 < 4; m4++) {   // de Boor on piece l (fpbspl), as bire_spline does
                const double x = t[l] + m4 * h;
                double hq[4] = {1., 0., 0., 0.}, hh[3];
                for (int j = 1; j <= 3; j++) {
                    for (int i = 0; i < j; i++) hh[i] = hq[i];
                    hq[0] = 0.;
                    for (int i = 0; i < j; i++) {
                        const int li = l + 1 + i, lj = li - j;
                        if (t[li] == t[lj]) { hq[i + 1] = 0.; continue; }
                        const double fq = hh[i] / (t[li] - t[lj]);
                        hq[i] += fq * (t[li] - x);
                        hq[i + 1] = fq * (x - t[lj]);
                    }
                }
                double v = 0.;
                for (int i = 0; i <= 3; i++) v += c[l - 3 + i] * hq[i];
                f[m4] = v;
            }
            const double d1a = (f[1] - f[0]) / h, d1b = (f[2] - f[1]) / h, d1c = (f[3] - f[2]) / h;
            const double d2a = (d1b - d1a) / (2. * h), d2b = (d1c - d1b) / (2. * h);
            rec[1] = h; rec[2] = f[0]; rec[3] = d1a; rec[4] = d2a; rec[5] = (d2b - d2a) / (3. * h);
        }
        out[7 * q] = t[l];
        for (int i = 0; i < 6; i++) out[7 * q + 1 + i] = rec[i];
    }
}

__device__ inline double bire_spline_lds(const BireSplines* sp, int base, int n, double x)
{
    int lo = 0, hi = n;  // last knot index with t[l] <= x (searchsorted right - 1)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sp->knot[base + mid] <= x) lo = mid + 1; else hi = mid;
    }
    int l = lo - 1;
    l = l < 3 ? 3 : (l > n - 5 ? n - 5 : l);
    const double* r = sp->piece[base + l];
    const double u = x - r[0], h = r[1];
    return r[2] + u * (r[3] + (u - h) * (r[4] + (u - 2. * h) * r[5]));
}

__global__ void __launch_bounds__(256)
bire_steps_kernel(BireBatch b, double* __restrict__ steps /* [total_steps][5] */, long long* __restrict__ log_norm)
{
    __shared__ BireSplines s_sp;
    const int ray = blockIdx.x;
    const int acc = b.n_points[ray];
    if (acc < 2) return;
    // blocks beyond the ray's last step have nothing to do (the grid is sized for the longest ray of the batch)
    if ((int)(blockIdx.y * (blockDim.x >> 6)) * 63 >= acc - 1) return;
    const int nk_all = b.n_knots[0] + b.n_knots[1] + b.n_knots[2];
    const bool lds_splines = b.spline_pieces && nk_all <= BIRE_MAX_KNOTS;
    if (lds_splines)
        for (int q = threadIdx.x; q < nk_all * 7; q += blockDim.x) {
            const double v = b.spline_pieces[q];
            if (q % 7 == 0) s_sp.knot[q / 7] = v; else s_sp.piece[q / 7][q % 7 - 1] = v;
        }
    __syncthreads();
    const double* A0 = b.x1 + 3 * (long)ray;
    const double* B0 = b.x2 + 3 * (long)ray;
    double A[3] = {A0[0], A0[1], A0[2]}, B[3] = {B0[0], B0[1], B0[2]};
    if (B[2] < A[2])  // set_start_and_end_point (:2057-2090): the path runs from the lower to the higher point
        for (int d = 0; d < 3; d++) { double t = A[d]; A[d] = B[d]; B[d] = t; }
    const double dX[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
    const double rho = sqrt(dX[0] * dX[0] + dX[1] * dX[1]);
    double cph = 1., sph = 0.;
    if (rho > 0) { cph = dX[0] / rho; sph = -(dX[1] / rho); }
    const IceConst m = b.ice;
    Pair2D p;
    p.y1 = A[0]; p.z1 = A[2];
    p.y2 = (cph * dX[0] + (-sph) * dX[1] + 0 * dX[2]) + A[0];
    p.z2 = (0 * dX[0] + 0 * dX[1] + 1 * dX[2]) + A[2];
    p.g1 = gamma_of_z(p.z1, m);
    p.g2 = gamma_of_z(p.z2, m);
    const C0State s = make_c0(b.C0[ray], m);
    const double C1 = C1_of(s, p, m);
    const double zstop = z_mirrored(p.y2, p.z2, s, C1, p);
    double sa = 0., ca = 1.;
    if (!isnan(b.angle_to_iceflow)) sincos(b.angle_to_iceflow * (M_PI / 180.), &sa, &ca);
    double* out = steps + 5 * b.step_offset[ray];
    // every lane of a wave runs the same number of iterations (wave reductions inside).  A wave takes 63 consecutive steps: lane
    // j evaluates path point i = first + j once, the step's other end comes from lane j + 1 (lane 63 only supplies a point)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int per_iter = gridDim.y * waves * 63;
    const int n_iter = (acc - 1 + per_iter - 1) / per_iter;
    for (int it = 0; it < n_iter; it++) {
        const int i = ((it * gridDim.y + blockIdx.y) * waves + wave) * 63 + lane;   // it = 0 covers the rays of up to gridDim.y * 252 steps
        double lg = 0.;
        double P0[3] = {0., 0., 0.}, P1[3];
        if (i < acc) bire_path_point(i, acc, p.z1, zstop, s, C1, m, p.y1, p.z1, A[0], A[1], cph, sph, ca, sa, P0);
        for (int d3 = 0; d3 < 3; d3++) P1[d3] = __shfl_down(P0[d3], 1);
        if (lane < 63 && i < acc - 1) {
        const double z = P0[2];
        const double n_nominal = (z <= 0) ? m.n_ice - m.delta_n * exp(z / m.z_0) : 1.;
        double nx, ny, nz;
        if (lds_splines) {
            nx = n_nominal + bire_spline_lds(&s_sp, 0, b.n_knots[0], -z) - b.n_ref;
            ny = n_nominal + bire_spline_lds(&s_sp, b.n_knots[0], b.n_knots[1], -z) - b.n_ref;
            nz = n_nominal + bire_spline_lds(&s_sp, b.n_knots[0] + b.n_knots[1], b.n_knots[2], -z) - b.n_ref;
        } else {
            nx = n_nominal + bire_spline(b.knots, b.coeffs, b.n_knots[0], -z) - b.n_ref;
            ny = n_nominal + bire_spline(b.knots + b.n_knots[0], b.coeffs + b.n_knots[0], b.n_knots[1], -z) - b.n_ref;
            nz = n_nominal + bire_spline(b.knots + b.n_knots[0] + b.n_knots[1], b.coeffs + b.n_knots[0] + b.n_knots[1],
                                         b.n_knots[2], -z) - b.n_ref;
        }
        double dir[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
        const double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] /= len; dir[1] /= len; dir[2] /= len;
        // effective indices (:2188-2207)
        const double sx = dir[0], sy = dir[1], sz = dir[2];
        const double nx2 = nx * nx, ny2 = ny * ny, nz2 = nz * nz;
        const double Aq = ny2 * nz2 * (-1 + sx * sx) + nx2 * (nz2 * (-1 + sy * sy) + ny2 * (-1 + sz * sz));
        const double Bq = sqrt(4 * nx2 * ny2 * nz2 * (nz2 * (-1 + sx * sx + sy * sy) + ny2 * (-1 + sx * sx + sz * sz) +
                                                      nx2 * (-1 + sy * sy + sz * sz)) + Aq * Aq);
        const double num = -2 * nx2 * ny2 * nz2;
        const double N1 = sqrt(num / (Aq - Bq)), N2 = sqrt(num / (Aq + Bq));
        // eigen-polarisations, theta / phi components (:2243-2337)
        const double narrow = 1e-9, wide = 1e-10;
        const bool c1 = fabs(N1 - nx) <= narrow || fabs(N1 - ny) <= narrow || fabs(N1 - nz) <= narrow;
        const bool c2 = fabs(N2 - nx) <= narrow || fabs(N2 - ny) <= narrow || fabs(N2 - nz) <= narrow;
        double a, bb, c, d;
        bool general = true;
        if (c1 || c2) {
            general = false;
            if (c1 && c2) { a = bb = c = d = 0.; }
            else if (fabs(N1 - nx) <= wide) { a = 0; bb = (dir[0] < 0) ? 1. : -1.; c = 1.; d = 0.; }
            else if (fabs(N1 - ny) <= narrow) { a = 0; bb = (dir[1] < 0) ? 1. : -1.; c = 1.; d = 0.; }
            else if (fabs(N2 - ny) <= narrow) { a = 1.; bb = 0.; c = 0.; d = (dir[1] < 0) ? -1. : 1.; }
            else if (fabs(N2 - nz) <= wide) { a = 0.; bb = -1.; c = (dir[2] < 0) ? -1. : 1.; d = 0.; }
            else general = true;
        }
        if (general) {
            double p1[3], p2[3];
            bire_simple(N1, dir, nx, ny, nz, p1);
            bire_simple(N2, dir, nx, ny, nz, p2);
            bire_on_sky(dir, p1, &a, &bb);
            bire_on_sky(dir, p2, &c, &d);
        }
        // np.isclose(a d - b c, 0) or NaN: the step is skipped (:2431-2433) -- marked by NaN in the delay
        double delay = len * N2 / 0.299792458 - len * N1 / 0.299792458;
        const double det = a * d - bb * c;
        if (fabs(det) <= 1e-8 || isnan(a) || isnan(bb) || isnan(c) || isnan(d)) delay = NAN;
        out[5 * (long)i + 0] = a;
        out[5 * (long)i + 1] = bb;
        out[5 * (long)i + 2] = c;
        out[5 * (long)i + 3] = d;
        out[5 * (long)i + 4] = delay;
        if (!isnan(delay)) {  // ||R^T diag(1, phase) R||_2 <= sigma_max(R)^2 = largest eigenvalue of R^T R
            const double T = a * a + bb * bb + c * c + d * d;
            lg = log(0.5 * (T + sqrt(fmax(T * T - 4. * det * det, 0.))));
        }
        }
        if (log_norm) {  // log of the product over the steps: an upper bound on the gain of the whole path
            // summed as integers (units of 2^-40, every term rounded up): integer addition is associative, so the bound -- and
            // with it every pruning decision -- does not depend on the order in which the waves arrive
            long long q = (long long)ceil(lg * BIRE_LOG_FIXED);
            for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
            if ((threadIdx.x & 63) == 0 && q != 0) atomicAdd((unsigned long long*)&log_norm[ray], (unsigned long long)q);
        }
    }
}

#define BIRE_TILE 256

// sin / cos of a small angle (|x| < 0.25: the delay of a 1 m step is ~1e-2 ns, so x = 2 pi f dt stays below 0.1 rad up to
// several GHz): Taylor polynomials, truncation < 2e-18; anything larger goes through sincos()
__device__ inline void bire_sincos(double x, double* sn, double* cs)
{
    if (fabs(x) < 0.25) {
        const double x2 = x * x;
        double ps = -1. / 39916800.;                 // -1/11!
        ps = fma(ps, x2, 1. / 362880.);
        ps = fma(ps, x2, -1. / 5040.);
        ps = fma(ps, x2, 1. / 120.);
        ps = fma(ps, x2, -1. / 6.);
        ps = fma(ps, x2, 1.);
        double pc = 1. / 479001600.;                 // 1/12!
        pc = fma(pc, x2, -1. / 3628800.);
        pc = fma(pc, x2, 1. / 40320.);
        pc = fma(pc, x2, -1. / 720.);
        pc = fma(pc, x2, 1. / 24.);
        pc = fma(pc, x2, -0.5);
        pc = fma(pc, x2, 1.);
        *sn = ps * x;
        *cs = pc;
    } else {
        sincos(x, sn, cs);
    }
}

// the same without the library call (which costs ~100 registers next to the lanes' spectra): larger angles are halved m times and
// brought back by m angle doublings, error ~2^m ulp -- the angles here are (lane or lane count) x 2 pi delay df, i.e. |x| < 0.25
// unless a single step delays by more than ~0.1 ns, and m <= 10 up to 250 rad
__device__ inline void bire_sincos_small(double x, double* sn, double* cs)
{
    int m = 0;
    while (fabs(x) >= 0.25 && m < 40) { x *= 0.5; m++; }
    const double x2 = x * x;
    double ps = -1. / 39916800.;
    ps = fma(ps, x2, 1. / 362880.);
    ps = fma(ps, x2, -1. / 5040.);
    ps = fma(ps, x2, 1. / 120.);
    ps = fma(ps, x2, -1. / 6.);
    ps = fma(ps, x2, 1.);
    double pc = 1. / 479001600.;
    pc = fma(pc, x2, -1. / 3628800.);
    pc = fma(pc, x2, 1. / 40320.);
    pc = fma(pc, x2, -1. / 720.);
    pc = fma(pc, x2, 1. / 24.);
    pc = fma(pc, x2, -0.5);
    pc = fma(pc, x2, 1.);
    double s = ps * x, c = pc;
    for (; m > 0; m--) {
        const double s2 = 2. * s * c, c2 = (c - s) * (c + s);
        s = s2;
        c = c2;
    }
    *sn = s;
    *cs = c;
}

#ifndef BIRE_BINS
#define BIRE_BINS 9   // frequency bins per lane: n_f = 2^p + 1 bins fill ceil(n_f / 9) lanes = 89 % of whole waves
#endif

// spectra [n_rays][2][n_f] complex, in place.  A lane owns the bins k = tid + j T (T lanes per ray, j < BIRE_BINS): the phase of a
// step is linear in the bin number, so per step a lane takes exp(i tid theta) once (polynomial) and walks its bins by multiplying
// with exp(i T theta), which the threads that stage the tile of step records into LDS compute once per step.
#ifndef BIRE_WAVES
#define BIRE_WAVES 2
#endif
__global__ void __launch_bounds__(512, BIRE_WAVES)
bire_propagate_kernel(BireBatch b, const double* __restrict__ steps, double2* __restrict__ spec, const int* __restrict__ ray_active,
                      int T)
{
    __shared__ __align__(16) double s_step[BIRE_TILE][8];   // a, b, c, d, theta, cos(T theta), sin(T theta), kind
    const int ray = blockIdx.x;
    if (ray_active && !ray_active[ray]) return;  // the ray's event cannot pass the candidate cut (general_bound_kernel)
    const int n_steps = b.n_points[ray] - 1;
    const int tid = blockIdx.y * blockDim.x + threadIdx.x;
    const int n_f = b.n_f;
    const int N = 2 * (n_f - 1);
    const double fs = b.sampling_rate;
    const double df = 1.0 / (N * (1. / fs));  // np.fft.rfftfreq
    double2* st = spec + (long)ray * 2 * n_f;
    double2 et[BIRE_BINS], ep[BIRE_BINS];
#pragma unroll
    for (int j = 0; j < BIRE_BINS; j++) {
        const int k = tid + j * T;
        const bool on = tid < T && k < n_f;
        et[j] = on ? st[k] : make_double2(0., 0.);
        ep[j] = on ? st[n_f + k] : make_double2(0., 0.);
    }
    const double* S = steps + 5 * b.step_offset[ray];
    for (int base = 0; base < n_steps; base += BIRE_TILE) {
        const int cnt = min(BIRE_TILE, n_steps - base);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
            const double* R = S + 5 * (long)(base + q);
            const double delay = R[4];
            // BaseTrace.apply_time_shift (base_trace.py:246-276): whole samples are rolled in the time domain (which drops
            // the imaginary parts of the DC and Nyquist bins), anything else is a phase ramp
            double kind = 0., D = delay;
            if (isnan(delay)) { kind = 3.; D = 0.; }   // skipped step
            else {
                const double x = delay * fs;
                if (fabs(rint(x) - x) < 1e-5) {
                    const double kk = rint(x);
                    D = kk / fs;
                    kind = ((long)kk % 2 == 0) ? 1. : 2.;
                }
            }
            const double theta = -2. * M_PI * D * df;
            double sn, cs;
            bire_sincos_small(theta * T, &sn, &cs);
            s_step[q][0] = R[0]; s_step[q][1] = R[1]; s_step[q][2] = R[2]; s_step[q][3] = R[3];
            s_step[q][4] = theta; s_step[q][5] = cs; s_step[q][6] = sn; s_step[q][7] = kind;
        }
        __syncthreads();
        if (tid >= T) continue;
        for (int i = 0; i < cnt; i++) {
            const double kind = s_step[i][7];
            if (kind == 3.) continue;
            const double a = s_step[i][0], bb = s_step[i][1], c = s_step[i][2], d = s_step[i][3];
            const double2 wT = make_double2(s_step[i][5], s_step[i][6]);
            double2 w;
            bire_sincos_small(s_step[i][4] * tid, &w.y, &w.x);
            if (kind == 0.) {
#pragma unroll
                for (int j = 0; j < BIRE_BINS; j++) {
                    const double2 b0 = make_double2(a * et[j].x + bb * ep[j].x, a * et[j].y + bb * ep[j].y);
                    double2 b1 = make_double2(c * et[j].x + d * ep[j].x, c * et[j].y + d * ep[j].y);
                    b1 = make_double2(b1.x * w.x - b1.y * w.y, b1.x * w.y + b1.y * w.x);
                    et[j] = make_double2(a * b0.x + c * b1.x, a * b0.y + c * b1.y);   // R^T
                    ep[j] = make_double2(bb * b0.x + d * b1.x, bb * b0.y + d * b1.y);
                    w = make_double2(w.x * wT.x - w.y * wT.y, w.x * wT.y + w.y * wT.x);
                }
            } else {   // a roll by whole samples: real factors +-1 at DC and Nyquist, imaginary parts dropped there
#pragma unroll
                for (int j = 0; j < BIRE_BINS; j++) {
                    const int k = tid + j * T;
                    const double2 b0 = make_double2(a * et[j].x + bb * ep[j].x, a * et[j].y + bb * ep[j].y);
                    double2 b1 = make_double2(c * et[j].x + d * ep[j].x, c * et[j].y + d * ep[j].y);
                    double2 ww = w;
                    if (k == 0 || k == n_f - 1) {
                        b1.y = 0.;
                        ww = make_double2((k == 0 || kind == 1.) ? 1. : -1., 0.);
                    }
                    b1 = make_double2(b1.x * ww.x - b1.y * ww.y, b1.x * ww.y + b1.y * ww.x);
                    et[j] = make_double2(a * b0.x + c * b1.x, a * b0.y + c * b1.y);
                    ep[j] = make_double2(bb * b0.x + d * b1.x, bb * b0.y + d * b1.y);
                    w = make_double2(w.x * wT.x - w.y * wT.y, w.x * wT.y + w.y * wT.x);
                }
            }
        }
    }
    if (tid < T) {
#pragma unroll
        for (int j = 0; j < BIRE_BINS; j++) {
            const int k = tid + j * T;
            if (k < n_f) { st[k] = et[j]; st[n_f + k] = ep[j]; }
        }
    }
    if (b.counters && tid == 0 && n_steps > 0) atomicAdd(&b.counters[1], (unsigned long long)n_steps * (unsigned long long)n_f);
}

void launch_birefringence_steps(hipStream_t s, const BireBatch& b, int max_points, double* steps, long long* log_norm)
{
    if (b.n_rays <= 0 || max_points < 2) return;
    int gy = (max_points - 1 + 251) / 252;   // 4 waves x 63 steps per block and pass
    if (gy > 64) gy = 64;
    if (log_norm) (void)hipMemsetAsync(log_norm, 0, sizeof(long long) * (size_t)b.n_rays, s);
    if (b.spline_pieces && b.n_knots[0] + b.n_knots[1] + b.n_knots[2] <= BIRE_MAX_KNOTS)
        hipLaunchKernelGGL(bire_pieces_kernel, dim3(1), dim3(128), 0, s, b);
    hipLaunchKernelGGL(bire_steps_kernel, dim3((unsigned)b.n_rays, (unsigned)gy), dim3(256), 0, s, b, steps, log_norm);
}
void launch_birefringence_propagate(hipStream_t s, const BireBatch& b, const double* steps, double2* spec, const int* active)
{
    if (b.n_rays <= 0) return;
    const int T = (b.n_f + BIRE_BINS - 1) / BIRE_BINS;            // lanes per ray
    const int gy = (T + 511) / 512;
    const int block = (((T + gy - 1) / gy) + 63) / 64 * 64;
    hipLaunchKernelGGL(bire_propagate_kernel, dim3((unsigned)b.n_rays, (unsigned)gy), dim3(block), 0, s, b, steps, spec, active, T);
}
void launch_birefringence(hipStream_t s, const BireBatch& b, int max_points, double* steps, double2* spec)
{
    if (b.n_rays <= 0 || max_points < 2) return;
    launch_birefringence_steps(s, b, max_points, steps, nullptr);
    launch_birefringence_propagate(s, b, steps, spec, nullptr);
}

}  // namespace nrhip
