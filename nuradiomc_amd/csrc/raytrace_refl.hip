// raytrace_refl.hip -- ray tracing with reflections off the bottom of an ice shelf (medium.reflection, e.g. Moore's Bay).
//
// ray_tracing.find_solutions (analyticraytracing.py:2118-2130) makes 1 + 2 n_reflections calls of the 2-D solution finder
// per pair: the plain one, then for every number of bottom reflections one for rays that start upwards (reflection_case 1)
// and one for rays that start downwards (2).  The calls are independent, so here every (pair, call) gets its own LANE
// (`find_kernel`: the same hybrid-Powell + two Brent searches as raytrace.hip, on the objective with the reflection
// loop), and a second kernel with one lane per (pair, solution slot) gathers the roots in call order and derives the
// per-solution records: the path is cut into segments between two bottom reflections (get_path_segments :1091-1159);
// length and time are sums over the segments of the closed forms, launch / receive angles come from the first / last
// segment, every segment contributes (z1, z2 mirrored, z_turn) integration limits for the attenuation kernels.
//
// The objective works on a private copy of the start point, as the reference's C++ twin does (cpp:405-470), which wrote
// the golden table reference_C0_MooresBay.pkl; the Python text shifts the shared array in place (:226-229) and thereby
// loses nearly all reflection_case = 2 roots (tests/golden/gen/gen_mooresbay.py).
#include "ray_device.h"
#include "nrhip_internal.h"
#include "root_device.h"

namespace nrhip {

// get_y_with_z_mirror(z, C0, C1) (:160-184)
__device__ inline double y_mirror_c1(double z, const C0State& s, const IceConst& m, double C1)
{
    double y_turn = s.y_turn0 + C1;
    if (z < s.z_turn) return y_of_gamma(gamma_of_z(z, m), s, m) + C1;
    return 2 * y_turn - (y_of_gamma(gamma_of_z(2 * s.z_turn - z, m), s, m) + C1);
}

// get_C_1(x1, C0) (:487)
__device__ inline double C1_at(double y, double z, const C0State& s, const IceConst& m) { return y - y_mirror_c1(z, s, m, 0.0); }

// get_delta_y (:204-272) with `refl` reflections off the layer at z_refl
__device__ __noinline__ double delta_y_refl(double logC0, const Pair2D& p, const IceConst& m, int refl, int rcase, double z_refl)
{
    double C0 = det_exp(logC0) + m.inv_n;
    if (C0 < m.inv_n) return -INFINITY;
    C0State s = make_c0(C0, m);
    double y1 = p.y1, z1 = p.z1;
    if (refl > 0 && rcase == 2) {  // starts downwards: continue the path to the left of x1, to where it passes z1 going up
        double y_turn = s.y_turn0 + C1_at(y1, z1, s, m);
        double dy = y_turn - y1;
        y1 = y1 - 2.0 * dy;
    }
    for (int i = 0; i < refl; i++) {  // restart from the point where the ray meets the bottom again (:281-291)
        double C1 = C1_at(y1, z1, s, m);
        y1 = y_mirror_c1(-z_refl + 2 * s.z_turn, s, m, C1);
        z1 = z_refl;
    }
    double C1 = C1_at(y1, z1, s, m);
    double y_turn = s.y_turn0 + C1;
    if (s.z_turn < p.z2) {
        double dz = s.z_turn - p.z2, dy = y_turn - p.y2;
        return -(sqrt(dz * dz + dy * dy) + 10 * fabs(dz));
    }
    double y2 = y_of_gamma(p.g2, s, m) + C1;
    if (y_turn > p.y2) return p.y2 - y2;
    return -1 * (p.y2 - (2 * y_turn - y2));
}

struct PairGeom {
    Pair2D p;
    double cph, sph;
    bool swap;
};

// set_start_and_end_point (:2057-2090)
__device__ inline PairGeom pair_geometry(const double* __restrict__ x1, const double* __restrict__ x2, const IceConst& m)
{
    PairGeom g;
    double A[3] = {x1[0], x1[1], x1[2]}, B[3] = {x2[0], x2[1], x2[2]};
    g.swap = B[2] < A[2];
    if (g.swap)
        for (int d = 0; d < 3; d++) { double t = A[d]; A[d] = B[d]; B[d] = t; }
    double dX[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
    double rho = sqrt(dX[0] * dX[0] + dX[1] * dX[1]);
    g.cph = 1.;
    g.sph = 0.;
    if (rho > 0) {
        g.cph = dX[0] / rho;
        g.sph = -(dX[1] / rho);
    }
    g.p.y1 = A[0];
    g.p.z1 = A[2];
    g.p.y2 = (g.cph * dX[0] + (-g.sph) * dX[1] + 0 * dX[2]) + A[0];
    g.p.z2 = (0 * dX[0] + 0 * dX[1] + 1 * dX[2]) + A[2];
    g.p.g1 = gamma_of_z(g.p.z1, m);
    g.p.g2 = gamma_of_z(g.p.z2, m);
    return g;
}

// call c of a pair: c = 0 plain; c = 2 r - 1: (r reflections, case 1); c = 2 r: (r, case 2)
__device__ inline void call_label(int c, int* refl, int* rcase)
{
    *refl = (c + 1) / 2;
    *rcase = (c == 0) ? 1 : 2 - (c & 1);
}

// one lane per (pair, call): up to 3 roots, sorted by C0
__global__ void __launch_bounds__(256)
find_refl_kernel(long n_pairs, int n_calls, const double* __restrict__ x1, const double* __restrict__ x2, int n_x2, IceConst m,
                 double z_refl, int* __restrict__ cand_n, double* __restrict__ cand_C0, int strict)
{
    const long n_items = n_pairs * n_calls;
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n_items; it += (long)gridDim.x * blockDim.x) {
        // lanes of a wave share the call (same branch structure), consecutive lanes are consecutive pairs
        const long i = it % n_pairs;
        const int c = (int)(it / n_pairs);
        const long i1 = (n_x2 > 0) ? i / n_x2 : i, i2 = (n_x2 > 0) ? i % n_x2 : i;
        PairGeom g = pair_geometry(x1 + 3 * i1, x2 + 3 * i2, m);
        const Pair2D& p = g.p;
        int refl, rcase;
        call_label(c, &refl, &rcase);
        int ns = 0;
        double lc[3];
        if (!(p.z2 > 0)) {
            auto dy = [&](double l) { return delta_y_refl(l, p, m, refl, rcase, z_refl); };
            auto dy2 = [&](double l) { double d = delta_y_refl(l, p, m, refl, rcase, z_refl); return d * d; };
            double fun;
            double xr = hybrd1(dy2, -1., 1e-6, &fun);
            const double d_hi = dy(xr + 0.0001), d_lo = dy(xr - 0.0001);
            if (fun < 1e-7) lc[ns++] = xr;
            else if (!strict && refl == 0 && d_lo != 0 && d_hi != 0 && !isnan(d_lo) && !isnan(d_hi) && signbit(d_lo) != signbit(d_hi))
                lc[ns++] = brentq(dy, xr - 0.0001, xr + 0.0001, d_lo, d_hi);   // the true set for the plain call (raytrace.hip)
            {
                double a = xr + 0.0001, b = 100.;
                double fa = d_hi, fb = dy(b);
                if (brent_bracket_ok(fa, fb)) lc[ns++] = brentq(dy, a, b, fa, fb);
            }
            {
                double a = -100., b = xr - 0.0001;
                double fa = dy(a), fb = d_lo;
                if (brent_bracket_ok(fa, fb)) lc[ns++] = brentq(dy, a, b, fa, fb);
            }
        }
        double c0[3];
        for (int k = 0; k < ns; k++) c0[k] = det_exp(lc[k]) + m.inv_n;
        for (int a = 1; a < ns; a++)
            for (int b = a; b > 0 && c0[b] < c0[b - 1]; b--) { double t = c0[b]; c0[b] = c0[b - 1]; c0[b - 1] = t; }
        const long o = i * n_calls + c;
        cand_n[o] = ns;
        for (int k = 0; k < 3; k++) cand_C0[3 * o + k] = (k < ns) ? c0[k] : NAN;
    }
}

struct Seg { double y1, z1, y2, z2; };

// get_path_segments (:1091-1159) of the path (ys, zs) -> (ye, ze)
__device__ inline int path_segments(double ys, double zs, double ye, double ze, const C0State& s, const IceConst& m, int refl,
                                    int rcase, double z_refl, Seg* segs)
{
    if (refl == 0) {
        segs[0] = Seg{ys, zs, ye, ze};
        return 1;
    }
    double y1 = ys, z1 = zs;
    if (rcase == 2) {
        double y_turn = s.y_turn0 + C1_at(y1, z1, s, m);
        double dy = y_turn - y1;
        y1 = y1 - 2 * dy;
    }
    int n = 0;
    for (int i = 0; i < refl + 1; i++) {
        double C1 = C1_at(y1, z1, s, m);
        double y2 = y_mirror_c1(-z_refl + 2 * s.z_turn, s, m, C1), z2 = z_refl;
        bool stop = false;
        if (y2 > ye) {
            stop = true;
            y2 = ye;
            z2 = ze;
        }
        segs[n++] = Seg{y1, z1, y2, z2};
        if (stop) break;
        y1 = y2;
        z1 = z2;
    }
    return n;
}

__device__ inline Pair2D seg_pair(double y1, double z1, double y2, double z2, const IceConst& m)
{
    Pair2D q;
    q.y1 = y1; q.z1 = z1; q.y2 = y2; q.z2 = z2;
    q.g1 = gamma_of_z(z1, m);
    q.g2 = gamma_of_z(z2, m);
    return q;
}

// get_angle(x, x_start, C0, reflection, reflection_case) (:1161-1193): the start of the last segment replaces x_start
__device__ inline void ray_sincos_refl(double y, double z, double ys, double zs, const C0State& s, const IceConst& m, int refl,
                                       int rcase, double z_refl, double* sn, double* cs)
{
    Seg segs[NRHIP_MAX_REFLECTIONS + 1];
    int n = path_segments(ys, zs, y, z, s, m, refl, rcase, z_refl, segs);
    Pair2D q = seg_pair(segs[n - 1].y1, segs[n - 1].z1, y, z, m);
    ray_sincos(y, z, s, C1_of(s, q, m), q, m, sn, cs);
}

// one lane per (pair, slot)
__global__ void __launch_bounds__(256)
records_refl_kernel(long n_pairs, int n_calls, int stride, const double* __restrict__ x1, const double* __restrict__ x2, int n_x2,
                    IceConst m, double z_refl, const int* __restrict__ cand_n, const double* __restrict__ cand_C0, int given,
                    ReflRecords out)
{
    const long n_items = n_pairs * stride;
    const int n_seg_max = (n_calls - 1) / 2 + 1;
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n_items; it += (long)gridDim.x * blockDim.x) {
        const long i = it / stride;
        const int slot = (int)(it % stride);
        const long k = it;
        double c0 = NAN;
        int refl = 0, rcase = 0, ns = 0;
        if (given) {
            ns = out.n_sol[i];
            if (slot < ns) { c0 = out.C0[k]; refl = out.reflection[k]; rcase = out.reflection_case[k]; }
        } else {
            // the solutions of the calls in call order (each sorted by C0); more than `stride` -> none (:2127-2130)
            for (int c = 0; c < n_calls; c++) {
                int nc = cand_n[i * n_calls + c];
                if (slot >= ns && slot < ns + nc) {
                    c0 = cand_C0[3 * (i * n_calls + c) + (slot - ns)];
                    call_label(c, &refl, &rcase);
                }
                ns += nc;
            }
            if (ns > stride) { ns = 0; c0 = NAN; }
            if (slot == 0) out.n_sol[i] = ns;
        }
        for (int j = 0; j < n_seg_max; j++) {
            out.seg_C0[k * n_seg_max + j] = NAN;
            for (int d = 0; d < 3; d++) out.seg_zint[((k * n_seg_max) + j) * 3 + d] = NAN;
        }
        if (slot >= ns || isnan(c0)) {
            out.type[k] = 0;
            out.reflection[k] = out.reflection_case[k] = 0;
            out.n_segments[k] = out.surface_mask[k] = 0;
            out.C0[k] = out.C1[k] = out.D[k] = out.T[k] = out.refl_angle[k] = NAN;
            for (int d = 0; d < 3; d++) out.launch[3 * k + d] = out.receive[3 * k + d] = NAN;
            continue;
        }
        const long i1 = (n_x2 > 0) ? i / n_x2 : i, i2 = (n_x2 > 0) ? i % n_x2 : i;
        PairGeom g = pair_geometry(x1 + 3 * i1, x2 + 3 * i2, m);
        const Pair2D& p = g.p;
        C0State st = make_c0(c0, m);
        const double C1 = C1_of(st, p, m);
        out.type[k] = solution_type(st, C1, p);
        out.C0[k] = c0;
        out.C1[k] = C1;
        out.reflection[k] = refl;
        out.reflection_case[k] = rcase;
        // segments: length, time, attenuation limits, surface reflections
        Seg segs[NRHIP_MAX_REFLECTIONS + 1];
        const int nseg = path_segments(p.y1, p.z1, p.y2, p.z2, st, m, refl, rcase, z_refl, segs);
        const double c_light = 0.299792458;
        double D = 0, cT = 0, refl_angle = NAN;
        int surface_mask = 0;
        for (int j = 0; j < nseg; j++) {
            // a first segment that starts downwards is integrated as its mirror image (:629-636, :720-727, :943-950)
            Pair2D q = (j == 0 && rcase == 2) ? seg_pair(p.y1, segs[j].z2, segs[j].y2, p.z1, m)
                                              : seg_pair(segs[j].y1, segs[j].z1, segs[j].y2, segs[j].z2, m);
            const double C1q = C1_of(st, q, m);
            const int tq = solution_type(st, C1q, q);
            double sL, cL, Dj, Tj;
            ray_sincos(q.y1, q.z1, st, C1q, q, m, &sL, &cL);
            path_length_time(st, C1q, tq, sL, q, m, &Dj, &Tj);
            D += Dj;
            cT += Tj * c_light;
            double* zi = out.seg_zint + ((k * n_seg_max) + j) * 3;
            zi[0] = q.z1;
            zi[1] = z_mirrored(q.y2, q.z2, st, C1q, q);
            zi[2] = st.z_turn;
            out.seg_C0[k * n_seg_max + j] = c0;
            // reflection at the surface inside this segment (:1201-1237): the angle is the same in every segment
            const double C1s = C1_at(segs[j].y1, segs[j].z1, st, m);
            const double y_turn = st.y_turn0 + C1s;
            if (st.z_turn >= 0 && y_turn > p.y1 && y_turn < p.y2) {
                Pair2D qs = seg_pair(segs[j].y1, segs[j].z1, y_turn, 0., m);
                double sr, cr;
                ray_sincos(y_turn, 0., st, C1s, qs, m, &sr, &cr);
                refl_angle = atan2(sr, cr);
                surface_mask |= 1 << j;
            }
        }
        out.D[k] = D;
        out.T[k] = cT / c_light;
        out.refl_angle[k] = refl_angle;
        out.n_segments[k] = nseg;
        out.surface_mask[k] = surface_mask;
        double sL, cL, s2, c2;
        ray_sincos_refl(p.y1, p.z1, p.y1, p.z1, st, m, refl, rcase, z_refl, &sL, &cL);
        ray_sincos_refl(p.y2, p.z2, p.y1, p.z1, st, m, refl, rcase, z_refl, &s2, &c2);
        double lv0 = sL, lv2 = cL, rv0 = -s2, rv2 = -c2;
        if (g.swap) {
            lv0 = -s2; lv2 = -c2;
            rv0 = sL;  rv2 = cL;
        }
        out.launch[3 * k + 0] = g.cph * lv0;
        out.launch[3 * k + 1] = -g.sph * lv0;
        out.launch[3 * k + 2] = lv2;
        out.receive[3 * k + 0] = g.cph * rv0;
        out.receive[3 * k + 1] = -g.sph * rv0;
        out.receive[3 * k + 2] = rv2;
    }
}

// att[ray][f] = product over the ray's path segments (get_attenuation_along_path :933-1089)
__global__ void __launch_bounds__(256)
segment_product_kernel(long n_rays, int n_seg_max, int n_freq, const double* __restrict__ seg_zint,
                       const double* __restrict__ seg_att, double* __restrict__ att)
{
    long it = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (it >= n_rays * n_freq) return;
    const long r = it / n_freq;
    const int f = (int)(it % n_freq);
    double a = NAN;
    for (int j = 0; j < n_seg_max; j++) {
        if (isnan(seg_zint[(r * n_seg_max + j) * 3])) continue;
        double v = seg_att[(r * n_seg_max + j) * n_freq + f];
        a = isnan(a) ? v : a * v;
    }
    att[it] = a;
}

void launch_find_refl(hipStream_t stream, long n_pairs, int n_reflections, const double* x1, const double* x2, int n_x2,
                      const IceConst& m, double z_refl, int* cand_n, double* cand_C0, bool reference_procedure)
{
    const int n_calls = 1 + 2 * n_reflections;
    long n_items = n_pairs * n_calls;
    if (n_items <= 0) return;
    long grid = (n_items + 255) / 256;
    if (grid > 256L * 64) grid = 256L * 64;
    hipLaunchKernelGGL(find_refl_kernel, dim3((unsigned)grid), dim3(256), 0, stream, n_pairs, n_calls, x1, x2, n_x2, m, z_refl,
                       cand_n, cand_C0, reference_procedure ? 1 : 0);
}

void launch_records_refl(hipStream_t stream, long n_pairs, int n_reflections, int stride, const double* x1, const double* x2,
                         int n_x2, const IceConst& m, double z_refl, const int* cand_n, const double* cand_C0, int given,
                         const ReflRecords& out)
{
    const int n_calls = 1 + 2 * n_reflections;
    long n_items = n_pairs * stride;
    if (n_items <= 0) return;
    long grid = (n_items + 255) / 256;
    if (grid > 256L * 64) grid = 256L * 64;
    hipLaunchKernelGGL(records_refl_kernel, dim3((unsigned)grid), dim3(256), 0, stream, n_pairs, n_calls, stride, x1, x2, n_x2, m,
                       z_refl, cand_n, cand_C0, given, out);
}

void launch_segment_product(hipStream_t stream, long n_rays, int n_seg_max, int n_freq, const double* seg_zint,
                            const double* seg_att, double* att)
{
    long n = n_rays * n_freq;
    if (n <= 0) return;
    hipLaunchKernelGGL(segment_product_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n_rays, n_seg_max, n_freq,
                       seg_zint, seg_att, att);
}

}  // namespace nrhip
