// raytrace.hip -- batched find_solutions + per-solution geometry on MI355X (gfx950).
//
// One LANE per (vertex, channel) pair: the reference's solution finder is an inherently serial
// procedure (a MINPACK hybrid-Powell iteration on (delta y)^2 started at logC0 = -1, followed by two
// Brent searches either side of where it stopped -- analyticraytracing.py:1477-1547), and WHICH roots
// are reported depends on where that first iteration stops.  To report the same solutions as the
// reference the kernel runs the same procedure per pair; 64 independent pairs per wavefront keep all
// lanes busy, and consecutive lanes are the channels of one event, so branches are mostly coherent.
// Inputs are read once, coalesced (24 B per lane), outputs are written once ([pair][2] records).
//
// The kernel is FP64-VALU / transcendental bound (about 1e2 objective evaluations x ~1e2 flops per
// pair against ~320 B of HBM traffic), not HBM bound.
#include <cstdlib>
#include "ray_device.h"
#include "nrhip_internal.h"
#include "root_device.h"

namespace nrhip {

// the 2-D geometry of a pair: set_start_and_end_point (:2057-2090, the higher point becomes the stop point) and the rotation by
// dPhi = -atan2(dy, dx) into the y-z plane, cos / sin taken algebraically (bit-reproducible)
struct PairGeom {
    double A0, A2;      // start point (x, z) after the swap
    double y2, z2;      // stop point in the rotated frame
    double cph, sph;
    double dist;        // |x2 - x1|
    bool swap;
};
__device__ __forceinline__ PairGeom pair_geometry(const double* __restrict__ x1, const double* __restrict__ x2, long i1, long i2)
{
    double A[3] = {x1[3 * i1], x1[3 * i1 + 1], x1[3 * i1 + 2]};
    double B[3] = {x2[3 * i2], x2[3 * i2 + 1], x2[3 * i2 + 2]};
    PairGeom g;
    g.swap = B[2] < A[2];
    if (g.swap) {
        for (int d = 0; d < 3; d++) { double t = A[d]; A[d] = B[d]; B[d] = t; }
    }
    const double dX[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
    const double rho = sqrt(dX[0] * dX[0] + dX[1] * dX[1]);
    g.cph = 1.;
    g.sph = 0.;
    if (rho > 0) {
        g.cph = dX[0] / rho;
        g.sph = -(dX[1] / rho);
    }
    g.A0 = A[0];
    g.A2 = A[2];
    g.y2 = (g.cph * dX[0] + (-g.sph) * dX[1] + 0 * dX[2]) + A[0];
    g.z2 = (0 * dX[0] + 0 * dX[1] + 1 * dX[2]) + A[2];
    g.dist = sqrt(dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2]);
    return g;
}

// Kernels: pair i = (event i / n_ch, channel i % n_ch) when n_ch > 0, else x2 is per pair.
//
// Two launches.  `raytrace_roots_kernel` runs the finder (hybrd + two Brent searches, ~1e2 calls of the objective per pair, which
// only sees the pair's six numbers in an LDS column) and writes the sorted launch parameters C0 and their count -- 20 B per pair.
// `raytrace_records_kernel` then makes the per-solution records (type, C1, D, T, angles, vectors) from C0, in the natural order
// of the pairs (coalesced) and with its own register budget.  As ONE kernel (rounds 1-2) the records code shared the finder's 80
// registers: the compiler hoisted ~25 polynomial constants of its atan2 / log / path-length code out of the pair loop, spilled
// them, and re-read them for every solution -- 1.6 KB of scratch reads per pair that miss the L2 (the scratch of all resident
// waves is 26 MB per XCD): 8 GB of fetches and 6 GB of write-backs per 5e6 pairs (`profiles/r03_pmc_traffic.csv`, 14.5 GB).
// det_exp as a call: inlined into the finder kernel its 16 polynomial constants are hoisted out of the pair loop and spilled
__device__ __noinline__ double det_exp_call(double x) { return det_exp(x); }

#ifndef NRHIP_RT_WAVES
#define NRHIP_RT_WAVES 6  // waves per SIMD the register budget is cut for (measured best)
#endif
__global__ void __launch_bounds__(256, NRHIP_RT_WAVES)
raytrace_roots_kernel(long n_pairs, const double* __restrict__ x1, const double* __restrict__ x2, int n_ch,
                      IceConst m_arg, RayRecords out, const double* __restrict__ max_dist, const int* __restrict__ perm,
                      unsigned long long* __restrict__ eval_count, int only_flagged, int strict)
{
    // strict: the reference's acceptance test alone decides about the first root (NRHIP_FINDER_REFERENCE, nrhip_ctx_set_ray_finder):
    // the list is then the reference's, lost roots included
    int n_eval = 0;   // calls of the objective by this lane (the FP64 view of bench.py prices the finder by them)
    __shared__ double sh_pair[6][256];   // the pair geometry the objective reads on every evaluation (see delta_y_lds)
    __shared__ IceConst sh_ice;          // the ice model for the (non-inlined) objective: a reference to the kernel argument would
                                         // be a per-lane copy in scratch (64 B x every resident lane, re-read by every call)
    double* const sp = &sh_pair[0][threadIdx.x];   // (only this lane reads its column: no barrier)
    if (threadIdx.x == 0) sh_ice = m_arg;
    __syncthreads();
    const IceConst& m = sh_ice;
    for (long iw = blockIdx.x * (long)blockDim.x + threadIdx.x; iw < n_pairs; iw += (long)gridDim.x * blockDim.x) {
        // perm (optional): events in an order that puts similar geometries (distance, depth) next to each other, so that
        // the lanes of a wave run similar numbers of root-finder iterations; results land at the original pair index
        long i1 = (n_ch > 0) ? iw / n_ch : iw;
        const long i2 = (n_ch > 0) ? iw % n_ch : iw;
        if (perm) i1 = perm[i1];
        const long i = (n_ch > 0) ? i1 * n_ch + i2 : iw;
        if (only_flagged && out.n_sol[i] != -1) continue;   // (the pairs raytrace_roots_fast_kernel left: receivers deeper than 10 z_0)
        bool search;
        {
            const PairGeom g = pair_geometry(x1, x2, i1, i2);
            sp[0] = g.A0; sp[256] = g.A2; sp[512] = g.y2; sp[768] = g.z2;
            sp[1024] = m.delta_n * det_exp_call(g.A2 / m.z_0); sp[1280] = m.delta_n * det_exp_call(g.z2 / m.z_0);   // gamma_of_z
            // speedup.distance_cut (simulation.py:155-163): showers farther from the antenna than their cut are not traced
            const bool too_far = max_dist && g.dist > max_dist[i1];
            // receiver in air: special branch of the reference (:1437-1460) not provided
            search = !(g.z2 > 0) && !too_far;
        }
        int ns = 0;
        double lc0 = 0., lc1 = 0., lc2 = 0.;
        if (search) {
            auto dy = [&](double l) { n_eval++; return delta_y_lds(l, sp, 256, m); };
            auto dy2 = [&](double l) { n_eval++; double d = delta_y_lds(l, sp, 256, m); return d * d; };
            double fun;
            const double xr = hybrd1(dy2, -1., 1e-6, &fun);
            // The reference keeps the iterate as a root if (delta y)^2 < 1e-7 there (:1483) and then looks for one more root
            // either side of it, 1e-4 away (:1498-1541).  Round 5: where that test fails although delta y changes its sign
            // between the two points 1e-4 either side -- delta y is continuous in log C0, so there IS a root between them, the
            // one the iteration was after -- the root is taken from that bracket (the TRUE solution set: DESIGN section 2,
            // tools/true_roots.py).  The (at most three) brackets of a lane are searched one after the
            // other by ONE call site of Brent's method: a wave spends the longest lane's searches, not the sum over the kinds.
            const double d_hi = dy(xr + 0.0001), d_top = dy(100.), d_bot = dy(-100.), d_lo = dy(xr - 0.0001);
            unsigned todo = 0;
            if (fun < 1e-7) { lc0 = xr; ns = 1; }
            else if (!strict && d_lo != 0 && d_hi != 0 && !isnan(d_lo) && !isnan(d_hi) && signbit(d_lo) != signbit(d_hi)) todo |= 1u;
            if (brent_bracket_ok(d_hi, d_top)) todo |= 2u;
            if (brent_bracket_ok(d_bot, d_lo)) todo |= 4u;
            while (todo) {
                const unsigned k = todo & (0u - todo);   // lowest pending bracket: the order of the reference's list
                todo ^= k;
                const double a = (k == 1u) ? xr - 0.0001 : ((k == 2u) ? xr + 0.0001 : -100.);
                const double b = (k == 1u) ? xr + 0.0001 : ((k == 2u) ? 100. : xr - 0.0001);
                const double fa = (k == 1u) ? d_lo : ((k == 2u) ? d_hi : d_bot);
                const double fb = (k == 1u) ? d_hi : ((k == 2u) ? d_top : d_lo);
                const double r = brentq(dy, a, b, fa, fb);
                if (ns == 0) lc0 = r; else if (ns == 1) lc1 = r; else lc2 = r;
                ns++;
            }
        }
        double c0a = NAN, c0b = NAN, c0c = NAN;
        if (ns > 0) c0a = det_exp_call(lc0) + m.inv_n;
        if (ns > 1) c0b = det_exp_call(lc1) + m.inv_n;
        if (ns > 2) c0c = det_exp_call(lc2) + m.inv_n;
        // sorted by C0 (<= 3 entries)
        if (ns > 1 && c0b < c0a) { double t = c0a; c0a = c0b; c0b = t; }
        if (ns > 2 && c0c < c0b) { double t = c0b; c0b = c0c; c0c = t; }
        if (ns > 1 && c0b < c0a) { double t = c0a; c0a = c0b; c0b = t; }
        if (ns > NRHIP_MAXS) {  // "too many solutions -> none" (:2127-2130)
            ns = 0;
            c0a = c0b = NAN;
        }
        out.n_sol[i] = ns;
        out.C0[i * NRHIP_MAXS] = c0a;
        out.C0[i * NRHIP_MAXS + 1] = c0b;
    }
    if (eval_count) {
        for (int off = 32; off > 0; off >>= 1) n_eval += __shfl_xor(n_eval, off);
        if ((threadIdx.x & 63) == 0 && n_eval) atomicAdd(eval_count, (unsigned long long)n_eval);
    }
}

// ---- the finder without the hybr stage (round 5; DESIGN section 2, "the true solution set") ---------------------------------
// delta_y(log C0) of analyticraytracing.py:204-272 is min(u, v) wherever the ray's turning point lies above the receiver:
//     u = x2.y - y(z2)                the receiver's offset from the ray on its way UP to the turning point,
//     v = (2 y_turn - y(z2)) - x2.y   the same on its way DOWN (mirrored branch, or after the reflection at the surface);
// u rises monotonically with C0 (a steeper launch reaches the receiver's depth earlier), v rises to ONE maximum -- the farthest
// point any ray reaches at that depth -- and falls (both checked on 1e6 random pairs in three ice models, tools/root_shapes.py).
// So the solutions are: the root of u (the direct ray) and the root of v beyond its maximum when the ray that turns AT the
// receiver's depth overshoots the receiver (v > 0 there); else the two roots of v either side of its maximum if that is
// positive; else none.  The reference looks for the same roots with scipy.optimize.root on (delta_y)^2 from log C0 = -1 --
// ~37 evaluations creeping onto a double root, stopped 1e-7 away from it and kept by a coin flip (:1479-1483) -- and two Brent
// searches either side of where that stopped.  Here every root comes out of a bracket, by the same Brent's method, to 2e-12:
// ~21 evaluations per pair instead of ~62, one logarithm fewer per evaluation (the depth of the turning point is never needed:
// the clamp at the surface is a comparison of gammas), and no root is lost to an acceptance test.
// Searches run in t = sqrt(log C0 - x_lo), x_lo = log(1 / n(z2) - 1 / n_ice) the launch parameter of the ray that turns at the
// receiver's depth: u and v start like sqrt(log C0 - x_lo) there.  Pairs whose receiver lies deeper than 10 z_0 (n(z) = n_ice
// to 1e-5: x_lo ill-conditioned; no detector is there) are flagged (n_sol = -1) and left to raytrace_roots_kernel, the
// reference's procedure.
#define NRHIP_T_START 3e-5
#define NRHIP_SHALLOW 4.5399929762484854e-05   // exp(-10)

// (u, v) at t; sp: the pair's LDS column (y1, z1, y2, z2, gamma1, gamma2, x_lo)
__device__ __noinline__ double2 uv_lds(double t, const double* __restrict__ sp, int stride, const IceConst& m)
{
    const double y1 = sp[0], y2p = sp[2 * stride], g1 = sp[4 * stride], g2 = sp[5 * stride], x_lo = sp[6 * stride];
    C0State s;
    const double C0 = det_exp(x_lo + t * t) + m.inv_n;
    s.C0 = C0;
    s.c = m.n2 - 1. / (C0 * C0);
    s.two_sc = 2 * sqrt(s.c);
    s.two_c = 2 * s.c;
    s.pref = m.z_0 / sqrt(m.n2 * C0 * C0 - 1);
    double gt = m.b * 0.5 - sqrt(m.qb2 - s.c);
    if (gt > m.delta_n) gt = m.delta_n;   // turning point above the surface: reflection at z = 0
    const double y_turn0 = y_of_gamma(gt, s, m);
    const double C1 = y1 - y_of_gamma(g1, s, m);
    const double y_turn = y_turn0 + C1;
    const double y2 = y_of_gamma(g2, s, m) + C1;
    return make_double2(y2p - y2, -1 * (y2p - (2 * y_turn - y2)));
}

#ifndef NRHIP_RTF_WAVES
#define NRHIP_RTF_WAVES 6   // waves per SIMD the register budget of the finder without the hybr stage is cut for
#endif
__global__ void __launch_bounds__(256, NRHIP_RTF_WAVES)
raytrace_roots_fast_kernel(long n_pairs, const double* __restrict__ x1, const double* __restrict__ x2, int n_ch,
                           IceConst m_arg, RayRecords out, const double* __restrict__ max_dist, const int* __restrict__ perm,
                           unsigned long long* __restrict__ eval_count, int event_major)
{
    int n_eval = 0;
    __shared__ double sh_pair[7][256];
    __shared__ IceConst sh_ice;
    double* const sp = &sh_pair[0][threadIdx.x];
    if (threadIdx.x == 0) sh_ice = m_arg;
    __syncthreads();
    const IceConst& m = sh_ice;
    for (long iw = blockIdx.x * (long)blockDim.x + threadIdx.x; iw < n_pairs; iw += (long)gridDim.x * blockDim.x) {
        // with the events in geometry-cell order (perm) the pairs are walked CHANNEL-major: the lanes of a wave are neighbouring
        // events seen from the same antenna -- the same kind of search and similar iteration counts (event-major, a wave of a
        // 24-channel station mixes antennas at -3 m and -100 m: 1.7 x the time per evaluation on the 35-station array)
        long i1, i2;
        if (n_ch > 0 && perm && !event_major) {
            const long n_ev = n_pairs / n_ch;
            i2 = iw / n_ev;
            i1 = perm[iw - i2 * n_ev];
        } else {
            i1 = (n_ch > 0) ? iw / n_ch : iw;
            i2 = (n_ch > 0) ? iw % n_ch : iw;
            if (perm) i1 = perm[i1];
        }
        const long i = (n_ch > 0) ? i1 * n_ch + i2 : iw;
        bool search, deep;
        {
            const PairGeom g = pair_geometry(x1, x2, i1, i2);
            const double g2 = m.delta_n * det_exp_call(g.z2 / m.z_0);
            sp[0] = g.A0; sp[256] = g.A2; sp[512] = g.y2; sp[768] = g.z2;
            sp[1024] = m.delta_n * det_exp_call(g.A2 / m.z_0); sp[1280] = g2;
            sp[1536] = det_log(1. / (m.n_ice - g2) - m.inv_n);   // x_lo: the ray that turns at the receiver's depth
            const bool too_far = max_dist && g.dist > max_dist[i1];
            search = !(g.z2 > 0) && !too_far;
            // (and pairs exactly above each other: their solutions are the vertical rays, log C0 -> infinity -- the reference's
            // procedure reports them at its search limits, the brackets here have no sign change to find)
            deep = !(g2 >= NRHIP_SHALLOW * m.delta_n) || !(g.y2 > g.A0);
        }
        if (search && deep) {   // left to the reference's procedure (raytrace_roots_kernel with only_flagged)
            out.n_sol[i] = -1;
            continue;
        }
        int ns = 0;
        double r0 = 0., r1 = 0.;
        if (search) {
            auto fv = [&](double t) { n_eval++; return uv_lds(t, sp, 256, m).y; };
            auto uv = [&](double t) { n_eval++; return uv_lds(t, sp, 256, m); };
            const double x_lo = sp[1536];
            const double ta = NRHIP_T_START, tm = sqrt(2. - x_lo), tt = sqrt(100. - x_lo);
            const double2 A = uv(ta), M = uv(tm);
            double2 T = make_double2(0., 0.);
            // the (at most two) brackets found below are searched afterwards by ONE inlined copy of Brent's method
            int n_br = 0, comp0 = 0, comp1 = 0;
            double a0 = 0., b0 = 0., fa0 = 0., fb0 = 0., a1 = 0., b1 = 0., fa1 = 0., fb1 = 0.;
            auto push = [&](int comp, double xa, double xb, double fa, double fb) {
                if (n_br == 0) { comp0 = comp; a0 = xa; b0 = xb; fa0 = fa; fb0 = fb; }
                else { comp1 = comp; a1 = xa; b1 = xb; fa1 = fa; fb1 = fb; }
                n_br++;
            };
            if (A.y > 0) {
                // the ray turning at the receiver's depth overshoots it: the direct ray (u rises through zero once) ...
                bool have_t = false;
                if (A.x < 0) {
                    if (M.x > 0) push(0, ta, tm, A.x, M.x);
                    else {
                        T = uv(tt); have_t = true;
                        if (T.x > 0) push(0, tm, tt, M.x, T.x);
                    }
                }
                // ... and the one root of v beyond its maximum
                if (M.y < 0) push(1, ta, tm, A.y, M.y);
                else {
                    if (!have_t) T = uv(tt);
                    if (T.y < 0) push(1, tm, tt, M.y, T.y);
                }
            } else {
                // v <= 0 at the lower end: is its maximum positive?  Brent's minimiser on -v over (a, b); a, b are always evaluated
                // points (fa, fb = v there, <= 0 so far); stops at the first v > 0
                const double CG = 0.3819660112501051;
                double a = ta, fa = A.y, b = tm, fb = M.y;
                double x, fx;
                if (M.y > A.y) {
                    T = uv(tt);
                    b = tt; fb = T.y;
                    x = tm; fx = -M.y;
                } else {
                    x = a + CG * (b - a);
                    fx = -fv(x);
                }
                double w = x, vv = x, fw = fx, fvv = fx, d = 0., e = 0.;
                bool found = (fx < 0);
                for (int it = 0; it < 60 && !found; it++) {
                    const double xm = 0.5 * (a + b), tol1 = 1e-6 * fabs(x) + 1e-7, tol2 = 2. * tol1;
                    if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
                    bool golden = true;
                    if (fabs(e) > tol1) {
                        const double rr = (x - w) * (fx - fvv);
                        double q = (x - vv) * (fx - fw);
                        double p = (x - vv) * q - (x - w) * rr;
                        q = 2. * (q - rr);
                        if (q > 0.) p = -p;
                        q = fabs(q);
                        const double etemp = e;
                        e = d;
                        if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
                            d = p / q;
                            const double xn = x + d;
                            if (xn - a < tol2 || b - xn < tol2) d = (xm - x >= 0) ? tol1 : -tol1;
                            golden = false;
                        }
                    }
                    if (golden) {
                        e = (x >= xm) ? a - x : b - x;
                        d = CG * e;
                    }
                    const double xu = (fabs(d) >= tol1) ? x + d : x + ((d >= 0) ? tol1 : -tol1);
                    const double fnew = -fv(xu);
                    if (fnew < 0) {   // v > 0: inside the interval of solutions, a < xu < b
                        if (xu < x) { b = x; fb = -fx; } else { a = x; fa = -fx; }
                        x = xu; fx = fnew;
                        found = true;
                        break;
                    }
                    if (fnew <= fx) {
                        if (xu >= x) { a = x; fa = -fx; } else { b = x; fb = -fx; }
                        vv = w; fvv = fw; w = x; fw = fx; x = xu; fx = fnew;
                    } else {
                        if (xu < x) { a = xu; fa = -fnew; } else { b = xu; fb = -fnew; }
                        if (fnew <= fw || w == x) { vv = w; fvv = fw; w = xu; fw = fnew; }
                        else if (fnew <= fvv || vv == x || vv == w) { vv = xu; fvv = fnew; }
                    }
                }
                if (found) {   // (an end with v == 0 exactly is a root itself: brentq returns it at once)
                    push(1, a, x, fa, -fx);
                    push(1, x, b, -fx, fb);
                }
            }
            for (int k = 0; k < n_br; k++) {
                const int comp = k ? comp1 : comp0;
                auto f = [&](double t) { n_eval++; const double2 q = uv_lds(t, sp, 256, m); return comp ? q.y : q.x; };
                const double r = brentq(f, k ? a1 : a0, k ? b1 : b0, k ? fa1 : fa0, k ? fb1 : fb0);
                if (ns == 0) r0 = r; else r1 = r;
                ns++;
            }
        }
        double c0a = NAN, c0b = NAN;
        if (ns > 0) c0a = det_exp_call(sp[1536] + r0 * r0) + m.inv_n;
        if (ns > 1) c0b = det_exp_call(sp[1536] + r1 * r1) + m.inv_n;
        if (ns > 1 && c0b < c0a) { double t = c0a; c0a = c0b; c0b = t; }
        out.n_sol[i] = ns;
        out.C0[i * NRHIP_MAXS] = c0a;
        out.C0[i * NRHIP_MAXS + 1] = c0b;
    }
    if (eval_count) {
        for (int off = 32; off > 0; off >>= 1) n_eval += __shfl_xor(n_eval, off);
        if ((threadIdx.x & 63) == 0 && n_eval) atomicAdd(eval_count, (unsigned long long)n_eval);
    }
}

#ifndef NRHIP_RTREC_WAVES
#define NRHIP_RTREC_WAVES 3
#endif
__global__ void __launch_bounds__(256, NRHIP_RTREC_WAVES)
raytrace_records_kernel(long n_pairs, const double* __restrict__ x1, const double* __restrict__ x2, int n_ch,
                        IceConst m, RayRecords out, const double* __restrict__ given_C0, const double* __restrict__ given_D,
                        const double* __restrict__ given_T)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_pairs; i += (long)gridDim.x * blockDim.x) {
        const long i1 = (n_ch > 0) ? i / n_ch : i;
        const long i2 = (n_ch > 0) ? i % n_ch : i;
        const PairGeom g = pair_geometry(x1, x2, i1, i2);
        Pair2D p;
        p.y1 = g.A0; p.z1 = g.A2; p.y2 = g.y2; p.z2 = g.z2;
        p.g1 = gamma_of_z(p.z1, m);
        p.g2 = gamma_of_z(p.z2, m);
        const bool swap = g.swap;
        const double cph = g.cph, sph = g.sph;
        int ns;
        double c0v[NRHIP_MAXS];
        int src0 = 0, src1 = 1;   // which given slot each solution came from (given_D / given_T are indexed like given_C0)
        if (given_C0) {  // ray_tracing.set_solution (:2092): launch parameters read back from a file, no root finding
            ns = 0;
            double c0a = NAN, c0b = NAN;
            for (int k = 0; k < NRHIP_MAXS; k++) {   // (NRHIP_MAXS = 2 slots)
                double v = given_C0[i * NRHIP_MAXS + k];
                if (!isnan(v)) {
                    if (ns == 0) { c0a = v; src0 = k; } else { c0b = v; src1 = k; }
                    ns++;
                }
            }
            if (ns > 1 && c0b < c0a) { double t = c0a; c0a = c0b; c0b = t; int q = src0; src0 = src1; src1 = q; }
            c0v[0] = c0a;
            c0v[1] = c0b;
            out.n_sol[i] = ns;
        } else {
            ns = out.n_sol[i];
            c0v[0] = out.C0[i * NRHIP_MAXS];
            c0v[1] = out.C0[i * NRHIP_MAXS + 1];
        }
#pragma unroll
        for (int s = 0; s < NRHIP_MAXS; s++) {
            long k = i * NRHIP_MAXS + s;
            if (s >= ns) {
                out.type[k] = 0;
                out.C0[k] = out.C1[k] = out.D[k] = out.T[k] = out.refl_angle[k] = NAN;
                for (int d = 0; d < 3; d++) out.launch[3 * k + d] = out.receive[3 * k + d] = NAN;
                continue;
            }
            const double c0s = c0v[s];
            C0State st = make_c0(c0s, m);
            double C1 = C1_of(st, p, m);
            int type = solution_type(st, C1, p);
            double sL, cL, s2, c2;
            ray_sincos(p.y1, p.z1, st, C1, p, m, &sL, &cL);  // launch angle  (:1195)
            ray_sincos(p.y2, p.z2, st, C1, p, m, &s2, &c2);  // receive angle = pi - this one (:1198)
            double D, T;
            path_length_time(st, C1, type, sL, p, m, &D, &T);
            // given path lengths / travel times (the reference's own numbers for its rays: its closed forms take the square root
            // of a fully cancelling difference at the turning point, 1e-8 of rounding noise in T that no second implementation
            // reproduces and that is a 1e-4 phase at 500 MHz -- DESIGN section 2)
            if (given_D) { const double v = given_D[i * NRHIP_MAXS + (s ? src1 : src0)]; if (!isnan(v)) D = v; }
            if (given_T) { const double v = given_T[i * NRHIP_MAXS + (s ? src1 : src0)]; if (!isnan(v)) T = v; }
            // 2-D -> 3-D via R^T (:2560-2624); for swapped end points launch and receive exchange roles
            double lv0 = sL, lv2 = cL, rv0 = -s2, rv2 = -c2;
            if (swap) {
                lv0 = -s2; lv2 = -c2;
                rv0 = sL;  rv2 = cL;
            }
            out.type[k] = type;
            out.C0[k] = c0s;
            out.C1[k] = C1;
            out.D[k] = D;
            out.T[k] = T;
            out.launch[3 * k + 0] = cph * lv0;
            out.launch[3 * k + 1] = -sph * lv0;
            out.launch[3 * k + 2] = lv2;
            out.receive[3 * k + 0] = cph * rv0;
            out.receive[3 * k + 1] = -sph * rv0;
            out.receive[3 * k + 2] = rv2;
            // surface reflection angle (:1201-1237); NaN encodes the reference's None
            double y_turn = st.y_turn0 + C1;
            double refl = NAN;
            if (st.z_turn >= 0 && y_turn > p.y1 && y_turn < p.y2) {
                double sr, cr;
                ray_sincos(y_turn, 0., st, C1, p, m, &sr, &cr);
                refl = atan2(sr, cr);
            }
            out.refl_angle[k] = refl;
        }
    }
}

// geometry cell of an event: NRHIP_GEO_SIDE^2 cells in (horizontal distance to the station's first antenna, vertex depth)
#define NRHIP_GEO_SIDE 128
__global__ void __launch_bounds__(256)
event_cell_kernel(int n_events, const double* __restrict__ vertex, const double* __restrict__ x2, int* __restrict__ cell,
                  int* __restrict__ hist)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    double dx = vertex[3 * (long)e] - x2[0], dy = vertex[3 * (long)e + 1] - x2[1];
    const int S = NRHIP_GEO_SIDE;
    int cr = (int)(sqrt(dx * dx + dy * dy) * (S / 5000.)), cz = (int)(-vertex[3 * (long)e + 2] * (S / 3000.));
    cr = cr < 0 ? 0 : (cr > S - 1 ? S - 1 : cr);
    cz = cz < 0 ? 0 : (cz > S - 1 ? S - 1 : cz);
    int c = cz * S + cr;
    cell[e] = c;
    atomicAdd(&hist[c], 1);
}

__global__ void __launch_bounds__(256)
event_perm_kernel(int n_events, const int* __restrict__ cell, int* __restrict__ cursor, int* __restrict__ perm)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    perm[atomicAdd(&cursor[cell[e]], 1)] = e;  // order inside a cell is irrelevant: outputs go to the original index
}

void launch_event_cells(hipStream_t stream, int n_events, const double* vertex, const double* x2, int* cell, int* hist)
{
    hipLaunchKernelGGL(event_cell_kernel, dim3((n_events + 255) / 256), dim3(256), 0, stream, n_events, vertex, x2, cell, hist);
}
void launch_event_perm(hipStream_t stream, int n_events, const int* cell, int* cursor, int* perm)
{
    hipLaunchKernelGGL(event_perm_kernel, dim3((n_events + 255) / 256), dim3(256), 0, stream, n_events, cell, cursor, perm);
}

void launch_raytrace(hipStream_t stream, long n_pairs, const double* x1, const double* x2, int n_ch,
                     const IceConst& m, const RayRecords& out, const double* max_dist, const int* perm, const double* given_C0,
                     unsigned long long* eval_count, const double* given_D, const double* given_T, bool reference_procedure, bool channel_major)
{
    if (n_pairs <= 0) return;
    int block = 256;
    long grid = (n_pairs + block - 1) / block;
    if (grid > 256L * 64) grid = 256L * 64;
    if (!given_C0) {
        // reference_procedure (nrhip_ctx_set_ray_finder(NRHIP_FINDER_REFERENCE); NRHIP_RT_REFERENCE_PROCEDURE=1 forces it for a whole
        // process): hybr + its acceptance test + two Brent searches for every pair, analyticraytracing.py:1476-1547 to the letter
        static const bool ref_proc = getenv("NRHIP_RT_REFERENCE_PROCEDURE") != nullptr && atoi(getenv("NRHIP_RT_REFERENCE_PROCEDURE")) != 0;
        if (ref_proc || reference_procedure) {
            hipLaunchKernelGGL(raytrace_roots_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_pairs, x1, x2, n_ch, m, out, max_dist, perm, eval_count, 0, 1);
        } else {
            static const int force_order = getenv("NRHIP_RT_EVENT_MAJOR") ? 1 : (getenv("NRHIP_RT_CHANNEL_MAJOR") ? 2 : 0);   // (either order: same results)
            const int event_major = force_order ? (force_order == 1) : !channel_major;
            hipLaunchKernelGGL(raytrace_roots_fast_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_pairs, x1, x2, n_ch, m, out, max_dist, perm, eval_count, event_major);
            // the flagged pairs (receiver deeper than 10 z_0, end points exactly above each other -- whatever the antennas' depth)
            // through the reference's procedure with the sign-change acceptance; with nothing flagged the launch reads one word per pair
            hipLaunchKernelGGL(raytrace_roots_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_pairs, x1, x2, n_ch, m, out, max_dist, perm, eval_count, 1, 0);
        }
    }
    hipLaunchKernelGGL(raytrace_records_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_pairs, x1, x2, n_ch, m, out, given_C0,
                       given_C0 ? given_D : nullptr, given_C0 ? given_T : nullptr);
}

}  // namespace nrhip
