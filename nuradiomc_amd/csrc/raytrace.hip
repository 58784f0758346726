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
#include "ray_device.h"
#include "nrhip_internal.h"
#include "root_device.h"

namespace nrhip {

// Kernel: pair i = (event i / n_ch, channel i % n_ch) when n_ch > 0, else x2 is per pair.
#ifndef NRHIP_RT_WAVES
#define NRHIP_RT_WAVES 6  // waves per SIMD the register budget is cut for (measured best)
#endif
__global__ void __launch_bounds__(256, NRHIP_RT_WAVES)
raytrace_kernel(long n_pairs, const double* __restrict__ x1, const double* __restrict__ x2, int n_ch,
                IceConst m, RayRecords out, const double* __restrict__ max_dist, const int* __restrict__ perm,
                const double* __restrict__ given_C0)
{
#ifndef NRHIP_RT_SCRATCH_PAIR
    __shared__ double sh_pair[6][256];   // the pair geometry the objective reads on every evaluation (see delta_y_lds)
#endif
    for (long iw = blockIdx.x * (long)blockDim.x + threadIdx.x; iw < n_pairs; iw += (long)gridDim.x * blockDim.x) {
        // perm (optional): events in an order that puts similar geometries (distance, depth) next to each other, so that
        // the lanes of a wave run similar numbers of root-finder iterations; results land at the original pair index
        long i1 = (n_ch > 0) ? iw / n_ch : iw;
        long i2 = (n_ch > 0) ? iw % n_ch : iw;
        if (perm) i1 = perm[i1];
        const long i = (n_ch > 0) ? i1 * n_ch + i2 : iw;
        double A[3] = {x1[3 * i1], x1[3 * i1 + 1], x1[3 * i1 + 2]};
        double B[3] = {x2[3 * i2], x2[3 * i2 + 1], x2[3 * i2 + 2]};
        // set_start_and_end_point (:2057-2090): the higher point becomes the stop point
        bool swap = B[2] < A[2];
        if (swap) {
            for (int d = 0; d < 3; d++) { double t = A[d]; A[d] = B[d]; B[d] = t; }
        }
        double dX[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
        // rotation by dPhi = -atan2(dy, dx) into the y-z plane, cos / sin taken algebraically (bit-reproducible)
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

        int ns = 0;
        double lc[3];
        // speedup.distance_cut (simulation.py:155-163): showers farther from the antenna than their cut are not traced
        const bool too_far = max_dist && sqrt(dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2]) > max_dist[i1];
        if (!(p.z2 > 0) && !too_far && !given_C0) {  // receiver in air: special branch of the reference (:1437-1460) not provided
#ifndef NRHIP_RT_SCRATCH_PAIR
            double* const sp = &sh_pair[0][threadIdx.x];   // (only this lane reads its column: no barrier)
            sp[0] = p.y1; sp[256] = p.z1; sp[512] = p.y2; sp[768] = p.z2; sp[1024] = p.g1; sp[1280] = p.g2;
            auto dy = [&](double l) { return delta_y_lds(l, sp, 256, m); };
            auto dy2 = [&](double l) { double d = delta_y_lds(l, sp, 256, m); return d * d; };
#else
            auto dy = [&](double l) { return delta_y(l, p, m); };
            auto dy2 = [&](double l) { double d = delta_y(l, p, m); return d * d; };
#endif
            double fun;
            double xr = hybrd1(dy2, -1., 1e-6, &fun);
            if (fun < 1e-7) lc[ns++] = xr;
            {
                double a = xr + 0.0001, b = 100.;
                double fa = dy(a), fb = dy(b);
                if (np_sign_differs(fa, fb) && signbit(fa) != signbit(fb)) lc[ns++] = brentq(dy, a, b, fa, fb);
            }
            {
                double a = -100., b = xr - 0.0001;
                double fa = dy(a), fb = dy(b);
                if (np_sign_differs(fa, fb) && signbit(fa) != signbit(fb)) lc[ns++] = brentq(dy, a, b, fa, fb);
            }
        }
        double c0[3];
        for (int k = 0; k < ns; k++) c0[k] = det_exp(lc[k]) + m.inv_n;
        if (given_C0) {  // ray_tracing.set_solution (:2092): launch parameters read back from a file, no root finding
            ns = 0;
            for (int k = 0; k < NRHIP_MAXS; k++) {
                double v = given_C0[i * NRHIP_MAXS + k];
                if (!isnan(v)) c0[ns++] = v;
            }
        }
        // sorted by C0 (insertion sort, <= 3 entries)
        for (int a = 1; a < ns; a++)
            for (int b = a; b > 0 && c0[b] < c0[b - 1]; b--) { double t = c0[b]; c0[b] = c0[b - 1]; c0[b - 1] = t; }
        if (ns > NRHIP_MAXS) ns = 0;  // "too many solutions -> none" (:2127-2130)
        out.n_sol[i] = ns;
        for (int s = 0; s < NRHIP_MAXS; s++) {
            long k = i * NRHIP_MAXS + s;
            if (s >= ns) {
                out.type[k] = 0;
                out.C0[k] = out.C1[k] = out.D[k] = out.T[k] = out.refl_angle[k] = NAN;
                for (int d = 0; d < 3; d++) out.launch[3 * k + d] = out.receive[3 * k + d] = NAN;
                continue;
            }
            C0State st = make_c0(c0[s], m);
            double C1 = C1_of(st, p, m);
            int type = solution_type(st, C1, p);
            double sL, cL, s2, c2;
            ray_sincos(p.y1, p.z1, st, C1, p, m, &sL, &cL);  // launch angle  (:1195)
            ray_sincos(p.y2, p.z2, st, C1, p, m, &s2, &c2);  // receive angle = pi - this one (:1198)
            double D, T;
            path_length_time(st, C1, type, sL, p, m, &D, &T);
            // 2-D -> 3-D via R^T (:2560-2624); for swapped end points launch and receive exchange roles
            double lv0 = sL, lv2 = cL, rv0 = -s2, rv2 = -c2;
            if (swap) {
                lv0 = -s2; lv2 = -c2;
                rv0 = sL;  rv2 = cL;
            }
            out.type[k] = type;
            out.C0[k] = c0[s];
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
                     const IceConst& m, const RayRecords& out, const double* max_dist, const int* perm, const double* given_C0)
{
    if (n_pairs <= 0) return;
    int block = 256;
    long grid = (n_pairs + block - 1) / block;
    if (grid > 256L * 64) grid = 256L * 64;
    hipLaunchKernelGGL(raytrace_kernel, dim3((unsigned)grid), dim3(block), 0, stream, n_pairs, x1, x2, n_ch, m, out, max_dist, perm, given_C0);
}

}  // namespace nrhip
