// arz.hip -- the ARZ time-domain Askaryan model on MI355X (gfx950): vector potential of a charge-excess profile and the
// electric-field trace derived from it (NuRadioMC/SignalGen/ARZ/ARZ.py: get_vector_potential :36-275, ARZ.get_time_trace
// :500-673).
//
// A(t) = -mu / (4 pi) int dz' Q(z') v_perp F_p(t_ret(z', t)) / R(z') is a trapezoid sum over the profile for each of the
// N + 1 observer times; stretches of the profile that radiate within +-1 ns of the observer time are refined 100x
// (np.arange / np.interp on the slice, restated below), so a time bin costs between ~5e2 and ~5e4 integrand evaluations
// (exp + pow each).  Mapping: grid = (ray, 1/8 of the observer times that can see the shower); inside a block every WAVE owns one observer
// time at a time and its 64 lanes stride over the points of the (refined) profile; the +-1 ns stretches are found from
// wave ballots (the profile points of a wave are consecutive, so a gap is a bit flip in the ballot word), the trapezoid
// rule is evaluated as sum_j y_j (z_{j+1} - z_{j-1}) / 2 so that every point is evaluated once, and a shuffle reduction
// gives the two non-zero components (x, z) of A.  The profile (depth, charge excess) sits in LDS.  A second kernel
// differentiates and rotates into the on-sky basis of the direction to the shower maximum.
// FP64 VALU / transcendental bound: ~24 B of HBM traffic per observer time against 1e3..1e5 exp/pow evaluations.
#include <hip/hip_runtime.h>
#include "arz.h"
#include "detmath.h"

namespace nrhip {

#ifndef ARZ_CHUNKS
#define ARZ_CHUNKS 4      // blocks per ray: each takes a quarter of the observer times that can see the shower (2 .. 8 measured: 407 .. 419 ms per 1e5 events)
#endif
#define ARZ_MAX_PROFILE 2048

static __device__ const double ARZ_RHO = 5.767155003928648e+39;   // 0.924 g / cm^3 in NuRadioReco units (ARZ.py:31)
static __device__ const double ARZ_XMU = 2.0133542226782937e-07;  // 12.566370e-7 N / A^2 (:32)
static __device__ const double ARZ_C = 0.299792458;               // m / ns (:33)

struct ArzRay {
    double X0, X2, R0, xntot, E_TeV, em_factor;
    double Af, freq_pos, freq_neg, exp_pos, exp_neg, t0_pos, t0_neg;
    double K, inv_t0_pos, inv_t0_neg;   // Af E_TeV fc / xntot em_factor; 1 / t0
    const double2 *tab_pos, *tab_neg;   // form-factor polynomials of this shower type (nullptr: evaluate directly)
    const double2 *far_pos, *far_neg;   // the same beyond 2.5 ns (cells of 1/16 ns)
};

// cell i of table (type, sign): Taylor coefficients about the cell centre c = (i + 1/2) / 512 ns of
// exp(-a / t0) + (1 + f a)^e = sum_n [exp(-c / t0) (-1 / t0)^n / n! + (1 + f c)^e binom(e, n) (f / (1 + f c))^n] (a - c)^n;
// with |a - c| <= 1/1024 ns and t0 >= 0.02 ns the first neglected term is 1.2e-13 of the exponential, far less of the power law
__global__ void arz_form_factor_table_kernel(const double* __restrict__ parameters, double* __restrict__ table)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * (ARZ_TABLE_CELLS + ARZ_FAR_CELLS)) return;
    // rows (type, sign) of ARZ_TABLE_CELLS near cells (1/512 ns) first, then the rows of ARZ_FAR_CELLS far cells (1/16 ns from 2.5 ns)
    const bool far = i >= 4 * ARZ_TABLE_CELLS;
    const int j = far ? i - 4 * ARZ_TABLE_CELLS : i, per = far ? ARZ_FAR_CELLS : ARZ_TABLE_CELLS;
    const int cell = j % per, sign = (j / per) & 1, typ = j / (2 * per);
    const double* P = parameters + 7 * typ;
    const double f = sign ? P[2] : P[1], e = sign ? P[4] : P[3], t0 = sign ? P[6] : P[5];
    const double c = far ? 2.5 + (cell + 0.5) * (1. / 16.) : (cell + 0.5) * (1. / 512.);
    const double y = 1. + f * c, ratio = f / y;
    double ce = exp(-c / t0), cp = pow(y, e);
    double* out = table + (size_t)i * 8;
    for (int n = 0; n < 7; n++) {
        out[n] = ce + cp;
        ce *= (-1. / t0) / (n + 1);
        cp *= ratio * (e - n) / (n + 1);
    }
    out[7] = 0.;
}


// The same integrand with the per-ray constants folded (one reciprocal of R instead of six divisions) and x^e as exp(e log x)
// through the table-free exp / log of detmath.h (<= 1 ulp each; |e log x| < 20, so the power is good to 1e-14): the
// evaluation count of the time-domain model is ~1e6 per ray, all of it this function.  The values differ from the spelling
// above by rounding only (1e-13 of the trace maximum measured against the reference's traces).
// rsqrt() of a positive finite number: the compiler's own expansion (hardware estimate + one third-order correction) without its
// test for zero / infinity / NaN inputs -- the same bits, four instructions less per evaluation
__device__ __forceinline__ double arz_rsqrt_pos(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double t = y0 * (-x);
    const double e = fma(t, y0, 1.0);
    const double u = y0 * e;
    const double p = fma(e, 0.375, 0.5);
    return fma(u, p, y0);
}

__device__ __forceinline__ void arz_integrand_fast(const ArzRay& r, double depth, double q, double tobs, double n_index,
                                                   double* yx, double* yz)
{
    const double z = depth * (1. / ARZ_RHO);
    const double dz = r.X2 - z;
    const double R2 = r.X0 * r.X0 + dz * dz;
    const double invR = arz_rsqrt_pos(R2);   // (R2 >= X0^2 > 0)
    const double R = R2 * invR;
    const double t = ((ARZ_C * tobs - n_index * R) - z) * (1. / ARZ_C);
    double F = 0.;
    if (t < 20. && t > -20.) {
        const double a = fabs(t);
        const bool pos = t > 0;
        if (a < ARZ_TABLE_CELLS * (1. / 512.) && r.tab_pos) {
#ifdef ARZ_PROBE_CONST_CELL   // measurement only (wrong values): what the table's memory latency costs
            const int ci = 0;
#else
            const int ci = (int)(a * 512.);
#endif
            const double dl = a - (ci + 0.5) * (1. / 512.);
            const double2* T = (pos ? r.tab_pos : r.tab_neg) + 4 * ci;
            const double2 c01 = T[0], c23 = T[1], c45 = T[2], c67 = T[3];
            double p = fma(c67.x, dl, c45.y);
            p = fma(p, dl, c45.x);
            p = fma(p, dl, c23.y);
            p = fma(p, dl, c23.x);
            p = fma(p, dl, c01.y);
            p = fma(p, dl, c01.x);
            F = r.K * p;
        } else if (r.tab_pos) {   // 2.5 ns <= |t| < 20 ns: the far table
            int ci = (int)((a - 2.5) * 16.);
            ci = ci > ARZ_FAR_CELLS - 1 ? ARZ_FAR_CELLS - 1 : ci;
            const double dl = a - (2.5 + (ci + 0.5) * (1. / 16.));
            const double2* T = (pos ? r.far_pos : r.far_neg) + 4 * ci;
            const double2 c01 = T[0], c23 = T[1], c45 = T[2], c67 = T[3];
            double p = fma(c67.x, dl, c45.y);
            p = fma(p, dl, c45.x);
            p = fma(p, dl, c23.y);
            p = fma(p, dl, c23.x);
            p = fma(p, dl, c01.y);
            p = fma(p, dl, c01.x);
            F = r.K * p;
        } else {
            const double e1 = det_exp_inrange(fmax(-a * (pos ? r.inv_t0_pos : r.inv_t0_neg), -745.));
            const double e2 = det_exp_inrange((pos ? r.exp_pos : r.exp_neg) * det_log(1. + (pos ? r.freq_pos : r.freq_neg) * a));
            F = r.K * (e1 + e2);
        }
    }
    const double ux = r.X0 * invR;
    const double g = ux * q * F * invR;
    *yx = -((dz * invR) * g);
    *yz = ux * g;
}

// integrand -v Q F_p / R at shower depth `depth` (g/cm^2 in internal units) for observer time tobs: x and z components
__device__ inline void arz_integrand(const ArzRay& r, double depth, double q, double tobs, double n_index, double fc,
                                     double* yx, double* yz, double* tt_out)
{
    const double z = depth / ARZ_RHO;
    const double R = sqrt(r.X0 * r.X0 + (r.X2 - z) * (r.X2 - z));
    const double arg = z - (ARZ_C * tobs - n_index * R);
    const double t = -arg / ARZ_C;
    *tt_out = t;
    double F = 0.;
    if (t < 20. && t > -20.) {
        const double a = fabs(t);
        double A;
        if (t > 0) A = r.Af * r.E_TeV * (exp(-a / r.t0_pos) + pow(1. + r.freq_pos * a, r.exp_pos));
        else A = r.Af * r.E_TeV * (exp(-a / r.t0_neg) + pow(1. + r.freq_neg * a, r.exp_neg));
        F = A * fc / r.xntot * r.em_factor;
    }
    const double ux = r.X0 / R, uz = (r.X2 - z) / R;
    *yx = -(ux * uz) * q * F / R;
    *yz = (ux * ux) * q * F / R;  // -v_z, v_z = -(u_x^2 + u_y^2)
}

__device__ inline double arz_tt(const ArzRay& r, double depth, double tobs, double n_index)
{
    const double z = depth / ARZ_RHO;
    const double R = sqrt(r.X0 * r.X0 + (r.X2 - z) * (r.X2 - z));
    return -(z - (ARZ_C * tobs - n_index * R)) / ARZ_C;
}

// np.interp(x, xp[is:ie], fp[is:ie]): constant beyond the slice's last node; `guess` = a node index near x
__device__ inline double arz_interp_slice(double x, const double* __restrict__ xp, const double* __restrict__ fp, int is, int ie,
                                          int guess)
{
    const int last = ie - 1;
    if (x <= xp[is]) return fp[is];
    if (x >= xp[last]) return fp[last];
    int k = guess < is ? is : (guess > last - 1 ? last - 1 : guess);
    while (k > is && xp[k] > x) k--;
    while (k < last - 1 && xp[k + 1] <= x) k++;
    const double slope = (fp[k + 1] - fp[k]) / (xp[k + 1] - xp[k]);
    return slope * (x - xp[k]) + fp[k];
}

// One stretch of the refined profile: coarse nodes [c0, c1) followed by n_fine points start + k * delta (slice [is, ie))
struct ArzPiece { int c0, c1, is, ie; long n_fine; double start, step, delta; };

#ifndef ARZ_WAVES
#define ARZ_WAVES 4   // waves per SIMD the register budget is cut for: 2 .. 8 measured, 4 is fastest (355 vs 410 .. 420 ms per 1e5 events at 2 / 3)
#endif
__global__ void __launch_bounds__(256, ARZ_WAVES)
arz_vector_potential_kernel(ArzBatch b, double* __restrict__ vp /* [n_rays][N + 1][2] */, int* __restrict__ status,
                            int* __restrict__ vp_range /* [n_rays][2] or nullptr */)
{
    extern __shared__ double lds[];
    const int ray = blockIdx.x;
    const double nidx = b.n_index_ray ? b.n_index_ray[ray] : b.n_index;
    const int nd = b.n_depth;
    double* s_depth = lds;
    double* s_ce = lds + nd;
    double* s_slope = lds + 2 * nd;   // slope of the profile between nodes k and k + 1 (np.interp's own expression)
    double* s_z = lds + 3 * nd;       // depth / rho: position along the shower axis
    double* s_nR = lds + 4 * nd;      // n * distance from the node to the observer
    __shared__ double s_w[4][4];   // per wave: sum, max value, h min, h max
    __shared__ int s_wi[4];        // per wave: index of the max
    const int nt = b.N + 1;
    const double theta = b.theta[ray];
    const int typ = b.shower_type[ray];  // 0 HAD, 1 EM
    // 20 deg cut (ARZ.py:603-607): the trace kernel writes zeros, nothing to integrate
    if (fabs(theta - acos(1. / nidx)) > b.maximum_angle) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const double* ce_g = b.profile_ce + (long)b.profile_index[ray] * nd;
    const double resc = b.rescale ? b.rescale[ray] : 1.;
    // profile -> LDS; sum and first maximum (np.argmax) by a wave / block reduction
    double part = 0., bestv = -1e300;
    int best = 0x7fffffff;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        const double q = ce_g[i] * resc;
        s_depth[i] = b.profile_depth[i];
        s_ce[i] = q;
        part += q;
        if (q > bestv) { best = i; bestv = q; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        part += __shfl_xor(part, off);
        const double ov = __shfl_xor(bestv, off);
        const int oi = __shfl_xor(best, off);
        if (ov > bestv || (ov == bestv && oi < best)) { bestv = ov; best = oi; }
    }
    if (lane == 0) { s_w[wave][0] = part; s_w[wave][1] = bestv; s_wi[wave] = best; }
    __syncthreads();
    for (int i = threadIdx.x; i < nd - 1; i += blockDim.x) s_slope[i] = (s_ce[i + 1] - s_ce[i]) / (s_depth[i + 1] - s_depth[i]);
    double sum = 0.;
    int im = 0x7fffffff;
    {
        double mv = -1e300;
        for (int w = 0; w < n_waves; w++) {
            sum += s_w[w][0];
            if (s_w[w][1] > mv || (s_w[w][1] == mv && s_wi[w] < im)) { mv = s_w[w][1]; im = s_wi[w]; }
        }
    }
    ArzRay r;
    {
        const double dist = b.distance[ray];
        const double dxmax = s_depth[im] / ARZ_RHO;
        r.X0 = dist * sin(theta);
        r.X2 = dist * cos(theta) + (b.shift_for_xmax ? dxmax : 0.);
        r.R0 = sqrt(r.X0 * r.X0 + r.X2 * r.X2);
        r.xntot = sum * (s_depth[1] / ARZ_RHO - s_depth[0] / ARZ_RHO);
        r.E_TeV = b.energy[ray] / 1e12;
        r.em_factor = (typ == 0) ? b.em_factor[ray] : 1.;
        const double* P = b.parameters + 7 * typ;
        r.Af = P[0]; r.freq_pos = P[1]; r.freq_neg = P[2]; r.exp_pos = P[3]; r.exp_neg = P[4]; r.t0_pos = P[5]; r.t0_neg = P[6];
    }
    // delay of the profile nodes relative to the first one, h_i = t_obs - tt_i: an observer time t with t - 20 ns above
    // all of them or t + 20 ns below all of them sees no node inside +-20 ns (the `continue` of ARZ.py:160-162).  Only the
    // observer times in between are distributed over the ARZ_CHUNKS blocks of this ray.
    double hmin = 1e300, hmax = -1e300;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        const double h = -arz_tt(r, s_depth[i], r.R0 / ARZ_C * nidx, nidx);
        hmin = fmin(hmin, h);
        hmax = fmax(hmax, h);
    }
    for (int off = 32; off > 0; off >>= 1) {
        hmin = fmin(hmin, __shfl_xor(hmin, off));
        hmax = fmax(hmax, __shfl_xor(hmax, off));
    }
    if (lane == 0) { s_w[wave][2] = hmin; s_w[wave][3] = hmax; }
    __syncthreads();
    for (int w = 0; w < n_waves; w++) {
        hmin = fmin(hmin, s_w[w][2]);
        hmax = fmax(hmax, s_w[w][3]);
    }
    const double fc = 4. * M_PI / (ARZ_XMU * sin(acos(1. / nidx)));
    const double factor = -ARZ_XMU / (4. * M_PI);
    r.K = r.Af * r.E_TeV * fc / r.xntot * r.em_factor;
    r.inv_t0_pos = 1. / r.t0_pos;
    r.inv_t0_neg = 1. / r.t0_neg;
    // the polynomials are good to 1e-12 for the published parameter sets (t0 >= 0.02 ns, |e| f <= 12 / ns); anything far from
    // those is evaluated directly
    const bool tab_ok = b.form_factor_table && fmin(r.t0_pos, r.t0_neg) >= 0.015 &&
                        fmax(fabs(r.exp_pos) * r.freq_pos, fabs(r.exp_neg) * r.freq_neg) <= 20.;
    r.tab_pos = tab_ok ? (const double2*)(b.form_factor_table + (size_t)(2 * typ) * ARZ_TABLE_CELLS * 8) : nullptr;
    r.tab_neg = tab_ok ? (const double2*)(b.form_factor_table + (size_t)(2 * typ + 1) * ARZ_TABLE_CELLS * 8) : nullptr;
    {
        const double* far0 = b.form_factor_table ? b.form_factor_table + (size_t)4 * ARZ_TABLE_CELLS * 8 : nullptr;
        r.far_pos = tab_ok ? (const double2*)(far0 + (size_t)(2 * typ) * ARZ_FAR_CELLS * 8) : nullptr;
        r.far_neg = tab_ok ? (const double2*)(far0 + (size_t)(2 * typ + 1) * ARZ_FAR_CELLS * 8) : nullptr;
    }
    const double inv_coarse = 1. / (s_depth[1] - s_depth[0]);
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        const double z = s_depth[i] / ARZ_RHO;
        s_z[i] = z;
        s_nR[i] = nidx * sqrt(r.X0 * r.X0 + (r.X2 - z) * (r.X2 - z));
    }
    __syncthreads();
    // observer times: arange(0, (N + 1) dt, dt) + dt / 2 - mean (:98-102)
    const int nt_raw = (int)ceil(((b.N + 1) * b.dt - 0.) / b.dt);
    const double mean = b.dt * (nt_raw - 1) * 0.5;
    // the observer times that can see the shower: hmin - 20 ns < t < hmax + 20 ns (1e-3 ns of slack, checked again per time)
    int it_lo = (int)floor((hmin - 20.001 + mean - 0.5 * b.dt) / b.dt), it_hi = (int)ceil((hmax + 20.001 + mean - 0.5 * b.dt) / b.dt);
    it_lo = max(it_lo, 0);
    it_hi = min(it_hi, nt - 1);
    // Only the observer times it_lo .. it_hi are written (all of them).  With vp_range the caller learns the window and takes the
    // vector potential outside it as zero (arz_trace_kernel); without it the launcher has zeroed the whole array.
    if (vp_range && blockIdx.y == 0 && threadIdx.x == 0) { vp_range[2 * ray] = it_lo; vp_range[2 * ray + 1] = it_hi; }
    unsigned n_eval = 0;   // integrand evaluations of this lane
    const int per = (it_hi - it_lo + 1 + ARZ_CHUNKS - 1) / ARZ_CHUNKS;
    const int it_end = min(it_hi + 1, it_lo + (int)(blockIdx.y + 1) * per);
    for (int it = it_lo + blockIdx.y * per + wave; it < it_end; it += n_waves) {
        const double t_bin = it * b.dt + 0.5 * b.dt - mean;
        if (t_bin - hmax > 20.001 || t_bin - hmin < -20.001) {   // no node within +-20 ns: A = 0 (written: the window below is dense)
            if (lane == 0) { vp[((long)ray * nt + it) * 2] = 0.; vp[((long)ray * nt + it) * 2 + 1] = 0.; }
            continue;
        }
        const double tobs = t_bin + (r.R0 / ARZ_C * nidx);
        const double ct = ARZ_C * tobs;
        // pass 1 over the profile nodes: anything within +-20 ns?  where does the +-1 ns condition flip?  The flip positions are
        // wave-uniform; only the first five are kept (more than four is the reference's NotImplementedError), in scalars
        int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, ni = 0;
        auto push = [&](int v) {
            if (ni == 0) i0 = v; else if (ni == 1) i1 = v; else if (ni == 2) i2 = v; else if (ni == 3) i3 = v; else if (ni == 4) i4 = v;
            ni++;
        };
        bool any20 = false, first_in = false;
        int prev_last = 0;
        for (int base = 0; base < nd; base += 64) {
            const int i = base + lane;
            double t = 1e300;
            if (i < nd) t = -(s_z[i] - (ct - s_nR[i])) / ARZ_C;
            const unsigned long long B20 = __ballot(t < 20. && t > -20.);
            const unsigned long long B1 = __ballot(t < 1. && t > -1.);
            any20 |= (B20 != 0ull);
            if (base == 0) first_in = (B1 & 1ull) != 0;
            const int cnt = min(64, nd - base);
            // flips between node base - 1 and base, then inside the chunk
            if (base > 0 && (prev_last != (int)(B1 & 1ull)) && ni < 16) push(base - 1);
            unsigned long long G = (B1 ^ (B1 >> 1));
            if (cnt < 64) G &= (cnt >= 2) ? ((1ull << (cnt - 1)) - 1ull) : 0ull;
            else G &= 0x7fffffffffffffffull;
            while (G && ni < 16) {
                const int bit = __ffsll((long long)G) - 1;
                push(base + bit);
                G &= G - 1ull;
            }
            prev_last = (int)((B1 >> (cnt - 1)) & 1ull);
        }
        double ax = 0., az = 0.;
        if (any20) {
            // up to three pieces: coarse nodes [c0, c1) followed by n_fine points start + k * delta (slice [is, ie))
            int pc0[3] = {0, 0, 0}, pc1[3] = {nd, 0, 0}, pis[3] = {0, 0, 0}, pie[3] = {0, 0, 0}, pnf[3] = {0, 0, 0};
            double pstart[3] = {0., 0., 0.}, pdelta[3] = {0., 0., 0.};
            int n_piece = 1;
            const double step = (s_depth[1] - s_depth[0]) / b.interp_factor2;
            if (b.interp_factor2 != 1. && ni != 0) {
                if (ni % 2 != 0) {  // a stretch that starts with the first / ends with the last node (:176-181)
                    if (first_in && i0 != 0) {
                        i4 = i3; i3 = i2; i2 = i1; i1 = i0; i0 = 0;
                        ni++;
                    } else {
                        const int last_idx = (ni == 1) ? i0 : (ni == 3 ? i2 : (ni == 5 ? i4 : -1));   // -1: beyond what is kept
                        if (last_idx != nd - 1) push(nd - 1);
                    }
                }
                if (ni % 2 == 0 && ni != 2 && ni != 4) {
                    if (lane == 0) atomicExch(&status[ray], 1);  // NotImplementedError in the reference (:207)
                    ni = 0;
                }
                if (ni == 2 || ni == 4) {
                    n_piece = 0;
                    int from = 0;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        if (2 * q < ni) {
                            const int is = q ? i2 : i0, ie = q ? i3 : i1;
                            const double start = s_depth[is];
                            pc0[q] = from; pc1[q] = is; pis[q] = is; pie[q] = ie;
                            pnf[q] = (int)ceil((s_depth[ie] - start) / step);
                            pstart[q] = start;
                            pdelta[q] = (start + step) - start;
                            from = ie;
                            n_piece = q + 1;
                        }
                    }
                    if (n_piece == 1) { pc0[1] = from; pc1[1] = nd; pnf[1] = 0; }
                    else { pc0[2] = from; pc1[2] = nd; pnf[2] = 0; }
                    n_piece++;
                }
            }
            // the merged grid: per piece its coarse nodes, then its fine points; g = index on the merged grid
            int off[4] = {0, 0, 0, 0};
#pragma unroll
            for (int ip = 0; ip < 3; ip++) off[ip + 1] = off[ip] + ((ip < n_piece) ? (pc1[ip] - pc0[ip]) + pnf[ip] : 0);
            const int M = off[3];
            auto depth_at = [&](int g) -> double {   // any point of the merged grid (used at the seams only)
                double res = 0.;
#pragma unroll
                for (int ip = 0; ip < 3; ip++) {
                    if (ip < n_piece && g >= off[ip] && g < off[ip + 1]) {
                        const int j = g - off[ip], nc = pc1[ip] - pc0[ip];
                        if (j < nc) res = s_depth[pc0[ip] + j];
                        else {
                            const int k = j - nc;
                            res = (k == 0) ? pstart[ip] : (k == 1 ? pstart[ip] + step : pstart[ip] + k * pdelta[ip]);
                        }
                    }
                }
                return res;
            };
            // trapezoid rule sum_j (z_{j+1} - z_j) (y_{j+1} + y_j) / 2 = sum_j y_j (z_{j+1} - z_{j-1}) / 2
#pragma unroll
            for (int ip = 0; ip < 3; ip++) {
                if (ip >= n_piece) continue;
                const int nc = pc1[ip] - pc0[ip], nf = pnf[ip];
                for (int j = lane; j < nc; j += 64) {
                    const int n = pc0[ip] + j, g = off[ip] + j;
                    const double tn = ((ct - s_nR[n]) - s_z[n]) * (1. / ARZ_C);
                    if (!(tn < 20.5 && tn > -20.5)) continue;   // F = 0 there (the exact comparison is made inside)
                    const double x = s_depth[n];
                    const double xl = (j > 0) ? s_depth[n - 1] : (g > 0 ? depth_at(g - 1) : x);
                    const double xr = (j + 1 < nc) ? s_depth[n + 1] : (g + 1 < M ? depth_at(g + 1) : x);
                    double yx, yz;
                    n_eval++;
                    arz_integrand_fast(r, x, s_ce[n], tobs, nidx, &yx, &yz);
                    const double w = (xr - xl) * (0.5 / ARZ_RHO);
                    ax += w * yx;
                    az += w * yz;
                }
                const int is = pis[ip], last = pie[ip] - 1;
                const double start = pstart[ip], delta = pdelta[ip];
                // (the slice's end nodes once per stretch: inside the loop they were two dependent LDS round trips per point)
                const double d_is = nf ? s_depth[is] : 0., d_last = nf ? s_depth[last] : 0., ce_is = nf ? s_ce[is] : 0., ce_last = nf ? s_ce[last] : 0.;
                for (int k = lane; k < nf; k += 64) {
                    const int g = off[ip] + nc + k;
                    const double xk = start + k * delta;
                    asm volatile("" :: "v"(xk));   // (selects, not two branches around a conversion and a multiply-add)
                    const double x = (k == 0) ? start : (k == 1 ? start + step : xk);
                    double q;   // np.interp on the slice [is, ie): constant beyond its last node
                    if (x <= d_is) q = ce_is;
                    else if (x >= d_last) q = ce_last;
                    else {
                        int kk = is + (int)((x - start) * inv_coarse);
                        kk = kk < is ? is : (kk > last - 1 ? last - 1 : kk);
                        // the guessed cell is nearly always the right one: its four numbers in ONE LDS round trip, np.interp's search
                        // (two dependent round trips per point) only where the guess is off
                        double dk = s_depth[kk], sl = s_slope[kk], ck = s_ce[kk];
                        const double dk1 = s_depth[kk + 1];
                        const bool cell_ok = ((kk <= is) | (dk <= x)) & ((kk >= last - 1) | (dk1 > x));   // (no short circuit: the reads stay together)
                        asm volatile("" :: "v"(sl), "v"(ck));   // (slope and value of the cell are requested with its ends, not after the test)
                        if (!cell_ok) {
                            while (kk > is && s_depth[kk] > x) kk--;
                            while (kk < last - 1 && s_depth[kk + 1] <= x) kk++;
                            dk = s_depth[kk]; sl = s_slope[kk]; ck = s_ce[kk];
                        }
                        q = sl * (x - dk) + ck;
                    }
                    double xl, xr;
                    if (k >= 3 && k + 1 < nf) {   // inside the stretch: the neighbours are fine points too
                        xl = start + (k - 1) * delta;
                        xr = start + (k + 1) * delta;
                    } else {
                        xl = (g > 0) ? depth_at(g - 1) : x;
                        xr = (g + 1 < M) ? depth_at(g + 1) : x;
                    }
                    double yx, yz;
                    n_eval++;
                    arz_integrand_fast(r, x, q, tobs, nidx, &yx, &yz);
                    const double w = (xr - xl) * (0.5 / ARZ_RHO);
                    ax += w * yx;
                    az += w * yz;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                ax += __shfl_xor(ax, off);
                az += __shfl_xor(az, off);
            }
        }
        if (lane == 0) {
            vp[((long)ray * nt + it) * 2] = ax * factor;
            vp[((long)ray * nt + it) * 2 + 1] = az * factor;
        }
    }
    if (b.eval_count) {
        for (int off = 32; off > 0; off >>= 1) n_eval += __shfl_xor(n_eval, off);
        if (lane == 0 && n_eval) atomicAdd(b.eval_count, (unsigned long long)n_eval);
    }
}

// E = -dA/dt, rotated into the on-sky basis of the direction to the shower maximum (:641-655); [n_rays][3][N]
__global__ void __launch_bounds__(256)
arz_trace_kernel(ArzBatch b, const double* __restrict__ vp, double* __restrict__ trace, const int* __restrict__ vp_range,
                 int* __restrict__ silent /* [n_rays] or nullptr: 1 = beyond the 20 degrees, the trace is all zeros and NOT written */)
{
    const int ray = blockIdx.x;
    const double nidx = b.n_index_ray ? b.n_index_ray[ray] : b.n_index;
    const int N = b.N, nt = N + 1, nd = b.n_depth;
    const double theta = b.theta[ray];
    double* out = trace + (long)ray * 3 * N;
    if (fabs(theta - acos(1. / nidx)) > b.maximum_angle) {
        if (silent) {   // (the caller honours the flag: half of the rays of a survey, 96 KB of zeros each)
            if (threadIdx.x == 0) silent[ray] = 1;
            return;
        }
        for (int i = threadIdx.x; i < 3 * N; i += blockDim.x) out[i] = 0.;
        return;
    }
    if (silent && threadIdx.x == 0) silent[ray] = 0;
    __shared__ double s_tp;
    if (threadIdx.x == 0) {
        double tp = theta;
        if (!b.shift_for_xmax) {  // theta_to_thetaprime (:299-315); argmax of the profile (a positive rescaling keeps it)
            const double* ce = b.profile_ce + (long)b.profile_index[ray] * nd;
            int im = 0;
            for (int i = 1; i < nd; i++)
                if (ce[i] > ce[im]) im = i;
            const double L = b.profile_depth[im] / ARZ_RHO, R = b.distance[ray];
            tp = atan2(R * sin(theta), R * cos(theta) - L);
        }
        s_tp = tp;
    }
    __syncthreads();
    const double ct = cos(s_tp), st = sin(s_tp);
    const double* v = vp + (long)ray * nt * 2;
    // (with vp_range: the vector potential was written for the observer times lo .. hi only and is zero -- not stored -- elsewhere)
    const int lo = vp_range ? vp_range[2 * ray] : 0, hi = vp_range ? vp_range[2 * ray + 1] : nt - 1;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const bool in0 = i >= lo && i <= hi, in1 = i + 1 >= lo && i + 1 <= hi;
        const double ax0 = in0 ? v[2 * i] : 0., az0 = in0 ? v[2 * i + 1] : 0.;
        const double ax1 = in1 ? v[2 * (i + 1)] : 0., az1 = in1 ? v[2 * (i + 1) + 1] : 0.;
        const double ex = -(ax1 - ax0) / b.dt, ez = -(az1 - az0) / b.dt;
        out[i] = st * ex + ct * ez;
        out[N + i] = ct * ex - st * ez;
        out[2 * N + i] = 0.;
    }
}

void launch_arz(hipStream_t s, const ArzBatch& b, double* vp, double* trace, int* status, int* vp_range, int* silent)
{
    if (b.n_rays <= 0) return;
    const int nt = b.N + 1;
    dim3 grid((unsigned)b.n_rays, ARZ_CHUNKS);
    // vp_range ([n_rays][2], optional): the window of observer times the kernel writes per ray; without it the array is zeroed first
    // (round 6: that memset was 155 GB of HBM writes per 2e4-event step of BASELINE config 4, read back as zeros by the trace kernel)
    if (!vp_range) (void)hipMemsetAsync(vp, 0, sizeof(double) * 2 * (size_t)nt * b.n_rays, s);
    if (b.form_factor_table)
        hipLaunchKernelGGL(arz_form_factor_table_kernel, dim3((4 * (ARZ_TABLE_CELLS + ARZ_FAR_CELLS) + 255) / 256), dim3(256), 0, s, b.parameters,
                           b.form_factor_table);
    const size_t lds = sizeof(double) * 5 * (size_t)b.n_depth;   // 80 KB at the 2048 depth bins the entry points admit
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)arz_vector_potential_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(arz_vector_potential_kernel, grid, dim3(256), lds, s, b, vp, status, vp_range);
    hipLaunchKernelGGL(arz_trace_kernel, dim3((unsigned)b.n_rays), dim3(256), 0, s, b, vp, trace, vp_range, silent);
}

}  // namespace nrhip
