// spectral.hip -- Askaryan emission, propagation effects, antenna + amplifier response and the threshold
// trigger on MI355X (gfx950), all FFT work staged in LDS (fft_device.h).
//
// Reference behaviour being provided (paths in nu-radio/NuRadioMC):
//   * NuRadioMC/simulation/simulation.py:187-206 (viewing angle, delta_C cut), :244-246 + :798-819
//     (polarisation), :259-285 (trace start time, candidate amplitude cut)
//   * NuRadioMC/SignalGen/parametrizations.py:92-275 via askaryan.get_frequency_spectrum (askaryan.py:143)
//   * NuRadioMC/SignalProp/analyticraytracing.py:2937-3033 (attenuation interpolation, Fresnel)
//   * NuRadioReco/modules/efieldToVoltageConverter.py:111-345 (common time grid of event-dependent length L,
//     sub-sample shift, rfft(L), VEL, 5 MHz cut, sum over rays)
//   * NuRadioReco/detector/antennapattern.py:1190-1307, :1580-1768 (analytic VPol / HPol)
//   * NuRadioReco/utilities/signal_processing.py:282-292 (analog Butterworth response)
//   * NuRadioReco/modules/trigger/simpleThreshold.py + highLowThreshold.get_majority_logic (:82-142)
//
// Kernel roles and rooflines are in DESIGN.md; in short every kernel here is LDS/FP64 bound, not HBM bound:
// per ray only ~0.5 KB of parameters enter and 8..100 B leave, the N- and L-point transforms never leave LDS.
#include "fft_device.h"
#include "conv_fft.h"
#include "spectral.h"
#include "wave_reduce.h"
#ifndef NRHIP_WAVE_DPP
#define NRHIP_WAVE_DPP 1  // 0: the wave sums of the bound kernels by __shfl_xor butterflies (ds_bpermute_b32), as up to round 5
#endif
#include <cstdlib>

namespace nrhip {

#ifdef NRHIP_CONV_TIMING  // debug builds: shader clocks per phase, summed over the blocks (thread 0), read by nrhip_debug_conv_clocks
__device__ unsigned long long g_conv_clk[16];
__device__ unsigned long long g_ct_mark[1024];
#define CT_DECL unsigned long long ct_t = __builtin_amdgcn_s_memtime()
#define CT(i) do { unsigned long long ct_n = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_conv_clk[i], ct_n - ct_t); ct_t = ct_n; } while (0)
#else
#define CT_DECL
#define CT(i)
#endif


// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
__device__ inline void sph2cart(double zen, double az, double v[3])
{
    double sz = sin(zen);
    v[0] = sz * cos(az);
    v[1] = sz * sin(az);
    v[2] = cos(zen);
}

// radiotools.helper.cartesian_to_spherical: theta = arccos(z / r) (0 if z / r >= 1), phi in [0, 2 pi)
__device__ inline void cart2sph(const double v[3], double* zen, double* az)
{
    double r = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double q = v[2] / r;
    *zen = (q < 1) ? acos(q) : 0.;
    double phi = atan2(v[1], v[0]);
    const double twopi = 2 * M_PI;
    while (phi >= twopi) phi -= twopi;
    while (phi < 0) phi += twopi;
    *az = phi;
}

// rows e_r, e_theta, e_phi (radiotools cstrafo; analyticraytracing.py:2363-2365)
__device__ inline void onsky_matrix(double zen, double az, double Mx[9])
{
    double ct = cos(zen), st = sin(zen), cp = cos(az), sp = sin(az);
    Mx[0] = st * cp; Mx[1] = st * sp; Mx[2] = ct;
    Mx[3] = ct * cp; Mx[4] = ct * sp; Mx[5] = -st;
    Mx[6] = -sp;     Mx[7] = cp;      Mx[8] = 0;
}

__device__ inline void mat3vec(const double A[9], const double v[3], double o[3])
{
    for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// antenna frame: incoming direction in the antenna-model frame (theta_ant) and the 2x2 map T from the raw
// (theta, phi) components of the model's VEL to the on-sky basis of the arrival direction
// (antennapattern.py:1218-1307); orthonormal bases are inverted by transposition
__device__ inline void antenna_frame(double zen, double az, const double* rot, const double* roti, double T[4],
                                     double* theta_ant, double* phi_ant = nullptr)
{
    double inc[3], incw[3], th_a, ph_a;
    sph2cart(zen, az, inc);
    mat3vec(rot, inc, incw);
    cart2sph(incw, &th_a, &ph_a);
    double Ma[9], Ms[9];
    onsky_matrix(th_a, ph_a, Ma);
    onsky_matrix(zen, az, Ms);
    for (int c = 0; c < 2; c++) {  // raw component c+1 (theta, phi) = row c+1 of Ma as a cartesian vector
        double v[3] = {Ma[3 * (c + 1)], Ma[3 * (c + 1) + 1], Ma[3 * (c + 1) + 2]}, g[3], o[3];
        mat3vec(roti, v, g);
        mat3vec(Ms, g, o);
        T[0 + c] = o[1];
        T[2 + c] = o[2];
    }
    *theta_ant = th_a;
    if (phi_ant) *phi_ant = ph_a;
}

// weights of the on-sky eTheta / ePhi field in the channel voltage and the response table of the ray.  The analytic
// models factorise: VEL_raw(f; theta, phi) = B_table(f) * d(theta, phi) (antennapattern.py:1672-1768), and the rotation to
// the on-sky basis (T) is frequency independent, so V(f) = B(f) (vfac_t E_theta(f) + vfac_p E_phi(f)).
__device__ inline void antenna_factors(int model, const double T[4], double th_a, double ph_a, double* vfac_t,
                                       double* vfac_p, int* table)
{
    double d_theta = 0., d_phi = 0.;
    if (model == 0) {          // VPol: theta component, sin(theta)
        d_theta = sin(th_a);
        *table = 0;
    } else if (model == 1) {   // HPol: phi component, sin^2(theta)
        d_phi = sin(th_a) * sin(th_a);
        *table = 1;
    } else {                   // LPDA: both components; phase regime by theta (front <= 45 deg < side <= 90 deg < back)
        d_theta = cos(th_a) * sin(ph_a) * cos(th_a / 2);
        d_phi = cos(th_a / 2) * cos(ph_a);
        *table = (th_a <= 45 * 0.017453292519943295) ? 2 : ((th_a <= 90 * 0.017453292519943295) ? 3 : 4);
    }
    *vfac_t = T[0] * d_theta + T[1] * d_phi;
    *vfac_p = T[2] * d_theta + T[3] * d_phi;
}

// ---- tabulated antenna patterns (AntennaPattern._get_antenna_response_vectorized_raw, antennapattern.py:1426-1577) ----
__device__ inline double2 clerp(double x, double x0, double x1, double2 y0, double2 y1)  // interpolate_linear, 'complex'
{
    if (x0 == x1) return y0;
    const double t = (x - x0) / (x1 - x0);
    // y0 + (y1 - y0) * (x - x0) / (x1 - x0), evaluated left to right like the reference
    double2 d = csub(y1, y0);
    d = cscale(d, x - x0);
    d = make_double2(d.x / (x1 - x0), d.y / (x1 - x0));
    (void)t;
    return cadd(y0, d);
}

struct TabAngles {
    bool ok;            // arrival direction inside the table's angular range
    int iT0, iT1, iP0, iP1;
    double theta, phi;  // after the reference's clamping / wrapping
};

__device__ inline TabAngles tab_angles(const AntTabDev& t, double theta, double phi)
{
    TabAngles a;
    const double th_lo = t.th[0], th_hi = t.th[t.nT - 1], ph_lo = t.ph[0], ph_hi = t.ph[t.nP - 1];
    while (phi < ph_lo) phi += 2 * M_PI;
    while (phi > ph_hi) phi -= 2 * M_PI;
    // radiotools.helper.is_equal(a, b, rel_precision = 1e-5)
    auto is_equal = [](double u, double v) { return (u == 0 || v == 0) ? (u == v) : (fabs((u - v) / v) < 1e-5); };
    if (is_equal(theta, th_hi)) theta = th_hi;
    if (is_equal(theta, th_lo)) theta = th_lo;
    a.ok = !(phi < ph_lo || phi > ph_hi || theta < th_lo || theta > th_hi);
    a.theta = theta;
    a.phi = phi;
    a.iT0 = a.iT1 = a.iP0 = a.iP1 = 0;
    if (a.ok) {
        if (th_hi != th_lo) {
            const double u = (theta - th_lo) / (th_hi - th_lo) * (t.nT - 1);
            a.iT0 = (int)floor(u);
            a.iT1 = (int)ceil(u);
        }
        if (ph_hi != ph_lo) {
            const double u = (phi - ph_lo) / (ph_hi - ph_lo) * (t.nP - 1);
            a.iP0 = (int)floor(u);
            a.iP1 = (int)ceil(u);
        }
    }
    return a;
}

// the angular part of the interpolation at frequency node iF: (VEL_theta, VEL_phi) of the table frame
__device__ inline void tab_node(const AntTabDev& t, const TabAngles& a, int iF, double2* vt, double2* vp)
{
    const long base = (long)iF * t.nT * t.nP;
    const long i00 = base + (long)a.iP0 * t.nT + a.iT0, i01 = base + (long)a.iP1 * t.nT + a.iT0;
    const long i10 = base + (long)a.iP0 * t.nT + a.iT1, i11 = base + (long)a.iP1 * t.nT + a.iT1;
    const double p0 = t.ph[a.iP0], p1 = t.ph[a.iP1], t0 = t.th[a.iT0], t1 = t.th[a.iT1];
    *vt = clerp(a.theta, t0, t1, clerp(a.phi, p0, p1, t.vt[i00], t.vt[i01]), clerp(a.phi, p0, p1, t.vt[i10], t.vt[i11]));
    *vp = clerp(a.theta, t0, t1, clerp(a.phi, p0, p1, t.vp[i00], t.vp[i01]), clerp(a.phi, p0, p1, t.vp[i10], t.vp[i11]));
}

// response at frequency f from the per-node angular results (nodes[0..nF) theta, nodes[nF..2nF) phi); zero out of range
__device__ inline void tab_response(const AntTabDev& t, const double2* __restrict__ nodes, double f, double2* vt, double2* vp)
{
    const double f_lo = t.fr[0], f_hi = t.fr[t.nF - 1];
    if (f < f_lo || f > f_hi) {
        *vt = *vp = make_double2(0., 0.);
        return;
    }
    const double u = (f - f_lo) / (f_hi - f_lo) * (t.nF - 1);
    const int i0 = (int)floor(u), i1 = (int)ceil(u);
    *vt = clerp(f, t.fr[i0], t.fr[i1], nodes[i0], nodes[i1]);
    *vp = clerp(f, t.fr[i0], t.fr[i1], nodes[t.nF + i0], nodes[t.nF + i1]);
}

// ---------------------------------------------------------------------------------------------------------
// ray selection: viewing angle + delta_C cut  (simulation.py:187-206)
// ---------------------------------------------------------------------------------------------------------
__device__ inline double viewing_angle(const double sd[3], const double* lv)
{
    double dot = sd[0] * lv[0] + sd[1] * lv[1] + sd[2] * lv[2];
    double n1 = sqrt(sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]);
    double n2 = sqrt(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]);
    double c = dot / (n1 * n2);
    if (c > 1) c = 1;
    if (c < -1) c = -1;
    return acos(c);
}

__device__ inline double n_index_at(double z, const IceConst& m) { return (z <= 0) ? n_of_z(z, m) : 1.; }

__global__ void __launch_bounds__(256)
select_rays_kernel(long n_pairs, int n_ch, const double* __restrict__ vertex, const double* __restrict__ zenith,
                   const double* __restrict__ azimuth, RayRecords rec, IceConst m, double delta_C_cut,
                   int* __restrict__ keep)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    long e = i / n_ch;
    int ns = rec.n_sol[i];
    double axis[3], sd[3];
    sph2cart(zenith[e], azimuth[e], axis);
    for (int d = 0; d < 3; d++) sd[d] = -1 * axis[d];
    double n_index = n_index_at(vertex[3 * e + 2], m);
    double cherenkov = acos(1. / n_index);
    const int S = rec.stride;
    for (int s = 0; s < S; s++) {
        int k = 0;
        if (s < ns) {
            double view = viewing_angle(sd, rec.launch + 3 * (i * S + s));
            k = !(fabs(view - cherenkov) > delta_C_cut);
        }
        keep[i * S + s] = k;
    }
}

// exclusive prefix sum of int flags, three small kernels (tile sums -> scan of tile sums -> per-tile scan)
constexpr int SCAN_TILE = 2048;

__global__ void __launch_bounds__(256) scan_tile_sums_kernel(long n, const int* __restrict__ in, int* __restrict__ tile_sum)
{
    __shared__ int red[256];
    long base = (long)blockIdx.x * SCAN_TILE;
    int s = 0;
    for (int i = threadIdx.x; i < SCAN_TILE; i += 256) {
        long k = base + i;
        if (k < n) s += in[k];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}

// exclusive scan of the tile sums, ONE block (the whole chip waits for it -- an array call makes ten of these scans per station):
// eight consecutive tiles per thread, a shuffle scan inside each wave, the sixteen wave totals through LDS -- two barriers per 8192
// tiles (the Hillis-Steele version of rounds 1-4 took twenty per 1024: 0.4 ms per scan of a 35-station call, 7 % of its kernel time)
__global__ void __launch_bounds__(1024) scan_tile_offsets_kernel(int n_tiles, int* __restrict__ tile_sum)
{
    constexpr int PER = 8;
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < n_tiles; base += 1024 * PER) {
        const int i0 = base + (int)threadIdx.x * PER;
        int v[PER], s_ = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            v[j] = (i0 + j < n_tiles) ? tile_sum[i0 + j] : 0;
            s_ += v[j];
        }
        int incl = s_;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w_ = 0; w_ < 16; w_++) {
            const int q = wsum[w_];
            if (w_ < wave) before += q;
            total += q;
        }
        int excl = carry + before + incl - s_;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (i0 + j < n_tiles) tile_sum[i0 + j] = excl;
            excl += v[j];
        }
        carry += total;
        __syncthreads();   // (wsum is rewritten by the next round)
    }
}

__global__ void __launch_bounds__(256) scan_within_tiles_kernel(long n, const int* __restrict__ in,
                                                                const int* __restrict__ tile_off, int* __restrict__ out)
{
    // each thread owns 8 consecutive elements of the tile
    __shared__ int part[256];
    long base = (long)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int v[8], s = 0;
    for (int j = 0; j < 8; j++) {
        long k = base + j;
        v[j] = (k < n) ? in[k] : 0;
        s += v[j];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int t = ((int)threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = tile_off[blockIdx.x] + part[threadIdx.x] - s;
    for (int j = 0; j < 8; j++) {
        long k = base + j;
        if (k < n) out[k] = run;
        run += v[j];
    }
}

// keep flags + exclusive scan -> list of kept slots (ordered: event, channel, solution)
__global__ void scatter_slots_kernel(long n_slots, const int* __restrict__ keep, const int* __restrict__ offset,
                                     int* __restrict__ ray_slot)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    if (keep[i]) ray_slot[offset[i]] = (int)i;
}

// Askaryan constants per ray: parametrizations.py:110-275 up to the frequency dependence
__device__ inline AskaryanConst askaryan_setup(int model, double energy, double theta, int shower_type,
                                               double n_index, double R, double k_L_in)
{
    AskaryanConst a;
    a.model = model;
    a.had = (shower_type == 0);
    const double eV = 1., MeV = 1e6, TeV = 1e12, cm = 0.01, MHz = 1e-3, GHz = 1., deg = 0.017453292519943295;
    const double g = 6.241509744511525e+33;  // NuRadioReco/utilities/units.py gram
    if (model == 0) {  // parametrizations.py:110-218
        const double E_C = 73.1 * MeV;
        const double rho = 0.924 * g / (cm * cm * cm);
        const double X_0 = 36.08 * g / (cm * cm);
        const double R_M = 10.57 * g / (cm * cm);
        const double c = 0.299792458;
        double k_E_bar, k_L, k_R_bar;
        double lE = log10(energy / eV);
        if (a.had) {
            double k_E_0 = 4.13e-16 * 1. / cm / (MHz * MHz);
            k_E_bar = k_E_0 * tanh((lE - 10.60) / 2.54);
            k_L = 31.25 * pow(energy / (1.e15 * eV), 3.01e-2);
            k_R_bar = 2.73 + tanh((12.92 - lE) / 1.72);
            a.beta = 2.57;
        } else {
            k_E_bar = 4.65e-16 * 1. / cm / (MHz * MHz);
            k_L = k_L_in;
            k_R_bar = 1.54;
            a.beta = 2.74;
        }
        a.a_pref = k_E_bar * energy / E_C * X_0 / rho * sin(theta);
        double nu_L = rho / k_L / X_0;
        double q = fabs(1 - n_index * cos(theta));
        const double cher_cut = 1.e-8;
        if (q < cher_cut) nu_L *= c / cher_cut;
        else nu_L *= c / q;
        a.nu_L = nu_L;
        a.nu_R = rho / k_R_bar / R_M * c / sqrt(n_index * n_index - 1);
        a.alpha = 1.27;
        a.scale = R;
        a.ln_nu_L = log(a.nu_L);
        a.ln_nu_R = log(a.nu_R);
        a.cL = exp(-a.beta * a.ln_nu_L);
        a.cR = exp(-a.alpha * a.ln_nu_R);
        a.pref2 = 0.5 * a.a_pref / a.scale;
    } else if (model == 1) {  // Alvarez2000, parametrizations.py:220-275
        a.cher = acos(1. / n_index);
        a.theta = theta;
        const double Elpm = 2e15 * eV;
        double eps = log10(energy / TeV);
        double dth;  // angular width * f / (500 MHz)  [rad]
        if (!a.had) {
            dth = 2.7 * deg * 500 * MHz * pow(Elpm / (0.14 * energy + Elpm), 0.3);
            a.scale = 1.;
        } else {
            double dd = 0;
            if (eps >= 0 && eps <= 2) dd = 500 * MHz * (2.07 - 0.33 * eps + 7.5e-2 * eps * eps) * deg;
            else if (eps > 2 && eps <= 5) dd = 500 * MHz * (1.74 - 1.21e-2 * eps) * deg;
            else if (eps > 5 && eps <= 7) dd = 500 * MHz * (4.23 - 0.785 * eps + 5.5e-2 * eps * eps) * deg;
            else if (eps > 7) dd = 500 * MHz * (4.23 - 0.785 * 7 + 5.5e-2 * 49) * (1 + (eps - 7) * 0.075) * deg;
            dth = dd;
            double f_eps = -1.27e-2 - 4.76e-2 * (eps + 3);
            f_eps += -2.07e-3 * (eps + 3) * (eps + 3) + 0.52 * sqrt(eps + 3);
            a.scale = (dd != 0) ? f_eps : 0.;
        }
        a.dth = dth;
        a.f0 = 1.15 * GHz;
        a.a_pref = 2.53e-7 * energy / TeV / MHz * (sin(theta) / sin(a.cher)) / R;
    } else {  // ZHS1992, parametrizations.py:92-108
        a.cher = acos(1. / n_index);
        a.theta = theta;
        a.a_pref = 1.1e-7 * energy / TeV / R / MHz;
    }
    return a;
}

// ---------------------------------------------------------------------------------------------------------
// per-ray parameters
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ray_setup_kernel(int n_rays, int n_ch, const int* __restrict__ ray_slot, const double* __restrict__ vertex,
                 const double* __restrict__ zenith, const double* __restrict__ azimuth, RayRecords rec, IceConst m,
                 StationDev st, RayWork w, EventIn evin, int ask_model, const int* __restrict__ foc_n_sol,
                 const double* __restrict__ foc_launch, double foc_dz, double foc_limit, double refl_coefficient,
                 double refl_phase, double pol_ephi)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int slot = ray_slot[r];
    long pair = slot / rec.stride;
    long e = pair / n_ch;
    int ch = (int)(pair - e * n_ch);
    double axis[3], sd[3];
    sph2cart(zenith[e], azimuth[e], axis);
    for (int d = 0; d < 3; d++) sd[d] = -1 * axis[d];
    const double* lv = rec.launch + 3 * (long)slot;
    const double* rv = rec.receive + 3 * (long)slot;
    w.ev[r] = (int)e;
    w.ch[r] = ch;
    w.sol[r] = slot % rec.stride;
    w.view[r] = viewing_angle(sd, lv);
    w.n_index[r] = n_index_at(vertex[3 * e + 2], m);
    w.R[r] = rec.D[slot];
    w.t0[r] = (evin.vertex_time ? evin.vertex_time[e] : 0.) + rec.T[slot] - 0.5 * st.N / st.fs;  // simulation.py:259-268
    w.C0[r] = rec.C0[slot];
    // polarisation = l x (s x l), normalised, in the on-sky basis of the launch direction (simulation.py:816-819)
    double sxl[3] = {sd[1] * lv[2] - sd[2] * lv[1], sd[2] * lv[0] - sd[0] * lv[2], sd[0] * lv[1] - sd[1] * lv[0]};
    double pol[3] = {lv[1] * sxl[2] - lv[2] * sxl[1], lv[2] * sxl[0] - lv[0] * sxl[2], lv[0] * sxl[1] - lv[1] * sxl[0]};
    double pn = sqrt(pol[0] * pol[0] + pol[1] * pol[1] + pol[2] * pol[2]);
    for (int d = 0; d < 3; d++) pol[d] /= pn;
    double lz, la, Mx[9], po[3];
    cart2sph(lv, &lz, &la);
    onsky_matrix(lz, la, Mx);
    mat3vec(Mx, pol, po);
    w.pol_theta[r] = po[1];
    w.pol_phi[r] = po[2];
    if (!isnan(pol_ephi)) {   // signal.polarization = 'custom' (simulation.py:821-825)
        const double et = sqrt(1 - pol_ephi * pol_ephi);
        const double nrm = sqrt(0 * 0 + et * et + pol_ephi * pol_ephi);
        w.pol_theta[r] = et / nrm;
        w.pol_phi[r] = pol_ephi / nrm;
    }
    // arrival direction (simulation.py:270)
    double zen, az;
    cart2sph(rv, &zen, &az);
    w.zen[r] = zen;
    w.az[r] = az;
    // Fresnel reflection off the surface, n_1 = n(-1 cm), n_2 = 1 (analyticraytracing.py:2990-2997,
    // geometryUtilities.py:208-263: conjugated coefficients, complex under total internal reflection)
    double ra = rec.refl_angle[slot];
    double2 rth = make_double2(1., 0.), rph = make_double2(1., 0.);
    if (!isnan(ra)) {
        double n1 = n_of_z(-0.01, m);
        double n = 1. / n1;
        double n2 = n * n;
        double sa = sin(ra), ca = cos(ra);
        double arg = n2 - sa * sa;
        double2 s = (arg >= 0) ? make_double2(sqrt(arg), 0.) : make_double2(0., sqrt(-arg));
        double2 num = make_double2(n2 * ca - s.x, -s.y), den = make_double2(n2 * ca + s.x, s.y);
        double dd = den.x * den.x + den.y * den.y;
        double2 q = make_double2((num.x * den.x + num.y * den.y) / dd, (num.y * den.x - num.x * den.y) / dd);
        rth = cconj(q);
        num = make_double2(ca - s.x, -s.y);
        den = make_double2(ca + s.x, s.y);
        dd = den.x * den.x + den.y * den.y;
        q = make_double2((num.x * den.x + num.y * den.y) / dd, (num.y * den.x - num.x * den.y) / dd);
        rph = cconj(q);
    }
    if (rec.reflection) {
        // paths with bottom reflections (analyticraytracing.py:2966-3009): one Fresnel factor per path segment that reflects at
        // the surface, and per bottom reflection the layer's coefficient and phase shift on both components
        int n_surf = __popc((unsigned)rec.surface_mask[slot]);
        double2 pt = make_double2(1., 0.), pp = make_double2(1., 0.);
        for (int q_ = 0; q_ < n_surf; q_++) { pt = cmul(pt, rth); pp = cmul(pp, rph); }
        const int n_refl = rec.reflection[slot];
        if (n_refl > 0) {
            const double coef = pow(refl_coefficient, (double)n_refl);
            const double phase = fmod(n_refl * refl_phase, 2 * M_PI);
            double sn, cs;
            sincos(phase, &sn, &cs);
            const double2 f = make_double2(coef * cs, coef * sn);
            pt = cmul(pt, f);
            pp = cmul(pp, f);
        }
        rth = pt;
        rph = pp;
    }
    w.r_theta[r] = rth;
    w.r_phi[r] = rph;
    double T[4], th_a, ph_a;
    antenna_frame(zen, az, st.rot + 9 * ch, st.rot_inv + 9 * ch, T, &th_a, &ph_a);
    for (int c = 0; c < 4; c++) w.vel_T[4 * (long)r + c] = T[c];
    w.theta_ant[r] = th_a;
    w.phi_ant[r] = ph_a;
    if (st.ant_model[ch] == 3) {  // tabulated pattern: frequency-dependent direction response, evaluated per ray later
        w.vfac_t[r] = w.vfac_p[r] = 0.;
        w.tab[r] = 0;
    } else {
        antenna_factors(st.ant_model[ch], T, th_a, ph_a, &w.vfac_t[r], &w.vfac_p[r], &w.tab[r]);
    }
    w.slot[r] = slot;
    AskaryanConst ac = askaryan_setup(ask_model, evin.energy[e], w.view[r], evin.shower_type[e], w.n_index[r], w.R[r],
                                      evin.k_L[e]);
    w.focus[r] = 1.;
    if (foc_n_sol) {
        // ray_tracing.get_focusing, numerical branch (analyticraytracing.py:2778-2888): launch angle of the same solution
        // of the trace to the receiver moved by foc_dz (second ray-tracing pass, tables foc_*)
        const double vz = vertex[3 * e + 2], cz = st.pos[3 * ch + 2];
        const double rec_ang = acos(-rv[2] / sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]));
        const double lau_ang = acos(lv[2] / sqrt(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]));
        double f = 1.0;
        if (slot % rec.stride < foc_n_sol[pair]) {
            const double* l1 = foc_launch + 3 * (long)slot;
            const double lau_ang1 = acos(l1[2] / sqrt(l1[0] * l1[0] + l1[1] * l1[1] + l1[2] * l1[2]));
            const double dzz = (cz + foc_dz) - cz;
            const double D = rec.D[slot];
            f = sqrt(D / sin(rec_ang) * fabs((lau_ang1 - lau_ang) / dzz));
            const double dx = st.pos[3 * ch] - vertex[3 * e], dy = st.pos[3 * ch + 1] - vertex[3 * e + 1], dz = cz - vz;
            const double radius = sqrt(dx * dx + dy * dy + dz * dz);
            const double sin_theta = sqrt(dx * dx + dy * dy) / radius;
            f *= sqrt((D * sin(lau_ang)) / (radius * sin_theta));
        }
        if (f > foc_limit) f = foc_limit;
        f *= sqrt(n_index_at(vz, m) / n_index_at(cz, m));
        ac.a_pref *= f;   // every parametrisation is linear in a_pref: spec[1:] *= focusing (:3011-3016)
        ac.pref2 *= f;
        w.focus[r] = f;
    }
    w.ask[r] = ac;
}

// integration limits for the attenuation kernel from the ray records
__global__ void __launch_bounds__(256)
ray_limits_from_slots_kernel(int n_rays, int n_ch, const int* __restrict__ ray_slot, const double* __restrict__ vertex,
                             const double* __restrict__ chan_pos, RayRecords rec, IceConst m, double* __restrict__ zint)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int slot = ray_slot[r];
    long pair = slot / rec.stride;
    long e = pair / n_ch;
    int ch = (int)(pair - e * n_ch);
    if (rec.reflection && rec.reflection[slot] > 0) {
        // path with bottom reflections: several segments, integrated separately (launch_attenuation_segments); the depth-binned
        // pruning bound takes no attenuation credit for it (zero path length in every bin -> factor 1)
        zint[3 * (long)r] = zint[3 * (long)r + 1] = zint[3 * (long)r + 2] = 0.;
        return;
    }
    double A[3] = {vertex[3 * e], vertex[3 * e + 1], vertex[3 * e + 2]};
    double B[3] = {chan_pos[3 * ch], chan_pos[3 * ch + 1], chan_pos[3 * ch + 2]};
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
    C0State s = make_c0(rec.C0[slot], m);
    double C1 = C1_of(s, p, m);
    zint[3 * r] = p.z1;
    zint[3 * r + 1] = z_mirrored(p.y2, p.z2, s, C1, p);
    zint[3 * r + 2] = s.z_turn;
}

// ---------------------------------------------------------------------------------------------------------
// Askaryan amplitude: the reference's irfft -> roll(N/2) -> rfft detour (parametrizations.py:208-214,
// askaryan.py:209-213) is the identity spec_k = i * X_k * (-1)^k * sqrt(2) for 0 < k < N/2 and 0 at DC and
// Nyquist, so only the real amplitude X_k is evaluated.
// ---------------------------------------------------------------------------------------------------------

// real amplitude X(f) such that the reference's spectrum bin is i * X * (-1)^k * sqrt(2) (ZHS1992: see phase)
// lnf = ln f: (f / nu)^beta is evaluated as exp(beta (ln f - ln nu)), ln f coming from a per-station table
__device__ inline double askaryan_amplitude(double f, double lnf, const AskaryanConst& a)
{
    if (a.model == 0) {
        double A = a.a_pref * f;
        double d_L = 1 / (1 + exp(a.beta * (lnf - a.ln_nu_L)));
        double d_R = 1 / (1 + exp(a.alpha * (lnf - a.ln_nu_R)));
        double s = A * d_L * d_R;
        s *= 0.5;
        s /= a.scale;
        return s;
    } else if (a.model == 1) {
        if (a.scale == 0.) return 0.;
        double E = a.a_pref * f / a.f0 / (1 + exp(1.44 * (lnf - log(a.f0))));
        double w = (a.theta - a.cher) / (a.dth / f);
        return 0.5 * a.scale * E * exp(-0.6931471805599453 * w * w);
    } else {
        double vv0 = f / 0.5;
        double w = (a.theta - a.cher) / (2.4 * 0.017453292519943295 / vv0);
        return 0.5 * a.a_pref * vv0 / (1 + 0.4 * vv0 * vv0) * exp(-0.5 * w * w);
    }
}

// np.interp(f, xp, fp) on the coarse attenuation grid (analyticraytracing.py:1078).  The grid is a linspace, so
// the segment index is guessed arithmetically and then corrected to the exact xp[lo] <= f < xp[lo+1]
__device__ inline double interp_att(double f, int n, const double* __restrict__ xp, const double* fp)
{
    if (f <= xp[0]) return fp[0];
    if (f >= xp[n - 1]) return fp[n - 1];
    int lo = (int)((f - xp[0]) / (xp[n - 1] - xp[0]) * (n - 1));
    if (lo > n - 2) lo = n - 2;
    if (lo < 0) lo = 0;
    while (lo > 0 && f < xp[lo]) lo--;
    while (lo < n - 2 && f >= xp[lo + 1]) lo++;
    double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (f - xp[lo]) + fp[lo];
}

// the same value with the segment index from the station's table and the segment slopes precomputed per ray
__device__ inline double interp_seg(double f, int lo, int n, const double* xp, const double* fp, const double* slope)
{
    if (f <= xp[0]) return fp[0];
    if (f >= xp[n - 1]) return fp[n - 1];
    return slope[lo] * (f - xp[lo]) + fp[lo];
}

// X(f_k) on the station's N-sample grid: Alvarez2009 from the f^p tables (one division per bin), others directly
__device__ inline double amplitude_bin(int k, double f, const AskaryanConst& a, const StationDev& st)
{
    if (a.model == 0) {
        const int stride = st.N / 2 + 1;
        double x = st.fpow[(a.had ? 0 : stride) + k] * a.cL;   // (f / nu_L)^beta
        double y = st.fpow[2 * stride + k] * a.cR;             // (f / nu_R)^alpha
        return a.pref2 * f / ((1 + x) * (1 + y));
    }
    return askaryan_amplitude(f, st.lnf[k], a);
}

// ---------------------------------------------------------------------------------------------------------
// Building blocks shared by the efield-maximum kernel and the channel kernel (block-cooperative, LDS)
// ---------------------------------------------------------------------------------------------------------
struct RayShared {
    AskaryanConst ask;
    double att[NRHIP_MAX_NFC];
    double slope[NRHIP_MAX_NFC];
    double xp[NRHIP_MAX_NFC];
};

// amp[k] = X_k * att(f_k) for k <= N/2 (0 at k = 0 and N/2): the real, component-independent part of the field.
// rs.ask and rs.att must be set (and synchronised) by the caller.  Returns this thread's partial sum of amp (for the
// sum-of-magnitudes bound on max |E(t)|).
__device__ inline double fill_amplitude(double* amp, const StationDev& st, RayShared& rs)
{
    const int N = st.N, nh = N / 2, n_fc = st.n_fc;
    const double df = 1.0 / (N * (1. / st.fs));
    for (int j = threadIdx.x; j < n_fc; j += blockDim.x) {
        rs.xp[j] = st.fcoarse[j];
        if (j < n_fc - 1) rs.slope[j] = (rs.att[j + 1] - rs.att[j]) / (st.fcoarse[j + 1] - st.fcoarse[j]);
    }
    __syncthreads();
    double part = 0.;
    if (rs.ask.model == 0) {
        // Alvarez2009 from the station's f^p tables; the table entries of the thread's NEXT bin are requested before the current
        // bin is evaluated (the loop has 2 .. 8 iterations per thread and would otherwise wait for L1 / L2 in each)
        const int stride = nh + 1, off_l = rs.ask.had ? 0 : stride;
        int kq = threadIdx.x;
        bool in = kq > 0 && kq < nh;
        double n_pl = in ? st.fpow[off_l + kq] : 0., n_pr = in ? st.fpow[2 * stride + kq] : 0.;
        int n_seg = in ? st.seg[kq] : 0;
        for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
            const bool cur = k > 0 && k < nh;
            const double pl = n_pl, pr = n_pr;
            const int seg = n_seg;
            kq = k + blockDim.x;
            in = kq > 0 && kq < nh;
            if (in) { n_pl = st.fpow[off_l + kq]; n_pr = st.fpow[2 * stride + kq]; n_seg = st.seg[kq]; }
            double v = 0.;
            if (cur) {
                const double f = k * df;
                const double x = pl * rs.ask.cL, y = pr * rs.ask.cR;   // amplitude_bin, model 0
                v = rs.ask.pref2 * f / ((1 + x) * (1 + y)) * interp_seg(f, seg, n_fc, rs.xp, rs.att, rs.slope);
            }
            if (amp) amp[k] = v;
            part += v;
        }
    } else {
        for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
            double v = 0.;
            if (k > 0 && k < nh) {
                double f = k * df;
                v = amplitude_bin(k, f, rs.ask, st) * interp_seg(f, st.seg[k], n_fc, rs.xp, rs.att, rs.slope);
            }
            if (amp) amp[k] = v;
            part += v;
        }
    }
    __syncthreads();
    return part;
}

// block-wide sum / max through LDS (red has blockDim.x doubles)
__device__ inline double block_sum(double v, double* red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}
__device__ inline double block_max(double v, double* red)
{
    // wave-level butterfly, then one LDS word per wave: two barriers instead of log2(blockDim) + 2 (max is exact in any order)
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < nw; i++) r = fmax(r, red[i]);
    __syncthreads();
    return r;
}

// the same with barriers that order LDS only (channel_conv_kernel: global stores of emitted traces stay in flight)
__device__ inline double block_max_lds(double v, double* red)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    double r = red[0];
    for (int i = 1; i < nw; i++) r = fmax(r, red[i]);
    lds_barrier();
    return r;
}

__device__ inline double cabs2(double2 a) { return sqrt(a.x * a.x + a.y * a.y); }

// max |E(t)| <= (fs / sqrt 2) (1 / N) 2 sum_k |G_k|,  |G_k| = sqrt 2 |pol r| amp_k   (triangle inequality on irfft)
// 1 / x for the bound kernels: hardware reciprocal estimate + one Newton step (relative error < 1e-12 for the normal,
// positive arguments it sees) instead of the ~3x longer IEEE division sequence; the sums built from it are inflated by
// BOUND_RCP_SLACK, so they stay upper bounds
#define BOUND_RCP_SLACK (1. + 1e-9)
#define BOUND_F32_SLACK (1. + 5e-4)
__device__ inline double bound_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.), r, r);
}

__device__ inline double efield_bound(double amp_sum, int N, double fs, double cmax)
{
    return (fs / 1.4142135623730951) * (2.0 / N) * (1.4142135623730951 * cmax * amp_sum);
}

// G(k): spectrum bin k (0..N/2) of one on-sky component, optionally with the sub-sample shift phase ramp
// exp(-2 pi i f rem) of BaseTrace.apply_time_shift (base_trace.py:246-276)
// `ramp` (optional, LDS): exp(-2 pi i f_k rem) = ramp[k & 63] * ramp[64 + (k >> 6)] (two-level table filled by
// field_time_domain: 64 + N / 128 + 1 sincospi per ray instead of one sincos per bin)
__device__ inline double2 field_bin(int k, double amp_k, int N, double fs, double pol, double2 rc, double rem,
                                    bool shift, int ask_model, double roll_bins, const double2* ramp = nullptr)
{
    const int nh = N / 2;
    if (k <= 0 || k >= nh) return make_double2(0., 0.);
    double a = amp_k * 1.4142135623730951;
    double2 s;
    if (ask_model == 2) {  // ZHS1992: +90 deg phase, roll by int(2 ns / dt) bins instead of N/2
        double sn, cs;
        sincospi(-2.0 * k * roll_bins / N, &sn, &cs);
        s = make_double2(-a * sn, a * cs);  // i * a * exp(-2 pi i k m / N)
    } else {
        s = make_double2(0., (k & 1) ? -a : a);  // i * a * (-1)^k
    }
    s = cscale(s, pol);
    s = cmul(s, rc);
    if (shift) {
        if (ramp) {
            s = cmul(s, cmul(ramp[k & 63], ramp[64 + (k >> 6)]));
        } else {
            double f = k * (1.0 / (N * (1. / fs)));
            double sn, cs;
            sincos(-2. * M_PI * rem * f, &sn, &cs);
            s = cmul(s, make_double2(cs, sn));
        }
    }
    return s;
}

// y_j = e[2j] + i e[2j+1] (j < N/2) of e = irfft_N(G) * fs / sqrt(2), left in x[nplan_idx(j)] (x in LDS, >= nplan_points(np))
#ifndef NRHIP_EFIELD_FUSE
#define NRHIP_EFIELD_FUSE 3   // efield_max_kernel: 256 threads on N / 2 = 2048 points = 2^3 points per thread and pass
#endif
template <int FUSE = 0>
__device__ inline void field_time_domain(double2* x, const double* amp, int N, const NPlan& np, double fs, double pol,
                                         double2 rc, double rem, bool shift, int ask_model, double roll_bins,
                                         const double2* __restrict__ tw)
{
    const int log2nh = np.log2nh;
    const int nh = N / 2;
#ifdef NRHIP_CONV_TIMING
    if (threadIdx.x == 0) g_ct_mark[blockIdx.x & 1023] = __builtin_amdgcn_s_memtime();
#endif
    // the sub-sample shift's phase ramp exp(-2 pi i f rem), f = k fs / N: w^k = w^(k & 63) * (w^64)^(k >> 6)
    __shared__ double2 s_ramp[64 + FFT_MAX / 64 + 1];   // N / 2 up to FFT_MAX bins
    const double2* ramp = nullptr;
    if (shift && blockDim.x >= 64 + (unsigned)(nh >> 6) + 1) {
        const int t = threadIdx.x;
        if (t < 64 + (nh >> 6) + 1) {
            const double f = (t < 64 ? t : 64 * (t - 64)) * (1.0 / (N * (1. / fs)));
            double sn, cs;
            sincospi(-2. * rem * f, &sn, &cs);
            s_ramp[t] = make_double2(cs, sn);
        }
        __syncthreads();
        ramp = s_ramp;
    }
    // (the twiddle of the thread's NEXT bin is requested before the current one is used: it comes from L1 / L2)
    double2 wnext = (threadIdx.x < (unsigned)nh) ? nplan_w(np, threadIdx.x, tw) : make_double2(1., 0.);
    for (int k = threadIdx.x; k < nh; k += blockDim.x) {
        const double2 wk = wnext;                // exp(-2 pi i k / N)
        if (k + (int)blockDim.x < nh) wnext = nplan_w(np, k + blockDim.x, tw);
        double2 Gk = field_bin(k, amp[k], N, fs, pol, rc, rem, shift, ask_model, roll_bins, ramp);
        double2 Gc = cconj(field_bin(nh - k, amp[nh - k], N, fs, pol, rc, rem, shift, ask_model, roll_bins, ramp));
        double2 ge = cscale(cadd(Gk, Gc), 0.5);
        double2 d = cscale(csub(Gk, Gc), 0.5);
        double2 go = cmul(d, cconj(wk));         // * exp(+2 pi i k / N)
        x[k] = make_double2(ge.x - go.y, ge.y + go.x);  // ge + i go
    }
    __syncthreads();
#ifdef NRHIP_CONV_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_conv_clk[10], __builtin_amdgcn_s_memtime() - g_ct_mark[blockIdx.x & 1023]);
#endif
    // inverse, natural -> bit-reversed (natural for lengths that are no power of two: nplan_idx); scale applied by the reader
    if (log2nh < 0) nplan_fft(x, np, tw, true);
    else if (FUSE > 2) fft_dif_fused_k<(FUSE > 2 ? FUSE : 3)>(x, log2nh, tw, true);
    else fft_dif(x, log2nh, tw, true);
}

// ---------------------------------------------------------------------------------------------------------
// kernel: rigorous upper bound on max |E(t)| per ray WITHOUT attenuation (attenuation factors are <= 1): events whose
// rays all stay below the candidate cut even un-attenuated can never become candidates (simulation.py:283-285), so
// their rays skip the attenuation quadrature and the time-domain transform.  One block per ray.
// ---------------------------------------------------------------------------------------------------------
// margin of the attenuation bound: the quadrature the bound must dominate is QUADPACK's at epsrel = 1e-2 (its estimate may be
// 1 % above... the computed integral at most 1 % below the true one)
#ifndef NRHIP_ATT_BOUND_MARGIN
#define NRHIP_ATT_BOUND_MARGIN 0.95
#endif
__device__ __noinline__ double2 efield_bound_fp64_ray(int N, double fs, int n_fc, const unsigned char* __restrict__ seg,
                                                       const double* __restrict__ fpow, const double* __restrict__ lnf,
                                                       const AskaryanConst& ask, const double* at, const double* at_slope,
                                                       const double* s_xp, int lane);   // (defined with efield_bound_kernel)
#define AB_RT 4  // rays per wave and pass: the frequency-grid tables are loaded once for AB_RT rays
#ifndef AB_G
#define AB_G 4   // bins per group of the two-sided sums
#endif
#define AB_RUN 8 // groups per lane at most (N / 2 <= 2048; longer traces take the plain sums)
#ifndef NRHIP_AB_WAVES
#define NRHIP_AB_WAVES 3
#endif
__global__ void __launch_bounds__(256, NRHIP_AB_WAVES)
amp_bound_kernel(int n_rays, RayWork w, StationDev st, IceConst m, const double* __restrict__ vertex,
                 const double* __restrict__ zint, double* __restrict__ bound, double* __restrict__ max_efield, double cut)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(AB_RT == 4, "two packed pairs of rays");
    __shared__ float4 blen[4][64];      // per wave and depth bin: path length of the tile's four rays inside the bin
    __shared__ float s_binv[63 * 32];   // the depth-bin table (n_fc <= 32; read from HBM otherwise)
    const bool binv_lds = st.n_att_bins > 0 && st.n_fc <= 32;
    if (binv_lds)
        for (int j = threadIdx.x; j < st.n_att_bins * st.n_fc; j += blockDim.x) s_binv[j] = (float)st.att_bin_inv[j];
    // per wave: upper bounds of the coarse attenuation factors and the slopes between them -- single precision, the tile's four rays
    // side by side (one 16-byte read per table and frequency bin), for the FP32 sums; double precision per ray for the general sums
    __shared__ float4 ubf[4][NRHIP_MAX_NFC], slf[4][NRHIP_MAX_NFC];
    __shared__ double ub[4][AB_RT][NRHIP_MAX_NFC];
    __shared__ double ub_slope[4][AB_RT][NRHIP_MAX_NFC];
    __shared__ double s_xp[NRHIP_MAX_NFC];
    __shared__ float s_xp_f[NRHIP_MAX_NFC];
    for (int j = threadIdx.x; j < st.n_fc; j += blockDim.x) { s_xp[j] = st.fcoarse[j]; s_xp_f[j] = (float)st.fcoarse[j]; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nh = st.N / 2, stride = nh + 1;
    // the station's frequency tables at the nodes of the two-sided sums (bin 1 + AB_G j): (segment, f^p hadronic, f^p electromagnetic,
    // f^q) per node, a lane's run of nodes and the first node of the next lane's run side by side (AB_RUN + 1 entries per lane: a
    // stride of 36 words between the lanes, no bank conflicts)
    __shared__ float4 s_node[64 * (AB_RUN + 1)];
    const int n_nodes = (nh - 1 + AB_G - 1) / AB_G, run = (n_nodes + 63) / 64;
    if (run <= AB_RUN)
        for (int q = threadIdx.x; q < 64 * (AB_RUN + 1); q += blockDim.x) {
            const int j = min(run * (q / (AB_RUN + 1)) + q % (AB_RUN + 1), n_nodes - 1), k = 1 + AB_G * j;
            s_node[q] = make_float4(__int_as_float(st.seg[k]), st.fpow_f[k], st.fpow_f[stride + k], st.fpow_f[2 * stride + k]);
        }
    __syncthreads();  // s_binv / s_xp / s_node are filled by all four waves and read by each of them in the first pass
    const double df = 1.0 / (st.N * (1. / st.fs));
    const int per_pass = gridDim.x * 4 * AB_RT;
    const int n_iter = (n_rays + per_pass - 1) / per_pass;
    // Two-sided sums over groups of AB_G bins (see below): only where a group holds at most one coarse frequency and the candidate
    // cut is there to compare with
    bool grouped = cut >= 0. && nh >= 16 * AB_G && run <= AB_RUN;
    for (int j = 0; j + 1 < st.n_fc; j++) grouped = grouped && (s_xp[j + 1] - s_xp[j]) > (AB_G + 1) * df;
    for (int it = 0; it < n_iter; it++) {
        const int rb = __builtin_amdgcn_readfirstlane(((it * gridDim.x + blockIdx.x) * 4 + wv) * AB_RT);
        if (rb >= n_rays) continue;   // (wave-uniform; the tables below are the wave's own)
        // per-ray scalars only (the full AskaryanConst records would cost ~40 VGPRs per ray).  All of the tile's loads are issued
        // before anything waits for one of them: `all2009 && (model == 0)` used to short-circuit into four dependent round trips
        // to memory per tile, on a kernel that holds three waves per SIMD to hide them
        double cL[AB_RT], cR[AB_RT], pf[AB_RT];
        int had[AB_RT], model_or = 0;
        for (int i = 0; i < AB_RT; i++) {
            const AskaryanConst& ai = w.ask[min(rb + i, n_rays - 1)];
            model_or |= ai.model;
            cL[i] = ai.cL; cR[i] = ai.cR; pf[i] = ai.pref2; had[i] = ai.had;
        }
        // the polarisation factor of the ray this lane finishes (lanes < AB_RT), requested here, needed behind the sums
        double cmax_mine = 0.;
        if (lane < AB_RT && rb + lane < n_rays) {
            const int r = rb + lane;
            const double pth = w.pol_theta[r], pph = w.pol_phi[r];
            const double2 rth = w.r_theta[r], rph = w.r_phi[r];
            cmax_mine = fmax(fabs(pth) * cabs2(rth), fabs(pph) * cabs2(rph));
        }
        const bool all2009 = model_or == 0;
        // Alvarez2009 with all scalars comfortably inside the single-precision range: the attenuation bounds and the 2047-term sum in
        // FP32 (twice the VALU rate, packed pairs of rays, 1-instruction reciprocal / exp); every term is within ~1e-5 and the sum of
        // positive terms within 2047 * 6e-8 of the exact one, the result is inflated by BOUND_F32_SLACK and stays an upper bound
        // ((1 + x)(1 + y) cannot overflow: x, y < 1e15)
        bool f32 = all2009;
        for (int i = 0; i < AB_RT; i++)
            f32 = f32 && pf[i] > 1e-25 && pf[i] < 1e25 && cL[i] > 1e-15 && cL[i] < 1e15 && cR[i] > 1e-15 && cR[i] < 1e15;
        // attenuation factor <= exp(-int ds / L) <= exp(-D / L_max(f)); 0.95 covers the reference's 1e-2 quadrature
        // tolerance; linear interpolation of upper bounds bounds the interpolated attenuation
        double ud[AB_RT] = {0., 0., 0., 0.};   // this lane's coarse frequency (lane < n_fc), the tile's rays
        float uf[AB_RT] = {0.f, 0.f, 0.f, 0.f};
        if (st.n_att_bins > 0) {
            // depth-resolved: int ds / L >= sum_b (path length inside depth bin b) * min_bin(1 / L).  The path climbs
            // from z1 to min(z_turn, z2m) and, if it turns, descends again to 2 z_turn - z2m; s(z) is the closed-form
            // path length (analyticraytracing.py:602-689) with n sin(theta) = 1 / C0.  Lane i evaluates both legs at
            // the bin edge -i w; bins the path does not reach contribute 0, parts below the table are ignored.
            // Single precision throughout (hardware exp / log / sqrt / reciprocal: a few ulp each): the bin lengths are good to ~1e-5,
            // their products with the table and the 63-term sums to ~1e-5 as well; the sum is shrunk by 1e-3 to stay a lower bound.
            float bl[AB_RT];
            // (the path parameters of the four rays requested together, ahead of the branches below: wave-uniform addresses)
            double C0v[AB_RT], z1v[AB_RT], z2v[AB_RT], ztv[AB_RT];
#pragma unroll
            for (int i = 0; i < AB_RT; i++) {
                const long rc = min(rb + i, n_rays - 1);
                C0v[i] = w.C0[rc]; z1v[i] = zint[3 * rc]; z2v[i] = zint[3 * rc + 1]; ztv[i] = zint[3 * rc + 2];
            }
#pragma unroll
            for (int i = 0; i < AB_RT; i++) {
                const int r = rb + i;
                float s1 = 0.f, s2 = 0.f;
                if (r < n_rays && lane <= st.n_att_bins) {
                    const double C0 = C0v[i], z1 = z1v[i], z2m = z2v[i], zt = ztv[i];
                    const double beta2d = __builtin_amdgcn_rcp(C0 * C0);   // (hardware reciprocal: ~1e-8, see the precision note above)
                    const float beta2 = (float)beta2d, alpha = (float)(m.n2 - beta2d), sa = __builtin_amdgcn_sqrtf(alpha);
                    const float n_ice = (float)m.n_ice, dn = (float)m.delta_n, z0 = (float)m.z_0, inv_z0 = __builtin_amdgcn_rcpf(z0);
                    const float nsa = n_ice * __builtin_amdgcn_rcpf(sa);
                    const double edge = -lane * st.att_bin_width;
                    const double top1 = fmin(zt, z2m), lo2 = (z2m > zt) ? 2 * zt - z2m : zt;
                    const float zz[2] = {(float)fmin(fmax(edge, z1), top1), (float)fmin(fmax(edge, lo2), zt)};
                    float sv[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const float nz = n_ice - dn * __expf(zz[q] * inv_z0);
                        const float gam = fmaxf(0.f, nz * nz - beta2);
                        const float l1 = __builtin_amdgcn_sqrtf(alpha * gam) + n_ice * nz - beta2, l2 = __builtin_amdgcn_sqrtf(gam) + nz;
                        sv[q] = nsa * (zz[q] - z0 * __logf(l1)) + z0 * __logf(l2);
                    }
                    s1 = sv[0];
                    s2 = sv[1];
                }
                const float d1 = s1 - __shfl_down(s1, 1), d2 = s2 - __shfl_down(s2, 1);
                bl[i] = (lane < st.n_att_bins) ? fmaxf(0.f, d1) + fmaxf(0.f, d2) : 0.f;
            }
            blen[wv][lane] = make_float4(bl[0], bl[1], bl[2], bl[3]);
            wave_lds_sync();
            // sum_b length_b / L_b per coarse frequency.  Table in LDS (n_fc <= 32): the two halves of the wave take half of the
            // depth bins each (lane = 32 half + frequency) and fold; otherwise lanes over the frequencies, table from memory.
            f2 I01 = f2{0.f, 0.f}, I23 = I01;
            if (binv_lds) {
                // only the depth bins one of the tile's rays crosses (lane = bin above: a ballot gives the range)
                const unsigned long long crossed = __ballot(bl[0] + bl[1] + bl[2] + bl[3] > 0.f);
                const int b_lo = crossed ? __builtin_ctzll(crossed) : 0, b_hi = crossed ? 64 - __builtin_clzll(crossed) : 0;
                const int fc = lane & 31, nb_half = (b_hi - b_lo + 1) >> 1;
                const int b0 = b_lo + (lane >> 5) * nb_half, b1 = min(b_hi, b0 + nb_half);
                if (fc < st.n_fc)
                    for (int b = b0; b < b1; b++) {
                        const float t = s_binv[b * st.n_fc + fc];
                        const float4 q = blen[wv][b];
                        I01 += f2{q.x, q.y} * t;
                        I23 += f2{q.z, q.w} * t;
                    }
                I01 = f2{wave_fold32(I01.x, I01.x), wave_fold32(I01.y, I01.y)};
                I23 = f2{wave_fold32(I23.x, I23.x), wave_fold32(I23.y, I23.y)};
            } else if (lane < st.n_fc) {
                for (int b = 0; b < st.n_att_bins; b++) {
                    const float t = (float)st.att_bin_inv[b * st.n_fc + lane];
                    const float4 q = blen[wv][b];
                    I01 += f2{q.x, q.y} * t;
                    I23 += f2{q.z, q.w} * t;
                }
            }
            if (lane < st.n_fc) {
                const float I[AB_RT] = {I01.x, I01.y, I23.x, I23.y};
                for (int i = 0; i < AB_RT; i++) {
                    if (rb + i >= n_rays) continue;
                    // (v_exp_f32 on an argument of up to ~1e2: a few 1e-6 of the value; inflated to stay above it)
                    if (f32) uf[i] = __expf(-(float)(NRHIP_ATT_BOUND_MARGIN * (1 - 1e-3)) * I[i]) * (1.f + 2e-5f);
                    else ud[i] = exp(-NRHIP_ATT_BOUND_MARGIN * (1 - 1e-3) * (double)I[i]);
                }
            }
        } else if (lane < st.n_fc) {
            for (int i = 0; i < AB_RT; i++) {
                const int r = rb + i;
                double u = 0.;
                if (r < n_rays) {
                    double zlo = fmin(vertex[3 * (long)w.ev[r] + 2], st.pos[3 * w.ch[r] + 2]);  // deepest point of the path
                    u = (zlo >= -st.att_bound_depth) ? exp(-NRHIP_ATT_BOUND_MARGIN * w.R[r] * st.inv_lmax[lane]) : 1.;
                }
                ud[i] = u;
                uf[i] = (float)u * (1.f + 2e-7f);
            }
        }
        if (f32) {
            if (lane < st.n_fc) ubf[wv][lane] = make_float4(uf[0], uf[1], uf[2], uf[3]);
            wave_lds_sync();
            if (lane < st.n_fc) {   // slope towards the next coarse frequency (0 behind the last)
                float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < st.n_fc - 1) {
                    const float4 nx = ubf[wv][lane + 1];
                    const float inv = __builtin_amdgcn_rcpf(s_xp_f[lane + 1] - s_xp_f[lane]);
                    sl = make_float4((nx.x - uf[0]) * inv, (nx.y - uf[1]) * inv, (nx.z - uf[2]) * inv, (nx.w - uf[3]) * inv);
                }
                slf[wv][lane] = sl;
            }
        } else {
            if (lane < st.n_fc)
                for (int i = 0; i < AB_RT; i++) ub[wv][i][lane] = ud[i];
            wave_lds_sync();
            for (int i = 0; i < AB_RT; i++)
                if (lane < st.n_fc - 1)
                    ub_slope[wv][i][lane] = (ub[wv][i][lane + 1] - ud[i]) / (s_xp[lane + 1] - s_xp[lane]);
        }
        wave_lds_sync();   // (every table behind this point is the wave's own: the waves of a block run apart)
        // (A two-level scheme -- a 64-term bound first, the full sum only for tiles near the candidate cut -- was built and measured
        // in round 4 and not kept: 38 % of the tiles decided early, the kernel slower)
        double part[AB_RT];
        for (int i = 0; i < AB_RT; i++) part[i] = 0.;
        const double x_first = s_xp[0], x_last = s_xp[st.n_fc - 1], dx_last = x_last - s_xp[st.n_fc - 2];
        // Two-sided sums first (round 4).  The summand is pf f r(f) U(f) with r = 1 / ((1 + cL f^p)(1 + cR f^q)) falling in f (the
        // f^p tables rise) and U the piecewise-linear attenuation bound, so the bins k_j .. k_j + n - 1 of a group (k_j = 1 + AB_G j)
        // add up to between n pf f(k_j) r(k_{j+1}) U_min and n pf f(k_j + n - 1) r(k_j) U_max, U_min / U_max over the two nodes and the
        // coarse frequency between them, if any: ONE evaluation per group of AB_G bins.  The bound is only ever COMPARED with the
        // candidate cut (event_possible_kernel): unless a ray of the tile has the cut between its two sums (a few per cent of the
        // tiles; the 2047-term sum below decides those) the comparison is made with the upper sum -- the same flags, the value
        // stored still an upper bound.
        bool decided = false;
        if (f32 && grouped) {
            f2 cL2[2], cR2[2], pf2[2], hm2[2], up2[2], lo2[2];
            for (int j = 0; j < 2; j++) {
                cL2[j] = f2{(float)cL[2 * j], (float)cL[2 * j + 1]};
                cR2[j] = f2{(float)cR[2 * j], (float)cR[2 * j + 1]};
                pf2[j] = f2{(float)pf[2 * j], (float)pf[2 * j + 1]};
                hm2[j] = f2{had[2 * j] ? 1.f : 0.f, had[2 * j + 1] ? 1.f : 0.f};
                up2[j] = lo2[j] = f2{0.f, 0.f};
            }
            const f2 one2 = f2{1.f, 1.f};
            const float xf_first = (float)x_first, xf_last = (float)x_last, dxf_last = (float)dx_last, dff = (float)df;
            // a lane owns a run of consecutive nodes (and evaluates the first node of the next lane's run itself): no exchange between lanes
            const int j0 = run * lane;
            struct Node { f2 r[2], U[2]; int lo; };
            auto eval = [&](int j) {   // node j at bin 1 + AB_G j: r and U of the four rays, the (clamped) segment; zeros behind the last node
                Node o;
                const int k = 1 + AB_G * min(j, n_nodes - 1);
                const float f = k * dff;
                const float4 nd = s_node[lane * (AB_RUN + 1) + (j - j0)];
                int lo = __float_as_int(nd.x);
                const float ph = nd.y, pe = nd.z, pr = nd.w;
                float dx = f - s_xp_f[lo];
                if (f <= xf_first) { lo = 0; dx = 0.f; }
                if (f >= xf_last) { lo = st.n_fc - 2; dx = dxf_last; }
                dx = fmaxf(dx, 0.f);
                const float dp = ph - pe;
                const float4 sl4 = slf[wv][lo], u4 = ubf[wv][lo];
                const float live = j < n_nodes ? 1.f : 0.f;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const f2 psel = hm2[q] * dp + pe;
                    const f2 den = (one2 + psel * cL2[q]) * (one2 + pr * cR2[q]);
                    o.r[q] = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)} * live;
                    const f2 sl = q ? f2{sl4.z, sl4.w} : f2{sl4.x, sl4.y};
                    const f2 u0 = q ? f2{u4.z, u4.w} : f2{u4.x, u4.y};
                    const f2 uv = sl * dx + u0;
                    o.U[q] = f2{fmaxf(uv.x, 0.f), fmaxf(uv.y, 0.f)};
                }
                o.lo = lo;
                return o;
            };
            Node cur = eval(j0);
            for (int t = 0; t < run; t++) {
                const int j = j0 + t;
                const Node nxt = eval(j + 1);
                // group j: bins k .. k + n - 1 (n = 0 behind the last node).  With a node behind it: both sums; the last group of all
                // (a few bins at the Nyquist end): the upper sum with U <= 1 (every attenuation bound is, up to its 2e-5 of slack)
                const int k = 1 + AB_G * j, n = max(0, min(AB_G, nh - k));
                const bool has_next = j + 1 < n_nodes;
                const float w_up = (float)n * ((k + n - 1) * dff) * (1.f + 1e-6f), w_lo = has_next ? (float)n * (k * dff) * (1.f - 1e-6f) : 0.f;
                // the coarse frequency between the two nodes, if the segment changes: its value is the table's
                const float4 ia4 = ubf[wv][min(cur.lo + 1, st.n_fc - 1)];
                const bool inner = nxt.lo != cur.lo;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const f2 ia = q ? f2{ia4.z, ia4.w} : f2{ia4.x, ia4.y};
                    const f2 mid = inner ? ia : cur.U[q];
                    f2 umax = f2{fmaxf(fmaxf(cur.U[q].x, nxt.U[q].x), mid.x), fmaxf(fmaxf(cur.U[q].y, nxt.U[q].y), mid.y)};
                    const f2 umin = f2{fminf(fminf(cur.U[q].x, nxt.U[q].x), mid.x), fminf(fminf(cur.U[q].y, nxt.U[q].y), mid.y)};
                    if (!has_next) umax = f2{1.f + 3e-5f, 1.f + 3e-5f};
                    up2[q] += (pf2[q] * w_up) * cur.r[q] * umax;
                    lo2[q] += (pf2[q] * w_lo) * nxt.r[q] * umin;
                }
                cur = nxt;
            }
            float upt[AB_RT] = {up2[0].x, up2[0].y, up2[1].x, up2[1].y}, lot[AB_RT] = {lo2[0].x, lo2[0].y, lo2[1].x, lo2[1].y};
            double up_mine = 0., lo_mine = 0.;
#if NRHIP_WAVE_DPP
            {   // the eight wave sums: 6 half-wave / row swaps + DPP adds (wave_reduce.h) instead of 48 LDS-crossbar shuffles
                static_assert(AB_RT == 4, "wave_fold4");
                const float ua = wave_fold4(upt[0], upt[1], upt[2], upt[3]), la = wave_fold4(lot[0], lot[1], lot[2], lot[3]);
                const float us[AB_RT] = {wave_row_value<0>(ua), wave_row_value<1>(ua), wave_row_value<2>(ua), wave_row_value<3>(ua)};
                const float ls[AB_RT] = {wave_row_value<0>(la), wave_row_value<1>(la), wave_row_value<2>(la), wave_row_value<3>(la)};
                for (int i = 0; i < AB_RT; i++)
                    if (lane == i) { up_mine = us[i]; lo_mine = ls[i]; }
            }
#else
            for (int i = 0; i < AB_RT; i++) {
                float a = upt[i], b = lot[i];
                for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (lane == i) { up_mine = a; lo_mine = b; }
            }
#endif
            // what the 2047-term sum would hand to efield_bound lies between these two (its FP32 rounding: 2047 x 6e-8, ours: 1e-5)
            const double x_up = (up_mine * BOUND_F32_SLACK) * BOUND_F32_SLACK + 1e-30;
            const double x_lo = (lo_mine * (1. - 5e-4)) * BOUND_F32_SLACK;
            const int r = rb + lane;
            bool open = false;
            double b_up = 0.;
            if (lane < AB_RT && r < n_rays) {
                const double cmax = cmax_mine;
                b_up = efield_bound(x_up * BOUND_RCP_SLACK, st.N, st.fs, cmax);
                const double b_lo = efield_bound(x_lo * BOUND_RCP_SLACK, st.N, st.fs, cmax);
                open = (b_up * (1 + 1e-6) > cut) && !(b_lo * (1 + 1e-6) > cut);
            }
#ifdef NRHIP_CONV_TIMING
            if (lane == 0) { atomicAdd(&g_conv_clk[13], 1ULL); if (__ballot(open) == 0ull) atomicAdd(&g_conv_clk[14], 1ULL); }
#endif
            if (__ballot(open) == 0ull) {
                decided = true;
                if (lane < AB_RT && r < n_rays) {
                    bound[r] = b_up;
                    max_efield[r] = -b_up;
                }
            }
        }
        if (decided) {
            wave_lds_sync();
            continue;
        }
        if (f32) {
            // the four rays of the tile as two packed pairs (v_pk_mul / v_pk_fma_f32: two single-precision operations per lane and
            // instruction; only the reciprocal is per ray)
            f2 cL2[2], cR2[2], pf2[2], hm2[2], acc2[2];
            for (int j = 0; j < 2; j++) {
                cL2[j] = f2{(float)cL[2 * j], (float)cL[2 * j + 1]};
                cR2[j] = f2{(float)cR[2 * j], (float)cR[2 * j + 1]};
                pf2[j] = f2{(float)pf[2 * j], (float)pf[2 * j + 1]};
                hm2[j] = f2{had[2 * j] ? 1.f : 0.f, had[2 * j + 1] ? 1.f : 0.f};
                acc2[j] = f2{0.f, 0.f};
            }
            const f2 one2 = f2{1.f, 1.f}, zero2 = f2{0.f, 0.f};
            const float xf_first = (float)x_first, xf_last = (float)x_last, dxf_last = (float)dx_last, dff = (float)df;
            // the station tables of the NEXT bin are requested before the current one is evaluated (L1 / L2 latency off the path)
            int kq = 1 + lane;
            int n_lo = (kq < nh) ? st.seg[kq] : 0;
            float n_ph = (kq < nh) ? st.fpow_f[kq] : 0.f, n_pe = (kq < nh) ? st.fpow_f[stride + kq] : 0.f,
                  n_pr = (kq < nh) ? st.fpow_f[2 * stride + kq] : 0.f;
            for (int k = 1 + lane; k < nh; k += 64) {
                const float f = k * dff;
                int lo = n_lo;
                const float ph = n_ph, pe = n_pe, pr = n_pr;
                kq = k + 64;
                if (kq < nh) {
                    n_lo = st.seg[kq];
                    n_ph = st.fpow_f[kq];
                    n_pe = st.fpow_f[stride + kq];
                    n_pr = st.fpow_f[2 * stride + kq];
                }
                float dx = f - s_xp_f[lo];
                if (f <= xf_first) { lo = 0; dx = 0.f; }
                if (f >= xf_last) { lo = st.n_fc - 2; dx = dxf_last; }
                dx = fmaxf(dx, 0.f);
                const float dp = ph - pe;
                const float4 sl4 = slf[wv][lo], u4 = ubf[wv][lo];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const f2 psel = hm2[j] * dp + pe;   // hadronic: f^2.57 table, electromagnetic: f^2.74 (exactly one of the two)
                    const f2 den = (one2 + psel * cL2[j]) * (one2 + pr * cR2[j]);
                    const f2 rc = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                    const f2 sl = j ? f2{sl4.z, sl4.w} : f2{sl4.x, sl4.y};
                    const f2 u0 = j ? f2{u4.z, u4.w} : f2{u4.x, u4.y};
                    const f2 uv = __builtin_elementwise_max(sl * dx + u0, zero2);
                    acc2[j] += (pf2[j] * f) * rc * uv;
                }
            }
            float pf32[AB_RT] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y};
            for (int i = 0; i < AB_RT; i++) part[i] = (double)pf32[i] * BOUND_F32_SLACK + 1e-30;  // + what FP32 may have flushed to zero
        } else {
            // rare (another emission model, or parameters outside the FP32 range): FP64, ray by ray, out of line (the sums of
            // efield_bound_kernel with the bounds in place of the attenuation) -- inlined, these loops cost the common path registers
            for (int i = 0; i < AB_RT; i++)
                part[i] = efield_bound_fp64_ray(st.N, st.fs, st.n_fc, st.seg, st.fpow, st.lnf, w.ask[min(rb + i, n_rays - 1)], ub[wv][i],
                                                ub_slope[wv][i], s_xp, lane).x;
        }
        // the four sums: wave totals by shuffles, then lane i < AB_RT finishes ray i (one copy of the epilogue instead of four)
        double mine = 0.;
#if NRHIP_WAVE_DPP
        {
            const double pa = wave_fold4(part[0], part[1], part[2], part[3]);
            const double ps[AB_RT] = {wave_row_value<0>(pa), wave_row_value<1>(pa), wave_row_value<2>(pa), wave_row_value<3>(pa)};
            for (int i = 0; i < AB_RT; i++)
                if (lane == i) mine = ps[i];
        }
#else
        for (int i = 0; i < AB_RT; i++) {
            double pt = part[i];
            for (int off = 32; off > 0; off >>= 1) pt += __shfl_xor(pt, off);
            if (lane == i) mine = pt;
        }
#endif
        {
            const int r = rb + lane;
            if (lane < AB_RT && r < n_rays) {
                double b = efield_bound(mine * BOUND_RCP_SLACK, st.N, st.fs, cmax_mine);
                bound[r] = b;
                max_efield[r] = -b;  // "not evaluated, at most b" until efield_max_kernel overwrites it
            }
        }
        wave_lds_sync();   // (every table behind this point is the wave's own: the waves of a block run apart)
    }
}

// ray range of every event group: rays are ordered by shower, the showers of a group are consecutive.
// group_begin == nullptr: every shower is its own group.
__global__ void group_ray_range_kernel(int n_groups, const int* __restrict__ group_begin, int n_ch,
                                       const int* __restrict__ slot_offset, int* __restrict__ grp_ray, int stride)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    long sh = group_begin ? group_begin[g] : g;
    grp_ray[g] = slot_offset[sh * n_ch * stride];
}

// per event: can any ray exceed the cut?  (1 + 1e-6 absorbs rounding of the bound and of exp(-integral) <= 1)
// A wave takes 64 consecutive event groups and walks THEIR rays -- a contiguous range -- with its lanes (coalesced); the group of a
// ray comes from a six-step search over the wave's 65 range starts (lane shuffles).  (Rounds 1-4: one thread per group walking its
// rays one after the other -- uncoalesced, 0.7 ... 1 ms per station call of an array for a few MB.)
struct GroupWave {
    int off;        // this lane's group: first ray
    int r_begin, r_end;
    __device__ GroupWave(const int* __restrict__ slot_offset, int g0, int n_groups)
    {
        const int lane = threadIdx.x & 63;
        off = slot_offset[min(g0 + lane, n_groups)];
        r_begin = __shfl(off, 0);
        r_end = slot_offset[min(g0 + 64, n_groups)];
    }
    // index (0 .. 63) of the group that holds ray r (r_begin <= r < r_end): the last lane whose range starts at or before r
    __device__ int group_of(int r) const
    {
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const int cand = lo + step;
            const int o = __shfl(off, cand & 63);
            if (cand < 64 && o <= r) lo = cand;
        }
        return lo;
    }
};

__global__ void __launch_bounds__(256)
event_possible_kernel(int n_events, int n_ch, const int* __restrict__ slot_offset, const double* __restrict__ bound,
                      double min_efield, int* __restrict__ ray_active, int own_only)
{
    __shared__ int s_any[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + wv) * 64;
    if (g0 >= n_events) return;
    const GroupWave gw(slot_offset, g0, n_events);
    if (own_only) {
        // two-stage attenuation: first only the rays that could make their event a candidate on their own; the other rays of the
        // events that did become candidates follow after the candidate cut (follower_flags_kernel)
        for (int r = gw.r_begin + lane; r < gw.r_end; r += 64) ray_active[r] = (bound[r] * (1 + 1e-6) > min_efield) ? 1 : 0;
        return;
    }
    s_any[wv][lane] = 0;
    wave_lds_sync();
    for (int base = gw.r_begin; base < gw.r_end; base += 64) {   // (all lanes take part in the shuffles)
        const int r = base + lane;
        const bool in = r < gw.r_end;
        const int j = gw.group_of(in ? r : gw.r_begin);
        if (in && bound[r] * (1 + 1e-6) > min_efield) s_any[wv][j] = 1;
    }
    wave_lds_sync();
    for (int base = gw.r_begin; base < gw.r_end; base += 64) {
        const int r = base + lane;
        const bool in = r < gw.r_end;
        const int j = gw.group_of(in ? r : gw.r_begin);
        if (in) ray_active[r] = s_any[wv][j];
    }
}

// second stage of the attenuation: the rays of candidate readouts whose attenuation has not been computed yet (their own bound is
// below the candidate cut, but the channel traces of a candidate event sum all its rays)
__global__ void __launch_bounds__(256)
follower_flags_kernel(int n_ev, EventOut ev, const double* __restrict__ att, int n_fc, int n_rays, int* __restrict__ flag,
                      int* __restrict__ ray_active)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ev || !ev.candidate[e]) return;
    const int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
    for (int r = r0; r < r1; r++)
        if (!ray_active[r] && isnan(att[(long)r * n_fc])) {
            flag[r] = 1;
            ray_active[r] = 1;
        }
}

__global__ void scatter_active_kernel(int n_rays, const int* __restrict__ active, const int* __restrict__ offset,
                                      int* __restrict__ list)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    if (active[r]) list[offset[r]] = r;
}

// The attenuation quadrature of direct rays needs ~1 interval, of reflected rays a few, of refracted rays ~10
// (break point at the turning depth): list the active rays class by class so that the rays sharing a wave do similar
// work.  flags has 3 n_rays + 1 entries (class-major); its exclusive scan gives the list positions.
__device__ inline int work_class(int type) { return type == 1 ? 0 : (type == 3 ? 1 : 2); }

__global__ void active_class_flags_kernel(int n_rays, const int* __restrict__ active, const int* __restrict__ ray_slot2,
                                          const int* __restrict__ slot_type, int* __restrict__ flags)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int c = work_class(slot_type[ray_slot2[r]]);
    int a = active[r] ? 1 : 0;
    for (int k = 0; k < 3; k++) flags[(long)k * n_rays + r] = (k == c) ? a : 0;
}

__global__ void scatter_active_class_kernel(int n_rays, const int* __restrict__ flags, const int* __restrict__ offset,
                                            int* __restrict__ list)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= 3L * n_rays) return;
    if (flags[i]) list[offset[i]] = (int)(i % n_rays);
}

// ---- active rays listed by PREDICTED quadrature work (round 3) ---------------------------------------------------------------
// QUADPACK's number of bisections of a ray is set by how close the path comes to where ds/dz diverges (the turning point of the
// ray curve, wherever the path itself ends): direct and surface-reflected rays by h = z_turn(C0, unclamped) - (highest point of the
// path) -- h >= 30 m: one rule, h < 1 m: ~7 rounds --, refracted rays (break point AT the turning depth: 7 ... 11 rounds) by the
// depth span below it (the tolerance is relative to the whole integral).  Measured on 38 189 rays of the survey with the CPU
// restatement: the rounds a wave spends on its slowest ray exceed the mean of its rays by 7.1 % with the three classes by type,
// 2.1 % with these nineteen (0 % with a perfect sort).  The class only orders the list: results do not depend on it.
// A stable counting sort in two light kernels: per block of 256 rays the class counts (class-major, so that one exclusive scan
// over [class][block] gives every block its base per class), then ranks inside the block by ballots.
#define QC_NC 20
__device__ inline int quad_class(int type, double C0, double z1, double z2m, double zt, const IceConst& m)
{
    // (single precision: only the order of the list depends on it)
    const float zt_true = (float)m.z_0 * __logf(((float)m.n_ice - 1.f / (float)C0) / (float)m.delta_n);
    if (type == 2) {
        // rounds ~ 35 + 1.3 log2(part of the path above the turning depth's mirror, zt - upper end) - 3.2 log2(depth span below zt),
        // between 7 and 11 (fit to the survey's rays, residual 0.4 rounds); nine classes of half a round
        const float span = fmaxf((float)(zt - fmin(z1, z2m)), 1.f), up = fmaxf((float)(zt - fmax(z1, 2. * zt - z2m)), 1e-3f);
        const float rounds = 35.f + 1.3f * __log2f(up) - 3.2f * __log2f(span);
        const int k = (int)floorf((rounds - 6.75f) * 2.f);
        return 10 + (k < 0 ? 0 : (k > 8 ? 8 : k));
    }
    const float h = (type == 1) ? zt_true - (float)fmax(z1, z2m) : zt_true;
    return h >= 30.f ? 0 : h >= 22.f ? 1 : h >= 14.f ? 2 : h >= 10.f ? 3 : h >= 6.f ? 4 : h >= 4.f ? 5 : h >= 3.f ? 6 : h >= 2.f ? 7 :
           h >= 1.f ? 8 : 9;
}

__global__ void __launch_bounds__(256)
quad_class_count_kernel(int n_rays, const int* __restrict__ active, const int* __restrict__ ray_slot2, const int* __restrict__ slot_type,
                        const double* __restrict__ C0, const double* __restrict__ zint, IceConst m, signed char* __restrict__ cls,
                        int* __restrict__ counts, int getenv_ascending)
{
    __shared__ int hist[QC_NC];
    if (threadIdx.x < QC_NC) hist[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;
    if (r < n_rays && active[r]) {
        // heaviest class first: the waves stride through the list, and the few that get one pair more than the others take it from
        // the END of the list -- which should be the cheapest rays, not the refracted ones with eleven rounds
        c = QC_NC - 1 - quad_class(slot_type[ray_slot2[r]], C0[r], zint[3 * (long)r], zint[3 * (long)r + 1], zint[3 * (long)r + 2], m);
        if (getenv_ascending) c = QC_NC - 1 - c;
        atomicAdd(&hist[c], 1);
    }
    if (r < n_rays) cls[r] = (signed char)c;
    __syncthreads();
    if (threadIdx.x < QC_NC) counts[(long)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

__global__ void __launch_bounds__(256)
quad_class_scatter_kernel(int n_rays, const signed char* __restrict__ cls, const int* __restrict__ offset, int* __restrict__ list)
{
    __shared__ int wcount[4][QC_NC];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = (r < n_rays) ? (int)cls[r] : -1;
    int rank = 0;
    for (int k = 0; k < QC_NC; k++) {
        const unsigned long long mk = __ballot(c == k);
        if (c == k) rank = __popcll(mk & ((1ULL << lane) - 1ULL));
        if (lane == 0) wcount[wv][k] = __popcll(mk);
    }
    __syncthreads();
    if (c >= 0) {
        int before = 0;
        for (int q = 0; q < wv; q++) before += wcount[q][c];
        list[offset[(long)c * gridDim.x + blockIdx.x] + before + rank] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernel: max |E(t)| per ray (candidate cut, simulation.py:283-285).  One block per active ray.
// With attenuation known the sum-of-magnitudes bound is re-evaluated; only rays whose bound exceeds the cut pay for
// the time-domain transform (others report the negated bound).  Real reflection coefficients make both on-sky
// components proportional to one real pulse: a single transform serves both.
// LDS: N/2 complex + (N/2 + 1) doubles.
// ---------------------------------------------------------------------------------------------------------
// FP64 sums of efield_bound_kernel for one ray (lane = frequency bin modulo 64): sum_k amp_k att_k and sum_k (amp_k att_k)^2
__device__ __noinline__ double2 efield_bound_fp64_ray(int N, double fs, int n_fc, const unsigned char* __restrict__ seg,
                                                       const double* __restrict__ fpow, const double* __restrict__ lnf,
                                                       const AskaryanConst& ask, const double* at, const double* at_slope,
                                                       const double* s_xp, int lane)
{
    const int nh = N / 2, stride = nh + 1;
    const double df = 1.0 / (N * (1. / fs));
    const double x_first = s_xp[0], x_last = s_xp[n_fc - 1], dx_last = x_last - s_xp[n_fc - 2];
    const bool is2009 = (ask.model == 0);
    double part = 0., sq = 0.;
    for (int k = 1 + lane; k < nh; k += 64) {
        const double f = k * df;
        int lo = seg[k];
        double dx = f - s_xp[lo];
        if (f <= x_first) { lo = 0; dx = 0.; }
        if (f >= x_last) { lo = n_fc - 2; dx = dx_last; }
        double amp;
        if (is2009) {
            const double x = (ask.had ? fpow[k] : fpow[stride + k]) * ask.cL, y = fpow[2 * stride + k] * ask.cR;
            amp = ask.pref2 * f * bound_rcp((1 + x) * (1 + y));
        } else {
            amp = askaryan_amplitude(f, lnf[k], ask);
        }
        const double v = amp * (at_slope[lo] * dx + at[lo]);
        part += v;
        sq += v * v;
    }
    return make_double2(part, sq);
}

// kernel: with attenuation known, the sum-of-magnitudes bound on max |E(t)| and the L2 norm of the unit-polarisation
// pulse of every active ray (one wave per AB_RT rays, frequency-grid tables loaded once per AB_RT rays).  Rays whose bound
// stays below the cut report the negated bound; the others are flagged for the time-domain transform.
#ifndef NRHIP_EB_WAVES
#define NRHIP_EB_WAVES 3
#endif
__global__ void __launch_bounds__(256, NRHIP_EB_WAVES)
efield_bound_kernel(int n_active, const int* __restrict__ active_list, RayWork w, StationDev st, double min_efield,
                    int exact, double* __restrict__ max_efield, int* __restrict__ need_fft)
{
    __shared__ double at[4][AB_RT][NRHIP_MAX_NFC];
    __shared__ double at_slope[4][AB_RT][NRHIP_MAX_NFC];
    __shared__ double s_xp[NRHIP_MAX_NFC];
    for (int j = threadIdx.x; j < st.n_fc; j += blockDim.x) s_xp[j] = st.fcoarse[j];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nh = st.N / 2, stride = nh + 1;
    const double df = 1.0 / (st.N * (1. / st.fs));
    const int per_pass = gridDim.x * 4 * AB_RT;
    const int n_iter = (n_active + per_pass - 1) / per_pass;
    for (int it = 0; it < n_iter; it++) {
        const int ib = ((it * gridDim.x + blockIdx.x) * 4 + wv) * AB_RT;
        int rr[AB_RT];
        for (int i = 0; i < AB_RT; i++) rr[i] = active_list[min(ib + i, n_active - 1)];
        for (int i = 0; i < AB_RT; i++)
            if (lane < st.n_fc) at[wv][i][lane] = w.att[(long)rr[i] * st.n_fc + lane];
        __syncthreads();
        for (int i = 0; i < AB_RT; i++)
            if (lane < st.n_fc - 1)
                at_slope[wv][i][lane] = (at[wv][i][lane + 1] - at[wv][i][lane]) / (s_xp[lane + 1] - s_xp[lane]);
        __syncthreads();
        if (ib < n_active) {
            double cL[AB_RT], cR[AB_RT], pf[AB_RT];
            int had[AB_RT];
            int model_or = 0;   // (no short circuit: the four records are requested together)
            for (int i = 0; i < AB_RT; i++) {
                const AskaryanConst& ai = w.ask[rr[i]];
                model_or |= ai.model;
                cL[i] = ai.cL; cR[i] = ai.cR; pf[i] = ai.pref2; had[i] = ai.had;
            }
            const bool all2009 = model_or == 0;
            double part[AB_RT], sq[AB_RT];
            for (int i = 0; i < AB_RT; i++) part[i] = sq[i] = 0.;
            const double x_first = s_xp[0], x_last = s_xp[st.n_fc - 1], dx_last = x_last - s_xp[st.n_fc - 2];
            bool f32 = all2009;  // as in amp_bound_kernel: FP32 sums, inflated
            for (int i = 0; i < AB_RT; i++)
                f32 = f32 && pf[i] > 1e-18 && pf[i] < 1e18 && cL[i] > 1e-15 && cL[i] < 1e15 && cR[i] > 1e-15 && cR[i] < 1e15;
            if (f32) {
                float cLf[AB_RT], cRf[AB_RT], pff[AB_RT], p32[AB_RT], q32[AB_RT];
                for (int i = 0; i < AB_RT; i++) { cLf[i] = (float)cL[i]; cRf[i] = (float)cR[i]; pff[i] = (float)pf[i]; p32[i] = q32[i] = 0.f; }
                const float xf_first = (float)x_first, xf_last = (float)x_last, dxf_last = (float)dx_last, dff = (float)df;
                // as in amp_bound_kernel: the tables of the next bin are requested before the current one is evaluated
                int kq = 1 + lane;
                int n_lo = (kq < nh) ? st.seg[kq] : 0;
                float n_ph = (kq < nh) ? st.fpow_f[kq] : 0.f, n_pe = (kq < nh) ? st.fpow_f[stride + kq] : 0.f,
                      n_pr = (kq < nh) ? st.fpow_f[2 * stride + kq] : 0.f;
                for (int k = 1 + lane; k < nh; k += 64) {
                    const float f = k * dff;
                    int lo = n_lo;
                    const float ph = n_ph, pe = n_pe, pr = n_pr;
                    kq = k + 64;
                    if (kq < nh) {
                        n_lo = st.seg[kq];
                        n_ph = st.fpow_f[kq];
                        n_pe = st.fpow_f[stride + kq];
                        n_pr = st.fpow_f[2 * stride + kq];
                    }
                    float dx = f - (float)s_xp[lo];
                    if (f <= xf_first) { lo = 0; dx = 0.f; }
                    if (f >= xf_last) { lo = st.n_fc - 2; dx = dxf_last; }
#pragma unroll
                    for (int i = 0; i < AB_RT; i++) {
                        const float x = (had[i] ? ph : pe) * cLf[i], y = pr * cRf[i];
                        const float amp = pff[i] * f * __builtin_amdgcn_rcpf((1.f + x) * (1.f + y));
                        const float v = amp * fmaxf((float)at_slope[wv][i][lo] * dx + (float)at[wv][i][lo], 0.f);
                        p32[i] += v;
                        q32[i] += v * v;
                    }
                }
                for (int i = 0; i < AB_RT; i++) {
                    part[i] = (double)p32[i] * BOUND_F32_SLACK + 1e-30;  // + what FP32 may have flushed to zero
                    sq[i] = (double)q32[i] * (BOUND_F32_SLACK * BOUND_F32_SLACK) + 1e-60;
                }
            } else {
                // rare (another emission model, or parameters outside the FP32 range): FP64, ray by ray, out of line -- inlined
                // four times this loop's registers pushed the common path above into scratch (1.3 GB of spill writes per launch)
                for (int i = 0; i < AB_RT; i++) {
                    const double2 r = efield_bound_fp64_ray(st.N, st.fs, st.n_fc, st.seg, st.fpow, st.lnf, w.ask[rr[i]], at[wv][i], at_slope[wv][i], s_xp, lane);
                    part[i] = r.x;
                    sq[i] = r.y;
                }
            }
#if NRHIP_WAVE_DPP
            const double pa = wave_fold4(part[0], part[1], part[2], part[3]), qa = wave_fold4(sq[0], sq[1], sq[2], sq[3]);
            const double pts[AB_RT] = {wave_row_value<0>(pa), wave_row_value<1>(pa), wave_row_value<2>(pa), wave_row_value<3>(pa)};
            const double s2s[AB_RT] = {wave_row_value<0>(qa), wave_row_value<1>(qa), wave_row_value<2>(qa), wave_row_value<3>(qa)};
#endif
            for (int i = 0; i < AB_RT; i++) {
#if NRHIP_WAVE_DPP
                const double pt = pts[i], s2 = s2s[i];
#else
                double pt = part[i], s2 = sq[i];
                for (int off = 32; off > 0; off >>= 1) {
                    pt += __shfl_xor(pt, off);
                    s2 += __shfl_xor(s2, off);
                }
#endif
                if (lane == 0 && ib + i < n_active) {
                    const int r = rr[i];
                    // Parseval: sum_t s(t)^2 = (fs^2 / 2) (1 / N) 2 sum_k |G_k|^2 with |G_k| = sqrt(2) amp_k
                    w.e_norm[r] = sqrt((st.fs * st.fs / st.N) * 2. * s2) * BOUND_RCP_SLACK;
                    double cmax = fmax(fabs(w.pol_theta[r]) * cabs2(w.r_theta[r]), fabs(w.pol_phi[r]) * cabs2(w.r_phi[r]));
                    double bnd = efield_bound(pt * BOUND_RCP_SLACK, st.N, st.fs, cmax);
                    bool need = exact || (bnd * (1 + 1e-6) > min_efield);
                    max_efield[r] = -bnd;
                    need_fft[r] = need ? 1 : 0;  // per ray (zero-initialised for inactive rays)
                }
            }
        }
        __syncthreads();
    }
}

// events with at least one undecided ray (the unit of work of efield_max_kernel)
__global__ void __launch_bounds__(256)
event_need_kernel(int n_events, int n_ch, const int* __restrict__ slot_offset, const int* __restrict__ need_ray,
                  int* __restrict__ ev_need, const double* __restrict__ max_efield, double min_efield, int exact)
{
    __shared__ int s_any[4][64], s_dec[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + wv) * 64;
    if (g0 > n_events) return;
    if (g0 == n_events) {   // the entry behind the last group (the scan's total)
        if (lane == 0) ev_need[n_events] = 0;
        return;
    }
    const GroupWave gw(slot_offset, g0, n_events);
    s_any[wv][lane] = 0;
    s_dec[wv][lane] = 0;
    wave_lds_sync();
    for (int base = gw.r_begin; base < gw.r_end; base += 64) {   // (all lanes take part in the shuffles)
        const int r = base + lane;
        const bool in = r < gw.r_end;
        const int j = gw.group_of(in ? r : gw.r_begin);
        if (in) {
            if (need_ray[r]) s_any[wv][j] = 1;
            if (max_efield[r] > min_efield) s_dec[wv][j] = 1;   // a ray whose sampled field already exceeds the cut makes the event a candidate
        }
    }
    wave_lds_sync();
    const int g = g0 + lane;
    if (g < n_events) ev_need[g] = (s_dec[wv][lane] && !exact) ? 0 : s_any[wv][lane];
    if (g == n_events) ev_need[g] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// kernel: the rays the sum-of-magnitudes bound leaves open, decided WITHOUT a transform where possible.  With real reflection
// coefficients the (Alvarez2009) pulse is, up to the factor c = max |pol r|,
//     e[N/2 + j] = -fs (2 / N) sum_k v_k sin(2 pi k j / N),   v_k = amplitude x attenuation of bin k   (i (-1)^k spectrum, N/2 roll)
// -- antisymmetric about the centre sample.  The samples j = 1 .. ES_NJ are summed directly (exact values of the trace: a lower
// bound on max |e|, and an upper bound on those samples), every later sample is bounded by summation by parts,
//     |sum_k v_k e^{i k theta}| <= (v_last + sum_k |v_{k+1} - v_k|) / |sin(theta / 2)|,   theta = 2 pi j / N,
// which falls like 1 / j while the pulse lives within a few samples of the centre.  So
//     low = max_j |S_j| - err   <=   max |e| / scale   <=   max(max_j |S_j| + err, TV / sin(pi (ES_NJ + 1) / N))  = up,
// and only rays with low <= cut <= up still need the N-point transform.  FP32 sums (terms within 1e-6, the sine recurrence
// within ES_NJ^2 ulp): err = 2e-3 of sum_k v_k covers them.  One wave per ray; lanes over the bins.
// ---------------------------------------------------------------------------------------------------------
#ifndef ES_NJ
#define ES_NJ 24
#endif
__global__ void __launch_bounds__(256)
efield_sample_kernel(int n_active, const int* __restrict__ active_list, RayWork w, StationDev st, double min_efield,
                     double* __restrict__ max_efield, int* __restrict__ need_fft, unsigned long long* __restrict__ sampled_count)
{
    unsigned n_sampled = 0;   // rays this wave summed (bench.py prices the stage by them)
    __shared__ double s_xp[NRHIP_MAX_NFC];
    __shared__ float at[4][NRHIP_MAX_NFC], at_slope[4][NRHIP_MAX_NFC];
    // the per-bin tables of the station in LDS (N <= 4096: 3 x 2049 floats + 2049 bytes): the inner loop then never waits for HBM / L2
    extern __shared__ __align__(16) unsigned char es_smem[];
    const int nh = st.N / 2, stride = nh + 1;
    float* s_fpow = (float*)es_smem;                       // [3][stride]
    unsigned char* s_seg = (unsigned char*)(s_fpow + 3 * stride);
    for (int j = threadIdx.x; j < 3 * stride; j += blockDim.x) s_fpow[j] = st.fpow_f[j];
    for (int j = threadIdx.x; j < stride; j += blockDim.x) s_seg[j] = st.seg[j];
    for (int j = threadIdx.x; j < st.n_fc; j += blockDim.x) s_xp[j] = st.fcoarse[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float dff = (float)(1.0 / (st.N * (1. / st.fs)));
    const float xf_first = (float)s_xp[0], xf_last = (float)s_xp[st.n_fc - 1], dxf_last = (float)(s_xp[st.n_fc - 1] - s_xp[st.n_fc - 2]);
    const float invN = 1.f / (float)st.N;
    const int n_waves = gridDim.x * 4;
    for (int ia = blockIdx.x * 4 + wv; ia < n_active; ia += n_waves) {
        const int r = active_list[ia];
        if (!need_fft[r]) continue;   // wave-uniform
        const AskaryanConst& ai = w.ask[r];
        const double2 rt = w.r_theta[r], rp = w.r_phi[r];
        const double cL = ai.cL, cR = ai.cR, pf = ai.pref2;
        if (ai.model != 0 || rt.y != 0. || rp.y != 0. || !(pf > 1e-18 && pf < 1e18 && cL > 1e-15 && cL < 1e15 && cR > 1e-15 && cR < 1e15))
            continue;
        if (lane < st.n_fc) at[wv][lane] = (float)w.att[(long)r * st.n_fc + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < st.n_fc - 1) at_slope[wv][lane] = (float)(((double)at[wv][lane + 1] - (double)at[wv][lane]) / (s_xp[lane + 1] - s_xp[lane]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float cLf = (float)cL, cRf = (float)cR, pff = (float)pf;
        const int had = ai.had;
        n_sampled++;
        auto value = [&](int k) -> float {   // v_k, 0 outside 1 .. nh - 1
            if (k < 1 || k >= nh) return 0.f;
            const float f = k * dff;
            int lo = s_seg[k];
            float dx = f - (float)s_xp[lo];
            if (f <= xf_first) { lo = 0; dx = 0.f; }
            if (f >= xf_last) { lo = st.n_fc - 2; dx = dxf_last; }
            const float x = (had ? s_fpow[k] : s_fpow[stride + k]) * cLf, y = s_fpow[2 * stride + k] * cRf;
            const float amp = pff * f * __builtin_amdgcn_rcpf((1.f + x) * (1.f + y));
            return amp * fmaxf(at_slope[wv][lo] * dx + at[wv][lo], 0.f);
        };
        float acc[ES_NJ];
#pragma unroll
        for (int j = 0; j < ES_NJ; j++) acc[j] = 0.f;
        float sum_v = 0.f, tv = 0.f;
        float v_prev_wave = 0.f;   // v of the bin before this pass's first one (v_0 = 0)
        for (int k = 1 + lane; k - lane <= nh; k += 64) {   // (wave-uniform trip count: the shuffles below need every lane; k = nh is visited)
            const float v = value(k);   // 0 beyond nh - 1
            // total variation sum_{k >= 1} |v_k - v_{k-1}| over k = 1 .. nh (v_0 = v_nh = 0): the neighbour's value comes from the
            // lane below (the previous pass's last lane for lane 0) instead of a second evaluation of the amplitude formula;
            // |v_1 - v_0| is not part of the bound's sum: lane 0 of the first pass skips it
#if NRHIP_WAVE_DPP
            float vm = wave_from_lane_below(v);
            if (lane == 0) vm = v_prev_wave;
            v_prev_wave = wave_lane_value(v, 63);
#else
            float vm = __shfl_up(v, 1);
            if (lane == 0) vm = v_prev_wave;
            v_prev_wave = __shfl(v, 63);
#endif
            sum_v += v;
            if (k >= 2 && k <= nh) tv += fabsf(v - vm);   // k = nh: |0 - v_{nh-1}| = v_last
            if (k >= nh) continue;
            const float ph = (float)k * invN;   // k / N < 1/2: exact enough for the hardware sine / cosine (argument in turns)
            const float s1 = __builtin_amdgcn_sinf(ph), c2 = 2.f * __builtin_amdgcn_cosf(ph);
            float sa = 0.f, sb = s1;            // sin(0), sin(theta); sin((j + 1) theta) = 2 cos(theta) sin(j theta) - sin((j - 1) theta)
#pragma unroll
            for (int j = 0; j < ES_NJ; j++) {
                acc[j] += v * sb;
                const float sc = c2 * sb - sa;
                sa = sb;
                sb = sc;
            }
        }
        float best = 0.f;
#if NRHIP_WAVE_DPP
        {   // ES_NJ + 2 wave sums, four to a register (wave_reduce.h); each row of `m` then holds the largest |S_j| of its slot
            static_assert(ES_NJ % 4 == 0, "wave_fold4");
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < ES_NJ; j += 4) m = fmaxf(m, fabsf(wave_fold4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3])));
            best = fmaxf(fmaxf(wave_lane_value(m, 0), wave_lane_value(m, 16)), fmaxf(wave_lane_value(m, 32), wave_lane_value(m, 48)));
            const float sv = wave_fold4(sum_v, tv, sum_v, tv);
            sum_v = wave_row_value<0>(sv);
            tv = wave_row_value<1>(sv);
        }
#else
#pragma unroll
        for (int j = 0; j < ES_NJ; j++) {
            float t = acc[j];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            best = fmaxf(best, fabsf(t));
        }
        for (int off = 32; off > 0; off >>= 1) {
            sum_v += __shfl_xor(sum_v, off);
            tv += __shfl_xor(tv, off);
        }
#endif
        if (lane == 0) {
            const double cmax = fmax(fabs(w.pol_theta[r]) * fabs(rt.x), fabs(w.pol_phi[r]) * fabs(rp.x));
            const double err = 2e-3 * (double)sum_v + 1e-30;
            const double tail = (double)tv * (1. + 1e-3) / sin(M_PI * (ES_NJ + 1) / st.N);
            const double low = efield_bound((double)best - err, st.N, st.fs, cmax);
            const double up = efield_bound(fmax((double)best + err, tail), st.N, st.fs, cmax);
            if (low > min_efield * (1 + 1e-9)) {
                max_efield[r] = low;    // "at least": the event is a candidate
                need_fft[r] = 0;
            } else if (up * (1 + 1e-9) < min_efield) {
                max_efield[r] = -up;    // "at most": this ray cannot make the event a candidate
                need_fft[r] = 0;
            }
        }
    }
    if (sampled_count && lane == 0 && n_sampled) atomicAdd(sampled_count, (unsigned long long)n_sampled);
}

// ---------------------------------------------------------------------------------------------------------
// kernel (round 6): efield_bound_kernel and efield_sample_kernel in one pass over the bins, the samples on the matrix cores.
// Both kernels above evaluate v_k = amplitude x attenuation of every bin of a ray; the first sums v_k and v_k^2, the second
// re-evaluates them for the three rays in four the first leaves open and pays 2 FP32 instructions per (bin, sample) for
//     S_j = sum_k v_k sin(2 pi j k / N).
// That sum is a matrix product: [32 rays x K bins] x [K bins x 32 columns] with the column table sin(2 pi j k / N) fixed per station,
// and column 0 = 1 hands over sum_k v_k as well.  One wave takes 32 rays: lane l evaluates v of ray l & 31 for one bin per step
// (lanes 0..31 the lower half of the bins, lanes 32..63 the upper half -- each lane walks consecutive bins, so the total
// variation needs no neighbour lane) and v_mfma_f32_32x32x2_f32 accumulates D[ray][j] += v[ray][k] B[k][j], an exact FP32 fma
// chain (64 cycles per bin and wave: 64 FLOP / clk / SIMD, twice what unpacked vector fmas reach; measured, it does NOT overlap the
// vector instructions of the SIMD's other waves -- the FP32-input MFMA runs at the vector rate -- so the kernel's time is the sum
// of the two: SQ counters in profiles/).  Every lane takes TWO consecutive bins per step (8-byte table
// reads; the packed arithmetic costs the same issue time as unpacked -- v_pk_*_f32 issue at half rate -- but fewer LDS
// instructions).  31 samples instead of 24 (a lower tail bound), and the epilogue of 32 rays runs in 32 lanes instead of one.
// Tables per block (LDS), bin k at index k - 1 (pairs start at even indices: 8-byte reads): t = position of f_k inside its
// coarse-grid segment (the interpolated attenuation is a0 + t (a1 - a0)), f^alpha, f^beta_had, f^beta_em, the segment index,
// sin(2 pi m / N) for m < N; per wave the attenuation values of its 32 rays [segment][ray] (stride 33).  Bins k >= N / 2 carry
// f^beta = inf: amplitude 0.
// Rays outside the FP32 range of the fast path: FP64 sums, ray by ray (efield_bound_fp64_ray), no sampling.
// ---------------------------------------------------------------------------------------------------------
#define ED_NJ 31
#define ED_ROW 33          // floats per row of the per-wave tables
#ifndef ED_MAX_WAVES
#define ED_MAX_WAVES 16
#endif
typedef float ed_f32x16 __attribute__((ext_vector_type(16)));
typedef float ed_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 ed_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ed_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ed_u4 __attribute__((ext_vector_type(4)));
#ifndef ED_BF16
#define ED_BF16 1   // 0: the product as v_mfma_f32_32x32x2_f32 (FP32 operands: runs at the vector rate and does not overlap vector work)
#endif
// two floats rounded to bfloat16 (v_cvt_pk_bf16_f32, round to nearest even): a in the low half
__device__ __forceinline__ unsigned ed_pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(ed_f2{a, b}, ed_bf16x2));
}
__host__ __device__ inline int ed_wave_floats(int n_fc) { return (n_fc > 32 ? n_fc : 32) * ED_ROW; }
// bins per half of the wave: the lower half walks k = 1 .. KH, the upper KH + 1 .. 2 KH (>= N / 2)
__host__ __device__ inline int ed_bins_per_half(int N) { return 8 * ((N / 2 + 15) / 16); }   // (a multiple of the 8 bins of a matrix operand)
__host__ __device__ inline size_t ed_table_bytes(int N)
{
    const size_t n_tab = 2 * (size_t)ed_bins_per_half(N);
    return n_tab * 16 + (size_t)(N + 4) * 4 + ((n_tab + 15) & ~(size_t)15);
}

__global__ void __launch_bounds__(64 * ED_MAX_WAVES)
efield_decide_kernel(int n_active, const int* __restrict__ active_list, RayWork w, StationDev st, double min_efield, int exact,
                     double* __restrict__ max_efield, int* __restrict__ need_fft, unsigned long long* __restrict__ sampled_count)
{
    extern __shared__ __align__(16) unsigned char ed_smem[];
    __shared__ double s_xp[NRHIP_MAX_NFC];
    const int N = st.N, nh = N / 2, stride = nh + 1, n_fc = st.n_fc;
    const int KH = ed_bins_per_half(N), n_tab = 2 * KH;
    float* s_t = (float*)ed_smem;                         // [n_tab] each, entry q = bin q + 1
    float* s_pr = s_t + n_tab;                            // f^alpha
    float* s_ph = s_pr + n_tab;                           // f^beta, hadronic
    float* s_pe = s_ph + n_tab;                           // f^beta, electromagnetic
    float* s_sin = s_pe + n_tab;                          // [N] sin(2 pi m / N), then 1.0 (column 0)
    unsigned char* s_lo = (unsigned char*)(s_sin + N + 4);   // [n_tab]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
    float* s_wave = (float*)(s_lo + ((n_tab + 15) & ~15)) + (size_t)wv * ed_wave_floats(n_fc);
    const double df = 1.0 / (N * (1. / st.fs));
    for (int j = threadIdx.x; j < n_fc; j += blockDim.x) s_xp[j] = st.fcoarse[j];
    __syncthreads();
    for (int q = threadIdx.x; q < n_tab; q += blockDim.x) {
        const int k = q + 1;
        float t = 0.f, pr = 0.f, ph = __builtin_inff(), pe = __builtin_inff();
        int lo = 0;
        if (k < nh) {
            const double f = k * df;
            lo = st.seg[k];
            double td = (f - s_xp[lo]) / (s_xp[lo + 1] - s_xp[lo]);
            if (f <= s_xp[0]) { lo = 0; td = 0.; }
            if (f >= s_xp[n_fc - 1]) { lo = n_fc - 2; td = 1.; }
            t = (float)td; pr = st.fpow_f[2 * stride + k]; ph = st.fpow_f[k]; pe = st.fpow_f[stride + k];
        }
        s_t[q] = t; s_pr[q] = pr; s_ph[q] = ph; s_pe[q] = pe;
        s_lo[q] = (unsigned char)lo;
    }
#if ED_BF16
    // the sines as two bfloat16 each (value = hi + lo to 2^-18), hi in the low half of the word
    for (int m = threadIdx.x; m < N + 4; m += blockDim.x) {
        const float sv = m < N ? (float)sinpi(2. * m / N) : 1.f;
        const unsigned hi = ed_pk_bf16(sv, 0.f) & 0xffffu;
        const unsigned lo = ed_pk_bf16(sv - __uint_as_float(hi << 16), 0.f) & 0xffffu;
        ((unsigned*)s_sin)[m] = hi | (lo << 16);
    }
#else
    for (int m = threadIdx.x; m < N; m += blockDim.x) s_sin[m] = (float)sinpi(2. * m / N);
    if (threadIdx.x < 4) s_sin[N + threadIdx.x] = 1.f;
#endif
    __syncthreads();

    const int i = lane & 31, h = lane >> 5;
    const int q0 = h * KH;                       // table index of the lane's first bin (bin q0 + 1)
    const float dff = (float)df;
    // column j = i of the table: sin(2 pi j k / N) = s_sin[(j k) mod N]; the lane's even and odd bins advance by 2 j each
    // (j < 32 <= N / 2); column 0 reads the constant 1 behind the table and stays there
    const unsigned sin_step = 8u * (unsigned)i, sin_wrap = (i == 0) ? 0u : 4u * (unsigned)N;
    const unsigned sin_a0 = (i == 0) ? 4u * (unsigned)N : 4u * (unsigned)(((long)i * (q0 + 1)) % N);
    const unsigned sin_b0 = (i == 0) ? 4u * (unsigned)N : 4u * (unsigned)(((long)i * (q0 + 2)) % N);
    const int n_tiles = (n_active + 31) / 32;
    unsigned n_sampled = 0;
    for (int tile = blockIdx.x * n_wv + wv; tile < n_tiles; tile += gridDim.x * n_wv) {
        const int ia = tile * 32 + i;
        const bool valid = ia < n_active;
        const int r = active_list[min(ia, n_active - 1)];
        const AskaryanConst& ai = w.ask[r];
        const double cL = ai.cL, cR = ai.cR, pf = ai.pref2;
        const bool fast = ai.model == 0 && pf > 1e-18 && pf < 1e18 && cL > 1e-15 && cL < 1e15 && cR > 1e-15 && cR < 1e15;
        const float* s_psel = ai.had != 0 ? s_ph : s_pe;
        wave_lds_sync();   // (the previous tile's readers of s_wave are done)
        for (int q = 0; q < 32; q += 2) {   // two rays per pass, lanes over the coarse frequencies
            const int rq = __shfl(r, q + h);
            for (int c = i; c < n_fc; c += 32) s_wave[c * ED_ROW + q + h] = (float)w.att[(long)rq * n_fc + c];
        }
        wave_lds_sync();
        const float cLf = fast ? (float)cL : 1.f, cRf = fast ? (float)cR : 1.f, pfd = fast ? (float)pf * dff : 0.f;
        ed_f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.f;
        ed_f2 sq2 = ed_f2{0.f, 0.f}, kf2 = ed_f2{(float)(q0 + 1), (float)(q0 + 2)};   // (bin numbers as floats: exact)
        float tv = 0.f, v_prev = 0.f, v_first = 0.f;
        unsigned sm_a = sin_a0, sm_b = sin_b0;
        const float* s_at_lane = s_wave + i;
#if ED_BF16
        ed_u4 a_hi, a_lo, b_hi, b_lo;   // 8 bins of this lane: v and the sines as bfloat16 pairs (hi, lo)
#endif
        auto step = [&](int q, bool first, int slot) {   // bins q + 1, q + 2
            const ed_f2 t = *(const ed_f2*)(s_t + q), pr = *(const ed_f2*)(s_pr + q), ps = *(const ed_f2*)(s_psel + q);
            const unsigned lo2 = *(const unsigned short*)(s_lo + q);
            const int ra = (lo2 & 0xffu) * ED_ROW, rb = (lo2 >> 8) * ED_ROW;
            const ed_f2 a0 = ed_f2{s_at_lane[ra], s_at_lane[rb]}, a1 = ed_f2{s_at_lane[ra + ED_ROW], s_at_lane[rb + ED_ROW]};
            const ed_f2 att = ed_f2{fmaf(t.x, a1.x - a0.x, a0.x), fmaf(t.y, a1.y - a0.y, a0.y)};   // (per bin: the pairs come in (a0, a1) order)
            const ed_f2 x = ps * cLf, yp = pr * cRf + 1.f;
            const ed_f2 den = x * yp + yp;
            const ed_f2 amp = (kf2 * pfd) * ed_f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            kf2 += 2.f;
            const ed_f2 v = amp * ed_f2{fmaxf(att.x, 0.f), fmaxf(att.y, 0.f)};
#if ED_BF16
            {   // v = hi + lo (two roundings to 8 bits: 2^-18 of v left over); the sines come split from the table
                const unsigned vh = ed_pk_bf16(v.x, v.y);
                a_hi[slot] = vh;
                a_lo[slot] = ed_pk_bf16(v.x - __uint_as_float(vh << 16), v.y - __uint_as_float(vh & 0xffff0000u));
                const unsigned da = *(const unsigned*)((const unsigned char*)s_sin + sm_a), db = *(const unsigned*)((const unsigned char*)s_sin + sm_b);
                b_hi[slot] = __builtin_amdgcn_perm(db, da, 0x05040100u);   // (hi_a, hi_b)
                b_lo[slot] = __builtin_amdgcn_perm(db, da, 0x07060302u);   // (lo_a, lo_b)
            }
#else
            const float bj_a = *(const float*)((const unsigned char*)s_sin + sm_a), bj_b = *(const float*)((const unsigned char*)s_sin + sm_b);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, bj_a, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, bj_b, acc, 0, 0, 0);
#endif
            sq2 += v * v;
            // total variation of the lane's run of bins (the step from the run's predecessor: after the loop)
            if (first) v_first = v.x;
            else tv += fabsf(v.x - v_prev);
            tv += fabsf(v.y - v.x);
            v_prev = v.y;
            const unsigned sa = sm_a + sin_step, sb = sm_b + sin_step;
            sm_a = min(sa, sa - sin_wrap);   // (unsigned: s - wrap is huge while s < wrap)
            sm_b = min(sb, sb - sin_wrap);
        };
        // 8 bins per lane, then the product of the group: v s = hi_v hi_s + lo_v hi_s + hi_v lo_s (+ 2^-18 v s), accumulated in FP32
        auto group = [&](int q, bool first) {
            step(q, first, 0);
            step(q + 2, false, 1);
            step(q + 4, false, 2);
            step(q + 6, false, 3);
#if ED_BF16
            const ed_bf16x8 ah = __builtin_bit_cast(ed_bf16x8, a_hi), al = __builtin_bit_cast(ed_bf16x8, a_lo);
            const ed_bf16x8 bh = __builtin_bit_cast(ed_bf16x8, b_hi), bl = __builtin_bit_cast(ed_bf16x8, b_lo);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
#endif
        };
        group(q0, true);
        for (int s = 8; s < KH; s += 8) group(q0 + s, false);
        const float sq = sq2.x + sq2.y;
        // the step at the head of each half's run: its first bin against the bin before it -- v_0 = 0 for the lower half, the lower
        // half's last bin for the upper one: tv = sum_{k = 1}^{N/2} |v_k - v_{k-1}| of the zero-padded sequence
        tv += fabsf(v_first - wave_from_lower_half(v_prev));   // (0 in the lower half)
        tv = wave_fold32(tv, tv);
        const float sq_ray = wave_fold32(sq, sq);
        // D[ray][j]: register q of lane (h, j) holds ray (q & 3) + 8 (q >> 2) + 4 h
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 16; q++) s_wave[((q & 3) + 8 * (q >> 2) + 4 * h) * ED_ROW + i] = acc[q];
        wave_lds_sync();
        float best = 0.f;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const float t = s_wave[i * ED_ROW + 16 * h + c];
            if (c > 0 || h) best = fmaxf(best, fabsf(t));
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(best), __float_as_int(best), false, false);
            best = fmaxf(__int_as_float(sw[0]), __int_as_float(sw[1]));
        }
        const float sum_v = s_wave[i * ED_ROW];
        bool sampled = false;
        if (valid && fast && h == 0) {
            const double2 rt = w.r_theta[r], rp = w.r_phi[r];
            const double pth = fabs(w.pol_theta[r]), pph = fabs(w.pol_phi[r]);
            const double pt = (double)sum_v * BOUND_F32_SLACK + 1e-30;   // + what FP32 may have flushed to zero
            const double s2 = (double)sq_ray * (BOUND_F32_SLACK * BOUND_F32_SLACK) + 1e-60;
            // Parseval: sum_t s(t)^2 = (fs^2 / 2) (1 / N) 2 sum_k |G_k|^2 with |G_k| = sqrt(2) amp_k
            w.e_norm[r] = sqrt((st.fs * st.fs / N) * 2. * s2) * BOUND_RCP_SLACK;
            const double cmax = fmax(pth * cabs2(rt), pph * cabs2(rp));
            const double bnd = efield_bound(pt * BOUND_RCP_SLACK, N, st.fs, cmax);
            bool need = exact || (bnd * (1 + 1e-6) > min_efield);
            double out = -bnd;
            if (need && !exact && rt.y == 0. && rp.y == 0.) {
                // the samples decide what they can: low <= max |e| / scale <= up (comment of efield_sample_kernel).  The products of
                // the split operands leave out 2^-18 of every term, the FP32 accumulation is within N / 2 ulp of the exact sum, v_k
                // itself within 2e-6: err = 1e-3 of sum_k v_k covers them several times
                sampled = true;
                // Beyond the samples, summation by parts: with z = exp(2 pi i j / N) and v_0 = v_{N/2} = 0,
                // (1 - z) sum_k v_k z^k = sum_k (v_k - v_{k-1}) z^k, so |S_j| <= tv / (2 sin(pi j / N)), largest at j = ED_NJ + 1 (half
                // of efield_sample_kernel's bound, which drops |v_1 - v_0| and the factor 2 of |1 - z|).  v_k as evaluated here is
                // within 2e-6 sum_k v_k (summed over k) of the exact amplitudes: + 4e-6 sum_v; FP32 differences and sums: 1e-3.
                // (The second-order bound, tv2 / (4 sin^2), was built and measured: it decides no further ray of config 2.)
                const double err = 1e-3 * (double)sum_v + 1e-30;
                const double tail = ((double)tv + 4e-6 * (double)sum_v) * (1. + 1e-3) / (2. * sin(M_PI * (ED_NJ + 1) / N));
                const double low = efield_bound((double)best - err, N, st.fs, cmax);
                const double up = efield_bound(fmax((double)best + err, tail), N, st.fs, cmax);
                if (low > min_efield * (1 + 1e-9)) {
                    out = low;      // "at least": the event is a candidate
                    need = false;
                } else if (up * (1 + 1e-9) < min_efield) {
                    out = -up;      // "at most": this ray cannot make the event a candidate
                    need = false;
                }
            }
            max_efield[r] = out;
            need_fft[r] = need ? 1 : 0;
        }
        n_sampled += __popcll(__ballot(sampled));
        // rays outside the fast path's range: the FP64 sums of efield_bound_kernel, one ray at a time over the wave
        unsigned long long slow = __ballot(valid && !fast && h == 0);
        while (slow) {
            const int q = __ffsll((long long)slow) - 1;
            slow &= slow - 1;
            const int rq = __shfl(r, q);
            double* at_d = (double*)s_wave;
            double* slope_d = at_d + n_fc;
            wave_lds_sync();
            if (lane < n_fc) at_d[lane] = w.att[(long)rq * n_fc + lane];
            wave_lds_sync();
            if (lane < n_fc - 1) slope_d[lane] = (at_d[lane + 1] - at_d[lane]) / (s_xp[lane + 1] - s_xp[lane]);
            wave_lds_sync();
            const double2 ps = efield_bound_fp64_ray(N, st.fs, n_fc, st.seg, st.fpow, st.lnf, w.ask[rq], at_d, slope_d, s_xp, lane);
            const double pt = wave_sum(ps.x), s2 = wave_sum(ps.y);
            if (lane == 0) {
                w.e_norm[rq] = sqrt((st.fs * st.fs / N) * 2. * s2) * BOUND_RCP_SLACK;
                const double cmax = fmax(fabs(w.pol_theta[rq]) * cabs2(w.r_theta[rq]), fabs(w.pol_phi[rq]) * cabs2(w.r_phi[rq]));
                const double bnd = efield_bound(pt * BOUND_RCP_SLACK, N, st.fs, cmax);
                max_efield[rq] = -bnd;
                need_fft[rq] = (exact || (bnd * (1 + 1e-6) > min_efield)) ? 1 : 0;
            }
        }
    }
    if (sampled_count && lane == 0 && n_sampled) atomicAdd(sampled_count, (unsigned long long)n_sampled);
}

__global__ void scatter_flagged_kernel(int n, const int* __restrict__ flag, const int* __restrict__ offset,
                                       int* __restrict__ list)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) list[offset[i]] = i;
}

// ---------------------------------------------------------------------------------------------------------
// kernel: max |E(t)| of the rays the bound could not decide (candidate cut, simulation.py:283-285): one block per listed
// ray, N-point field in the time domain.  Real reflection coefficients make both on-sky components proportional to one
// real pulse: a single transform serves both.  LDS: N/2 complex + (N/2 + 1) doubles.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
efield_max_kernel(const int* __restrict__ n_list, const int* __restrict__ ev_list, const int* __restrict__ need_ray,
                  const int* __restrict__ slot_offset, RayWork w, EventIn evin, StationDev st, int ask_model,
                  const double2* __restrict__ tw, int log2nh, double min_efield, int exact, double* __restrict__ max_efield,
                  unsigned long long* __restrict__ xform_count, double* __restrict__ amp_scratch)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = st.N, nh = N / 2;
    double2* x = (double2*)smem;
    // (amplitude table in a row of HBM scratch when the Bluestein transform takes all of the LDS: ray_amp_in_hbm)
    double* amp = amp_scratch ? amp_scratch + (long)blockIdx.x * (nh + 1) : (double*)(x + nplan_points(st.np));
    __shared__ RayShared rs;
    __shared__ double red[256];
    const int n_ev = *n_list;
    // unit of work: one event.  The candidate flag is an OR over the event's rays, so once one ray exceeds the cut the
    // remaining undecided rays keep their "at most" value (unless every maximum is wanted).
    for (int le = blockIdx.x; le < n_ev; le += gridDim.x) {
      const int e = ev_list[le];
      const int r0 = slot_offset[e], r1 = slot_offset[e + 1];
      bool done = false;
      for (int r = r0; r < r1 && !done; r++) {
        if (!need_ray[r]) continue;
        if (threadIdx.x == 0) rs.ask = w.ask[r];
        for (int i = threadIdx.x; i < st.n_fc; i += blockDim.x) rs.att[i] = w.att[(long)r * st.n_fc + i];
        __syncthreads();
        fill_amplitude(amp, st, rs);
        const double2 rt = w.r_theta[r], rp = w.r_phi[r];
        const double pt = w.pol_theta[r], pp = w.pol_phi[r];
        const bool both_real = (rt.y == 0. && rp.y == 0.);
        double mx = 0.;
        const double scale = st.fs / 1.4142135623730951 / nh;
        if (both_real) {
            field_time_domain<NRHIP_EFIELD_FUSE>(x, amp, N, st.np, st.fs, 1.0, make_double2(1., 0.), 0., false, ask_model,
                              floor(2.0 * st.fs), tw);
            double cm = fmax(fabs(pt * rt.x), fabs(pp * rp.x));
            for (int j = threadIdx.x; j < nh; j += blockDim.x) {
                double2 y = x[j];
                mx = fmax(mx, fmax(fabs(y.x * scale), fabs(y.y * scale)));
            }
            mx *= cm;
            __syncthreads();
        } else {
            for (int comp = 0; comp < 2; comp++) {
                field_time_domain<NRHIP_EFIELD_FUSE>(x, amp, N, st.np, st.fs, comp ? pp : pt, comp ? rp : rt, 0., false, ask_model,
                                  floor(2.0 * st.fs), tw);
                for (int j = threadIdx.x; j < nh; j += blockDim.x) {
                    double2 y = x[j];
                    mx = fmax(mx, fmax(fabs(y.x * scale), fabs(y.y * scale)));
                }
                __syncthreads();
            }
        }
        mx = block_max(mx, red);
        if (threadIdx.x == 0) {
            max_efield[r] = mx;
            if (xform_count) atomicAdd(&xform_count[2], 1ULL);
        }
        if (!exact && mx > min_efield) done = true;  // block-uniform (block_max broadcasts)
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernel: per event common time grid (efieldToVoltageConverter.py:120-169) + candidate flag
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
event_grid_kernel(int n_events, int n_ch, const int* __restrict__ slot_offset, RayWork w, StationDev st,
                  const double* __restrict__ max_efield, double min_efield, EventOut ev)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    int r0 = slot_offset[e], r1 = slot_offset[e + 1];  // ray range of the event group (group_ray_range_kernel)
    double tmin = INFINITY, tmax = -INFINITY;
    int cand = 0;
    for (int r = r0; r < r1; r++) {
        double t0 = w.t0[r] + st.cable[w.ch[r]];
        tmin = fmin(tmin, t0);
        tmax = fmax(tmax, t0 + st.N / st.fs);
        if (max_efield[r] > min_efield) cand = 1;
    }
    int L = 0;
    if (r1 > r0) {
        double max_len = st.readout_length;  // longest detector readout window (n_samples / sampling rate)
        tmin -= st.pre_pulse;
        tmax += st.post_pulse;
        while (tmax - tmin < max_len) tmax += st.post_pulse;
        double res = 1. / st.fs;
        L = (int)rint((tmax - tmin) / res);
        if (L % 2 != 0) L += 1;
    }
    ev.n_rays[e] = r1 - r0;
    ev.ray_begin[e] = r0;
    ev.candidate[e] = (unsigned char)(cand && r1 > r0);
    ev.L[e] = L;
    ev.t_min[e] = (r1 > r0) ? tmin : NAN;
}

// candidate events -> int flags, flags of the trace lengths in use (index L / 2), number of rays in candidate events
__global__ void __launch_bounds__(256)
candidate_flags_kernel(int n_events, int n_half, EventOut ev, int* __restrict__ cflag, int* __restrict__ lflag,
                       long long* __restrict__ n_cand_rays)
{
    // grid-stride, ONE atomic per block: one per wave of a million-event list is 15 625 atomics on one word -- 0.18 ms at the
    // ~88 per microsecond a single address takes, most of this kernel's time
    __shared__ int s_nr[4];
    int nr = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e <= n_events; e += gridDim.x * blockDim.x) {
        if (e == n_events) { cflag[e] = 0; break; }
        int c = ev.candidate[e] ? 1 : 0;
        cflag[e] = c;
        if (c) {
            const int h = ev.L[e] / 2;
            if (h < n_half) lflag[h] = 1;
            else atomicMax((unsigned long long*)(n_cand_rays + 1), (unsigned long long)h);  // too long: the host reports it
            nr += ev.n_rays[e];
        }
    }
    nr = wave_sum_i32(nr);
    if ((threadIdx.x & 63) == 0) s_nr[threadIdx.x >> 6] = nr;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long tot = (long long)s_nr[0] + s_nr[1] + s_nr[2] + s_nr[3];
        if (tot) atomicAdd((unsigned long long*)n_cand_rays, (unsigned long long)tot);
    }
}

// compacted candidate list (event order), ascending list of distinct lengths, per-event index into it
__global__ void __launch_bounds__(256)
candidate_lists_kernel(int n_events, int n_half, EventOut ev, const int* __restrict__ cflag, const int* __restrict__ coff,
                       const int* __restrict__ lflag, const int* __restrict__ loff, int* __restrict__ cand,
                       int* __restrict__ len_index, int* __restrict__ lens)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_events) {
        int li = -1;
        if (cflag[i]) {
            cand[coff[i]] = i;
            const int h = ev.L[i] / 2;
            li = (h < n_half) ? loff[h] : -1;
        }
        len_index[i] = li;
    }
    if (i < n_half && lflag[i]) lens[loff[i]] = 2 * i;
}

// complex rational filter response prod_i B_i(j f) / A_i(j f) applied successively (signal.freqs)
__device__ inline double2 apply_filters(double2 v, double f, const FilterSet& fl)
{
    for (int i = 0; i < fl.n; i++) {
        if (fl.kind[i] == 2) {  // rectangular
            if (!(fl.b[i][0] <= f && f <= fl.b[i][1])) return make_double2(0., 0.);
            continue;
        }
        if (fl.kind[i] == 4) continue;  // gaussian_tapered depends on the whole grid: applied by the table builder
        if (fl.kind[i] == 3) {
            // measured response (RNO_G/analog_components.py:83-104): scipy interp1d(ff, gain, fill_value=0) and
            // interp1d(ff, unwrap(phase), fill_value=0), gain times the temperature correction c0 + c1 f^5
            const double* t = fl.pool + 3 * (long)fl.na[i];
            const int n = fl.nb[i];
            double g = 0., ph = 0.;
            if (f >= t[0] && f <= t[3 * (n - 1)]) {
                int lo = 0, hi = n - 1;   // largest lo with t[lo] <= f (searchsorted side='left' minus one, clipped as interp1d does)
                while (hi - lo > 1) {
                    int mid = (lo + hi) / 2;
                    if (t[3 * mid] < f) lo = mid;
                    else hi = mid;
                }
                const double x0 = t[3 * lo], x1 = t[3 * hi];
                const double sl_g = (t[3 * hi + 1] - t[3 * lo + 1]) / (x1 - x0), sl_p = (t[3 * hi + 2] - t[3 * lo + 2]) / (x1 - x0);
                g = sl_g * (f - x0) + t[3 * lo + 1];
                ph = sl_p * (f - x0) + t[3 * lo + 2];
            }
            g *= fl.b[i][0] + fl.b[i][1] * (f * f * f * f * f);
            double sn, cs;
            sincos(ph, &sn, &cs);
            v = cmul(v, make_double2(g * cs, g * sn));
            continue;
        }
        if (!(f > 0)) return make_double2(0., 0.);
        double2 num = make_double2(0., 0.), den = make_double2(0., 0.);
        const double2 jw = make_double2(0., f);
        for (int k = 0; k < fl.nb[i]; k++) num = cadd(cmul(num, jw), make_double2(fl.b[i][k], 0.));
        for (int k = 0; k < fl.na[i]; k++) den = cadd(cmul(den, jw), make_double2(fl.a[i][k], 0.));
        double dd = den.x * den.x + den.y * den.y;
        double2 h = make_double2((num.x * den.x + num.y * den.y) / dd, (num.y * den.x - num.x * den.y) / dd);
        if (fl.kind[i] == 1) h = make_double2(cabs2(h), 0.);
        v = cmul(v, h);
    }
    return v;
}

// (see czt_inverse_blocks)
__host__ __device__ inline bool czt_inverse_chunked(int m, int M) { return m + 1 > M - 1024; }

// ---------------------------------------------------------------------------------------------------------
// kernel: per distinct trace length L -- Bluestein tables and the analytic antenna magnitudes on the L grid
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
length_tables_kernel(int n_len, const int* __restrict__ lengths, const int* __restrict__ slots, StationDev st,
                     const FilterSet* __restrict__ fls,
                     const double2* __restrict__ tw, const double2* __restrict__ w16, LengthTables tab)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    __shared__ double red[1024];
    const int M = FFT_MAX, nh = st.N / 2;
    for (int il = blockIdx.x; il < n_len; il += gridDim.x) {
        const int L = lengths[il], m = L / 2;
        const long is = slots ? slots[il] : il;   // row of the tables this length lives in
        // forward transform of the packed N/2-point block onto m output bins, modulus m, sign -1; when nh + m - 1 exceeds the
        // transform length the outputs come in blocks of M - nh + 1 (channel_kernel), all served by this one table
        czt_build_table(x, FFT_LOG2_MAX, nh, min(m, M - nh + 1), m, -1., tw);
        for (int i = threadIdx.x; i < M; i += blockDim.x) tab.B_fwd[is * M + i] = x[i];
        __syncthreads();
        // inverse transform of m + 1 bins onto blocks of P = M - (m + 1) + 1 samples, modulus L, sign +1 (chunks of M / 2 bins onto
        // blocks of M / 2 samples for the longest traces: czt_inverse_blocks)
        if (czt_inverse_chunked(m, M)) czt_build_table(x, FFT_LOG2_MAX, M / 2, M / 2, L, +1., tw);
        else czt_build_table(x, FFT_LOG2_MAX, m + 1, M - m, L, +1., tw);
        for (int i = threadIdx.x; i < M; i += blockDim.x) tab.B_inv[is * M + i] = x[i];
        __syncthreads();
        // every phase factor of this length: E[j] = exp(-2 pi i j / (2 L)); filter chain on the L grid
        for (int j = threadIdx.x; j < 2 * L; j += blockDim.x) {
            double sn, cs;
            sincospi((double)j / (double)L, &sn, &cs);
            tab.E[is * NRHIP_E_STRIDE + j] = make_double2(cs, -sn);
        }
        const double df = 1.0 / (L * (1. / st.fs));
        for (int fs = 0; fs < st.n_fsets; fs++) {
            const FilterSet& fl = fls[fs];
            double2* Hs = tab.H + (is * st.n_fsets + fs) * NRHIP_SPEC_STRIDE;
            for (int k = threadIdx.x; k <= m; k += blockDim.x) Hs[k] = apply_filters(make_double2(1., 0.), k * df, fl);
            for (int i = 0; i < fl.n; i++) {
                if (fl.kind[i] != 4) continue;
                // gaussian_tapered (signal_processing.py:310-321) on THIS grid of n = m + 1 frequencies: the pass band
                // (1 inside b[0] .. b[1]) convolved (mode 'same') with signal.windows.gaussian(n, int(round(roll_width / df))),
                // divided by its maximum.  Direct sums over the pass-band bins; terms beyond 40 sigma are exactly 0 in binary64.
                const int n = m + 1, c = (n - 1) / 2;
                const double sigma = (double)(int)rint(fl.b[i][2] / df), mid = 0.5 * (n - 1);
                int j_lo = (int)ceil(fl.b[i][0] / df), j_hi = (int)floor(fl.b[i][1] / df);
                while (j_lo > 0 && (j_lo - 1) * df >= fl.b[i][0]) j_lo--;       // exactly the bins with b0 <= k df <= b1
                while (j_lo * df < fl.b[i][0]) j_lo++;
                while (j_hi < m && (j_hi + 1) * df <= fl.b[i][1]) j_hi++;
                while (j_hi >= 0 && j_hi * df > fl.b[i][1]) j_hi--;
                if (j_lo < 0) j_lo = 0;
                if (j_hi > m) j_hi = m;
                double* gt = (double*)smem;   // the tapered pass band on the grid
                double lmax = 0.;
                for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                    // full convolution index k + c, window index t = k + c - j for pass-band bin j
                    double acc_ = 0.;
                    int ja = j_lo, jb = j_hi;
                    const double reach = 40. * sigma + 1.;
                    const int t_hi = (int)floor(mid + reach), t_lo = (int)ceil(mid - reach);   // |t - mid| <= reach
                    if (k + c - ja > t_hi) ja = k + c - t_hi;
                    if (k + c - jb < t_lo) jb = k + c - t_lo;
                    for (int j = ja; j <= jb; j++) {
                        const int t = k + c - j;
                        if (t < 0 || t >= n) continue;
                        const double u = (t - mid) / sigma;
                        acc_ += exp(-0.5 * (u * u));
                    }
                    gt[k] = acc_;
                    lmax = fmax(lmax, acc_);
                }
                red[threadIdx.x] = lmax;
                __syncthreads();
                for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) {
                    if ((int)threadIdx.x < s_) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s_]);
                    __syncthreads();
                }
                const double gmax = red[0];
                for (int k = threadIdx.x; k <= m; k += blockDim.x) Hs[k] = cscale(Hs[k], gt[k] / gmax);
                __syncthreads();
            }
        }
        __syncthreads();   // the H tables are read back below (same block)
        for (int k = threadIdx.x; k < NRHIP_SPEC_STRIDE; k += blockDim.x)
            tab.Cf[is * NRHIP_SPEC_STRIDE + k] = chirp(k, m, -1.);
        for (int n = threadIdx.x; n < FFT_MAX; n += blockDim.x) tab.Ci[is * FFT_MAX + n] = chirp(n, L, +1.);
        // analytic antenna magnitude * phase on the L grid (antennapattern.py:1672-1768), models 0 VPol, 1 HPol;
        // the "remove DC offset" cut below 5 MHz (efieldToVoltageConverter.py:313) is folded in
        for (int model = 0; model < NRHIP_N_ANT_TAB; model++) {
            if (!((st.tab_mask >> model) & 1)) continue;
            __syncthreads();
            double* mag = (double*)smem;
            int index = 0;
            if (model != 1) {  // np.argmax(freq > cutoff): first bin above 220 MHz (VPol) / 110 MHz (LPDA), 0 if none
                const double cutoff = (model == 0) ? 0.22 : 0.11;
                index = m + 1;
                for (int k = 0; k <= m; k++) {
                    if (k * df > cutoff) { index = k; break; }
                }
                if (index == m + 1) index = 0;
            }
            double lmax = 0.;
            for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                double f = k * df, v = 0.;
                if (k > 0) {
                    if (model != 1) {
                        double gain = (model == 0) ? 1.0 / sqrt(f) : 1.0;   // LPDA: flat gain
                        v = sqrt(gain) / f;
                        if (k < index) v *= 0.5 - 0.5 * cos(2. * M_PI * k / (2 * index - 1));  // hann(2 index)[k]
                    } else {
                        double sn = sin(f / 0.5 * M_PI / 2);
                        v = 1.0 * (sn * sn);
                        if (f > 0.5 * 2) v = 0.;
                    }
                    lmax = fmax(lmax, v);
                }
                mag[k] = v;
            }
            red[threadIdx.x] = lmax;
            __syncthreads();
            for (int s = blockDim.x / 2; s > 0; s >>= 1) {
                if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
                __syncthreads();
            }
            double vmax = red[0];
            double max_vel = model == 0 ? 0.18 : (model == 1 ? 0.055 : 0.55);
            double h2s[NRHIP_MAX_FSETS] = {0., 0., 0., 0.};
            for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                double f = k * df, v = mag[k];
                if (k > 0) v *= max_vel / vmax;
                double ph;
                if (model == 0) ph = 2.086 - 117.917 * f + 74.567 / 2 * (f * f) - 64.343 / 3 * (f * f * f);
                else if (model == 1) ph = 0.321 - 11.400 * f + 39.590 / 2 * (f * f) - 38.181 / 3 * (f * f * f);
                else if (model == 2) {  // parametric_phase (antennapattern.py:1643-1650): front lobe, side, back
                    ph = 100 * ((f - 0.4) * (f - 0.4)) - 20;
                    if (f > 0.4) ph -= 0.00007 * ((f - 0.4) * (f - 0.4));
                } else if (model == 3) ph = 40 * ((f - 0.95) * (f - 0.95)) - 40;
                else ph = 50 * ((f - 0.95) * (f - 0.95)) - 50;
                double sn, cs;
                sincos(ph, &sn, &cs);
                if (f < 0.005) v = 0.;
                double2 vv = make_double2(v * cs, v * sn);
                tab.vel[(is * NRHIP_N_ANT_TAB + model) * NRHIP_SPEC_STRIDE + k] = vv;
                // |antenna x filter|^2 for the impulse-response norm (irfft keeps only the real part of DC / Nyquist)
                for (int fs = 0; fs < st.n_fsets; fs++) {
                    double2 hk = cmul(vv, tab.H[(is * st.n_fsets + fs) * NRHIP_SPEC_STRIDE + k]);
                    h2s[fs] += (k == 0 || k == m) ? hk.x * hk.x : 2. * (hk.x * hk.x + hk.y * hk.y);
                }
            }
            __syncthreads();
            for (int fs = 0; fs < st.n_fsets; fs++) {
                const double h2 = block_sum(h2s[fs], red);
                if (threadIdx.x == 0) tab.hnorm[(is * st.n_fsets + fs) * NRHIP_N_ANT_TAB + model] = sqrt(h2 / L);
            }
        }
        // Lengths up to FFT_MAX: the channel voltage is the circular convolution (period L) of the summed, placed field
        // traces with g = (fs / sqrt 2) irfft_L(antenna x filter).  g comes from one chirp-z inverse; its spectrum on the
        // 2 FFT_MAX-point grid (real transform via the packed FFT_MAX-point complex one) is what channel_conv_kernel
        // multiplies with.  Factors folded in: 1/2 of each even/odd split (two of them), 1/FFT_MAX of the inverse.
        if (tab.G && L <= FFT_MAX) {
            const double2* E = tab.E + is * NRHIP_E_STRIDE;
            const double2* Ci = tab.Ci + is * FFT_MAX;
            const double2* Bi = tab.B_inv + is * M;
            const unsigned LL = (unsigned)L;
            const int P = M - m;
            const double scale = st.fs / 1.4142135623730951 / L;
            for (int fm = 0; fm < st.n_fsets * NRHIP_N_ANT_TAB; fm++) {
                const int fs = fm / NRHIP_N_ANT_TAB, model = fm % NRHIP_N_ANT_TAB;
                if (!((st.fset_tab_mask[fs] >> model) & 1)) continue;
                const double2* Hf = tab.H + (is * st.n_fsets + fs) * NRHIP_SPEC_STRIDE;
                const double2* vel = tab.vel + (is * NRHIP_N_ANT_TAB + model) * NRHIP_SPEC_STRIDE;
                double2* G = tab.G + ((is * st.n_fsets + fs) * NRHIP_N_ANT_TAB + model) * NRHIP_G_STRIDE;
                double* gtmp = (double*)G;  // L doubles of the impulse response, overwritten by its spectrum below
                __syncthreads();
                for (int n0 = 0; n0 < L; n0 += P) {
                    for (int k = threadIdx.x; k < M; k += blockDim.x) {
                        double2 v = make_double2(0., 0.);
                        if (k <= m) {
                            v = cmul(vel[k], Hf[k]);
                            if (k == 0 || k == m) v = make_double2(v.x, 0.);
                            else v = cscale(v, 2.);
                            if (n0 != 0) {
                                unsigned kn = ((unsigned)k * (unsigned)n0) % LL;
                                v = cmul(v, cconj(E[2 * kn]));
                            }
                            v = cmul(v, Ci[k]);
                        }
                        x[k] = v;
                    }
                    __syncthreads();
                    czt_convolve(x, FFT_LOG2_MAX, Bi, tw);
                    int np = min(P, L - n0);
                    for (int n = threadIdx.x; n < np; n += blockDim.x) {
                        double2 u = cmul(x[n], Ci[n]);
                        gtmp[n0 + n] = u.x * (1.0 / M) * scale;
                    }
                    __syncthreads();
                }
                for (int j = threadIdx.x; j < M; j += blockDim.x)
                    x[j] = (j < m) ? make_double2(gtmp[2 * j], gtmp[2 * j + 1]) : make_double2(0., 0.);
                __syncthreads();
                fft_dif(x, FFT_LOG2_MAX, tw, false);
                const double nrm = 1.0 / (8.0 * M);
                for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
                    int p = bitrev(k, FFT_LOG2_MAX), q = (k == 0) ? p : bitrev(M - k, FFT_LOG2_MAX);
                    double2 A = x[p], Bc = cconj(x[q]);
                    double2 Ee = cadd(A, Bc), D = csub(A, Bc);
                    double2 O = make_double2(D.y, -D.x);
                    double2 wO = cmul(w16[k], O);
                    double2 Xk = cscale(cadd(Ee, wO), nrm), Xm = cscale(cconj(csub(Ee, wO)), nrm);
                    if (k == 0) { Xk.y = 0.; Xm.y = 0.; }
                    G[k] = Xk;
                    G[M - k] = Xm;
                }
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernel: one thread per (candidate event, channel) item -- Cauchy-Schwarz prefilter.  The channel trace is
// sum_r vfac_r (e_r (*) g), so |V(t)| <= ||g||_2 sum_r |vfac_r| ||e_r||_2.  Items that cannot reach the threshold report the
// negated bound; the others are flagged for channel_conv_kernel (L <= FFT_MAX only; longer traces stay with channel_kernel).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
channel_prefilter_kernel(int n_items, const int* __restrict__ item_event, RayWork w, EventOut ev,
                         const int* __restrict__ ev_len_index, StationDev st, double threshold, const double* __restrict__ hnorm,
                         int exact, double* __restrict__ maxV, int* __restrict__ need, int skip_off)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const int e = item_event[item / st.n_ch], ch = item % st.n_ch;
    const int L = ev.L[e], il = ev_len_index[e];
    if (L > FFT_MAX || st.ant_model[ch] == 3) { need[item] = 0; maxV[item] = NAN; return; }  // chirp-z kernel: long traces, tabulated patterns
    if (skip_off && st.trig_on && !st.trig_on[ch]) {  // not a trigger channel: nothing of it decides anything
        need[item] = 0;
        maxV[item] = NAN;
        return;
    }
    int flag = 1;
    if (!exact) {
        const int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
        double bnd = 0.;
        for (int r = r0; r < r1; r++) {
            if (w.ch[r] != ch) continue;
            bnd += w.e_norm[r] * (fabs(w.vfac_t[r] * w.pol_theta[r]) * cabs2(w.r_theta[r]) +
                                  fabs(w.vfac_p[r] * w.pol_phi[r]) * cabs2(w.r_phi[r])) *
                   hnorm[((long)il * st.n_fsets + (st.ch_fset ? st.ch_fset[ch] : 0)) * NRHIP_N_ANT_TAB + w.tab[r]];
        }
        maxV[item] = -bnd;   // pruned items keep it; for the others channel_conv_kernel orders the event's channels by it
        if (!(bnd * (1 + 1e-9) >= threshold)) flag = 0;
    }
    need[item] = flag;
}

// phased-array trigger: the window power of a beam is at most (window / divisor) (sum over the array's channels of max |V_c|)^2, so
// an event whose channel bounds (channel_prefilter_kernel, -maxV) add up to less than amp_cut cannot trigger: none of its traces
// is needed (they stay zero, its beam powers read 0 = "below the threshold")
__global__ void pa_event_prune_kernel(int n_cand, int n_ch, const unsigned char* __restrict__ trig_on, const double* __restrict__ maxV,
                                      int* __restrict__ need, double amp_cut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    double sum = 0.;
    for (int ch = 0; ch < n_ch; ch++)
        if (!trig_on || trig_on[ch]) {
            const double b = -maxV[(long)i * n_ch + ch];
            sum += (b == b) ? b : INFINITY;   // (a channel the prefilter could not bound keeps the event)
        }
    if (!(sum * (1 + 1e-9) >= amp_cut))
        for (int ch = 0; ch < n_ch; ch++) need[(long)i * n_ch + ch] = 0;
}

__global__ void scatter_item_list_kernel(int n_items, const int* __restrict__ need, const int* __restrict__ offset,
                                         int* __restrict__ list)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    if (need[i]) list[offset[i]] = i;
}

// The event list of channel_conv_kernel in the order of the events' trace lengths (round 6).  The response spectrum G_L a channel is
// multiplied with depends on the event's length L alone (1075 distinct lengths in 1e6 events of the survey, 131 KB each): handed out in
// list order, the 256 resident blocks worked on 256 different lengths and every channel fetched its G_L from HBM (10.7 of the
// kernel's 15.6 GB per launch, and an HBM latency in the spectrum pass); in length order neighbouring blocks share it in the L2.
// A counting sort on L / 2: histogram, scan, scatter through per-length cursors (the order inside a length is whatever the atomics
// give -- the events are independent).
__global__ void length_hist_kernel(const int* __restrict__ n_list, const int* __restrict__ list, const int* __restrict__ item_event,
                                   const int* __restrict__ ev_L, int* __restrict__ hist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_list) return;
    atomicAdd(&hist[min(ev_L[item_event[list[i]]] >> 1, FFT_MAX / 2)], 1);
}
__global__ void length_scatter_kernel(const int* __restrict__ n_list, const int* __restrict__ list, const int* __restrict__ item_event,
                                      const int* __restrict__ ev_L, int* __restrict__ cursor, int* __restrict__ sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_list) return;
    const int c = list[i];
    sorted[atomicAdd(&cursor[min(ev_L[item_event[c]] >> 1, FFT_MAX / 2)], 1)] = c;
}

// candidate events with at least one channel left to evaluate (the unit of work of channel_conv_kernel)
// n_coinc > 1 (majority logic, production mode): an n-fold coincidence needs n channels that can raise a flag at all -- an event
// with fewer channels left by the prefilter cannot trigger, none of its channels is evaluated (their maxima keep the bound)
__global__ void channel_event_flags_kernel(int n_cand, int n_ch, int* __restrict__ need, int* __restrict__ ev_need, int n_coinc)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    int cnt = 0;
    for (int ch = 0; ch < n_ch; ch++) cnt += need[c * n_ch + ch] != 0;
    if (cnt > 0 && cnt < n_coinc) {
        for (int ch = 0; ch < n_ch; ch++) need[c * n_ch + ch] = 0;
        cnt = 0;
    }
    ev_need[c] = cnt > 0;
}

// ---------------------------------------------------------------------------------------------------------
// kernel: one (candidate event, channel) item per block iteration, trace lengths L <= FFT_MAX and N <= FFT_MAX / 2.
//   S[n] (LDS, real, period L) = sum over the channel's rays and on-sky components of
//                                vfac * (N-point field in the time domain, with sub-sample shift) placed at the start bin
//   V = S (*)_L g  through ONE real 2 FFT_MAX-point transform pair (packed complex FFT_MAX-point FFTs in LDS):
//       rfft(S zero-padded) * G  ->  irfft  ->  V[n] = y[n] + y[n + L]
//   |V| >= threshold (last sample excluded, see majority logic)
// Mathematically the reference's rfft_L / * VEL / sum / * filter / irfft_L (efieldToVoltageConverter.py:214-341 and
// channelBandPassFilter), without any length-L transform per item.  LDS: FFT_MAX complex (128 KB); the N/2-point field
// buffer and the amplitude array live in its upper half until the big transform starts.
// ---------------------------------------------------------------------------------------------------------
#ifndef CONV_NT
#define CONV_NT 512
#endif
#define CONV_MAX_ORDER 64  // stations with more channels are evaluated in channel order
#define CONV_STAGE_RAYS 192  // rays of an event whose channel / antenna-table numbers are kept in LDS (more: read from HBM per channel)
// threads per block: 512 for the full-capacity kernel (one block per CU), 256 for the half-capacity one (two blocks per CU: the same
// eight waves per CU, each with the 256 registers the transforms want, but two independent barrier domains)
#define CONV_THREADS(log2cap) ((log2cap) == FFT_LOG2_MAX ? CONV_NT : CONV_NT / 2)
// What channel_conv_kernel needs to know about a candidate event before it can start on it, in ONE record per list entry (round 6).
// The kernel used to find these out by itself: thread 0 took a list index from the queue, then read list -> candidate -> event ->
// length, then the needed channels and their bounds one by one for the best-first order; a barrier; then every thread read the
// event's table index, ray range and start time -- eight dependent trips to the L2 / HBM per event with the whole block waiting
// (10 % of the kernel's time).  Now a block claims `claim` consecutive list entries with one atomic and a wave reads their
// records with one request per lane.
#define CONV_CLAIM_MAX 8
struct ConvHdr {
    int c, e, L, il, r0, nre;        // candidate index, event, common trace length, its table row, first ray, number of rays
    double t_min;
    int n_order, pad[3];             // channels in `order` (best first), or the number of needed channels (coincidence, > CONV_MAX_ORDER)
    unsigned char order[CONV_MAX_ORDER];
};
static_assert(sizeof(ConvHdr) == 7 * 16, "seven 16-byte words per record");
__global__ void __launch_bounds__(256)
conv_header_kernel(const int* __restrict__ n_list, const int* __restrict__ list, const int* __restrict__ item_event, EventOut ev,
                   const int* __restrict__ ev_len_index, int n_ch, const int* __restrict__ need, const double* __restrict__ maxV,
                   int best_first, int coinc, int exact, ConvHdr* __restrict__ hdr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_list) return;
    ConvHdr h;
    const int c = list[i], e = item_event[c];
    h.c = c; h.e = e; h.L = ev.L[e]; h.il = ev_len_index[e]; h.r0 = ev.ray_begin[e]; h.nre = ev.n_rays[e]; h.t_min = ev.t_min[e];
    h.pad[0] = h.pad[1] = h.pad[2] = 0;
    for (int k = 0; k < CONV_MAX_ORDER; k++) h.order[k] = 0;
    int cnt = 0;
    const int base = c * n_ch;
    if (best_first) {
        // plain OR: the channel with the largest bound first (the event is done at the first trigger).  Coincidence: the WEAKEST
        // first -- the kernel's early stop counts silent channels, and the likely loud one is then the one it saves
        // (the prefilter leaves -bound in maxV)
        for (int ch = 0; ch < n_ch; ch++) {
            if (!need[base + ch]) continue;
            const double b = coinc ? maxV[base + ch] : -maxV[base + ch];
            int k = cnt++;
            while (k > 0 && (coinc ? maxV[base + h.order[k - 1]] : -maxV[base + h.order[k - 1]]) < b) {
                h.order[k] = h.order[k - 1];
                k--;
            }
            h.order[k] = (unsigned char)ch;
        }
    } else if (coinc && !exact) {   // (more channels than the ordered list holds: channel order, counted)
        for (int ch = 0; ch < n_ch; ch++) cnt += need[base + ch] != 0;
    }
    h.n_order = cnt;
    hdr[i] = h;
}

// one N/2-point transform of the convolution kernel: an on-sky component of a ray (or both at once when the reflection
// coefficients are real), where it starts on the event's grid and what it is scaled with
struct ConvJob {
    int r, sbin, shift;
    double pol, vfac, rem;
    double2 rc;
};
// dynamic LDS of channel_conv_kernel<log2cap>: the padded buffer and, behind it, the emission constants / attenuation rows of the
// rays of a batch (4 with 512 threads, 2 with 256)
static inline int conv_lds_bytes(int log2cap)
{
    return conv_lds_elems(1 << log2cap) * 16 + (log2cap == FFT_LOG2_MAX ? 4 : 2) * (int)sizeof(RayShared);
}
// amplitude X_k att(f_k) of bin k exactly as fill_amplitude() forms it
__device__ inline double conv_amplitude(int k, int nh, double df, const StationDev& st, const RayShared& rs, double pl, double pr, int seg)
{
    if (k <= 0 || k >= nh) return 0.;
    const double f = k * df;
    if (rs.ask.model == 0) {
        const double x = pl * rs.ask.cL, y = pr * rs.ask.cR;   // amplitude_bin, model 0
        return rs.ask.pref2 * f / ((1 + x) * (1 + y)) * interp_seg(f, seg, st.n_fc, rs.xp, rs.att, rs.slope);
    }
    return amplitude_bin(k, f, rs.ask, st) * interp_seg(f, seg, st.n_fc, rs.xp, rs.att, rs.slope);
}
#pragma clang fp contract(off)   // (see conv_fft.h: the kernel's two instantiations must do the same arithmetic)
// ---- wave-private ray transforms of the convolution kernel (conv_fft.h): spectrum + first stages, and last stages + placement ----
// The rare emission models (Alvarez2000, ZHS1992: exp / log per bin, ZHS's own phase) stay out of line: inlined into every bin of
// every unrolled item they made the kernel three times its size (instruction fetch) for a path the surveys never take.
__device__ __noinline__ double conv_amplitude_rare(int k, double f, const AskaryanConst* a, const double* lnf)
{
    return askaryan_amplitude(f, lnf[k], *a);
}
__device__ __noinline__ double2 conv_zhs_phase(int k, double roll, int N)
{
    double sn, cs;
    sincospi(-2.0 * k * roll / N, &sn, &cs);
    return make_double2(-sn, cs);
}
// The packed spectrum x_k = ge_k + i go_k of field_time_domain() is built by the thread that runs the first log2(NBK) stages on it:
// item i0 (0 < i0 < 256) owns the bins i0 + 512 j of the transform and their mirror partners nh - k = (512 - i0) + 512 (NBK - 1 - j)
// (both members of a mirror pair need both amplitudes); item 0 the two groups that are their own mirrors (i0 = 0 and 256).  The phase
// ramp exp(-2 pi i f_k rem) of the sub-sample shift comes from two sincospi per thread (bin lt and the thread stride) and products.
// (Tried as real function calls like the passes of conv_fft.h -- the interface is made for it: values, and LDS places as byte offsets
// into the dynamic LDS -- and measured slower, 14.4 against 12.2 ms: the spill traffic moves to the call sites.)  The station's
// scalars and tables a ray needs:
struct RayStation { int N, n_fc; double fs; const double* fpow; const unsigned char* seg; const double* lnf; };
template <int NBK>
__device__ __forceinline__ void ray_build(int xjob_off, int rg_off, int lt, const ConvJob job, const RayStation st, int ask_model,
                                       const double2* __restrict__ tw, int log2nh)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* xjob = (double2*)(smem + xjob_off);
    const RayShared& rg = *(const RayShared*)(smem + rg_off);
    constexpr int TR = 64 * NBK, IT = 4 / NBK, LWR = (NBK == 4) ? 2 : 1, BS = ray_blk_stride(NBK);
    static_assert(NBK == 2 || NBK == 4, "N / 2 = 1024 or 2048");
    const int N = st.N, nh = N / 2, stride = nh + 1, off_l = rg.ask.had ? 0 : stride;
    const double df = 1.0 / (N * (1. / st.fs));
    const bool m0 = rg.ask.model == 0;
    const double roll = floor(2.0 * st.fs);
    const int twn = FFT_MAX / N;
    const double al = job.shift ? -2. * job.rem * df : 0.;
    double2 rho_a = make_double2(1., 0.), rho_t = rho_a, rho_512 = rho_a, rho_nh = rho_a;
    if (job.shift) {
        double sn, cs;
        sincospi(al * lt, &sn, &cs);
        rho_a = make_double2(cs, sn);
        sincospi(al * TR, &sn, &cs);
        rho_t = make_double2(cs, sn);
        const double2 r2 = cmulx(rho_t, rho_t), r4 = cmulx(r2, r2);
        rho_nh = cmulx(r4, r4);                           // nh = 8 TR
        rho_512 = (NBK == 4) ? r2 : r4;
    }
    // station tables of a mirror pair (k, nh - k), 0 < k < nh: requested for all pairs of an item before the first is used
    struct PairIn { double pl1, pr1, pl2, pr2; int sg1, sg2; double2 w1, w2; };
    auto pair_load = [&](int k) -> PairIn {
        const int k2 = nh - k;
        PairIn q;
        q.pl1 = st.fpow[off_l + k]; q.pr1 = st.fpow[2 * stride + k]; q.sg1 = st.seg[k]; q.w1 = gload(&tw[k * twn]);
        q.pl2 = st.fpow[off_l + k2]; q.pr2 = st.fpow[2 * stride + k2]; q.sg2 = st.seg[k2]; q.w2 = gload(&tw[k2 * twn]);
        return q;
    };
    // amplitude X_k att(f_k) (conv_amplitude), 0 < k < nh; the quotient through a reciprocal estimate and two Newton steps (~1 ulp)
    auto ampl = [&](int k, double pl, double pr, int seg) -> double {
        const double f = k * df;
        // (interp_seg with its multiply-add written out)
        const double at = (f <= rg.xp[0]) ? rg.att[0] : ((f >= rg.xp[st.n_fc - 1]) ? rg.att[st.n_fc - 1] : fma(rg.slope[seg], f - rg.xp[seg], rg.att[seg]));
        if (m0) {
            const double den = fma(pl, rg.ask.cL, 1.0) * fma(pr, rg.ask.cR, 1.0);
            double rc = __builtin_amdgcn_rcp(den);
            rc = fma(fma(-den, rc, 1.0), rc, rc);
            rc = fma(fma(-den, rc, 1.0), rc, rc);
            return rg.ask.pref2 * f * rc * at;
        }
        return conv_amplitude_rare(k, f, &rg.ask, st.lnf) * at;
    };
    // one on-sky component's spectrum bin (field_bin with the ramp value handed in), 0 < k < nh
    auto fbin = [&](int k, double amp, const double2 rk) -> double2 {
        const double a = amp * 1.4142135623730951;
        double2 sv = make_double2(0., (k & 1) ? -a : a);
        if (ask_model == 2) sv = cscale(conv_zhs_phase(k, roll, N), a);
        sv = cscale(sv, job.pol);
        sv = cmulx(sv, job.rc);
        if (job.shift) sv = cmulx(sv, rk);
        return sv;
    };
    // packed values of the mirror pair (k, nh - k); rk = ramp at bin k
    auto pair = [&](int k, const PairIn& q, const double2 rk, double2& Xk, double2& Xm) {
        const int k2 = nh - k;
        const double2 F1 = fbin(k, ampl(k, q.pl1, q.pr1, q.sg1), rk);
        const double2 F2 = fbin(k2, ampl(k2, q.pl2, q.pr2, q.sg2), cmulcx(rho_nh, rk));
        {
            const double2 Gc = cconj(F2);
            const double2 ge = cscale(cadd(F1, Gc), 0.5), d = cscale(csub(F1, Gc), 0.5);
            const double2 go = cmulcx(d, q.w1);
            Xk = make_double2(ge.x - go.y, ge.y + go.x);
        }
        {
            const double2 Gc = cconj(F1);
            const double2 ge = cscale(cadd(F2, Gc), 0.5), d = cscale(csub(F2, Gc), 0.5);
            const double2 go = cmulcx(d, q.w2);
            Xm = make_double2(ge.x - go.y, ge.y + go.x);
        }
    };
    // first LWR stages (spans nh / 2 .. 512) of the inverse transform on the group at i0 (elements i0 + 512 j), then store
    auto stages_store = [&](double2 (&a)[NBK], int i0) {
#pragma unroll
        for (int e = 0; e < LWR; e++) {
            const int half = NBK >> (e + 1);
#pragma unroll
            for (int b = 0; b < NBK; b += 2 * half)
#pragma unroll
                for (int q = 0; q < half; q++) dif_bf_c(a[b + q], a[b + q + half], gload(&tw[(i0 + 512 * q) * (FFT_MAX >> (log2nh - e))]));
        }
#pragma unroll
        for (int j = 0; j < NBK; j++) xjob[j * BS + i0] = a[j];
    };
    // Item 0 takes the two groups that are their own mirrors: bins 256 (j + 1), j < NBK, pair up as (256, nh - 256), (512, nh - 512),
    // ... (nh / 2 with itself; bin 0 is empty) -- as many pairs as any other item has, computed by the SAME instructions with other
    // bin numbers and ramps, the results dealt to the two groups by selects: a branch of its own made the item's wave (and with it
    // every wave at the barrier behind) take twice the time.
    const double2 rho_256 = (NBK == 4) ? rho_t : cmulx(rho_t, rho_t);
    double2 rho_i = rho_a;
#pragma unroll 1
    for (int u = 0; u < IT; u++) {
        const int i0 = lt + TR * u;
        const bool sp = i0 == 0;
        double2 P[NBK], Q[NBK];
        {
            // (the station tables of pair j + 1 are requested before pair j is evaluated: two pairs' worth of registers, not NBK)
            double2 g = rho_i, q = rho_256;
            PairIn nxt = pair_load(sp ? 256 : i0);
#pragma unroll
            for (int j = 0; j < NBK; j++) {
                const int kj = sp ? 256 * (j + 1) : i0 + 512 * j;
                const PairIn cur = nxt;
                if (j + 1 < NBK) {
                    nxt = pair_load(sp ? 256 * (j + 2) : i0 + 512 * (j + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
                pair(kj, cur, make_double2(sp ? q.x : g.x, sp ? q.y : g.y), P[j], Q[j]);
                g = cmulx(g, rho_512);
                q = cmulx(q, rho_256);
            }
        }
        double2 A[NBK], B[NBK];
        const double2 zero = make_double2(0., 0.);
        auto sel = [&](const double2 x, const double2 y) { return make_double2(sp ? x.x : y.x, sp ? x.y : y.y); };
        if (NBK == 4) {
            // general: A[j] = X[i0 + 512 j] = P[j], B[3 - j] = Q[j];  item 0 (P[j] = X[256 (j + 1)], Q[j] = X[nh - 256 (j + 1)]):
            // A = X[0, 512, 1024, 1536] = {0, P1, P3, Q1}, B = X[256, 768, 1280, 1792] = {P0, P2, Q2, Q0}
            A[0] = sel(zero, P[0]); A[1] = P[1]; A[2] = sel(P[3], P[2]); A[3] = sel(Q[1], P[3]);
            B[0] = sel(P[0], Q[3]); B[1] = sel(P[2], Q[2]); B[2] = sel(Q[2], Q[1]); B[3] = Q[0];
        } else {
            // general: A = {P0, P1}, B = {Q1, Q0};  item 0: A = X[0, 512] = {0, P1}, B = X[256, 768] = {P0, Q0}
            A[0] = sel(zero, P[0]); A[1] = P[1];
            B[0] = sel(P[0], Q[1]); B[1] = Q[0];
        }
        stages_store(A, i0);
        stages_store(B, sp ? 256 : 512 - i0);
        rho_i = cmulx(rho_i, rho_t);
    }
}
// the two wave-private passes of a ray transform on the wave's 512-point block (byte offset into the dynamic LDS)
__device__ __forceinline__ void ray_p23(int zb_off, const double2* __restrict__ cft, int lane)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* zb = (double2*)(smem + zb_off);
    ray_p2(zb, cft, lane);
    ray_p3(zb, cft, lane);
}
// last three stages of a ray's transform + placement of its samples on the event's grid: thread lt owns the samples lt + (nh / 8) m
template <int NBK>
__device__ __forceinline__ void ray_place(int xjob_off, int lt, int nh, const ConvJob jq, int L)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const double2* xjob = (const double2*)(smem + xjob_off);
    double* S = (double*)smem;
    constexpr int LWR = (NBK == 4) ? 2 : 1, BS = ray_blk_stride(NBK);
    const int blk = (NBK == 4) ? (((lt & 1) << 1) | ((lt >> 1) & 1)) : (lt & 1);
    const double2* zb = xjob + blk * BS + (lt >> LWR);
    double2 a[8];
#pragma unroll
    for (int c = 0; c < 8; c++) a[c] = zb[65 * c];
    dif8_tail_c(a);
    const double sc = jq.vfac / nh;   // the fs / sqrt(2) of freq2time cancels against time2freq's sqrt(2) / fs
    const int K = nh >> 3;
    // The sixteen samples of a thread are sixteen different places (j differs, N <= L): all of them are read before the first is
    // written.  Written as read-modify-write one by one -- the compiler must assume the places alias -- this was sixteen LDS round
    // trips in a row per thread and job (round 5: 14 % of the kernel's time).
    double* s0[8];
    double* s1[8];
    double v0[8], v1[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int j = lt + K * br3(r);
        int i0 = jq.sbin + 2 * j;
        if (i0 >= L) i0 -= L;
        int i1 = i0 + 1;
        if (i1 >= L) i1 -= L;
        s0[r] = S + 2 * conv_pad(i0 >> 1) + (i0 & 1);
        s1[r] = S + 2 * conv_pad(i1 >> 1) + (i1 & 1);
        v0[r] = *s0[r];
        v1[r] = *s1[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        v0[r] = fma(a[r].x, sc, v0[r]);
        v1[r] = fma(a[r].y, sc, v1[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        *s0[r] = v0[r];
        *s1[r] = v1[r];
    }
}
// Output pass of a channel without coincidence logic: V[n] = (y[n] + y[n + L]) * vscale from the convolution buffer (the kernel's
// dynamic LDS), maximum |V| and threshold flag of this thread, the samples written to `em` (traces of a triggered event) and / or
// `tr` (dump_traces) if given.  Out of line on purpose: inside the kernel -- 400 spilled scalars -- the loop came out with a dozen
// spill reloads and a wait for the previous store per iteration (9 % of the kernel's time for twenty LDS reads per thread).
struct ConvOut { double vmax; int trig; };
__device__ __noinline__ ConvOut conv_output_pass(int L, double vscale, double threshold, int ch_on, double* __restrict__ em,
                                                 double* __restrict__ tr, const double* __restrict__ add, int with_signal)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const double2* z = (const double2*)smem;
    ConvOut o = {0., 0};
    // two samples per step: L is even, so the samples 2 i, 2 i + 1 and their partners L + 2 i, L + 2 i + 1 are two complex elements
    const int hl = L >> 1, nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < hl; i0 += 4 * nt) {   // four steps' LDS reads in flight at a time
        double2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * nt;
            if (i < hl && with_signal) { a[u] = z[conv_pad(i)]; b[u] = z[conv_pad(i + hl)]; }
            else { a[u] = make_double2(0., 0.); b[u] = a[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * nt;
            if (i >= hl) break;
            double v0 = (a[u].x + b[u].x) * vscale, v1 = (a[u].y + b[u].y) * vscale;
            if (add) { v0 += gload(add + 2 * i); v1 += gload(add + 2 * i + 1); }   // thermal noise of the channel (a row of HBM scratch)
            if (em) gstore((double2*)(em + 2 * i), make_double2(v0, v1));
            if (tr) { gstore(tr + 2 * i, v0); gstore(tr + 2 * i + 1, v1); }
            const double a0 = fabs(v0), a1 = fabs(v1);
            o.vmax = fmax(o.vmax, fmax(a0, a1));
            if (ch_on && (a0 >= threshold || (2 * i + 1 < L - 1 && a1 >= threshold))) o.trig = 1;
        }
    }
    return o;
}
// Output pass of a channel WITH coincidence logic: per-channel flags (simpleThreshold.py:14-29 / highLowThreshold.py:13-80),
// OR-dilated over the coincidence window (get_majority_logic :82-150: flag i stays up for w_coinc samples), counted per sample in
// cnt; returns the thread's maximum |V|.  Two passes over the trace, a contiguous run of <= RUN samples per thread: (1) samples (the
// two halves of the linear convolution folded on the fly, read-only), the high / low window as the distances to the last sample
// above / below (the w_hl - 1 samples in front of the run looked at once: O(run + w) instead of O(run x w)), the running index of
// the last raised flag (registers); (2) after a scan over the runs' totals, the coincidence count.  Two barriers (were six).  Out
// of line like conv_output_pass (its own register allocation).
// The per-sample coincidence counts of an event (channels whose dilated flag is up) live in the REGISTERS of the thread that owns the
// sample's run (round 6): a thread's run is the same for every channel of the event (L and the block size fix it), at most 17
// samples -- one byte each in five words, handed to the pass and back by value.  They used to sit in a row of HBM scratch: zeroed
// per event, read-modify-written sample by sample through flat loads (a dependent round trip each) by every channel that raised a
// flag, read back by stride for the majority logic, with two barriers that order global memory per event.
#define CONV_CNT_WORDS 5
struct CoincOut { double vmax; int any_flag; unsigned cw[CONV_CNT_WORDS]; };   // (all in registers: a reference parameter would come back through the stack)
template <int RUN, bool EXTRA>
__device__ __noinline__ CoincOut conv_coinc_pass(int L, double vscale, double threshold, int ch_on, TriggerDev trg, double* __restrict__ tr,
                                                 const double* __restrict__ add, int with_signal, unsigned cw0, unsigned cw1, unsigned cw2,
                                                 unsigned cw3, unsigned cw4, int* scan)
{
    static_assert(RUN + 1 <= 4 * CONV_CNT_WORDS, "one byte per sample of a thread's run");
    unsigned cw[CONV_CNT_WORDS] = {cw0, cw1, cw2, cw3, cw4};
    // Written for SIZE: the channel loop of channel_conv_kernel with its transforms is about as large as the instruction cache, and
    // this pass -- once 2200 instructions of unrolled branches around the run's samples, executed once per channel -- took 1.5 x
    // the forward transform's time on the 2-of-5 high / low arrays; a version with all reads in flight but MORE code was slower
    // still.  Hence rolled loops of four samples, selects instead of branches, the trace dump / noise rows (EXTRA) in their own
    // instantiation, and the per-sample counts in a second walk that only channels with a raised flag take.
    extern __shared__ __align__(16) unsigned char smem[];
    const double* S = (const double*)smem;
    const int NT = blockDim.x;
    const int nb = (trg.type == 0) ? L : L - 1;
    // runs of an ODD number of samples: the threads of a wave read the buffer at a stride of `chunk` doubles, and an even stride
    // (16 for L = 4096 on 256 threads) puts all 64 lanes on two LDS bank positions
    const int chunk = ((L + NT - 1) / NT) | 1, b0 = threadIdx.x * chunk, b1 = min(b0 + chunk, L);
    const double vs = with_signal ? vscale : 0.;
    const bool hl = trg.type != 0;
    auto folded = [&](int n) {   // y[n] + y[n + L]
        const int n2 = n + L;
        return S[2 * conv_pad(n >> 1) + (n & 1)] + S[2 * conv_pad(n2 >> 1) + (n2 & 1)];
    };
    double vmax = 0.;
    // one walk over the run: the index of the last raised flag at or before every sample goes to `sink`
    auto walk = [&](bool first, auto&& sink) {   // sink(sample, index of the last raised flag, word and byte of the sample in the run)
        int last_hi = -(1 << 30), last_lo = -(1 << 30), run = -1;
        if (hl && b0 < b1) {   // the w_hl - 1 samples in front of the run (the reference pads with zeros in front)
#pragma unroll 1
            for (int k = b0 - (trg.w_hl - 1); k < b0; k++) {
                double x = (k >= 0) ? folded(k) * vs : 0.;
                if (EXTRA && add && k >= 0) x += add[k];
                last_hi = (x >= trg.high) ? k : last_hi;
                last_lo = (x <= trg.low) ? k : last_lo;
            }
        }
        int word = 0;
#pragma unroll 1
        for (int i0 = b0; i0 < b1; i0 += 4, word++) {
            double x[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + q;
                x[q] = folded(i < b1 ? i : 0) * vs;
            }
            if (EXTRA) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + q;
                    if (add && i < b1) x[q] += add[i];
                    if (tr && first && i < b1) tr[i] = x[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + q;
                const bool ok = i < b1;
                const double ax = fabs(x[q]);
                vmax = ok ? fmax(vmax, ax) : vmax;
                last_hi = (ok && x[q] >= trg.high) ? i : last_hi;
                last_lo = (ok && x[q] <= trg.low) ? i : last_lo;
                const bool flag = hl ? ((i - last_hi < trg.w_hl) && (i - last_lo < trg.w_hl)) : (ax >= threshold);
                run = (ok && i < nb && flag && ch_on) ? i : run;
                if (ok) sink(i, run, word, q);
            }
        }
        return run;
    };
    const int run = walk(true, [](int, int, int, int) {});
    // running maximum across the runs: a wave-level scan of the runs' last values (shuffles), the waves' totals through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = run;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl = max(incl, t);
    }
    if (lane == 63) scan[wv] = incl;
    int before = __shfl_up(incl, 1);
    if (lane == 0) before = -1;
    lds_barrier();
    int top = -1;   // has the channel raised a flag at all?  (the same for every thread: the waves' totals)
    for (int q = 0; q < NT / 64; q++) {
        const int t = scan[q];
        if (q < wv) before = max(before, t);
        top = max(top, t);
    }
    if (top >= 0) {   // the coincidence count: flag i stays up for w_coinc samples
        const int wc = min(trg.w_coinc, nb), i_end = min(b1, nb - 1);
        const double keep = vmax;
        (void)walk(false, [&](int i, int r, int word, int q) {
            const int a = max(r, before);
            const unsigned inc = (i < i_end && a >= 0 && i - a < wc) ? (1u << (8 * q)) : 0u;
#pragma unroll
            for (int v = 0; v < CONV_CNT_WORDS; v++) cw[v] += (v == word) ? inc : 0u;   // (selects: no indexed register array)
        });
        vmax = keep;
    }
    lds_barrier();
    CoincOut o;
    o.vmax = vmax;
    o.any_flag = top >= 0 ? 1 : 0;
#pragma unroll
    for (int v = 0; v < CONV_CNT_WORDS; v++) o.cw[v] = cw[v];
    return o;
}
template <int NT, class FV, class FS>
__device__ __forceinline__ void czt_inverse_blocks(double2* x, const double2* __restrict__ Bi, const double2* __restrict__ tw,
                                                   const double2* __restrict__ E, const double2* __restrict__ Ci, int L, int m, int M,
                                                   FV&& spectrum, FS&& sink);   // (defined with the chirp-z channel kernel below)
// LOG2CAP: log2 of the complex points the LDS buffer holds.  13 (FFT_MAX): any event of up to FFT_MAX samples, 133 KB, one block per
// CU.  12: events of up to FFT_MAX / 2 samples only (the N = 2048 workloads), 68 KB + 9 KB static: TWO blocks per CU, so one block's
// barriers and LDS round trips are covered by the other's arithmetic.
// WR: the rays' transforms wave-private (N / 2 = 1024 or 2048) -- a compile-time choice, so that the instantiation the surveys run does
// not carry the batched block-wide ray path's registers (and the other one not the wave-private path's).
// NZ: with the thermal-noise trace per channel (full-capacity instantiation only), likewise compile-time.
// WR is N / 1024 (2 or 4: the 512-point blocks of a ray's transform) or 0 for the batched block-wide ray path.
// MODE: 0 the plain OR of simple thresholds, 1 the same with the traces of a triggered event emitted, 2 coincidence logic.
template <int LOG2CAP, int WR, bool NZ, int MODE>
__global__ void __launch_bounds__(CONV_THREADS(LOG2CAP), 2)
channel_conv_kernel(const int* __restrict__ n_list, const ConvHdr* __restrict__ hdr, int claim, const int* __restrict__ need,
                    const int* __restrict__ item_event, RayWork w, EventIn evin, EventOut ev,
                    const int* __restrict__ ev_len_index, StationDev st, int ask_model, TriggerDev trg,
                    const double2* __restrict__ tw, const double2* __restrict__ w16, LengthTables tab, int log2nh,
                    ChannelOut out, int exact, int* __restrict__ coinc_cnt, double2* __restrict__ conv_acc,
                    unsigned long long* __restrict__ xform_count, int* __restrict__ queue, int l_min, NoiseDev nz,
                    double* __restrict__ noise_buf)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int M = 1 << LOG2CAP;
    constexpr int NT = CONV_THREADS(LOG2CAP);   // threads of the block
    constexpr int BM = (LOG2CAP == FFT_LOG2_MAX) ? 4 : 2;   // transforms per batch at most
    const double threshold = trg.threshold;
    constexpr bool coinc = MODE == 2;   // (= trg.coincidence(): the launcher's choice)
    double2* acc = conv_acc + (long)blockIdx.x * FFT_MAX;
    (void)coinc_cnt;   // (the per-sample coincidence counts live in registers since round 6: CONV_CNT_WORDS)
    __shared__ int s_scan[NT];
    __shared__ int s_first;
    const int N = st.N, nh = N / 2;
    // the convolution buffer in the block-padded layout of conv_fft.h: complex element i at PZ(i), real sample n at PS(n); the field
    // buffer and the amplitudes sit behind the L <= FFT_MAX samples (FFT_MAX / 2 complex elements, padded) until the big transform starts
    double2* z = (double2*)smem;
    double* S = (double*)smem;
#define PZ(i) conv_pad(i)
#define PS(n) (2 * conv_pad((n) >> 1) + ((n) & 1))
    double2* xs = z + (conv_pad(M / 2) + 8);
    const double2* cft = w16 + (FFT_MAX / 2 + 1);   // per-pass twiddle tables of conv_fft.h, behind w16
    // up to four transforms at a time: a group of NT / B threads per transform builds its spectrum (amplitudes on the fly,
    // bins k and N/2 - k together), ONE batched transform runs them all, the placements follow in ray order.  B is what fits the
    // 64 KB behind the event's samples: 4 transforms of <= 1024 points, 2 of 2048, 1 of 4096.
    RayShared* rs4 = (RayShared*)(smem + (size_t)conv_lds_elems(M) * 16);   // [BM], behind the buffer (dynamic LDS: the ray functions reach it by offset)
    __shared__ ConvJob s_jobs[64];
    __shared__ int s_njob, s_nadv;
    __shared__ double2 s_ramp4[BM][64 + FFT_MAX / 4 / 64 + 1];
    // B is what fits behind the event's samples: M / 2 complex elements (conv_lds_elems)
    const int log2B = (LOG2CAP == FFT_LOG2_MAX) ? ((nh <= 1024) ? 2 : (nh <= 2048 ? 1 : 0)) : ((nh <= 1024) ? 1 : 0), B = 1 << log2B;
    // wave-private ray transforms for N / 2 = 1024, 2048 (other lengths: the batched block-wide transform below)
    constexpr int NBKr = WR ? WR : 2;   // (2: a valid template argument for the code the batched variant never runs)
    const int TR = nh >> 3;
    constexpr bool wave_rays = WR != 0;   // (the launcher: WR = N / 1024 only for N / 2 = 1024, 2048 with N / 16 <= NT)
    const int Bw = wave_rays ? min(BM, NT / TR) : 0;
    __shared__ double red2[2][NT / 64];   // channel maximum: one word per wave, two phases' worth (see the flags phase)
    __shared__ int s_trig2[2];
    if (threadIdx.x == 0) { s_trig2[0] = 0; s_trig2[1] = 0; }
    int flag_par = 0;
    const int n_list_events = *n_list;
    // unit of work: one candidate event; its channels are evaluated in sequence and -- the trigger being an OR over
    // channels -- the remaining ones are skipped (maxV = NaN) once one has triggered, unless every trace is wanted
    __shared__ long long s_emit_off;
    // The channels an event still needs come strongest Cauchy-Schwarz bound first (ConvHdr::order): the trigger is an OR over the
    // channels, so the one most likely to fire ends the event soonest (any order gives the same mask).
    __shared__ unsigned short s_rct[CONV_STAGE_RAYS];   // (channel << 4) | antenna table of the event's rays
    const bool best_first = !exact && st.n_ch <= CONV_MAX_ORDER;   // the needed channels of an event in the order of their bounds
    // Events are handed out through a counter in HBM (zero at launch; their cost varies with the number of channels and rays):
    // `claim` consecutive list entries per atomic, their records (ConvHdr: built by conv_header_kernel, the best-first channel
    // order included) read by the first wave, one 16-byte word per lane.  Two sets of records: a wave may run ahead through the
    // barrier-free skip path into the next claim while the others still read this one's.
    __shared__ ConvHdr s_hdr[2][CONV_CLAIM_MAX];
    __shared__ int s_claim[2];
    int par = 1, kc = claim;   // (the first pass claims)
    CT_DECL;
    for (;;) {
      if (kc == claim) {
          par ^= 1;
          kc = 0;
          if (threadIdx.x < 64) {
              int base = 0;
              if (threadIdx.x == 0) base = atomicAdd(queue, claim);
              base = __shfl(base, 0);
              const int k = (int)threadIdx.x / 7, part = (int)threadIdx.x - 7 * k;
              if (k < claim && base + k < n_list_events) {
                  typedef int conv_i4v __attribute__((ext_vector_type(4)));
                  ((conv_i4v*)&s_hdr[par][k])[part] = ((const __attribute__((address_space(1))) conv_i4v*)(hdr + base + k))[part];
              }
              if (threadIdx.x == 0) s_claim[par] = base;
          }
      }
      lds_barrier();
      const int le = s_claim[par] + kc;
      const ConvHdr& H = s_hdr[par][kc];
      kc++;
      if (le >= n_list_events) break;
      // the two instantiations share one list: each takes the events of its length class (l_min < L <= M); the others cost this test
      if (!(H.L > l_min && H.L <= M)) continue;
      const int ev_item0 = H.c * st.n_ch;
      const int ev_e = H.e, ev_L = H.L;
      // Touch what the channel steps of this event will read from HBM -- the rays' emission constants, attenuation rows and per-ray
      // scalars, one word per 64 bytes -- so that ONE round trip brings all of it into the L2 while the block sets up; the steps'
      // own dependent loads (ray list, job constants) then take L2 latencies.  The sum is only there to keep the loads.
      int touch_acc = 0;
      {
          const int r0e = H.r0, nre = H.nre;
          const char* pa = (const char*)(w.ask + r0e);
          for (long i = (long)threadIdx.x * 64; i < (long)nre * (long)sizeof(AskaryanConst); i += (long)NT * 64) touch_acc += *(const int*)(pa + i);
          const char* pt = (const char*)(w.att + (long)r0e * st.n_fc);
          for (long i = (long)threadIdx.x * 64; i < (long)nre * st.n_fc * 8; i += (long)NT * 64) touch_acc += *(const int*)(pt + i);
          if (threadIdx.x < 9 * 8) {
              const int which = threadIdx.x >> 3, part = threadIdx.x & 7;
              const char* pb = which == 0 ? (const char*)(w.ch + r0e) : which == 1 ? (const char*)(w.tab + r0e) :
                               which == 2 ? (const char*)(w.vfac_t + r0e) : which == 3 ? (const char*)(w.vfac_p + r0e) :
                               which == 4 ? (const char*)(w.r_theta + r0e) : which == 5 ? (const char*)(w.r_phi + r0e) :
                               which == 6 ? (const char*)(w.pol_theta + r0e) : which == 7 ? (const char*)(w.pol_phi + r0e) :
                               (const char*)(w.t0 + r0e);
              const int el = which < 2 ? 4 : ((which == 4 || which == 5) ? 16 : 8);
              if (part * 64 < nre * el) touch_acc += *(const int*)(pb + part * 64);
          }
      }
      // what every channel step of the event needs: read once (the steps used to fetch these scalars, and walk the rays' channel and
      // antenna-table numbers in HBM, one dependent load after the other, per channel)
      const int evx_il = H.il, evx_r0 = H.r0, evx_r1 = H.r0 + H.nre;
      const double evx_t_min = H.t_min;
      const bool rays_staged = evx_r1 - evx_r0 <= CONV_STAGE_RAYS;
      if (rays_staged)
          for (int i = threadIdx.x; i < evx_r1 - evx_r0; i += blockDim.x) s_rct[i] = (unsigned short)((w.ch[evx_r0 + i] << 4) | w.tab[evx_r0 + i]);
      lds_barrier();
      unsigned cw[CONV_CNT_WORDS] = {0u, 0u, 0u, 0u, 0u};   // coincidence counts of this thread's run of samples (conv_coinc_pass)
      // has an earlier channel of this event triggered?  A per-thread copy, refreshed between two barriers after every
      // evaluated channel: the shared flag itself may already have been reset for the NEXT event by a wave that ran ahead
      // through the barrier-free skip path below
      bool ev_trig = false;
      const int n_steps = best_first ? H.n_order : st.n_ch;
      // n-fold coincidence, production mode: once the channels that raised a flag plus the channels still to come are fewer than n
      // the event cannot trigger any more -- the rest of its channels is not transformed (the mask is the same; a 2-fold
      // coincidence on two candidate channels ends after the first one that stays silent)
      const bool coinc_stop = coinc && !exact && out.trace == nullptr;
      const int m_need = H.n_order;
      int n_done = 0, n_flagged = 0;
      // emission of a triggered event's traces: once a channel has triggered, ALL channels of the event are evaluated (in channel
      // order, the pruned ones included) and written into the block reserved for the event
      const bool can_emit = MODE == 1 && out.emit != nullptr && !exact;
      bool emitting = false;
      int c_star = -1;
      long long e_off = -1;
      for (int step = 0;; step++) {
        int ch;
        if (!emitting) {
            if (step >= n_steps) break;
            ch = best_first ? (int)H.order[step] : step;
        } else {
            if (step >= st.n_ch) break;
            ch = step;
            if (ch == c_star) continue;   // its trace went out when it triggered
        }
        const int item = ev_item0 + ch;
        if (!emitting && !best_first && !need[item]) continue;   // (the ordered list holds the needed channels only)
        const bool ch_on = !st.trig_on || st.trig_on[ch];  // triggered_channels of the reference's trigger modules
        const int e = ev_e;
        if (!emitting && !exact && !coinc && ev_trig) {
            if (threadIdx.x == 0) out.maxV[item] = NAN;
            continue;
        }
        const int L = ev_L, il = evx_il;
        const double t_min = evx_t_min;
        const double res = 1. / st.fs;
        // the circular convolution over L samples is a linear one of 2 L: when that fits FFT_MAX real points the packed transform has
        // FFT_MAX / 2 complex points (one stage and half the LDS traffic less); the response spectrum on that grid is every other
        // bin of the table (the even bins of a zero-padded sequence's transform are the transform of the shorter padding)
        const bool half_size = 2 * L <= M;
        const int log2Mr = half_size ? LOG2CAP - 1 : LOG2CAP, Mr = 1 << log2Mr, gs = FFT_MAX / Mr;
        const double vscale = (double)gs;   // the table carries the 1 / FFT_MAX of the un-normalised transform pair
        int r0 = evx_r0, r1 = evx_r1;
        // antenna response tables among this channel's rays (one, except for LPDAs seeing rays in different lobes)
        int tabs = 0, n_used = 0;
        if (rays_staged) {
            for (int i = 0; i < r1 - r0; i++) {
                const int ct = s_rct[i];
                if ((ct >> 4) == ch) { tabs |= 1 << (ct & 15); n_used++; }
            }
        } else {
            for (int r = r0; r < r1; r++)
                if (w.ch[r] == ch) { tabs |= 1 << w.tab[r]; n_used++; }
        }
        const bool multi = (tabs & (tabs - 1)) != 0;
        bool first_tab = true;
        // Thermal noise (channelGenericNoiseAdder before the filter chain, simulation.py:594-606; noise.h): its trace irfft_L(noise x
        // filter response) is an arbitrary-length transform -- ONE inverse chirp-z in this buffer (the per-length tables the chirp-z
        // channel kernel uses), parked in a row of HBM scratch and added when the channel's samples are read out.  Only the
        // full-capacity instantiation can hold the 8192-point chirp convolution: with noise every event runs in it.
        bool noisy = false;
        double* const nbuf = noise_buf ? noise_buf + (long)blockIdx.x * FFT_MAX : nullptr;
        if (NZ && LOG2CAP == FFT_LOG2_MAX && nz.on && nbuf) noisy = nz.amplitude[ch] > 0.;
        if constexpr (NZ && LOG2CAP == FFT_LOG2_MAX) if (noisy) {
            const int grp = nz.ev_group ? nz.ev_group[e] : e;
            const long long gid = nz.group_id ? nz.group_id[grp] : nz.group_offset + grp;
            const int sub = nz.ev_sub ? nz.ev_sub[e] : 0;
            const double2* Bi = tab.B_inv + (long)il * FFT_MAX;
            const double2* E = tab.E + (long)il * NRHIP_E_STRIDE;
            const double2* Ci = tab.Ci + (long)il * FFT_MAX;
            const double2* Hf = tab.H + ((long)il * st.n_fsets + (st.ch_fset ? st.ch_fset[ch] : 0)) * NRHIP_SPEC_STRIDE;
            const double nscale = st.fs / 1.4142135623730951 / L * (1.0 / FFT_MAX);
            const int mL = L / 2;
            // the inverse chirp-z of czt_inverse_blocks (m + 1 bins -> blocks of P = M - m samples, the block's input pre-multiplied by
            // exp(+2 pi i k n0 / L)), its 8192-point convolution through the transform pair of conv_fft.h (the Bluestein spectrum in the
            // bit-reversed order its table has)
            const int Pn = FFT_MAX - mL;
            for (int n0 = 0; n0 < L; n0 += Pn) {
                lds_barrier();
                for (int k = threadIdx.x; k <= mL; k += NT) {
                    double2 v = cmul(noise_bin(nz, gid, sub, ch, k, L, st.fs), Hf[k]);
                    if (k == 0 || k == mL) v = make_double2(v.x, 0.);   // Hermitian folding of irfft
                    else v = cscale(v, 2.);
                    if (n0 != 0) {
                        const unsigned kn = ((unsigned)k * (unsigned)n0) % (unsigned)L;
                        v = cmul(v, cconj(E[2 * kn]));                         // exp(+2 pi i k n0 / L)
                    }
                    z[PZ(k)] = cmul(v, Ci[k]);                                 // chirp(k; L, +)
                }
                lds_barrier();
                if (mL + 1 > FFT_MAX / 2) conv_fwd<FFT_LOG2_MAX, NT, true>(tw, cft, mL + 1);
                else conv_fwd<FFT_LOG2_MAX, NT, false>(tw, cft, mL + 1);
                conv_mid_plain<FFT_LOG2_MAX, NT, true>(Bi);
                conv_inv<FFT_LOG2_MAX, NT>(tw, cft);
                const int np = min(Pn, L - n0);
                for (int n = threadIdx.x; n < np; n += NT) nbuf[n0 + n] = cmul(z[PZ(n)], Ci[n]).x * nscale;
            }
            __syncthreads();   // (the noise samples are read back by other threads: a barrier that covers global memory)
        }
        for (int tb = 0; tb < NRHIP_N_ANT_TAB; tb++) {
            if (!((tabs >> tb) & 1)) continue;
            const double2* G = tab.G + (((long)il * st.n_fsets + (st.ch_fset ? st.ch_fset[ch] : 0)) * NRHIP_N_ANT_TAB + tb) * NRHIP_G_STRIDE;
            lds_barrier();
            CT(0);
            for (int n = threadIdx.x; n < L; n += blockDim.x) S[PS(n)] = 0.;
            lds_barrier();
            CT(1);
            {
                const double df = 1.0 / (N * (1. / st.fs));
                // the transforms of this (channel, antenna table), listed by the first wave: lane i looks at ray r_chunk + i (at most two
                // transforms each), a wave scan numbers them; the list holds 64, the next round resumes at the first ray that did not fit
                for (int r_chunk = r0, r_adv = 64; r_chunk < r1; r_chunk += r_adv) {
                  if (threadIdx.x < 64) {
                      const int lane = threadIdx.x, r = r_chunk + lane;
                      ConvJob jb[2];
                      int nj_l = 0;
                      if (r < r1 && (rays_staged ? (int)s_rct[r - r0] == ((ch << 4) | tb) : (w.ch[r] == ch && w.tab[r] == tb))) {
                          const double vt = w.vfac_t[r], vp = w.vfac_p[r];
                          const double2 rt = w.r_theta[r], rp = w.r_phi[r];
                          const double pt = w.pol_theta[r], pp = w.pol_phi[r];
                          const double wt = fabs(vt * pt) * cabs2(rt), wp = fabs(vp * pp) * cabs2(rp);
                          // real reflection coefficients: both on-sky components are the same real pulse -> one transform
                          const bool one = (rt.y == 0. && rp.y == 0.);
                          // start bin and sub-sample remainder (efieldToVoltageConverter.py:214-218)
                          const double start_time = w.t0[r] - t_min + st.cable[ch] + 0;
                          const long start_bin = (long)rint(start_time / res);
                          ConvJob job;
                          job.r = r;
                          int sb32 = (int)start_bin % L;   // (|start_bin| is a few L at most: 32-bit remainder instead of two 64-bit ones)
                          job.sbin = sb32 < 0 ? sb32 + L : sb32;
                          job.rem = start_time - start_bin * res;
                          job.shift = !(fabs(rint(job.rem * st.fs) - job.rem * st.fs) < 1e-5);
                          for (int comp = 0; comp < (one ? 1 : 2); comp++) {
                              if (!one && (comp ? wp : wt) <= 1e-13 * (comp ? wt : wp)) continue;
                              job.pol = one ? 1. : (comp ? pp : pt);
                              job.rc = one ? make_double2(1., 0.) : (comp ? rp : rt);
                              job.vfac = one ? (vt * pt * rt.x + vp * pp * rp.x) : (comp ? vp : vt);
                              jb[nj_l++] = job;
                          }
                      }
                      int incl = nj_l;   // inclusive scan over the wave
                      for (int off = 1; off < 64; off <<= 1) {
                          const int v = __shfl_up(incl, off);
                          if (lane >= off) incl += v;
                      }
                      const int first = incl - nj_l;
                      const bool fits = incl <= 64;
                      const unsigned long long fb = __ballot(fits);   // lanes 0 .. m - 1 fit (the scan is monotone)
                      const int m = (fb == ~0ull) ? 64 : __ffsll((long long)~fb) - 1;
                      if (fits && nj_l > 0) s_jobs[first] = jb[0];
                      if (fits && nj_l > 1) s_jobs[first + 1] = jb[1];
                      if (lane == (m > 0 ? m - 1 : 0)) { s_njob = (m > 0) ? incl : 0; s_nadv = (m > 0) ? m : 1; }
                  }
                  lds_barrier();
                  CT(11);
                  const int n_jobs = s_njob;
                  r_adv = s_nadv;
                  if (wave_rays) {
                    // wave-private transforms (conv_fft.h): Bw jobs at a time, TR = nh / 8 threads each.  No staging barrier: every
                    // wave of a job fills the job's RayShared itself (the same values), builds its share of the spectrum with the
                    // first stages (ray_build), runs the 512-point block it owns, and the job's threads place the samples
                    for (int j0 = 0; j0 < n_jobs; j0 += Bw) {
                      const int nj = min(Bw, n_jobs - j0);
                      const int tid = conv_opaque((int)threadIdx.x);
                      const int g = tid / TR, lt = tid - g * TR, lane = tid & 63;
                      const bool mine = g < nj;
                      const int BS = ray_blk_stride(NBKr);
                      double2* xjob = xs + (long)g * NBKr * BS;
                      if (mine) {
                          const ConvJob job = s_jobs[j0 + g];
                          RayShared& rg = rs4[g];
                          static_assert(sizeof(AskaryanConst) % 8 == 0, "copied word by word");
                          if (lane < (int)(sizeof(AskaryanConst) / 8))   // one word per lane (a struct copy by one lane is 19 loads in a row)
                              ((double*)&rg.ask)[lane] = ((const double*)&w.ask[job.r])[lane];
                          for (int i = lane; i < st.n_fc; i += 64) {
                              const double a0 = w.att[(long)job.r * st.n_fc + i], x0 = st.fcoarse[i];
                              rg.att[i] = a0;
                              rg.xp[i] = x0;
                              if (i < st.n_fc - 1) rg.slope[i] = (w.att[(long)job.r * st.n_fc + i + 1] - a0) / (st.fcoarse[i + 1] - x0);
                          }
                          wave_lds_sync();
                          CT(2);
                          const RayStation rst = {st.N, st.n_fc, st.fs, st.fpow, st.seg, st.lnf};
                          const int xo = (int)((const unsigned char*)xjob - smem), ro = (int)((const unsigned char*)&rg - smem);
                          const double2* two = conv_opaque(tw);
                          if (NBKr == 4) ray_build<4>(xo, ro, lt, job, rst, ask_model, two, log2nh);
                          else ray_build<2>(xo, ro, lt, job, rst, ask_model, two, log2nh);
                      }
                      lds_barrier();
                      CT(3);
                      if (mine) ray_p23((int)((const unsigned char*)(xs + (long)(tid >> 6) * BS) - smem), conv_opaque(cft), lane);
                      lds_barrier();
                      for (int q = 0; q < nj; q++) {
                          if (g == q) {
                              const ConvJob jq = s_jobs[j0 + q];
                              const int xo = (int)((const unsigned char*)xjob - smem);
                              if (NBKr == 4) ray_place<4>(xo, lt, nh, jq, L);
                              else ray_place<2>(xo, lt, nh, jq, L);
                          }
                          lds_barrier();
                      }
                      CT(4);
                    }
                  } else
                  for (int j0 = 0; j0 < n_jobs; j0 += B) {
                    const int nj = min(B, n_jobs - j0);
                    const ConvJob* s_job = s_jobs + j0;
                    // thread groups as large as the batch allows: 512 / 256 / 128 threads per transform for 1 / 2 / 3-4 of them
                    const int GTe = (nj <= 1) ? NT : (nj == 2 ? NT / 2 : NT / 4);
                    const int g = threadIdx.x / GTe, lt = threadIdx.x - g * GTe;
                    const bool mine = g < nj;
                    const ConvJob job = s_job[mine ? g : 0];
                    RayShared& rg = rs4[mine ? g : 0];
                    if (mine) {
                        if (lt == 0) rg.ask = w.ask[job.r];
                        for (int i = lt; i < st.n_fc; i += GTe) {   // attenuation factors and the slopes between them (fill_amplitude's)
                            const double a0 = w.att[(long)job.r * st.n_fc + i], x0 = st.fcoarse[i];
                            rg.att[i] = a0;
                            rg.xp[i] = x0;
                            if (i < st.n_fc - 1) rg.slope[i] = (w.att[(long)job.r * st.n_fc + i + 1] - a0) / (st.fcoarse[i + 1] - x0);
                        }
                        // the sub-sample shift's phase ramp exp(-2 pi i f rem), f = k fs / N: w^k = w^(k & 63) * (w^64)^(k >> 6)
                        if (job.shift)
                            for (int t = lt; t < 64 + (nh >> 6) + 1; t += GTe) {
                                const double f = (t < 64 ? t : 64 * (t - 64)) * (1.0 / (N * (1. / st.fs)));
                                double sn, cs;
                                sincospi(-2. * job.rem * f, &sn, &cs);
                                s_ramp4[g][t] = make_double2(cs, sn);
                            }
                    }
                    lds_barrier();
                    CT(2);
                    if (mine) {
                        // spectrum of the packed half-length transform (field_time_domain), bins k and N/2 - k by the same thread:
                        // both need both amplitudes.  Station tables of the next iteration are requested ahead.
                        double2* xg = xs + (long)g * nh;
                        const int stride = nh + 1, off_l = rg.ask.had ? 0 : stride;
                        const bool m0 = rg.ask.model == 0;
                        const double2* ramp = job.shift ? s_ramp4[g] : nullptr;
                        const double roll = floor(2.0 * st.fs);
                        int k = lt;
                        double a_pl = 0., a_pr = 0., b_pl = 0., b_pr = 0.;
                        int a_sg = 0, b_sg = 0;
                        double2 a_w = make_double2(1., 0.), b_w = a_w;
                        auto fetch = [&](int kk) {
                            const int k2 = nh - kk;
                            if (kk > 0 && kk < nh) { if (m0) { a_pl = st.fpow[off_l + kk]; a_pr = st.fpow[2 * stride + kk]; } a_sg = st.seg[kk]; }
                            if (k2 > 0 && k2 < nh) { if (m0) { b_pl = st.fpow[off_l + k2]; b_pr = st.fpow[2 * stride + k2]; } b_sg = st.seg[k2]; }
                            a_w = tw[kk * (FFT_MAX / N)];
                            if (k2 < nh) b_w = tw[k2 * (FFT_MAX / N)];
                        };
                        if (k <= nh / 2) fetch(k);
                        for (; k <= nh / 2; k += GTe) {
                            const double pl1 = a_pl, pr1 = a_pr, pl2 = b_pl, pr2 = b_pr;
                            const int sg1 = a_sg, sg2 = b_sg;
                            const double2 w1 = a_w, w2 = b_w;
                            if (k + GTe <= nh / 2) fetch(k + GTe);
                            const int k2 = nh - k;
                            const double amp1 = conv_amplitude(k, nh, df, st, rg, pl1, pr1, sg1);
                            const double amp2 = conv_amplitude(k2, nh, df, st, rg, pl2, pr2, sg2);
                            const double2 F1 = field_bin(k, amp1, N, st.fs, job.pol, job.rc, job.rem, job.shift, ask_model, roll, ramp);
                            const double2 F2 = field_bin(k2, amp2, N, st.fs, job.pol, job.rc, job.rem, job.shift, ask_model, roll, ramp);
                            {
                                const double2 Gc = cconj(F2);
                                const double2 ge = cscale(cadd(F1, Gc), 0.5), d = cscale(csub(F1, Gc), 0.5);
                                const double2 go = cmul(d, cconj(w1));         // * exp(+2 pi i k / N)
                                xg[k] = make_double2(ge.x - go.y, ge.y + go.x);  // ge + i go
                            }
                            if (k != 0 && k2 != k) {
                                const double2 Gc = cconj(F1);
                                const double2 ge = cscale(cadd(F2, Gc), 0.5), d = cscale(csub(F2, Gc), 0.5);
                                const double2 go = cmul(d, cconj(w2));
                                xg[k2] = make_double2(ge.x - go.y, ge.y + go.x);
                            }
                        }
                    }
                    lds_barrier();
                    // inverse transforms of the whole batch, natural -> bit-reversed inside each block; scale applied by the reader
                    fft_dif_batched<3>(xs, log2nh + log2B, log2B, tw, true);
                    CT(3);
                    for (int q = 0; q < nj; q++) {
                        const ConvJob jq = s_job[q];
                        const double2* xq = xs + (long)q * nh;
                        const double c = jq.vfac / nh;  // the fs/sqrt(2) of freq2time cancels against time2freq's sqrt(2)/fs
                        for (int j = threadIdx.x; j < nh; j += blockDim.x) {
                            const double2 y = xq[bitrev(j, log2nh)];
                            int i0 = jq.sbin + 2 * j;
                            if (i0 >= L) i0 -= L;
                            int i1 = i0 + 1;
                            if (i1 >= L) i1 -= L;
                            S[PS(i0)] += y.x * c;
                            S[PS(i1)] += y.y * c;
                        }
                        lds_barrier();
                    }
                    CT(4);
                  }
                }
            }
            // (only the L / 2 complex elements that hold the L real samples are read by the forward transform: what lies behind them in the
            // buffer -- the previous transform's output -- counts as zero there; every placement above ended with a barrier)
            CT(5);
            // forward transform, real-transform split * G * merge, first stages of the inverse (conv_fft.h)
            // (forward transform and spectrum pass as ONE out-of-line function, the first response values requested a pass earlier,
            // was measured in round 6: 10.59 against 10.61 ms -- the two calls stay)
            if (half_size) { conv_fwd<LOG2CAP - 1, NT>(tw, cft, L >> 1); CT(6); conv_mid<LOG2CAP - 1, NT>(G, w16); }
            else { conv_fwd<LOG2CAP, NT>(tw, cft, L >> 1); CT(6); conv_mid<LOG2CAP, NT>(G, w16); }
            if (multi) {  // sum the tables' contributions (the rest of the inverse is linear) in global scratch of this block
                for (int k = threadIdx.x; k < Mr; k += blockDim.x) acc[k] = first_tab ? z[PZ(k)] : cadd(acc[k], z[PZ(k)]);
                first_tab = false;
            }
        }
        double vmax = 0.;
        int trig = 0;
        const bool sig = n_used > 0;   // (a channel without rays still has its noise)
        if (sig || noisy) {
            if (sig) {
                if (threadIdx.x == 0 && xform_count) {  // work actually done (roofline accounting of bench.py)
                    atomicAdd(&xform_count[0], 1ULL);
                    atomicAdd(&xform_count[1], (unsigned long long)n_used);
                }
                if (multi) {
                    lds_barrier();
                    for (int k = threadIdx.x; k < Mr; k += blockDim.x) z[PZ(k)] = acc[k];
                }
                lds_barrier();
                CT(7);
                if (half_size) conv_inv<LOG2CAP - 1, NT>(tw, cft);
                else conv_inv<LOG2CAP, NT>(tw, cft);
                CT(8);
            }
            if (!coinc) {
                const ConvOut co = conv_output_pass(L, vscale, threshold, ch_on ? 1 : 0, emitting ? out.emit + e_off + (long long)ch * L : nullptr,
                                                    out.trace ? out.trace + out.trace_offset[item] : nullptr, noisy ? nbuf : nullptr, sig ? 1 : 0);
                vmax = co.vmax;
                trig = co.trig;
            } else {
                CoincOut co;
                if (out.trace || noisy)
                    co = conv_coinc_pass<M / NT, true>(L, vscale, threshold, ch_on ? 1 : 0, trg, out.trace ? out.trace + out.trace_offset[item] : nullptr,
                                                       noisy ? nbuf : nullptr, sig ? 1 : 0, cw[0], cw[1], cw[2], cw[3], cw[4], s_scan);
                else
                    co = conv_coinc_pass<M / NT, false>(L, vscale, threshold, ch_on ? 1 : 0, trg, nullptr, nullptr, sig ? 1 : 0, cw[0], cw[1], cw[2],
                                                        cw[3], cw[4], s_scan);
                vmax = co.vmax;
                n_flagged += co.any_flag;
#pragma unroll
                for (int v = 0; v < CONV_CNT_WORDS; v++) cw[v] = co.cw[v];
            }
        }
        else if (emitting) {   // a channel without rays: zeros, as the reference's empty channels
            double* const em = out.emit + e_off + (long long)ch * L;
            for (int n = threadIdx.x; n < L; n += blockDim.x) em[n] = 0.;
        }
        n_done++;
        CT(12);
        // maximum and trigger flag of the channel with ONE barrier: wave-level reduction, a word per wave and a flag in the buffer of
        // this phase's parity (the other buffer is cleared for the next phase), every thread reads the result for itself
        {
            const int fp = flag_par;
            flag_par ^= 1;
            for (int off = 32; off > 0; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off));
            const bool wave_trig = __ballot(trig != 0) != 0ull;
            if ((threadIdx.x & 63) == 0) {
                red2[fp][threadIdx.x >> 6] = vmax;
                if (wave_trig) s_trig2[fp] = 1;
            }
            lds_barrier();
            double vm = red2[fp][0];
            for (int i = 1; i < NT / 64; i++) vm = fmax(vm, red2[fp][i]);
            const bool ch_trig = s_trig2[fp] != 0;
            if (threadIdx.x == 0) {
                s_trig2[fp ^ 1] = 0;
                out.maxV[item] = vm;
                if (ch_trig) out.triggered[e] = 1;
            }
            ev_trig = ev_trig || ch_trig;
        }
        CT(9);
        if (can_emit && ev_trig && !emitting) {
            // the event has just triggered on channel ch: reserve n_ch x L samples, write this channel's trace (still in S), then
            // go through all the other channels
            if (threadIdx.x == 0) {
                const unsigned long long want = (unsigned long long)st.n_ch * (unsigned long long)L;
                const unsigned long long o = atomicAdd(&out.emit_cursor[0], want);
                long long off = (long long)o;
                if (o + want > (unsigned long long)out.emit_cap) {
                    off = -2;   // full: the host sees the count and falls back to the second pass for these events
                    atomicAdd(&out.emit_cursor[1], 1ULL);
                } else {
                    atomicAdd(&out.emit_cursor[2], 1ULL);
                }
                out.emit_offset[e] = off;
                s_emit_off = off;
            }
            lds_barrier();
            e_off = s_emit_off;
            if (e_off >= 0) {
                double* const em = out.emit + e_off + (long long)ch * L;
                (void)conv_output_pass(L, vscale, threshold, 0, em, nullptr, nullptr, 1);
                emitting = true;
                c_star = ch;
                step = -1;   // restart: every channel in channel order
            }
            lds_barrier();
        }
        if (coinc_stop && n_flagged + (m_need - n_done) < trg.n_coinc) {   // (block-uniform: counts of whole channels)
            if (threadIdx.x == 0) {   // the rest is not evaluated
                if (best_first) { for (int s2 = step + 1; s2 < n_steps; s2++) out.maxV[ev_item0 + H.order[s2]] = NAN; }
                else { for (int c2 = ch + 1; c2 < st.n_ch; c2++) if (need[ev_item0 + c2]) out.maxV[ev_item0 + c2] = NAN; }
            }
            break;
        }
      }
      asm volatile("" :: "v"(touch_acc));
      if (coinc) {  // majority logic over the channels of the event: every thread looks at the counts of its own run of samples
          if (threadIdx.x == 0) s_first = 0x7fffffff;
          lds_barrier();
          const int nb = (trg.type == 0) ? ev_L : ev_L - 1;
          const int chunk = ((ev_L + (int)blockDim.x - 1) / (int)blockDim.x) | 1, b0 = (int)threadIdx.x * chunk;   // (conv_coinc_pass's runs)
          int first = 0x7fffffff;
#pragma unroll
          for (int v = CONV_CNT_WORDS - 1; v >= 0; v--)
#pragma unroll
              for (int q = 3; q >= 0; q--) {
                  const int i = b0 + 4 * v + q;
                  if (4 * v + q < chunk && i < nb - 1 && (int)((cw[v] >> (8 * q)) & 0xffu) >= trg.n_coinc) first = i;   // (descending: the smallest stays)
              }
          if (first != 0x7fffffff) atomicMin(&s_first, first);
          lds_barrier();
          if (threadIdx.x == 0 && s_first != 0x7fffffff) {
              out.triggered[ev_e] = 1;
              out.trigger_bin[ev_e] = s_first;
          }
          lds_barrier();
      }
    }
}
#undef PZ
#undef PS
#pragma clang fp contract(fast)

// y_j = e[2j] + i e[2j+1] of a real N-sample trace in HBM, optionally delayed by the sub-sample remainder `rem` through the
// Fourier shift theorem on the N grid (rfft -> * exp(-2 pi i f rem) -> irfft, base_trace.py:273-276).  Without the shift the
// packed trace is left in natural order, with it in bit-reversed order and unscaled by 1 / (N / 2).  x: FFT_MAX complex (LDS).
__device__ inline void trace_to_packed(double2* x, const double* __restrict__ tr, int N, const NPlan& np, double fs, double rem,
                                       bool shift, const double2* __restrict__ tw, double2* __restrict__ gbuf = nullptr)
{
    const int M = FFT_MAX, nh = N / 2;
    const double res = 1. / fs;
    for (int j = threadIdx.x; j < nh; j += blockDim.x) x[j] = make_double2(tr[2 * j], tr[2 * j + 1]);
    __syncthreads();
    if (!shift) return;
    nplan_fft(x, np, tw, false);     // Y in bit-reversed (power of two) or natural order: nplan_idx
    // G'(k), k = 0..nh: in the upper half of the buffer, or -- N / 2 + 1 values no longer fit there (N >= 8192) -- in the caller's
    // row of HBM scratch
    double2* G = (nh + 1 <= M / 2) ? x + M / 2 : gbuf;
    for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
        int ka = (k == nh) ? 0 : k, kb = (k == 0 || k == nh) ? 0 : nh - k;
        double2 Y1 = x[nplan_idx(np, ka)], Y2 = cconj(x[nplan_idx(np, kb)]);
        double2 ge = cscale(cadd(Y1, Y2), 0.5);
        double2 d = cscale(csub(Y1, Y2), 0.5);
        double2 go = make_double2(d.y, -d.x);  // d / i
        double2 wk = nplan_w(np, k, tw);       // exp(-2 pi i k / N)
        double2 g = cadd(ge, cmul(go, wk));
        double f = k * (1.0 / (N * res));
        double sn, cs;
        sincos(-2. * M_PI * rem * f, &sn, &cs);
        g = cmul(g, make_double2(cs, sn));
        if (k == 0 || k == nh) g.y = 0.;  // irfft uses the real part of DC and Nyquist only
        G[k] = g;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nh; k += blockDim.x) {
        double2 Gk = G[k], Gc = cconj(G[nh - k]);
        double2 ge = cscale(cadd(Gk, Gc), 0.5);
        double2 d = cscale(csub(Gk, Gc), 0.5);
        double2 go = cmul(d, cconj(nplan_w(np, k, tw)));
        x[k] = make_double2(ge.x - go.y, ge.y + go.x);
    }
    __syncthreads();
    nplan_fft(x, np, tw, true);
}

// inverse chirp-z of channel_kernel: out[n] = sum_{k <= m} V(k) exp(+2 pi i k n / L), n < L, handed to sink(n, value) for every n.
// m + 1 input bins and P outputs share the M-point convolution (m + P <= M): blocks of P = M - m outputs, the input of block n0
// pre-multiplied by exp(+2 pi i k n0 / L).  When that would leave fewer than 1024 outputs per block (L > 14 336) the input is cut
// into chunks of K1 = M / 2 bins as well: out[n0 + n] = sum_c exp(+2 pi i c n / L) sum_{k'} (V(c + k') exp(+2 pi i (c + k') n0 / L))
// exp(+2 pi i k' n / L) -- every inner sum the same convolution (table built for K1 inputs and P = M / 2 outputs), the chunk sums
// added in registers.  Below that size the operations, and hence the bits, are those of the single-chunk code.

template <int NT, class FV, class FS>
__device__ __forceinline__ void czt_inverse_blocks(double2* x, const double2* __restrict__ Bi, const double2* __restrict__ tw,
                                                   const double2* __restrict__ E, const double2* __restrict__ Ci, int L, int m, int M,
                                                   FV&& spectrum, FS&& sink)
{
    const unsigned LL = (unsigned)L;
    if (!czt_inverse_chunked(m, M)) {
        const int P = M - m;
        for (int n0 = 0; n0 < L; n0 += P) {
            __syncthreads();
#pragma unroll 2
            for (int k = threadIdx.x; k < M; k += NT) {
                double2 v = make_double2(0., 0.);
                if (k <= m) {
                    v = spectrum(k);
                    if (n0 != 0) {
                        unsigned kn = ((unsigned)k * (unsigned)n0) % LL;
                        v = cmul(v, cconj(E[2 * kn]));                         // exp(+2 pi i k n0 / L)
                    }
                    v = cmul(v, Ci[k]);                                        // chirp(k; L, +)
                }
                x[k] = v;
            }
            __syncthreads();
            czt_convolve_t<NT>(x, Bi, tw);
            const int np = min(P, L - n0);
#pragma unroll 4
            for (int n = threadIdx.x; n < np; n += NT) sink(n0 + n, cmul(x[n], Ci[n]));
        }
        __syncthreads();
        return;
    }
    const int K1 = M / 2, P = M / 2;
    constexpr int PER = (FFT_MAX / 2 + NT - 1) / NT;   // outputs of a block per thread
    for (int n0 = 0; n0 < L; n0 += P) {
        const int np = min(P, L - n0);
        double2 part[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) part[q] = make_double2(0., 0.);
        for (int c0 = 0; c0 <= m; c0 += K1) {
            __syncthreads();
#pragma unroll 2
            for (int k1 = threadIdx.x; k1 < M; k1 += NT) {
                double2 v = make_double2(0., 0.);
                const int k = c0 + k1;
                if (k1 < K1 && k <= m) {
                    v = spectrum(k);
                    if (n0 != 0) {
                        unsigned kn = ((unsigned)k * (unsigned)n0) % LL;
                        v = cmul(v, cconj(E[2 * kn]));                         // exp(+2 pi i k n0 / L)
                    }
                    v = cmul(v, Ci[k1]);                                       // chirp(k'; L, +)
                }
                x[k1] = v;
            }
            __syncthreads();
            czt_convolve_t<NT>(x, Bi, tw);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int n = threadIdx.x + q * NT;
                if (n < np) {
                    double2 u = cmul(x[n], Ci[n]);
                    if (c0 != 0) {
                        unsigned cn = ((unsigned)c0 * (unsigned)n) % LL;
                        u = cmul(u, cconj(E[2 * cn]));                         // exp(+2 pi i c0 n / L)
                    }
                    part[q] = cadd(part[q], u);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int n = threadIdx.x + q * NT;
            if (n < np) sink(n0 + n, part[q]);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// kernel: one (candidate event, channel) item per block iteration.
//   for every ray of the channel and both on-sky components:
//       N-point field in time domain (with sub-sample shift) -> chirp-z onto the event's L grid
//       -> * VEL -> accumulate channel spectrum (global scratch of this block, L2 resident)
//   * filters -> chirp-z back to L samples -> |V| >= threshold (last sample excluded, see majority logic)
// LDS: FFT_MAX complex (128 KB) + (N/2 + 1) doubles.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
channel_kernel(int n_items, const int* __restrict__ item_event, RayWork w, EventIn evin, EventOut ev,
               const int* __restrict__ ev_len_index, StationDev st, FilterSet fl, int ask_model, double threshold,
               const double2* __restrict__ tw, LengthTables tab, double2* __restrict__ scratch, int log2nh,
               ChannelOut out, int exact, int skip_upto, double2* __restrict__ tab_nodes,
               const double* __restrict__ ray_traces, FilterSet envf, double* __restrict__ env_trace, NoiseDev nz,
               double* __restrict__ amp_scratch, const int* __restrict__ item_need)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int M = FFT_MAX, N = st.N, nh = N / 2;
    double2* x = (double2*)smem;
    // the ray's amplitude table: behind the transform buffer in LDS, or (N > 4096: 128 + 32 KB do not fit) a row of HBM scratch
    double* amp = amp_scratch ? amp_scratch + (long)blockIdx.x * (nh + 1) : (double*)(x + M);
    __shared__ RayShared rs;
    __shared__ double red[1024];
    __shared__ int s_trig;
    double2* __restrict__ acc = scratch + (long)blockIdx.x * 2 * NRHIP_SPEC_STRIDE;
    double2* __restrict__ zbuf = acc + NRHIP_SPEC_STRIDE;   // the forward transform's outputs when they come in blocks
    const long vel_stride = NRHIP_SPEC_STRIDE;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int e = item_event[item / st.n_ch], ch = item % st.n_ch;
        const int L = ev.L[e], m = L / 2, il = ev_len_index[e];
        const bool tabulated = (st.ant_model[ch] == 3);
        if (L <= skip_upto && !tabulated && !ray_traces) continue;  // done by channel_conv_kernel
        if (item_need && !item_need[item]) continue;   // general path: the channel cannot reach the threshold (general_prefilter_kernel)
        double2* nodes = tab_nodes ? tab_nodes + (long)blockIdx.x * 2 * st.max_tab_freq : nullptr;
        const double t_min = ev.t_min[e];
        const double res = 1. / st.fs;
        const double2* Bf = tab.B_fwd + (long)il * M;
        const double2* Bi = tab.B_inv + (long)il * M;
        const double2* E = tab.E + (long)il * NRHIP_E_STRIDE;   // E[j] = exp(-2 pi i j / (2 L))
        const int fset = st.ch_fset ? st.ch_fset[ch] : 0;
        const double2* Hf = tab.H + ((long)il * st.n_fsets + fset) * NRHIP_SPEC_STRIDE;
        const double2* Cf = tab.Cf + (long)il * NRHIP_SPEC_STRIDE;
        const double2* Ci = tab.Ci + (long)il * FFT_MAX;
        const unsigned LL = (unsigned)L;
        int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
        if (!exact && !tabulated) {
            // Cauchy-Schwarz: the channel trace is sum_r vfac_r (e_r (*) h_L), so |V(t)| <= ||h_L||_2 sum_r |vfac_r| ||e_r||_2.
            // If even that cannot reach the threshold nothing of this item needs to be transformed.
            double bnd = 0.;
            for (int r = r0; r < r1; r++) {
                if (w.ch[r] != ch) continue;
                bnd += w.e_norm[r] * (fabs(w.vfac_t[r] * w.pol_theta[r]) * cabs2(w.r_theta[r]) +
                                      fabs(w.vfac_p[r] * w.pol_phi[r]) * cabs2(w.r_phi[r])) *
                       tab.hnorm[((long)il * st.n_fsets + fset) * NRHIP_N_ANT_TAB + w.tab[r]];
            }
            if (!(bnd * (1 + 1e-9) >= threshold)) {
                if (threadIdx.x == 0) out.maxV[item] = -bnd;
                continue;
            }
        }
        for (int k = threadIdx.x; k <= m; k += blockDim.x) acc[k] = make_double2(0., 0.);
        if (threadIdx.x == 0) s_trig = 0;
        __syncthreads();
        int n_used = 0;
        for (int r = r0; r < r1; r++) {
            if (w.ch[r] != ch) continue;
            n_used++;
            if (!ray_traces) {
                if (threadIdx.x == 0) rs.ask = w.ask[r];
                for (int i = threadIdx.x; i < st.n_fc; i += blockDim.x) rs.att[i] = w.att[(long)r * st.n_fc + i];
                __syncthreads();
                fill_amplitude(amp, st, rs);
            }
            // start bin and sub-sample remainder (efieldToVoltageConverter.py:214-218)
            double start_time = w.t0[r] - t_min + st.cable[ch] + 0;
            long start_bin = (long)rint(start_time / res);
            const unsigned sbin = (unsigned)(((start_bin % (long)L) + (long)L) % (long)L);
            double rem = start_time - start_bin * res;
            bool shift = !(fabs(rint(rem * st.fs) - rem * st.fs) < 1e-5);
            const double2* vel = tab.vel + ((long)il * NRHIP_N_ANT_TAB + w.tab[r]) * vel_stride;
            double Tt = w.vfac_t[r], Tp = w.vfac_p[r];
            const double dir = 1.;
            const double* TT = w.vel_T + 4 * (long)r;
            const AntTabDev* at = tabulated ? &st.ant_tabs[st.ant_tab_index[ch]] : nullptr;
            bool tab_ok = false;
            if (tabulated) {
                // angular interpolation once per frequency node of the table; the frequency interpolation follows per bin
                const TabAngles ta = tab_angles(*at, w.theta_ant[r], w.phi_ant[r]);
                tab_ok = ta.ok;
                if (ta.ok)
                    for (int iF = threadIdx.x; iF < at->nF; iF += blockDim.x) tab_node(*at, ta, iF, &nodes[iF], &nodes[at->nF + iF]);
                Tt = Tp = 1.;  // both on-sky components are transformed
                __syncthreads();
            }
            // weight of each on-sky component in the channel voltage; a component below 1e-13 of the other one
            // (e.g. the e_phi response of a vertical dipole, 1e-17 from the rotation round-off) is not transformed
            const double wt = fabs(Tt * w.pol_theta[r]) * cabs2(w.r_theta[r]);
            const double wp = fabs(Tp * w.pol_phi[r]) * cabs2(w.r_phi[r]);
            for (int comp = 0; comp < 2; comp++) {
                double pol = comp ? w.pol_phi[r] : w.pol_theta[r];
                double2 rc = comp ? w.r_phi[r] : w.r_theta[r];
                double vfac = (comp ? Tp : Tt) * dir;
                if (!ray_traces && (comp ? wp : wt) <= 1e-13 * (comp ? wt : wp)) continue;
                bool natural = false;
                double sc = 1.0 / nh;  // the fs/sqrt(2) of freq2time cancels against time2freq's sqrt(2)/fs
                if (ray_traces) {
                    // the ray's electric-field trace of this component comes from HBM (time-domain emission models,
                    // birefringence: general_trace_kernel); the on-sky factors are already in it
                    trace_to_packed(x, ray_traces + ((long)r * 2 + comp) * N, N, st.np, st.fs, rem, shift, tw, zbuf);
                    natural = !shift;
                    sc = (shift ? 1.0 / nh : 1.0) * (1.4142135623730951 / st.fs);  // time2freq
                } else {
                    field_time_domain(x, amp, N, st.np, st.fs, pol, rc, rem, shift, ask_model, floor(2.0 * st.fs), tw);
                }
                // gather y_j (bit-reversed positions) -> registers, then lay out a_j = y_j * chirp_j, zero pad
                double2 yreg[FFT_MAX / 512];   // N / 2 <= 7168 points on 512 threads
                int cnt = 0;
                for (int j = threadIdx.x; j < nh; j += blockDim.x) yreg[cnt++] = natural ? x[j] : x[nplan_idx(st.np, j)];
                __syncthreads();
                // the m output bins Z(0 .. m - 1) fit one convolution if nh + m - 1 <= M; longer common traces take them in blocks of
                // Pf: Z(k0 + k') = sum_j (y_j exp(-2 pi i j k0 / m)) exp(-2 pi i j k' / m) -- the same chirp table on the
                // phase-ramped input -- collected in zbuf (HBM scratch of the block) for the untangling below, which pairs k and m - k
                const int Pf = min(m, M - nh + 1);
                const bool blocked = Pf < m;
                for (int k0 = 0; k0 < m; k0 += Pf) {
                    cnt = 0;
#pragma unroll 4
                    for (int j = threadIdx.x; j < M; j += blockDim.x) {
                        double2 v = make_double2(0., 0.);
                        if (j < nh) {
                            v = cmul(cscale(yreg[cnt++], sc), Cf[j]);  // chirp(j; m, -)
                            if (k0) v = cmul(v, E[(4u * ((unsigned)j * (unsigned)k0 % (unsigned)m)) % (2u * LL)]);   // exp(-2 pi i j k0 / m), 2 L = 4 m
                        }
                        x[j] = v;
                    }
                    __syncthreads();
                    czt_convolve_t<512>(x, Bf, tw);
                    if (blocked) {
                        const int nb = min(Pf, m - k0);
                        for (int k = threadIdx.x; k < nb; k += blockDim.x) zbuf[k0 + k] = cscale(cmul(x[k], Cf[k]), 1.0 / M);
                        __syncthreads();
                    }
                }
                // Z(k) = chirp(k) x[k] / M ; untangle even/odd samples, apply the start-bin phase, VEL, 5 MHz cut
#pragma unroll 4
                for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                    int k1 = (k == m) ? 0 : k, k2 = (k == 0 || k == m) ? 0 : m - k;
                    double2 Z1 = blocked ? zbuf[k1] : cscale(cmul(x[k1], Cf[k1]), 1.0 / M);
                    double2 Z2 = cconj(blocked ? zbuf[k2] : cscale(cmul(x[k2], Cf[k2]), 1.0 / M));
                    double2 Ee = cscale(cadd(Z1, Z2), 0.5);
                    double2 d = cscale(csub(Z1, Z2), 0.5);
                    double2 Eo = make_double2(d.y, -d.x);  // d / i
                    double2 X = cadd(Ee, cmul(Eo, E[2 * k]));                 // exp(-2 pi i k / L)
                    unsigned ks = ((unsigned)k * sbin) % LL;                     // both factors < 2^14
                    X = cmul(X, E[2 * ks]);                                       // exp(-2 pi i k s / L)
                    double2 v;
                    if (tabulated) {
                        // VEL on the L grid for this ray, rotated to the on-sky basis (T), 5 MHz cut (efieldToVoltageConverter.py:313)
                        const double f = k * (1.0 / (L * (1. / st.fs)));
                        double2 vt = make_double2(0., 0.), vp = vt;
                        if (tab_ok && !(f < 0.005)) tab_response(*at, nodes, f, &vt, &vp);
                        const double2 coef = comp ? cadd(cscale(vt, TT[2]), cscale(vp, TT[3])) : cadd(cscale(vt, TT[0]), cscale(vp, TT[1]));
                        v = cmul(coef, X);
                    } else {
                        v = cmul(cscale(vel[k], vfac), X);
                    }
                    acc[k] = cadd(acc[k], v);
                }
                __syncthreads();
            }
        }
        // thermal noise (channelGenericNoiseAdder, added before the filter chain: simulation.py:594-606) on every channel that is not
        // noiseless, with or without a ray
        const bool noisy = nz.on && nz.amplitude[ch] > 0.;
        if (noisy) {
            const int grp = nz.ev_group ? nz.ev_group[e] : e;
            const long long gid = nz.group_id ? nz.group_id[grp] : nz.group_offset + grp;
            const int sub = nz.ev_sub ? nz.ev_sub[e] : 0;
            __syncthreads();
            for (int k = threadIdx.x; k <= m; k += blockDim.x) acc[k] = cadd(acc[k], noise_bin(nz, gid, sub, ch, k, L, st.fs));
            __syncthreads();
        }
        // filters, then back to the time domain in blocks of P samples -- unless the sum-of-magnitudes bound
        // max |V(t)| <= (fs / sqrt 2) (1 / L) (|V_0| + |V_m| + 2 sum |V_k|) already shows that no sample can reach the
        // threshold (the trace then is not needed: its maximum is reported as the negated bound)
        double vmax = 0.;
        int trig = 0;
        bool need_trace = (n_used > 0) || noisy;
        if (n_used > 0 && !exact) {
            double part = 0.;
            for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                double2 v = cmul(acc[k], Hf[k]);
                part += ((k == 0 || k == m) ? fabs(v.x) : 2. * cabs2(v));
            }
            double bnd = block_sum(part, red) * (st.fs / 1.4142135623730951 / L);
            if (!(bnd * (1 + 1e-9) >= threshold)) {
                need_trace = false;
                vmax = -bnd;
            }
        }
        if (need_trace) {
            const double scale = st.fs / 1.4142135623730951 / L;
            const bool ch_on = (!st.trig_on || st.trig_on[ch]);
            czt_inverse_blocks<512>(x, Bi, tw, E, Ci, L, m, M,
                [&](int k) {
                    double2 v = cmul(acc[k], Hf[k]);
                    // Hermitian folding of irfft: DC and Nyquist real and single, the rest doubled
                    if (k == 0 || k == m) v = make_double2(v.x, 0.);
                    else v = cscale(v, 2.);
                    return v;
                },
                [&](int ng, double2 u) {
                    const double v = u.x * (1.0 / M) * scale;
                    if (out.trace) out.trace[out.trace_offset[item] + ng] = v;
                    const double av = fabs(v);
                    vmax = fmax(vmax, av);
                    if (ng < L - 1 && av >= threshold && ch_on) trig = 1;
                });
        }
        if (env_trace && (n_used > 0 || noisy)) {
            // envelope trigger (envelopeTrigger.py:14-31 on channel.get_filtered_trace(passband, 'butter', order)): the channel
            // spectrum through the trigger's band pass, then the analytic signal -- the one-sided sum the inverse chirp-z forms anyway
            // (DC and Nyquist once and real, the bins between twice: scipy.signal.hilbert's weights); its modulus is the envelope
            const double scale = st.fs / 1.4142135623730951 / L;
            const double dfL = 1.0 / (L * (1. / st.fs));
            czt_inverse_blocks<512>(x, Bi, tw, E, Ci, L, m, M,
                [&](int k) {
                    double2 v = cmul(cmul(acc[k], Hf[k]), apply_filters(make_double2(1., 0.), k * dfL, envf));
                    if (k == 0 || k == m) v = make_double2(v.x, 0.);
                    else v = cscale(v, 2.);
                    return v;
                },
                [&](int ng, double2 u) { env_trace[out.trace_offset[item] + ng] = cabs2(u) * (1.0 / M) * scale; });
        }
        if (trig) s_trig = 1;
        double vm = need_trace ? block_max(vmax, red) : vmax;
        if (threadIdx.x == 0) {
            out.maxV[item] = vm;
            if (s_trig) out.triggered[e] = 1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernel: per-efield voltages on the native N grid and their Hilbert-envelope maximum
// (efieldToVoltageConverterPerEfield.py:28-101 -> channelAddCableDelay -> filter chain ->
// simulation._calculate_amp_per_ray_solution :1868-1886).  One block (256) per candidate event group, its rays in turn:
// V_k = B_tab(f_k) H(f_k) (vfac_t G_theta,k + vfac_p G_phi,k) (5 MHz cut inside B), analytic signal = N-point complex
// inverse transform of the one-sided spectrum (scipy.signal.hilbert), envelope maximum and the time of its first maximum.
// The antenna / filter tables on the N grid are the length tables of "L = N".  LDS: N complex + (N/2 + 1) doubles.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ray_envelope_kernel(const int* __restrict__ n_cand, const int* __restrict__ item_event, RayWork w, EventOut ev,
                    StationDev st, int ask_model, const double2* __restrict__ tw, LengthTables tab,
                    const int* __restrict__ len_index_N, int log2n, double* __restrict__ max_env,
                    double* __restrict__ signal_time, const double2* __restrict__ spec, double2* __restrict__ tab_nodes,
                    double* __restrict__ amp_scratch)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = st.N, nh = N / 2;
    double2* x = (double2*)smem;
    double2* nodes = tab_nodes ? tab_nodes + (long)blockIdx.x * 2 * st.max_tab_freq : nullptr;
    const bool pow2 = st.np.log2nh >= 0;
    // (N > 4096: the N-point buffer -- or Bluestein's 8192 points -- takes the LDS, the amplitude table a row of HBM scratch)
    double* amp = amp_scratch ? amp_scratch + (long)blockIdx.x * (nh + 1) : (double*)(x + (pow2 ? N : nplan_points(st.np)));
    __shared__ RayShared rs;
    __shared__ double red[256];
    __shared__ int red_i[256];
    const int il = *len_index_N;
    const int n_ev = *n_cand;
    for (int ie = blockIdx.x; ie < n_ev; ie += gridDim.x) {
      const int e = item_event[ie];
      const int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
      for (int r = r0; r < r1; r++) {
        if (threadIdx.x == 0) rs.ask = w.ask[r];
        for (int i = threadIdx.x; i < st.n_fc; i += blockDim.x) rs.att[i] = w.att[(long)r * st.n_fc + i];
        __syncthreads();
        if (!spec) fill_amplitude(amp, st, rs);
        // general path (time-domain emission model, birefringence): the ray's on-sky spectra are in HBM, polarisation, Fresnel
        // coefficients, attenuation and propagation included ([ray][eTheta, ePhi][N / 2 + 1], general_spectrum_kernel)
        const double2* Et = spec ? spec + (long)r * 2 * (nh + 1) : nullptr;
        const double2* Ep = spec ? Et + (nh + 1) : nullptr;
        const double2* vel = tab.vel + ((long)il * NRHIP_N_ANT_TAB + w.tab[r]) * NRHIP_SPEC_STRIDE;
        const double2* Hf = tab.H + ((long)il * st.n_fsets + (st.ch_fset ? st.ch_fset[w.ch[r]] : 0)) * NRHIP_SPEC_STRIDE;
        const double vt = w.vfac_t[r], vp = w.vfac_p[r], pt = w.pol_theta[r], pp = w.pol_phi[r];
        const double2 rt = w.r_theta[r], rp = w.r_phi[r];
        const double scale = st.fs / 1.4142135623730951 / N;  // freq2time
        // tabulated antenna pattern of the ray's channel: the angular interpolation once per frequency node of the table (as
        // channel_kernel does), the frequency interpolation per bin; rotated to the on-sky basis by the ray's T
        const bool tabulated = nodes && st.ant_model[w.ch[r]] == 3;
        const AntTabDev* at = tabulated ? &st.ant_tabs[st.ant_tab_index[w.ch[r]]] : nullptr;
        const double* TT = w.vel_T + 4 * (long)r;
        bool tab_ok = false;
        if (tabulated) {
            const TabAngles ta = tab_angles(*at, w.theta_ant[r], w.phi_ant[r]);
            tab_ok = ta.ok;
            if (ta.ok)
                for (int iF = threadIdx.x; iF < at->nF; iF += blockDim.x) tab_node(*at, ta, iF, &nodes[iF], &nodes[at->nF + iF]);
            __syncthreads();
        }
        auto one_sided = [&](int k) -> double2 {
            double2 v = make_double2(0., 0.);
            if (k > 0 && k < nh) {
                double2 Gt, Gp;
                if (spec) { Gt = Et[k]; Gp = Ep[k]; }
                else {
                    Gt = field_bin(k, amp[k], N, st.fs, pt, rt, 0., false, ask_model, floor(2.0 * st.fs));
                    Gp = field_bin(k, amp[k], N, st.fs, pp, rp, 0., false, ask_model, floor(2.0 * st.fs));
                }
                if (tabulated) {
                    const double f = k * (1.0 / (N * (1. / st.fs)));
                    double2 at_t = make_double2(0., 0.), at_p = at_t;
                    if (tab_ok && !(f < 0.005)) tab_response(*at, nodes, f, &at_t, &at_p);   // 5 MHz cut (efieldToVoltageConverterPerEfield.py)
                    const double2 ct = cadd(cscale(at_t, TT[0]), cscale(at_p, TT[1])), cp = cadd(cscale(at_t, TT[2]), cscale(at_p, TT[3]));
                    v = cscale(cmul(Hf[k], cadd(cmul(ct, Gt), cmul(cp, Gp))), 2. * scale);
                } else {
                    double2 E = cadd(cscale(Gt, vt), cscale(Gp, vp));
                    v = cscale(cmul(cmul(vel[k], Hf[k]), E), 2. * scale);
                }
            }
            return v;
        };
        double mx = -1.;
        int imx = 0;
        if (pow2) {
            for (int k = threadIdx.x; k < N; k += blockDim.x) x[k] = one_sided(k);
            __syncthreads();
            fft_dif(x, log2n, tw, true);  // inverse, natural -> bit-reversed
            for (int n = threadIdx.x; n < N; n += blockDim.x) {  // n ascending per thread: first maximum kept
                double a = cabs2(x[bitrev(n, log2n)]);
                if (a > mx) { mx = a; imx = n; }
            }
        } else {
            // a[n] = sum_{k < nh} V_k e^{2 pi i k n / N}: the even samples are the nh-point inverse transform of V, the odd ones
            // that of V_k e^{2 pi i k / N} (two Bluestein transforms of the station's plan)
            for (int par = 0; par < 2; par++) {
                __syncthreads();
                for (int k = threadIdx.x; k < nh; k += blockDim.x) {
                    double2 v = one_sided(k);
                    if (par) v = cmul(v, cconj(nplan_w(st.np, k, tw)));
                    x[k] = v;
                }
                __syncthreads();
                nplan_fft(x, st.np, tw, true);
                for (int j = threadIdx.x; j < nh; j += blockDim.x) {
                    const double a = cabs2(x[nplan_idx(st.np, j)]);   // (natural order after Bluestein, block-permuted after the odd-radix plan)
                    const int n = 2 * j + par;
                    if (a > mx || (a == mx && n < imx)) { mx = a; imx = n; }
                }
            }
        }
        red[threadIdx.x] = mx;
        red_i[threadIdx.x] = imx;
        __syncthreads();
        for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) {
            if ((int)threadIdx.x < s_) {
                double a = red[threadIdx.x + s_];
                int ia = red_i[threadIdx.x + s_];
                if (a > red[threadIdx.x] || (a == red[threadIdx.x] && ia < red_i[threadIdx.x])) {
                    red[threadIdx.x] = a;
                    red_i[threadIdx.x] = ia;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            max_env[r] = red[0];
            signal_time[r] = w.t0[r] + st.cable[w.ch[r]] + red_i[0] / st.fs;
        }
        __syncthreads();
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// kernel: efieldToVoltageConverter.run (efieldToVoltageConverter.py:111-345) for ONE station event on arbitrary
// electric-field traces (the module-level drop-in; the simulation path uses channel_kernel, which generates the
// fields itself).  One block (512) per channel; every efield of the channel: sub-sample shift (FFT phase ramp on the
// N grid, skipped within 1e-5 of a sample like BaseTrace.apply_time_shift), chirp-z onto the L grid, VEL, 5 MHz cut,
// sum; optionally the station's filter chain; back to L samples.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
efield_channel_kernel(int n_efields, const double* __restrict__ traces, const double* __restrict__ t0,
                      const double* __restrict__ zenith, const double* __restrict__ azimuth,
                      const int* __restrict__ channel, StationDev st, int L, double t_min, int apply_filter,
                      const double2* __restrict__ tw, LengthTables tab, double2* __restrict__ scratch, int log2nh,
                      double* __restrict__ V, double2* __restrict__ tab_nodes)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int M = FFT_MAX, N = st.N, nh = N / 2, m = L / 2;
    const int ch = blockIdx.x;
    double2* x = (double2*)smem;
    double2* acc = scratch + (long)blockIdx.x * 2 * NRHIP_SPEC_STRIDE;
    double2* zbuf = acc + NRHIP_SPEC_STRIDE;   // the forward transform's outputs when they come in blocks (and trace_to_packed's row)
    const double2 *Bf = tab.B_fwd, *Bi = tab.B_inv, *E = tab.E, *Cf = tab.Cf, *Ci = tab.Ci;
    const double2* Hf = tab.H + (long)(st.ch_fset ? st.ch_fset[ch] : 0) * NRHIP_SPEC_STRIDE;
    const unsigned LL = (unsigned)L;
    const double res = 1. / st.fs;
    __shared__ double sT[4], s_th, s_ph, s_vt, s_vp;
    __shared__ int s_tab;
    const bool tabulated = (st.ant_model[ch] == 3);
    const AntTabDev* at = tabulated ? &st.ant_tabs[st.ant_tab_index[ch]] : nullptr;
    double2* nodes = tabulated ? tab_nodes + (long)blockIdx.x * 2 * st.max_tab_freq : nullptr;
    for (int k = threadIdx.x; k <= m; k += blockDim.x) acc[k] = make_double2(0., 0.);
    __syncthreads();
    int n_used = 0;
    for (int e = 0; e < n_efields; e++) {
        if (channel[e] != ch) continue;
        n_used++;
        if (threadIdx.x == 0) {
            double ph_a;
            antenna_frame(zenith[e], azimuth[e], st.rot + 9 * ch, st.rot_inv + 9 * ch, sT, &s_th, &ph_a);
            s_ph = ph_a;
            if (tabulated) { s_vt = s_vp = 1.; s_tab = 0; }
            else antenna_factors(st.ant_model[ch], sT, s_th, ph_a, &s_vt, &s_vp, &s_tab);
        }
        __syncthreads();
        bool tab_ok = false;
        if (tabulated) {
            const TabAngles ta = tab_angles(*at, s_th, s_ph);
            tab_ok = ta.ok;
            if (ta.ok)
                for (int iF = threadIdx.x; iF < at->nF; iF += blockDim.x) tab_node(*at, ta, iF, &nodes[iF], &nodes[at->nF + iF]);
            __syncthreads();
        }
        double start_time = t0[e] - t_min + st.cable[ch] + 0;
        long start_bin = (long)rint(start_time / res);
        const unsigned sbin = (unsigned)(((start_bin % (long)L) + (long)L) % (long)L);
        double rem = start_time - start_bin * res;
        bool shift = !(fabs(rint(rem * st.fs) - rem * st.fs) < 1e-5);
        const double dir = 1., Tt = s_vt, Tp = s_vp;
        const double2* vel = tab.vel + (long)s_tab * NRHIP_SPEC_STRIDE;
        for (int comp = 0; comp < 2; comp++) {
            const double vfac = (comp ? Tp : Tt) * dir;
            if (vfac == 0.) continue;
            // y_j = e[2j] + i e[2j+1], sub-sample shift on the N grid (base_trace.py:273-276); then exactly channel_kernel's steps
            // for a ray trace from HBM: a_j = y_j chirp_j sqrt(2) / fs (time2freq), the m output bins in blocks of Pf
            trace_to_packed(x, traces + ((long)e * 2 + comp) * N, N, st.np, st.fs, rem, shift, tw, zbuf);
            const double sc = (shift ? 1.0 / nh : 1.0) * (1.4142135623730951 / st.fs);
            double2 yreg[FFT_MAX / 512];   // N / 2 <= 7168 points on 512 threads
            int cnt = 0;
            for (int j = threadIdx.x; j < nh; j += blockDim.x) yreg[cnt++] = shift ? x[nplan_idx(st.np, j)] : x[j];
            __syncthreads();
            const int Pf = min(m, M - nh + 1);
            const bool blocked = Pf < m;
            for (int k0 = 0; k0 < m; k0 += Pf) {
                cnt = 0;
                for (int j = threadIdx.x; j < M; j += blockDim.x) {
                    double2 v = make_double2(0., 0.);
                    if (j < nh) {
                        v = cmul(cscale(yreg[cnt++], sc), Cf[j]);
                        if (k0) v = cmul(v, E[(4u * ((unsigned)j * (unsigned)k0 % (unsigned)m)) % (2u * LL)]);   // exp(-2 pi i j k0 / m)
                    }
                    x[j] = v;
                }
                __syncthreads();
                czt_convolve_t<512>(x, Bf, tw);
                if (blocked) {
                    const int nb = min(Pf, m - k0);
                    for (int k = threadIdx.x; k < nb; k += blockDim.x) zbuf[k0 + k] = cscale(cmul(x[k], Cf[k]), 1.0 / M);
                    __syncthreads();
                }
            }
            for (int k = threadIdx.x; k <= m; k += blockDim.x) {
                int k1 = (k == m) ? 0 : k, k2 = (k == 0 || k == m) ? 0 : m - k;
                double2 Z1 = blocked ? zbuf[k1] : cscale(cmul(x[k1], Cf[k1]), 1.0 / M);
                double2 Z2 = cconj(blocked ? zbuf[k2] : cscale(cmul(x[k2], Cf[k2]), 1.0 / M));
                double2 Ee = cscale(cadd(Z1, Z2), 0.5);
                double2 d = cscale(csub(Z1, Z2), 0.5);
                double2 Eo = make_double2(d.y, -d.x);
                double2 X = cadd(Ee, cmul(Eo, E[2 * k]));
                unsigned ks = ((unsigned)k * sbin) % LL;
                X = cmul(X, E[2 * ks]);
                if (tabulated) {
                    const double f = k * (1.0 / (L * res));
                    double2 vt = make_double2(0., 0.), vp = vt;
                    if (tab_ok && !(f < 0.005)) tab_response(*at, nodes, f, &vt, &vp);
                    const double2 coef = comp ? cadd(cscale(vt, sT[2]), cscale(vp, sT[3])) : cadd(cscale(vt, sT[0]), cscale(vp, sT[1]));
                    acc[k] = cadd(acc[k], cmul(coef, X));
                } else {
                    acc[k] = cadd(acc[k], cmul(cscale(vel[k], vfac), X));
                }
            }
            __syncthreads();
        }
    }
    // back to the time domain (freq2time: irfft * fs / sqrt 2)
    const double scale = st.fs / 1.4142135623730951 / L;
    czt_inverse_blocks<512>(x, Bi, tw, E, Ci, L, m, M,
        [&](int k) {
            double2 v = make_double2(0., 0.);
            if (n_used > 0) {
                v = apply_filter ? cmul(acc[k], Hf[k]) : acc[k];
                if (k == 0 || k == m) v = make_double2(v.x, 0.);
                else v = cscale(v, 2.);
            }
            return v;
        },
        [&](int ng, double2 u) { V[(long)ch * L + ng] = u.x * (1.0 / M) * scale; });
}

// ---------------------------------------------------------------------------------------------------------
// stand-alone Askaryan spectrum (askaryan.get_frequency_spectrum) for the drop-in Python API
// ---------------------------------------------------------------------------------------------------------
__global__ void askaryan_spectrum_kernel(int n, const double* __restrict__ energy, const double* __restrict__ theta,
                                         const int* __restrict__ shower_type, const double* __restrict__ n_index,
                                         const double* __restrict__ R, const double* __restrict__ k_L, int model, int N,
                                         double dt, double2* __restrict__ spec)
{
    const int nf = N / 2 + 1;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        __shared__ AskaryanConst a;
        if (threadIdx.x == 0) a = askaryan_setup(model, energy[i], theta[i], shower_type[i], n_index[i], R[i], k_L[i]);
        __syncthreads();
        for (int k = threadIdx.x; k < nf; k += blockDim.x) {
            double amp1 = (k > 0 && k < N / 2) ? askaryan_amplitude(k * (1.0 / (N * dt)), log(k * (1.0 / (N * dt))), a) : 0.;
            spec[(long)i * nf + k] = field_bin(k, amp1, N, 1. / dt, 1.0, make_double2(1., 0.), 0., false, model,
                                               floor(2.0 / dt));
        }
        __syncthreads();
    }
}

// test hook: out[k] = sum_j in[j] exp(sgn 2 pi i j k / Q), k < n_out, one transform per block
__global__ void __launch_bounds__(1024)
czt_test_kernel(int n_batch, int n_in, int n_out, int Q, double sgn, const double2* __restrict__ in,
                double2* __restrict__ outp, const double2* __restrict__ tw, double2* __restrict__ Bscratch)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    const int M = FFT_MAX;
    double2* B = Bscratch + (long)blockIdx.x * M;
    czt_build_table(x, FFT_LOG2_MAX, n_in, n_out, Q, sgn, tw);
    for (int i = threadIdx.x; i < M; i += blockDim.x) B[i] = x[i];
    __syncthreads();
    for (int b = blockIdx.x; b < n_batch; b += gridDim.x) {
        for (int j = threadIdx.x; j < M; j += blockDim.x)
            x[j] = (j < n_in) ? cmul(in[(long)b * n_in + j], chirp(j, Q, sgn)) : make_double2(0., 0.);
        __syncthreads();
        czt_convolve(x, FFT_LOG2_MAX, B, tw);
        for (int k = threadIdx.x; k < n_out; k += blockDim.x)
            outp[(long)b * n_out + k] = cscale(cmul(x[k], chirp(k, Q, sgn)), 1.0 / M);
        __syncthreads();
    }
}

// tables of the N / 2-point Bluestein transforms of a station whose N / 2 is not a power of two (NPlan): one block
__global__ void __launch_bounds__(1024)
nplan_tables_kernel(int nh, int log2p, double2* __restrict__ wN, double2* __restrict__ cw, double2* __restrict__ Bf,
                    double2* __restrict__ Bi, const double2* __restrict__ tw)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    const int P = 1 << log2p;
    for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
        double sn, cs;
        sincospi(-(double)k / (double)nh, &sn, &cs);   // exp(-2 pi i k / N), N = 2 nh
        wN[k] = make_double2(cs, sn);
        if (cw && k < nh) cw[k] = chirp(k, nh, -1.);
    }
    if (!cw) return;   // the mixed-radix plan of the longest traces only needs wN
    for (int pass = 0; pass < 2; pass++) {
        czt_build_table(x, log2p, nh, nh, nh, pass ? 1. : -1., tw);
        double2* B = pass ? Bi : Bf;
        for (int i = threadIdx.x; i < P; i += blockDim.x) B[i] = x[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static inline unsigned grid_for(long n, int block) { return (unsigned)((n + block - 1) / block); }
static void set_big_lds();
__global__ void general_spectrum_kernel(int n_rays, RayWork w, StationDev st, int ask_model, const double* __restrict__ arz_trace,
                                        const double2* __restrict__ tw, int log2nh, double2* __restrict__ spec, double* __restrict__ amp_scratch,
                                        const int* __restrict__ silent);
__global__ void general_trace_kernel(int n_rays, StationDev st, const double2* __restrict__ spec, const double2* __restrict__ tw,
                                     int log2nh, double* __restrict__ traces, double* __restrict__ max_efield,
                                     const int* __restrict__ active, const double* __restrict__ bound);

void launch_select_rays(hipStream_t s, long n_pairs, int n_ch, const double* vertex, const double* zen, const double* az,
                        const RayRecords& rec, const IceConst& m, double cut, int* keep)
{
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(select_rays_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, s, n_pairs, n_ch, vertex, zen, az,
                       rec, m, cut, keep);
}
long scan_tiles(long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
void launch_exclusive_scan(hipStream_t s, long n, const int* in, int* out, int* tile_tmp)
{
    if (n <= 0) return;
    long nt = scan_tiles(n);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(256), 0, s, n, in, tile_tmp);
    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, s, (int)nt, tile_tmp);
    hipLaunchKernelGGL(scan_within_tiles_kernel, dim3((unsigned)nt), dim3(256), 0, s, n, in, tile_tmp, out);
}
void launch_scatter_slots(hipStream_t s, long n_slots, const int* keep, const int* offset, int* ray_slot)
{
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(scatter_slots_kernel, dim3(grid_for(n_slots, 256)), dim3(256), 0, s, n_slots, keep, offset, ray_slot);
}
void launch_ray_setup(hipStream_t s, int n_rays, int n_ch, const int* ray_slot, const double* vertex, const double* zen,
                      const double* az, const RayRecords& rec, const IceConst& m, const StationDev& st, const RayWork& w,
                      const EventIn& evin, int ask_model, const int* foc_n_sol, const double* foc_launch, double foc_dz,
                      double foc_limit, double refl_coefficient, double refl_phase, double pol_ephi)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(ray_setup_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, n_ch, ray_slot, vertex, zen,
                       az, rec, m, st, w, evin, ask_model, foc_n_sol, foc_launch, foc_dz, foc_limit, refl_coefficient, refl_phase,
                       pol_ephi);
}

// ---- paths with bottom reflections: attenuation = product over the path segments ----------------------------------------
// per kept ray the launch parameter and integration limits of its segments, gathered from the slot tables
__global__ void gather_segments_kernel(int n_rays, int NS, const int* __restrict__ ray_slot, const double* __restrict__ seg_C0,
                                       const double* __restrict__ seg_zint, double* __restrict__ ray_seg_C0,
                                       double* __restrict__ ray_seg_zint)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_rays * NS) return;
    const long r = i / NS, j = i % NS, q = (long)ray_slot[r] * NS + j;
    ray_seg_C0[i] = seg_C0[q];
    for (int d = 0; d < 3; d++) ray_seg_zint[3 * i + d] = seg_zint[3 * q + d];
}
__global__ void segment_items_kernel(int n_active, int NS, const int* __restrict__ active_list, int* __restrict__ items)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_active * NS) return;
    items[i] = active_list[i / NS] * NS + (int)(i % NS);
}
__global__ void segment_product_rays_kernel(int n_active, int NS, int n_fc, const int* __restrict__ active_list,
                                            const double* __restrict__ ray_seg_C0, const double* __restrict__ seg_att,
                                            double* __restrict__ att)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_active * n_fc) return;
    const long r = active_list[i / n_fc];
    const int f = (int)(i % n_fc);
    double a = NAN;
    for (int j = 0; j < NS; j++) {
        if (isnan(ray_seg_C0[r * NS + j])) continue;
        const double v = seg_att[(r * NS + j) * n_fc + f];
        a = isnan(a) ? v : a * v;
    }
    att[r * n_fc + f] = a;
}
void launch_gather_segments(hipStream_t s, int n_rays, int NS, const int* ray_slot, const double* seg_C0, const double* seg_zint,
                            double* ray_seg_C0, double* ray_seg_zint)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(gather_segments_kernel, dim3(grid_for((long)n_rays * NS, 256)), dim3(256), 0, s, n_rays, NS, ray_slot, seg_C0,
                       seg_zint, ray_seg_C0, ray_seg_zint);
}
void launch_segment_items(hipStream_t s, int n_active, int NS, const int* active_list, int* items)
{
    if (n_active <= 0) return;
    hipLaunchKernelGGL(segment_items_kernel, dim3(grid_for((long)n_active * NS, 256)), dim3(256), 0, s, n_active, NS, active_list, items);
}
void launch_segment_product_rays(hipStream_t s, int n_active, int NS, int n_fc, const int* active_list, const double* ray_seg_C0,
                                 const double* seg_att, double* att)
{
    if (n_active <= 0) return;
    hipLaunchKernelGGL(segment_product_rays_kernel, dim3(grid_for((long)n_active * n_fc, 256)), dim3(256), 0, s, n_active, NS, n_fc,
                       active_list, ray_seg_C0, seg_att, att);
}
// speedup.distance_cut for the reflection finder (which has no such input): pairs farther apart than the shower's cut get no solution
__global__ void distance_cut_pairs_kernel(long n_pairs, int n_ch, const double* __restrict__ vertex, const double* __restrict__ pos,
                                          const double* __restrict__ max_dist, int* __restrict__ n_sol)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const long e = i / n_ch;
    const int c = (int)(i % n_ch);
    const double dx = pos[3 * c] - vertex[3 * e], dy = pos[3 * c + 1] - vertex[3 * e + 1], dz = pos[3 * c + 2] - vertex[3 * e + 2];
    if (sqrt(dx * dx + dy * dy + dz * dz) > max_dist[e]) n_sol[i] = 0;
}
void launch_distance_cut_pairs(hipStream_t s, long n_pairs, int n_ch, const double* vertex, const double* pos, const double* max_dist,
                               int* n_sol)
{
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(distance_cut_pairs_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, s, n_pairs, n_ch, vertex, pos, max_dist, n_sol);
}
void launch_ray_limits_from_slots(hipStream_t s, int n_rays, int n_ch, const int* ray_slot, const double* vertex,
                                  const double* chan_pos, const RayRecords& rec, const IceConst& m, double* zint)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(ray_limits_from_slots_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, n_ch, ray_slot,
                       vertex, chan_pos, rec, m, zint);
}
static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

void launch_amp_bound(hipStream_t s, int n_rays, const RayWork& w, const StationDev& st, const IceConst& m,
                      const double* vertex, const double* zint, double* bound, double* max_efield, double cut)
{
    if (n_rays <= 0) return;
    int grid = (n_rays + 4 * AB_RT - 1) / (4 * AB_RT);
    if (grid > 256 * 32) grid = 256 * 32;
    if (getenv("NRHIP_AMP_BOUND_EXACT")) cut = -1.;   // (A / B: always the 2047-term sum)
    hipLaunchKernelGGL(amp_bound_kernel, dim3(grid), dim3(256), 0, s, n_rays, w, st, m, vertex, zint, bound, max_efield, cut);
}
void launch_group_ray_range(hipStream_t s, int n_groups, const int* group_begin, int n_ch, const int* slot_offset, int* grp_ray,
                            int stride)
{
    hipLaunchKernelGGL(group_ray_range_kernel, dim3(grid_for(n_groups + 1, 256)), dim3(256), 0, s, n_groups, group_begin, n_ch,
                       slot_offset, grp_ray, stride);
}
void launch_follower_list(hipStream_t s, int n_ev, const EventOut& ev, const double* att, int n_fc, int n_rays, int* flag, int* offset,
                          int* scan_tmp, int* list, int* ray_active)
{
    if (n_ev <= 0 || n_rays <= 0) return;
    (void)hipMemsetAsync(flag, 0, sizeof(int) * ((size_t)n_rays + 1), s);
    hipLaunchKernelGGL(follower_flags_kernel, dim3(grid_for(n_ev, 256)), dim3(256), 0, s, n_ev, ev, att, n_fc, n_rays, flag, ray_active);
    if (!list) return;   // flags only: the caller lists them by quadrature work (launch_quad_class_list)
    launch_exclusive_scan(s, (long)n_rays + 1, flag, offset, scan_tmp);
    hipLaunchKernelGGL(scatter_flagged_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, flag, offset, list);
}
// sum-of-magnitudes bound and Parseval norm (what the channel prefilter multiplies) of the listed rays with their computed attenuation
void launch_efield_bound_list(hipStream_t s, int n_list, const int* list, const RayWork& w, const StationDev& st, double min_efield,
                              double* max_efield, int* need_scratch)
{
    if (n_list <= 0) return;
    int gridA = (n_list + 4 * AB_RT - 1) / (4 * AB_RT);
    if (gridA > 256 * 32) gridA = 256 * 32;
    hipLaunchKernelGGL(efield_bound_kernel, dim3(gridA), dim3(256), 0, s, n_list, list, w, st, min_efield, 0, max_efield, need_scratch);
}
void launch_event_possible(hipStream_t s, int n_events, int n_ch, const int* slot_offset, const double* bound,
                           double min_efield, int* ray_active, int own_only)
{
    if (n_events <= 0) return;
    hipLaunchKernelGGL(event_possible_kernel, dim3(grid_for(n_events, 256)), dim3(256), 0, s, n_events, n_ch, slot_offset,
                       bound, min_efield, ray_active, own_only);   // (a block = four waves of 64 groups each)
}
void launch_active_class_flags(hipStream_t s, int n_rays, const int* active, const int* ray_slot2, const int* slot_type,
                               int* flags)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(active_class_flags_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, active,
                       ray_slot2, slot_type, flags);
}
// the active rays in the order of their predicted quadrature work; counts / offset: QC_NC * ceil(n_rays / 256) + 1 ints, cls: n_rays
// bytes; the number of active rays ends up in offset[quad_class_entries(n_rays) - 1]
long quad_class_entries(int n_rays) { return (long)QC_NC * ((n_rays + 255) / 256) + 1; }
void launch_quad_class_list(hipStream_t s, int n_rays, const int* active, const int* ray_slot2, const int* slot_type, const double* C0,
                            const double* zint, const IceConst& m, signed char* cls, int* counts, int* offset, int* scan_tmp, int* list)
{
    if (n_rays <= 0) return;
    const int nb = (n_rays + 255) / 256;
    static const int ascending = getenv("NRHIP_ATT_ASCENDING") ? 1 : 0;   // (experiment switch: lightest class first)
    hipLaunchKernelGGL(quad_class_count_kernel, dim3(nb), dim3(256), 0, s, n_rays, active, ray_slot2, slot_type, C0, zint, m, cls, counts, ascending);
    (void)hipMemsetAsync(counts + (long)QC_NC * nb, 0, sizeof(int), s);
    launch_exclusive_scan(s, (long)QC_NC * nb + 1, counts, offset, scan_tmp);
    hipLaunchKernelGGL(quad_class_scatter_kernel, dim3(nb), dim3(256), 0, s, n_rays, cls, offset, list);
}
void launch_scatter_active_class(hipStream_t s, int n_rays, const int* flags, const int* offset, int* list)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(scatter_active_class_kernel, dim3(grid_for(3L * n_rays, 256)), dim3(256), 0, s, n_rays, flags,
                       offset, list);
}
void launch_scatter_active(hipStream_t s, int n_rays, const int* active, const int* offset, int* list)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(scatter_active_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, active, offset, list);
}
// waves per block of efield_decide_kernel: what fits the CU's LDS beside the station's tables (0: the kernel does not apply)
static int efield_decide_waves(const StationDev& st)
{
    if (st.N < 64 || st.n_fc < 2) return 0;
    const long room = 160 * 1024 - 1024 - (long)ed_table_bytes(st.N);
    long nw = room / ((long)ed_wave_floats(st.n_fc) * 4);
    if (nw > ED_MAX_WAVES) nw = ED_MAX_WAVES;
    return nw >= 4 ? (int)nw : 0;
}
void launch_efield_max(hipStream_t s, int n_active, const int* active_list, int n_rays, int n_events,
                       const int* slot_offset, const RayWork& w, const EventIn& evin, const StationDev& st, int ask_model,
                       const double2* tw, double min_efield, int exact, double* max_efield, int* need_ray, int* ev_need,
                       int* ev_offset, int* scan_tmp, int* ev_list, unsigned long long* xform_count, double* amp_scratch)
{
    if (n_active <= 0) return;
    int nh = st.N / 2;
    (void)hipMemsetAsync(need_ray, 0, sizeof(int) * (size_t)n_rays, s);
    int gridA = (n_active + 4 * AB_RT - 1) / (4 * AB_RT);
    if (gridA > 256 * 32) gridA = 256 * 32;
    static const bool two_kernels = getenv("NRHIP_EFIELD_TWO_KERNELS") != nullptr;   // (experiment switch: the round-5 pair)
    const int ed_waves = efield_decide_waves(st);
    if (ask_model == 0 && ed_waves > 0 && !two_kernels) {
        // bound, Parseval norm and the samples next to the pulse centre in one pass, 32 rays per wave (matrix cores)
        set_big_lds();
        const int n_tiles = (n_active + 31) / 32;
        int gridD = (n_tiles + ed_waves - 1) / ed_waves;
        if (gridD > 256) gridD = 256;
        hipLaunchKernelGGL(efield_decide_kernel, dim3(gridD), dim3(64 * ed_waves),
                           ed_table_bytes(st.N) + (size_t)ed_waves * ed_wave_floats(st.n_fc) * 4, s, n_active, active_list, w, st,
                           min_efield, exact, max_efield, need_ray, xform_count ? xform_count + 4 : nullptr);
    } else {
    hipLaunchKernelGGL(efield_bound_kernel, dim3(gridA), dim3(256), 0, s, n_active, active_list, w, st, min_efield, exact,
                       max_efield, need_ray);
    if (!exact && ask_model == 0) {   // decide what the direct samples next to the pulse centre can decide
        int gridS = (n_active + 3) / 4;
        if (gridS > 256 * 16) gridS = 256 * 16;
        hipLaunchKernelGGL(efield_sample_kernel, dim3(gridS), dim3(256), (size_t)(nh + 1) * 13, s, n_active, active_list, w, st,
                           min_efield, max_efield, need_ray, xform_count ? xform_count + 4 : nullptr);
    }
    }
    hipLaunchKernelGGL(event_need_kernel, dim3(grid_for(n_events + 1, 256)), dim3(256), 0, s, n_events, st.n_ch, slot_offset,
                       need_ray, ev_need, max_efield, min_efield, exact);
    launch_exclusive_scan(s, (long)n_events + 1, ev_need, ev_offset, scan_tmp);
    hipLaunchKernelGGL(scatter_flagged_kernel, dim3(grid_for(n_events, 256)), dim3(256), 0, s, n_events, ev_need, ev_offset,
                       ev_list);
    set_big_lds();
    size_t lds = (size_t)nplan_points(st.np) * 16 + (amp_scratch ? 0 : (size_t)(nh + 1) * 8);
    int grid = n_events < 256 * 16 ? n_events : 256 * 16;
    if (amp_scratch && grid > RAY_AMP_ROWS) grid = RAY_AMP_ROWS;
    hipLaunchKernelGGL(efield_max_kernel, dim3(grid), dim3(256), lds, s, ev_offset + n_events, ev_list, need_ray, slot_offset,
                       w, evin, st, ask_model, tw, ilog2(nh), min_efield, exact, max_efield, xform_count, amp_scratch);
}
void launch_event_grid(hipStream_t s, int n_events, int n_ch, const int* slot_offset, const RayWork& w, const StationDev& st,
                       const double* max_efield, double min_efield, const EventOut& ev)
{
    if (n_events <= 0) return;
    hipLaunchKernelGGL(event_grid_kernel, dim3(grid_for(n_events, 256)), dim3(256), 0, s, n_events, n_ch, slot_offset, w, st,
                       max_efield, min_efield, ev);
}
void launch_candidate_flags(hipStream_t s, int n_events, int n_half, const EventOut& ev, int* cflag, int* lflag,
                            long long* n_cand_rays)
{
    int grid = grid_for(n_events + 1, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(candidate_flags_kernel, dim3(grid), dim3(256), 0, s, n_events, n_half, ev, cflag, lflag, n_cand_rays);
}
void launch_candidate_lists(hipStream_t s, int n_events, int n_half, const EventOut& ev, const int* cflag, const int* coff,
                            const int* lflag, const int* loff, int* cand, int* len_index, int* lens)
{
    int n = n_events > n_half ? n_events : n_half;
    hipLaunchKernelGGL(candidate_lists_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, n_events, n_half, ev, cflag, coff,
                       lflag, loff, cand, len_index, lens);
}
static int fft_pad_host(int i) { return i + (i >> 5) + (i >> 7); }   // fft_pad() of fft_device.h
// The instantiations of channel_conv_kernel: (capacity 13 | 12) x (ray path 0 | 2 | 4) x (noise: capacity 13 only) x (mode 0 | 1 | 2)
using ConvKernelPtr = decltype(&channel_conv_kernel<FFT_LOG2_MAX, 0, false, 0>);
template <int LOG2CAP, int WR, bool NZ>
static ConvKernelPtr conv_kernel_mode(int mode)
{
    return mode == 2 ? channel_conv_kernel<LOG2CAP, WR, NZ, 2> : mode == 1 ? channel_conv_kernel<LOG2CAP, WR, NZ, 1> : channel_conv_kernel<LOG2CAP, WR, NZ, 0>;
}
template <int LOG2CAP, bool NZ>
static ConvKernelPtr conv_kernel_rays(int wr, int mode)
{
    return wr == 4 ? conv_kernel_mode<LOG2CAP, 4, NZ>(mode) : wr == 2 ? conv_kernel_mode<LOG2CAP, 2, NZ>(mode) : conv_kernel_mode<LOG2CAP, 0, NZ>(mode);
}
static ConvKernelPtr conv_kernel_pick(int log2cap, int wr, bool noise, int mode)
{
    if (log2cap != FFT_LOG2_MAX) return conv_kernel_rays<FFT_LOG2_MAX - 1, false>(wr, mode);
    return noise ? conv_kernel_rays<FFT_LOG2_MAX, true>(wr, mode) : conv_kernel_rays<FFT_LOG2_MAX, false>(wr, mode);
}
template <class F>
static void conv_kernel_for_each(F&& f)
{
    for (int wr : {0, 2, 4})
        for (int mode : {0, 1, 2}) {
            f(conv_kernel_pick(FFT_LOG2_MAX, wr, false, mode), FFT_LOG2_MAX);
            f(conv_kernel_pick(FFT_LOG2_MAX, wr, true, mode), FFT_LOG2_MAX);
            f(conv_kernel_pick(FFT_LOG2_MAX - 1, wr, false, mode), FFT_LOG2_MAX - 1);
        }
}
static bool g_attr_set = false;
static void set_big_lds()
{
    if (g_attr_set) return;
    // static + dynamic LDS must stay within the 160 KB (163840 B) of a CU
    (void)hipFuncSetAttribute((const void*)length_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)channel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              FFT_MAX * 16 + (FFT_MAX / 2 + 1) * 8);
    conv_kernel_for_each([](auto kern, int log2cap) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes(log2cap));
    });
    (void)hipFuncSetAttribute((const void*)ray_envelope_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)czt_test_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)efield_channel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    // trace lengths that are no power of two: Bluestein on up to FFT_MAX / 2 points inside the ray kernels
    // (N = 8192: 4096 points + 4097 amplitudes = 96 KB; N between 4098 and 8190: Bluestein on FFT_MAX points, amplitudes in HBM)
    (void)hipFuncSetAttribute((const void*)efield_max_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)general_spectrum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)general_trace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    (void)hipFuncSetAttribute((const void*)efield_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (FFT_MAX + 1) * 13);
    (void)hipFuncSetAttribute((const void*)efield_decide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    (void)hipGetLastError();
    g_attr_set = true;
}
void launch_length_tables(hipStream_t s, int n_len, const int* lengths, const StationDev& st, const FilterSet* fls,
                          const double2* tw, const double2* w16, const LengthTables& tab, const int* slots)
{
    if (n_len <= 0) return;
    set_big_lds();
    int grid = n_len < 256 ? n_len : 256;
    hipLaunchKernelGGL(length_tables_kernel, dim3(grid), dim3(1024), (size_t)FFT_MAX * 16, s, n_len, lengths, slots, st, fls, tw, w16,
                       tab);
}

// per-event index into this call's sorted length list -> row of the station's table cache
__global__ void length_slot_kernel(int n_events, const int* __restrict__ ev_L, const int* __restrict__ slotmap, int* __restrict__ len_index)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_events && len_index[e] >= 0) len_index[e] = slotmap[ev_L[e] / 2];
}
void launch_length_slots(hipStream_t s, int n_events, const int* ev_L, const int* slotmap, int* len_index)
{
    if (n_events <= 0) return;
    hipLaunchKernelGGL(length_slot_kernel, dim3(grid_for(n_events, 256)), dim3(256), 0, s, n_events, ev_L, slotmap, len_index);
}
// blocks of the channel kernels (per-block scratch in HBM is sized by it): two per CU for the half-capacity convolution kernel, the
// 128 KB kernels use half of them
int channel_grid_blocks() { return 512; }
void launch_channel(hipStream_t s, int n_items, const int* item_event, const RayWork& w, const EventIn& evin,
                    const EventOut& ev, const int* ev_len_index, const StationDev& st, const FilterSet& fl, int ask_model,
                    const TriggerDev& trig, const double2* tw, const double2* w16, const LengthTables& tab, double2* scratch,
                    const ChannelOut& out, int exact, int max_length, int* need, int* need_offset, int* scan_tmp,
                    int* item_list, int* coinc_cnt, double2* conv_acc, unsigned long long* xform_count, double2* tab_nodes,
                    const double* ray_traces, int skip_off, const FilterSet* envf, double* env_trace, const NoiseDev* noise,
                    bool conv_split, double pa_amp_cut, double* amp_scratch, double* noise_buf, const int* item_need, void* conv_ws)
{
    if (skip_off < 0) skip_off = !exact;  // channels outside the trigger set are evaluated only when everything is
    if (n_items <= 0) return;
    set_big_lds();
    int nh = st.N / 2;
    int grid = n_items < channel_grid_blocks() / 2 ? n_items : channel_grid_blocks() / 2;   // 128 KB of LDS: one block per CU
    // traces up to FFT_MAX samples: prefilter, then one real convolution per listed item; longer ones (or
    // NRHIP_CHANNEL_CZT=1): chirp-z per ray (plain OR of simple thresholds only; the caller checks)
    int skip_upto = 0;
    // (the convolution kernel's ray stage is radix 2: trace lengths that are no power of two take the chirp-z kernel)
    if (tab.G && conv_ws && st.N <= FFT_MAX / 2 && st.np.log2nh >= 0 && !getenv("NRHIP_CHANNEL_CZT") && !ray_traces && !env_trace &&
        (!(noise && noise->on) || (noise_buf && !getenv("NRHIP_NOISE_CZT")))) {
        const bool pa_prune = pa_amp_cut >= 0.;   // (then the bounds are wanted although every kept item is evaluated exactly)
        hipLaunchKernelGGL(channel_prefilter_kernel, dim3(grid_for(n_items, 256)), dim3(256), 0, s, n_items, item_event, w, ev,
                           ev_len_index, st, pa_prune ? 0. : trig.prefilter(), tab.hnorm, pa_prune ? 0 : exact, out.maxV, need, skip_off);
        const int n_cand = n_items / st.n_ch;
        if (pa_prune)
            hipLaunchKernelGGL(pa_event_prune_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, n_cand, st.n_ch, st.trig_on, out.maxV,
                               need, pa_amp_cut);
        int* ev_need = need + n_items;  // [n_cand + 1]
        hipLaunchKernelGGL(channel_event_flags_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, n_cand, st.n_ch, need,
                           ev_need, (exact || pa_prune || st.max_tab_freq > 0) ? 1 : trig.n_coinc);   // (tabulated patterns: some channels are the chirp-z kernel's)
        (void)hipMemsetAsync(ev_need + n_cand, 0, sizeof(int), s);
        launch_exclusive_scan(s, (long)n_cand + 1, ev_need, need_offset, scan_tmp);
        hipLaunchKernelGGL(scatter_item_list_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, n_cand, ev_need, need_offset,
                           item_list);
        // the list in the order of the trace lengths (length_hist_kernel), then one record per entry (conv_header_kernel)
        constexpr int NK = FFT_MAX / 2 + 1;
        ConvHdr* hdr = (ConvHdr*)conv_ws;
        int *hist = (int*)(hdr + n_cand), *cursor = hist + NK + 1, *tmp = cursor + NK + 1, *sorted = tmp + scan_tiles(NK + 1);
        // (only for long lists: the six small launches of the sort cost a station call of an array -- ~15 k candidate events over
        // ~1000 lengths, whose few events per length miss the L2 together anyway -- more than it gains: config 5 258 ms unsorted
        // against 265 sorted per 3e5 events, config 3 103 against 106; the 1e6-event survey: -1 % time, -60 % of the kernel's reads)
        if (n_cand >= 32768 && !getenv("NRHIP_CONV_LIST_ORDER")) {
            (void)hipMemsetAsync(hist, 0, sizeof(int) * (NK + 1), s);
            hipLaunchKernelGGL(length_hist_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, need_offset + n_cand, item_list, item_event, ev.L, hist);
            launch_exclusive_scan(s, NK + 1, hist, cursor, tmp);
            hipLaunchKernelGGL(length_scatter_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, need_offset + n_cand, item_list, item_event, ev.L,
                               cursor, sorted);
            item_list = sorted;
        }
        const bool coinc_mode = trig.coincidence();
        hipLaunchKernelGGL(conv_header_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, need_offset + n_cand, item_list, item_event, ev,
                           ev_len_index, st.n_ch, need, out.maxV, (!exact && st.n_ch <= CONV_MAX_ORDER) ? 1 : 0, coinc_mode ? 1 : 0, exact, hdr);
        // list entries a block claims per atomic.  Measured (1e6-event survey, 83 k candidate events on 256 blocks): 1 -> 10.8 ms,
        // 2 -> 10.9, 4 -> 11.4, 8 -> 11.5 -- the coarser hand-out costs more at the tail of the list than the atomics it saves, so
        // one entry per claim it is; what pays is the record (one trip instead of eight)
        int claim = 1;
        if (getenv("NRHIP_CONV_CLAIM")) claim = std::max(1, std::min(CONV_CLAIM_MAX, atoi(getenv("NRHIP_CONV_CLAIM"))));
        // events of up to FFT_MAX / 2 samples go to the half-capacity instantiation (two blocks per CU), longer ones to the full one;
        // both walk the same list with their own counter
        // (with thermal noise every event takes the full-capacity instantiation: the noise trace is an 8192-point chirp convolution)
        const bool with_noise = noise && noise->on && noise_buf;
        const NoiseDev nz_off{0, 0ull, nullptr, nullptr, 0, nullptr, nullptr};
        const int conv_mode = trig.coincidence() ? 2 : (out.emit ? 1 : 0);   // (compile-time in the kernel)
        const bool small = st.N < FFT_MAX / 2 && conv_split && !with_noise && !getenv("NRHIP_CONV_ONE_BLOCK");   // L >= N: nothing to take otherwise
        const bool large = !small || max_length > FFT_MAX / 2;
        int* queue = ev_need + n_cand;   // the scan's zero sentinel: free again, and 0; the slot behind it for the second launch
        (void)hipMemsetAsync(queue + 1, 0, sizeof(int), s);
        if (small) {
            int blocks = channel_grid_blocks();
            if (getenv("NRHIP_CONV_SMALL_BLOCKS")) blocks = atoi(getenv("NRHIP_CONV_SMALL_BLOCKS"));
            const int cgrid = n_cand < blocks ? n_cand : blocks;
            // (wave-private ray transforms: N / 16 threads per ray have to fit the block)
            const bool wr_s = (nh == 1024 || nh == 2048) && (nh >> 3) <= CONV_THREADS(FFT_LOG2_MAX - 1) && !getenv("NRHIP_CONV_OLD_RAYS");
            auto kern_s = conv_kernel_pick(FFT_LOG2_MAX - 1, !wr_s ? 0 : (nh == 2048 ? 4 : 2), false, conv_mode);
            hipLaunchKernelGGL(kern_s, dim3(cgrid), dim3(CONV_THREADS(FFT_LOG2_MAX - 1)), (size_t)conv_lds_bytes(FFT_LOG2_MAX - 1), s,
                               need_offset + n_cand, hdr, claim, need, item_event, w, evin, ev, ev_len_index, st, ask_model, trig, tw, w16,
                               tab, ilog2(nh), out, exact, coinc_cnt, conv_acc, xform_count, queue, 0, nz_off, nullptr);
        }
        if (large) {
            const int cgrid = n_cand < channel_grid_blocks() / 2 ? n_cand : channel_grid_blocks() / 2;
            const bool wr_l = (nh == 1024 || nh == 2048) && (nh >> 3) <= CONV_NT && !getenv("NRHIP_CONV_OLD_RAYS");
            const int wr_n = !wr_l ? 0 : (nh == 2048 ? 4 : 2);
            auto kern_l = conv_kernel_pick(FFT_LOG2_MAX, wr_n, with_noise, conv_mode);
            hipLaunchKernelGGL(kern_l, dim3(cgrid), dim3(CONV_NT), (size_t)conv_lds_bytes(FFT_LOG2_MAX), s,
                               need_offset + n_cand, hdr, claim, need, item_event, w, evin, ev, ev_len_index, st, ask_model, trig, tw, w16,
                               tab, ilog2(nh), out, exact, coinc_cnt, conv_acc, xform_count, queue + (small ? 1 : 0),
                               small ? FFT_MAX / 2 : 0, with_noise ? *noise : nz_off, with_noise ? noise_buf : nullptr);
        }
        skip_upto = FFT_MAX;
        if (max_length <= FFT_MAX && !st.ant_tabs) return;
    }
    size_t lds = (size_t)FFT_MAX * 16 + (amp_scratch ? 0 : (size_t)(nh + 1) * 8);
    hipLaunchKernelGGL(channel_kernel, dim3(grid), dim3(512), lds, s, n_items, item_event, w, evin, ev, ev_len_index, st, fl,
                       ask_model, trig.threshold, tw, tab, scratch, ilog2(nh), out, exact, skip_upto, tab_nodes, ray_traces,
                       envf ? *envf : fl, env_trace, noise ? *noise : NoiseDev{0, 0ull, nullptr, nullptr, 0, nullptr, nullptr},
                       amp_scratch, item_need);
}
// ---------------------------------------------------------------------------------------------------------
// General emission / propagation path (time-domain emission models such as ARZ, birefringence): the on-sky spectra of every
// kept ray are materialised in HBM ([ray][eTheta, ePhi][N/2 + 1] complex), optionally propagated through the birefringent
// ice (birefringence.hip works on exactly this layout), turned into time traces ([ray][2][N]) and handed to the chirp-z
// channel kernel, which then reads fields instead of generating them.  calculate_sim_efield (simulation.py:221-290):
// spectrum = askaryan(...), E = pol x spectrum, apply_propagation_effects (attenuation, Fresnel, birefringence).
// ---------------------------------------------------------------------------------------------------------
// kernel: spectra.  arz_trace != nullptr: the Askaryan spectrum is fft.time2freq of the ray's eTheta ARZ trace
// ([ray][3][N], arz.hip), else the parametrisation's.  One block (256) per ray; LDS: N/2 complex + (N/2 + 1) doubles.
__global__ void __launch_bounds__(256)
general_spectrum_kernel(int n_rays, RayWork w, StationDev st, int ask_model, const double* __restrict__ arz_trace,
                        const double2* __restrict__ tw, int log2nh, double2* __restrict__ spec, double* __restrict__ amp_scratch,
                        const int* __restrict__ silent)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = st.N, nh = N / 2, n_f = nh + 1;
    double2* x = (double2*)smem;
    double* amp = amp_scratch ? amp_scratch + (long)blockIdx.x * (nh + 1) : (double*)(x + nplan_points(st.np));
    __shared__ RayShared rs;
    const double df = 1.0 / (N * (1. / st.fs));
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        if (arz_trace && silent && silent[r]) {   // beyond the emission model's 20 degrees: the trace is zero (and was not written)
            double2* Ez = spec + (long)r * 2 * n_f;
            for (int k = threadIdx.x; k < 2 * n_f; k += blockDim.x) Ez[k] = make_double2(0., 0.);
            continue;   // (block-uniform)
        }
        __syncthreads();
        if (threadIdx.x == 0) rs.ask = w.ask[r];
        for (int i = threadIdx.x; i < st.n_fc; i += blockDim.x) rs.att[i] = w.att[(long)r * st.n_fc + i];
        __syncthreads();
        const double2 rt = w.r_theta[r], rp = w.r_phi[r];
        const double pt = w.pol_theta[r], pp = w.pol_phi[r];
        double2* Eo = spec + (long)r * 2 * n_f;
        if (!arz_trace) {
            fill_amplitude(amp, st, rs);
            for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
                const double2 G = field_bin(k, amp[k], N, st.fs, 1.0, make_double2(1., 0.), 0., false, ask_model, floor(2.0 * st.fs));
                Eo[k] = cscale(cmul(G, rt), pt);
                Eo[n_f + k] = cscale(cmul(G, rp), pp);
            }
        } else {
            for (int j = threadIdx.x; j < st.n_fc; j += blockDim.x) {
                rs.xp[j] = st.fcoarse[j];
                if (j < st.n_fc - 1) rs.slope[j] = (rs.att[j + 1] - rs.att[j]) / (st.fcoarse[j + 1] - st.fcoarse[j]);
            }
            const double* tr = arz_trace + ((long)r * 3 + 1) * N;
            const double foc = w.focus[r];
            for (int j = threadIdx.x; j < nh; j += blockDim.x) x[j] = make_double2(tr[2 * j], tr[2 * j + 1]);
            __syncthreads();
            nplan_fft(x, st.np, tw, false);  // packed half-length transform (element j at nplan_idx(j))
            const double sc = 1.4142135623730951 / st.fs;  // fft.time2freq: rfft / fs * sqrt 2
            for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
                const int ka = (k == nh) ? 0 : k, kb = (k == 0 || k == nh) ? 0 : nh - k;
                const double2 Y1 = x[nplan_idx(st.np, ka)], Y2 = cconj(x[nplan_idx(st.np, kb)]);
                const double2 ge = cscale(cadd(Y1, Y2), 0.5);
                const double2 d = cscale(csub(Y1, Y2), 0.5);
                const double2 go = make_double2(d.y, -d.x);
                const double2 wk = nplan_w(st.np, k, tw);
                // attenuation: np.interp on the coarse grid for f > 0, 1 at DC (analyticraytracing.py:1075-1080)
                // (and the focusing factor on every bin but DC: spec[1:] *= focusing, :3011-3016)
                const double a = (k == 0) ? 1. : interp_seg(k * df, st.seg[k], st.n_fc, rs.xp, rs.att, rs.slope) * foc;
                const double2 S = cscale(cadd(ge, cmul(go, wk)), sc * a);
                Eo[k] = cscale(cmul(S, rt), pt);
                Eo[n_f + k] = cscale(cmul(S, rp), pp);
            }
        }
    }
}

// kernel: spectra -> traces e = irfft(E) fs / sqrt 2 ([ray][2][N]) and max |e| per ray (the candidate cut of
// simulation.py:283-285 looks at all components of the trace).  One block (256) per ray.
__global__ void __launch_bounds__(256)
general_trace_kernel(int n_rays, StationDev st, const double2* __restrict__ spec, const double2* __restrict__ tw, int log2nh,
                     double* __restrict__ traces, double* __restrict__ max_efield, const int* __restrict__ active,
                     const double* __restrict__ bound)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = st.N, nh = N / 2, n_f = nh + 1;
    double2* x = (double2*)smem;
    __shared__ double red[256];
    const double scale = st.fs / 1.4142135623730951 / nh;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        if (active && !active[r]) {  // not evaluated: "at most `bound`" (its event cannot become a candidate); no bound given: a
            if (threadIdx.x == 0 && bound) max_efield[r] = -bound[r];   // second round, the entry stays what the first left
            continue;
        }
        double mx = 0.;
        for (int comp = 0; comp < 2; comp++) {
            const double2* E = spec + ((long)r * 2 + comp) * n_f;
            __syncthreads();
            for (int k = threadIdx.x; k < nh; k += blockDim.x) {
                double2 Gk = E[k], Gc = cconj(E[nh - k]);
                if (k == 0) { Gk.y = 0.; Gc.y = 0.; }  // irfft ignores the imaginary parts of DC and Nyquist
                const double2 ge = cscale(cadd(Gk, Gc), 0.5);
                const double2 d = cscale(csub(Gk, Gc), 0.5);
                const double2 go = cmul(d, cconj(nplan_w(st.np, k, tw)));
                x[k] = make_double2(ge.x - go.y, ge.y + go.x);
            }
            __syncthreads();
            nplan_fft(x, st.np, tw, true);
            double* out = traces + ((long)r * 2 + comp) * N;
            for (int j = threadIdx.x; j < nh; j += blockDim.x) {
                const double2 y = x[nplan_idx(st.np, j)];
                const double e0 = y.x * scale, e1 = y.y * scale;
                out[2 * j] = e0;
                out[2 * j + 1] = e1;
                mx = fmax(mx, fmax(fabs(e0), fabs(e1)));
            }
        }
        mx = block_max(mx, red);
        if (threadIdx.x == 0) max_efield[r] = mx;
    }
}

// kernel: upper bound on max |e(t)| of a ray AFTER a linear propagation step whose 2-norm gain is at most exp(log_gain):
// |e(t)| <= (fs / sqrt 2) (1 / N) (|E_0| + |E_N/2| + 2 sum_k |E_k|) per component, |E'_k| <= gain ||(E_theta,k, E_phi,k)||_2.
// Lets the birefringent propagation (the expensive part of the general path) skip the events that cannot become candidates.
__global__ void __launch_bounds__(256)
general_bound_kernel(int n_rays, StationDev st, const double2* __restrict__ spec, const long long* __restrict__ log_gain,
                     double* __restrict__ bound, double* __restrict__ e_norm)
{
    __shared__ double red[256];
    const int N = st.N, nh = N / 2, n_f = nh + 1;
    for (int r = blockIdx.x; r < n_rays; r += gridDim.x) {
        const double2* Et = spec + (long)r * 2 * n_f;
        const double2* Ep = Et + n_f;
        double part = 0., part2 = 0.;
        for (int k = threadIdx.x; k <= nh; k += blockDim.x) {
            const double2 a = Et[k], b = Ep[k];
            const double m2 = a.x * a.x + a.y * a.y + b.x * b.x + b.y * b.y;
            const double m = sqrt(m2);
            part += (k == 0 || k == nh) ? m : 2. * m;
            part2 += (k == 0 || k == nh) ? m2 : 2. * m2;
        }
        const double sum = block_sum(part, red);
        const double sum2 = e_norm ? block_sum(part2, red) : 0.;
        if (threadIdx.x == 0) {
            const double gain = exp(log_gain ? (double)log_gain[r] * (1. / 1099511627776.0) : 0.);
            bound[r] = (st.fs / 1.4142135623730951 / N) * sum * gain * (1. + 1e-9);
            // L2 norm of the ray's field after the propagation, at most (Parseval on the N-sample trace e = irfft(E) fs / sqrt 2, both
            // on-sky components together; the path's gain bounds what the birefringent propagation can do to it): what the
            // Cauchy-Schwarz bound of a channel multiplies with
            if (e_norm) e_norm[r] = sqrt(st.fs * st.fs / (2. * N) * sum2) * gain * (1. + 1e-9);
        }
        __syncthreads();
    }
}

// General path, production mode: which channels of the candidate readouts can reach the trigger threshold at all?
// |V_c(t)| <= sum over the channel's rays of ||g_L|| sqrt(vfac_t^2 + vfac_p^2) ||e_r|| (Cauchy-Schwarz, the two on-sky components as a
// 2-vector).  The others are not evaluated -- and their rays, unless they decided the candidate cut, never propagated.
__global__ void __launch_bounds__(256)
general_prefilter_kernel(int n_items, const int* __restrict__ item_event, RayWork w, EventOut ev, const int* __restrict__ ev_len_index,
                         StationDev st, double threshold, const double* __restrict__ hnorm, double* __restrict__ maxV,
                         int* __restrict__ need)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const int e = item_event[item / st.n_ch], ch = item % st.n_ch;
    const int L = ev.L[e], il = ev_len_index[e];
    if (st.trig_on && !st.trig_on[ch]) { need[item] = 0; maxV[item] = NAN; return; }   // not a trigger channel
    if (L > FFT_MAX || st.ant_model[ch] == 3) { need[item] = 1; return; }               // no norm table: evaluated
    const int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
    double bnd = 0.;
    for (int r = r0; r < r1; r++) {
        if (w.ch[r] != ch) continue;
        const double vt = w.vfac_t[r], vp = w.vfac_p[r];
        bnd += w.e_norm[r] * sqrt(vt * vt + vp * vp) *
               hnorm[((long)il * st.n_fsets + (st.ch_fset ? st.ch_fset[ch] : 0)) * NRHIP_N_ANT_TAB + w.tab[r]];
    }
    const int flag = (bnd * (1 + 1e-9) >= threshold) ? 1 : 0;
    if (!flag) maxV[item] = -bnd;
    need[item] = flag;
}
// ... and the rays of those channels that have not been propagated yet
__global__ void __launch_bounds__(256)
general_mark_rays_kernel(int n_cand, const int* __restrict__ item_event, EventOut ev, RayWork w, int n_ch, const int* __restrict__ need,
                         int* __restrict__ propagated, int* __restrict__ fresh)
{
    const int ic = blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= n_cand) return;
    const int e = item_event[ic];
    const int r0 = ev.ray_begin[e], r1 = r0 + ev.n_rays[e];
    for (int r = r0; r < r1; r++) {
        const int want = need[(long)ic * n_ch + w.ch[r]] && !propagated[r];
        fresh[r] = want;
        if (want) propagated[r] = 1;
    }
}

// kernel: what the ARZ and birefringence kernels need per ray, gathered from the ray tables
__global__ void __launch_bounds__(256)
general_gather_kernel(int n_rays, int n_ch, RayWork w, EventIn evin, StationDev st, const double* __restrict__ vertex,
                      const int* __restrict__ shower_profile, const double* __restrict__ shower_rescale, int em_formula,
                      double* __restrict__ energy, int* __restrict__ type, double* __restrict__ em_factor,
                      int* __restrict__ profile, double* __restrict__ rescale, double* __restrict__ x1, double* __restrict__ x2,
                      int* __restrict__ n_steps)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int sh = w.ev[r], ch = w.ch[r];
    const double E = evin.energy[sh];
    energy[r] = E;
    type[r] = evin.shower_type[sh];
    // ARZ.em_fraction (ARZ.py:436-447), hadronic showers of the ARZ2020 parameter set only
    double f = 1.;
    if (em_formula && evin.shower_type[sh] == 0) {
        const double eps = log10(E / 1.);
        f = -21.98905 - 2.32492 * eps;
        f += 0.019650 * (eps * eps) + 13.76152 * sqrt(eps);
    }
    em_factor[r] = f;
    profile[r] = shower_profile ? shower_profile[sh] : 0;
    rescale[r] = shower_rescale ? shower_rescale[sh] : 1.;
    for (int d = 0; d < 3; d++) {
        x1[3 * (long)r + d] = vertex[3 * (long)sh + d];
        x2[3 * (long)r + d] = st.pos[3 * ch + d];
    }
    const int acc = (int)(w.R[r] / 1.);  // int(path length / m) path points (analyticraytracing.py:2417)
    n_steps[r] = acc > 1 ? acc - 1 : 0;
}

__global__ void int_to_long_kernel(int n, const int* __restrict__ in, long* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void steps_to_points_kernel(int n, const int* __restrict__ n_steps, int* __restrict__ n_points)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) n_points[i] = n_steps[i] > 0 ? n_steps[i] + 1 : 0;
}

void launch_general_spectrum(hipStream_t s, int n_rays, const RayWork& w, const StationDev& st, int ask_model,
                             const double* arz_trace, const double2* tw, double2* spec, double* amp_scratch, const int* silent)
{
    if (n_rays <= 0) return;
    const int nh = st.N / 2;
    int grid = n_rays < 256 * 64 ? n_rays : 256 * 64;
    if (amp_scratch && grid > RAY_AMP_ROWS) grid = RAY_AMP_ROWS;
    set_big_lds();
    hipLaunchKernelGGL(general_spectrum_kernel, dim3(grid), dim3(256), (size_t)nplan_points(st.np) * 16 + (amp_scratch ? 0 : (size_t)(nh + 1) * 8), s, n_rays, w, st,
                       ask_model, arz_trace, tw, ilog2(nh), spec, amp_scratch, silent);
}
void launch_general_trace(hipStream_t s, int n_rays, const StationDev& st, const double2* spec, const double2* tw,
                          double* traces, double* max_efield, const int* active, const double* bound)
{
    if (n_rays <= 0) return;
    const int nh = st.N / 2;
    int grid = n_rays < 256 * 64 ? n_rays : 256 * 64;
    set_big_lds();
    hipLaunchKernelGGL(general_trace_kernel, dim3(grid), dim3(256), (size_t)nplan_points(st.np) * 16, s, n_rays, st, spec, tw, ilog2(nh), traces,
                       max_efield, active, bound);
}
void launch_general_bound(hipStream_t s, int n_rays, const StationDev& st, const double2* spec, const long long* log_gain,
                          double* bound, double* e_norm)
{
    if (n_rays <= 0) return;
    int grid = n_rays < 256 * 64 ? n_rays : 256 * 64;
    hipLaunchKernelGGL(general_bound_kernel, dim3(grid), dim3(256), 0, s, n_rays, st, spec, log_gain, bound, e_norm);
}
void launch_general_prefilter(hipStream_t s, int n_items, const int* item_event, const RayWork& w, const EventOut& ev,
                              const int* ev_len_index, const StationDev& st, double threshold, const double* hnorm, double* maxV,
                              int* need, int n_rays, int* propagated, int* fresh)
{
    if (n_items <= 0) return;
    hipLaunchKernelGGL(general_prefilter_kernel, dim3(grid_for(n_items, 256)), dim3(256), 0, s, n_items, item_event, w, ev, ev_len_index,
                       st, threshold, hnorm, maxV, need);
    (void)hipMemsetAsync(fresh, 0, sizeof(int) * (size_t)n_rays, s);
    const int n_cand = n_items / st.n_ch;
    hipLaunchKernelGGL(general_mark_rays_kernel, dim3(grid_for(n_cand, 256)), dim3(256), 0, s, n_cand, item_event, ev, w, st.n_ch, need,
                       propagated, fresh);
}
void launch_general_gather(hipStream_t s, int n_rays, int n_ch, const RayWork& w, const EventIn& evin, const StationDev& st,
                           const double* vertex, const int* shower_profile, const double* shower_rescale, int em_formula,
                           double* energy, int* type, double* em_factor, int* profile, double* rescale, double* x1, double* x2,
                           int* n_steps, int* n_points)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(general_gather_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, n_ch, w, evin, st, vertex,
                       shower_profile, shower_rescale, em_formula, energy, type, em_factor, profile, rescale, x1, x2, n_steps);
    hipLaunchKernelGGL(steps_to_points_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, n_steps, n_points);
}
// Rays the time-domain model leaves without a signal (more than `maximum_angle` off the Cherenkov cone, ARZ.py:603-607: an all-zero
// trace, the very comparison of arz_vector_potential_kernel): a zero spectrum stays zero through any linear propagation, so their
// path steps are neither computed nor applied (n_points = 0 is "nothing to do" for both birefringence kernels)
__global__ void silent_rays_kernel(int n_rays, const double* __restrict__ view, const double* __restrict__ n_index_ray, double n_index,
                                   double maximum_angle, int* __restrict__ n_steps, int* __restrict__ n_points)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const double nidx = n_index_ray ? n_index_ray[r] : n_index;
    if (fabs(view[r] - acos(1. / nidx)) > maximum_angle) { n_steps[r] = 0; n_points[r] = 0; }
}
void launch_silent_rays(hipStream_t s, int n_rays, const double* view, const double* n_index_ray, double n_index, double maximum_angle,
                        int* n_steps, int* n_points)
{
    if (n_rays <= 0) return;
    hipLaunchKernelGGL(silent_rays_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, s, n_rays, view, n_index_ray, n_index, maximum_angle,
                       n_steps, n_points);
}
void launch_int_to_long(hipStream_t s, int n, const int* in, long* out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(int_to_long_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, n, in, out);
}

// ---------------------------------------------------------------------------------------------------------
// kernel: phased-array trigger core (NuRadioReco/modules/phasedarray/phasedArrayBase.py: phase_signals :183-215, power_sum
// :217-271, decision of phased_trigger :455-496, mode 'power_sum', no digitisation, no upsampling).  One block per candidate
// event: per beam the coherent sum of the np.roll-ed channel traces (read from the dumped traces in HBM) is built in LDS,
// then every thread squares and sums its sliding windows; any window above the threshold triggers the event.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
phased_array_kernel(int n_cand, const int* __restrict__ item_event, int n_ch, const int* __restrict__ ev_L,
                    const double* __restrict__ trace, const long* __restrict__ trace_offset, int n_pa,
                    const int* __restrict__ pa_channel, int n_beams, const int* __restrict__ rolls, int window, int step,
                    double divisor, double threshold, unsigned char* __restrict__ triggered, double* __restrict__ pa_max)
{
    extern __shared__ double coh[];
    __shared__ double red[256];
    for (int i = blockIdx.x; i < n_cand; i += gridDim.x) {
        const int e = item_event[i], L = ev_L[e];
        const int n_frames = (L - window) / step > 0 ? (L - window) / step : 0;
        double any = 0.;
        for (int b = 0; b < n_beams; b++) {
            __syncthreads();
            for (int n = threadIdx.x; n < L; n += blockDim.x) {
                double sum = 0.;
                for (int c = 0; c < n_pa; c++) {  // np.roll(trace, r)[n] = trace[(n - r) mod L]
                    int k = (n - rolls[b * n_pa + c]) % L;
                    if (k < 0) k += L;
                    sum += trace[trace_offset[(long)i * n_ch + pa_channel[c]] + k];
                }
                coh[n] = sum;
            }
            __syncthreads();
            double mx = -INFINITY;
            for (int f = threadIdx.x; f < n_frames; f += blockDim.x) {
                double p = 0.;
                for (int j = 0; j < window; j++) {
                    const double v = coh[f * step + j];
                    p += v * v;
                }
                mx = fmax(mx, p / divisor);
            }
            mx = block_max(mx, red);
            if (threadIdx.x == 0) pa_max[(long)i * n_beams + b] = mx;
            if (mx > threshold) any = 1.;
        }
        if (threadIdx.x == 0 && any > 0.) triggered[e] = 1;
    }
}

void launch_phased_array(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                         const long* trace_offset, int n_pa, const int* pa_channel, int n_beams, const int* rolls, int window,
                         int step, double divisor, double threshold, int max_length, unsigned char* triggered, double* pa_max)
{
    if (n_cand <= 0) return;
    (void)hipFuncSetAttribute((const void*)phased_array_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FFT_MAX * 8);
    int grid = n_cand < 256 * 16 ? n_cand : 256 * 16;
    hipLaunchKernelGGL(phased_array_kernel, dim3(grid), dim3(256), (size_t)max_length * 8, s, n_cand, item_event, n_ch, ev_L, trace,
                       trace_offset, n_pa, pa_channel, n_beams, rolls, window, step, divisor, threshold, triggered, pa_max);
}

// ---------------------------------------------------------------------------------------------------------
// The digitised phased-array trigger (phasedArrayTrigger with apply_digitization and up-sampling): per (candidate event, array
// channel) the channel trace goes through the trigger ADC of analogToDigitalConverter.get_digital_trace (:254-373) and
// signal_processing.digital_upsampling (:111-190, method 'fft'):
//   (1) resampling to 5 GHz (signal_processing.resample :71-108: two scipy.signal.resample calls, up by p then down by q with
//       p / q = 5 GHz / f_s) -- the trigonometric interpolant of the trace (Nyquist bin halved) sampled at num2 = (p L) // q points;
//   (2) linear interpolation at the ADC's sample times (downsampling_linear_interpolation :432-463, scipy interp1d);
//   (3) floor((V - V_min) / lsb), clipped to 0 .. 2^bits - 1, + floor(V_min / lsb) (perfect_floor_comparator :14-110), volts or counts;
//   (4) up-sampling by an integer factor: again a trigonometric interpolant (Nyquist halved), rounded for ADC counts.
// No arbitrary-length FFT is needed: only the 2 n_adc values of (1) that (2) reads are evaluated, as direct sums over the trace's
// L / 2 + 1 spectrum bins (themselves a direct sum over the L samples); phases are advanced by complex multiplication and taken
// afresh from sincospi every 32 steps, so every value is within ~1e-14 of the reference's FFT result.  One block per item;
// LDS: trace (L doubles) + spectrum (L / 2 + 1 complex) + ADC trace and its spectrum.
// ---------------------------------------------------------------------------------------------------------
__device__ inline double2 turn_phase(long num, long den)   // exp(2 pi i num / den)
{
    double sn, cs;
    sincospi(2. * (double)(num % den) / (double)den, &sn, &cs);
    return make_double2(cs, sn);
}

__global__ void __launch_bounds__(1024)
pa_digitize_kernel(int n_cand, const int* __restrict__ item_event, int n_ch, const int* __restrict__ ev_L,
                   const double* __restrict__ trace, const long* __restrict__ trace_offset, int n_pa, const int* __restrict__ pa_channel,
                   double fs, PaAdc adc, double* __restrict__ pa_trace, int* __restrict__ pa_len)
{
#pragma clang fp contract(off)   // counts are compared sample by sample with the CPU restatement: no fused rounding here
    extern __shared__ double pd_lds[];
    const int n_items = n_cand * n_pa;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ic = item / n_pa, c = item % n_pa;
        const int e = item_event[ic], L = ev_L[e], m = L / 2;
        const double* x = trace + trace_offset[(long)ic * n_ch + pa_channel[c]];
        double* sx = pd_lds;                              // [L] (+ 2: the region is reused below)
        double2* X = (double2*)(pd_lds + L + 2 + (L & 1));   // [m + 1]
        __syncthreads();
        for (int n = threadIdx.x; n < L; n += blockDim.x) sx[n] = x[n];
        __syncthreads();
        // spectrum X_k = sum_n x[n] exp(-2 pi i k n / L)
        // (two bins per thread and loop: two independent recurrences keep the FP64 pipe busy, one LDS read serves both)
        for (int k0 = threadIdx.x; k0 <= m; k0 += 2 * blockDim.x) {
            const int k1 = k0 + blockDim.x;
            const bool two = k1 <= m;
            double2 a0 = make_double2(0., 0.), a1 = a0, w0 = make_double2(1., 0.), w1 = w0;
            const double2 s0 = cconj(turn_phase(k0, L)), s1 = cconj(turn_phase(two ? k1 : 0, L));
            for (int n = 0; n < L; n++) {
                if ((n & 31) == 0) {
                    w0 = cconj(turn_phase((long)k0 * n, L));
                    w1 = cconj(turn_phase((long)(two ? k1 : 0) * n, L));
                }
                const double v = sx[n];
                a0.x += v * w0.x; a0.y += v * w0.y;
                a1.x += v * w1.x; a1.y += v * w1.y;
                w0 = cmul(w0, s0);
                w1 = cmul(w1, s1);
            }
            X[k0] = a0;
            if (two) X[k1] = a1;
        }
        __syncthreads();
        // (1) + (2): the ADC samples
        const bool resampled = !(fabs(adc.adc_fs - fs) <= 1e-8 + 1e-5 * fabs(fs));   // np.allclose
        const bool to5 = resampled && 5.0 > fs;
        long num2 = L;
        if (to5) {
            long n1 = (adc.p != 1) ? (long)adc.p * L : L;
            num2 = (adc.q != 1) ? n1 / adc.q : n1;
        }
        const long len5 = to5 ? (num2 - (num2 & 1)) : L;      // an odd number of samples loses the last one
        const double cur = to5 ? 5.0 : fs;
        int n_adc = resampled ? (int)((adc.adc_fs / cur) * (double)len5) : L;
        // the ADC trace d [n_adc] and, later, its spectrum D [n_adc / 2 + 1] reuse LDS that is dead by then: after the 5 GHz
        // interpolant is defined by X the samples sx are no longer read (d, D there; the host checks 2 n_adc + 4 <= L); without that
        // resampling X is not used at all (d there, D over sx once d is complete)
        double* d = to5 ? pd_lds : (double*)X;
        auto value5_pair = [&](long i, double& y0, double& y1) {   // samples i and i + 1 of the (re)sampled trace
            if (!to5) { y0 = sx[i]; y1 = sx[i + 1]; return; }
            double acc0 = X[0].x, acc1 = X[0].x;
            double2 w0 = make_double2(1., 0.), w1 = w0;
            const double2 s0 = turn_phase(i, num2), s1 = turn_phase(i + 1, num2);
            for (int k = 1; k <= m; k++) {
                if ((k & 31) == 1) { w0 = turn_phase((long)k * i, num2); w1 = turn_phase((long)k * (i + 1), num2); }
                else { w0 = cmul(w0, s0); w1 = cmul(w1, s1); }
                const double2 xk = X[k];
                const double t0 = xk.x * w0.x - xk.y * w0.y, t1 = xk.x * w1.x - xk.y * w1.y;
                acc0 += (k == m) ? t0 : 2. * t0;
                acc1 += (k == m) ? t1 : 2. * t1;
            }
            y0 = acc0 / L;
            y1 = acc1 / L;
        };
        const double lsb = (adc.vmax - adc.vmin) / (double)((1 << adc.n_bits) - 1);
        const double vmin_adc = floor(adc.vmin / lsb);
        const int n_dig = n_adc - (n_adc & 1);
        for (int j = threadIdx.x; j < n_dig; j += blockDim.x) {
            double v;
            if (resampled) {
                const double tn = (double)j / adc.adc_fs;
                long lo = (long)floor(tn * cur);
                while (lo > 0 && (double)lo / cur >= tn) lo--;          // times[lo] < t <= times[lo + 1] (searchsorted, side 'left')
                while ((double)(lo + 1) / cur < tn) lo++;
                if (lo > len5 - 2) lo = len5 - 2;
                if (lo < 0) lo = 0;
                const double xlo = (double)lo / cur, xhi = (double)(lo + 1) / cur;
                double ylo, yhi;
                value5_pair(lo, ylo, yhi);
                const double slope = (yhi - ylo) / (xhi - xlo);
                v = slope * (tn - xlo) + ylo;
            } else {
                v = sx[j];
            }
            if (adc.n_bits == 0) {   // no trigger ADC (apply_digitization = False): the analog trace goes on to the up-sampling
                d[j] = v;
                continue;
            }
            double cnt = floor((v - adc.vmin) / lsb);
            cnt = fmin(fmax(cnt, 0.), (double)((1 << adc.n_bits) - 1)) + vmin_adc;
            d[j] = adc.counts ? cnt : lsb * cnt;
        }
        __syncthreads();
        // (4) up-sampling
        double* out = pa_trace + (long)item * adc.stride;
        int n_up = n_dig;
        // digital_upsampling rounds its result when the input "is digital" (np.allclose(trace, np.round(trace)) :153, :182-183):
        // ADC counts always are; an analog trace is when every sample is within 1e-8 + 1e-5 |round| of an integer (a silent channel)
        bool round_out = adc.counts != 0;
        if (adc.n_bits == 0 && adc.upsampling >= 2) {
            int ok = 1;
            for (int j = threadIdx.x; j < n_dig; j += blockDim.x) {
                const double r = rint(d[j]);
                ok &= (fabs(d[j] - r) <= 1e-8 + 1e-5 * fabs(r)) ? 1 : 0;
            }
            round_out = __syncthreads_and(ok) != 0;
        }
        if (adc.upsampling >= 2) {
            const int md = n_dig / 2;
            double2* D = to5 ? (double2*)(pd_lds + n_dig + (n_dig & 1)) : (double2*)pd_lds;
            for (int k = threadIdx.x; k <= md; k += blockDim.x) {
                double2 acc = make_double2(0., 0.), w = make_double2(1., 0.);
                const double2 step = cconj(turn_phase(k, n_dig));
                for (int n = 0; n < n_dig; n++) {
                    if ((n & 31) == 0) w = cconj(turn_phase((long)k * n, n_dig));
                    acc.x += d[n] * w.x;
                    acc.y += d[n] * w.y;
                    w = cmul(w, step);
                }
                D[k] = acc;
            }
            __syncthreads();
            n_up = n_dig * adc.upsampling;
            for (int n = threadIdx.x; n < n_up; n += blockDim.x) {
                double acc = D[0].x;
                double2 w = make_double2(1., 0.);
                const double2 step = turn_phase(n, n_up);
                for (int k = 1; k <= md; k++) {
                    if ((k & 31) == 1) w = turn_phase((long)k * n, n_up);
                    else w = cmul(w, step);
                    const double t = D[k].x * w.x - D[k].y * w.y;
                    acc += (k == md) ? t : 2. * t;
                }
                acc /= n_dig;
                out[n] = round_out ? rint(acc) : acc;
            }
        } else {
            for (int n = threadIdx.x; n < n_dig; n += blockDim.x) out[n] = d[n];
        }
        if (threadIdx.x == 0) pa_len[item] = n_up - (n_up & 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same chain through the in-LDS chirp-z transform (O(L log L) per transform instead of the direct sums): four transforms per
// item, each "out[k] = sum_j in[j] exp(sgn 2 pi i j k / Q)" evaluated in blocks of P outputs with one Bluestein table per
// (trace length, stage):
//   stage 1: the trace's spectrum              n_in = L,            Q = L,      sgn -, P = FFT_MAX - L + 1
//   stage 2: the 5 GHz samples                 n_in = L / 2 + 1,    Q = num2,   sgn +, P = FFT_MAX - L / 2      (real part)
//   stage 3: (no transform) linear interpolation at the ADC times + floor comparator
//   stage 4: the ADC trace's spectrum          n_in = n_dig,        Q = n_dig,  sgn -, P = FFT_MAX - n_dig + 1
//   stage 5: the up-sampled trace              n_in = n_dig / 2 + 1, Q = n_up,  sgn +, P = FFT_MAX - n_dig / 2  (real part, rounded for counts)
// Blocks after the first shift the input by exp(sgn 2 pi i j k0 / Q).  Intermediate arrays live in HBM (a few 100 KB per item).
// ---------------------------------------------------------------------------------------------------------
struct PaSizes { int L0, m0, cyc, L, m, n_dig, n_up; long num2, len5; };
// L0: samples of the channel trace.  With a clock offset (analogToDigitalConverter.py:327-340) the trace is delayed by
// clock_offset / adc_fs through its spectrum and loses the cyc = round(delay * fs) (made even) samples that wrapped round
// (signal_processing.delay_trace :401-472): everything behind works on L = L0 - cyc samples.
__device__ __host__ inline PaSizes pa_sizes(int L0, double fs, const PaAdc& adc)
{
    PaSizes z;
    z.L0 = L0;
    z.m0 = L0 / 2;
    z.cyc = 0;
    if (adc.clock_offset) {
        z.cyc = (int)rint((adc.clock_offset / adc.adc_fs) * fs);
        if (z.cyc & 1) z.cyc++;
    }
    const int L = L0 - z.cyc;
    z.L = L;
    z.m = L / 2;
    long n1 = (adc.p != 1) ? (long)adc.p * L : L;
    z.num2 = (adc.q != 1) ? n1 / adc.q : n1;
    z.len5 = z.num2 - (z.num2 & 1);
    const int n_adc = (int)((adc.adc_fs / 5.0) * (double)z.len5);
    z.n_dig = n_adc - (n_adc & 1);
    z.n_up = z.n_dig * (adc.upsampling >= 2 ? adc.upsampling : 1);
    return z;
}
// stage -> (n_in, Q, sgn, P, n_out); returns log2 of the convolution length: the smallest power of two that takes the whole
// transform in one block of outputs (n_in + n_out - 1 points), FFT_MAX with several blocks of P outputs beyond that.
// Stages 6 and 7 (clock offset only, between 1 and 2): the delayed trace's samples cyc .. L0 - 1 from the phase-ramped spectrum
// (n_in = L0 / 2 + 1, Q = L0, sgn +, output k is sample cyc + k), then the spectrum of those L samples (n_in = L, Q = L, sgn -).
__device__ __host__ inline int pa_stage(const PaSizes& z, int stage, int* n_in, long* Q, double* sgn, int* P, long* n_out)
{
    if (stage == 1) { *n_in = z.L0; *Q = z.L0; *sgn = -1.; *n_out = z.m0 + 1; }
    else if (stage == 2) { *n_in = z.m + 1; *Q = z.num2; *sgn = 1.; *n_out = z.len5; }
    else if (stage == 4) { *n_in = z.n_dig; *Q = z.n_dig; *sgn = -1.; *n_out = z.n_dig / 2 + 1; }
    else if (stage == 5) { *n_in = z.n_dig / 2 + 1; *Q = z.n_up; *sgn = 1.; *n_out = z.n_up; }
    else if (stage == 6) { *n_in = z.m0 + 1; *Q = z.L0; *sgn = 1.; *n_out = z.L; }
    else { *n_in = z.L; *Q = z.L; *sgn = -1.; *n_out = z.m + 1; }
    int log2m = FFT_LOG2_MAX;
    if (*n_in + *n_out - 1 <= FFT_MAX) {
        log2m = 11;   // (the transform pair of conv_fft.h comes in 2048, 4096 and 8192 points)
        while ((1l << log2m) < *n_in + *n_out - 1) log2m++;
    }
    *P = (1 << log2m) - *n_in + 1;
    return log2m;
}
// table row of a stage's transform: 1, 2, 4, 5, 6, 7 -> 0 .. 5
__device__ __host__ inline int pa_stage_row(int stage) { return stage < 3 ? stage - 1 : stage - 2; }

struct PaWork {
    double2* X;      // [item][xs]   trace spectra
    double* x5;      // [item][s5]   5 GHz samples
    double* d;       // [item][sd]   ADC traces
    double2* D;      // [item][sd / 2 + 1]
    int xs, s5, sd;
};

// Tables of the lengths in `lens` (slots of the station's table cache): [slot][PA_TABLES][FFT_MAX] -- rows 0..5 the Bluestein
// spectra of the transforms (2^log2m points, natural bin order: conv_mid_plain), rows 6..11 their chirps exp(sgn i pi j^2 / Q), j < 2^log2m (the
// factor of every input and every output sample: two sincospi per point and transform when evaluated in place)
__global__ void __launch_bounds__(1024)
pa_tables_kernel(int n_len, const int* __restrict__ lens, const int* __restrict__ slots, double fs, PaAdc adc,
                 const double2* __restrict__ tw, double2* __restrict__ Btab)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    const int il = blockIdx.x, st = blockIdx.y;   // table row 0..5 -> stages 1, 2, 4, 5, 6, 7
    if (il >= n_len) return;
    if (st >= 4 && !adc.clock_offset) return;
    const PaSizes z = pa_sizes(lens[il], fs, adc);
    int n_in, P;
    long Q, n_out;
    double sgn;
    const int log2m = pa_stage(z, st < 2 ? st + 1 : st + 2, &n_in, &Q, &sgn, &P, &n_out);
    if (P < 1) return;
    czt_build_table(x, log2m, n_in, P, Q, sgn, tw);
    double2* B = Btab + ((long)slots[il] * PA_TABLES + st) * FFT_MAX;
    double2* C = B + (long)(PA_TABLES / 2) * FFT_MAX;
    for (int i = threadIdx.x; i < (1 << log2m); i += blockDim.x) {
        B[bitrev(i, log2m)] = x[i];
        C[i] = chirp(i, Q, sgn);
    }
}

// One transform of the chain: block (item, block of P outputs); 512 threads, the buffer in conv_fft.h's layout.  The convolution is
// the transform pair of channel_conv_kernel (wave-private 1024-point sub-transforms, conv_fft.h) with a plain spectrum product in
// the middle pass: 2^log2m / 16 threads carry it, all 512 fill the buffer and read it out.
template <int LOG2M>
__device__ __forceinline__ void pa_convolve(const double2* __restrict__ Bn, const double2* __restrict__ tw, const double2* __restrict__ cft,
                                            bool full)
{
    if (full) conv_fwd<LOG2M, 512, true>(tw, cft, 1 << LOG2M);
    else conv_fwd<LOG2M, 512, false>(tw, cft, 1 << LOG2M);
    conv_mid_plain<LOG2M, 512>(Bn);
    conv_inv<LOG2M, 512>(tw, cft);
}

template <int STAGE>
__global__ void __launch_bounds__(512)
pa_czt_stage_kernel(int item0, int n_cand, const int* __restrict__ item_event, int n_ch, const int* __restrict__ ev_L,
                    const int* __restrict__ slotmap, const double* __restrict__ trace, const long* __restrict__ trace_offset, int n_pa,
                    const int* __restrict__ pa_channel, double fs, PaAdc adc, const double2* __restrict__ tw,
                    const double2* __restrict__ cft, const double2* __restrict__ Btab, PaWork wk, double* __restrict__ pa_trace,
                    unsigned long long* __restrict__ conv_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    constexpr int stage = STAGE, NT = 512, U = FFT_MAX / NT;
    const int item = item0 + blockIdx.x, n_items = n_cand * n_pa, wi = blockIdx.x;   // wi: row of the chunk's work arrays
    if (item >= n_items) return;
    const int ic = item / n_pa, c = item % n_pa;
    const int e = item_event[ic], L = ev_L[e];
    const PaSizes z = pa_sizes(L, fs, adc);
    int n_in, P;
    long Q, n_out;
    double sgn;
    const int log2m = pa_stage(z, stage, &n_in, &Q, &sgn, &P, &n_out), M = 1 << log2m;
    const long k0 = (long)blockIdx.y * P;
    if (k0 >= n_out) return;
    // FP64 operations of the convolution: transform pair 2 x 5 M log2 M, three complex products per point (bench.py's roofline)
    if (threadIdx.x == 0 && conv_count) atomicAdd(conv_count, (unsigned long long)M * (unsigned long long)(10 * log2m + 18));
    const int st = pa_stage_row(stage);
    const double2* B = Btab + ((long)slotmap[L / 2] * PA_TABLES + st) * FFT_MAX;
    const double2* C = B + (long)(PA_TABLES / 2) * FFT_MAX;
    const double* tr = trace + trace_offset[(long)ic * n_ch + pa_channel[c]];
    const double2* Xi = wk.X + (long)wi * wk.xs;
    const double* di = wk.d + (long)wi * wk.sd;
    const double2* Di = wk.D + (long)wi * (wk.sd / 2 + 1);
    const double* x5i = wk.x5 + (long)wi * wk.s5;
    const double delay_bins = (adc.clock_offset / adc.adc_fs) * fs;   // the clock offset in samples of the trace
    // the inputs of a thread are 512 apart: with several blocks of outputs their factors exp(sgn 2 pi i j k0 / Q) follow from the
    // first one by a constant rotation (two sincospi per thread instead of one per point; <= 16 steps of the recurrence)
    double2 sh = make_double2(1., 0.), sh_step = sh;
    const long ks = k0 + (stage == 6 ? z.cyc : 0);   // first output index of this block (stage 6 starts behind the wrapped samples)
    if (ks) {
        auto turn = [&](long j) {
            double sn, cs;
            sincospi(2. * (((double)j * (double)ks < 4.5e15 && Q < (1l << 30)) ? mod_exact((double)j * (double)ks, (double)Q) : (double)(((long long)j * ks) % Q)) / (double)Q, &sn, &cs);
            return make_double2(cs, sgn * sn);
        };
        sh = turn(threadIdx.x);
        sh_step = turn(NT);
    }
    const bool full = n_in > M / 2;
    const int n_fill = full ? M : M / 2;   // (the forward transform does not read the upper half unless told to)
    {
        // all loads of the thread first (one round trip to HBM / L2 for the block instead of one per 512 points), then the products
        double2 vin[U], cin[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = threadIdx.x + u * NT;
            double2 v = make_double2(0., 0.), cc = v;
            if (j < n_in) {
                if (stage == 1) v = make_double2(tr[j], 0.);
                else if (stage == 2 || stage == 6) v = Xi[j];
                else if (stage == 4) v = make_double2(di[j], 0.);
                else if (stage == 5) v = Di[j];
                else v = make_double2(x5i[j], 0.);   // stage 7: the delayed samples
                cc = C[j];
            }
            vin[u] = v;
            cin[u] = cc;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = threadIdx.x + u * NT;
            if (j >= n_fill) break;
            double2 v = vin[u];
            if (stage == 2) v = cscale(v, ((j == 0 || j == z.m) ? 1. : 2.) / z.L);
            else if (stage == 5) v = cscale(v, ((j == 0 || j == z.n_dig / 2) ? 1. : 2.) / z.n_dig);
            else if (stage == 6) {   // irfft weights and the delay's phase ramp exp(-2 pi i f_j delay), f_j = j fs / L0
                double sn, cs;
                sincospi(-2. * ((double)j * delay_bins / z.L0), &sn, &cs);
                v = cmul(cscale(v, ((j == 0 || j == z.m0) ? 1. : 2.) / z.L0), make_double2(cs, sn));
            }
            v = cmul(v, cin[u]);
            if (ks) { v = cmul(v, sh); sh = cmul(sh, sh_step); }   // outputs ks .. : in[j] exp(sgn 2 pi i j ks / Q)
            x[conv_pad(j)] = v;
        }
    }
    __syncthreads();
    if (log2m == 13) pa_convolve<13>(B, tw, cft, full);
    else if (log2m == 12) pa_convolve<12>(B, tw, cft, full);
    else pa_convolve<11>(B, tw, cft, full);
    const int cnt = (int)((n_out - k0 < P) ? n_out - k0 : P);
    const double inv_m = 1.0 / M;
    {
        double2 cout[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = threadIdx.x + u * NT;
            cout[u] = (k < cnt) ? C[k] : make_double2(0., 0.);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = threadIdx.x + u * NT;
            if (k >= cnt) break;
            // out[k0 + k] = chirp(k) x[k] / M  (the chirp belongs to the shifted problem: index k)
            const double2 o = cscale(cmul(x[conv_pad(k)], cout[u]), inv_m);
            if (stage == 1 || stage == 7) wk.X[(long)wi * wk.xs + k0 + k] = o;
            else if (stage == 2 || stage == 6) wk.x5[(long)wi * wk.s5 + k0 + k] = o.x;
            else if (stage == 4) wk.D[(long)wi * (wk.sd / 2 + 1) + k0 + k] = o;
            else pa_trace[(long)item * adc.stride + k0 + k] = adc.counts ? rint(o.x) : o.x;
        }
    }
}

// stage 3: the ADC samples from the 5 GHz samples (downsampling_linear_interpolation + perfect_floor_comparator)
__global__ void __launch_bounds__(256)
pa_adc_sample_kernel(int item0, int n_cand, const int* __restrict__ item_event, const int* __restrict__ ev_L, int n_pa, double fs,
                     PaAdc adc, PaWork wk, double* __restrict__ pa_trace, int* __restrict__ pa_len)
{
#pragma clang fp contract(off)   // counts are compared sample by sample with the CPU restatement: no fused rounding here
    const int item = item0 + blockIdx.x, n_items = n_cand * n_pa, wi = blockIdx.x;
    if (item >= n_items) return;
    const int e = item_event[item / n_pa];
    const PaSizes z = pa_sizes(ev_L[e], fs, adc);
    const double* x5 = wk.x5 + (long)wi * wk.s5;
    const double lsb = (adc.vmax - adc.vmin) / (double)((1 << adc.n_bits) - 1);
    const double vmin_adc = floor(adc.vmin / lsb);
    for (int j = threadIdx.x; j < z.n_dig; j += blockDim.x) {
        const double tn = (double)j / adc.adc_fs;
        long lo = (long)floor(tn * 5.0);
        while (lo > 0 && (double)lo / 5.0 >= tn) lo--;          // times[lo] < t <= times[lo + 1] (searchsorted, side 'left')
        while ((double)(lo + 1) / 5.0 < tn) lo++;
        if (lo > z.len5 - 2) lo = z.len5 - 2;
        if (lo < 0) lo = 0;
        const double xlo = (double)lo / 5.0, xhi = (double)(lo + 1) / 5.0;
        const double ylo = x5[lo], yhi = x5[lo + 1];
        const double slope = (yhi - ylo) / (xhi - xlo);
        const double v = slope * (tn - xlo) + ylo;
        double cnt = floor((v - adc.vmin) / lsb);
        cnt = fmin(fmax(cnt, 0.), (double)((1 << adc.n_bits) - 1)) + vmin_adc;
        const double dv = adc.counts ? cnt : lsb * cnt;
        wk.d[(long)wi * wk.sd + j] = dv;
        if (adc.upsampling < 2) pa_trace[(long)item * adc.stride + j] = dv;
    }
    if (threadIdx.x == 0) pa_len[item] = z.n_up - (z.n_up & 1);
}

// 'lin' and 'fir' up-sampling of the ADC traces (signal_processing.digital_upsampling :163-190, upsampling_fir :192-234); one block
// per (candidate event, array channel).  'lin': np.interp of the trace on cur_t = arange(0, n / f, 1 / f) at
// new_t = arange(0, n / f, 1 / (f up)) (numpy's lengths ceil(stop / step), values i * step; beyond the last node the last value);
// 'fir': out[i] = up sum_j zp[j] h[i + off - j] with the zero-stuffed trace zp[j up] = d[j], off = len(h) // 2 - 1 (np.convolve
// 'full', sliced).  ADC counts are rounded (the sums are exact: coefficients are multiples of 1 / coeff_gain); an odd length
// loses its last sample.
__global__ void __launch_bounds__(256)
pa_upsample_kernel(int n_items, PaAdc adc, const double* __restrict__ adc_trace, int stride_in, const int* __restrict__ len_in,
                   double* __restrict__ pa_trace, int* __restrict__ pa_len)
{
#pragma clang fp contract(off)   // counts are compared sample by sample with the CPU restatement: no fused rounding here
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const double* d = adc_trace + (long)item * stride_in;
        double* out = pa_trace + (long)item * adc.stride;
        const int n = len_in[item], up = adc.upsampling;
        int n_new;
        bool round_out = adc.counts != 0;   // "is_digital_trace" of digital_upsampling (:153), see pa_digitize_kernel
        if (adc.n_bits == 0) {
            int ok = 1;
            for (int j = threadIdx.x; j < n; j += blockDim.x) {
                const double r = rint(d[j]);
                ok &= (fabs(d[j] - r) <= 1e-8 + 1e-5 * fabs(r)) ? 1 : 0;
            }
            round_out = __syncthreads_and(ok) != 0;
        }
        if (adc.up_method == 1) {
            const double dt = 1 / adc.adc_fs, stop = dt * n, dtn = 1 / (adc.adc_fs * up);
            n_new = (int)ceil(stop / dtn);
            if (n_new > adc.stride) n_new = adc.stride;
            for (int i = threadIdx.x; i < n_new; i += blockDim.x) {
                const double x = i * dtn;
                int j = (int)floor(x * adc.adc_fs);
                j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
                while (j > 0 && j * dt > x) j--;
                while (j < n - 1 && (j + 1) * dt <= x) j++;
                double v;
                if (x > (n - 1) * dt || j == n - 1) v = d[n - 1];
                else if (j * dt == x) v = d[j];
                else {
                    const double slope = (d[j + 1] - d[j]) / ((j + 1) * dt - j * dt);
                    v = slope * (x - j * dt) + d[j];
                }
                out[i] = round_out ? rint(v) : v;
            }
        } else {
            n_new = n * up;
            const int nh = adc.n_up_taps, off = nh / 2 - 1;
            for (int i = threadIdx.x; i < n_new; i += blockDim.x) {
                // zp index m = j up contributes h[i + off - m]: 0 <= i + off - m < nh
                int j_lo = (i + off - (nh - 1) + up - 1);
                j_lo = j_lo > 0 ? j_lo / up : 0;
                int j_hi = (i + off) / up;
                if (i + off < 0) j_hi = -1;
                if (j_hi > n - 1) j_hi = n - 1;
                double acc = 0.;
                for (int j = j_lo; j <= j_hi; j++) acc += d[j] * adc.up_taps[i + off - j * up];
                acc *= up;
                out[i] = round_out ? rint(acc) : acc;
            }
        }
        if (threadIdx.x == 0) pa_len[item] = n_new - (n_new & 1);
    }
}

void launch_pa_upsample(hipStream_t s, int n_items, const PaAdc& adc, const double* adc_trace, int stride_in, const int* len_in,
                        double* pa_trace, int* pa_len)
{
    if (n_items <= 0) return;
    hipLaunchKernelGGL(pa_upsample_kernel, dim3(n_items < 8192 ? n_items : 8192), dim3(256), 0, s, n_items, adc, adc_trace, stride_in,
                       len_in, pa_trace, pa_len);
}

// beams and power windows on the digitised, up-sampled traces (phase_signals with the saturation of ADC counts :183-215,
// power_sum with its rounding :217-271, the decision of phased_trigger :455-496)
__global__ void __launch_bounds__(256)
phased_array_digital_kernel(int n_cand, const int* __restrict__ item_event, const double* __restrict__ pa_trace,
                            const int* __restrict__ pa_len, int n_pa, int n_beams, const int* __restrict__ rolls, int window, int step,
                            double divisor, double threshold, PaAdc adc, unsigned char* __restrict__ triggered, double* __restrict__ pa_max)
{
    extern __shared__ double coh[];
    __shared__ double red[256];
    for (int i = blockIdx.x; i < n_cand; i += gridDim.x) {
        const int e = item_event[i], Lu = pa_len[(long)i * n_pa];
        const int n_frames = (Lu - window) / step > 0 ? (Lu - window) / step : 0;
        const double hi = (double)((1 << (adc.saturation_bits - 1)) - 1), lo = -(double)(1 << (adc.saturation_bits - 1));
        double any = 0.;
        for (int b = 0; b < n_beams; b++) {
            __syncthreads();
            for (int n = threadIdx.x; n < Lu; n += blockDim.x) {
                double sum = 0.;
                for (int c = 0; c < n_pa; c++) {
                    int k = (n - rolls[b * n_pa + c]) % Lu;
                    if (k < 0) k += Lu;
                    sum += pa_trace[((long)i * n_pa + c) * adc.stride + k];
                }
                if (adc.counts && adc.saturation_bits > 0) sum = fmin(fmax(sum, lo), hi);
                coh[n] = sum;
            }
            __syncthreads();
            double mx = -INFINITY;
            if (adc.mode == 2) {
                // hilbert_envelope with ideal_transformer = True (:339-345): imag(scipy.signal.hilbert(coh)) is the circular
                // convolution of coh with g = imag(ifft(h)), h the one-sided spectrum weights; for an even length (pa_len always is)
                // g[m] = (2 / n) cot(pi m / n) for odd m and 0 for even m (sum_{k=1}^{n/2-1} sin(2 pi k m / n) in closed form).
                // g's odd entries go to LDS behind the beam, then n^2 / 2 multiply-adds per beam; envelope sqrt(x^2 + im^2)
                // (g is odd about n / 2: cot(pi (n - m) / n) = -cot(pi m / n), so the odd m <= n / 2 are kept: n / 4 + 1 values)
                double* g = coh + Lu;
                const int n_g = Lu / 4 + 1;
                for (int j = threadIdx.x; j < n_g; j += blockDim.x) {
                    double sn, cs;
                    sincospi((double)(2 * j + 1) / (double)Lu, &sn, &cs);
                    g[j] = (2. / Lu) * (cs / sn);
                }
                __syncthreads();
                for (int n = threadIdx.x; n < Lu; n += blockDim.x) {
                    double im = 0.;
                    // s runs over the samples of the other parity; d = (n - s) mod Lu is odd
                    int d = (n & 1) ? n : n - 1 + ((n == 0) ? Lu : 0);   // s = 0 (n odd) or s = 1 (n even)
                    for (int s2 = (n & 1) ? 0 : 1; s2 < Lu; s2 += 2) {
                        const double gv = (2 * d <= Lu) ? g[d >> 1] : -g[(Lu - d) >> 1];
                        im += coh[s2] * gv;
                        d -= 2;
                        if (d < 0) d += Lu;
                    }
                    if (adc.counts) im = rint(im);
                    const double c = coh[n];
                    double env = sqrt(c * c + im * im);
                    if (adc.counts) env = rint(env);
                    mx = fmax(mx, env);
                }
            } else if (adc.mode == 1) {
                // hilbert_envelope (:337-367): imaginary part by the FIR transformer (np.convolve 'full', centred), the magnitude
                // estimate max + 3/8 min of the two SIGNED sequences as the reference writes it, rounded for counts
                const int nh = adc.n_hil_taps, half = nh / 2;
                for (int n = threadIdx.x; n < Lu; n += blockDim.x) {
                    const int j_lo = max(0, n + half - (nh - 1)), j_hi = min(Lu - 1, n + half);
                    double im = 0.;
                    for (int j = j_lo; j <= j_hi; j++) im += coh[j] * adc.hil_taps[n + half - j];
                    if (adc.counts) im = rint(im);
                    const double c = coh[n];
                    double env = fmax(c, im) + (3. / 8.) * fmin(c, im);
                    if (adc.counts) env = rint(env);
                    mx = fmax(mx, env);
                }
            } else
            for (int f = threadIdx.x; f < n_frames; f += blockDim.x) {
                double p = 0.;
                for (int j = 0; j < window; j++) {
                    const double v = coh[f * step + j];
                    p += v * v;
                }
                p /= divisor;
                if (adc.counts) p = rint(p);
                mx = fmax(mx, p);
            }
            mx = block_max(mx, red);
            if (threadIdx.x == 0) pa_max[(long)i * n_beams + b] = mx;
            if (mx > (adc.counts ? trunc(threshold) : threshold)) any = 1.;
        }
        if (threadIdx.x == 0 && any > 0.) triggered[e] = 1;
    }
}

void launch_phased_array_beams(hipStream_t s, int n_cand, const int* item_event, int n_pa, int n_beams, const int* rolls_up, int window,
                               int step, double divisor, double threshold, const PaAdc& adc, const double* pa_trace, const int* pa_len,
                               unsigned char* triggered, double* pa_max)
{
    if (n_cand <= 0) return;
    set_big_lds();
    const size_t lds = (size_t)adc.stride * (adc.mode == 2 ? 10 : 8) + 128;   // the beam (+ a quarter of a beam of Hilbert kernel values)
    (void)hipFuncSetAttribute((const void*)phased_array_digital_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(phased_array_digital_kernel, dim3(n_cand < 4096 ? n_cand : 4096), dim3(256), lds, s, n_cand,
                       item_event, pa_trace, pa_len, n_pa, n_beams, rolls_up, window, step, divisor, threshold, adc, triggered, pa_max);
}

void launch_phased_array_digital(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                                 const long* trace_offset, int n_pa, const int* pa_channel, int n_beams, const int* rolls_up, int window,
                                 int step, double divisor, double threshold, int max_length, double fs, const PaAdc& adc,
                                 double* pa_trace, int* pa_len, unsigned char* triggered, double* pa_max, bool with_beams)
{
    if (n_cand <= 0) return;
    set_big_lds();
    const size_t lds1 = (size_t)(2 * max_length + 16) * 8 + 64;
    (void)hipFuncSetAttribute((const void*)pa_digitize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    const int n_items = n_cand * n_pa;
    hipLaunchKernelGGL(pa_digitize_kernel, dim3(n_items < 4096 ? n_items : 4096), dim3(1024), lds1, s, n_cand, item_event, n_ch, ev_L, trace,
                       trace_offset, n_pa, pa_channel, fs, adc, pa_trace, pa_len);
    if (with_beams)
        launch_phased_array_beams(s, n_cand, item_event, n_pa, n_beams, rolls_up, window, step, divisor, threshold, adc, pa_trace, pa_len,
                                  triggered, pa_max);
}

// whether the chirp-z digitiser takes this configuration: resampling through 5 GHz, transforms with at least 1024 outputs per block
bool pa_czt_applies(int max_length, double fs, const PaAdc& adc)
{
    const bool resampled = !(fabs(adc.adc_fs - fs) <= 1e-8 + 1e-5 * fabs(fs));
    if (!resampled || !(5.0 > fs)) return false;
    const PaSizes z = pa_sizes(max_length, fs, adc);
    return max_length <= FFT_MAX - 1023 && z.n_dig <= FFT_MAX - 1023 && z.n_dig >= 2;
}

size_t pa_czt_work_bytes(int max_length, double fs, const PaAdc& adc, int chunk)
{
    const PaSizes z = pa_sizes(max_length, fs, adc);
    const size_t xs = z.m0 + 2, s5 = std::max<long>(z.num2, z.L) + 2, sd = z.n_dig + 4;
    return (size_t)chunk * (xs * 16 + s5 * 8 + sd * 8 + (sd / 2 + 1) * 16) + 256;
}

void launch_pa_czt_tables(hipStream_t s, int n_len, const int* lens, const int* slots, double fs, const PaAdc& adc, const double2* tw,
                          double2* Btab)
{
    if (n_len <= 0) return;
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)pa_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    hipLaunchKernelGGL(pa_tables_kernel, dim3(n_len, PA_TABLES / 2), dim3(1024), (size_t)FFT_MAX * 16, s, n_len, lens, slots, fs, adc, tw, Btab);
}

void launch_phased_array_digital_czt(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const int* slotmap,
                                     const double* trace, const long* trace_offset, int n_pa, const int* pa_channel, int n_beams,
                                     const int* rolls_up, int window, int step, double divisor, double threshold, int max_length, double fs,
                                     const PaAdc& adc, const double2* tw, const double2* cft, const double2* Btab, void* work, int chunk, double* pa_trace,
                                     int* pa_len, unsigned char* triggered, double* pa_max, bool with_beams,
                                     unsigned long long* conv_count)
{
    if (n_cand <= 0) return;
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    (void)hipFuncSetAttribute((const void*)pa_czt_stage_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_elems(FFT_MAX) * 16);
    const PaSizes z = pa_sizes(max_length, fs, adc);
    PaWork wk;
    wk.xs = z.m0 + 2; wk.s5 = (int)std::max<long>(z.num2, z.L) + 2; wk.sd = z.n_dig + 4;
    unsigned char* w = (unsigned char*)work;
    wk.X = (double2*)w;                 w += (size_t)chunk * wk.xs * 16;
    wk.D = (double2*)w;                 w += (size_t)chunk * (wk.sd / 2 + 1) * 16;
    wk.x5 = (double*)w;                 w += (size_t)chunk * wk.s5 * 8;
    wk.d = (double*)w;
    const int n_items = n_cand * n_pa;
    for (int item0 = 0; item0 < n_items; item0 += chunk) {
        const unsigned nb = (unsigned)std::min(chunk, n_items - item0);
        const int order_plain[5] = {1, 2, 3, 4, 5}, order_clock[7] = {1, 6, 7, 2, 3, 4, 5};
        const int n_order = adc.clock_offset ? 7 : 5;
        for (int io = 0; io < n_order; io++) {
            const int stage = adc.clock_offset ? order_clock[io] : order_plain[io];
            if (stage == 3) {
                hipLaunchKernelGGL(pa_adc_sample_kernel, dim3(nb), dim3(256), 0, s, item0, n_cand, item_event, ev_L, n_pa, fs, adc, wk,
                                   pa_trace, pa_len);
                if (adc.upsampling < 2) break;
                continue;
            }
            // output blocks, convolution length and threads of the longest trace (n_in + n_out grows with the length: it needs the most)
            int n_in, P; long Q, n_out; double sgn;
            const int log2m = pa_stage(z, stage, &n_in, &Q, &sgn, &P, &n_out), M = 1 << log2m;
            const unsigned by = (unsigned)((n_out + P - 1) / P);
            auto kern = stage == 1 ? pa_czt_stage_kernel<1> : stage == 2 ? pa_czt_stage_kernel<2> : stage == 4 ? pa_czt_stage_kernel<4>
                        : stage == 5 ? pa_czt_stage_kernel<5> : stage == 6 ? pa_czt_stage_kernel<6> : pa_czt_stage_kernel<7>;
            hipLaunchKernelGGL(kern, dim3(nb, by), dim3(512), (size_t)conv_lds_elems(M) * 16, s, item0, n_cand, item_event, n_ch, ev_L,
                               slotmap, trace, trace_offset, n_pa, pa_channel, fs, adc, tw, cft, Btab, wk, pa_trace, conv_count);
        }
    }
    if (with_beams)
        launch_phased_array_beams(s, n_cand, item_event, n_pa, n_beams, rolls_up, window, step, divisor, threshold, adc, pa_trace, pa_len,
                                  triggered, pa_max);
}

// ---------------------------------------------------------------------------------------------------------
// kernel: high/low and n-fold coincidence triggers (highLowThreshold.py:13-150, simpleThreshold.py) on channel traces that
// sit in HBM -- the stations / events the fused logic of channel_conv_kernel cannot take: common traces longer than FFT_MAX
// samples (bottom reflections: up to 16 128) and tabulated antenna patterns, whose traces come out of the chirp-z
// channel_kernel.  One block per candidate event; per channel the flags, their OR-dilation over the coincidence window (running
// maximum of the index of the last raised flag) and the per-sample channel count, then the majority -- the very steps of
// channel_conv_kernel's coincidence branch.  LDS: 2 x max_length ints.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
trace_trigger_kernel(int n_cand, const int* __restrict__ item_event, int n_ch, const int* __restrict__ ev_L,
                     const double* __restrict__ trace, const long* __restrict__ trace_offset, TriggerDev trg,
                     const unsigned char* __restrict__ trig_on, unsigned char* __restrict__ triggered, int* __restrict__ trigger_bin)
{
    extern __shared__ int tt_lds[];
    __shared__ int s_scan[256];
    __shared__ int s_first;
    for (int ic = blockIdx.x; ic < n_cand; ic += gridDim.x) {
        const int e = item_event[ic], L = ev_L[e];
        int* A = tt_lds;
        int* cnt = tt_lds + L;
        const int nb = (trg.type == 1) ? L - 1 : L;
        __syncthreads();
        for (int i = threadIdx.x; i < L; i += blockDim.x) cnt[i] = 0;
        if (threadIdx.x == 0) s_first = 0x7fffffff;
        __syncthreads();
        for (int ch = 0; ch < n_ch; ch++) {
            if (trig_on && !trig_on[ch]) continue;   // triggered_channels of the reference's trigger modules
            const double* V = trace + trace_offset[(long)ic * n_ch + ch];
            for (int i = threadIdx.x; i < nb; i += blockDim.x) {
                bool flag;
                if (trg.type == 0) {
                    flag = fabs(V[i]) >= trg.threshold;
                } else if (trg.type == 2) {   // V is the Hilbert envelope of the band-passed trace (envelopeTrigger.py:31: strictly above)
                    flag = V[i] > trg.threshold;
                } else {
                    bool hi = false, lo = false;
                    for (int k = max(0, i - trg.w_hl + 1); k <= i; k++) {
                        hi = hi || (V[k] >= trg.high);
                        lo = lo || (V[k] <= trg.low);
                    }
                    if (i - trg.w_hl + 1 < 0) {  // the reference pads with zeros in front
                        hi = hi || (0. >= trg.high);
                        lo = lo || (0. <= trg.low);
                    }
                    flag = hi && lo;
                }
                A[i] = flag ? i : -1;
            }
            __syncthreads();
            {   // inclusive running maximum of A[0 .. nb)
                const int chunk = (nb + 255) / 256, b0 = threadIdx.x * chunk, b1 = min(b0 + chunk, nb);
                int run = -1;
                for (int i = b0; i < b1; i++) { run = max(run, A[i]); A[i] = run; }
                s_scan[threadIdx.x] = run;
                __syncthreads();
                for (int off = 1; off < 256; off <<= 1) {
                    int v = ((int)threadIdx.x >= off) ? s_scan[threadIdx.x - off] : -1;
                    __syncthreads();
                    s_scan[threadIdx.x] = max(s_scan[threadIdx.x], v);
                    __syncthreads();
                }
                const int before = threadIdx.x > 0 ? s_scan[threadIdx.x - 1] : -1;
                for (int i = b0; i < b1; i++) A[i] = max(A[i], before);
                __syncthreads();
            }
            const int wc = min(trg.w_coinc, nb);
            for (int i = threadIdx.x; i < nb - 1; i += blockDim.x)
                if (A[i] >= 0 && i - A[i] < wc) cnt[i] += 1;
            __syncthreads();
        }
        int first = 0x7fffffff;
        for (int i = threadIdx.x; i < nb - 1; i += blockDim.x)
            if (cnt[i] >= trg.n_coinc) first = min(first, i);
        if (first != 0x7fffffff) atomicMin(&s_first, first);
        __syncthreads();
        if (threadIdx.x == 0 && s_first != 0x7fffffff) {
            triggered[e] = 1;
            trigger_bin[e] = s_first;
        }
    }
}

void launch_trace_trigger(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                          const long* trace_offset, const TriggerDev& trg, const unsigned char* trig_on, int max_length,
                          unsigned char* triggered, int* trigger_bin)
{
    if (n_cand <= 0) return;
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)trace_trigger_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * FFT_MAX * 4 + 1024);
    int grid = n_cand < 256 * 16 ? n_cand : 256 * 16;
    hipLaunchKernelGGL(trace_trigger_kernel, dim3(grid), dim3(256), (size_t)2 * max_length * 4, s, n_cand, item_event, n_ch, ev_L, trace,
                       trace_offset, trg, trig_on, triggered, trigger_bin);
}

void launch_ray_envelope(hipStream_t s, int n_cand_max, const int* n_cand, const int* item_event, const RayWork& w,
                         const EventOut& ev, const StationDev& st, int ask_model, const double2* tw, const LengthTables& tab,
                         const int* len_index_N, double* max_env, double* signal_time, const double2* spec, double2* tab_nodes,
                         double* amp_scratch)
{
    if (n_cand_max <= 0) return;
    set_big_lds();
    size_t lds = (size_t)(st.np.log2nh >= 0 ? st.N : nplan_points(st.np)) * 16 + (amp_scratch ? 0 : (size_t)(st.N / 2 + 1) * 8);
    int grid = n_cand_max < 256 * 4 ? n_cand_max : 256 * 4;
    if (amp_scratch && grid > RAY_AMP_ROWS) grid = RAY_AMP_ROWS;
    if (tab_nodes && grid > channel_grid_blocks()) grid = channel_grid_blocks();   // (rows of the node scratch)
    hipLaunchKernelGGL(ray_envelope_kernel, dim3(grid), dim3(256), lds, s, n_cand, item_event, w, ev, st, ask_model, tw, tab,
                       len_index_N, ilog2(st.N), max_env, signal_time, spec, tab_nodes, amp_scratch);
}
// ---------------------------------------------------------------------------------------------------------
// kernel: what the output writer keeps of a triggered station event's channel traces (the dumped traces never leave the device):
// the trigger bin of the simple threshold on any channel -- first sample of the first L - 1 (get_majority_logic drops the last one,
// highLowThreshold.py:82-150) with |V| >= threshold on some channel --, then per channel the read-out window the reference cuts
// (channelReadoutWindowCutter.run :28-137: the trace rolled so that the window of n_window samples starts pre_bins before the
// trigger) and of that window max |V| and the maximum of its Hilbert envelope |scipy.signal.hilbert(w)|
// (channelSignalReconstructor -> output_writer_hdf5.py:215-320 "maximum_amplitudes", "maximum_amplitudes_envelope").
// One block (256) per item; n_window a power of two: forward transform of the window (imaginary part 0), the one-sided weights
// (1, 2 ... 2, 1, 0 ... 0), inverse transform, modulus.  LDS: n_window complex.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
readout_window_kernel(int n_items, int n_ch, const int* __restrict__ item_event, const int* __restrict__ ev_L,
                      const double* __restrict__ trace, const long* __restrict__ trace_offset, int n_window, int log2n, int pre_bins,
                      double threshold, const double2* __restrict__ tw, int* __restrict__ trigger_bin, double* __restrict__ max_amp,
                      double* __restrict__ max_env)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* x = (double2*)smem;
    __shared__ double red[256];
    __shared__ int s_first;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int L = ev_L[item_event[item]];
        if (threadIdx.x == 0) s_first = 0x7fffffff;
        __syncthreads();
        int first = 0x7fffffff;
        for (int c = 0; c < n_ch; c++) {
            const double* V = trace + trace_offset[(long)item * n_ch + c];
            for (int n = threadIdx.x; n < L - 1 && n < first; n += blockDim.x)
                if (fabs(V[n]) >= threshold) first = n;
        }
        if (first != 0x7fffffff) atomicMin(&s_first, first);
        __syncthreads();
        const int tbin = (s_first == 0x7fffffff) ? -1 : s_first;
        if (threadIdx.x == 0) trigger_bin[item] = tbin;
        const int nw = min(n_window, L);
        if (tbin < 0 || nw != n_window) {   // no sample reaches the threshold / a common trace shorter than the window: the caller's business
            for (int c = threadIdx.x; c < n_ch; c += blockDim.x) max_amp[(long)item * n_ch + c] = max_env[(long)item * n_ch + c] = NAN;
            __syncthreads();
            continue;
        }
        int s0 = (tbin - pre_bins) % L;
        if (s0 < 0) s0 += L;
        for (int c = 0; c < n_ch; c++) {
            const double* V = trace + trace_offset[(long)item * n_ch + c];
            double am = 0.;
            for (int j = threadIdx.x; j < n_window; j += blockDim.x) {
                int q = s0 + j;
                if (q >= L) q -= L;
                const double v = V[q];
                am = fmax(am, fabs(v));
                x[j] = make_double2(v, 0.);
            }
            __syncthreads();
            fft_dif(x, log2n, tw, false);   // natural -> bit-reversed
            for (int k = threadIdx.x; k < n_window; k += blockDim.x) {
                const double wgt = (k == 0 || k == n_window / 2) ? 1. : (k < n_window / 2 ? 2. : 0.);
                const int q = bitrev(k, log2n);
                x[q] = make_double2(x[q].x * wgt, x[q].y * wgt);
            }
            __syncthreads();
            fft_dit(x, log2n, tw, true);    // bit-reversed -> natural, unscaled
            double em = 0.;
            for (int j = threadIdx.x; j < n_window; j += blockDim.x) em = fmax(em, cabs2(x[j]));
            am = block_max(am, red);
            em = block_max(em, red);
            if (threadIdx.x == 0) {
                max_amp[(long)item * n_ch + c] = am;
                max_env[(long)item * n_ch + c] = em * (1.0 / n_window);
            }
            __syncthreads();
        }
    }
}

void launch_readout_windows(hipStream_t s, int n_items, int n_ch, const int* item_event, const int* ev_L, const double* trace,
                            const long* trace_offset, int n_window, int pre_bins, double threshold, const double2* tw,
                            int* trigger_bin, double* max_amp, double* max_env)
{
    if (n_items <= 0) return;
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)readout_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    hipLaunchKernelGGL(readout_window_kernel, dim3(n_items < 2048 ? n_items : 2048), dim3(256), (size_t)n_window * 16, s, n_items, n_ch,
                       item_event, ev_L, trace, trace_offset, n_window, ilog2(n_window), pre_bins, threshold, tw, trigger_bin, max_amp,
                       max_env);
}
void launch_efield_channel(hipStream_t s, int n_efields, const double* traces, const double* t0, const double* zen,
                           const double* az, const int* channel, const StationDev& st, int L, double t_min, int apply_filter,
                           const double2* tw, const LengthTables& tab, double2* scratch, double* V, double2* tab_nodes)
{
    set_big_lds();
    int nh = st.N / 2;
    hipLaunchKernelGGL(efield_channel_kernel, dim3(st.n_ch), dim3(512), (size_t)FFT_MAX * 16, s, n_efields, traces, t0, zen, az,
                       channel, st, L, t_min, apply_filter, tw, tab, scratch, ilog2(nh), V, tab_nodes);
}
void launch_askaryan_spectrum(hipStream_t s, int n, const double* energy, const double* theta, const int* type,
                              const double* n_index, const double* R, const double* k_L, int model, int N, double dt,
                              double2* spec)
{
    if (n <= 0) return;
    int grid = n < 4096 ? n : 4096;
    hipLaunchKernelGGL(askaryan_spectrum_kernel, dim3(grid), dim3(256), 0, s, n, energy, theta, type, n_index, R, k_L, model,
                       N, dt, spec);
}
void launch_nplan_tables(hipStream_t s, int nh, int log2p, double2* wN, double2* cw, double2* Bf, double2* Bi, const double2* tw)
{
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)nplan_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFT_MAX * 16);
    hipLaunchKernelGGL(nplan_tables_kernel, dim3(1), dim3(1024), (size_t)(1 << log2p) * 16, s, nh, log2p, wN, cw, Bf, Bi, tw);
}
void launch_czt_test(hipStream_t s, int n_batch, int n_in, int n_out, int Q, double sgn, const double2* in, double2* out,
                     const double2* tw, double2* Bscratch, int grid)
{
    set_big_lds();
    hipLaunchKernelGGL(czt_test_kernel, dim3(grid), dim3(1024), (size_t)FFT_MAX * 16, s, n_batch, n_in, n_out, Q, sgn, in,
                       out, tw, Bscratch);
}

// test hook of wave_reduce.h: per wave 8 values per lane -> doubles out[w][WAVE_TEST_OUT]: [0..7] wave_fold4 sums (FP64), [8..15]
// the same in FP32, [16 + lane] wave_sum of value 0 per lane (FP64), [80 + lane] the same in FP32, [144 + lane] the lane below's value,
// [208] lane 63's value, [209 + lane] the value of lane - 32
__global__ void __launch_bounds__(256) wave_reduce_test_kernel(int n_waves, const double* __restrict__ in, double* __restrict__ out)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    double v[8];
    float f[8];
    for (int i = 0; i < 8; i++) {
        v[i] = in[((long)w * 8 + i) * 64 + lane];
        f[i] = (float)v[i];
    }
    double* o = out + (long)w * WAVE_TEST_OUT;
    const double a = wave_fold4(v[0], v[1], v[2], v[3]), b = wave_fold4(v[4], v[5], v[6], v[7]);
    const float af = wave_fold4(f[0], f[1], f[2], f[3]), bf = wave_fold4(f[4], f[5], f[6], f[7]);
    const double s64[8] = {wave_row_value<0>(a), wave_row_value<1>(a), wave_row_value<2>(a), wave_row_value<3>(a),
                           wave_row_value<0>(b), wave_row_value<1>(b), wave_row_value<2>(b), wave_row_value<3>(b)};
    const float s32[8] = {wave_row_value<0>(af), wave_row_value<1>(af), wave_row_value<2>(af), wave_row_value<3>(af),
                          wave_row_value<0>(bf), wave_row_value<1>(bf), wave_row_value<2>(bf), wave_row_value<3>(bf)};
    if (lane < 8) {
        o[lane] = s64[lane];
        o[8 + lane] = (double)s32[lane];
    }
    o[16 + lane] = wave_sum(v[0]);
    o[80 + lane] = (double)wave_sum(f[0]);
    o[144 + lane] = (double)wave_from_lane_below(f[1]);
    if (lane == 0) o[208] = (double)wave_lane_value(f[1], 63);
    o[209 + lane] = (double)wave_from_lower_half(f[2]);
}
void launch_wave_reduce_test(hipStream_t s, int n_waves, const double* in, double* out)
{
    hipLaunchKernelGGL(wave_reduce_test_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, s, n_waves, in, out);
}


#ifdef NRHIP_CONV_TIMING
// ---- probe: the transform pair of channel_conv_kernel alone (tools/conv_pair_probe.py) -- one 512-thread block per CU, n_iter pairs each
// on an event of L samples; shader clocks of wave 0 and of wave 5 per phase (forward, spectrum pass, inverse), wall time by the caller
__global__ void __launch_bounds__(CONV_NT, 2)
conv_pair_probe_kernel(const double2* __restrict__ tw, const double2* __restrict__ w16, const double2* __restrict__ G, int n_iter, int L,
                       int variant, unsigned long long* __restrict__ clk)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    const double2* cft = w16 + (FFT_MAX / 2 + 1);
    for (int i = threadIdx.x; i < conv_lds_elems(FFT_MAX); i += blockDim.x) z[i] = make_double2(1e-3 * (i & 15), 1e-3);
    __syncthreads();
    unsigned long long c[3] = {0, 0, 0};
    for (int it = 0; it < n_iter; it++) {
        const double2* Gi = G + (variant == 1 ? (long)((blockIdx.x * 131 + it * 7) % 64) * NRHIP_G_STRIDE : 0);   // 1: a cold-ish table per pair
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        conv_fwd<FFT_LOG2_MAX, CONV_NT>(tw, cft, L >> 1);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        conv_mid<FFT_LOG2_MAX, CONV_NT>(Gi, w16);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        conv_inv<FFT_LOG2_MAX, CONV_NT>(tw, cft);
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        c[0] += t1 - t0; c[1] += t2 - t1; c[2] += t3 - t2;
    }
    if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 5)) {
        const int o = (threadIdx.x >> 6) == 0 ? 0 : 3;
        for (int q = 0; q < 3; q++) atomicAdd(&clk[o + q], c[q]);
    }
}
void launch_conv_pair_probe(hipStream_t s, const double2* tw, const double2* w16, const double2* G, int n_iter, int L, int variant,
                            unsigned long long* clk)
{
    set_big_lds();
    (void)hipFuncSetAttribute((const void*)conv_pair_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes(FFT_LOG2_MAX));
    hipLaunchKernelGGL(conv_pair_probe_kernel, dim3(channel_grid_blocks() / 2), dim3(CONV_NT), (size_t)conv_lds_bytes(FFT_LOG2_MAX), s, tw, w16, G,
                       n_iter, L, variant, clk);
}
#endif
}  // namespace nrhip

#ifdef NRHIP_CONV_TIMING
extern "C" int nrhip_debug_conv_clocks(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(nrhip::g_conv_clk), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nrhip::g_conv_clk), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
