// earth.hip -- per-event Earth-absorption weight (NuRadioMC/utilities/earth_attenuation.py:12-60, called per event group by
// simulation.py:880-903) with the 'ctw' and 'ghandi' cross sections (NuRadioMC/utilities/cross_sections.py:64-120, :280-281,
// :301-311, :393-421).
//
// Chord modes ('core_mantle_crust', 'PREM'; PREM.slant_depth :183-240): a wave per event.  The column density is the
// trapezoid rule over n_steps = int(distance / step) (+1) samples spread from the vertex to the surface, up to 25 500 of them
// for a chord through the whole Earth; lane l of the wave evaluates the samples base + l of a 64-sample chunk (chunks
// overlap by one sample), the interval sums stay in registers and are folded over the wave at the end.  Every sample's
// radius is computed with the reference's operations in the reference's order (no contraction; the two 3-element BLAS
// inner products as the fma chain that ddot's tail loop is on x86-64 with FMA): the surface sample has r = R up to rounding and counts with the
// outermost layer's density or with 0 depending on that rounding.
// Closed-form modes ('simple' :63-86, 'core_mantle_crust_simple' :89-130): a lane per event.
#include "earth.h"
#include "../../include/nrhip.h"

namespace nrhip {

// cross_sections.param (:64-120), 'ctw': log10(sigma / cm^2) = c1 + c2 l + c3 l^2 + c4 / l, l = ln(log10(E / GeV) - c0)
__device__ inline double ctw_param(double energy, const double c0, const double c1, const double c2, const double c3, const double c4)
{
    const double epsilon = log10(energy / 1e9);
    const double l_eps = log(epsilon - c0);
    const double crscn = c1 + c2 * l_eps + c3 * (l_eps * l_eps) + c4 / l_eps;
    return pow(10., crscn) * (0.01 * 0.01);
}

// get_nu_cross_section(E, flavors, inttype='total', cross_section_type='ctw') (:301-311): nc + cc
__device__ inline double ctw_total(double energy, int flavor)
{
    if (energy < 1e4 * 1e9) return __builtin_nan("");  // :69-76: not valid below 1e4 GeV, NaN
    if (flavor >= 0)
        return ctw_param(energy, -1.826, -17.31, -6.448, 1.431, -18.61) + ctw_param(energy, -1.826, -17.31, -6.406, 1.431, -17.91);
    return ctw_param(energy, -1.033, -15.95, -7.296, 1.569, -18.30) + ctw_param(energy, -1.033, -15.95, -7.247, 1.569, -17.72);
}

// get_nu_cross_section(..., inttype='total') of the cross-section models evaluated here
__device__ inline double total_cross_section(int type, double energy, int flavor)
{
    if (type == NRHIP_XS_GHANDI) return 7.84e-36 * (0.01 * 0.01) * pow(energy / 1e9, 0.363);  // :280-281
    if (type == NRHIP_XS_GIVEN) return energy;   // the caller evaluated its (tabulated) model: `energy` carries sigma [m^2]
    return ctw_total(energy, flavor);
}

__device__ inline double dot3_blas(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));  // the BLAS ddot tail on three elements
}

// PREM.density (:171-181) times nothing: lower <= r < upper picks the layer, 0 outside every layer
__device__ inline double earth_density(double r, const EarthModelDev& m)
{
    double lower = 0.;
    for (int k = 0; k < m.n_layers; k++) {
        const double upper = m.radii[k];
        if (lower <= r && r < upper) {
            if (m.coef[k][1] == 0. && m.coef[k][2] == 0. && m.coef[k][3] == 0.) return m.coef[k][0];  // constant layer: same double
            const double x = r / m.earth_radius;
            return ((m.coef[k][0] + m.coef[k][1] * x) + m.coef[k][2] * (x * x)) + m.coef[k][3] * (x * x * x);
        }
        lower = upper;
    }
    return 0.;
}

__global__ void __launch_bounds__(256)
earth_chord_kernel(EarthBatch b, EarthModelDev m, double* __restrict__ weight, double* __restrict__ slant_out)
{
    const int lane = threadIdx.x & 63;
    const long ev = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ev >= b.n) return;
    const double R = m.earth_radius;
    double d0 = b.direction[3 * ev], d1 = b.direction[3 * ev + 1], d2 = b.direction[3 * ev + 2];
    const double norm = sqrt(dot3_blas(d0, d1, d2, d0, d1, d2));  // direction /= np.linalg.norm(direction) (:211)
    d0 /= norm; d1 /= norm; d2 /= norm;
    const double e0 = b.endpoint[3 * ev], e1 = b.endpoint[3 * ev + 1], e2 = b.endpoint[3 * ev + 2] + R;
    const double dot_prod = dot3_blas(e0, e1, e2, d0, d1, d2);
    const double discriminant = dot_prod * dot_prod - ((e0 * e0 + e1 * e1) + e2 * e2) + R * R;
    double slant = 0.;
    const double distance = discriminant > 0. ? -dot_prod + sqrt(discriminant) : 0.;
    if (discriminant > 0. && distance > 0.) {
        long n_steps = (long)(distance / b.step);
        if (fmod(distance, b.step) != 0.) n_steps += 1;
        if (n_steps > 1) {
            const double tstep = 1. / (double)(n_steps - 1);  // np.linspace(0, 1, n): arange(n) * step, last sample = 1
            double acc = 0.;
            for (long base = 0; base < n_steps - 1; base += 63) {
                const long i = base + lane;
                double t = 0., y = 0.;
                if (i < n_steps) {
                    t = (i == n_steps - 1) ? 1. : (double)i * tstep;
                    const double td = t * distance;
                    const double x = e0 + td * d0, yy = e1 + td * d1, z = e2 + td * d2;
                    const double r = sqrt((x * x + yy * yy) + z * z);
                    y = earth_density(r, m) * distance;
                }
                const double t_next = __shfl_down(t, 1);
                const double y_next = __shfl_down(y, 1);
                if (lane < 63 && i + 1 < n_steps) acc += (t_next - t) * (y_next + y) / 2.0;  // np.trapz
            }
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            slant = acc;
        }
    }
    if (lane == 0) {
        if (slant_out) slant_out[ev] = slant;
        if (weight) {
            // get_interaction_length(density=1.) (:393-421): m_n / sigma / density
            const double L_int = b.nucleon_mass / total_cross_section(b.cross_section_type, b.energy[ev], b.flavor[ev]) / 1.;
            weight[ev] = exp(-slant / L_int);
        }
    }
}

__global__ void __launch_bounds__(256)
earth_closed_form_kernel(EarthBatch b, double* __restrict__ weight)
{
    const long ev = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ev >= b.n) return;
    const double th = b.zenith[ev];
    double w = 1.;
    if (!(th <= 0.5 * M_PI)) {  // coming from below
        if (b.mode == NRHIP_EARTH_SIMPLE) {
            const double sigma = total_cross_section(b.cross_section_type, b.energy[ev], 0);  // flavors=0 (:83)
            const double d = -2 * b.simple_radius * cos(th);
            w = exp(-d * sigma * b.simple_density / b.amu);
        } else {
            const double sigma = total_cross_section(b.cross_section_type, b.energy[ev], b.flavor[ev]);
            const double RE = b.layer_radii[2];
            const double s = sin(M_PI - th);
            if (th <= b.layer_theta[0]) {          // only the outer layer
                const double d_outer = -2 * RE * cos(th);
                w = exp(-d_outer * sigma * b.layer_density[2] / b.amu);
            } else if (th <= b.layer_theta[1]) {   // outer and middle layer
                const double d_middle = 2 * sqrt(b.layer_radii[1] * b.layer_radii[1] - b.layer_radii[2] * b.layer_radii[2] * s * s);
                const double d_outer = -2 * RE * cos(th) - d_middle;
                w = exp(-d_outer * sigma * b.layer_density[2] / b.amu - d_middle * sigma * b.layer_density[1] / b.amu);
            } else {                               // all three layers
                const double d_inner = 2 * sqrt(b.layer_radii[0] * b.layer_radii[0] - b.layer_radii[2] * b.layer_radii[2] * s * s);
                const double d_middle = 2 * sqrt(b.layer_radii[1] * b.layer_radii[1] - b.layer_radii[2] * b.layer_radii[2] * s * s) - d_inner;
                const double d_outer = -2 * RE * cos(th) - d_middle - d_inner;
                w = exp(-d_outer * sigma * b.layer_density[2] / b.amu - d_middle * sigma * b.layer_density[1] / b.amu -
                        d_inner * sigma * b.layer_density[0] / b.amu);
            }
        }
    }
    weight[ev] = w;
}

void launch_earth_weights(hipStream_t s, const EarthBatch& b, const EarthModelDev& model, double* weight, double* slant_depth)
{
    if (b.n <= 0) return;
    if (b.mode == NRHIP_EARTH_SIMPLE || b.mode == NRHIP_EARTH_CORE_MANTLE_CRUST_SIMPLE) {
        hipLaunchKernelGGL(earth_closed_form_kernel, dim3((unsigned)((b.n + 255) / 256)), dim3(256), 0, s, b, weight);
    } else {
        hipLaunchKernelGGL(earth_chord_kernel, dim3((unsigned)((b.n + 3) / 4)), dim3(256), 0, s, b, model, weight, slant_depth);
    }
}

}  // namespace nrhip
