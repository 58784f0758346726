// arz.h -- batch descriptor of the ARZ kernels (arz.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

struct ArzBatch {
    long n_rays;
    const double *energy, *theta, *distance;  // [n_rays]
    const int* shower_type;                   // [n_rays] 0 HAD, 1 EM
    const double* em_factor;                  // [n_rays] energy fraction of the electromagnetic component (HAD only)
    const int* profile_index;                 // [n_rays] row of profile_ce
    const double* rescale;                    // [n_rays] amplitude factor of the profile (E / E_library) or nullptr
    int n_profiles, n_depth;
    const double* profile_depth;              // [n_depth]
    const double* profile_ce;                 // [n_profiles][n_depth]
    const double* parameters;                 // [2][7]: (Af, freq_pos, freq_neg, exp_pos, exp_neg, t0_pos, t0_neg) for HAD, EM
    int N;
    double dt, n_index, interp_factor2;
    int shift_for_xmax;
    double maximum_angle;
    const double* n_index_ray = nullptr;      // [n_rays] index of refraction at the shower (overrides n_index)
    double* form_factor_table = nullptr;      // [ARZ_TABLE_DOUBLES] scratch the launch fills (arz_form_factor_table_kernel); nullptr = none
    unsigned long long* eval_count = nullptr; // (nullable) += evaluations of the integrand (the FP64 view of bench.py prices the model by them)
};

// piecewise degree-6 Taylor polynomials of the form factor exp(-|t| / t0) + (1 + f |t|)^e per (shower type, sign of t):
// cells of 1/512 ns up to |t| = 2.5 ns, 8 doubles per cell (one 64 B line)
#define ARZ_TABLE_CELLS 1280
// beyond 2.5 ns the form factor is the smooth power-law tail (the exponential is < 1e-16 there): cells of 1/16 ns up to 20.5 ns,
// the same degree-6 polynomials (truncation < 2e-11 of the tail's own value) instead of exp(e log(1 + f |t|)) per point
#define ARZ_FAR_CELLS 288
#define ARZ_TABLE_DOUBLES (4 * (ARZ_TABLE_CELLS + ARZ_FAR_CELLS) * 8)

void launch_arz(hipStream_t s, const ArzBatch& b, double* vp, double* trace, int* status, int* vp_range = nullptr,
                int* silent = nullptr);

}  // namespace nrhip
