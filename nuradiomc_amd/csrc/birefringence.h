// birefringence.h -- batch descriptor of the birefringence kernels (birefringence.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "ray_device.h"

namespace nrhip {

struct BireBatch {
    long n_rays;
    const double *x1, *x2;     // [n_rays][3] end points of the rays
    const double* C0;          // [n_rays]
    const int* n_points;       // [n_rays] acc = int(path length / m)
    const long* step_offset;   // [n_rays] first step record of the ray
    IceConst ice;
    const double *knots, *coeffs;  // the three depth splines (nx, ny, nz), concatenated
    int n_knots[3];
    double n_ref;              // 1.78 (analyticraytracing.py:2421)
    double angle_to_iceflow;   // deg, NaN = none
    int n_f;
    double sampling_rate;
    double* spline_pieces = nullptr;   // [BIRE_MAX_KNOTS][7] scratch the launch fills (bire_pieces_kernel); nullptr or too many knots: de Boor
    unsigned long long* counters = nullptr;   // (nullable) [0] += path steps made by bire_steps_kernel, [1] += (step, frequency bin) pairs bire_propagate_kernel applied
};
#define BIRE_MAX_KNOTS 96

void launch_birefringence(hipStream_t s, const BireBatch& b, int max_points, double* steps, double2* spec);
// the two halves separately: log_norm[ray] (nullable) = log of an upper bound on the 2-norm gain of the ray's whole path;
// active (nullable): rays with active[ray] == 0 are left untouched
#define BIRE_LOG_FIXED 1099511627776.0  // 2^40: log_norm is a fixed-point sum (deterministic whatever the order of the atomics)
void launch_birefringence_steps(hipStream_t s, const BireBatch& b, int max_points, double* steps, long long* log_norm);
void launch_birefringence_propagate(hipStream_t s, const BireBatch& b, const double* steps, double2* spec, const int* active);

}  // namespace nrhip
