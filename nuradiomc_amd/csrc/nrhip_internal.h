// nrhip_internal.h -- shared declarations between the kernels and the C-ABI layer (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "ray_device.h"

#define NRHIP_MAXS 2  // 2 + 4 * n_reflections ray solutions per pair (propagation_base_class.py:424-429)

namespace nrhip {

// Per-(pair, solution) ray records in HBM, [n_pairs][NRHIP_MAXS] (the reference's HDF5 table layout:
// NaN / 0 padding for missing solutions).  136 B per solution + 4 B per pair.
struct RayRecords {
    int* n_sol;          // [n_pairs]
    int* type;           // [n_pairs][MAXS]  1 direct, 2 refracted, 3 reflected, 0 = none
    double* C0;          // [n_pairs][MAXS]
    double* C1;
    double* D;           // path length [m]
    double* T;           // travel time [ns]
    double* launch;      // [n_pairs][MAXS][3]
    double* receive;     // [n_pairs][MAXS][3]
    double* refl_angle;  // surface reflection zenith angle, NaN = none
};

void launch_raytrace(hipStream_t stream, long n_pairs, const double* x1, const double* x2, int n_ch,
                     const IceConst& m, const RayRecords& out, const double* max_dist = nullptr,
                     const int* perm = nullptr, const double* given_C0 = nullptr);
void launch_event_cells(hipStream_t stream, int n_events, const double* vertex, const double* x2, int* cell, int* hist);
void launch_event_perm(hipStream_t stream, int n_events, const int* cell, int* cursor, int* perm);


}  // namespace nrhip
