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
    // solution slots per pair: NRHIP_MAXS, or 2 + 4 n_reflections with reflections off the bottom of an ice shelf; then also
    // per slot the number of bottom reflections and bit j of surface_mask = path segment j reflects at the surface
    int stride = NRHIP_MAXS;
    const int* reflection = nullptr;
    const int* surface_mask = nullptr;
};

void launch_raytrace(hipStream_t stream, long n_pairs, const double* x1, const double* x2, int n_ch,
                     const IceConst& m, const RayRecords& out, const double* max_dist = nullptr,
                     const int* perm = nullptr, const double* given_C0 = nullptr, unsigned long long* eval_count = nullptr,
                     const double* given_D = nullptr, const double* given_T = nullptr, bool reference_procedure = false,
                     bool channel_major = false);
// channel_major (with perm): the finder walks the pairs channel by channel -- a wave sees neighbouring events from ONE antenna --
// instead of event by event; pays for stations whose antennas sit at very different depths (measured: the 24-channel RNO-G-like
// station -21 % of the stage; a string of dipoles 1 m apart +3 %, its stores no longer coalesce)
// ---- reflections off the bottom of an ice shelf (raytrace_refl.hip) ----
#define NRHIP_MAX_REFLECTIONS 4
struct ReflRecords {   // [n_pairs][2 + 4 n_reflections]
    int *n_sol, *type, *reflection, *reflection_case, *n_segments, *surface_mask;
    double *C0, *C1, *D, *T, *launch, *receive, *refl_angle;
    double* seg_zint;  // [n_pairs][stride][n_reflections + 1][3]: attenuation limits (z1, z2 mirrored, z_turn) per path segment
    double* seg_C0;    // [n_pairs][stride][n_reflections + 1]: C0 of the segment's ray, NaN = no such segment
};
void launch_find_refl(hipStream_t stream, long n_pairs, int n_reflections, const double* x1, const double* x2, int n_x2,
                      const IceConst& m, double z_refl, int* cand_n, double* cand_C0, bool reference_procedure = false);
// stride = solution slots per pair: 2 + 4 n_reflections after launch_find_refl; any value with given records
void launch_records_refl(hipStream_t stream, long n_pairs, int n_reflections, int stride, const double* x1, const double* x2,
                         int n_x2, const IceConst& m, double z_refl, const int* cand_n, const double* cand_C0, int given,
                         const ReflRecords& out);
void launch_segment_product(hipStream_t stream, long n_rays, int n_seg_max, int n_freq, const double* seg_zint,
                            const double* seg_att, double* att);

void launch_event_cells(hipStream_t stream, int n_events, const double* vertex, const double* x2, int* cell, int* hist);
void launch_event_perm(hipStream_t stream, int n_events, const int* cell, int* cursor, int* perm);


}  // namespace nrhip
