/*
 * nrhip.h -- C ABI of libnrhip.so: the MI355X (gfx950) hot path for NuRadioMC-style simulations.
 *
 * Plain C, caller-allocated buffers, int status returns (0 = ok, <0 = error, see nrhip_last_error),
 * no exceptions, no torch / numpy types.  One context per (process, GPU); a context is not
 * thread-safe, different contexts are independent.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the
 * nu-radio/NuRadioMC tree).  The only native boundary the reference itself has is the Cython wrapper
 * NuRadioMC/SignalProp/CPPAnalyticRayTracing/wrapper.pyx:3-6 (find_solutions2,
 * get_attenuation_along_path2, get_attenuation_length_wrapper -- scalar, one ray / one frequency per
 * call, callee-allocated leaked arrays); the batch functions below are what a maintainer binds instead
 * (see INTEGRATION.md).
 *
 * Units are NuRadioMC's: metre, nanosecond, GHz, radian, eV, volt.  All floating point is IEEE
 * binary64; spectra are interleaved (re, im) binary64 pairs.
 *
 * Pointers marked HOST are ordinary process memory; pointers marked DEV are device (HBM) pointers of
 * the context's GPU, e.g. obtained from nrhip_malloc or from any HIP allocation (torch data_ptr()).
 */
#ifndef NRHIP_H
#define NRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRHIP_MAX_SOLUTIONS 2 /* 2 + 4 * n_reflections, n_reflections = 0 (propagation_base_class.py:424-429) */

/* attenuation model ids: NuRadioMC/utilities/attenuation.py:14 (SP1 1, GL1 2, MB1 3, GL2 4) */
#define NRHIP_ATT_SP1 1
#define NRHIP_ATT_GL1 2
#define NRHIP_ATT_MB1 3
#define NRHIP_ATT_GL2 4

/* Askaryan models (NuRadioMC/SignalGen/parametrizations.py:24-26) */
#define NRHIP_ASK_ALVAREZ2009 0
#define NRHIP_ASK_ALVAREZ2000 1
#define NRHIP_ASK_ZHS1992 2

/* shower types */
#define NRHIP_SHOWER_HAD 0
#define NRHIP_SHOWER_EM 1

/* analytic antenna models (NuRadioReco/detector/antennapattern.py:1580-1768) */
#define NRHIP_ANT_VPOL 0
#define NRHIP_ANT_HPOL 1

typedef struct nrhip_ctx nrhip_ctx;

/* ---- context -------------------------------------------------------------------------------------
 * Stands in for constructing `ray_tracing(medium, attenuation_model, ...)`
 * (NuRadioMC/SignalProp/analyticraytracing.py:1938-2041) with an IceModelSimple medium
 * n(z) = n_ice - delta_n * exp(z / z_0) (NuRadioMC/utilities/medium_base.py:254-277).           */
int nrhip_ctx_create(int device, double n_ice, double delta_n, double z_0, int attenuation_model,
                     nrhip_ctx** out);
void nrhip_ctx_destroy(nrhip_ctx* ctx);
const char* nrhip_last_error(void);
int nrhip_device_count(void);
int nrhip_synchronize(nrhip_ctx* ctx);

/* device memory helpers (so a host language without a HIP binding can keep data resident) */
int nrhip_malloc(nrhip_ctx* ctx, uint64_t bytes, void** dev_ptr);
int nrhip_free(nrhip_ctx* ctx, void* dev_ptr);
int nrhip_memcpy_h2d(nrhip_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes);
int nrhip_memcpy_d2h(nrhip_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes);

/* ---- ray tracing ----------------------------------------------------------------------------------
 * Batched ray_tracing.set_start_and_end_point + find_solutions + get_solution_type + get_path_length +
 * get_travel_time + get_launch_vector + get_receive_vector + get_reflection_angle
 * (analyticraytracing.py:2057, :2118, :2132, :2650, :2697, :2560, :2593, :2626), replacing
 * wrapper.pyx find_solutions (:8-27).
 *
 * Pair i runs from x1[i] (emitter / vertex) to x2[i] (receiver) when n_x2 == 0; when n_x2 > 0 the
 * pairs are the outer product (vertex i / n_x2, receiver i % n_x2) and x1 holds n_pairs / n_x2 rows.
 * Outputs are [n_pairs][NRHIP_MAX_SOLUTIONS](...) tables sorted by C0, padded with type 0 / NaN, the
 * layout of the reference's HDF5 station tables.  refl_angle is NaN where the reference returns None.
 * All pointers HOST.                                                                              */
int nrhip_find_solutions_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2,
                               int32_t n_x2, int32_t* n_sol, int32_t* type, double* C0, double* C1,
                               double* D, double* T, double* launch, double* receive,
                               double* refl_angle);

/* Batched ray_tracing.get_attenuation on an explicit frequency list
 * (analyticraytracing.py:2744 -> get_attenuation_along_path :933-1089, Python branch), replacing the
 * per-frequency wrapper.pyx get_attenuation_along_path (:30-31).
 * Ray r goes from x1[r] to x2[r] with launch parameter C0[r]; att is [n_rays][n_freq];
 * neval (may be NULL) receives the number of integrand evaluations per item.  All pointers HOST.   */
int nrhip_attenuation_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2,
                            const double* C0, int32_t n_freq, const double* freqs, double* att,
                            int32_t* neval);

/* attenuation.get_attenuation_length(z, f, model) (NuRadioMC/utilities/attenuation.py:145-262),
 * replacing wrapper.pyx get_attenuation_length (:33-34); elementwise over n values.  HOST.         */
int nrhip_attenuation_length(nrhip_ctx* ctx, int64_t n, const double* z, const double* freq, double* L);

#ifdef __cplusplus
}
#endif
#endif /* NRHIP_H */
