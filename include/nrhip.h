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

/* attenuation model ids: NuRadioMC/utilities/attenuation.py:14 (SP1 1, GL1 2, MB1 3, GL2 4, GL3 5) */
#define NRHIP_ATT_SP1 1
#define NRHIP_ATT_GL1 2
#define NRHIP_ATT_MB1 3
#define NRHIP_ATT_GL2 4
#define NRHIP_ATT_GL3 5   /* needs nrhip_ctx_set_gl3_table */

/* Askaryan models (NuRadioMC/SignalGen/parametrizations.py:24-26) */
#define NRHIP_ASK_ALVAREZ2009 0
#define NRHIP_ASK_ALVAREZ2000 1
#define NRHIP_ASK_ZHS1992 2
#define NRHIP_ASK_ARZ2019 3   /* time-domain models: need nrhip_station_set_arz + nrhip_station_set_shower_profiles */
#define NRHIP_ASK_ARZ2020 4

/* shower types */
#define NRHIP_SHOWER_HAD 0
#define NRHIP_SHOWER_EM 1

/* analytic antenna models (NuRadioReco/detector/antennapattern.py:1580-1768) */
#define NRHIP_ANT_VPOL 0
#define NRHIP_ANT_HPOL 1
#define NRHIP_ANT_LPDA 2   /* analytic_LPDA (antennapattern.py:1676-1713): both VEL components, three phase regimes */
#define NRHIP_ANT_TABLE 3  /* tabulated pattern (AntennaPattern, antennapattern.py:1338-1577): nrhip_antenna_table */

typedef struct nrhip_ctx nrhip_ctx;

/* A tabulated vector effective length on a regular (frequency, theta, phi) grid, the content of the reference's antenna
 * pickles (get_pickle_antenna_response, antennapattern.py:540-632): tri-linear complex interpolation
 * (_get_antenna_response_vectorized_raw :1426-1577), zero outside the frequency range.  Flat index of the complex
 * tables: iF * n_theta * n_phi + iP * n_theta + iT; values interleaved (re, im).  orientation: boresight (theta, phi) and
 * tine-plane normal (theta, phi) of the frame the table was simulated in (:1190-1216).  HOST pointers (copied).     */
typedef struct {
    int32_t n_freq, n_theta, n_phi;
    const double* freqs;      /* [n_freq] ascending, equidistant [GHz] */
    const double* thetas;     /* [n_theta] ascending, equidistant [rad] */
    const double* phis;       /* [n_phi] ascending, equidistant [rad] */
    const double* vel_theta;  /* [n_freq * n_theta * n_phi][2] */
    const double* vel_phi;
    double orientation[4];
} nrhip_antenna_table;

/* ---- context -------------------------------------------------------------------------------------
 * Stands in for constructing `ray_tracing(medium, attenuation_model, ...)`
 * (NuRadioMC/SignalProp/analyticraytracing.py:1938-2041) with an IceModelSimple medium
 * n(z) = n_ice - delta_n * exp(z / z_0) (NuRadioMC/utilities/medium_base.py:254-277).           */
int nrhip_ctx_create(int device, double n_ice, double delta_n, double z_0, int attenuation_model,
                     nrhip_ctx** out);
void nrhip_ctx_destroy(nrhip_ctx* ctx);

/* GL3 (attenuation.py:206-221): L(z, f) = slope(depth) * f + offset(depth) with the depth table of
 * NuRadioMC/utilities/data/GL3_params.csv (rows: depth [m, positive, increasing], slope, offset; linear interpolation,
 * end values outside).  Its path integral follows the reference's speed-optimised scheme (analyticraytracing.py:998-1064:
 * 10 m segment sums, QUADPACK on ds around the turning depth).  HOST pointers.                                        */
int nrhip_ctx_set_gl3_table(nrhip_ctx* ctx, int32_t n, const double* depth, const double* slope, const double* offset);

/* Which solution finder find_solutions runs (analyticraytracing.py:1400-1547) -- for every later call on this context:
 *   NRHIP_FINDER_TRUE_ROOTS (default): every root of the path objective out of a bracket.  The list is the mathematically true
 *     one; the reference's list is a subset of it (it loses its first root on 0.2 ... 0.4 % of random pairs, see below).
 *   NRHIP_FINDER_REFERENCE: the reference's procedure to the letter -- scipy.optimize.root(tol=1e-6) on (delta y)^2 from
 *     log C0 = -1, the iterate kept only if (delta y)^2 < 1e-7 there (:1479-1483), then one Brent search either side, 1e-4 away
 *     (:1498-1541) -- for every pair and every call with bottom reflections.  Counts, types and order are then the reference's
 *     wherever the last bits of its libm agree with the kernels' exp / log (the acceptance test sits ~1e-7 off a double root).
 *     About three times the objective evaluations of the default.                                                              */
#define NRHIP_FINDER_TRUE_ROOTS 0
#define NRHIP_FINDER_REFERENCE 1
int nrhip_ctx_set_ray_finder(nrhip_ctx* ctx, int32_t finder);
const char* nrhip_last_error(void);
int nrhip_device_count(void);
int nrhip_synchronize(nrhip_ctx* ctx);

/* device memory helpers (so a host language without a HIP binding can keep data resident) */
int nrhip_malloc(nrhip_ctx* ctx, uint64_t bytes, void** dev_ptr);
int nrhip_free(nrhip_ctx* ctx, void* dev_ptr);
int nrhip_memset(nrhip_ctx* ctx, void* dev_dst, int32_t value, uint64_t bytes);   /* asynchronous on the context's stream */
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

/* ray_tracing.set_solution (analyticraytracing.py:2092-2116): the same tables from launch parameters that were found
 * earlier and stored (ray_tracing_C0 of the reference's output files): C0_in [n_pairs][NRHIP_MAX_SOLUTIONS], NaN = no
 * solution in that slot; no root finding.  All pointers HOST.                                              */
int nrhip_ray_records_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                            const double* C0_in, int32_t* n_sol, int32_t* type, double* C0, double* C1, double* D,
                            double* T, double* launch, double* receive, double* refl_angle);

/* ---- reflections off the bottom of an ice shelf (medium.reflection, e.g. mooresbay_simple: -576 m) -------------------
 * ray_tracing(medium, n_reflections=n).find_solutions (analyticraytracing.py:2118-2130): the plain 2-D solution finder,
 * then per number of bottom reflections r = 1..n one call for rays that start upwards (reflection_case 1) and one for
 * rays that start downwards (2) (ray_tracing_2D.find_solutions :1400-1547 with get_delta_y's reflection loop :204-272,
 * C++ twin cpp:405-470, :602-874); more than 2 + 4 n solutions -> none.  Replaces wrapper.pyx find_solutions(x1, x2,
 * n_ice, delta_n, z_0, reflection, reflection_case, ice_reflection).
 * Tables are [n_pairs][2 + 4 n_reflections] in the order the reference lists the solutions (call order, C0 ascending
 * inside a call), padded with type 0 / NaN.  D, T: sums over the path segments between two bottom reflections
 * (get_path_segments :1091-1159, get_path_length_analytic :602-690, get_travel_time_analytic :692-783); launch / receive:
 * get_launch_vector :2560, get_receive_vector :2593; refl_angle: zenith angle of the reflections at the SURFACE (the same
 * in every segment that has one, NaN = none), n_segments the number of path segments and bit j of surface_mask whether
 * segment j has one (get_reflection_angle :1201-1237 returns one entry, angle or None, per segment).  apply_propagation_effects (:2966-3009) then multiplies
 * the eTheta / ePhi spectra by r_p / r_s once per surface reflection and by (reflection_coefficient e^{i phase_shift})
 * once per bottom reflection.  z_reflection < 0: depth of the reflective layer.  Any output pointer may be NULL.
 * All pointers HOST.                                                                                                */
int nrhip_find_solutions_reflections_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                                           int32_t n_reflections, double z_reflection, int32_t* n_sol, int32_t* type,
                                           double* C0, double* C1, int32_t* reflection, int32_t* reflection_case, double* D,
                                           double* T, double* launch, double* receive, double* refl_angle,
                                           int32_t* n_segments, int32_t* surface_mask);

/* ray_tracing.set_solution (:2092-2116) with bottom reflections: n_sol, C0, reflection, reflection_case are INPUTS
 * ([n_pairs][2 + 4 n_reflections]); everything else is derived from them as above.                                  */
int nrhip_ray_records_reflections_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                                        int32_t n_reflections, double z_reflection, int32_t* n_sol, int32_t* type,
                                        double* C0, double* C1, int32_t* reflection, int32_t* reflection_case, double* D,
                                        double* T, double* launch, double* receive, double* refl_angle,
                                        int32_t* n_segments, int32_t* surface_mask);

/* get_attenuation_along_path with bottom reflections (:933-1089): the product over the path segments of
 * exp(-int ds / L(z, f)); one ray per row (x1, x2, C0, reflection, reflection_case), att [n_rays][n_freq].
 * segment_att (may be NULL): the factors of the single segments, [n_rays][max(reflection) + 1][n_freq], NaN where a ray
 * has fewer segments (the reference interpolates every segment's factors to the full frequency grid before it
 * multiplies them, :1078-1084).  HOST pointers. */
int nrhip_attenuation_reflections_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2, const double* C0,
                                        const int32_t* reflection, const int32_t* reflection_case, double z_reflection,
                                        int32_t n_freq, const double* freqs, double* att, double* segment_att);

/* ---- ARZ time-domain Askaryan model ---------------------------------------------------------------------------------
 * ARZ.get_time_trace (NuRadioMC/SignalGen/ARZ/ARZ.py:500-673) for a batch of (shower, ray) pairs: vector potential of the
 * charge-excess profile (get_vector_potential :36-275: trapezoid rule over the profile, stretches radiating within +-1 ns
 * of the observer time refined interp_factor2 (100) times), E = -dA/dt, rotated to the on-sky basis (eR, eTheta, ePhi) of
 * the direction to the shower maximum; zero trace more than maximum_angle (20 deg) off the Cherenkov angle.
 * shower_type 0 HAD / 1 EM; em_factor = ARZ.em_fraction(E) (:436-447, HAD only); profile_index: row of profile_ce
 * [n_profiles][n_depth] on the common grid profile_depth (NuRadioReco units: depth in g/cm^2 * 6.2415e37); rescale
 * (NULL = 1): E / E_library; parameters [2][7] = (Af, freq_pos, freq_neg, exp_pos, exp_neg, t0_pos, t0_neg) for HAD, EM
 * (:394-434).  trace [n_rays][3][N]; vector_potential (may be NULL) [n_rays][N + 1][2] = (A_x, A_z), A_y = 0.
 * The library handling (closest energy, random profile number, same_shower) stays on the host.  HOST pointers.   */
int nrhip_arz_time_trace_batch(nrhip_ctx* ctx, int64_t n_rays, const double* energy, const double* theta, const double* distance,
                               const int32_t* shower_type, const double* em_factor, const int32_t* profile_index,
                               const double* rescale, int32_t n_profiles, int32_t n_depth, const double* profile_depth,
                               const double* profile_ce, const double* parameters, int32_t N, double dt, double n_index,
                               double interp_factor2, int32_t shift_for_xmax, double maximum_angle, double* trace,
                               double* vector_potential);

/* ---- birefringence ---------------------------------------------------------------------------------------------------
 * ray_tracing.get_pulse_propagation_birefringence (analyticraytracing.py:2369-2445) for a batch of rays: the path of ray i
 * (x1, x2, C0; no bottom reflections) is cut into int(path_length[i] / m) points (get_path :1239-1291, :2148-2163,
 * optionally rotated by angle_to_iceflow [deg], NaN = none); per step the principal indices n(z) + spline_j(-z) - n_ref
 * (j = x, y, z: cubic B-splines (knots, coeffs) of medium_base.IceModelBirefringence :378-420, the three of them
 * concatenated, n_knots[j] entries each; n_ref = 1.78), the effective indices (:2165-2207) and eigen-polarisations
 * (:2243-2367) give E <- R^T diag(1, time shift by t_1 - t_0) R E on the eTheta / ePhi spectra (time shift as
 * BaseTrace.apply_time_shift, base_trace.py:246-276).
 * spectra: [n_rays][2][n_f] interleaved complex, eTheta then ePhi, on np.fft.rfftfreq(2 (n_f - 1), 1 / sampling_rate),
 * modified in place.  step_records (may be NULL): [sum_i (int(path_length[i]) - 1)][5] = (a, b, c, d, t_1 - t_0) per step,
 * rows R = ((a, b), (c, d)); NaN delay = step skipped as in the reference (:2431-2433).  HOST pointers.            */
int nrhip_birefringence_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2, const double* C0,
                              const double* path_length, const int32_t n_knots[3], const double* knots, const double* coeffs,
                              double n_ref, double angle_to_iceflow, int32_t n_f, double sampling_rate, double* spectra,
                              double* step_records);

/* Batched ray_tracing.get_attenuation on an explicit frequency list
 * (analyticraytracing.py:2744 -> get_attenuation_along_path :933-1089, Python branch), replacing the
 * per-frequency wrapper.pyx get_attenuation_along_path (:30-31).
 * Ray r goes from x1[r] to x2[r] with launch parameter C0[r]; att is [n_rays][n_freq];
 * neval (may be NULL) receives the number of integrand evaluations per item.  All pointers HOST.   */
int nrhip_attenuation_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2,
                            const double* C0, int32_t n_freq, const double* freqs, double* att,
                            int32_t* neval);

/* Diagnostic: how many rays of the last nrhip_attenuation_batch call on this context the dense quadrature kernel (LDS lists of
 * 12 intervals per frequency) handed on to the general kernel (QUADPACK's limit of 50) -- results are the same either way.  */
int64_t nrhip_attenuation_last_overflow(nrhip_ctx* ctx);

/* attenuation.get_attenuation_length(z, f, model) (NuRadioMC/utilities/attenuation.py:145-262),
 * replacing wrapper.pyx get_attenuation_length (:33-34); elementwise over n values.  HOST.         */
int nrhip_attenuation_length(nrhip_ctx* ctx, int64_t n, const double* z, const double* freq, double* L);

/* ---- Askaryan emission ------------------------------------------------------------------------------
 * Batched askaryan.get_frequency_spectrum(energy, theta, N, dt, shower_type, n_index, R, model, k_L=...)
 * (NuRadioMC/SignalGen/askaryan.py:143-213 -> parametrizations.py:29-278): spectrum is [n][N/2+1] complex
 * (interleaved re, im).  shower_type NRHIP_SHOWER_*, model NRHIP_ASK_*; k_L is read for Alvarez2009 EM
 * showers only (the random draw of parametrizations.py:160-173 stays on the host).  All pointers HOST.   */
int nrhip_askaryan_spectrum_batch(nrhip_ctx* ctx, int64_t n, const double* energy, const double* theta,
                                  const int32_t* shower_type, const double* n_index, const double* R,
                                  const double* k_L, int32_t model, int32_t N, double dt, double* spectrum);

/* ---- station + whole hot path -----------------------------------------------------------------------
 * A station is the flat-array form of what the reference reads from `det` for one station
 * (get_relative_position / get_cable_delay / get_antenna_model / get_antenna_orientation /
 * get_number_of_samples / get_sampling_frequency), plus the analog filter chain that the user's
 * _detector_simulation_filter_amp applies (NuRadioMC/examples/01_Veff_simulation/T02RunSimulation.py:18-22)
 * as rational responses polyval(b, j f) / polyval(a, j f), coefficients highest power first, rows of
 * NRHIP_MAX_POLY doubles.                                                                                */
#define NRHIP_MAX_POLY 24
#define NRHIP_MAX_FILTERS 4

typedef struct {
    int32_t n_channels;
    const double* position;       /* [n_channels][3]  relative + absolute station position             */
    const double* cable_delay;    /* [n_channels]                                                       */
    const int32_t* antenna_model; /* [n_channels]     NRHIP_ANT_*                                       */
    const double* orientation;    /* [n_channels][4]  orientation theta, phi, rotation theta, phi [rad] */
    int32_t n_samples;            /* N: samples per simulated trace at the internal sampling rate: any even number 16 ... 8192, or 10240 / 12288 / 14336 */
    double sampling_rate;         /* internal sampling rate [GHz]                                       */
    double readout_length;        /* longest detector readout window [ns] (n_samples / sampling frequency) */
    double pre_pulse_time;        /* efieldToVoltageConverter.begin(pre_pulse_time = 200 ns)            */
    double post_pulse_time;       /* efieldToVoltageConverter.begin(post_pulse_time = 400 ns)           */
    int32_t n_att_freq;           /* coarse attenuation grid (analyticraytracing.py:885-931)            */
    const double* att_freq;
    const double* att_bound_inv_length; /* [n_att_freq] or NULL: 1 / (largest attenuation length between the surface and
                                     att_bound_depth at that frequency) [1/m].  Only used to prune: exp(-0.95 D / L_max)
                                     bounds the attenuation factor of a ray of path length D from above            */
    double att_bound_depth;       /* [m] > 0: the bound is applied to rays that stay above -att_bound_depth          */
    int32_t n_filters;
    const int32_t* filter_nb;
    const int32_t* filter_na;
    const double* filter_b;       /* [n_filters][NRHIP_MAX_POLY]                                        */
    const double* filter_a;
    /* optional depth-resolved pruning bound: depth bins [-(b+1) w, -b w], b < n, w = att_bound_bin_width [m] (n <= 63);
       att_bound_bin_inv_length[b][f] <= min over the bin of 1 / L_att(z, f).  A ray's attenuation factor then is bounded
       from above by exp(-0.95 sum_b (path length inside bin b) * that) -- much tighter than the single-length bound, and
       used instead of it when given (n = 0: not given) */
    int32_t att_bound_n_bins;
    double att_bound_bin_width;
    const double* att_bound_bin_inv_length; /* [att_bound_n_bins][n_att_freq] */
    /* [n_filters] or NULL (all rational): NRHIP_FILTER_RATIONAL b(jf)/a(jf) (butter, cheby1: signal.butter / cheby1
       (analog=True) + freqs), NRHIP_FILTER_ABS its modulus (butterabs), NRHIP_FILTER_RECTANGULAR pass band
       filter_b[0] <= f <= filter_b[1] (signal_processing.get_filter_response :237-333) */
    const int32_t* filter_kind;
    /* tabulated antenna patterns: channels with antenna_model == NRHIP_ANT_TABLE use antenna_tables[antenna_table_index[c]].
       Their voltages are computed per ray on the event's L grid (chirp-z kernel), without the pruning bounds.       */
    int32_t n_antenna_tables;
    const nrhip_antenna_table* antenna_tables;
    const int32_t* antenna_table_index; /* [n_channels] or NULL */
    /* per-channel filter chains (channelBandPassFilter's per-channel dict arguments, channelBandPassFilter.py:40-66; different
       amplifiers per channel, RNO_G/hardwareResponseIncorporator.py:40-135): n_filter_sets > 1 chains, channel c uses chain
       channel_filter_set[c], chain s has set_n_filters[s] stages and the stages of all chains follow each other in filter_nb /
       filter_na / filter_b / filter_a / filter_kind (n_filters is ignored then).  n_filter_sets <= 1: the one chain above.   */
    int32_t n_filter_sets;
    const int32_t* channel_filter_set;   /* [n_channels] */
    const int32_t* set_n_filters;        /* [n_filter_sets] */
    /* measured responses for NRHIP_FILTER_TABULATED stages: rows (frequency [GHz] ascending, linear gain, unwrapped phase [rad]);
       a stage uses filter_nb rows starting at row filter_na */
    int32_t n_filter_table_points;
    const double* filter_table;          /* [n_filter_table_points][3] */
} nrhip_station_desc;
#define NRHIP_FILTER_RATIONAL 0
#define NRHIP_FILTER_ABS 1
#define NRHIP_FILTER_RECTANGULAR 2
/* gain(f) exp(i phase(f)) of a measured amplifier / signal chain: scipy interp1d(linear, 0 outside the table) of gain and of
   the unwrapped phase, gain times the temperature correction filter_b[0] + filter_b[1] f^5 (f in GHz) --
   NuRadioReco/detector/RNO_G/analog_components.load_amp_response :10-104, what hardwareResponseIncorporator.get_filter(...,
   sim_to_data=True) multiplies the channel spectra with (:93-135) */
#define NRHIP_FILTER_TABULATED 3
/* 'gaussian_tapered' (signal_processing.py:310-321): pass band filter_b[0] .. filter_b[1] [GHz] convolved (mode 'same') with
   signal.windows.gaussian(n_frequencies, int(round(filter_b[2] / df))) on the frequency grid the filter is applied on (it
   depends on the trace length), normalised to a maximum of 1 */
#define NRHIP_FILTER_GAUSSIAN_TAPERED 4

typedef struct {
    int32_t askaryan_model;       /* NRHIP_ASK_*                                                        */
    double delta_C_cut;           /* config speedup.delta_C_cut [rad]                                   */
    double min_efield_amplitude;  /* absolute candidate cut [V/m] = speedup.min_efield_amplitude * Vrms_efield */
    double trigger_threshold;     /* simple threshold [V] on any channel                                */
    int32_t dump_traces;          /* != 0: keep the channel voltage traces of the chunk for nrhip_sim_fetch */
    int32_t no_pruning;           /* != 0: evaluate attenuation and max |E(t)| for EVERY kept ray (parity tests); by
                                     default rays of events that provably cannot pass the candidate cut are skipped */
    /* trigger (NuRadioReco/modules/trigger/simpleThreshold.py, highLowThreshold.py), all channels of the station:
       NRHIP_TRIG_SIMPLE    per-sample |V| >= trigger_threshold;
       NRHIP_TRIG_HIGH_LOW  a sample >= threshold_high AND a sample <= threshold_low inside a sliding window of
                            high_low_window (get_high_low_triggers :13-80);
       then get_majority_logic (:82-150): per-channel flags OR-dilated over coinc_window, summed over channels,
       >= n_coincidences.  n_coincidences <= 1 with NRHIP_TRIG_SIMPLE is the plain OR (trigger_type 0 and all-zero
       fields reproduce the previous behaviour).  trigger times: nrhip_sim_fetch("ev_trigger_time").              */
    int32_t trigger_type;
    int32_t n_coincidences;
    double threshold_high, threshold_low;   /* [V] */
    double high_low_window, coinc_window;   /* [ns] */
    /* speedup.amp_per_ray_solution (simulation.py:523, _calculate_amp_per_ray_solution :1868-1886): for every ray of the
       candidate event groups the per-efield voltage on the native N grid (efieldToVoltageConverterPerEfield.py:28-101,
       cable delay, filter chain), its Hilbert-envelope maximum and the time of that maximum --
       nrhip_sim_fetch("ray_max_amp_envelope" / "ray_signal_time"), NaN for rays of other events.                    */
    int32_t amp_per_ray;
    /* propagation.focusing / focusing_limit (analyticraytracing.py:2778-2888, :3011-3016): every ray's field is scaled by
       the ray-convergence factor from a second trace to the receiver moved by 1 cm (numerical branch), at most
       focusing_limit, times sqrt(n(vertex) / n(receiver)).  Doubles the ray-tracing work.                          */
    int32_t focusing;
    double focusing_limit;
    /* two-phase runs (the host draws stateful random shower parameters -- Alvarez2009's k_L, the ARZ profile number -- in the
       order in which the reference meets the showers, simulation.py:221-242, between the phases):
       select_only != 0: stop after ray tracing and the delta_C cut; tables "pair_n_sol", "slot_*", "slot_keep" and
       "shower_first_channel" (int32 [n_showers]: first channel of this station with a kept ray, -1 = none) are fetchable,
       `triggered` is zeroed, stats carries n_pairs / n_rays.
       reuse_ray_tables != 0: skip ray tracing and the cut -- the tables of the previous call with the same vertex and
       max_distance pointers, shower and group counts, station position, delta_C_cut and reflection set-up are used (anything
       else fails).  A pointer identifies a list only while its buffer lives: do not free and re-upload between the phases. */
    int32_t select_only;
    int32_t reuse_ray_tables;
    /* != 0: `triggered` is not zeroed first, triggers are OR-ed into it -- the event-group mask of an array simulated station
       by station (simulation.py:1500: "each station is treated independently"; an event group is kept when any station
       triggered); stats->n_triggered then counts the accumulated mask */
    int32_t accumulate_triggered;
    /* reflections off the bottom of an ice shelf inside the batched path (propagation.n_reflections with a medium that has a
       reflective layer, e.g. mooresbay_simple: z_reflection = -576 m, reflection_coefficient 0.82, reflection_phase_shift pi;
       medium_base.py:IceModelSimple, analyticraytracing.py:2118-2130, :2966-3009): ray tables get 2 + 4 n_reflections solution
       slots per pair ("slot_*" tables, plus "slot_reflection", "slot_reflection_case", "slot_n_segments"), a ray's attenuation is
       the product over its path segments, its field is multiplied by r_p / r_s once per surface reflection and by
       reflection_coefficient e^{i reflection_phase_shift} once per bottom reflection.  0 = none.  With `focusing` the second
       trace lists the bottom-reflected solutions too (analyticraytracing.py:2835-2840).  Not together with birefringence. */
    int32_t n_reflections;
    double z_reflection;
    double reflection_coefficient;
    double reflection_phase_shift;
    /* > 0: simulation.group_into_events (simulation.py:906-947) -- the signals of a group at this station are sorted by start
       time (electric-field start + cable delay) and cut into sub-events wherever two consecutive ones are more than this many ns
       apart; every sub-event gets its own readout window, channel sums and trigger decision, a group triggers when one of its
       sub-events does.  The candidate cut stays per group.  The ev_* / item_* tables are then per sub-event ("ev_group",
       "ev_sub_event": group and index of each; stats->n_sub_events).  0: one readout per group.  Not with ARZ / birefringence
       or the phased-array trigger. */
    double split_event_time_diff;
    /* != 0: thermal noise on every channel of the candidate events before the filter chain and the trigger
       (channelGenericNoiseAdder.bandlimited_noise, type 'rayleigh', as simulation.apply_det_response adds it: simulation.py:594-606);
       per-channel amplitudes from nrhip_station_set_noise.  The draws come from a counter-based Philox4x32-10 generator keyed by
       noise_seed and counted by (event group id, sub-event, channel, frequency bin): an event's noise does not depend on batching or
       on the number of GPUs; against the reference (one sequential numpy stream in loop order) the agreement is statistical.
       Group ids: noise_group_id (DEV int64 [n_groups]) or, if NULL, noise_group_offset + index.  With noise every channel of every
       candidate event is evaluated (chirp-z kernel), the result-neutral pruning of the channel stage does not apply. */
    int32_t noise;
    uint64_t noise_seed;
    int64_t noise_group_offset;
    const int64_t* noise_group_id;
    /* != 0: config signal.polarization = 'custom' (simulation.py:821-825, calculate_polarization_vector): every ray's field is
       polarised along (0, sqrt(1 - ePhi^2), ePhi) / norm in the on-sky basis of its launch direction instead of l x (s x l).
       0: 'auto'. */
    int32_t custom_polarization;
    double polarization_ephi;
    /* != 0 (production mode with the plain OR of simple thresholds): the moment an event triggers, the convolution kernel
       evaluates ALL its channels and writes their traces into a compact buffer -- what the reference stores for triggered events
       (output_writer_hdf5.py:215-320 after channelReadoutWindowCutter) -- so that a survey needs no second pass over them.  Tables
       afterwards: "emit_offset" (int64 [n_events]: first sample of the event's [n_channels][L] block, -1 none, -2 the buffer was
       full) and "emit_trace" (double, the samples reserved); stats->n_emitted_events / n_emit_overflow tell whether every triggered
       event got its block (events decided by the chirp-z kernel -- common traces longer than 8192 samples, tabulated antennas,
       noise, the general path -- and coincidence triggers do not emit: run those through dump_traces).  emit_capacity_samples:
       size of the buffer (0: 4e8 samples = 3.2 GB, grown by the caller when n_emit_overflow > 0). */
    int32_t emit_triggered_traces;
    int64_t emit_capacity_samples;
    /* given ray solutions (ray_tracing.set_solution, analyticraytracing.py:2092-2116: launch parameters read back from a file
       instead of found): DEV double [n_showers * n_channels][2], NaN = no solution in that slot.  Non-NULL: the root search is
       skipped -- no hybr / Brent, no distance cut (the given rays have passed it) -- and the records of exactly these launch
       parameters (type, C1, path length, travel time, vectors) are made as for found ones; everything after the ray tracer is
       unchanged.  This is how the channel traces are compared with the reference's on the reference's OWN rays
       (tests/test_gpu_chain.py::test_reference_rays_through_the_batched_path).  Not with bottom reflections; the second trace
       of `focusing` still searches (the reference's does).  NULL: the finder runs.
       given_D / given_T (with given_C0 only; same layout, NULL or NaN entries = computed): path length [m] and travel time [ns] of the
       given rays, taken as given instead of evaluated from C0.  The reference's closed forms (analyticraytracing.py:602-783) take
       sqrt(n(z_turn)^2 - beta^2) of a difference that cancels completely at the turning point of a refracted ray; what is left is
       rounding noise of the libm in use, 1e-8 relative in D and T (measured against 60-digit arithmetic: either side is exact or
       1e-8 off, at random), i.e. 1e-4 rad of phase per GHz and km -- traces can be compared with the reference's at 1e-6 only on
       the reference's own D and T. */
    const double* given_C0;
    const double* given_D;
    const double* given_T;
} nrhip_sim_config;
#define NRHIP_TRIG_SIMPLE 0
#define NRHIP_TRIG_HIGH_LOW 1
#define NRHIP_TRIG_PHASED_ARRAY 2   /* needs nrhip_station_set_phased_array; trigger_threshold is the power threshold */
#define NRHIP_TRIG_ENVELOPE 3       /* envelopeTrigger.py: needs nrhip_station_set_envelope_trigger; Hilbert envelope of the band-passed
                                       channel trace > trigger_threshold, then the majority logic (n_coincidences, coinc_window) */

#define NRHIP_N_STAGES 9
/* stage_ms: device time (HIP events on the context's stream) of 0 ray tracing, 1 ray selection + setup,
 * 2 un-attenuated amplitude bound + active-ray list, 3 attenuation, 4 candidate cut (efield maximum),
 * 5 event grid (+ host hand-off), 6 per-length tables, 7 channel voltages + trigger, 8 whole call.       */
typedef struct {
    int64_t n_events, n_pairs, n_rays, n_candidate_events, n_triggered, n_channel_items, n_distinct_lengths;
    int64_t n_candidate_rays;
    int64_t n_active_rays;        /* rays that went through the attenuation quadrature */
    int64_t n_integrand_evals;    /* attenuation integrand evaluations (QUADPACK's neval summed over all items) */
    int64_t n_channel_transforms; /* (event, channel) items whose voltage trace was actually computed              */
    int64_t n_ray_transforms;     /* rays whose field went into one of those traces                                  */
    int64_t n_efield_transforms;  /* rays whose field was transformed for the candidate cut                          */
    int32_t max_length;
    int32_t n_sub_events;         /* readouts: event groups, or their sub-events with split_event_time_diff */
    double stage_ms[NRHIP_N_STAGES];
    /* nrhip_sim_config.emit_triggered_traces: events whose traces were written, events that found the buffer full, samples reserved */
    int64_t n_emitted_events, n_emit_overflow, n_emitted_samples;
    int64_t n_objective_evals;    /* calls of the ray finder's objective delta_y(log C0) (hybrd + Brent, summed over the pairs) */
    int64_t n_adc_convolution_flops;   /* FP64 operations of the chirp convolutions in the phased array's trigger-ADC chain (M (10 log2 M + 18) per M-point convolution) */
    /* general path (ARZ time-domain emission, birefringence): evaluations of the vector-potential integrand (ARZ.py:216-266, one per
       profile point and observer time), 1 m path steps whose records were made (analyticraytracing.py:2402-2413), and (path step,
       frequency bin) pairs the propagation applied (both rounds) */
    int64_t n_arz_evals, n_bire_steps, n_bire_step_bins;
    int64_t n_efield_sampled;     /* rays whose field was sampled next to the pulse centre for the candidate cut (efield_sample_kernel) */
} nrhip_sim_stats;

typedef struct nrhip_station nrhip_station;

/* A station belongs to its context: destroy every station BEFORE nrhip_ctx_destroy of that context. */
int nrhip_station_create(nrhip_ctx* ctx, const nrhip_station_desc* desc, nrhip_station** out);
void nrhip_station_destroy(nrhip_station* st);

/* The per-call tables of the last nrhip_simulate_events call (ray records, per-ray tables, traces: what nrhip_sim_fetch reads)
 * stay resident in the station object and are reused by the next call, and so does the cache of per-length tables (one row
 * of about 1 MB or more per distinct common-trace length the station has met; it only grows).  With many stations alive on one
 * GPU (an array simulated station by station), or after a long survey with many distinct lengths, release them: returns the
 * number of bytes given back (workspace + table cache; the next call rebuilds the table rows it needs). */
int64_t nrhip_station_release_workspace(nrhip_station* st);

/* Arrays of identical stations (BASELINE configs 3-5): move the station object to the next station's antenna positions
 * ([n_channels][3], HOST) and call nrhip_simulate_events again -- antenna types, orientations, cable delays, filters, tables
 * and the workspace are shared by all stations of the array, only the positions differ. */
int nrhip_station_set_positions(nrhip_station* st, const double* position);

/* ---- general emission / propagation inside nrhip_simulate_events (BASELINE config 4) -----------------------------------
 * With askaryan_model NRHIP_ASK_ARZ2019 / ARZ2020 and / or a birefringence model set, nrhip_simulate_events materialises
 * the on-sky spectra of every ray that passes the delta_C cut (calculate_sim_efield, simulation.py:221-290: emission,
 * polarisation, apply_propagation_effects = attenuation, Fresnel factors, birefringence), turns them into electric-field
 * traces and feeds those to the channel / trigger stage; the result-neutral pruning of the parametrised path does not
 * apply (every kept ray is evaluated).  Tables afterwards: "ray_spectra" [ray][2][N/2 + 1] complex, "ray_traces" [ray][2][N].
 *
 * nrhip_station_set_arz: the charge-excess profiles (as nrhip_arz_time_trace_batch: common depth grid, [n_profiles][n_depth]),
 * model parameters [2][7] (HAD, EM), interp_factor2, em_formula != 0: hadronic showers are scaled by ARZ.em_fraction
 * (ARZ2020).  nrhip_station_set_shower_profiles: per shower of the NEXT simulate call the profile row and the amplitude
 * factor E / E_library (the host picks them as ARZ.get_time_trace does: closest library energy, random / given number).
 * nrhip_station_set_birefringence: the three depth splines as in nrhip_birefringence_batch; n_knots == NULL switches it off.
 * Limits: simple threshold trigger, no focusing with ARZ, no amp_per_ray.  HOST pointers (copied).                      */
/* triggered_channels of the reference's threshold triggers (simpleThreshold.run :48-148, highLowThreshold.run :160-335): only
 * these channels raise flags / count towards the coincidence; the others are not evaluated at all unless traces are dumped
 * (their maxima then read NaN).  n = 0: every channel (the default).  HOST pointer (copied).                        */
int nrhip_station_set_trigger_channels(nrhip_station* st, int32_t n, const int32_t* channels);

/* Phased-array trigger (NuRadioReco/modules/phasedarray/phasedArrayBase.py, mode 'power_sum' without digitisation and
 * upsampling: phase_signals :183-215, power_sum :217-271, phased_trigger :455-496): per beam the traces of the n_pa trigger
 * channels are rolled by rolls[beam][channel] samples (calculate_time_delays :58-124, the host computes them) and summed, the
 * mean power of sliding windows (window samples, every step samples; averaging_divisor 0 = window) is compared with
 * nrhip_sim_config.trigger_threshold (trigger_type NRHIP_TRIG_PHASED_ARRAY).  Table afterwards: "pa_max_power"
 * [candidate event][beam] (maximum_amps).  n_pa = 0 switches it off.  HOST pointers (copied).                       */
/* The band pass of the envelope trigger (NuRadioReco/modules/trigger/envelopeTrigger.py:47-136 filters every channel with
 * channel.get_filtered_trace(passband, 'butter', order) before the Hilbert envelope): the analog Butterworth design as one rational
 * stage polyval(b, j f) / polyval(a, j f), highest power first, f in GHz (what nrhip_station_desc.filter_b / filter_a hold).
 * nb <= 0 switches it off. */
int nrhip_station_set_envelope_trigger(nrhip_station* st, int32_t nb, int32_t na, const double* b, const double* a);
/* Digitisation and up-sampling in front of the phased array (phasedArrayTrigger.run(apply_digitization=True, adc_kwargs=...,
 * upsampling_kwargs=dict(upsampling_method='fft', upsampling_factor=...)); call after nrhip_station_set_phased_array):
 * analogToDigitalConverter.get_digital_trace (:254-373) with the trigger ADC -- resampling to 5 GHz (5 GHz / sampling rate =
 * resample_p / resample_q, Fraction(...).limit_denominator(5000) as signal_processing.resample :71-108), linear interpolation at the
 * ADC sample times, perfect floor comparator with n_bits between v_min and v_max, output in volts (output_counts = 0) or ADC counts --
 * then signal_processing.digital_upsampling (:111-190, 'fft') by upsampling_factor (1: none); beams with rolls_up [n_beams][n_pa]
 * (whole-sample shifts at adc_sampling_frequency * upsampling_factor), saturation of count sums at saturation_bits (phase_signals
 * :183-215), window powers rounded for counts (power_sum :217-271).  adc_sampling_frequency <= 0 switches the digitisation off.
 * Tables afterwards: "pa_digital_trace" [n_candidates][n_pa][stride], "pa_digital_length" [n_candidates][n_pa], "pa_max_power".
 * n_bits = 0 (with adc_sampling_frequency = the simulation's rate): no digitisation -- phased_trigger(apply_digitization=False)
 * with upsampling_kwargs / mode 'hilbert_env' (:312-321): the analog channel traces take the same up-sampling, beams and envelope. */
int nrhip_station_set_phased_array_adc(nrhip_station* st, double adc_sampling_frequency, int32_t n_bits, double v_min, double v_max,
                                       int32_t output_counts, int32_t upsampling_factor, int32_t saturation_bits, int32_t resample_p,
                                       int32_t resample_q, const int32_t* rolls_up);
/* Other processing options of the digitised phased array (after nrhip_station_set_phased_array_adc, which resets them):
 * upsampling_method 0 = 'fft', 1 = 'lin' (np.interp on the two time grids, digital_upsampling :167-170), 2 = 'fir' (upsampling_fir
 * :192-234: zero stuffing and the low pass up_taps[n_up_taps] -- scipy.signal.firwin(filter_taps, f_adc / 2, fs = f_adc * factor),
 * rounded to 1 / coeff_gain and trimmed of zeros, as the caller designed it -- times the factor);
 * mode 0 = 'power_sum', 1 = 'hilbert_env' (PhasedArrayBase.hilbert_envelope :337-367 with ideal_transformer = False: the FIR
 * transformer hilbert_taps[n_hilbert_taps] on every beam, max + 3/8 min of (signal, transformed signal), rounded for ADC counts;
 * trigger_threshold then compares with the envelope and "pa_max_power" holds the envelope maxima), 2 = 'hilbert_env' with
 * ideal_transformer = True (:339-345: imag(scipy.signal.hilbert(beam)) as the circular convolution with its closed-form kernel,
 * envelope sqrt(beam^2 + imag^2); no taps). */
int nrhip_station_set_phased_array_processing(nrhip_station* st, int32_t upsampling_method, int32_t n_up_taps, const double* up_taps,
                                              int32_t mode, int32_t n_hilbert_taps, const double* hilbert_taps);
/* clock_offset of the phased-array trigger modules (phasedArrayTrigger.run(..., clock_offset=), phasedArrayTrigger.py:32,124, handed
 * to analogToDigitalConverter.get_digital_trace :327-340): the channel trace is delayed by clock_offset / adc_sampling_frequency in
 * front of the digitiser (signal_processing.delay_trace :401-472: phase ramp on the spectrum, the round(delay * sampling rate)
 * samples -- made even -- that wrapped round cut off).  Whole, non-negative clock cycles; after nrhip_station_set_phased_array_adc
 * (which resets it to 0). */
int nrhip_station_set_phased_array_clock_offset(nrhip_station* st, int32_t clock_offset);
/* `amplitude` of the noise adder per channel [n_channels] (simulation.py:596-600: Vrms / sqrt(norm / max_freq), norm = int |H|^2 df,
 * max_freq = sampling rate / 2; 0 = noiseless channel).  n <= 0 removes them. */
int nrhip_station_set_noise(nrhip_station* st, int32_t n, const double* amplitude);
int nrhip_station_set_phased_array(nrhip_station* st, int32_t n_pa, const int32_t* channels, int32_t n_beams,
                                   const int32_t* rolls, int32_t window, int32_t step, int32_t averaging_divisor);

int nrhip_station_set_arz(nrhip_station* st, int32_t n_profiles, int32_t n_depth, const double* profile_depth,
                          const double* profile_ce, const double* parameters, double interp_factor2, int32_t em_formula);
int nrhip_station_set_shower_profiles(nrhip_station* st, int64_t n_showers, const int32_t* profile_index, const double* rescale);
int nrhip_station_set_birefringence(nrhip_station* st, const int32_t* n_knots, const double* knots, const double* coeffs,
                                    double n_ref, double angle_to_iceflow);

/* The per-event hot path for single-shower event groups, in the order of simulation.run()
 * (NuRadioMC/simulation/simulation.py:1454-1600): for every channel calculate_sim_efield (:93-292: ray
 * tracing, delta_C cut, Askaryan spectrum, polarisation, attenuation, Fresnel, candidate cut), then
 * efieldToVoltageConverter.run (efieldToVoltageConverter.py:111-345) on the event's common time grid, the
 * filter chain and the trigger selected in cfg (simple threshold, high/low, n-fold coincidence).
 * Production mode decides candidate / trigger per event and skips whatever a rigorous bound or the OR-logic of the
 * decision makes unnecessary; cfg->dump_traces / cfg->no_pruning evaluate everything.
 * Event inputs are DEV pointers (resident in HBM): vertex [n][3], zenith / azimuth of the shower axis,
 * shower energy, shower_type (int32), k_L.  triggered is a DEV uint8 [n] mask.  stats is HOST (may be NULL). */
int nrhip_simulate_events(nrhip_ctx* ctx, nrhip_station* st, const nrhip_sim_config* cfg, int64_t n_events,
                          const double* vertex, const double* zenith, const double* azimuth,
                          const double* energy, const int32_t* shower_type, const double* k_L,
                          uint8_t* triggered, nrhip_sim_stats* stats);

/* The same for event groups of several showers (nu_e CC: hadronic + electromagnetic shower; secondary interactions):
 * calculate_sim_efield loops over the showers of the group per channel (simulation.py:143), every later step -- candidate
 * flag, common time grid, channel sums, trigger -- is per group.  Showers are given in group order; group_begin is a DEV
 * int32 [n_groups + 1] array of first-shower indices (NULL: one shower per group, n_groups == n_showers); vertex_time
 * [n_showers] DEV or NULL (0) enters the trace start time (simulation.py:259-268).  max_distance [n_showers] DEV or NULL:
 * speedup.distance_cut (simulation.py:155-163, :1398-1409) -- a shower farther from an antenna than this is skipped for
 * that channel (the host evaluates the energy polynomial, incl. the energy sum of neighbouring showers).
 * triggered: DEV uint8 [n_groups];
 * the ev_* / item_* tables of nrhip_sim_fetch are per group -- or per sub-event with cfg->split_event_time_diff > 0
 * (simulation.group_into_events); a readout whose common trace exceeds the supported length fails the call. */
int nrhip_simulate_event_groups(nrhip_ctx* ctx, nrhip_station* st, const nrhip_sim_config* cfg, int64_t n_showers,
                                const double* vertex, const double* zenith, const double* azimuth,
                                const double* energy, const int32_t* shower_type, const double* k_L,
                                const double* vertex_time, const double* max_distance, int64_t n_groups,
                                const int32_t* group_begin, uint8_t* triggered, nrhip_sim_stats* stats);

/* What the reference's output writer keeps of the channel traces of a triggered station event (output_writer_hdf5.py:215-320:
 * "maximum_amplitudes", "maximum_amplitudes_envelope", the trigger time), computed on the device from the traces the last call kept
 * (nrhip_sim_config.dump_traces) -- the traces themselves need not leave it.  Per item (candidate event of the last call, in
 * "item_event" order): trigger_bin = first sample of the first L - 1 with |V| >= threshold on any channel (the majority logic drops
 * the last sample, highLowThreshold.py:82-150; -1: none); per channel the read-out window of n_window samples starting pre_bins
 * before the trigger, cyclic (channelReadoutWindowCutter.run :28-137 with whole-sample shifts), its max |V| and the maximum of its
 * Hilbert envelope |scipy.signal.hilbert| (trace_utilities.get_hilbert_envelope).  n_window: a power of two, 16 .. 8192; items
 * whose common trace is shorter than the window, or without a trigger, get NaN.  HOST outputs [n_items], [n_items][n_channels]. */
int nrhip_readout_windows(nrhip_ctx* ctx, nrhip_station* st, int32_t n_window, int32_t pre_bins, double threshold,
                          int64_t n_items, int32_t* trigger_bin, double* max_amp, double* max_env);

/* Copy one intermediate table of the LAST nrhip_simulate_events call to HOST memory (parity tests,
 * output writers).  Names: ray_event ray_channel ray_solution ray_view ray_pol_theta ray_pol_phi ray_zenith
 * ray_azimuth ray_t0 ray_r_theta ray_r_phi ray_att ray_max_efield ray_C0 ray_D (int32 / double / complex);
 * ev_n_rays ev_L ev_candidate ev_t_min ev_trigger_bin; item_event item_maxV trace_offset trace;
 * ray_max_amp_envelope ray_signal_time (cfg->amp_per_ray).  ray_max_efield: > 0 exact, < 0 "at most" (not evaluated);
 * item_maxV: >= 0 exact, < 0 "at most", NaN not evaluated because an earlier channel of the event had triggered.
 * Returns the number of bytes available (>= 0) or < 0; copies min(bytes, available).                     */
int64_t nrhip_sim_fetch(nrhip_station* st, const char* name, void* host_dst, uint64_t bytes);

/* ---- per-event Earth-absorption weight (row f2) ----------------------------------------------------------------------
 * earth_attenuation.get_weight (NuRadioMC/utilities/earth_attenuation.py:12-60; one call per event group from
 * simulation.py:880-903) for n events at once; cross sections evaluated on the device: NRHIP_XS_CTW ('ctw',
 * NuRadioMC/utilities/cross_sections.py:64-120, :301-311) and NRHIP_XS_GHANDI ('ghandi', :280-281); interaction length
 * as :393-421.
 *   NRHIP_EARTH_SIMPLE                    get_simple_weight (:63-86)
 *   NRHIP_EARTH_CORE_MANTLE_CRUST_SIMPLE  get_core_mantle_crust_weight (:89-130)
 *   NRHIP_EARTH_CHORD                     'core_mantle_crust' / 'PREM' (:39-57): exp(-slant_depth / L_int), slant_depth =
 *                                         PREM.slant_depth(endpoint, direction, step) (:183-240) in the layered density
 *                                         model `model` (PREM.density :171-181)
 * endpoint [n][3]: vertex, z < 0 below the surface; direction [n][3]: hp.spherical_to_cartesian(zenith, azimuth) of the
 * arrival direction as the caller computed it (normalised on the device as :211 does) -- both only read in the chord
 * mode.  flavor: PDG code (< 0 antiparticle).  nucleon_mass = constants.m_p * units.kg.  weight [n] and slant_depth [n]
 * (chord mode only) may each be NULL.  'ctw' below 1e4 GeV gives NaN (cross_sections.py:69-76).  HOST pointers.
 * A cross section of 0 (what the reference's 'csms' returns for inttype='total') gives weight 1.               */
#define NRHIP_EARTH_SIMPLE 0
#define NRHIP_EARTH_CORE_MANTLE_CRUST_SIMPLE 1
#define NRHIP_EARTH_CHORD 2
#define NRHIP_EARTH_MAX_LAYERS 16
#define NRHIP_XS_CTW 0
#define NRHIP_XS_GHANDI 1
#define NRHIP_XS_GIVEN 2   /* `energy` holds the total cross section [m^2] of each event: tabulated models ('csms' :123-229,
                              'hedis_bgr18' :283-299, a data file) are evaluated by the caller                             */
typedef struct {
    int32_t n_layers;
    int32_t reserved;
    double earth_radius;
    double radii[NRHIP_EARTH_MAX_LAYERS];    /* layer k: radii[k-1] <= r < radii[k] (radii[-1] = 0); density 0 outside */
    double coef[NRHIP_EARTH_MAX_LAYERS][4];  /* rho(x) = ((c0 + c1 x) + c2 x^2) + c3 x^3, x = r / earth_radius          */
} nrhip_earth_model;
int nrhip_earth_weights_batch(nrhip_ctx* ctx, int64_t n, const double* zenith, const double* energy, const int32_t* flavor,
                              const double* endpoint, const double* direction, int32_t mode, int32_t cross_section_type,
                              const nrhip_earth_model* model, double step, double nucleon_mass, double* weight,
                              double* slant_depth);

/* efieldToVoltageConverter.run(evt, station, det) (NuRadioReco/modules/efieldToVoltageConverter.py:111-345) for ONE
 * station event on arbitrary electric fields -- the module-level drop-in.  efield e: time-domain traces
 * traces[e][0] = eTheta, traces[e][1] = ePhi (n_samples of the station each), trace start time t0[e] [ns], arrival
 * direction, channel index (position in the station).  The caller passes the common time grid of the event (t_min, L),
 * computed as the module does (:120-169).  V receives [n_channels][L] voltage traces (channels without efield: zeros).
 * apply_filters != 0 additionally applies the station's filter chain (what a following channelBandPassFilter.run does).
 * All pointers HOST.  n_samples <= 4096.                                                                          */
int nrhip_efield_to_voltage(nrhip_ctx* ctx, nrhip_station* st, int32_t n_efields, const double* traces,
                            const double* t0, const double* zenith, const double* azimuth, const int32_t* channel,
                            int32_t apply_filters, int32_t L, double t_min, double* V);

/* ---- multi-GPU: one process per GPU, events sharded, ONE collective ------------------------------
 * The reference scales out as N independent processes over a split event list (NuRadioMC/utilities/runner.py:9-15) and merges
 * their output files afterwards (utilities/merge_hdf5.py); here rank r of W simulates its contiguous slice of the list and the
 * per-rank uint8 triggered masks are all-gathered over xGMI -- RCCL, bound directly (librccl.so is opened on first use).
 * nrhip_comm_get_unique_id: rank 0 makes the communicator id and hands the 128 bytes to the other processes through any
 * host channel (nuradiomc_amd/comm.py: a TCP socket on MASTER_ADDR); nrhip_comm_create: collective over all ranks, on the
 * context's GPU and stream.  Buffers are DEV pointers of that GPU; calls are asynchronous on the context's stream
 * (nrhip_synchronize), nrhip_comm_barrier returns when every rank's stream has drained.
 * nrhip_comm_allgather_u8: recv[r * count_per_rank + i] = send_r[i] (pad unequal shards to the largest one).          */
#define NRHIP_COMM_ID_BYTES 128
typedef struct nrhip_comm nrhip_comm;
int nrhip_comm_get_unique_id(uint8_t id[NRHIP_COMM_ID_BYTES]);
int nrhip_comm_create(nrhip_ctx* ctx, const uint8_t id[NRHIP_COMM_ID_BYTES], int32_t rank, int32_t world_size, nrhip_comm** out);
void nrhip_comm_destroy(nrhip_comm* comm);
int nrhip_comm_barrier(nrhip_comm* comm);
int nrhip_comm_allgather_u8(nrhip_comm* comm, const uint8_t* send, uint8_t* recv, int64_t count_per_rank);
int nrhip_comm_allreduce_i64_sum(nrhip_comm* comm, int64_t* buf, int32_t n);
int nrhip_comm_allreduce_f64_max(nrhip_comm* comm, double* buf, int32_t n);

/* ---- station-level selection of event groups for arrays (BASELINE configs 3-5) ------------------------------------------------
 * With speedup.distance_cut a shower is skipped for a channel when |vertex - antenna| > max_distance[shower]
 * (simulation.py:155-163); a group none of whose showers lies within max_distance + radius of the station centre (radius =
 * largest |relative antenna position|) is therefore skipped for every channel of the station -- the quick cut the reference
 * has at :1503-1509.  nrhip_cull_groups lists the groups in range (keep_index [n_keep], ascending; group_begin_out
 * [n_keep + 1] = first shower of every kept group in the gathered list), nrhip_gather_groups copies their showers into compact
 * arrays (any src / dst pair may be NULL; shower_index [n_showers_out], optional: source row of every gathered shower),
 * nrhip_simulate_event_groups runs on those, nrhip_mask_scatter_or ORs the compact triggered mask back:
 * dst[index[k]] |= src[k].  All array pointers DEV; centre, n_keep, n_showers_out HOST.  group_begin NULL: one shower per
 * group. */
int nrhip_cull_groups(nrhip_ctx* ctx, int64_t n_showers, int64_t n_groups, const int32_t* group_begin, const double* vertex,
                      const double* max_distance, const double centre[3], double radius, int32_t* keep_index,
                      int32_t* group_begin_out, int64_t* n_keep, int64_t* n_showers_out);
/* the same lists for the groups flagged in a DEV uint8 mask [n_groups] -- e.g. the triggered groups of a survey, whose showers
 * nrhip_gather_groups then copies into a compact list for the second pass that stores what the reference writes for triggered
 * events (all channel traces: nrhip_sim_config.dump_traces; output_writer_hdf5.py:215-320) without a host round trip of the list */
int nrhip_select_groups(nrhip_ctx* ctx, int64_t n_showers, int64_t n_groups, const int32_t* group_begin, const uint8_t* mask,
                        int32_t* keep_index, int32_t* group_begin_out, int64_t* n_keep, int64_t* n_showers_out);
int nrhip_gather_groups(nrhip_ctx* ctx, int64_t n_keep, const int32_t* keep_index, const int32_t* group_begin,
                        const int32_t* group_begin_out, const double* vertex, const double* zenith, const double* azimuth,
                        const double* energy, const int32_t* shower_type, const double* k_L, const double* vertex_time,
                        const double* max_distance, double* o_vertex, double* o_zenith, double* o_azimuth, double* o_energy,
                        int32_t* o_shower_type, double* o_k_L, double* o_vertex_time, double* o_max_distance,
                        int32_t* shower_index);
int nrhip_mask_scatter_or(nrhip_ctx* ctx, int64_t n, const int32_t* index, const uint8_t* src, uint8_t* dst);
/* out[k] = offset + index[k] (index NULL: offset + k) as DEV int64: the original group ids of a gathered list, e.g. for
 * nrhip_sim_config.noise_group_id (the noise of an event group must not depend on which compact list it travels in) */
int nrhip_index_to_i64(nrhip_ctx* ctx, int64_t n, const int32_t* index, int64_t offset, int64_t* out);
/* out[k] = src[index[k]] (DEV int64): caller-given group ids (nrhip_sim_config.noise_group_id) of a gathered list */
int nrhip_gather_i64(nrhip_ctx* ctx, int64_t n, const int32_t* index, const int64_t* src, int64_t* out);

/* dst[i] = overwrite ? src[i] : dst[i] | src[i] on DEV uint8 masks (event-group mask of an array = OR over its stations) */
int nrhip_mask_or(nrhip_ctx* ctx, int64_t n, uint8_t* dst, const uint8_t* src, int32_t overwrite);

/* test hook for the in-LDS chirp-z transform: out[b][k] = sum_j in[b][j] exp(sgn 2 pi i j k / Q). HOST. */
int nrhip_debug_czt(nrhip_ctx* ctx, int32_t n_batch, int32_t n_in, int32_t n_out, int32_t Q, double sgn,
                    const double* in, double* out);
/* test hook for the wave-level sums of the bound kernels (csrc/wave_reduce.h: v_permlane32_swap / v_permlane16_swap / DPP adds):
 * in HOST [n_waves][8][64], out HOST [n_waves][273] = 8 sums in FP64, the same 8 in FP32, the sum of value 0 as every lane sees it
 * (FP64, FP32), value 1 of the lane below per lane, value 1 of lane 63, value 2 of lane - 32 per lane. */
int nrhip_debug_wave_sums(nrhip_ctx* ctx, int32_t n_waves, const double* in, double* out);

#ifdef __cplusplus
}
#endif
#endif /* NRHIP_H */
