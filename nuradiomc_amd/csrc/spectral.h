// spectral.h -- device-side tables of the spectral stages (shared by spectral.hip and pipeline.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "nrhip_internal.h"
#include "fft_device.h"

#define NRHIP_MAX_NFC 64        // coarse attenuation frequencies per ray (n_freq + n_freq / 2 <= 64)
#define NRHIP_MAX_FILTERS 4
#define NRHIP_MAX_POLY 24
#define NRHIP_MAX_FSETS 4       // distinct filter chains per station (channels sharing an amplifier type share a set)
#define NRHIP_SPEC_STRIDE 16384  // max L / 2 + 1 spectrum bins per channel: common traces of at most 32766 samples
#define NRHIP_E_STRIDE 65536    // 2 L phase-table entries per length
#define NRHIP_N_ANT_TAB 5       // antenna response tables per length: VPol, HPol, LPDA front / side / back lobe phase
#define NRHIP_G_STRIDE 8200     // FFT_MAX + 1 bins of the 2 FFT_MAX-point real transform of the impulse response (padded)

#include "noise.h"

namespace nrhip {

// tabulated antenna pattern in HBM (nrhip_antenna_table)
struct AntTabDev {
    int nF, nT, nP;
    const double *fr, *th, *ph;
    const double2 *vt, *vp;  // flat index iF * nT * nP + iP * nT + iT
};

// station description in HBM (small, read through the scalar / L1 caches)
struct StationDev {
    int n_ch, N, n_fc;
    double fs, pre_pulse, post_pulse, readout_length, att_bound_depth;
    const double* pos;        // [n_ch][3]
    const double* cable;      // [n_ch]
    const unsigned char* trig_on;  // [n_ch] 1 = the channel takes part in the threshold triggers; nullptr = all do
    const int* ant_model;     // [n_ch]   0 analytic_VPol, 1 analytic_HPol, 2 analytic_LPDA
    const AntTabDev* ant_tabs;   // tabulated patterns (ant_model 3): table of channel c = ant_tabs[ant_tab_index[c]]
    const int* ant_tab_index;    // [n_ch]
    int max_tab_freq;            // largest n_freq of the tables (scratch sizing)
    int tab_mask;             // bit t set: antenna table t (NRHIP_N_ANT_TAB) is needed by some channel
    int n_fsets;              // filter chains of the station (>= 1)
    const int* ch_fset;       // [n_ch] filter chain of every channel (nullptr: chain 0 for all)
    int fset_tab_mask[NRHIP_MAX_FSETS];  // tab_mask restricted to the channels of one chain
    const double* rot;        // [n_ch][9] inv(E) A   (antennapattern.py:1190-1216)
    const double* rot_inv;    // [n_ch][9]
    const double* fcoarse;    // [n_fc] attenuation frequency grid
    const double* lnf;        // [N/2 + 1] ln f_k of the N-sample grid (entry 0 unused)
    const double* inv_lmax;   // [n_fc] 1 / max_z L_att(z, f) (0 = unknown): upper bound exp(-0.95 D / L_max) on attenuation
    int n_att_bins;           // depth-binned attenuation bound (0: not given)
    double att_bin_width;
    const double* att_bin_inv; // [n_att_bins][n_fc] lower bounds of 1 / L_att inside depth bin b
    const double* fpow;       // [3][N/2 + 1] f_k^p for p = 2.57, 2.74, 1.27 (Alvarez2009: beta had / em, alpha)
    const float* fpow_f;      // the same in single precision (bound kernels)
    const unsigned char* seg; // [N/2 + 1] coarse-grid segment lo of f_k: fcoarse[lo] <= f_k < fcoarse[lo + 1]
    NPlan np;                 // the N / 2-point transforms of the ray stages (any even N)
};

// analog filter chain: response_i(f) = polyval(b_i, j f) / polyval(a_i, j f), highest power first
struct FilterSet {
    int n;
    // 0 rational, 1 |rational|, 2 rectangular pass band b[0] <= f <= b[1],
    // 3 tabulated (gain, unwrapped phase) on nb grid points starting at pool[3 * na], linear interpolation, 0 outside, times
    //   the correction b[0] + b[1] f^5 (RNO_G/analog_components.load_amp_response),
    // 4 gaussian_tapered: pass band b[0] .. b[1] convolved with a Gaussian of sigma b[2] on the grid it is asked for
    int kind[NRHIP_MAX_FILTERS];
    const double* pool;  // tabulated responses of the station: (f [GHz], gain, phase [rad]) triples
    int nb[NRHIP_MAX_FILTERS], na[NRHIP_MAX_FILTERS];
    double b[NRHIP_MAX_FILTERS][NRHIP_MAX_POLY], a[NRHIP_MAX_FILTERS][NRHIP_MAX_POLY];
};

// Askaryan emission constants of one ray (everything of the parametrisation that does not depend on frequency)
struct AskaryanConst {
    int model;      // 0 Alvarez2009, 1 Alvarez2000, 2 ZHS1992
    int had;
    double a_pref;  // everything that multiplies f
    double nu_L, beta, nu_R, alpha;  // Alvarez2009
    double dth, cher, theta, f0, scale, roll;  // Alvarez2000 / ZHS1992
    double ln_nu_L, ln_nu_R;
    double cL, cR, pref2;  // Alvarez2009 on the station's frequency grid: nu_L^-beta, nu_R^-alpha, a_pref / (2 scale)
};

// per kept ray (SoA, ordered by event, channel, solution)
struct RayWork {
    AskaryanConst* ask;  // [n]
    int *ev, *ch, *sol, *slot;
    double *view, *n_index, *R, *t0, *C0;
    double *pol_theta, *pol_phi;
    double2 *r_theta, *r_phi;
    double *zen, *az;
    double *vel_T;       // [n][4]
    double *theta_ant, *phi_ant;
    double *vfac_t, *vfac_p;  // [n] weight of the on-sky eTheta / ePhi field in the channel voltage (direction + frame)
    int *tab;            // [n] antenna response table of the ray (0 VPol, 1 HPol, 2..4 LPDA phase regime)
    double *att;         // [n][n_fc]
    double *e_norm;      // [n] L2 norm of the unit-polarisation field trace, sqrt(sum_t s(t)^2) (set by efield_max_kernel)
    double *focus;       // [n] focusing factor of the ray (1 without focusing): the parametrisations carry it in a_pref, the
                         //     time-domain emission models multiply their spectrum with it (analyticraytracing.py:3011-3016)
};

struct EventIn {
    const double* energy;
    const int* shower_type;  // 0 HAD, 1 EM
    const double* k_L;       // Alvarez2009 EM showers; ignored otherwise
    const double* vertex_time;  // [n_showers] or nullptr (0)
};

struct EventOut {
    int *n_rays, *ray_begin, *L;
    unsigned char* candidate;
    double* t_min;
};

struct LengthTables {
    double2* B_fwd;  // [n_len][FFT_MAX]
    double2* B_inv;  // [n_len][FFT_MAX]
    double2* vel;    // [n_len][NRHIP_N_ANT_TAB][NRHIP_SPEC_STRIDE]  analytic antenna response on the L grid (0 below 5 MHz)
    double2* E;      // [n_len][NRHIP_E_STRIDE]        exp(-2 pi i j / (2 L)), j < 2 L: every chirp / phase factor
    double2* H;      // [n_len][n_fsets][NRHIP_SPEC_STRIDE]  filter chain response on the L grid
    double2* Cf;     // [n_len][NRHIP_SPEC_STRIDE]     forward chirp exp(-i pi k^2 / (L/2)), contiguous in k
    double2* Ci;     // [n_len][FFT_MAX]               inverse chirp exp(+i pi n^2 / L), contiguous in n
    double* hnorm;   // [n_len][n_fsets][NRHIP_N_ANT_TAB]          L2 norm of the (antenna x filter) impulse response on the L grid
    double2* G;      // [n_len][n_fsets][NRHIP_N_ANT_TAB][NRHIP_G_STRIDE]  (L <= FFT_MAX) spectrum on the 2 FFT_MAX grid of the L-periodic impulse
                     //                                response irfft_L(antenna x filter), all scale factors folded in
};

// trigger logic of the station (nrhip_sim_config), times in samples
struct TriggerDev {
    int type;            // 0 simple threshold, 1 high/low, 2 Hilbert envelope of the band-passed trace (decided on dumped traces only)
    int n_coinc;         // channels required inside the coincidence window
    double threshold;    // simple: |V| >= threshold
    double high, low;    // high/low: a sample >= high and a sample <= low inside w_hl samples
    int w_hl, w_coinc;   // window lengths [samples]
    __host__ __device__ bool coincidence() const { return type != 0 || n_coinc > 1; }  // anything but the plain OR of simple thresholds
    __host__ __device__ double prefilter() const { return type == 0 ? threshold : (high > -low ? high : -low); }
};

struct ChannelOut {
    int* trigger_bin;           // [n_events] first triggered sample of the event's common trace (-1: none / not evaluated)
    double* maxV;               // [n_items]
    unsigned char* triggered;   // [n_events]
    double* trace;              // optional dump
    const long* trace_offset;   // [n_items]
    // production mode, plain OR of thresholds: the traces of ALL channels of the events that trigger, written by the convolution
    // kernel the moment an event triggers (compact buffer, space reserved through a cursor) -- what the reference stores for
    // triggered events -- instead of a second pass over them
    double* emit = nullptr;             // [emit_cap] samples
    long long* emit_offset = nullptr;   // [n_events] first sample of the event's n_ch x L block (-1: none, -2: the buffer was full)
    unsigned long long* emit_cursor = nullptr;   // [3]: samples reserved, events that did not fit, events written
    long long emit_cap = 0;
};

void launch_select_rays(hipStream_t s, long n_pairs, int n_ch, const double* vertex, const double* zen, const double* az,
                        const RayRecords& rec, const IceConst& m, double cut, int* keep);
long scan_tiles(long n);
void launch_exclusive_scan(hipStream_t s, long n, const int* in, int* out, int* tile_tmp);
void launch_scatter_slots(hipStream_t s, long n_slots, const int* keep, const int* offset, int* ray_slot);
void launch_ray_setup(hipStream_t s, int n_rays, int n_ch, const int* ray_slot, const double* vertex, const double* zen,
                      const double* az, const RayRecords& rec, const IceConst& m, const StationDev& st, const RayWork& w,
                      const EventIn& evin, int ask_model, const int* foc_n_sol = nullptr, const double* foc_launch = nullptr,
                      double foc_dz = 0., double foc_limit = 0., double refl_coefficient = 1., double refl_phase = 0.,
                      double pol_ephi = NAN /* signal.polarization 'custom': ePhi; NaN = 'auto' */);
void launch_gather_segments(hipStream_t s, int n_rays, int NS, const int* ray_slot, const double* seg_C0, const double* seg_zint,
                            double* ray_seg_C0, double* ray_seg_zint);
void launch_segment_items(hipStream_t s, int n_active, int NS, const int* active_list, int* items);
void launch_segment_product_rays(hipStream_t s, int n_active, int NS, int n_fc, const int* active_list, const double* ray_seg_C0,
                                 const double* seg_att, double* att);
void launch_distance_cut_pairs(hipStream_t s, long n_pairs, int n_ch, const double* vertex, const double* pos, const double* max_dist,
                               int* n_sol);
void launch_ray_limits_from_slots(hipStream_t s, int n_rays, int n_ch, const int* ray_slot, const double* vertex,
                                  const double* chan_pos, const RayRecords& rec, const IceConst& m, double* zint);
void launch_amp_bound(hipStream_t s, int n_rays, const RayWork& w, const StationDev& st, const IceConst& m,
                      const double* vertex, const double* zint, double* bound, double* max_efield, double cut);
void launch_group_ray_range(hipStream_t s, int n_groups, const int* group_begin, int n_ch, const int* slot_offset, int* grp_ray,
                            int stride = NRHIP_MAXS);
void launch_event_possible(hipStream_t s, int n_events, int n_ch, const int* slot_offset, const double* bound,
                           double min_efield, int* ray_active, int own_only = 0);
void launch_follower_list(hipStream_t s, int n_ev, const EventOut& ev, const double* att, int n_fc, int n_rays, int* flag, int* offset,
                          int* scan_tmp, int* list, int* ray_active);
void launch_efield_bound_list(hipStream_t s, int n_list, const int* list, const RayWork& w, const StationDev& st, double min_efield,
                              double* max_efield, int* need_scratch);
void launch_scatter_active(hipStream_t s, int n_rays, const int* active, const int* offset, int* list);
void launch_active_class_flags(hipStream_t s, int n_rays, const int* active, const int* ray_slot2, const int* slot_type,
                               int* flags);
long quad_class_entries(int n_rays);
void launch_quad_class_list(hipStream_t s, int n_rays, const int* active, const int* ray_slot2, const int* slot_type, const double* C0,
                            const double* zint, const IceConst& m, signed char* cls, int* counts, int* offset, int* scan_tmp, int* list);
void launch_scatter_active_class(hipStream_t s, int n_rays, const int* flags, const int* offset, int* list);
void launch_efield_max(hipStream_t s, int n_active, const int* active_list, int n_rays, int n_events,
                       const int* slot_offset, const RayWork& w, const EventIn& evin, const StationDev& st, int ask_model,
                       const double2* tw, double min_efield, int exact, double* max_efield, int* need_ray, int* ev_need,
                       int* ev_offset, int* scan_tmp, int* ev_list, unsigned long long* xform_count, double* amp_scratch = nullptr);
void launch_event_grid(hipStream_t s, int n_events, int n_ch, const int* slot_offset, const RayWork& w, const StationDev& st,
                       const double* max_efield, double min_efield, const EventOut& ev);
void launch_candidate_flags(hipStream_t s, int n_events, int n_half, const EventOut& ev, int* cflag, int* lflag,
                            long long* n_cand_rays);  // n_cand_rays[2]: rays in candidate events, largest L / 2 >= n_half
void launch_candidate_lists(hipStream_t s, int n_events, int n_half, const EventOut& ev, const int* cflag, const int* coff,
                            const int* lflag, const int* loff, int* cand, int* len_index, int* lens);
void launch_length_tables(hipStream_t s, int n_len, const int* lengths, const StationDev& st, const FilterSet* fls,
                          const double2* tw, const double2* w16, const LengthTables& tab, const int* slots = nullptr);
void launch_length_slots(hipStream_t s, int n_events, const int* ev_L, const int* slotmap, int* len_index);   // fls: DEV [st.n_fsets]
int channel_grid_blocks();
void launch_general_spectrum(hipStream_t s, int n_rays, const RayWork& w, const StationDev& st, int ask_model,
                             const double* arz_trace, const double2* tw, double2* spec, double* amp_scratch = nullptr,
                             const int* silent = nullptr);
void launch_general_trace(hipStream_t s, int n_rays, const StationDev& st, const double2* spec, const double2* tw,
                          double* traces, double* max_efield, const int* active = nullptr, const double* bound = nullptr);
void launch_general_bound(hipStream_t s, int n_rays, const StationDev& st, const double2* spec, const long long* log_gain,
                          double* bound, double* e_norm = nullptr);
void launch_general_prefilter(hipStream_t s, int n_items, const int* item_event, const RayWork& w, const EventOut& ev,
                              const int* ev_len_index, const StationDev& st, double threshold, const double* hnorm, double* maxV,
                              int* need, int n_rays, int* propagated, int* fresh);
void launch_general_gather(hipStream_t s, int n_rays, int n_ch, const RayWork& w, const EventIn& evin, const StationDev& st,
                           const double* vertex, const int* shower_profile, const double* shower_rescale, int em_formula,
                           double* energy, int* type, double* em_factor, int* profile, double* rescale, double* x1, double* x2,
                           int* n_steps, int* n_points);
void launch_silent_rays(hipStream_t s, int n_rays, const double* view, const double* n_index_ray, double n_index, double maximum_angle,
                        int* n_steps, int* n_points);
void launch_int_to_long(hipStream_t s, int n, const int* in, long* out);
void launch_phased_array(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                         const long* trace_offset, int n_pa, const int* pa_channel, int n_beams, const int* rolls, int window,
                         int step, double divisor, double threshold, int max_length, unsigned char* triggered, double* pa_max);
void launch_channel(hipStream_t s, int n_items, const int* item_event, const RayWork& w, const EventIn& evin,
                    const EventOut& ev, const int* ev_len_index, const StationDev& st, const FilterSet& fl, int ask_model,
                    const TriggerDev& trig, const double2* tw, const double2* w16, const LengthTables& tab, double2* scratch,
                    const ChannelOut& out, int exact, int max_length, int* need, int* need_offset, int* scan_tmp,
                    int* item_list, int* coinc_cnt, double2* conv_acc, unsigned long long* xform_count, double2* tab_nodes,
                    const double* ray_traces = nullptr, int skip_off = -1, const FilterSet* envf = nullptr, double* env_trace = nullptr,
                    const NoiseDev* noise = nullptr, bool conv_split = true, double pa_amp_cut = -1., double* amp_scratch = nullptr, double* noise_buf = nullptr,
                    const int* item_need = nullptr, void* conv_ws = nullptr);
// bytes of launch_channel's conv_ws: one 112-byte record per candidate event for the convolution kernel, then the counting sort of
// the event list by trace length (histogram, cursors, scan scratch, sorted list).  Required when the convolution kernel runs.
inline size_t conv_ws_bytes(long n_cand) { return (size_t)n_cand * 112 + 4 * (size_t)(2 * (FFT_MAX / 2 + 2) + scan_tiles(FFT_MAX / 2 + 2) + n_cand + 8) + 64; }
// channel_kernel's amplitude table lives in HBM scratch (rows of N / 2 + 1 doubles per block) when N > 4096
inline bool channel_amp_in_hbm(int n_samples) { return n_samples / 2 > 2048; }
// efield_max_kernel / general_spectrum_kernel: N / 2 no power of two and above 2048 -- the Bluestein transform takes FFT_MAX points,
// 128 KB of LDS; the amplitude tables of at most RAY_AMP_ROWS blocks then sit in HBM scratch
#define RAY_AMP_ROWS 1024
inline bool ray_amp_in_hbm(int n_samples) { const int nh = n_samples / 2; return nh > 2048 && ((nh & (nh - 1)) != 0 || nh > 4096); }
// trigger ADC + up-sampling of the phased array (pa_digitize_kernel)
struct PaAdc {
    double adc_fs, vmin, vmax;   // ADC sampling rate [GHz], voltage range
    int n_bits, counts, upsampling, saturation_bits, p, q, stride;  // 5 GHz / f_s = p / q; stride: samples per output trace
    int up_method = 0, n_up_taps = 0;     // 0 'fft', 1 'lin', 2 'fir' (taps up_taps)
    int mode = 0, n_hil_taps = 0;         // 0 'power_sum', 1 'hilbert_env' (FIR transformer hil_taps)
    const double *up_taps = nullptr, *hil_taps = nullptr;   // device
    int clock_offset = 0;                 // whole ADC clock cycles the trace is delayed by in front of the digitiser (>= 0)
};
// 'lin' / 'fir' up-sampling of ADC traces [n_items][stride_in] (lengths len_in) into pa_trace [n_items][adc.stride]
void launch_pa_upsample(hipStream_t s, int n_items, const PaAdc& adc, const double* adc_trace, int stride_in, const int* len_in,
                        double* pa_trace, int* pa_len);
// the chirp-z version of the same chain (O(L log L) transforms; tables per trace length in the station's cache)
constexpr int PA_TABLES = 12;  // rows of FFT_MAX complex numbers per length slot: 6 Bluestein spectra, 6 chirps (two of each for the clock offset only)
bool pa_czt_applies(int max_length, double fs, const PaAdc& adc);
size_t pa_czt_work_bytes(int max_length, double fs, const PaAdc& adc, int chunk);
void launch_pa_czt_tables(hipStream_t s, int n_len, const int* lens, const int* slots, double fs, const PaAdc& adc, const double2* tw,
                          double2* Btab);
void launch_phased_array_digital_czt(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const int* slotmap,
                                     const double* trace, const long* trace_offset, int n_pa, const int* pa_channel, int n_beams,
                                     const int* rolls_up, int window, int step, double divisor, double threshold, int max_length, double fs,
                                     const PaAdc& adc, const double2* tw, const double2* cft, const double2* Btab, void* work, int chunk, double* pa_trace,
                                     int* pa_len, unsigned char* triggered, double* pa_max, bool with_beams = true,
                                     unsigned long long* conv_count = nullptr);
void launch_phased_array_beams(hipStream_t s, int n_cand, const int* item_event, int n_pa, int n_beams, const int* rolls_up, int window,
                               int step, double divisor, double threshold, const PaAdc& adc, const double* pa_trace, const int* pa_len,
                               unsigned char* triggered, double* pa_max);
void launch_phased_array_digital(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                                 const long* trace_offset, int n_pa, const int* pa_channel, int n_beams, const int* rolls_up, int window,
                                 int step, double divisor, double threshold, int max_length, double fs, const PaAdc& adc,
                                 double* pa_trace, int* pa_len, unsigned char* triggered, double* pa_max, bool with_beams = true);
void launch_trace_trigger(hipStream_t s, int n_cand, const int* item_event, int n_ch, const int* ev_L, const double* trace,
                          const long* trace_offset, const TriggerDev& trg, const unsigned char* trig_on, int max_length,
                          unsigned char* triggered, int* trigger_bin);
void launch_readout_windows(hipStream_t s, int n_items, int n_ch, const int* item_event, const int* ev_L, const double* trace,
                            const long* trace_offset, int n_window, int pre_bins, double threshold, const double2* tw,
                            int* trigger_bin, double* max_amp, double* max_env);
void launch_ray_envelope(hipStream_t s, int n_cand_max, const int* n_cand, const int* item_event, const RayWork& w,
                         const EventOut& ev, const StationDev& st, int ask_model, const double2* tw, const LengthTables& tab,
                         const int* len_index_N, double* max_env, double* signal_time, const double2* spec = nullptr, double2* tab_nodes = nullptr, double* amp_scratch = nullptr);
void launch_efield_channel(hipStream_t s, int n_efields, const double* traces, const double* t0, const double* zen,
                           const double* az, const int* channel, const StationDev& st, int L, double t_min, int apply_filter,
                           const double2* tw, const LengthTables& tab, double2* scratch, double* V, double2* tab_nodes);
void launch_askaryan_spectrum(hipStream_t s, int n, const double* energy, const double* theta, const int* type,
                              const double* n_index, const double* R, const double* k_L, int model, int N, double dt,
                              double2* spec);
void launch_nplan_tables(hipStream_t s, int nh, int log2p, double2* wN, double2* cw, double2* Bf, double2* Bi, const double2* tw);
void launch_czt_test(hipStream_t s, int n_batch, int n_in, int n_out, int Q, double sgn, const double2* in, double2* out,
                     const double2* tw, double2* Bscratch, int grid);
#define WAVE_TEST_OUT 273
void launch_wave_reduce_test(hipStream_t s, int n_waves, const double* in, double* out);
void launch_attenuation_items(hipStream_t stream, long n_rays, const double* C0, const double* zint, int n_freq,
                              const double* freqs, int model, const IceConst& m, double* att, int* neval,
                              const int* ray_index = nullptr, unsigned long long* eval_counter = nullptr,
                              const double* gl3 = nullptr, int gl3_n = 0, int* overflow = nullptr);

}  // namespace nrhip
