// ctx.h -- the opaque context / station objects behind include/nrhip.h
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "spectral.h"

// grow-only device buffer
struct DevArray {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct nrhip_ctx {
    int device;
    hipStream_t stream;
    nrhip::IceConst ice;
    int att_model;
    double2* twiddle = nullptr;  // exp(-2 pi i k / FFT_MAX), k < FFT_MAX / 2
    double* gl3 = nullptr;       // GL3 depth table [3][gl3_n] (depth, slope, offset), nrhip_ctx_set_gl3_table
    int gl3_n = 0;
    long last_att_overflow = 0;  // rays the dense quadrature kernel left to the general one (last nrhip_attenuation_batch)
    double2* w16 = nullptr;      // exp(-2 pi i k / (2 FFT_MAX)), k <= FFT_MAX / 2 (real <-> packed-complex FFT split)
    std::set<struct nrhip_station*> stations;  // alive stations: nrhip_ctx_destroy releases what they hold on the GPU
    DevArray cull_ws;            // scratch of nrhip_cull_groups (flags, sizes, scans)
    int ray_finder = 0;          // NRHIP_FINDER_TRUE_ROOTS | NRHIP_FINDER_REFERENCE (nrhip_ctx_set_ray_finder)
};

// frees the station's device memory and events and detaches it from its context (the host object stays until
// nrhip_station_destroy): whichever of the two destroy calls comes first, nothing dangles
extern "C" void nrhip_station_detach(struct nrhip_station* s);

int nrhip_fail(const char* what, hipError_t e);
int nrhip_fail_msg(const char* what);
#define HIPCHK(x)                                         \
    do {                                                  \
        hipError_t e_ = (x);                              \
        if (e_ != hipSuccess) return nrhip_fail(#x, e_);  \
    } while (0)

struct nrhip_station {
    nrhip_ctx* ctx;
    nrhip::StationDev dev;
    nrhip::FilterSet filters[NRHIP_MAX_FSETS];
    DevArray d_filter_pool, d_ch_fset, d_filtersets;  // tabulated responses, per-channel chain index, the chains in HBM
    std::vector<double> h_pos, h_cable;
    DevArray d_pos, d_cable, d_model, d_rot, d_rot_inv, d_fc, d_lnf, d_invl, d_fpow, d_fpow_f, d_seg, d_attbin, d_anttabs, d_anttab_index;
    DevArray d_nplan;   // tables of the N / 2-point Bluestein transforms (N / 2 not a power of two)
    std::vector<DevArray> d_tabdata;  // arrays of the tabulated antenna patterns
    // general emission / propagation path (nrhip_station_set_arz / _set_birefringence / _set_shower_profiles)
    DevArray d_arz_depth, d_arz_ce, d_arz_par, d_bire_knots, d_bire_coeffs, d_shower_profile, d_shower_rescale;
    int arz_n_profiles = 0, arz_n_depth = 0, arz_em_formula = 0;
    double arz_interp_factor2 = 100.;
    int bire_n_knots[3] = {0, 0, 0};
    double bire_n_ref = 1.78, bire_angle = 0.;
    int64_t n_shower_profiles = 0;
    // phased-array trigger (nrhip_station_set_phased_array)
    DevArray d_pa_channel, d_pa_rolls, d_pa_mask, d_trig_on;
    // per-length tables (Bluestein chirps, antenna / filter responses on the L grid, impulse-response spectra) are station
    // constants: built once per distinct trace length and kept for the station's lifetime (like FFT plans); slot_of[L / 2] = row
    struct LengthTableCache {
        DevArray B_fwd, B_inv, vel, E, H, Cf, Ci, hnorm, G, slotmap;
        std::vector<int> slot_of;
        int n_slots = 0, cap = 0;
        void release()
        {
            B_fwd.release(); B_inv.release(); vel.release(); E.release(); H.release(); Cf.release(); Ci.release(); hnorm.release();
            G.release(); slotmap.release();
            slot_of.clear();
            n_slots = cap = 0;
        }
    } tabcache;
    nrhip::FilterSet env_filter;   // band pass of the envelope trigger (nrhip_station_set_envelope_trigger)
    bool env_set = false;
    nrhip::PaAdc pa_adc;           // trigger ADC + up-sampling of the phased array (nrhip_station_set_phased_array_adc)
    bool pa_adc_set = false;
    DevArray d_pa_rolls_up;        // beam rolls at the up-sampled ADC rate
    DevArray d_pa_up_taps, d_pa_hil_taps;   // FIR taps of the 'fir' up-sampling / the Hilbert transformer (nrhip_station_set_phased_array_processing)
    DevArray pa_B;                 // Bluestein tables of the chirp-z digitiser [slot of tabcache][4][FFT_MAX]
    std::vector<char> pa_built;    // per slot: tables present
    int pa_B_cap = 0;
    DevArray d_noise_amp;          // per-channel amplitude of the noise adder (nrhip_station_set_noise)
    bool noise_set = false;
    int pa_n_channels = 0, pa_n_beams = 0, pa_window = 0, pa_step = 0, pa_divisor = 0;
    // which way the convolution kernel runs this station's short events (two half-capacity blocks per CU, or everything in the
    // full-capacity one): both give the same bits, so the faster one is found by timing one call of each (calls 2 and 3 with
    // enough candidate events) and kept.  0 undecided, 1 split, 2 one block per CU
    int conv_mode = 0, conv_calls = 0;
    double conv_ms_per_event[2] = {0., 0.};
    // the same for the attenuation in two stages or one (pipeline.hip): 0 undecided, 1 two stages, 2 one
    int att_mode = 0, att_calls = 0;
    double att_ms_per_ray[2] = {0., 0.};
    // workspace of the last simulated chunk (kept for nrhip_sim_fetch and reused between calls)
    std::map<std::string, DevArray> ws;
    std::map<std::string, size_t> ws_bytes;  // valid bytes of the last chunk
    std::vector<int> h_lengths;              // distinct trace lengths of the last chunk
    int64_t last_dump_items = -1;            // candidate events whose traces the LAST simulate call kept (dump_traces), -1: it kept none
    // what the ray tables in the workspace belong to (nrhip_sim_config.reuse_ray_tables)
    // (pointers only identify a buffer as long as it has not been freed and re-allocated: the key below also holds the sizes, the
    // cuts, the reflection set-up and the station's position generation, and select_only calls are the only producers)
    int64_t rays_n_showers = -1;
    int64_t rays_n_groups = -1;
    double rays_delta_C = 0.;
    const double* rays_vertex = nullptr;
    const double* rays_max_distance = nullptr;
    int rays_n_reflections = 0;
    double rays_z_reflection = 0.;
    long rays_generation = -1;    // `generation` at the time the tables were made
    long generation = 0;          // bumped by every setter that changes the geometry (nrhip_station_set_positions)
    hipEvent_t evt[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 0..9 stage marks, 10..11 around the second stage of the attenuation
    DevArray& buf(const std::string& name) { return ws[name]; }
};
