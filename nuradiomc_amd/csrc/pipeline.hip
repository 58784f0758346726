// pipeline.hip -- host orchestration of the per-event hot path and its C-ABI entry points.
//
// All event data stays resident in HBM; one call processes the whole batch (1e6 events of a 5-channel station
// need ~4 GB of the 288 GB).  Two small device->host hand-offs per call size the later launches: the number
// of kept rays (4 B) and the per-event trace length + candidate flag (5 B per event), from which the host builds
// the list of distinct trace lengths (each needs one chirp table, built on device).
#include "../../include/nrhip.h"
#include "ctx.h"
#include "arz.h"
#include "birefringence.h"
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

using namespace nrhip;

static int ensure_twiddle(nrhip_ctx* ctx)
{
    if (ctx->twiddle) return 0;
    std::vector<double2> h(FFT_MAX / 2);
    for (int k = 0; k < FFT_MAX / 2; k++) {
        long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)FFT_MAX;
        h[k] = make_double2((double)cosl(a), (double)sinl(a));
    }
    HIPCHK(hipMalloc((void**)&ctx->twiddle, sizeof(double2) * h.size()));
    HIPCHK(hipMemcpyAsync(ctx->twiddle, h.data(), sizeof(double2) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    std::vector<double2> h2(FFT_MAX / 2 + 1);
    for (int k = 0; k <= FFT_MAX / 2; k++) {
        long double a = -3.14159265358979323846264338327950288L * (long double)k / (long double)FFT_MAX;
        h2[k] = make_double2((double)cosl(a), (double)sinl(a));
    }
    // behind w16: the per-pass twiddle tables of the convolution kernel's wave-private transforms (conv_fft.h) -- entries of the
    // master table h, so the butterflies see the same values whatever pass they are grouped into
    for (int s = 0; s < 15; s++)
        for (int l = 0; l < 64; l++) {
            const int span = s < 8 ? 512 : (s < 12 ? 256 : (s < 14 ? 128 : 64));   // twiddle W_(2 span)^pos = h[pos * FFT_MAX / (2 span)]
            const int pos = l + 64 * (s < 8 ? s : (s < 12 ? s - 8 : (s < 14 ? s - 12 : 0)));
            h2.push_back(h[pos * (FFT_MAX / (2 * span))]);
        }
    for (int s = 0; s < 7; s++)
        for (int c = 0; c < 8; c++) {
            const int span = s < 4 ? 32 : (s < 6 ? 16 : 8);
            const int pos = c + 8 * (s < 4 ? s : (s < 6 ? s - 4 : 0));
            h2.push_back(h[pos * (FFT_MAX / (2 * span))]);
        }
    HIPCHK(hipMalloc((void**)&ctx->w16, sizeof(double2) * h2.size()));
    HIPCHK(hipMemcpyAsync(ctx->w16, h2.data(), sizeof(double2) * h2.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

static void sph2cart_h(double zen, double az, double v[3])
{
    v[0] = std::sin(zen) * std::cos(az);
    v[1] = std::sin(zen) * std::sin(az);
    v[2] = std::cos(zen);
}
static void cross_h(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static bool inv3(const double A[9], double o[9])
{
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (std::fabs(det) < 1e-12) return false;
    double id = 1.0 / det;
    o[0] = (A[4] * A[8] - A[5] * A[7]) * id;
    o[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    o[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    o[3] = (A[5] * A[6] - A[3] * A[8]) * id;
    o[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    o[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    o[6] = (A[3] * A[7] - A[4] * A[6]) * id;
    o[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    o[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}
static void mat3mul(const double A[9], const double B[9], double o[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

template <class T>
static int upload(nrhip_ctx* ctx, DevArray& d, const T* h, size_t n)
{
    HIPCHK(d.reserve(n * sizeof(T)));
    HIPCHK(hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

extern "C" {

int nrhip_station_create(nrhip_ctx* ctx, const nrhip_station_desc* d, nrhip_station** out)
{
    if (!ctx || !d || !out) return nrhip_fail_msg("nrhip_station_create: NULL argument");
    if (d->n_channels <= 0) return nrhip_fail_msg("nrhip_station_create: station has no channels");
    if (d->n_samples <= 0 || d->n_samples % 2 != 0)
        return nrhip_fail_msg("nrhip_station_create: traces must have an even number of samples");
    // any even length like the reference (NuRadioReco/framework/base_trace.py:117-121); the in-LDS transforms hold 16 .. 8192 samples
    int nh = d->n_samples / 2;
    // above 8192 samples: N / 2 = 3, 5 or 7 times a power of two, at most 7168 (one odd-radix pass + radix-2 transforms in the
    // 128 KB of LDS; Bluestein would need 16 384 points).  N = 10 240 = 2 * 5 * 1024 is the case the reference's users run
    int odd = nh, two = 0;
    while (odd % 2 == 0) { odd /= 2; two++; }
    const bool big_mixed = nh > FFT_MAX / 2 && nh <= 7 * FFT_MAX / 8 && (odd == 3 || odd == 5 || odd == 7);
    if (nh < 8 || (nh > FFT_MAX / 2 && !big_mixed))
        return nrhip_fail_msg("nrhip_station_create: n_samples must be an even number between 16 and 8192 (or 10240, 12288, 14336)");
    // (N / 2 no power of two: Bluestein on the next power of two >= N - 1, at most FFT_MAX points = the whole LDS of the ray kernels,
    // whose amplitude tables then sit in HBM scratch)
    if (d->n_att_freq <= 0 || d->n_att_freq > NRHIP_MAX_NFC) return nrhip_fail_msg("nrhip_station_create: bad n_att_freq");
    HIPCHK(hipSetDevice(ctx->device));
    if (ensure_twiddle(ctx)) return -1;
    nrhip_station* s = new nrhip_station();
    s->ctx = ctx;
    int n = d->n_channels;
    std::vector<double> rot(9 * n), roti(9 * n);
    int tab_mask = 0;
    // model frame of the analytic antennas: boresight (0, 0), tines normal (90 deg, 0) (antennapattern.py:1612-1636)
    double e1[3], e2[3], e3[3];
    sph2cart_h(0., 0., e1);
    sph2cart_h(90 * 0.017453292519943295, 0., e2);
    cross_h(e1, e2, e3);
    double E[9] = {e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], e3[0], e3[1], e3[2]}, Ei[9];
    if (!inv3(E, Ei)) { delete s; return nrhip_fail_msg("nrhip_station_create: singular antenna model frame"); }
    for (int c = 0; c < n; c++) {
        const int am = d->antenna_model[c];
        if (am != NRHIP_ANT_VPOL && am != NRHIP_ANT_HPOL && am != NRHIP_ANT_LPDA && am != NRHIP_ANT_TABLE) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: antenna model not implemented (analytic_VPol, analytic_HPol, analytic_LPDA, tabulated)");
        }
        if (am == NRHIP_ANT_TABLE) {
            if (!d->antenna_table_index || !d->antenna_tables || d->antenna_table_index[c] < 0 ||
                d->antenna_table_index[c] >= d->n_antenna_tables) {
                delete s;
                return nrhip_fail_msg("nrhip_station_create: tabulated antenna without a valid table index");
            }
            // the table's own simulation frame (antennapattern.py:1190-1201)
            const double* mo = d->antenna_tables[d->antenna_table_index[c]].orientation;
            double m1[3], m2[3], m3[3];
            sph2cart_h(mo[0], mo[1], m1);
            sph2cart_h(mo[2], mo[3], m2);
            cross_h(m1, m2, m3);
            if (std::sqrt(m3[0] * m3[0] + m3[1] * m3[1] + m3[2] * m3[2]) < 0.9) {
                delete s;
                return nrhip_fail_msg("orientation of antenna not properly defined in WIPL-D orientation file");
            }
            double Em[9] = {m1[0], m1[1], m1[2], m2[0], m2[1], m2[2], m3[0], m3[1], m3[2]};
            if (!inv3(Em, Ei)) { delete s; return nrhip_fail_msg("nrhip_station_create: singular antenna model frame"); }
        } else {
            inv3(E, Ei);
            tab_mask |= am == NRHIP_ANT_LPDA ? 0x1c : (1 << am);
        }
        const double* o = d->orientation + 4 * c;
        double a1[3], a2[3], a3[3];
        sph2cart_h(o[0], o[1], a1);
        sph2cart_h(o[2], o[3], a2);
        cross_h(a1, a2, a3);
        double nrm = std::sqrt(a3[0] * a3[0] + a3[1] * a3[1] + a3[2] * a3[2]);
        if (nrm < 0.9) {  // antennapattern.py:1209-1211
            delete s;
            return nrhip_fail_msg("orientation of antenna not properly defined detector description");
        }
        double A[9] = {a1[0], a1[1], a1[2], a2[0], a2[1], a2[2], a3[0], a3[1], a3[2]};
        mat3mul(Ei, A, &rot[9 * c]);
        if (!inv3(&rot[9 * c], &roti[9 * c])) { delete s; return nrhip_fail_msg("nrhip_station_create: singular antenna rotation"); }
    }
    std::vector<double> lnf(nh + 1, 0.);
    {
        const double df = 1.0 / (d->n_samples * (1. / d->sampling_rate));
        for (int k = 1; k <= nh; k++) lnf[k] = std::log(k * df);
    }
    // tables on the N-sample frequency grid: f^p of the Alvarez2009 form factors, coarse-grid segment of np.interp
    std::vector<double> fpow(3 * (size_t)(nh + 1), 0.);
    std::vector<float> fpow_f(3 * (size_t)(nh + 1), 0.f);
    std::vector<unsigned char> seg(nh + 1, 0);
    {
        const double df = 1.0 / (d->n_samples * (1. / d->sampling_rate));
        const double pw[3] = {2.57, 2.74, 1.27};
        const int nfc = d->n_att_freq;
        for (int k = 1; k <= nh; k++) {
            const double f = k * df;
            for (int r = 0; r < 3; r++) {
                fpow[r * (size_t)(nh + 1) + k] = std::pow(f, pw[r]);
                fpow_f[r * (size_t)(nh + 1) + k] = (float)fpow[r * (size_t)(nh + 1) + k];
            }
            int lo = 0;
            while (lo < nfc - 2 && f >= d->att_freq[lo + 1]) lo++;
            seg[k] = (unsigned char)lo;
        }
    }
    std::vector<double> invl(d->n_att_freq, 0.);
    if (d->att_bound_inv_length)
        for (int k = 0; k < d->n_att_freq; k++) invl[k] = d->att_bound_inv_length[k] > 0 ? d->att_bound_inv_length[k] : 0.;
    s->h_pos.assign(d->position, d->position + 3 * n);
    s->h_cable.assign(d->cable_delay, d->cable_delay + n);
    if (upload(ctx, s->d_pos, d->position, 3 * n) || upload(ctx, s->d_cable, d->cable_delay, n) ||
        upload(ctx, s->d_model, d->antenna_model, n) || upload(ctx, s->d_rot, rot.data(), 9 * n) ||
        upload(ctx, s->d_rot_inv, roti.data(), 9 * n) || upload(ctx, s->d_fc, d->att_freq, d->n_att_freq) ||
        upload(ctx, s->d_lnf, lnf.data(), lnf.size()) || upload(ctx, s->d_invl, invl.data(), invl.size()) ||
        upload(ctx, s->d_fpow, fpow.data(), fpow.size()) || upload(ctx, s->d_fpow_f, fpow_f.data(), fpow_f.size()) ||
        upload(ctx, s->d_seg, seg.data(), seg.size())) {
        delete s;
        return -1;
    }
    int n_bins = 0;
    if (d->att_bound_n_bins > 0 && d->att_bound_bin_inv_length && d->att_bound_bin_width > 0) {
        if (d->att_bound_n_bins > 63) { delete s; return nrhip_fail_msg("nrhip_station_create: att_bound_n_bins > 63"); }
        n_bins = d->att_bound_n_bins;
        if (upload(ctx, s->d_attbin, d->att_bound_bin_inv_length, (size_t)n_bins * d->n_att_freq)) { delete s; return -1; }
    }
    // tabulated antenna patterns -> HBM
    std::vector<AntTabDev> tabs(std::max(d->n_antenna_tables, 0));
    int max_tab_freq = 0;
    for (int t = 0; t < (int)tabs.size(); t++) {
        const nrhip_antenna_table& a = d->antenna_tables[t];
        if (a.n_freq < 2 || a.n_theta < 1 || a.n_phi < 1 || !a.freqs || !a.thetas || !a.phis || !a.vel_theta || !a.vel_phi) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: malformed antenna table");
        }
        const size_t nv = (size_t)a.n_freq * a.n_theta * a.n_phi;
        s->d_tabdata.resize(s->d_tabdata.size() + 5);
        DevArray* da = &s->d_tabdata[s->d_tabdata.size() - 5];
        if (upload(ctx, da[0], a.freqs, a.n_freq) || upload(ctx, da[1], a.thetas, a.n_theta) || upload(ctx, da[2], a.phis, a.n_phi) ||
            upload(ctx, da[3], a.vel_theta, 2 * nv) || upload(ctx, da[4], a.vel_phi, 2 * nv)) {
            delete s;
            return -1;
        }
        tabs[t] = AntTabDev{a.n_freq, a.n_theta, a.n_phi, da[0].as<double>(), da[1].as<double>(), da[2].as<double>(),
                            da[3].as<double2>(), da[4].as<double2>()};
        max_tab_freq = std::max(max_tab_freq, a.n_freq);
    }
    std::vector<int> tab_index(n, 0);
    if (d->antenna_table_index)
        for (int c = 0; c < n; c++) tab_index[c] = d->antenna_model[c] == NRHIP_ANT_TABLE ? d->antenna_table_index[c] : 0;
    if (upload(ctx, s->d_anttab_index, tab_index.data(), n) ||
        (!tabs.empty() && upload(ctx, s->d_anttabs, tabs.data(), tabs.size()))) {
        delete s;
        return -1;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    StationDev& v = s->dev;
    bool any_tab = false;
    for (int c = 0; c < n; c++) any_tab |= d->antenna_model[c] == NRHIP_ANT_TABLE;
    v.ant_tabs = any_tab ? s->d_anttabs.as<AntTabDev>() : nullptr;
    v.ant_tab_index = s->d_anttab_index.as<int>();
    v.max_tab_freq = max_tab_freq;
    v.n_att_bins = n_bins;
    v.att_bin_width = n_bins ? d->att_bound_bin_width : 0.;
    v.att_bin_inv = n_bins ? s->d_attbin.as<double>() : nullptr;
    v.n_ch = n;
    v.tab_mask = tab_mask;
    v.N = d->n_samples;
    v.n_fc = d->n_att_freq;
    v.fs = d->sampling_rate;
    v.pre_pulse = d->pre_pulse_time;
    v.post_pulse = d->post_pulse_time;
    v.readout_length = d->readout_length;
    v.att_bound_depth = (d->att_bound_inv_length && d->att_bound_depth > 0) ? d->att_bound_depth : -1.;
    v.pos = s->d_pos.as<double>();
    v.cable = s->d_cable.as<double>();
    v.trig_on = nullptr;
    v.ant_model = s->d_model.as<int>();
    v.rot = s->d_rot.as<double>();
    v.rot_inv = s->d_rot_inv.as<double>();
    v.fcoarse = s->d_fc.as<double>();
    v.lnf = s->d_lnf.as<double>();
    v.inv_lmax = s->d_invl.as<double>();
    v.fpow = s->d_fpow.as<double>();
    v.fpow_f = s->d_fpow_f.as<float>();
    v.seg = s->d_seg.as<unsigned char>();
    // the N / 2-point transforms of the ray stages: radix 2, or Bluestein tables for any other length
    v.np.nh = nh;
    v.np.log2nh = -1;
    v.np.log2p = 0;
    v.np.wN = v.np.cw = v.np.Bf = v.np.Bi = nullptr;
    for (int l = 0; l <= FFT_LOG2_MAX; l++)
        if ((1 << l) == nh) v.np.log2nh = l;
    if (v.np.log2nh < 0 && big_mixed) {
        v.np.radix = odd;
        v.np.log2m = two;
        if (s->d_nplan.reserve(sizeof(double2) * ((size_t)nh + 1)) != hipSuccess) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: out of device memory (transform tables)");
        }
        double2* base = s->d_nplan.as<double2>();
        launch_nplan_tables(ctx->stream, nh, 0, base, nullptr, nullptr, nullptr, ctx->twiddle);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: building the transform tables failed");
        }
        v.np.wN = base;
    } else if (v.np.log2nh < 0) {
        while ((1 << v.np.log2p) < 2 * nh - 1) v.np.log2p++;
        const size_t P = (size_t)1 << v.np.log2p;
        if (s->d_nplan.reserve(sizeof(double2) * (2 * P + 2 * (size_t)nh + 2)) != hipSuccess) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: out of device memory (transform tables)");
        }
        double2* base = s->d_nplan.as<double2>();
        launch_nplan_tables(ctx->stream, nh, v.np.log2p, base, base + nh + 1, base + 2 * nh + 2, base + 2 * nh + 2 + P, ctx->twiddle);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: building the transform tables failed");
        }
        v.np.wN = base;
        v.np.cw = base + nh + 1;
        v.np.Bf = base + 2 * nh + 2;
        v.np.Bi = base + 2 * nh + 2 + P;
    }
    // filter chains: one for all channels (n_filter_sets <= 1), or up to NRHIP_MAX_FSETS of them with a per-channel index;
    // the stages of all chains follow each other in the filter_* arrays
    const int n_sets = d->n_filter_sets > 1 ? d->n_filter_sets : 1;
    if (n_sets > NRHIP_MAX_FSETS) { delete s; return nrhip_fail_msg("nrhip_station_create: too many filter chains"); }
    if (d->n_filter_table_points > 0) {
        if (!d->filter_table || upload(ctx, s->d_filter_pool, d->filter_table, (size_t)3 * d->n_filter_table_points)) {
            delete s;
            return nrhip_fail_msg("nrhip_station_create: tabulated filter responses could not be copied");
        }
    }
    int stage0 = 0;
    for (int fs = 0; fs < n_sets; fs++) {
        FilterSet& f = s->filters[fs];
        memset(&f, 0, sizeof f);
        f.pool = s->d_filter_pool.as<double>();
        f.n = (d->n_filter_sets > 1) ? d->set_n_filters[fs] : d->n_filters;
        if (f.n < 0 || f.n > NRHIP_MAX_FILTERS) { delete s; return nrhip_fail_msg("nrhip_station_create: too many filters"); }
        for (int i = 0; i < f.n; i++) {
            const int q = stage0 + i;
            f.kind[i] = d->filter_kind ? d->filter_kind[q] : 0;
            if (f.kind[i] < 0 || f.kind[i] > 4) { delete s; return nrhip_fail_msg("nrhip_station_create: unknown filter kind"); }
            f.nb[i] = d->filter_nb[q];
            f.na[i] = d->filter_na[q];
            if (f.kind[i] == NRHIP_FILTER_TABULATED) {
                if (f.nb[i] < 2 || f.na[i] < 0 || f.na[i] + f.nb[i] > d->n_filter_table_points) {
                    delete s;
                    return nrhip_fail_msg("nrhip_station_create: tabulated filter stage outside the response table");
                }
                memcpy(f.b[i], d->filter_b + (size_t)q * NRHIP_MAX_POLY, sizeof(double) * 2);
                continue;
            }
            if (f.nb[i] < 1 || f.nb[i] > NRHIP_MAX_POLY || f.na[i] < 1 || f.na[i] > NRHIP_MAX_POLY) {
                delete s;
                return nrhip_fail_msg("nrhip_station_create: filter polynomial too long");
            }
            memcpy(f.b[i], d->filter_b + (size_t)q * NRHIP_MAX_POLY, sizeof(double) * f.nb[i]);
            memcpy(f.a[i], d->filter_a + (size_t)q * NRHIP_MAX_POLY, sizeof(double) * f.na[i]);
        }
        stage0 += f.n;
    }
    std::vector<int> ch_fset(n, 0);
    for (int fs = 0; fs < NRHIP_MAX_FSETS; fs++) v.fset_tab_mask[fs] = 0;
    for (int c = 0; c < n; c++) {
        if (d->n_filter_sets > 1) {
            if (!d->channel_filter_set || d->channel_filter_set[c] < 0 || d->channel_filter_set[c] >= n_sets) {
                delete s;
                return nrhip_fail_msg("nrhip_station_create: channel_filter_set outside the filter chains");
            }
            ch_fset[c] = d->channel_filter_set[c];
        }
        const int am = d->antenna_model[c];
        if (am != NRHIP_ANT_TABLE) v.fset_tab_mask[ch_fset[c]] |= am == NRHIP_ANT_LPDA ? 0x1c : (1 << am);
    }
    if (upload(ctx, s->d_ch_fset, ch_fset.data(), (size_t)n) || upload(ctx, s->d_filtersets, s->filters, (size_t)n_sets)) {
        delete s;
        return -1;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    v.n_fsets = n_sets;
    v.ch_fset = n_sets > 1 ? s->d_ch_fset.as<int>() : nullptr;
    ctx->stations.insert(s);
    *out = s;
    return 0;
}

void nrhip_station_destroy(nrhip_station* s)
{
    if (!s) return;
    nrhip_station_detach(s);
    delete s;
}

void nrhip_station_detach(nrhip_station* s)
{
    if (!s || !s->ctx) return;  // already detached by nrhip_ctx_destroy
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (auto& kv : s->ws) kv.second.release();
    for (auto& e : s->evt) if (e) (void)hipEventDestroy(e);
    s->d_pos.release(); s->d_cable.release(); s->d_model.release();
    s->d_rot.release(); s->d_rot_inv.release(); s->d_fc.release(); s->d_lnf.release(); s->d_invl.release();
    s->d_nplan.release();
    s->d_fpow.release(); s->d_fpow_f.release(); s->d_seg.release(); s->d_attbin.release(); s->d_anttabs.release(); s->d_anttab_index.release();
    for (auto& a : s->d_tabdata) a.release();
    s->d_arz_depth.release(); s->d_arz_ce.release(); s->d_arz_par.release(); s->d_bire_knots.release();
    s->d_bire_coeffs.release(); s->d_shower_profile.release(); s->d_shower_rescale.release();
    s->d_pa_channel.release(); s->d_pa_rolls.release(); s->d_pa_mask.release(); s->d_trig_on.release();
    s->d_filter_pool.release(); s->d_ch_fset.release(); s->d_filtersets.release();
    s->tabcache.release();
    s->d_noise_amp.release();
    s->d_pa_rolls_up.release();
    s->d_pa_up_taps.release(); s->d_pa_hil_taps.release();
    s->pa_B.release();
    s->ws.clear();
    s->ws_bytes.clear(); s->last_dump_items = -1;
    s->ctx->stations.erase(s);
    s->ctx = nullptr;
}

int nrhip_station_set_positions(nrhip_station* s, const double* position)
{
    if (!s || !s->ctx || !position) return nrhip_fail_msg("nrhip_station_set_positions: NULL argument or station without a context");
    const int n = s->dev.n_ch;
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));  // no call of this station may still read the old positions
    s->h_pos.assign(position, position + 3 * n);
    HIPCHK(hipMemcpyAsync(s->d_pos.p, s->h_pos.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->ws_bytes.clear(); s->last_dump_items = -1;  // the tables of the last call belong to the old positions
    s->rays_n_showers = -1;
    s->generation++;
    return 0;
}

int64_t nrhip_station_release_workspace(nrhip_station* s)
{
    if (!s || !s->ctx) return 0;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    int64_t freed = 0;
    for (auto& kv : s->ws) {
        freed += (int64_t)kv.second.cap;
        kv.second.release();
    }
    s->ws_bytes.clear(); s->last_dump_items = -1;
    // the per-length table cache only grows while the station lives (one row of ~1 MB or more per distinct common-trace length):
    // it is given back here too; the next call rebuilds the rows of the lengths it meets (about 1 us per length)
    auto& tc = s->tabcache;
    for (DevArray* a : {&tc.B_fwd, &tc.B_inv, &tc.vel, &tc.E, &tc.H, &tc.Cf, &tc.Ci, &tc.hnorm, &tc.G, &tc.slotmap, &s->pa_B})
        freed += (int64_t)a->cap;
    tc.release();
    s->pa_B.release();
    s->pa_built.clear();
    s->pa_B_cap = 0;
    s->rays_n_showers = -1;
    return freed;
}

int nrhip_station_set_trigger_channels(nrhip_station* s, int32_t n, const int32_t* channels)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_trigger_channels: NULL argument or station without a context");
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    if (n <= 0 || !channels) {
        s->dev.trig_on = nullptr;
        return 0;
    }
    std::vector<unsigned char> on(s->dev.n_ch, 0);
    for (int i = 0; i < n; i++) {
        if (channels[i] < 0 || channels[i] >= s->dev.n_ch) return nrhip_fail_msg("nrhip_station_set_trigger_channels: bad channel index");
        on[channels[i]] = 1;
    }
    if (upload(s->ctx, s->d_trig_on, on.data(), on.size())) return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->dev.trig_on = s->d_trig_on.as<unsigned char>();
    return 0;
}

int nrhip_station_set_envelope_trigger(nrhip_station* s, int32_t nb, int32_t na, const double* b, const double* a)
{
    if (!s) return nrhip_fail_msg("nrhip_station_set_envelope_trigger: NULL argument");
    if (nb <= 0) {
        s->env_set = false;
        return 0;
    }
    if (!b || !a || na <= 0 || nb > NRHIP_MAX_POLY || na > NRHIP_MAX_POLY)
        return nrhip_fail_msg("nrhip_station_set_envelope_trigger: bad polynomial sizes");
    FilterSet f;
    memset(&f, 0, sizeof f);
    f.n = 1;
    f.kind[0] = 0;
    f.nb[0] = nb;
    f.na[0] = na;
    for (int i = 0; i < nb; i++) f.b[0][i] = b[i];
    for (int i = 0; i < na; i++) f.a[0][i] = a[i];
    s->env_filter = f;
    s->env_set = true;
    return 0;
}

int nrhip_station_set_phased_array_adc(nrhip_station* s, double adc_fs, int32_t n_bits, double v_min, double v_max, int32_t output_counts,
                                       int32_t upsampling_factor, int32_t saturation_bits, int32_t resample_p, int32_t resample_q,
                                       const int32_t* rolls_up)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_phased_array_adc: NULL argument or station without a context");
    s->pa_adc_set = false;
    if (!(adc_fs > 0)) return 0;
    if (s->pa_n_channels <= 0) return nrhip_fail_msg("nrhip_station_set_phased_array_adc: nrhip_station_set_phased_array comes first");
    // n_bits = 0: no digitisation (phased_trigger(apply_digitization=False) with up-sampling or the envelope mode): the channel traces
    // at the simulation's rate go through the same up-sampling / beam / envelope kernels as volts
    const bool analog = n_bits == 0;
    if (analog && !(fabs(adc_fs - s->dev.fs) <= 1e-8 + 1e-5 * fabs(s->dev.fs)))
        return nrhip_fail_msg("nrhip_station_set_phased_array_adc: without digitisation (n_bits = 0) the rate is the simulation's");
    if (analog) { v_min = 0.; v_max = 1.; output_counts = 0; }
    if (!rolls_up || n_bits < 0 || n_bits > 24 || !(v_max > v_min) || upsampling_factor < 1 || resample_p < 1 || resample_q < 1)
        return nrhip_fail_msg("nrhip_station_set_phased_array_adc: bad ADC description");
    if (!analog && adc_fs > 0.49 * s->dev.fs) return nrhip_fail_msg("nrhip_station_set_phased_array_adc: the ADC must sample at less than 0.49 of the simulation's rate");
    HIPCHK(hipSetDevice(s->ctx->device));
    if (upload(s->ctx, s->d_pa_rolls_up, rolls_up, (size_t)s->pa_n_beams * s->pa_n_channels)) return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->pa_adc = PaAdc{adc_fs, v_min, v_max, n_bits, output_counts ? 1 : 0, upsampling_factor, saturation_bits, resample_p, resample_q, 0};
    s->pa_adc.up_method = s->pa_adc.mode = s->pa_adc.n_up_taps = s->pa_adc.n_hil_taps = 0;
    s->pa_adc.up_taps = s->pa_adc.hil_taps = nullptr;
    s->pa_adc_set = true;
    s->pa_built.clear();   // the digitiser's transform tables depend on these rates
    return 0;
}

int nrhip_station_set_phased_array_clock_offset(nrhip_station* s, int32_t clock_offset)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_phased_array_clock_offset: NULL argument or station without a context");
    if (!s->pa_adc_set) return nrhip_fail_msg("nrhip_station_set_phased_array_clock_offset: nrhip_station_set_phased_array_adc comes first");
    if (clock_offset < 0) return nrhip_fail_msg("nrhip_station_set_phased_array_clock_offset: the clock offset must not be negative");
    // the delayed trace loses round(delay * fs) samples: at most half of the shortest trace there can be
    if ((clock_offset / s->pa_adc.adc_fs) * s->dev.fs + 2 > s->dev.N / 2)
        return nrhip_fail_msg("nrhip_station_set_phased_array_clock_offset: the delay is longer than half a trace");
    if (clock_offset && s->pa_adc.n_bits == 0)
        return nrhip_fail_msg("nrhip_station_set_phased_array_clock_offset: a clock offset needs the trigger ADC (n_bits > 0)");
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->pa_adc.clock_offset = clock_offset;
    s->pa_built.clear();   // the transform tables depend on the trace length behind the delay
    return 0;
}

int nrhip_station_set_phased_array_processing(nrhip_station* s, int32_t upsampling_method, int32_t n_up_taps, const double* up_taps,
                                              int32_t mode, int32_t n_hilbert_taps, const double* hilbert_taps)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_phased_array_processing: NULL argument or station without a context");
    if (!s->pa_adc_set) return nrhip_fail_msg("nrhip_station_set_phased_array_processing: nrhip_station_set_phased_array_adc comes first");
    if (upsampling_method < 0 || upsampling_method > 2)   // NotImplementedError of digital_upsampling :178-180
        return nrhip_fail_msg("nrhip_station_set_phased_array_processing: Interpolation method must be lin, fft, or fir");
    if (mode < 0 || mode > 2) return nrhip_fail_msg("nrhip_station_set_phased_array_processing: mode must be either 'power_sum' or 'hilbert_env'");
    if (upsampling_method == 2 && (!up_taps || n_up_taps < 1 || n_up_taps > 1024))
        return nrhip_fail_msg("nrhip_station_set_phased_array_processing: the 'fir' up-sampling needs 1..1024 filter taps");
    if (mode == 1 && (!hilbert_taps || n_hilbert_taps < 1 || n_hilbert_taps > 1024 || n_hilbert_taps % 2 == 0))
        return nrhip_fail_msg("nrhip_station_set_phased_array_processing: Num taps MUST be odd for a hilbert transformer");
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->pa_adc.up_method = upsampling_method;
    s->pa_adc.mode = mode;
    s->pa_built.clear();   // the transform tables of the up-sampling stage depend on the method
    s->pa_adc.n_up_taps = s->pa_adc.n_hil_taps = 0;
    s->pa_adc.up_taps = s->pa_adc.hil_taps = nullptr;
    if (upsampling_method == 2) {
        if (upload(s->ctx, s->d_pa_up_taps, up_taps, (size_t)n_up_taps)) return -1;
        s->pa_adc.n_up_taps = n_up_taps;
        s->pa_adc.up_taps = s->d_pa_up_taps.as<double>();
    }
    if (mode == 1) {
        if (upload(s->ctx, s->d_pa_hil_taps, hilbert_taps, (size_t)n_hilbert_taps)) return -1;
        s->pa_adc.n_hil_taps = n_hilbert_taps;
        s->pa_adc.hil_taps = s->d_pa_hil_taps.as<double>();
    }
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return 0;
}

int nrhip_station_set_noise(nrhip_station* s, int32_t n, const double* amplitude)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_noise: NULL argument or station without a context");
    if (n <= 0 || !amplitude) {
        s->noise_set = false;
        return 0;
    }
    if (n != s->dev.n_ch) return nrhip_fail_msg("nrhip_station_set_noise: one amplitude per channel is needed");
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    if (upload(s->ctx, s->d_noise_amp, amplitude, (size_t)n)) return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->noise_set = true;
    return 0;
}

int nrhip_station_set_phased_array(nrhip_station* s, int32_t n_pa, const int32_t* channels, int32_t n_beams, const int32_t* rolls,
                                   int32_t window, int32_t step, int32_t averaging_divisor)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_phased_array: NULL argument or station without a context");
    s->pa_n_channels = 0;
    s->pa_adc_set = false;
    if (n_pa <= 0) return 0;
    if (!channels || !rolls || n_beams < 1 || window < 1 || step < 1)
        return nrhip_fail_msg("nrhip_station_set_phased_array: channels, rolls, n_beams >= 1, window >= 1 and step >= 1 are required");
    for (int c = 0; c < n_pa; c++)
        if (channels[c] < 0 || channels[c] >= s->dev.n_ch) return nrhip_fail_msg("nrhip_station_set_phased_array: bad channel index");
    HIPCHK(hipSetDevice(s->ctx->device));
    std::vector<unsigned char> mask(s->dev.n_ch, 0);
    for (int c = 0; c < n_pa; c++) mask[channels[c]] = 1;
    if (upload(s->ctx, s->d_pa_channel, channels, (size_t)n_pa) || upload(s->ctx, s->d_pa_rolls, rolls, (size_t)n_beams * n_pa) ||
        upload(s->ctx, s->d_pa_mask, mask.data(), mask.size()))
        return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->pa_n_channels = n_pa;
    s->pa_n_beams = n_beams;
    s->pa_window = window;
    s->pa_step = step;
    s->pa_divisor = averaging_divisor > 0 ? averaging_divisor : window;
    return 0;
}

int nrhip_station_set_arz(nrhip_station* s, int32_t n_profiles, int32_t n_depth, const double* profile_depth,
                          const double* profile_ce, const double* parameters, double interp_factor2, int32_t em_formula)
{
    if (!s || !s->ctx || !profile_depth || !profile_ce || !parameters) return nrhip_fail_msg("nrhip_station_set_arz: NULL argument or station without a context");
    if (n_profiles < 1 || n_depth < 2 || n_depth > 2048) return nrhip_fail_msg("nrhip_station_set_arz: profiles need 2..2048 depth bins");
    HIPCHK(hipSetDevice(s->ctx->device));
    if (upload(s->ctx, s->d_arz_depth, profile_depth, (size_t)n_depth) ||
        upload(s->ctx, s->d_arz_ce, profile_ce, (size_t)n_profiles * n_depth) || upload(s->ctx, s->d_arz_par, parameters, (size_t)14))
        return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->arz_n_profiles = n_profiles;
    s->arz_n_depth = n_depth;
    s->arz_interp_factor2 = interp_factor2;
    s->arz_em_formula = em_formula;
    return 0;
}

int nrhip_station_set_shower_profiles(nrhip_station* s, int64_t n_showers, const int32_t* profile_index, const double* rescale)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_shower_profiles: NULL argument or station without a context");
    s->n_shower_profiles = 0;
    if (n_showers <= 0 || !profile_index || !rescale) return 0;
    for (int64_t i = 0; i < n_showers; i++)
        if (profile_index[i] < 0 || profile_index[i] >= s->arz_n_profiles)
            return nrhip_fail_msg("nrhip_station_set_shower_profiles: profile index outside the library set with nrhip_station_set_arz");
    HIPCHK(hipSetDevice(s->ctx->device));
    if (upload(s->ctx, s->d_shower_profile, profile_index, (size_t)n_showers) ||
        upload(s->ctx, s->d_shower_rescale, rescale, (size_t)n_showers))
        return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    s->n_shower_profiles = n_showers;
    return 0;
}

int nrhip_station_set_birefringence(nrhip_station* s, const int32_t* n_knots, const double* knots, const double* coeffs,
                                    double n_ref, double angle_to_iceflow)
{
    if (!s || !s->ctx) return nrhip_fail_msg("nrhip_station_set_birefringence: NULL argument or station without a context");
    s->bire_n_knots[0] = s->bire_n_knots[1] = s->bire_n_knots[2] = 0;
    if (!n_knots) return 0;
    if (!knots || !coeffs) return nrhip_fail_msg("nrhip_station_set_birefringence: NULL argument");
    size_t nk = 0;
    for (int j = 0; j < 3; j++) {
        if (n_knots[j] < 8) return nrhip_fail_msg("nrhip_station_set_birefringence: a cubic spline needs at least 8 knots");
        nk += (size_t)n_knots[j];
    }
    HIPCHK(hipSetDevice(s->ctx->device));
    if (upload(s->ctx, s->d_bire_knots, knots, nk) || upload(s->ctx, s->d_bire_coeffs, coeffs, nk)) return -1;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    for (int j = 0; j < 3; j++) s->bire_n_knots[j] = n_knots[j];
    s->bire_n_ref = n_ref;
    s->bire_angle = angle_to_iceflow;
    return 0;
}

// first channel of the station on which a shower has a ray that passes the delta_C cut (-1: none): what the host needs to
// walk the showers in the order in which the reference meets them (simulation.py:1454-1600 with :143-242)
__global__ void shower_first_channel_kernel(long n_showers, int n_ch, const int* __restrict__ keep, int* __restrict__ first,
                                            int stride)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_showers) return;
    int f = -1;
    for (int c = 0; c < n_ch && f < 0; c++)
        for (int s = 0; s < stride; s++)
            if (keep[(i * n_ch + c) * stride + s]) f = c;
    first[i] = f;
}

// ---- split_event_time_diff (simulation.group_into_events :906-947) -------------------------------------------------------------
// The signals of one event group at one station are sorted by their start time (electric-field start + cable delay); where two
// consecutive start times are more than split_event_time_diff apart a new (sub-)event begins, with its own readout window and
// trigger decision.  One thread per group: number of sub-events, and the group's rays reordered sub-event by sub-event (original
// order inside each) so that every sub-event is a contiguous ray range again.  Groups that do not split keep their order.
__global__ void sub_event_split_kernel(int n_groups, const int* __restrict__ grp_ray, const double* __restrict__ t0,
                                       const int* __restrict__ ch, const double* __restrict__ cable, double split,
                                       const int* __restrict__ slot_in, int* __restrict__ slot_out, int* __restrict__ sub_sorted,
                                       int* __restrict__ order, int* __restrict__ sub_of, int* __restrict__ n_sub,
                                       int* __restrict__ any_split)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = grp_ray[g], r1 = grp_ray[g + 1];
    double tmin = INFINITY, tmax = -INFINITY;
    for (int r = r0; r < r1; r++) {
        const double t = t0[r] + cable[ch[r]];
        tmin = fmin(tmin, t);
        tmax = fmax(tmax, t);
    }
    if (!(tmax - tmin > split)) {   // no gap can exceed the limit
        for (int r = r0; r < r1; r++) { slot_out[r] = slot_in[r]; sub_sorted[r] = 0; }
        n_sub[g] = 1;
        return;
    }
    // insertion sort of the ray indices by start time (stable), sub-event index = number of gaps passed
    for (int r = r0; r < r1; r++) {
        const double t = t0[r] + cable[ch[r]];
        int k = r;
        while (k > r0 && t0[order[k - 1]] + cable[ch[order[k - 1]]] > t) { order[k] = order[k - 1]; k--; }
        order[k] = r;
    }
    int ns = 0;
    sub_of[order[r0]] = 0;
    for (int k = r0 + 1; k < r1; k++) {
        const double d = (t0[order[k]] + cable[ch[order[k]]]) - (t0[order[k - 1]] + cable[ch[order[k - 1]]]);
        if (d > split) ns++;
        sub_of[order[k]] = ns;
    }
    n_sub[g] = ns + 1;
    int pos = r0;
    for (int sidx = 0; sidx <= ns; sidx++)
        for (int r = r0; r < r1; r++)
            if (sub_of[r] == sidx) { slot_out[pos] = slot_in[r]; sub_sorted[pos] = sidx; pos++; }
    if (ns > 0) atomicOr(any_split, 1);
}

// ray range and group of every sub-event (ev_base = exclusive scan of n_sub)
__global__ void sub_event_ranges_kernel(int n_groups, const int* __restrict__ grp_ray, const int* __restrict__ sub_sorted,
                                        const int* __restrict__ ev_base, int* __restrict__ sub_ray, int* __restrict__ ev_group,
                                        int* __restrict__ ev_sub)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int r0 = grp_ray[g], r1 = grp_ray[g + 1];
    int e = ev_base[g];
    sub_ray[e] = r0;
    ev_group[e] = g;
    ev_sub[e] = 0;
    for (int r = r0 + 1; r < r1; r++)
        if (sub_sorted[r] != sub_sorted[r - 1]) {
            e++;
            sub_ray[e] = r;
            ev_group[e] = g;
            ev_sub[e] = sub_sorted[r];
        }
    if (g == n_groups - 1) sub_ray[ev_base[n_groups]] = r1;
}

// the candidate flag belongs to the event group (simulation.py:1556-1560 tests the station's whole sim station before
// group_into_events): every sub-event of a candidate group is evaluated
__global__ void sub_event_candidate_kernel(int n_groups, const int* __restrict__ ev_base, unsigned char* __restrict__ candidate,
                                           const int* __restrict__ n_rays)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int e0 = ev_base[g], e1 = ev_base[g + 1];
    if (e1 - e0 < 2) return;
    int any = 0;
    for (int e = e0; e < e1; e++) any |= candidate[e];
    for (int e = e0; e < e1; e++) candidate[e] = (unsigned char)(any && n_rays[e] > 0);
}

__global__ void sub_event_trigger_kernel(int n_ev, const int* __restrict__ ev_group, const unsigned char* __restrict__ sub_triggered,
                                         unsigned char* __restrict__ triggered)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_ev && sub_triggered[e]) triggered[ev_group[e]] = 1;
}

static thread_local char g_ws_fail[160] = "";
#define WS(name, type, count)                                                                       \
    ([&]() -> type* {                                                                               \
        DevArray& b_ = st->buf(name);                                                               \
        if (b_.reserve((size_t)(count) * sizeof(type) + 16) != hipSuccess) {                        \
            snprintf(g_ws_fail, sizeof g_ws_fail, "nrhip_simulate_events: out of device memory (%s: %.3g GB)", name,  \
                     (double)((size_t)(count) * sizeof(type)) * 1e-9);                              \
            return (type*)nullptr;                                                                  \
        }                                                                                           \
        st->ws_bytes[name] = (size_t)(count) * sizeof(type);                                        \
        return b_.as<type>();                                                                       \
    })()
#define NEED(ptr) if (!(ptr)) return nrhip_fail_msg(g_ws_fail[0] ? g_ws_fail : "nrhip_simulate_events: out of device memory")
#define LCHK(what)                                                          \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) return nrhip_fail("launch " what, e_);        \
    } while (0)

int nrhip_simulate_events(nrhip_ctx* ctx, nrhip_station* st, const nrhip_sim_config* cfg, int64_t n_events,
                          const double* vertex, const double* zenith, const double* azimuth, const double* energy,
                          const int32_t* shower_type, const double* k_L, uint8_t* triggered, nrhip_sim_stats* stats)
{
    return nrhip_simulate_event_groups(ctx, st, cfg, n_events, vertex, zenith, azimuth, energy, shower_type, k_L, nullptr,
                                       nullptr, n_events, nullptr, triggered, stats);
}

int nrhip_simulate_event_groups(nrhip_ctx* ctx, nrhip_station* st, const nrhip_sim_config* cfg, int64_t n_showers,
                                const double* vertex, const double* zenith, const double* azimuth, const double* energy,
                                const int32_t* shower_type, const double* k_L, const double* vertex_time,
                                const double* max_distance, int64_t n_groups, const int32_t* group_begin,
                                uint8_t* triggered, nrhip_sim_stats* stats)
{
    if (!ctx || !st || !cfg) return nrhip_fail_msg("nrhip_simulate_events: NULL argument");
    if (!st->ctx) return nrhip_fail_msg("nrhip_simulate_events: the station's context has been destroyed");
    if (st->ctx != ctx) return nrhip_fail_msg("nrhip_simulate_events: station belongs to another context");
    if (n_showers < 0 || n_groups < 0 || n_groups > n_showers) return nrhip_fail_msg("nrhip_simulate_events: bad sizes");
    if (!group_begin && n_groups != n_showers)
        return nrhip_fail_msg("nrhip_simulate_events: group_begin is required when groups hold several showers");
    const int64_t n_events = n_showers;  // ray stages work per shower, decision stages per event group
    if (cfg->askaryan_model < 0 || cfg->askaryan_model > NRHIP_ASK_ARZ2020)
        return nrhip_fail_msg("nrhip_simulate_events: Askaryan model not implemented");
    nrhip_sim_stats S;
    memset(&S, 0, sizeof S);
    S.n_events = n_groups;
    st->ws_bytes.clear();
    st->last_dump_items = -1;
    if (stats) *stats = S;
    if (n_events == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    const StationDev& sd = st->dev;
    const int n_ch = sd.n_ch;
    // solution slots per pair: 2, or 2 + 4 n with n reflections off the bottom of an ice shelf (propagation_base_class.py:424-429)
    const int n_refl = cfg->n_reflections;
    if (n_refl < 0 || n_refl > NRHIP_MAX_REFLECTIONS) return nrhip_fail_msg("nrhip_simulate_events: n_reflections must be 0..4");
    if (n_refl > 0 && cfg->given_C0)
        return nrhip_fail_msg("nrhip_simulate_events: given ray solutions (given_C0) are not available together with bottom reflections");
    if (n_refl > 0 && !(cfg->z_reflection < 0))
        return nrhip_fail_msg("nrhip_simulate_events: reflections off the bottom are requested, but no reflective layer is given (z_reflection)");
    const int S_ = n_refl > 0 ? 2 + 4 * n_refl : NRHIP_MAXS, NS_ = n_refl + 1;
    const long n_pairs = n_events * n_ch, n_slots = n_pairs * S_;
    if (n_slots > 2000000000L) return nrhip_fail_msg("nrhip_simulate_events: batch too large, split the event list");
    S.n_pairs = n_pairs;
    unsigned long long* rt_eval_counter = nullptr;
    unsigned long long* general_counters = nullptr;   // [0] ARZ integrand evaluations, [1] (step, bin) pairs of the birefringent propagation
    // general path with birefringence, production mode: the propagation in two rounds (see "two rounds" below)
    bool two_rounds = false;
    BireBatch bb_keep{};
    double *steps_keep = nullptr, *traces_keep = nullptr;
    double2* spec_keep = nullptr;
    int* gactive_keep = nullptr;
    // general emission / propagation path?
    const bool arz = cfg->askaryan_model == NRHIP_ASK_ARZ2019 || cfg->askaryan_model == NRHIP_ASK_ARZ2020;
    const bool bire = st->bire_n_knots[0] > 0;
    const bool general = arz || bire;
    if (arz && st->arz_n_profiles <= 0)
        return nrhip_fail_msg("nrhip_simulate_events: the ARZ models need a shower library (nrhip_station_set_arz)");
    if (arz && !cfg->select_only && st->n_shower_profiles != n_events)
        return nrhip_fail_msg("nrhip_simulate_events: the ARZ models need one profile per shower (nrhip_station_set_shower_profiles)");
    if (cfg->n_reflections > 0 && bire)
        return nrhip_fail_msg("nrhip_simulate_events: bottom reflections are not available together with birefringence (the reference's own loop covers only the first part of such a path: tests/golden/ref_bire_reflection_probe.txt)");
    const bool phased = cfg->trigger_type == NRHIP_TRIG_PHASED_ARRAY;
    const bool envelope = cfg->trigger_type == NRHIP_TRIG_ENVELOPE;
    const bool noise = cfg->noise != 0;
    if (noise && !st->noise_set) return nrhip_fail_msg("nrhip_simulate_events: noise needs the per-channel amplitudes (nrhip_station_set_noise)");
    if (envelope && !st->env_set)
        return nrhip_fail_msg("nrhip_simulate_events: the envelope trigger needs its band pass (nrhip_station_set_envelope_trigger)");
    if (phased && st->pa_n_channels <= 0)
        return nrhip_fail_msg("nrhip_simulate_events: the phased-array trigger needs its channels and beams (nrhip_station_set_phased_array)");
    for (auto& e : st->evt) if (!e) HIPCHK(hipEventCreate(&e));
#define MARK(i) HIPCHK(hipEventRecord(st->evt[i], sm))
    MARK(0);
    if (!cfg->accumulate_triggered) HIPCHK(hipMemsetAsync(triggered, 0, n_groups, sm));

    // 1. ray tracing for every (event, channel) pair
    // (the finder without the hybr stage serves receivers down to 10 z_0; deeper receivers and end points exactly above each other
    // -- whatever the antennas' depth -- are flagged by it and taken by the reference's procedure in a second launch, which reads one
    // word per pair when nothing is flagged.  nrhip_ctx_set_ray_finder(NRHIP_FINDER_REFERENCE): the reference's procedure for all.)
    // antennas at very different depths (surface LPDAs next to deep dipoles): the finder walks the pairs channel-major
    bool spread_antennas = false;
    {
        double zlo = 0., zhi = -1e30;
        for (int c = 0; c < n_ch; c++) { zlo = std::min(zlo, st->h_pos[3 * c + 2]); zhi = std::max(zhi, st->h_pos[3 * c + 2]); }
        spread_antennas = n_ch > 1 && zhi - zlo > 20.;
    }
    RayRecords rec;
    NEED(rec.n_sol = WS("pair_n_sol", int, n_pairs));
    NEED(rec.type = WS("slot_type", int, n_slots));
    NEED(rec.C0 = WS("slot_C0", double, n_slots));
    NEED(rec.C1 = WS("slot_C1", double, n_slots));
    NEED(rec.D = WS("slot_D", double, n_slots));
    NEED(rec.T = WS("slot_T", double, n_slots));
    NEED(rec.launch = WS("slot_launch", double, 3 * n_slots));
    NEED(rec.receive = WS("slot_receive", double, 3 * n_slots));
    NEED(rec.refl_angle = WS("slot_refl_angle", double, n_slots));
    // events in geometry-cell order for the root finders (counting sort over 128 x 128 cells)
    int *geo_cell, *geo_hist, *geo_off, *geo_tmp, *geo_perm;
    NEED(geo_cell = WS("geo_cell", int, n_events));
    NEED(geo_hist = WS("geo_hist", int, 16384 + 1));
    NEED(geo_off = WS("geo_offset", int, 16384 + 1));
    NEED(geo_tmp = WS("scan_tmp7", int, scan_tiles(16384 + 1)));
    NEED(geo_perm = WS("geo_perm", int, n_events));
    // cfg->reuse_ray_tables: the ray records and the delta_C selection of the previous call on the SAME shower list, station
    // position and cuts are still in the workspace (two-phase runs: nrhip_sim_config.select_only first)
    const bool reuse = cfg->reuse_ray_tables != 0;
    if (reuse && (st->rays_n_showers != n_showers || st->rays_delta_C != cfg->delta_C_cut || st->rays_vertex != vertex ||
                  st->rays_n_groups != n_groups || st->rays_max_distance != max_distance || st->rays_n_reflections != n_refl ||
                  st->rays_z_reflection != cfg->z_reflection || st->rays_generation != st->generation))
        return nrhip_fail_msg("nrhip_simulate_events: reuse_ray_tables without matching ray tables of a previous call");
    st->rays_n_showers = -1;
    rec.stride = S_;
    double *seg_zint = nullptr, *seg_C0 = nullptr;
    if (n_refl > 0) {
        int *rf, *rc, *nseg, *smask, *cand_n;
        double* cand_C0;
        const int n_calls = 1 + 2 * n_refl;
        NEED(rf = WS("slot_reflection", int, n_slots));
        NEED(rc = WS("slot_reflection_case", int, n_slots));
        NEED(nseg = WS("slot_n_segments", int, n_slots));
        NEED(smask = WS("slot_surface_mask", int, n_slots));
        NEED(seg_zint = WS("slot_segment_limits", double, (size_t)n_slots * NS_ * 3));
        NEED(seg_C0 = WS("slot_segment_C0", double, (size_t)n_slots * NS_));
        NEED(cand_n = WS("refl_candidates_n", int, (size_t)n_pairs * n_calls));
        NEED(cand_C0 = WS("refl_candidates_C0", double, (size_t)n_pairs * n_calls * 3));
        rec.reflection = rf;
        rec.surface_mask = smask;
        if (!reuse) {
            // ray_tracing(medium, n_reflections = n).find_solutions (analyticraytracing.py:2118-2130): the plain finder plus, per
            // number of bottom reflections, one search for rays starting upwards and one for rays starting downwards
            ReflRecords rr{rec.n_sol, rec.type, rf, rc, nseg, smask, rec.C0, rec.C1, rec.D, rec.T, rec.launch, rec.receive,
                           rec.refl_angle, seg_zint, seg_C0};
            launch_find_refl(sm, n_pairs, n_refl, vertex, sd.pos, n_ch, ctx->ice, cfg->z_reflection, cand_n, cand_C0, ctx->ray_finder == NRHIP_FINDER_REFERENCE);
            launch_records_refl(sm, n_pairs, n_refl, S_, vertex, sd.pos, n_ch, ctx->ice, cfg->z_reflection, cand_n, cand_C0, 0, rr);
            if (max_distance) launch_distance_cut_pairs(sm, n_pairs, n_ch, vertex, sd.pos, max_distance, rec.n_sol);
            LCHK("raytrace (bottom reflections)");
        }
    } else if (!reuse) {
        HIPCHK(hipMemsetAsync(geo_hist, 0, sizeof(int) * 16385, sm));
        launch_event_cells(sm, (int)n_events, vertex, sd.pos, geo_cell, geo_hist);
        launch_exclusive_scan(sm, 16385, geo_hist, geo_off, geo_tmp);
        launch_event_perm(sm, (int)n_events, geo_cell, geo_off, geo_perm);
        LCHK("geometry order");
        if (stats) {
            NEED(rt_eval_counter = WS("rt_eval_counter", unsigned long long, 1));
            HIPCHK(hipMemsetAsync(rt_eval_counter, 0, sizeof(unsigned long long), sm));
        }
        launch_raytrace(sm, n_pairs, vertex, sd.pos, n_ch, ctx->ice, rec, max_distance, geo_perm, cfg->given_C0, rt_eval_counter,
                        cfg->given_D, cfg->given_T, ctx->ray_finder == NRHIP_FINDER_REFERENCE, spread_antennas);
        LCHK("raytrace");
    }
    MARK(1);

    // 2. delta_C cut -> ordered list of kept rays
    int *keep, *offset;
    NEED(keep = WS("slot_keep", int, n_slots + 1));
    NEED(offset = WS("slot_offset", int, n_slots + 1));
    int* scan_tmp;
    NEED(scan_tmp = WS("scan_tmp", int, scan_tiles(n_slots + 1)));
    if (!reuse) {
        HIPCHK(hipMemsetAsync(keep + n_slots, 0, sizeof(int), sm));
        launch_select_rays(sm, n_pairs, n_ch, vertex, zenith, azimuth, rec, ctx->ice, cfg->delta_C_cut, keep);
        LCHK("select_rays");
        launch_exclusive_scan(sm, n_slots + 1, keep, offset, scan_tmp);
        LCHK("scan");
    }
    int n_rays = 0;
    HIPCHK(hipMemcpyAsync(&n_rays, offset + n_slots, sizeof(int), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    S.n_rays = n_rays;
    st->rays_n_showers = n_showers;
    st->rays_n_groups = n_groups;
    st->rays_delta_C = cfg->delta_C_cut;
    st->rays_vertex = vertex;
    st->rays_max_distance = max_distance;
    st->rays_n_reflections = n_refl;
    st->rays_z_reflection = cfg->z_reflection;
    st->rays_generation = st->generation;
    if (cfg->select_only) {
        int* first;
        NEED(first = WS("shower_first_channel", int, n_showers));
        hipLaunchKernelGGL(shower_first_channel_kernel, dim3((unsigned)((n_showers + 255) / 256)), dim3(256), 0, sm, (long)n_showers,
                           n_ch, keep, first, S_);
        LCHK("shower_first_channel");
        HIPCHK(hipStreamSynchronize(sm));
        if (stats) *stats = S;
        return 0;
    }

    EventIn evin{energy, shower_type, k_L, vertex_time};
    // ray range of every event group (rays are ordered by shower; a group's showers are consecutive)
    int* grp_ray;
    NEED(grp_ray = WS("group_ray_begin", int, n_groups + 1));
    launch_group_ray_range(sm, (int)n_groups, group_begin, n_ch, offset, grp_ray, S_);
    LCHK("group ranges");

    RayWork w;
    const size_t nr = (size_t)std::max(n_rays, 1);
    int* ray_slot;
    NEED(ray_slot = WS("ray_slot", int, nr));
    NEED(w.ask = WS("ray_askaryan", AskaryanConst, nr));
    NEED(w.ev = WS("ray_event", int, nr));
    NEED(w.ch = WS("ray_channel", int, nr));
    NEED(w.sol = WS("ray_solution", int, nr));
    NEED(w.slot = WS("ray_slot2", int, nr));
    NEED(w.view = WS("ray_view", double, nr));
    NEED(w.n_index = WS("ray_n_index", double, nr));
    NEED(w.R = WS("ray_D", double, nr));
    NEED(w.t0 = WS("ray_t0", double, nr));
    NEED(w.C0 = WS("ray_C0", double, nr));
    NEED(w.pol_theta = WS("ray_pol_theta", double, nr));
    NEED(w.pol_phi = WS("ray_pol_phi", double, nr));
    NEED(w.r_theta = WS("ray_r_theta", double2, nr));
    NEED(w.r_phi = WS("ray_r_phi", double2, nr));
    NEED(w.zen = WS("ray_zenith", double, nr));
    NEED(w.az = WS("ray_azimuth", double, nr));
    NEED(w.vel_T = WS("ray_vel_T", double, 4 * nr));
    NEED(w.theta_ant = WS("ray_theta_ant", double, nr));
    NEED(w.phi_ant = WS("ray_phi_ant", double, nr));
    NEED(w.vfac_t = WS("ray_vfac_theta", double, nr));
    NEED(w.vfac_p = WS("ray_vfac_phi", double, nr));
    NEED(w.tab = WS("ray_antenna_table", int, nr));
    NEED(w.att = WS("ray_att", double, nr * sd.n_fc));
    NEED(w.e_norm = WS("ray_e_norm", double, nr));
    NEED(w.focus = WS("ray_focus", double, nr));
    double *zint, *max_efield;
    NEED(zint = WS("ray_zint", double, 3 * nr));
    NEED(max_efield = WS("ray_max_efield", double, nr));

    const int* foc_n_sol = nullptr;
    const double* foc_launch = nullptr;
    const double foc_dz = -0.01;  // get_focusing(dz = -1 cm)
    if (n_rays > 0) {
        launch_scatter_slots(sm, n_slots, keep, offset, ray_slot);
        LCHK("scatter");
        if (cfg->focusing) {
            // second trace to the receivers moved by dz (analyticraytracing.py:2778-2888); only n_sol and launch are used
            RayRecords rec2;
            double* pos2;
            NEED(pos2 = WS("foc_positions", double, 3 * n_ch));
            std::vector<double> hp(st->h_pos);
            for (int c = 0; c < n_ch; c++) hp[3 * c + 2] += foc_dz;
            HIPCHK(hipMemcpyAsync(pos2, hp.data(), sizeof(double) * 3 * n_ch, hipMemcpyHostToDevice, sm));
            NEED(rec2.n_sol = WS("foc_n_sol", int, n_pairs));
            NEED(rec2.type = WS("foc_type", int, n_slots));
            NEED(rec2.C0 = WS("foc_C0", double, n_slots));
            NEED(rec2.C1 = WS("foc_C1", double, n_slots));
            NEED(rec2.D = WS("foc_D", double, n_slots));
            NEED(rec2.T = WS("foc_T", double, n_slots));
            NEED(rec2.launch = WS("foc_launch", double, 3 * n_slots));
            NEED(rec2.receive = WS("foc_receive", double, 3 * n_slots));
            NEED(rec2.refl_angle = WS("foc_refl_angle", double, n_slots));
            if (n_refl > 0) {
                // the reference's second tracer is built with the same n_reflections (analyticraytracing.py:2835-2840): solution
                // iS of its list -- plain solutions, then per number of reflections the two starting directions -- is the slot
                int *rf2, *rc2, *nseg2, *smask2, *cand_n2;
                double *zint2, *segC2, *cand_C2;
                const int n_calls = 1 + 2 * n_refl;
                NEED(rf2 = WS("foc_reflection", int, n_slots));
                NEED(rc2 = WS("foc_reflection_case", int, n_slots));
                NEED(nseg2 = WS("foc_n_segments", int, n_slots));
                NEED(smask2 = WS("foc_surface_mask", int, n_slots));
                NEED(zint2 = WS("foc_segment_limits", double, (size_t)n_slots * NS_ * 3));
                NEED(segC2 = WS("foc_segment_C0", double, (size_t)n_slots * NS_));
                NEED(cand_n2 = WS("foc_candidates_n", int, (size_t)n_pairs * n_calls));
                NEED(cand_C2 = WS("foc_candidates_C0", double, (size_t)n_pairs * n_calls * 3));
                ReflRecords rr2{rec2.n_sol, rec2.type, rf2, rc2, nseg2, smask2, rec2.C0, rec2.C1, rec2.D, rec2.T, rec2.launch, rec2.receive,
                                rec2.refl_angle, zint2, segC2};
                launch_find_refl(sm, n_pairs, n_refl, vertex, pos2, n_ch, ctx->ice, cfg->z_reflection, cand_n2, cand_C2, ctx->ray_finder == NRHIP_FINDER_REFERENCE);
                launch_records_refl(sm, n_pairs, n_refl, S_, vertex, pos2, n_ch, ctx->ice, cfg->z_reflection, cand_n2, cand_C2, 0, rr2);
            } else {
                launch_raytrace(sm, n_pairs, vertex, pos2, n_ch, ctx->ice, rec2, max_distance, geo_perm, nullptr, nullptr, nullptr, nullptr,
                                ctx->ray_finder == NRHIP_FINDER_REFERENCE, spread_antennas);
            }
            LCHK("raytrace (focusing)");
            HIPCHK(hipStreamSynchronize(sm));  // hp goes out of scope
            foc_n_sol = rec2.n_sol;
            foc_launch = rec2.launch;
        }
        launch_ray_setup(sm, n_rays, n_ch, ray_slot, vertex, zenith, azimuth, rec, ctx->ice, sd, w, evin,
                         arz ? NRHIP_ASK_ALVAREZ2009 : cfg->askaryan_model, foc_n_sol, foc_launch, foc_dz, cfg->focusing_limit > 0 ? cfg->focusing_limit : 2.,
                         cfg->reflection_coefficient, cfg->reflection_phase_shift, cfg->custom_polarization ? cfg->polarization_ephi : NAN);
        LCHK("ray_setup");
    }
    // split_event_time_diff: sub-events of the groups (stages 3-4 stay per group: the candidate flag is the group's)
    long n_ev = n_groups;            // readouts: one per group, or one per sub-event
    const int* ev_ray = grp_ray;     // their ray ranges
    int* ev_group = nullptr;         // sub-event -> group (NULL: identity)
    int* ev_sub_dev = nullptr;       // sub-event index inside its group
    int* ev_base = nullptr;          // group -> first sub-event
    if (cfg->split_event_time_diff > 0 && n_rays > 0) {
        int *slot_new, *sub_sorted, *order, *sub_of, *n_sub, *etmp, *any_split;
        NEED(slot_new = WS("ray_slot_split", int, nr));
        NEED(sub_sorted = WS("ray_sub_event", int, nr));
        NEED(order = WS("ray_time_order", int, nr));
        NEED(sub_of = WS("ray_sub_event_unsorted", int, nr));
        NEED(n_sub = WS("group_n_sub_events", int, n_groups + 1));
        NEED(ev_base = WS("group_first_sub_event", int, n_groups + 1));
        NEED(etmp = WS("scan_tmp8", int, scan_tiles(n_groups + 1)));
        NEED(any_split = WS("any_split", int, 1));
        HIPCHK(hipMemsetAsync(any_split, 0, sizeof(int), sm));
        HIPCHK(hipMemsetAsync(n_sub + n_groups, 0, sizeof(int), sm));
        hipLaunchKernelGGL(sub_event_split_kernel, dim3((unsigned)((n_groups + 127) / 128)), dim3(128), 0, sm, (int)n_groups, grp_ray,
                           w.t0, w.ch, sd.cable, cfg->split_event_time_diff, ray_slot, slot_new, sub_sorted, order, sub_of, n_sub,
                           any_split);
        launch_exclusive_scan(sm, n_groups + 1, n_sub, ev_base, etmp);
        int h_split[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(&h_split[0], any_split, sizeof(int), hipMemcpyDeviceToHost, sm));
        HIPCHK(hipMemcpyAsync(&h_split[1], ev_base + n_groups, sizeof(int), hipMemcpyDeviceToHost, sm));
        HIPCHK(hipStreamSynchronize(sm));
        if (h_split[0]) {
            // the work table in the new ray order: the slot list is permuted inside the split groups and the per-ray set-up redone
            HIPCHK(hipMemcpyAsync(ray_slot, slot_new, sizeof(int) * (size_t)n_rays, hipMemcpyDeviceToDevice, sm));
            launch_ray_setup(sm, n_rays, n_ch, ray_slot, vertex, zenith, azimuth, rec, ctx->ice, sd, w, evin,
                             arz ? NRHIP_ASK_ALVAREZ2009 : cfg->askaryan_model,
                             foc_n_sol, foc_launch, foc_dz, cfg->focusing_limit > 0 ? cfg->focusing_limit : 2.,
                             cfg->reflection_coefficient, cfg->reflection_phase_shift,
                             cfg->custom_polarization ? cfg->polarization_ephi : NAN);
            n_ev = h_split[1];
            int *sub_ray, *ev_sub;
            NEED(sub_ray = WS("sub_event_ray_begin", int, n_ev + 1));
            NEED(ev_group = WS("ev_group", int, n_ev));
            NEED(ev_sub = WS("ev_sub_event", int, n_ev));
            ev_sub_dev = ev_sub;
            hipLaunchKernelGGL(sub_event_ranges_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, sm, (int)n_groups,
                               grp_ray, sub_sorted, ev_base, sub_ray, ev_group, ev_sub);
            ev_ray = sub_ray;
            LCHK("sub-events");
        } else {
            ev_base = nullptr;
        }
    }
    S.n_sub_events = (int32_t)n_ev;
    EventOut ev;
    NEED(ev.n_rays = WS("ev_n_rays", int, n_ev));
    NEED(ev.ray_begin = WS("ev_ray_begin", int, n_ev));
    NEED(ev.L = WS("ev_L", int, n_ev));
    NEED(ev.candidate = WS("ev_candidate", unsigned char, n_ev));
    NEED(ev.t_min = WS("ev_t_min", double, n_ev));
    int* trigger_bin;
    NEED(trigger_bin = WS("ev_trigger_bin", int, n_ev));
    HIPCHK(hipMemsetAsync(trigger_bin, 0xFF, sizeof(int) * n_ev, sm));
    unsigned char* ev_triggered = triggered;   // per readout; OR-ed into the groups' mask at the end when groups were split
    if (ev_group) {
        NEED(ev_triggered = WS("ev_triggered", unsigned char, n_ev));
        HIPCHK(hipMemsetAsync(ev_triggered, 0, (size_t)n_ev, sm));
    }
    MARK(2);
    // 3. which rays can matter at all?  un-attenuated sum-of-magnitudes bound per ray -> per event -> active ray list
    int n_active = 0;
    int* active_list = nullptr;
    double* bound;
    NEED(bound = WS("ray_bound", double, nr));
    // attenuation in two stages (result-neutral): first the rays whose own bound exceeds the candidate cut (they decide which
    // events become candidates), after the candidate cut the other rays of the candidate readouts (the channel traces of a
    // candidate sum all its rays) -- the rays that are active only because an event-mate might have made the event a candidate,
    // and did not, are never integrated (10 % of the quadratures of the survey)
    // (whether that pays depends on the workload -- few candidates, few followers: one launch less of everything; many candidates:
    // the second stage repeats fixed costs for little saved work -- so, like conv_mode, a station times one call of each and keeps
    // the faster; the bits are the same)
    // (not with dump_traces: there every active ray gets its exact max |e| and the followers of a second stage would only get a
    // bound -- which scheme runs is a timing decision, and no fetchable table may depend on one)
    const bool two_stage_ok = !general && n_refl == 0 && !cfg->no_pruning && !cfg->dump_traces;
    const bool att_tunable = two_stage_ok && n_rays >= 20000 && !getenv("NRHIP_ATT_ONE_STAGE") && !getenv("NRHIP_ATT_TWO_STAGE");
    int att_trial = -1;
    bool two_stage = two_stage_ok && st->att_mode != 2;
    if (getenv("NRHIP_ATT_ONE_STAGE")) two_stage = false;
    else if (getenv("NRHIP_ATT_TWO_STAGE")) two_stage = two_stage_ok;
    else if (att_tunable && st->att_mode == 0 && st->att_calls++ >= 1) {
        att_trial = (st->att_ms_per_ray[0] == 0.) ? 0 : 1;
        two_stage = att_trial == 0;
    }
    int* ractive = nullptr;
    if (n_rays > 0) {
        int *roff, *rtmp, *cflags;
        NEED(ractive = WS("ray_active", int, nr + 1));
        // (the lists by predicted work count QC_NC classes per block of 256 rays: more than 3 n + 1 entries for a handful of rays)
        const size_t n_cls = std::max<size_t>(3 * nr + 1, (size_t)quad_class_entries(n_rays));
        NEED(cflags = WS("ray_active_class", int, n_cls));
        NEED(roff = WS("ray_active_offset", int, n_cls));
        NEED(rtmp = WS("scan_tmp2", int, scan_tiles((long)n_cls)));
        NEED(active_list = WS("ray_active_list", int, nr));
        launch_ray_limits_from_slots(sm, n_rays, n_ch, ray_slot, vertex, sd.pos, rec, ctx->ice, zint);
        LCHK("ray_limits");
        if (general) {  // every kept ray is evaluated
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ractive, 1, (size_t)n_rays, sm));
            HIPCHK(hipMemsetAsync(bound, 0xFF, sizeof(double) * nr, sm));
        } else {
            // (cut: the candidate cut the bound will be compared with -- rays far below it get a 64-term bound only; < 0: all exact)
            launch_amp_bound(sm, n_rays, w, sd, ctx->ice, vertex, zint, bound, max_efield,
                             cfg->no_pruning ? -1.0 : cfg->min_efield_amplitude);
            LCHK("amp_bound");
            launch_event_possible(sm, (int)n_groups, n_ch, grp_ray, bound, cfg->no_pruning ? -1.0 : cfg->min_efield_amplitude,
                                  ractive, two_stage ? 1 : 0);
            LCHK("event_possible");
        }
        // active rays listed by predicted quadrature work: wave-mates do similar numbers of bisections (spectral.hip, quad_class)
        if (n_refl == 0 && !getenv("NRHIP_ATT_TYPE_CLASSES")) {
            signed char* qcls;
            NEED(qcls = WS("ray_quad_class", signed char, nr));
            launch_quad_class_list(sm, n_rays, ractive, w.slot, rec.type, w.C0, zint, ctx->ice, qcls, cflags, roff, rtmp, active_list);
            LCHK("active list");
            HIPCHK(hipMemcpyAsync(&n_active, roff + quad_class_entries(n_rays) - 1, sizeof(int), hipMemcpyDeviceToHost, sm));
        } else {   // (with bottom reflections the path is made of segments: by solution type, as in rounds 1-2)
            launch_active_class_flags(sm, n_rays, ractive, w.slot, rec.type, cflags);
            HIPCHK(hipMemsetAsync(cflags + 3L * n_rays, 0, sizeof(int), sm));
            launch_exclusive_scan(sm, 3L * n_rays + 1, cflags, roff, rtmp);
            launch_scatter_active_class(sm, n_rays, cflags, roff, active_list);
            LCHK("active list");
            HIPCHK(hipMemcpyAsync(&n_active, roff + 3L * n_rays, sizeof(int), hipMemcpyDeviceToHost, sm));
        }
        HIPCHK(hipMemsetAsync(w.att, 0xFF, nr * sd.n_fc * sizeof(double), sm));  // NaN = not evaluated
        HIPCHK(hipStreamSynchronize(sm));
    }
    S.n_active_rays = n_active;
    unsigned long long* eval_counter = nullptr;
    unsigned long long* xform_count;  // [0] channel traces computed, [1] rays in them, [2] rays transformed for the candidate cut,
                                      // [3] 8192-point chirp convolutions of the trigger-ADC chain
    NEED(xform_count = WS("transform_count", unsigned long long, 5));   // ([4]: rays sampled by efield_sample_kernel)
    HIPCHK(hipMemsetAsync(xform_count, 0, 5 * sizeof(unsigned long long), sm));
    MARK(3);
    if (n_active > 0) {
        // attenuation on the coarse frequency grid, active rays only
        NEED(eval_counter = WS("att_eval_counter", unsigned long long, 1));
        HIPCHK(hipMemsetAsync(eval_counter, 0, sizeof(unsigned long long), sm));
        if (ctx->att_model == NRHIP_ATT_GL3 && (!ctx->gl3 || sd.n_fc > 32))
            return nrhip_fail_msg("nrhip_simulate_events: GL3 needs nrhip_ctx_set_gl3_table and at most 32 attenuation frequencies");
        if (n_refl > 0) {
            // get_attenuation_along_path with bottom reflections (:933-1089): the product over the path segments, each with its
            // own launch parameter and limits (segment tables of the ray records)
            double *rs_C0, *rs_zint, *rs_att;
            int* items;
            NEED(rs_C0 = WS("ray_segment_C0", double, nr * NS_));
            NEED(rs_zint = WS("ray_segment_limits", double, nr * NS_ * 3));
            NEED(rs_att = WS("ray_segment_att", double, nr * NS_ * sd.n_fc));
            NEED(items = WS("ray_segment_items", int, (size_t)n_active * NS_));
            launch_gather_segments(sm, n_rays, NS_, ray_slot, seg_C0, seg_zint, rs_C0, rs_zint);
            launch_segment_items(sm, n_active, NS_, active_list, items);
            int* att_ovf;
            NEED(att_ovf = WS("att_overflow", int, 2 * (size_t)n_active * NS_ + 1));
            launch_attenuation_items(sm, (long)n_active * NS_, rs_C0, rs_zint, sd.n_fc, sd.fcoarse, ctx->att_model, ctx->ice, rs_att,
                                     nullptr, items, eval_counter, ctx->gl3, ctx->gl3_n, att_ovf);
            launch_segment_product_rays(sm, n_active, NS_, sd.n_fc, active_list, rs_C0, rs_att, w.att);
        } else {
            int* att_ovf;
            NEED(att_ovf = WS("att_overflow", int, 2 * (size_t)n_active + 1));
            launch_attenuation_items(sm, n_active, w.C0, zint, sd.n_fc, sd.fcoarse, ctx->att_model, ctx->ice, w.att, nullptr,
                                     active_list, eval_counter, ctx->gl3, ctx->gl3_n, att_ovf);
        }
        LCHK("attenuation");
    }
    MARK(4);
    const double* ray_traces = nullptr;
    const double2* general_spec = nullptr;   // the rays' on-sky spectra of the general path (amp_per_ray reads them)
    if (general && n_rays > 0) {
        // 4. general path: spectra of all kept rays -> (birefringence) -> traces and their maxima
        const int n_f = sd.N / 2 + 1;
        double2* spec;
        double *traces, *g_energy, *g_em, *g_resc, *g_x1, *g_x2;
        int *g_type, *g_prof, *g_nsteps, *g_npoints;
        NEED(spec = WS("ray_spectra", double2, nr * 2 * n_f));
        NEED(traces = WS("ray_traces", double, nr * 2 * sd.N));
        NEED(g_energy = WS("gen_energy", double, nr));
        NEED(g_em = WS("gen_em_factor", double, nr));
        NEED(g_resc = WS("gen_rescale", double, nr));
        NEED(g_x1 = WS("gen_x1", double, 3 * nr));
        NEED(g_x2 = WS("gen_x2", double, 3 * nr));
        NEED(g_type = WS("gen_type", int, nr));
        NEED(g_prof = WS("gen_profile", int, nr));
        NEED(g_nsteps = WS("gen_n_steps", int, nr));
        NEED(g_npoints = WS("gen_n_points", int, nr));
        if (stats) {   // work counters of the emission / propagation kernels (bench.py prices the stage by them)
            NEED(general_counters = WS("general_counters", unsigned long long, 2));
            HIPCHK(hipMemsetAsync(general_counters, 0, 2 * sizeof(unsigned long long), sm));
        }
        launch_general_gather(sm, n_rays, n_ch, w, evin, sd, vertex, arz ? st->d_shower_profile.as<int>() : nullptr,
                              arz ? st->d_shower_rescale.as<double>() : nullptr, st->arz_em_formula, g_energy, g_type, g_em,
                              g_prof, g_resc, g_x1, g_x2, g_nsteps, g_npoints);
        LCHK("general gather");
        const double* arz_trace = nullptr;
        const int* arz_silent = nullptr;   // rays beyond the emission model's 20 degrees: zero traces that are not stored
        if (arz) {
            double *vp, *atr;
            int* ast;
            NEED(vp = WS("arz_vector_potential", double, nr * (sd.N + 1) * 2));
            NEED(atr = WS("arz_traces", double, nr * 3 * sd.N));
            NEED(ast = WS("arz_status", int, nr));
            int* vpr;   // per ray: the window of observer times whose vector potential is stored (zero elsewhere, never written)
            NEED(vpr = WS("arz_vp_window", int, 2 * nr));
            int* sil;
            NEED(sil = WS("arz_silent", int, nr));
            arz_silent = sil;
            HIPCHK(hipMemsetAsync(ast, 0, sizeof(int) * nr, sm));
            ArzBatch ab{(long)n_rays, g_energy, w.view, w.R, g_type, g_em, g_prof, g_resc, st->arz_n_profiles, st->arz_n_depth,
                        st->d_arz_depth.as<double>(), st->d_arz_ce.as<double>(), st->d_arz_par.as<double>(), sd.N, 1. / sd.fs,
                        1.78, st->arz_interp_factor2, 0, 20. * 0.017453292519943295, w.n_index};
            NEED(ab.form_factor_table = WS("arz_form_factor_table", double, (size_t)ARZ_TABLE_DOUBLES));
            ab.eval_count = general_counters;
            launch_arz(sm, ab, vp, atr, ast, vpr, sil);
            LCHK("arz");
            // rays beyond the model's 20 degrees carry no signal: no path steps for them (half of config 4's rays)
            if (bire && !getenv("NRHIP_BIRE_ALL_RAYS"))
                launch_silent_rays(sm, n_rays, ab.theta, ab.n_index_ray, ab.n_index, ab.maximum_angle, g_nsteps, g_npoints);
            std::vector<int> hs(n_rays);
            HIPCHK(hipMemcpyAsync(hs.data(), ast, sizeof(int) * nr, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            for (int v : hs)
                if (v) return nrhip_fail_msg("nrhip_simulate_events: ARZ: length of indices is not 2 nor 4 (more than two stretches of a profile radiate within 1 ns)");
            arz_trace = atr;
        }
        double* ray_amp = nullptr;
        if (ray_amp_in_hbm(sd.N)) NEED(ray_amp = WS("ray_amp_scratch", double, (size_t)RAY_AMP_ROWS * (sd.N / 2 + 1)));
        launch_general_spectrum(sm, n_rays, w, sd, cfg->askaryan_model, arz_trace, ctx->twiddle, spec, ray_amp, arz_silent);
        LCHK("general spectrum");
        if (bire) {
            std::vector<int> hn(n_rays);
            HIPCHK(hipMemcpyAsync(hn.data(), g_nsteps, sizeof(int) * nr, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            std::vector<long> off(nr + 1, 0);
            int max_points = 0;
            for (size_t i = 0; i < nr; i++) {
                off[i + 1] = off[i] + hn[i];
                max_points = std::max(max_points, hn[i] + 1);
            }
            long* d_off;
            double* steps;
            NEED(d_off = WS("bire_step_offset", long, nr + 1));
            NEED(steps = WS("bire_steps", double, (size_t)std::max<long>(off[nr], 1) * 5));
            HIPCHK(hipMemcpyAsync(d_off, off.data(), sizeof(long) * (nr + 1), hipMemcpyHostToDevice, sm));
            BireBatch bb{(long)n_rays, g_x1, g_x2, w.C0, g_npoints, d_off, ctx->ice, st->d_bire_knots.as<double>(),
                         st->d_bire_coeffs.as<double>(), {st->bire_n_knots[0], st->bire_n_knots[1], st->bire_n_knots[2]},
                         st->bire_n_ref, st->bire_angle, n_f, sd.fs};
            NEED(bb.spline_pieces = WS("bire_spline_pieces", double, (size_t)BIRE_MAX_KNOTS * 7));
            bb.counters = general_counters;
            S.n_bire_steps = (int64_t)off[nr];
            // the gain of the whole path is bounded by the product of the steps' ||R||^2: events none of whose rays can exceed
            // the candidate cut even so skip the propagation (result-neutral, like the bounds of the parametrised path)
            long long* log_gain;  // fixed point, BIRE_LOG_FIXED
            int* gactive;
            NEED(log_gain = WS("bire_log_gain", long long, nr));
            NEED(gactive = WS("ray_propagated", int, nr));
            launch_birefringence_steps(sm, bb, max_points, steps, log_gain);
            // Two rounds (round 4): the propagation is the most expensive stage of this path, and most rays of a possible event
            // decide nothing.  First only the rays whose own bound exceeds the candidate cut (they alone can make the event a
            // candidate: the flags are exact); later, for the candidate readouts, the rays of the channels that can reach the
            // trigger threshold at all (Cauchy-Schwarz with the field norms bounded through the path gain) -- the other channels are
            // not evaluated, their rays never propagated.  Only where a plain threshold decides and nothing else wants the traces.
            two_rounds = !cfg->no_pruning && !cfg->dump_traces && !cfg->amp_per_ray && !phased && !envelope && !noise &&
                         cfg->trigger_type == NRHIP_TRIG_SIMPLE && cfg->n_coincidences <= 1 && !(cfg->split_event_time_diff > 0) &&
                         n_refl == 0 && !getenv("NRHIP_GENERAL_ONE_ROUND");
            launch_general_bound(sm, n_rays, sd, spec, log_gain, bound, two_rounds ? w.e_norm : nullptr);
            launch_event_possible(sm, (int)n_groups, n_ch, grp_ray, bound,
                                  (cfg->no_pruning || cfg->dump_traces) ? -1.0 : cfg->min_efield_amplitude, gactive, two_rounds ? 1 : 0);
            launch_birefringence_propagate(sm, bb, steps, spec, gactive);
            LCHK("birefringence");
            HIPCHK(hipStreamSynchronize(sm));  // `off` goes out of scope
            if (two_rounds) { bb_keep = bb; steps_keep = steps; spec_keep = spec; traces_keep = traces; gactive_keep = gactive; }
            launch_general_trace(sm, n_rays, sd, spec, ctx->twiddle, traces, max_efield, gactive, bound);
        } else {
            launch_general_trace(sm, n_rays, sd, spec, ctx->twiddle, traces, max_efield);
        }
        LCHK("general trace");
        ray_traces = traces;
        general_spec = spec;
    }
    if (!general && n_active > 0) {
        // 4. candidate cut on max |E(t)|
        int *need_ray, *ev_need, *ev_off, *ev_tmp, *ev_list;
        NEED(need_ray = WS("ray_need_transform", int, nr));
        NEED(ev_need = WS("ev_need_transform", int, n_groups + 1));
        NEED(ev_off = WS("ev_need_offset", int, n_groups + 1));
        NEED(ev_tmp = WS("scan_tmp3", int, scan_tiles(n_groups + 1)));
        NEED(ev_list = WS("ev_transform_list", int, n_groups));
        double* ray_amp = nullptr;
        if (ray_amp_in_hbm(sd.N)) NEED(ray_amp = WS("ray_amp_scratch", double, (size_t)RAY_AMP_ROWS * (sd.N / 2 + 1)));
        launch_efield_max(sm, n_active, active_list, n_rays, (int)n_groups, grp_ray, w, evin, sd, cfg->askaryan_model,
                          ctx->twiddle, cfg->min_efield_amplitude, (cfg->no_pruning || cfg->dump_traces) ? 1 : 0, max_efield,
                          need_ray, ev_need, ev_off, ev_tmp, ev_list, xform_count, ray_amp);
        LCHK("efield_max");
    }
    MARK(5);
    // 5. common time grid per event
    launch_event_grid(sm, (int)n_ev, n_ch, ev_ray, w, sd, max_efield, cfg->min_efield_amplitude, ev);
    if (ev_group)
        hipLaunchKernelGGL(sub_event_candidate_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, sm, (int)n_groups, ev_base,
                           ev.candidate, ev.n_rays);
    LCHK("event_grid");
    // candidate list, distinct trace lengths and the per-event table index are built on the device; the host only
    // learns the counts (and the few distinct lengths) it needs to size the next launches
    const int n_half = NRHIP_SPEC_STRIDE;  // possible values of L / 2
    int *cflag, *coff, *ctmp, *lflag, *loff, *ltmp, *d_lens, *d_len_index, *d_cand;
    long long* d_ncr;
    NEED(cflag = WS("cand_flag", int, n_ev + 1));
    NEED(coff = WS("cand_offset", int, n_ev + 1));
    NEED(ctmp = WS("scan_tmp5", int, scan_tiles(n_ev + 1)));
    NEED(lflag = WS("len_flag", int, n_half + 1));
    NEED(loff = WS("len_offset", int, n_half + 1));
    NEED(ltmp = WS("scan_tmp6", int, scan_tiles(n_half + 1)));
    NEED(d_lens = WS("lengths", int, n_half));
    NEED(d_len_index = WS("ev_len_index", int, n_ev));
    NEED(d_cand = WS("item_event", int, n_ev));
    NEED(d_ncr = WS("cand_ray_count", long long, 2));
    HIPCHK(hipMemsetAsync(lflag, 0, sizeof(int) * (n_half + 1), sm));
    HIPCHK(hipMemsetAsync(d_ncr, 0, 2 * sizeof(long long), sm));
    launch_candidate_flags(sm, (int)n_ev, n_half, ev, cflag, lflag, d_ncr);
    if (cfg->amp_per_ray)  // the per-efield voltages live on the N grid: tables of "L = N" are built with the others
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(lflag + sd.N / 2), 1, 1, sm));
    launch_exclusive_scan(sm, n_ev + 1, cflag, coff, ctmp);
    launch_exclusive_scan(sm, n_half + 1, lflag, loff, ltmp);
    launch_candidate_lists(sm, (int)n_ev, n_half, ev, cflag, coff, lflag, loff, d_cand, d_len_index, d_lens);
    LCHK("candidate lists");
    // second stage of the attenuation: list of the candidate readouts' rays that have none yet
    int *fol_flag = nullptr, *fol_off = nullptr, *fol_list = nullptr;
    int n_followers = 0;
    if (two_stage && n_rays > 0 && n_active > 0) {
        int* fol_tmp;
        const size_t n_fcls = std::max<size_t>(nr + 1, (size_t)quad_class_entries(n_rays));
        NEED(fol_flag = WS("follower_flag", int, nr + 1));
        NEED(fol_off = WS("follower_offset", int, n_fcls));
        NEED(fol_tmp = WS("scan_tmp9", int, scan_tiles((long)n_fcls)));
        NEED(fol_list = WS("follower_list", int, nr));
        if (!getenv("NRHIP_ATT_TYPE_CLASSES")) {   // like the first stage's list: in the order of the predicted quadrature work
            launch_follower_list(sm, (int)n_ev, ev, w.att, sd.n_fc, n_rays, fol_flag, fol_off, fol_tmp, nullptr, ractive);
            int* fol_counts;
            signed char* qcls;
            NEED(fol_counts = WS("ray_active_class", int, std::max<size_t>(3 * nr + 1, n_fcls)));
            NEED(qcls = WS("ray_quad_class", signed char, nr));
            launch_quad_class_list(sm, n_rays, fol_flag, w.slot, rec.type, w.C0, zint, ctx->ice, qcls, fol_counts, fol_off, fol_tmp, fol_list);
            LCHK("follower list");
            HIPCHK(hipMemcpyAsync(&n_followers, fol_off + quad_class_entries(n_rays) - 1, sizeof(int), hipMemcpyDeviceToHost, sm));
        } else {
            launch_follower_list(sm, (int)n_ev, ev, w.att, sd.n_fc, n_rays, fol_flag, fol_off, fol_tmp, fol_list, ractive);
            LCHK("follower list");
            HIPCHK(hipMemcpyAsync(&n_followers, fol_off + n_rays, sizeof(int), hipMemcpyDeviceToHost, sm));
        }
    }
    int h_counts[2] = {0, 0};
    long long h_ncr[2] = {0, 0};
    std::vector<int> lens(n_half);
    HIPCHK(hipMemcpyAsync(&h_counts[0], coff + n_ev, sizeof(int), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipMemcpyAsync(&h_counts[1], loff + n_half, sizeof(int), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipMemcpyAsync(h_ncr, d_ncr, 2 * sizeof(long long), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipMemcpyAsync(lens.data(), d_lens, sizeof(int) * n_half, hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    const int n_cand = h_counts[0];
    lens.resize(h_counts[1]);
    st->ws_bytes["lengths"] = sizeof(int) * lens.size();
    st->ws_bytes["item_event"] = sizeof(int) * (size_t)n_cand;
    if (n_followers > 0) {
        int* att_ovf2;
        NEED(att_ovf2 = WS("att_overflow2", int, 2 * (size_t)n_followers + 1));
        MARK(10);
        launch_attenuation_items(sm, n_followers, w.C0, zint, sd.n_fc, sd.fcoarse, ctx->att_model, ctx->ice, w.att, nullptr, fol_list,
                                 eval_counter, ctx->gl3, ctx->gl3_n, att_ovf2);
        MARK(11);
        // their sum-of-magnitudes bound and Parseval norm with the computed attenuation (the channel prefilter multiplies the latter)
        launch_efield_bound_list(sm, n_followers, fol_list, w, sd, cfg->min_efield_amplitude, max_efield, fol_flag);
        LCHK("attenuation (second stage)");
        S.n_active_rays += n_followers;
    }

    MARK(6);
    MARK(7);
    MARK(8);
    const int maxL = h_ncr[1] > 0 ? (int)(2 * h_ncr[1]) : (lens.empty() ? 0 : lens.back());
    S.n_candidate_events = n_cand;
    S.max_length = maxL;
    S.n_candidate_rays = h_ncr[0];
    st->h_lengths = lens;
    if (n_cand > 0) {
        const int nh = sd.N / 2;
        // (the per-length tables hold 16384 bins; the forward chirp-z takes its outputs in blocks, the inverse one its outputs and,
        // for the longest traces, its inputs: every N shares the same limit)
        const int m_max = NRHIP_SPEC_STRIDE - 1;
        if (maxL / 2 > m_max)
            return nrhip_fail_msg("nrhip_simulate_events: an event's common trace is longer than 32766 samples (the per-length tables hold 16384 bins)");
        (void)nh;
        S.n_distinct_lengths = (int64_t)lens.size();
        MARK(6);
        // the tables of the lengths this station has not met before are built now, into new rows of its cache
        auto& tc = st->tabcache;
        if (tc.slot_of.empty()) tc.slot_of.assign(n_half, -1);
        std::vector<int> new_len, new_slot;
        for (int L_ : lens) {
            int& slot = tc.slot_of[L_ / 2];
            if (slot < 0) {
                slot = tc.n_slots++;
                new_len.push_back(L_);
                new_slot.push_back(slot);
            }
        }
        const size_t row[9] = {(size_t)FFT_MAX * 16, (size_t)FFT_MAX * 16, NRHIP_N_ANT_TAB * (size_t)NRHIP_SPEC_STRIDE * 16,
                               (size_t)NRHIP_E_STRIDE * 16, sd.n_fsets * (size_t)NRHIP_SPEC_STRIDE * 16, (size_t)NRHIP_SPEC_STRIDE * 16,
                               (size_t)FFT_MAX * 16, sd.n_fsets * (size_t)NRHIP_N_ANT_TAB * 8,
                               sd.n_fsets * (size_t)NRHIP_N_ANT_TAB * NRHIP_G_STRIDE * 16};
        DevArray* arr[9] = {&tc.B_fwd, &tc.B_inv, &tc.vel, &tc.E, &tc.H, &tc.Cf, &tc.Ci, &tc.hnorm, &tc.G};
        if (tc.n_slots > tc.cap) {   // grow, keeping the rows that exist
            const int old_rows = tc.n_slots - (int)new_len.size();
            const int new_cap = std::max(tc.n_slots, std::max(tc.cap + tc.cap / 2, 64));
            for (int a = 0; a < 9; a++) {
                DevArray fresh;
                if (fresh.reserve(row[a] * (size_t)new_cap) != hipSuccess)
                    return nrhip_fail_msg("nrhip_simulate_events: out of device memory (per-length tables)");
                if (old_rows > 0) HIPCHK(hipMemcpyAsync(fresh.p, arr[a]->p, row[a] * (size_t)old_rows, hipMemcpyDeviceToDevice, sm));
                HIPCHK(hipStreamSynchronize(sm));
                arr[a]->release();
                *arr[a] = fresh;
            }
            tc.cap = new_cap;
        }
        LengthTables tab;
        tab.B_fwd = tc.B_fwd.as<double2>(); tab.B_inv = tc.B_inv.as<double2>(); tab.vel = tc.vel.as<double2>();
        tab.E = tc.E.as<double2>(); tab.H = tc.H.as<double2>(); tab.Cf = tc.Cf.as<double2>(); tab.Ci = tc.Ci.as<double2>();
        tab.hnorm = tc.hnorm.as<double>(); tab.G = tc.G.as<double2>();
        if (tc.slotmap.reserve(sizeof(int) * (size_t)n_half) != hipSuccess)
            return nrhip_fail_msg("nrhip_simulate_events: out of device memory (length slot map)");
        if (!new_len.empty()) {
            int *d_new_len, *d_new_slot;
            NEED(d_new_len = WS("new_lengths", int, new_len.size()));
            NEED(d_new_slot = WS("new_length_slots", int, new_len.size()));
            HIPCHK(hipMemcpyAsync(d_new_len, new_len.data(), sizeof(int) * new_len.size(), hipMemcpyHostToDevice, sm));
            HIPCHK(hipMemcpyAsync(d_new_slot, new_slot.data(), sizeof(int) * new_slot.size(), hipMemcpyHostToDevice, sm));
            HIPCHK(hipMemcpyAsync(tc.slotmap.p, tc.slot_of.data(), sizeof(int) * (size_t)n_half, hipMemcpyHostToDevice, sm));
            launch_length_tables(sm, (int)new_len.size(), d_new_len, sd, st->d_filtersets.as<FilterSet>(), ctx->twiddle, ctx->w16, tab,
                                 d_new_slot);
            HIPCHK(hipStreamSynchronize(sm));   // the host vectors go out of scope
        }
        launch_length_slots(sm, (int)n_ev, ev.L, tc.slotmap.as<int>(), d_len_index);
        LCHK("length_tables");
        if (cfg->amp_per_ray && n_rays > 0) {
            double *max_env, *sig_time;
            NEED(max_env = WS("ray_max_amp_envelope", double, nr));
            NEED(sig_time = WS("ray_signal_time", double, nr));
            HIPCHK(hipMemsetAsync(max_env, 0xFF, sizeof(double) * nr, sm));
            HIPCHK(hipMemsetAsync(sig_time, 0xFF, sizeof(double) * nr, sm));
            double* env_amp = nullptr;      // N > 4096: the kernel's amplitude table in HBM scratch
            if (sd.N > FFT_MAX / 2) NEED(env_amp = WS("ray_amp_scratch", double, (size_t)RAY_AMP_ROWS * (sd.N / 2 + 1)));
            double2* env_nodes = nullptr;   // tabulated patterns: the angular interpolation at the table's frequency nodes, per block
            if (sd.ant_tabs) NEED(env_nodes = WS("antenna_table_nodes", double2, (size_t)channel_grid_blocks() * 2 * sd.max_tab_freq));
            launch_ray_envelope(sm, n_cand, coff + n_ev, d_cand, w, ev, sd, cfg->askaryan_model, ctx->twiddle, tab,
                                tc.slotmap.as<int>() + sd.N / 2, max_env, sig_time, general_spec, env_nodes, env_amp);
            LCHK("ray_envelope");
        }
        MARK(7);
        // 6. channel voltages + trigger
        const int n_items = n_cand * n_ch;
        S.n_channel_items = n_items;
        TriggerDev trg;
        trg.type = cfg->trigger_type == NRHIP_TRIG_HIGH_LOW ? 1 : (envelope ? 2 : 0);
        trg.n_coinc = cfg->n_coincidences > 1 ? cfg->n_coincidences : 1;
        trg.threshold = phased ? INFINITY : cfg->trigger_threshold;  // phased array: the channel stage only produces the traces
        trg.high = cfg->threshold_high;
        trg.low = cfg->threshold_low;
        trg.w_hl = std::max(1, (int)std::lrint(cfg->high_low_window * sd.fs));
        trg.w_coinc = std::max(1, (int)std::lrint(cfg->coinc_window * sd.fs));
        // high/low and coincidence triggers are fused into channel_conv_kernel; where that kernel does not run (common traces longer
        // than FFT_MAX samples, tabulated antenna patterns, the general path) the channel stage only produces the traces and
        // trace_trigger_kernel decides on them
        const bool post_trigger = trg.coincidence() && !phased &&
                                  (maxL > FFT_MAX || sd.N > FFT_MAX / 2 || sd.np.log2nh < 0 || getenv("NRHIP_CHANNEL_CZT") || sd.ant_tabs ||
                                   general || envelope || noise);
        TriggerDev trg_ch = trg;
        if (post_trigger) {
            trg_ch.type = 0;
            trg_ch.n_coinc = 1;
            trg_ch.threshold = INFINITY;
        }
        double* env_trace = nullptr;
        NoiseDev nz{noise ? 1 : 0, cfg->noise_seed, st->d_noise_amp.as<double>(), (const long long*)cfg->noise_group_id,
                    (long long)cfg->noise_group_offset, ev_group, ev_sub_dev};
        ChannelOut co;
        NEED(co.maxV = WS("item_maxV", double, n_items));
        co.trigger_bin = trigger_bin;
        co.triggered = ev_triggered;
        co.trace = nullptr;
        co.trace_offset = nullptr;
        if (cfg->dump_traces || phased || post_trigger) {
            std::vector<int> hL(n_ev), cand(n_cand);
            HIPCHK(hipMemcpyAsync(hL.data(), ev.L, sizeof(int) * n_ev, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipMemcpyAsync(cand.data(), d_cand, sizeof(int) * n_cand, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            std::vector<long> off(n_items + 1, 0);
            for (int i = 0; i < n_items; i++) off[i + 1] = off[i] + hL[cand[i / n_ch]];
            long* d_off;
            NEED(d_off = WS("trace_offset", long, n_items + 1));
            HIPCHK(hipMemcpyAsync(d_off, off.data(), sizeof(long) * (n_items + 1), hipMemcpyHostToDevice, sm));
            NEED(co.trace = WS("trace", double, std::max<long>(off[n_items], 1)));
            // channels without a ray (and channels outside the trigger set) are not written by the kernels: zeros, as the
            // reference's empty channels
            HIPCHK(hipMemsetAsync(co.trace, 0, sizeof(double) * (size_t)std::max<long>(off[n_items], 1), sm));
            co.trace_offset = d_off;
            st->last_dump_items = n_cand;
            if (envelope) {   // the envelopes of the band-passed channel traces, same layout
                NEED(env_trace = WS("envelope_trace", double, std::max<long>(off[n_items], 1)));
                HIPCHK(hipMemsetAsync(env_trace, 0, sizeof(double) * (size_t)std::max<long>(off[n_items], 1), sm));
            }
            HIPCHK(hipStreamSynchronize(sm));  // `off` goes out of scope
        }
        // traces of the events that trigger, written by the convolution kernel itself (nrhip_sim_config.emit_triggered_traces)
        const bool emit = cfg->emit_triggered_traces && !cfg->dump_traces && !cfg->no_pruning && !general && !phased && !post_trigger &&
                          !noise && !trg.coincidence();
        unsigned long long* emit_cursor = nullptr;
        if (emit) {
            co.emit_cap = cfg->emit_capacity_samples > 0 ? (long long)cfg->emit_capacity_samples : 400000000LL;
            // never more than every candidate event could need
            co.emit_cap = std::min<long long>(co.emit_cap, (long long)n_cand * n_ch * (long long)maxL);
            NEED(co.emit = WS("emit_trace", double, (size_t)std::max<long long>(co.emit_cap, 1)));
            NEED(co.emit_offset = WS("emit_offset", long long, n_ev));
            NEED(emit_cursor = WS("emit_cursor", unsigned long long, 3));
            co.emit_cursor = emit_cursor;
            HIPCHK(hipMemsetAsync(co.emit_offset, 0xFF, sizeof(long long) * (size_t)n_ev, sm));
            HIPCHK(hipMemsetAsync(emit_cursor, 0, 3 * sizeof(unsigned long long), sm));
        }
        double2* scratch;
        NEED(scratch = WS("channel_scratch", double2, (size_t)channel_grid_blocks() * 2 * NRHIP_SPEC_STRIDE));
        double* amp_scratch = nullptr;   // N > 4096: the chirp-z kernel's amplitude table does not fit behind its 128 KB buffer
        if (channel_amp_in_hbm(sd.N)) NEED(amp_scratch = WS("channel_amp_scratch", double, (size_t)channel_grid_blocks() * (sd.N / 2 + 1)));
        int *it_need, *it_off, *it_tmp, *it_list;
        NEED(it_need = WS("item_need", int, (size_t)n_items + n_cand + 2));
        NEED(it_off = WS("item_need_offset", int, (size_t)n_items + 1));
        NEED(it_tmp = WS("scan_tmp4", int, scan_tiles((long)n_items + 1)));
        NEED(it_list = WS("item_list", int, (size_t)n_items));
        unsigned char* it_sorted;   // the convolution kernel's event records and its list in the order of the trace lengths (spectral.hip)
        NEED(it_sorted = WS("conv_event_records", unsigned char, conv_ws_bytes(n_cand)));
        double* conv_noise = nullptr;   // the noise trace of the channel a block of the convolution kernel is working on
        if (noise) NEED(conv_noise = WS("conv_noise_trace", double, (size_t)channel_grid_blocks() * FFT_MAX));
        double2* conv_acc;  // frequency-domain sum over antenna tables (LPDA channels seeing rays in different lobes)
        NEED(conv_acc = WS("conv_table_sum", double2, (sd.tab_mask & 0x1c) ? (size_t)channel_grid_blocks() * FFT_MAX : 1));
        int* coinc_cnt;
        NEED(coinc_cnt = WS("coincidence_count", int, trg.coincidence() ? (size_t)channel_grid_blocks() * FFT_MAX : 1));
        double2* tab_nodes = nullptr;  // per block: the angular interpolation of a tabulated pattern at its frequency nodes
        if (sd.ant_tabs) NEED(tab_nodes = WS("antenna_table_nodes", double2, (size_t)channel_grid_blocks() * 2 * sd.max_tab_freq));
        // short events of the convolution kernel: split over two instantiations or not (same bits either way; ctx.h)
        const bool conv_tunable = sd.N < FFT_MAX / 2 && n_cand >= 2000 && !cfg->dump_traces && !getenv("NRHIP_CONV_ONE_BLOCK") && !getenv("NRHIP_CONV_SPLIT");
        int conv_trial = -1;
        bool conv_split = st->conv_mode != 2;
        if (getenv("NRHIP_CONV_SPLIT")) conv_split = true;
        else if (conv_tunable && st->conv_mode == 0 && st->conv_calls++ >= 1) {
            conv_trial = (st->conv_ms_per_event[0] == 0.) ? 0 : 1;
            conv_split = conv_trial == 0;
        }
        StationDev sd_ch = sd;
        if (phased) sd_ch.trig_on = st->d_pa_mask.as<unsigned char>();  // only the array's channels need traces (unless all are dumped)
        // analog phased array in production mode: events whose channel bounds cannot add up to the power threshold are not transformed
        // (window power <= (window / divisor) (sum_c max |V_c|)^2); the digitised array (comparator, up-sampling overshoot) and noise
        // have no such bound
        double pa_amp_cut = -1.;
        if (phased && !st->pa_adc_set && !noise && !general && !cfg->dump_traces && !cfg->no_pruning && cfg->trigger_threshold > 0) {
            const double divisor = st->pa_divisor > 0 ? (double)st->pa_divisor : (double)st->pa_window;
            pa_amp_cut = sqrt(cfg->trigger_threshold * divisor / (double)st->pa_window);
        }
        const int* general_need = nullptr;
        if (two_rounds) {
            int* fresh;
            NEED(fresh = WS("ray_propagated_late", int, nr));
            launch_general_prefilter(sm, n_items, d_cand, w, ev, d_len_index, sd_ch, trg_ch.threshold, tab.hnorm, co.maxV, it_need, n_rays,
                                     gactive_keep, fresh);
            launch_birefringence_propagate(sm, bb_keep, steps_keep, spec_keep, fresh);
            launch_general_trace(sm, n_rays, sd, spec_keep, ctx->twiddle, traces_keep, max_efield, fresh, nullptr);
            LCHK("birefringence (second round)");
            general_need = it_need;
        }
        launch_channel(sm, n_items, d_cand, w, evin, ev, d_len_index, sd_ch, st->filters[0], arz ? NRHIP_ASK_ALVAREZ2009 : cfg->askaryan_model,
                       trg_ch, ctx->twiddle, ctx->w16, tab, scratch, co,
                       (cfg->no_pruning || cfg->dump_traces || general || phased || post_trigger || noise) ? 1 : 0, maxL,
                       it_need, it_off, it_tmp, it_list, coinc_cnt, conv_acc, xform_count, tab_nodes, ray_traces,
                       (phased || post_trigger) ? (cfg->dump_traces ? 0 : 1) : -1, envelope ? &st->env_filter : nullptr, env_trace,
                       noise ? &nz : nullptr, conv_split, pa_amp_cut, amp_scratch, conv_noise, general_need, it_sorted);
        LCHK("channel");
        if (post_trigger) {
            if (maxL > 2 * FFT_MAX)
                return nrhip_fail_msg("nrhip_simulate_events: coincidence / high-low / envelope triggers on dumped traces take common traces of at most 16384 samples");
            launch_trace_trigger(sm, n_cand, d_cand, n_ch, ev.L, envelope ? env_trace : co.trace, co.trace_offset, trg, sd.trig_on, maxL,
                                 ev_triggered, trigger_bin);
            LCHK("trace trigger");
        }
        if (phased) {
            double* pa_max;
            NEED(pa_max = WS("pa_max_power", double, (size_t)n_cand * st->pa_n_beams));
            if (st->pa_adc_set) {
                // trigger ADC + up-sampling in front of the beams
                if (maxL > 9000) return nrhip_fail_msg("nrhip_simulate_events: the digitised phased array takes common traces of at most 9000 samples");
                PaAdc adc = st->pa_adc;
                const bool to5 = 5.0 > sd.fs;
                const double len5max = to5 ? (double)adc.p * maxL / adc.q : (double)maxL, cur = to5 ? 5.0 : sd.fs;
                adc.stride = adc.upsampling * ((int)(adc.adc_fs / cur * len5max) + 2) + 2;
                // a beam (and, for the ideal Hilbert transformer, a quarter beam of kernel values) has to fit the LDS of a CU
                if ((size_t)adc.stride * (adc.mode == 2 ? 10 : 8) + 128 + 2048 > 163840)
                    return nrhip_fail_msg(adc.mode == 2 ? "nrhip_simulate_events: the ideal Hilbert transformer takes up-sampled beams of at most 16 000 samples"
                                                        : "nrhip_simulate_events: the digitised phased array takes up-sampled beams of at most 20 000 samples");
                double* pa_trace;
                int* pa_len;
                NEED(pa_trace = WS("pa_digital_trace", double, (size_t)n_cand * st->pa_n_channels * adc.stride));
                NEED(pa_len = WS("pa_digital_length", int, (size_t)n_cand * st->pa_n_channels));
                // 'lin' / 'fir' up-sampling: the digitisers stop at the ADC trace (factor 1, own buffer), pa_upsample_kernel follows
                const PaAdc adc_final = adc;
                double* pa_final = pa_trace;
                int* len_final = pa_len;
                const bool own_upsampling = adc.up_method != 0 && adc.upsampling >= 2;
                if (own_upsampling) {
                    adc.upsampling = 1;
                    adc.stride = (int)(adc.adc_fs / cur * len5max) + 4;
                    NEED(pa_trace = WS("pa_adc_trace", double, (size_t)n_cand * st->pa_n_channels * adc.stride));
                    NEED(pa_len = WS("pa_adc_length", int, (size_t)n_cand * st->pa_n_channels));
                }
                const bool with_beams = !own_upsampling;
                if (pa_czt_applies(maxL, sd.fs, adc) && !getenv("NRHIP_PA_DIRECT")) {
                    // chirp-z transforms; their tables join the station's per-length cache (same slots)
                    if (st->pa_B_cap < tc.cap) {
                        DevArray fresh;
                        if (fresh.reserve((size_t)tc.cap * PA_TABLES * FFT_MAX * 16) != hipSuccess)
                            return nrhip_fail_msg("nrhip_simulate_events: out of device memory (digitiser tables)");
                        if (st->pa_B_cap > 0)
                            HIPCHK(hipMemcpyAsync(fresh.p, st->pa_B.p, (size_t)st->pa_B_cap * PA_TABLES * FFT_MAX * 16, hipMemcpyDeviceToDevice, sm));
                        HIPCHK(hipStreamSynchronize(sm));
                        st->pa_B.release();
                        st->pa_B = fresh;
                        st->pa_B_cap = tc.cap;
                    }
                    st->pa_built.resize((size_t)tc.cap, 0);
                    std::vector<int> pl, ps;
                    for (int L_ : lens) {
                        const int slot = tc.slot_of[L_ / 2];
                        if (!st->pa_built[slot]) { st->pa_built[slot] = 1; pl.push_back(L_); ps.push_back(slot); }
                    }
                    if (!pl.empty()) {
                        int *d_pl, *d_ps;
                        NEED(d_pl = WS("pa_new_lengths", int, pl.size()));
                        NEED(d_ps = WS("pa_new_slots", int, ps.size()));
                        HIPCHK(hipMemcpyAsync(d_pl, pl.data(), sizeof(int) * pl.size(), hipMemcpyHostToDevice, sm));
                        HIPCHK(hipMemcpyAsync(d_ps, ps.data(), sizeof(int) * ps.size(), hipMemcpyHostToDevice, sm));
                        launch_pa_czt_tables(sm, (int)pl.size(), d_pl, d_ps, sd.fs, adc, ctx->twiddle, st->pa_B.as<double2>());
                        HIPCHK(hipStreamSynchronize(sm));
                    }
                    const int chunk = std::min<long>((long)n_cand * st->pa_n_channels, 16384);
                    unsigned char* work;
                    NEED(work = WS("pa_digitiser_work", unsigned char, pa_czt_work_bytes(maxL, sd.fs, adc, chunk)));
                    launch_phased_array_digital_czt(sm, n_cand, d_cand, n_ch, ev.L, tc.slotmap.as<int>(), co.trace, co.trace_offset,
                                                    st->pa_n_channels, st->d_pa_channel.as<int>(), st->pa_n_beams,
                                                    st->d_pa_rolls_up.as<int>(), st->pa_window, st->pa_step, (double)st->pa_divisor,
                                                    cfg->trigger_threshold, maxL, sd.fs, adc, ctx->twiddle, ctx->w16 + (FFT_MAX / 2 + 1), st->pa_B.as<double2>(), work,
                                                    chunk, pa_trace, pa_len, ev_triggered, pa_max, with_beams, xform_count + 3);
                } else if (adc.clock_offset)
                    return nrhip_fail_msg("nrhip_simulate_events: the clock offset of the trigger ADC needs the chirp-z digitiser (sampling rate below "
                                          "5 GHz, an ADC rate other than the simulation's, common traces of at most 7169 samples)");
                else
                launch_phased_array_digital(sm, n_cand, d_cand, n_ch, ev.L, co.trace, co.trace_offset, st->pa_n_channels,
                                            st->d_pa_channel.as<int>(), st->pa_n_beams, st->d_pa_rolls_up.as<int>(), st->pa_window,
                                            st->pa_step, (double)st->pa_divisor, cfg->trigger_threshold, maxL, sd.fs, adc, pa_trace, pa_len,
                                            ev_triggered, pa_max, with_beams);
                if (own_upsampling) {
                    launch_pa_upsample(sm, n_cand * st->pa_n_channels, adc_final, pa_trace, adc.stride, pa_len, pa_final, len_final);
                    launch_phased_array_beams(sm, n_cand, d_cand, st->pa_n_channels, st->pa_n_beams, st->d_pa_rolls_up.as<int>(),
                                              st->pa_window, st->pa_step, (double)st->pa_divisor, cfg->trigger_threshold, adc_final,
                                              pa_final, len_final, ev_triggered, pa_max);
                }
            } else
            launch_phased_array(sm, n_cand, d_cand, n_ch, ev.L, co.trace, co.trace_offset, st->pa_n_channels,
                                st->d_pa_channel.as<int>(), st->pa_n_beams, st->d_pa_rolls.as<int>(), st->pa_window, st->pa_step,
                                (double)st->pa_divisor, cfg->trigger_threshold, maxL, ev_triggered, pa_max);
            LCHK("phased array");
        }
        if (ev_group)
            hipLaunchKernelGGL(sub_event_trigger_kernel, dim3((unsigned)((n_ev + 255) / 256)), dim3(256), 0, sm, (int)n_ev, ev_group,
                               ev_triggered, triggered);
        MARK(8);
        unsigned long long h_emit[3] = {0ull, 0ull, 0ull};
        if (emit) HIPCHK(hipMemcpyAsync(h_emit, emit_cursor, sizeof h_emit, hipMemcpyDeviceToHost, sm));
        HIPCHK(hipStreamSynchronize(sm));  // host vectors used by async copies above stay alive until here
        if (emit) {
            S.n_emit_overflow = (int64_t)h_emit[1];
            S.n_emitted_samples = (int64_t)std::min<unsigned long long>(h_emit[0], (unsigned long long)co.emit_cap);
            st->ws_bytes["emit_trace"] = (size_t)S.n_emitted_samples * sizeof(double);
            S.n_emitted_events = (int64_t)h_emit[2];
        }
        if (conv_trial >= 0) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, st->evt[7], st->evt[8]) == hipSuccess && ms > 0.f) {
                st->conv_ms_per_event[conv_trial] = (double)ms / (double)n_cand;
                if (st->conv_ms_per_event[0] > 0. && st->conv_ms_per_event[1] > 0.)
                    // (a tie goes to the split form: the half-capacity instantiation moves half the HBM bytes at the same speed --
                    // config 5, PMC passes of both forms: 2.0 against 4.3 TB per step in this kernel)
                    st->conv_mode = (st->conv_ms_per_event[0] <= 1.03 * st->conv_ms_per_event[1]) ? 1 : 2;
            }
        }
    }
    MARK(9);
    if (stats) {
        // count triggers on device-resident mask (cheap D2H of n bytes only when asked for stats)
        std::vector<unsigned char> ht(n_groups);
        HIPCHK(hipMemcpyAsync(ht.data(), triggered, n_groups, hipMemcpyDeviceToHost, sm));
        HIPCHK(hipStreamSynchronize(sm));
        int64_t nt = 0;
        for (unsigned char t : ht) nt += t;
        S.n_triggered = nt;
        if (rt_eval_counter) {
            unsigned long long ne = 0;
            HIPCHK(hipMemcpyAsync(&ne, rt_eval_counter, sizeof ne, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            S.n_objective_evals = (int64_t)ne;
        }
        if (general_counters) {
            unsigned long long gc[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(gc, general_counters, sizeof gc, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            S.n_arz_evals = (int64_t)gc[0];
            S.n_bire_step_bins = (int64_t)gc[1];
        }
        if (eval_counter) {
            unsigned long long ne = 0;
            // (on the station's stream: a copy on the null stream would also wait for the other lanes' streams, array.py)
            HIPCHK(hipMemcpyAsync(&ne, eval_counter, sizeof ne, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            S.n_integrand_evals = (int64_t)ne;
        }
        {
            unsigned long long xc[5] = {0, 0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(xc, xform_count, sizeof xc, hipMemcpyDeviceToHost, sm));
            HIPCHK(hipStreamSynchronize(sm));
            S.n_channel_transforms = (int64_t)xc[0];
            S.n_ray_transforms = (int64_t)xc[1];
            S.n_efield_transforms = (int64_t)xc[2];
            S.n_adc_convolution_flops = (int64_t)xc[3];
            S.n_efield_sampled = (int64_t)xc[4];
        }
        for (int i = 0; i < 8; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, st->evt[i], st->evt[i + 1]) == hipSuccess) S.stage_ms[i] = ms;
        }
        if (n_followers > 0) {   // the quadrature's second launch belongs to the attenuation stage (its evaluations are counted there)
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, st->evt[10], st->evt[11]) == hipSuccess) { S.stage_ms[3] += ms; S.stage_ms[5] -= ms; }
        }
        float tot = 0.f;
        if (hipEventElapsedTime(&tot, st->evt[0], st->evt[9]) == hipSuccess) S.stage_ms[8] = tot;
        *stats = S;
    } else {
        HIPCHK(hipStreamSynchronize(sm));
    }
    if (att_trial >= 0) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, st->evt[3], st->evt[6]) == hipSuccess && ms > 0.f) {
            st->att_ms_per_ray[att_trial] = (double)ms / (double)n_rays;
            if (st->att_ms_per_ray[0] > 0. && st->att_ms_per_ray[1] > 0.)
                st->att_mode = (st->att_ms_per_ray[0] <= st->att_ms_per_ray[1]) ? 1 : 2;
        }
    }
    return 0;
}

int nrhip_readout_windows(nrhip_ctx* ctx, nrhip_station* st, int32_t n_window, int32_t pre_bins, double threshold,
                          int64_t n_items, int32_t* trigger_bin, double* max_amp, double* max_env)
{
    if (!ctx || !st || !trigger_bin || !max_amp || !max_env) return nrhip_fail_msg("nrhip_readout_windows: NULL argument");
    if (st->ctx != ctx) return nrhip_fail_msg("nrhip_readout_windows: station belongs to another context");
    if (n_items == 0) return 0;
    if (n_window < 16 || n_window > FFT_MAX || (n_window & (n_window - 1)))
        return nrhip_fail_msg("nrhip_readout_windows: the read-out window must be a power of two of 16 .. 8192 samples");
    for (const char* k : {"item_event", "trace", "trace_offset", "ev_L"})
        if (st->ws_bytes.find(k) == st->ws_bytes.end())
            return nrhip_fail_msg("nrhip_readout_windows: the last call kept no traces (nrhip_sim_config.dump_traces)");
    const int n_ch = st->dev.n_ch;
    // (the workspace tables above are registered per call -- ws_bytes is cleared when a simulate call starts -- and the count of
    // the call that wrote the traces is kept explicitly: a later call without dump_traces leaves nothing to read out)
    if (st->last_dump_items < 0) return nrhip_fail_msg("nrhip_readout_windows: the last call kept no traces (nrhip_sim_config.dump_traces)");
    if (n_items > st->last_dump_items) return nrhip_fail_msg("nrhip_readout_windows: more items than the last call has");
    if ((int64_t)(st->ws_bytes["item_event"] / sizeof(int)) < n_items ||
        (int64_t)(st->ws_bytes["trace_offset"] / sizeof(long)) < n_items * n_ch + 1)
        return nrhip_fail_msg("nrhip_readout_windows: more items than the last call has");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    int* d_bin;
    double *d_amp, *d_env;
    NEED(d_bin = WS("window_trigger_bin", int, n_items));
    NEED(d_amp = WS("window_max_amp", double, n_items * n_ch));
    NEED(d_env = WS("window_max_env", double, n_items * n_ch));
    launch_readout_windows(sm, (int)n_items, n_ch, st->ws["item_event"].as<int>(), st->ws["ev_L"].as<int>(), st->ws["trace"].as<double>(),
                           st->ws["trace_offset"].as<long>(), n_window, pre_bins, threshold, ctx->twiddle, d_bin, d_amp, d_env);
    LCHK("readout windows");
    HIPCHK(hipMemcpyAsync(trigger_bin, d_bin, sizeof(int) * (size_t)n_items, hipMemcpyDeviceToHost, sm));
    HIPCHK(hipMemcpyAsync(max_amp, d_amp, sizeof(double) * (size_t)n_items * n_ch, hipMemcpyDeviceToHost, sm));
    HIPCHK(hipMemcpyAsync(max_env, d_env, sizeof(double) * (size_t)n_items * n_ch, hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    return 0;
}

int64_t nrhip_sim_fetch(nrhip_station* st, const char* name, void* host_dst, uint64_t bytes)
{
    if (!st || !name) return nrhip_fail_msg("nrhip_sim_fetch: NULL argument");
    if (!st->ctx) return nrhip_fail_msg("nrhip_sim_fetch: the station's context has been destroyed");
    auto it = st->ws_bytes.find(name);
    if (it == st->ws_bytes.end()) return nrhip_fail_msg("nrhip_sim_fetch: no such table in the last simulated batch");
    size_t avail = it->second;
    size_t n = std::min<size_t>(avail, bytes);
    if (n && host_dst) {
        if (hipSetDevice(st->ctx->device) != hipSuccess) return -1;
        hipError_t e = hipMemcpyAsync(host_dst, st->ws[name].p, n, hipMemcpyDeviceToHost, st->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(st->ctx->stream);
        if (e != hipSuccess) return nrhip_fail("hipMemcpy", e);
    }
    return (int64_t)avail;
}

int nrhip_efield_to_voltage(nrhip_ctx* ctx, nrhip_station* st, int32_t n_efields, const double* traces, const double* t0,
                            const double* zenith, const double* azimuth, const int32_t* channel, int32_t apply_filters,
                            int32_t L, double t_min, double* V)
{
    if (!ctx || !st || !V) return nrhip_fail_msg("nrhip_efield_to_voltage: NULL argument");
    if (st->ctx != ctx) return nrhip_fail_msg("nrhip_efield_to_voltage: station belongs to another context");
    const StationDev& sd = st->dev;
    if (n_efields <= 0) return nrhip_fail_msg("station has no efields");  // LookupError in the reference (:117-118)
    // the limits of the batched path (nrhip_simulate_events): any station trace length, common traces of at most 32766 samples
    if (L <= 0 || L % 2 != 0 || L / 2 > NRHIP_SPEC_STRIDE - 1)
        return nrhip_fail_msg("nrhip_efield_to_voltage: common trace length unsupported (odd, or longer than 32766 samples)");
    for (int e = 0; e < n_efields; e++)
        if (channel[e] < 0 || channel[e] >= sd.n_ch) return nrhip_fail_msg("nrhip_efield_to_voltage: bad channel index");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    st->ws_bytes.clear();
    st->last_dump_items = -1;
    double *d_tr, *d_t0, *d_zen, *d_az, *d_V;
    int *d_ch, *d_len;
    NEED(d_tr = WS("ev_traces", double, (size_t)n_efields * 2 * sd.N));
    NEED(d_t0 = WS("ev_t0", double, n_efields));
    NEED(d_zen = WS("ev_zen", double, n_efields));
    NEED(d_az = WS("ev_az", double, n_efields));
    NEED(d_ch = WS("ev_ch", int, n_efields));
    NEED(d_V = WS("ev_V", double, (size_t)sd.n_ch * L));
    NEED(d_len = WS("lengths", int, 1));
    HIPCHK(hipMemcpyAsync(d_tr, traces, sizeof(double) * n_efields * 2 * sd.N, hipMemcpyHostToDevice, sm));
    HIPCHK(hipMemcpyAsync(d_t0, t0, sizeof(double) * n_efields, hipMemcpyHostToDevice, sm));
    HIPCHK(hipMemcpyAsync(d_zen, zenith, sizeof(double) * n_efields, hipMemcpyHostToDevice, sm));
    HIPCHK(hipMemcpyAsync(d_az, azimuth, sizeof(double) * n_efields, hipMemcpyHostToDevice, sm));
    HIPCHK(hipMemcpyAsync(d_ch, channel, sizeof(int) * n_efields, hipMemcpyHostToDevice, sm));
    HIPCHK(hipMemcpyAsync(d_len, &L, sizeof(int), hipMemcpyHostToDevice, sm));
    LengthTables tab;
    NEED(tab.B_fwd = WS("tab_B_fwd", double2, (size_t)FFT_MAX));
    NEED(tab.B_inv = WS("tab_B_inv", double2, (size_t)FFT_MAX));
    NEED(tab.vel = WS("tab_vel", double2, NRHIP_N_ANT_TAB * (size_t)NRHIP_SPEC_STRIDE));
    NEED(tab.E = WS("tab_E", double2, (size_t)NRHIP_E_STRIDE));
    NEED(tab.H = WS("tab_H", double2, sd.n_fsets * (size_t)NRHIP_SPEC_STRIDE));
    NEED(tab.Cf = WS("tab_Cf", double2, (size_t)NRHIP_SPEC_STRIDE));
    NEED(tab.Ci = WS("tab_Ci", double2, (size_t)FFT_MAX));
    NEED(tab.hnorm = WS("tab_hnorm", double, sd.n_fsets * NRHIP_N_ANT_TAB));
    tab.G = nullptr;  // the generic path always goes through the chirp-z kernel
    launch_length_tables(sm, 1, d_len, sd, st->d_filtersets.as<FilterSet>(), ctx->twiddle, ctx->w16, tab);
    LCHK("length_tables");
    double2* scratch;
    NEED(scratch = WS("channel_scratch", double2, (size_t)std::max(sd.n_ch, channel_grid_blocks()) * 2 * NRHIP_SPEC_STRIDE));
    double2* tab_nodes = nullptr;
    if (sd.ant_tabs) NEED(tab_nodes = WS("antenna_table_nodes", double2, (size_t)sd.n_ch * 2 * sd.max_tab_freq));
    launch_efield_channel(sm, n_efields, d_tr, d_t0, d_zen, d_az, d_ch, sd, L, t_min, apply_filters, ctx->twiddle, tab,
                          scratch, d_V, tab_nodes);
    LCHK("efield_channel");
    HIPCHK(hipMemcpyAsync(V, d_V, sizeof(double) * sd.n_ch * L, hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    return 0;
}

int nrhip_askaryan_spectrum_batch(nrhip_ctx* ctx, int64_t n, const double* energy, const double* theta,
                                  const int32_t* shower_type, const double* n_index, const double* R, const double* k_L,
                                  int32_t model, int32_t N, double dt, double* spectrum)
{
    if (!ctx) return nrhip_fail_msg("nrhip_askaryan_spectrum_batch: ctx is NULL");
    if (model < 0 || model > 2) return nrhip_fail_msg("model unknown");  // NotImplementedError in askaryan.py:136
    if (N <= 0 || N % 2) return nrhip_fail_msg("nrhip_askaryan_spectrum_batch: N must be even");
    if (n <= 0) return 0;
    for (int64_t i = 0; i < n; i++)
        if (shower_type[i] != NRHIP_SHOWER_HAD && shower_type[i] != NRHIP_SHOWER_EM)
            return nrhip_fail_msg("shower type is not implemented");  // parametrizations.py:130
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nf = N / 2 + 1;
    DevArray dE, dth, dty, dn, dR, dk, ds;
    int rc = upload(ctx, dE, energy, n) || upload(ctx, dth, theta, n) || upload(ctx, dty, shower_type, n) ||
             upload(ctx, dn, n_index, n) || upload(ctx, dR, R, n) || upload(ctx, dk, k_L, n);
    if (!rc && ds.reserve(n * nf * 16) != hipSuccess) rc = nrhip_fail_msg("out of device memory");
    if (!rc) {
        launch_askaryan_spectrum(ctx->stream, (int)n, dE.as<double>(), dth.as<double>(), dty.as<int>(), dn.as<double>(),
                                 dR.as<double>(), dk.as<double>(), model, N, dt, ds.as<double2>());
        hipError_t e = hipMemcpyAsync(spectrum, ds.p, n * nf * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = nrhip_fail("askaryan copy", e);
    }
    dE.release(); dth.release(); dty.release(); dn.release(); dR.release(); dk.release(); ds.release();
    return rc;
}

int nrhip_debug_czt(nrhip_ctx* ctx, int32_t n_batch, int32_t n_in, int32_t n_out, int32_t Q, double sgn, const double* in,
                    double* out)
{
    if (!ctx) return nrhip_fail_msg("nrhip_debug_czt: ctx is NULL");
    if (n_in + n_out - 1 > FFT_MAX) return nrhip_fail_msg("nrhip_debug_czt: n_in + n_out - 1 exceeds 8192");
    HIPCHK(hipSetDevice(ctx->device));
    if (ensure_twiddle(ctx)) return -1;
    int grid = std::min(n_batch, 64);
    DevArray di, dout, dB;
    int rc = upload(ctx, di, (const double2*)in, (size_t)n_batch * n_in);
    if (!rc && (dout.reserve((size_t)n_batch * n_out * 16) != hipSuccess || dB.reserve((size_t)grid * FFT_MAX * 16) != hipSuccess))
        rc = nrhip_fail_msg("out of device memory");
    if (!rc) {
        launch_czt_test(ctx->stream, n_batch, n_in, n_out, Q, sgn, di.as<double2>(), dout.as<double2>(), ctx->twiddle,
                        dB.as<double2>(), grid);
        hipError_t e = hipMemcpyAsync(out, dout.p, (size_t)n_batch * n_out * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = nrhip_fail("czt", e);
    }
    di.release(); dout.release(); dB.release();
    return rc;
}

int nrhip_debug_wave_sums(nrhip_ctx* ctx, int32_t n_waves, const double* in, double* out)
{
    if (!ctx) return nrhip_fail_msg("nrhip_debug_wave_sums: ctx is NULL");
    if (n_waves <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    DevArray di, dout;
    int rc = upload(ctx, di, in, (size_t)n_waves * 8 * 64);
    if (!rc && dout.reserve((size_t)n_waves * WAVE_TEST_OUT * 8) != hipSuccess) rc = nrhip_fail_msg("out of device memory");
    if (!rc) {
        launch_wave_reduce_test(ctx->stream, n_waves, di.as<double>(), dout.as<double>());
        hipError_t e = hipMemcpyAsync(out, dout.p, (size_t)n_waves * WAVE_TEST_OUT * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = nrhip_fail("wave sums", e);
    }
    di.release(); dout.release();
    return rc;
}

}  // extern "C"

#ifdef NRHIP_CONV_TIMING
namespace nrhip {
void launch_conv_pair_probe(hipStream_t s, const double2* tw, const double2* w16, const double2* G, int n_iter, int L, int variant,
                            unsigned long long* clk);
}
// tools/conv_pair_probe.py: n_iter transform pairs per block on every CU; clocks[6] = (forward, spectrum pass, inverse) of wave 0 and
// of wave 5 summed over the blocks, *ms = wall time of the launch
extern "C" int nrhip_debug_conv_pair(nrhip_ctx* ctx, int n_iter, int L, int variant, unsigned long long* clocks6, float* ms)
{
    if (ensure_twiddle(ctx)) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    double2* G;
    unsigned long long* clk;
    const size_t ng = (size_t)64 * NRHIP_G_STRIDE;
    HIPCHK(hipMalloc((void**)&G, ng * sizeof(double2)));
    HIPCHK(hipMalloc((void**)&clk, 6 * sizeof(unsigned long long)));
    std::vector<double2> h(ng, make_double2(1.0 / 8192, 0.));
    HIPCHK(hipMemcpy(G, h.data(), ng * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(clk, 0, 6 * sizeof(unsigned long long)));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    nrhip::launch_conv_pair_probe(ctx->stream, ctx->twiddle, ctx->w16, G, 2, L, variant, clk);   // warm-up
    HIPCHK(hipMemsetAsync(clk, 0, 6 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipEventRecord(a, ctx->stream));
    nrhip::launch_conv_pair_probe(ctx->stream, ctx->twiddle, ctx->w16, G, n_iter, L, variant, clk);
    HIPCHK(hipEventRecord(b, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(ms, a, b));
    HIPCHK(hipMemcpy(clocks6, clk, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(G); (void)hipFree(clk); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}
#endif
