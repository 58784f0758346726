// api.hip -- C-ABI layer of libnrhip.so (see include/nrhip.h for the contract).
#include "../../include/nrhip.h"
#include "nrhip_internal.h"
#include "ctx.h"
#include "arz.h"
#include "birefringence.h"
#include "earth.h"
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace nrhip {
void launch_ray_limits(hipStream_t stream, long n_rays, const double* x1, const double* x2, const double* C0,
                       const IceConst& m, double* zint);
void launch_attenuation_items(hipStream_t stream, long n_rays, const double* C0, const double* zint, int n_freq,
                              const double* freqs, int model, const IceConst& m, double* att, int* neval,
                              const int* ray_index, unsigned long long* eval_counter, const double* gl3, int gl3_n,
                              int* overflow);   // overflow: 1 + 2 n_rays ints (NULL: the general kernel only)
void launch_attenuation_length(hipStream_t stream, long n, const double* z, const double* f, int model, double* L,
                               const double* gl3, int gl3_n);
}  // namespace nrhip

static thread_local char g_err[512] = "";

int nrhip_fail(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -1;
}
int nrhip_fail_msg(const char* what)
{
    snprintf(g_err, sizeof g_err, "%s", what);
    return -2;
}
static int fail_msg(const char* what) { return nrhip_fail_msg(what); }

// RAII device buffer for the host-pointer convenience entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() { return (T*)p; }
};

extern "C" {

const char* nrhip_last_error(void) { return g_err; }

int nrhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int nrhip_ctx_create(int device, double n_ice, double delta_n, double z_0, int attenuation_model, nrhip_ctx** out)
{
    if (!out) return fail_msg("nrhip_ctx_create: out is NULL");
    if (!(n_ice > 1.) || !(delta_n > 0.) || !(z_0 > 0.))
        return fail_msg("nrhip_ctx_create: the analytic ray tracer needs an exponential (non-uniform) ice model");
    if (attenuation_model < 1 || attenuation_model > 5)
        return fail_msg("nrhip_ctx_create: attenuation model not implemented (SP1=1, GL1=2, MB1=3, GL2=4, GL3=5)");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail_msg("nrhip_ctx_create: no such GPU");
    HIPCHK(hipSetDevice(device));
    nrhip_ctx* c = new nrhip_ctx();
    c->device = device;
    c->ice = nrhip::make_ice(n_ice, delta_n, z_0);
    c->att_model = attenuation_model;
    hipError_t e = hipStreamCreate(&c->stream);
    if (e != hipSuccess) {
        delete c;
        return nrhip_fail("hipStreamCreate", e);
    }
    *out = c;
    return 0;
}

void nrhip_ctx_destroy(nrhip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    while (!ctx->stations.empty()) nrhip_station_detach(*ctx->stations.begin());  // stations outliving their context
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->gl3) (void)hipFree(ctx->gl3);
    if (ctx->twiddle) (void)hipFree(ctx->twiddle);
    if (ctx->w16) (void)hipFree(ctx->w16);
    ctx->cull_ws.release();
    delete ctx;
}

int nrhip_ctx_set_ray_finder(nrhip_ctx* ctx, int32_t finder)
{
    if (!ctx) return fail_msg("nrhip_ctx_set_ray_finder: NULL context");
    if (finder != NRHIP_FINDER_TRUE_ROOTS && finder != NRHIP_FINDER_REFERENCE) return fail_msg("nrhip_ctx_set_ray_finder: unknown finder");
    if (finder != ctx->ray_finder)
        for (auto* s : ctx->stations) s->generation++;   // ray tables kept for reuse_ray_tables were made by the other finder
    ctx->ray_finder = finder;
    return 0;
}

int nrhip_ctx_set_gl3_table(nrhip_ctx* ctx, int32_t n, const double* depth, const double* slope, const double* offset)
{
    if (!ctx || !depth || !slope || !offset) return fail_msg("nrhip_ctx_set_gl3_table: NULL argument");
    if (n < 2 || n > 100000) return fail_msg("nrhip_ctx_set_gl3_table: bad table length");
    for (int i = 1; i < n; i++)
        if (!(depth[i] > depth[i - 1])) return fail_msg("nrhip_ctx_set_gl3_table: depths must increase");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->gl3) (void)hipFree(ctx->gl3);
    ctx->gl3 = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->gl3, sizeof(double) * 3 * n));
    HIPCHK(hipMemcpy(ctx->gl3, depth, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->gl3 + n, slope, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->gl3 + 2 * n, offset, sizeof(double) * n, hipMemcpyHostToDevice));
    ctx->gl3_n = n;
    return 0;
}

int nrhip_synchronize(nrhip_ctx* ctx)
{
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nrhip_malloc(nrhip_ctx* ctx, uint64_t bytes, void** dev_ptr)
{
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMalloc(dev_ptr, bytes ? bytes : 8));
    return 0;
}
int nrhip_free(nrhip_ctx* ctx, void* dev_ptr)
{
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipFree(dev_ptr));
    return 0;
}
int nrhip_memset(nrhip_ctx* ctx, void* dev_dst, int32_t value, uint64_t bytes)
{
    if (!ctx) return fail_msg("nrhip_memset: ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes) HIPCHK(hipMemsetAsync(dev_dst, value, bytes, ctx->stream));
    return 0;
}
int nrhip_memcpy_h2d(nrhip_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes)
{
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int nrhip_memcpy_d2h(nrhip_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes)
{
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int ray_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2, const double* C0_in,
                     int32_t* n_sol, int32_t* type, double* C0, double* C1, double* D, double* T, double* launch,
                     double* receive, double* refl_angle);

int nrhip_find_solutions_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                               int32_t* n_sol, int32_t* type, double* C0, double* C1, double* D, double* T,
                               double* launch, double* receive, double* refl_angle)
{
    return ray_batch(ctx, n_pairs, x1, x2, n_x2, nullptr, n_sol, type, C0, C1, D, T, launch, receive, refl_angle);
}

int nrhip_ray_records_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                            const double* C0_in, int32_t* n_sol, int32_t* type, double* C0, double* C1, double* D,
                            double* T, double* launch, double* receive, double* refl_angle)
{
    if (!C0_in) return fail_msg("nrhip_ray_records_batch: C0_in is NULL");
    return ray_batch(ctx, n_pairs, x1, x2, n_x2, C0_in, n_sol, type, C0, C1, D, T, launch, receive, refl_angle);
}

static int ray_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2, const double* C0_in,
                     int32_t* n_sol, int32_t* type, double* C0, double* C1, double* D, double* T, double* launch,
                     double* receive, double* refl_angle)
{
    if (!ctx) return fail_msg("nrhip_find_solutions_batch: ctx is NULL");
    if (n_pairs < 0 || n_x2 < 0) return fail_msg("nrhip_find_solutions_batch: negative size");
    if (n_pairs == 0) return 0;
    if (n_x2 > 0 && n_pairs % n_x2 != 0) return fail_msg("nrhip_find_solutions_batch: n_pairs not a multiple of n_x2");
    HIPCHK(hipSetDevice(ctx->device));
    const int S = NRHIP_MAXS;
    size_t n1 = (n_x2 > 0) ? n_pairs / n_x2 : n_pairs, n2 = (n_x2 > 0) ? n_x2 : n_pairs;
    DevBuf dx1, dx2, dns, dty, dC0, dC1, dD, dT, dla, dre, dra;
    HIPCHK(dx1.alloc(n1 * 24));
    HIPCHK(dx2.alloc(n2 * 24));
    HIPCHK(dns.alloc(n_pairs * 4));
    HIPCHK(dty.alloc(n_pairs * S * 4));
    HIPCHK(dC0.alloc(n_pairs * S * 8));
    HIPCHK(dC1.alloc(n_pairs * S * 8));
    HIPCHK(dD.alloc(n_pairs * S * 8));
    HIPCHK(dT.alloc(n_pairs * S * 8));
    HIPCHK(dla.alloc(n_pairs * S * 24));
    HIPCHK(dre.alloc(n_pairs * S * 24));
    HIPCHK(dra.alloc(n_pairs * S * 8));
    HIPCHK(hipMemcpyAsync(dx1.p, x1, n1 * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dx2.p, x2, n2 * 24, hipMemcpyHostToDevice, ctx->stream));
    nrhip::RayRecords r{dns.as<int>(), dty.as<int>(), dC0.as<double>(), dC1.as<double>(), dD.as<double>(),
                        dT.as<double>(), dla.as<double>(), dre.as<double>(), dra.as<double>()};
    DevBuf dgiven;
    if (C0_in) {
        HIPCHK(dgiven.alloc(n_pairs * S * 8));
        HIPCHK(hipMemcpyAsync(dgiven.p, C0_in, n_pairs * S * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    nrhip::launch_raytrace(ctx->stream, n_pairs, dx1.as<double>(), dx2.as<double>(), n_x2, ctx->ice, r, nullptr, nullptr,
                           C0_in ? dgiven.as<double>() : nullptr, nullptr, nullptr, nullptr, ctx->ray_finder == NRHIP_FINDER_REFERENCE);
    HIPCHK(hipGetLastError());
#define D2H(dst, src, bytes) if (dst) HIPCHK(hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, ctx->stream))
    D2H(n_sol, dns, n_pairs * 4);
    D2H(type, dty, n_pairs * S * 4);
    D2H(C0, dC0, n_pairs * S * 8);
    D2H(C1, dC1, n_pairs * S * 8);
    D2H(D, dD, n_pairs * S * 8);
    D2H(T, dT, n_pairs * S * 8);
    D2H(launch, dla, n_pairs * S * 24);
    D2H(receive, dre, n_pairs * S * 24);
    D2H(refl_angle, dra, n_pairs * S * 8);
#undef D2H
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nrhip_attenuation_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2, const double* C0,
                            int32_t n_freq, const double* freqs, double* att, int32_t* neval)
{
    if (!ctx) return fail_msg("nrhip_attenuation_batch: ctx is NULL");
    if (n_rays < 0 || n_freq < 0) return fail_msg("nrhip_attenuation_batch: negative size");
    if (n_rays == 0 || n_freq == 0) return 0;
    for (int i = 0; i < n_freq; i++)
        if (!(freqs[i] > 0)) return fail_msg("nrhip_attenuation_batch: frequencies must be > 0 (DC is 1 by definition)");
    if (ctx->att_model == NRHIP_ATT_GL3 && !ctx->gl3) return fail_msg("nrhip_attenuation_batch: GL3 needs nrhip_ctx_set_gl3_table");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf dx1, dx2, dC0, dz, df, da, dn, dovf;
    HIPCHK(dovf.alloc((2 * (size_t)n_rays + 1) * 4));
    HIPCHK(hipMemsetAsync(dovf.p, 0, sizeof(int), ctx->stream));
    HIPCHK(dx1.alloc(n_rays * 24));
    HIPCHK(dx2.alloc(n_rays * 24));
    HIPCHK(dC0.alloc(n_rays * 8));
    HIPCHK(dz.alloc(n_rays * 24));
    HIPCHK(df.alloc(n_freq * 8));
    HIPCHK(da.alloc(n_rays * n_freq * 8));
    HIPCHK(dn.alloc(n_rays * n_freq * 4));
    HIPCHK(hipMemcpyAsync(dx1.p, x1, n_rays * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dx2.p, x2, n_rays * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dC0.p, C0, n_rays * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(df.p, freqs, n_freq * 8, hipMemcpyHostToDevice, ctx->stream));
    nrhip::launch_ray_limits(ctx->stream, n_rays, dx1.as<double>(), dx2.as<double>(), dC0.as<double>(), ctx->ice,
                             dz.as<double>());
    nrhip::launch_attenuation_items(ctx->stream, n_rays, dC0.as<double>(), dz.as<double>(), n_freq, df.as<double>(),
                                    ctx->att_model, ctx->ice, da.as<double>(), dn.as<int>(), nullptr, nullptr, ctx->gl3,
                                    ctx->gl3_n, dovf.as<int>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(att, da.p, n_rays * n_freq * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (neval) HIPCHK(hipMemcpyAsync(neval, dn.p, n_rays * n_freq * 4, hipMemcpyDeviceToHost, ctx->stream));
    int h_ovf = 0;
    HIPCHK(hipMemcpyAsync(&h_ovf, dovf.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->last_att_overflow = h_ovf;
    return 0;
}

int64_t nrhip_attenuation_last_overflow(nrhip_ctx* ctx) { return ctx ? ctx->last_att_overflow : -1; }

// ---- reflections off the bottom of an ice shelf ----
static int refl_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2, int32_t n_reflections,
                      double z_reflection, int given, int32_t* n_sol, int32_t* type, double* C0, double* C1,
                      int32_t* reflection, int32_t* reflection_case, double* D, double* T, double* launch, double* receive,
                      double* refl_angle, int32_t* n_segments, int32_t* surface_mask)
{
    const char* who = given ? "nrhip_ray_records_reflections_batch" : "nrhip_find_solutions_reflections_batch";
    if (!ctx) return fail_msg((std::string(who) + ": ctx is NULL").c_str());
    if (n_pairs < 0 || n_x2 < 0) return fail_msg((std::string(who) + ": negative size").c_str());
    if (n_reflections < 0 || n_reflections > NRHIP_MAX_REFLECTIONS)
        return fail_msg((std::string(who) + ": n_reflections must be 0..4").c_str());
    if (n_reflections > 0 && !(z_reflection < 0))  // AttributeError in the reference (:1421-1423)
        return fail_msg((std::string(who) + ": reflections off the bottom are requested, but the ice model does not specify a reflective layer").c_str());
    if (n_pairs == 0) return 0;
    if (n_x2 > 0 && n_pairs % n_x2 != 0) return fail_msg((std::string(who) + ": n_pairs not a multiple of n_x2").c_str());
    if (given && (!n_sol || !C0 || !reflection || !reflection_case)) return fail_msg((std::string(who) + ": records missing").c_str());
    HIPCHK(hipSetDevice(ctx->device));
    const int S = 2 + 4 * n_reflections, n_calls = 1 + 2 * n_reflections, NS = n_reflections + 1;
    size_t n1 = (n_x2 > 0) ? n_pairs / n_x2 : n_pairs, n2 = (n_x2 > 0) ? n_x2 : n_pairs;
    const size_t nk = (size_t)n_pairs * S;
    DevBuf dx1, dx2, dcn, dcc, dns, dty, drf, drc, dsu, dsm, dC0, dC1, dD, dT, dla, dre, dra, dsz, dsc;
    HIPCHK(dx1.alloc(n1 * 24));
    HIPCHK(dx2.alloc(n2 * 24));
    HIPCHK(dcn.alloc((size_t)n_pairs * n_calls * 4));
    HIPCHK(dcc.alloc((size_t)n_pairs * n_calls * 24));
    HIPCHK(dns.alloc(n_pairs * 4));
    HIPCHK(dty.alloc(nk * 4));
    HIPCHK(drf.alloc(nk * 4));
    HIPCHK(drc.alloc(nk * 4));
    HIPCHK(dsu.alloc(nk * 4));
    HIPCHK(dsm.alloc(nk * 4));
    HIPCHK(dC0.alloc(nk * 8));
    HIPCHK(dC1.alloc(nk * 8));
    HIPCHK(dD.alloc(nk * 8));
    HIPCHK(dT.alloc(nk * 8));
    HIPCHK(dla.alloc(nk * 24));
    HIPCHK(dre.alloc(nk * 24));
    HIPCHK(dra.alloc(nk * 8));
    HIPCHK(dsz.alloc(nk * NS * 24));
    HIPCHK(dsc.alloc(nk * NS * 8));
    HIPCHK(hipMemcpyAsync(dx1.p, x1, n1 * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dx2.p, x2, n2 * 24, hipMemcpyHostToDevice, ctx->stream));
    nrhip::ReflRecords r{dns.as<int>(), dty.as<int>(), drf.as<int>(), drc.as<int>(), dsu.as<int>(), dsm.as<int>(), dC0.as<double>(),
                         dC1.as<double>(), dD.as<double>(), dT.as<double>(), dla.as<double>(), dre.as<double>(),
                         dra.as<double>(), dsz.as<double>(), dsc.as<double>()};
    if (given) {
        HIPCHK(hipMemcpyAsync(dns.p, n_sol, n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(dC0.p, C0, nk * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(drf.p, reflection, nk * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(drc.p, reflection_case, nk * 4, hipMemcpyHostToDevice, ctx->stream));
    } else {
        nrhip::launch_find_refl(ctx->stream, n_pairs, n_reflections, dx1.as<double>(), dx2.as<double>(), n_x2, ctx->ice,
                                z_reflection, dcn.as<int>(), dcc.as<double>(), ctx->ray_finder == NRHIP_FINDER_REFERENCE);
    }
    nrhip::launch_records_refl(ctx->stream, n_pairs, n_reflections, S, dx1.as<double>(), dx2.as<double>(), n_x2, ctx->ice,
                               z_reflection, dcn.as<int>(), dcc.as<double>(), given, r);
    HIPCHK(hipGetLastError());
#define D2H(dst, src, bytes) if (dst) HIPCHK(hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, ctx->stream))
    D2H(n_sol, dns, n_pairs * 4);
    D2H(type, dty, nk * 4);
    D2H(reflection, drf, nk * 4);
    D2H(reflection_case, drc, nk * 4);
    D2H(n_segments, dsu, nk * 4);
    D2H(surface_mask, dsm, nk * 4);
    D2H(C0, dC0, nk * 8);
    D2H(C1, dC1, nk * 8);
    D2H(D, dD, nk * 8);
    D2H(T, dT, nk * 8);
    D2H(launch, dla, nk * 24);
    D2H(receive, dre, nk * 24);
    D2H(refl_angle, dra, nk * 8);
#undef D2H
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nrhip_find_solutions_reflections_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                                           int32_t n_reflections, double z_reflection, int32_t* n_sol, int32_t* type,
                                           double* C0, double* C1, int32_t* reflection, int32_t* reflection_case, double* D,
                                           double* T, double* launch, double* receive, double* refl_angle,
                                           int32_t* n_segments, int32_t* surface_mask)
{
    return refl_batch(ctx, n_pairs, x1, x2, n_x2, n_reflections, z_reflection, 0, n_sol, type, C0, C1, reflection,
                      reflection_case, D, T, launch, receive, refl_angle, n_segments, surface_mask);
}

int nrhip_ray_records_reflections_batch(nrhip_ctx* ctx, int64_t n_pairs, const double* x1, const double* x2, int32_t n_x2,
                                        int32_t n_reflections, double z_reflection, int32_t* n_sol, int32_t* type,
                                        double* C0, double* C1, int32_t* reflection, int32_t* reflection_case, double* D,
                                        double* T, double* launch, double* receive, double* refl_angle,
                                        int32_t* n_segments, int32_t* surface_mask)
{
    return refl_batch(ctx, n_pairs, x1, x2, n_x2, n_reflections, z_reflection, 1, n_sol, type, C0, C1, reflection,
                      reflection_case, D, T, launch, receive, refl_angle, n_segments, surface_mask);
}

int nrhip_attenuation_reflections_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2, const double* C0,
                                        const int32_t* reflection, const int32_t* reflection_case, double z_reflection,
                                        int32_t n_freq, const double* freqs, double* att, double* segment_att)
{
    if (!ctx || !x1 || !x2 || !C0 || !reflection || !reflection_case || !freqs || !att)
        return fail_msg("nrhip_attenuation_reflections_batch: NULL argument");
    if (n_rays < 0 || n_freq < 0) return fail_msg("nrhip_attenuation_reflections_batch: negative size");
    if (n_rays == 0 || n_freq == 0) return 0;
    for (int i = 0; i < n_freq; i++)
        if (!(freqs[i] > 0)) return fail_msg("nrhip_attenuation_reflections_batch: frequencies must be > 0 (DC is 1 by definition)");
    if (ctx->att_model == NRHIP_ATT_GL3 && !ctx->gl3)
        return fail_msg("nrhip_attenuation_reflections_batch: GL3 needs nrhip_ctx_set_gl3_table");
    int max_refl = 0;
    std::vector<int32_t> one(n_rays);
    for (int64_t i = 0; i < n_rays; i++) {
        if (reflection[i] < 0 || reflection[i] > NRHIP_MAX_REFLECTIONS)
            return fail_msg("nrhip_attenuation_reflections_batch: reflection must be 0..4");
        max_refl = std::max(max_refl, (int)reflection[i]);
        one[i] = std::isnan(C0[i]) ? 0 : 1;
    }
    if (max_refl > 0 && !(z_reflection < 0))
        return fail_msg("nrhip_attenuation_reflections_batch: reflections off the bottom are requested, but the ice model does not specify a reflective layer");
    HIPCHK(hipSetDevice(ctx->device));
    const int NS = max_refl + 1;
    const size_t nk = (size_t)n_rays;  // one solution slot per ray
    DevBuf dx1, dx2, dns, dty, drf, drc, dsu, dsm, dC0, dC1, dD, dT, dla, dre, dra, dsz, dsc, df, dsa, da, dovf;
    HIPCHK(dovf.alloc((2 * (size_t)nk * NS + 1) * 4));
    HIPCHK(dx1.alloc(nk * 24));
    HIPCHK(dx2.alloc(nk * 24));
    HIPCHK(dns.alloc(nk * 4));
    HIPCHK(dty.alloc(nk * 4));
    HIPCHK(drf.alloc(nk * 4));
    HIPCHK(drc.alloc(nk * 4));
    HIPCHK(dsu.alloc(nk * 4));
    HIPCHK(dsm.alloc(nk * 4));
    HIPCHK(dC0.alloc(nk * 8));
    HIPCHK(dC1.alloc(nk * 8));
    HIPCHK(dD.alloc(nk * 8));
    HIPCHK(dT.alloc(nk * 8));
    HIPCHK(dla.alloc(nk * 24));
    HIPCHK(dre.alloc(nk * 24));
    HIPCHK(dra.alloc(nk * 8));
    HIPCHK(dsz.alloc(nk * NS * 24));
    HIPCHK(dsc.alloc(nk * NS * 8));
    HIPCHK(df.alloc((size_t)n_freq * 8));
    HIPCHK(dsa.alloc(nk * NS * n_freq * 8));
    HIPCHK(da.alloc(nk * n_freq * 8));
    HIPCHK(hipMemcpyAsync(dx1.p, x1, nk * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dx2.p, x2, nk * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dns.p, one.data(), nk * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dC0.p, C0, nk * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(drf.p, reflection, nk * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(drc.p, reflection_case, nk * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(df.p, freqs, (size_t)n_freq * 8, hipMemcpyHostToDevice, ctx->stream));
    nrhip::ReflRecords r{dns.as<int>(), dty.as<int>(), drf.as<int>(), drc.as<int>(), dsu.as<int>(), dsm.as<int>(), dC0.as<double>(),
                         dC1.as<double>(), dD.as<double>(), dT.as<double>(), dla.as<double>(), dre.as<double>(),
                         dra.as<double>(), dsz.as<double>(), dsc.as<double>()};
    nrhip::launch_records_refl(ctx->stream, n_rays, max_refl, 1, dx1.as<double>(), dx2.as<double>(), 0, ctx->ice, z_reflection,
                               nullptr, nullptr, 1, r);
    nrhip::launch_attenuation_items(ctx->stream, (long)nk * NS, dsc.as<double>(), dsz.as<double>(), n_freq, df.as<double>(),
                                    ctx->att_model, ctx->ice, dsa.as<double>(), nullptr, nullptr, nullptr, ctx->gl3, ctx->gl3_n,
                                    dovf.as<int>());
    nrhip::launch_segment_product(ctx->stream, n_rays, NS, n_freq, dsz.as<double>(), dsa.as<double>(), da.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(att, da.p, nk * n_freq * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (segment_att) HIPCHK(hipMemcpyAsync(segment_att, dsa.p, nk * NS * n_freq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nrhip_arz_time_trace_batch(nrhip_ctx* ctx, int64_t n_rays, const double* energy, const double* theta, const double* distance,
                               const int32_t* shower_type, const double* em_factor, const int32_t* profile_index,
                               const double* rescale, int32_t n_profiles, int32_t n_depth, const double* profile_depth,
                               const double* profile_ce, const double* parameters, int32_t N, double dt, double n_index,
                               double interp_factor2, int32_t shift_for_xmax, double maximum_angle, double* trace,
                               double* vector_potential)
{
    if (!ctx || !energy || !theta || !distance || !shower_type || !em_factor || !profile_index || !profile_depth || !profile_ce ||
        !parameters || !trace)
        return fail_msg("nrhip_arz_time_trace_batch: NULL argument");
    if (n_rays < 0) return fail_msg("nrhip_arz_time_trace_batch: negative size");
    if (n_rays == 0) return 0;
    if (N < 2 || N % 2 != 0 || !(dt > 0)) return fail_msg("nrhip_arz_time_trace_batch: N must be even and dt > 0");
    if (n_depth < 2 || n_depth > 2048 || n_profiles < 1)
        return fail_msg("nrhip_arz_time_trace_batch: charge-excess profiles need 2..2048 depth bins");
    for (int64_t i = 0; i < n_rays; i++) {
        if (shower_type[i] != 0 && shower_type[i] != 1)  // NotImplementedError in the reference (ARZ.py:635-641)
            return fail_msg("nrhip_arz_time_trace_batch: showers of this type are not implemented. Use 'HAD', 'EM'");
        if (profile_index[i] < 0 || profile_index[i] >= n_profiles) return fail_msg("nrhip_arz_time_trace_batch: bad profile index");
    }
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf dE, dth, dR, dty, dem, dpi, drs, dpd, dpc, dpar, dvp, dtr, dst, dtab;
    const size_t nr = (size_t)n_rays;
    HIPCHK(dtab.alloc((size_t)ARZ_TABLE_DOUBLES * 8));
    HIPCHK(dE.alloc(nr * 8)); HIPCHK(dth.alloc(nr * 8)); HIPCHK(dR.alloc(nr * 8)); HIPCHK(dty.alloc(nr * 4));
    HIPCHK(dem.alloc(nr * 8)); HIPCHK(dpi.alloc(nr * 4)); HIPCHK(drs.alloc(nr * 8));
    HIPCHK(dpd.alloc((size_t)n_depth * 8)); HIPCHK(dpc.alloc((size_t)n_profiles * n_depth * 8)); HIPCHK(dpar.alloc(14 * 8));
    HIPCHK(dvp.alloc(nr * (N + 1) * 2 * 8)); HIPCHK(dtr.alloc(nr * 3 * N * 8)); HIPCHK(dst.alloc(nr * 4));
    hipStream_t s = ctx->stream;
#define H2D(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, s))
    H2D(dE, energy, nr * 8); H2D(dth, theta, nr * 8); H2D(dR, distance, nr * 8); H2D(dty, shower_type, nr * 4);
    H2D(dem, em_factor, nr * 8); H2D(dpi, profile_index, nr * 4);
    if (rescale) H2D(drs, rescale, nr * 8);
    H2D(dpd, profile_depth, (size_t)n_depth * 8); H2D(dpc, profile_ce, (size_t)n_profiles * n_depth * 8);
    H2D(dpar, parameters, 14 * 8);
#undef H2D
    HIPCHK(hipMemsetAsync(dst.p, 0, nr * 4, s));
    nrhip::ArzBatch b{(long)n_rays, dE.as<double>(), dth.as<double>(), dR.as<double>(), dty.as<int>(), dem.as<double>(),
                      dpi.as<int>(), rescale ? drs.as<double>() : nullptr, n_profiles, n_depth, dpd.as<double>(),
                      dpc.as<double>(), dpar.as<double>(), N, dt, n_index, interp_factor2, shift_for_xmax, maximum_angle};
    b.form_factor_table = getenv("NRHIP_ARZ_DIRECT") ? nullptr : dtab.as<double>();
    nrhip::launch_arz(s, b, dvp.as<double>(), dtr.as<double>(), dst.as<int>());
    HIPCHK(hipGetLastError());
    std::vector<int> st(nr);
    HIPCHK(hipMemcpyAsync(st.data(), dst.p, nr * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(trace, dtr.p, nr * 3 * N * 8, hipMemcpyDeviceToHost, s));
    if (vector_potential) HIPCHK(hipMemcpyAsync(vector_potential, dvp.p, nr * (N + 1) * 2 * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < nr; i++)
        if (st[i]) return fail_msg("nrhip_arz_time_trace_batch: length of indices is not 2 nor 4 (more than two stretches of the profile radiate within 1 ns)");
    return 0;
}

int nrhip_birefringence_batch(nrhip_ctx* ctx, int64_t n_rays, const double* x1, const double* x2, const double* C0,
                              const double* path_length, const int32_t n_knots[3], const double* knots, const double* coeffs,
                              double n_ref, double angle_to_iceflow, int32_t n_f, double sampling_rate, double* spectra,
                              double* step_records)
{
    if (!ctx || !x1 || !x2 || !C0 || !path_length || !n_knots || !knots || !coeffs || !spectra)
        return fail_msg("nrhip_birefringence_batch: NULL argument");
    if (n_rays < 0) return fail_msg("nrhip_birefringence_batch: negative size");
    if (n_rays == 0) return 0;
    if (n_f < 2 || !(sampling_rate > 0)) return fail_msg("nrhip_birefringence_batch: n_f >= 2 and sampling_rate > 0 required");
    for (int j = 0; j < 3; j++)
        if (n_knots[j] < 8) return fail_msg("nrhip_birefringence_batch: a cubic spline needs at least 8 knots");
    std::vector<int> npts(n_rays);
    std::vector<long> off(n_rays + 1, 0);
    int max_points = 0;
    for (int64_t i = 0; i < n_rays; i++) {
        if (!(path_length[i] >= 0) || !(C0[i] > 0)) return fail_msg("nrhip_birefringence_batch: rays need C0 > 0 and a path length");
        npts[i] = (int)(path_length[i] / 1.);  // acc = int(D / units.m) (:2417)
        off[i + 1] = off[i] + std::max(npts[i] - 1, 0);
        max_points = std::max(max_points, npts[i]);
    }
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t nr = (size_t)n_rays, nk = (size_t)n_knots[0] + n_knots[1] + n_knots[2];
    DevBuf dx1, dx2, dC0, dnp, dof, dkn, dco, dst, dsp;
    HIPCHK(dx1.alloc(nr * 24)); HIPCHK(dx2.alloc(nr * 24)); HIPCHK(dC0.alloc(nr * 8)); HIPCHK(dnp.alloc(nr * 4));
    HIPCHK(dof.alloc(nr * 8)); HIPCHK(dkn.alloc(nk * 8)); HIPCHK(dco.alloc(nk * 8));
    HIPCHK(dst.alloc((size_t)std::max<long>(off[n_rays], 1) * 40)); HIPCHK(dsp.alloc(nr * 2 * n_f * 16));
#define H2D(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, s))
    H2D(dx1, x1, nr * 24); H2D(dx2, x2, nr * 24); H2D(dC0, C0, nr * 8); H2D(dnp, npts.data(), nr * 4);
    H2D(dof, off.data(), nr * 8); H2D(dkn, knots, nk * 8); H2D(dco, coeffs, nk * 8); H2D(dsp, spectra, nr * 2 * n_f * 16);
#undef H2D
    nrhip::BireBatch b{(long)n_rays, dx1.as<double>(), dx2.as<double>(), dC0.as<double>(), dnp.as<int>(), dof.as<long>(),
                       ctx->ice, dkn.as<double>(), dco.as<double>(), {n_knots[0], n_knots[1], n_knots[2]}, n_ref,
                       angle_to_iceflow, n_f, sampling_rate};
    DevBuf dpieces;
    HIPCHK(dpieces.alloc((size_t)BIRE_MAX_KNOTS * 7 * 8));
    b.spline_pieces = dpieces.as<double>();
    nrhip::launch_birefringence(s, b, max_points, dst.as<double>(), dsp.as<double2>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(spectra, dsp.p, nr * 2 * n_f * 16, hipMemcpyDeviceToHost, s));
    if (step_records && off[n_rays] > 0) HIPCHK(hipMemcpyAsync(step_records, dst.p, (size_t)off[n_rays] * 40, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int nrhip_earth_weights_batch(nrhip_ctx* ctx, int64_t n, const double* zenith, const double* energy, const int32_t* flavor,
                              const double* endpoint, const double* direction, int32_t mode, int32_t cross_section_type,
                              const nrhip_earth_model* model, double step, double nucleon_mass, double* weight,
                              double* slant_depth)
{
    if (!ctx || !zenith || !energy || !flavor) return fail_msg("nrhip_earth_weights_batch: NULL argument");
    if (n < 0) return fail_msg("nrhip_earth_weights_batch: negative size");
    if (mode != NRHIP_EARTH_SIMPLE && mode != NRHIP_EARTH_CORE_MANTLE_CRUST_SIMPLE && mode != NRHIP_EARTH_CHORD)
        return fail_msg("nrhip_earth_weights_batch: mode not supported");  // NotImplementedError (earth_attenuation.py:58-60)
    if (cross_section_type != NRHIP_XS_CTW && cross_section_type != NRHIP_XS_GHANDI && cross_section_type != NRHIP_XS_GIVEN)
        return fail_msg("nrhip_earth_weights_batch: Cross-section not defined");  // cross_sections.py:387-389
    const bool chord = mode == NRHIP_EARTH_CHORD;
    if (chord) {
        if (!endpoint || !direction || !model) return fail_msg("nrhip_earth_weights_batch: the chord mode needs endpoint, direction and model");
        if (model->n_layers < 1 || model->n_layers > NRHIP_EARTH_MAX_LAYERS || !(model->earth_radius > 0))
            return fail_msg("nrhip_earth_weights_batch: 1..16 layers and earth_radius > 0 required");
        if (!(step > 0)) return fail_msg("nrhip_earth_weights_batch: step > 0 required");
    } else if (!weight) {
        return fail_msg("nrhip_earth_weights_batch: NULL argument");
    }
    if (!(nucleon_mass > 0)) return fail_msg("nrhip_earth_weights_batch: nucleon_mass > 0 required");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t nn = (size_t)n;
    DevBuf dz, dE, dfl, dep, ddr, dw, dsd;
    HIPCHK(dz.alloc(nn * 8)); HIPCHK(dE.alloc(nn * 8)); HIPCHK(dfl.alloc(nn * 4)); HIPCHK(dw.alloc(nn * 8));
#define H2D(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, s))
    H2D(dz, zenith, nn * 8); H2D(dE, energy, nn * 8); H2D(dfl, flavor, nn * 4);
    if (chord) {
        HIPCHK(dep.alloc(nn * 24)); HIPCHK(ddr.alloc(nn * 24)); HIPCHK(dsd.alloc(nn * 8));
        H2D(dep, endpoint, nn * 24); H2D(ddr, direction, nn * 24);
    }
#undef H2D
    nrhip::EarthBatch b{};
    b.n = (long)n; b.zenith = dz.as<double>(); b.energy = dE.as<double>(); b.flavor = dfl.as<int>();
    b.endpoint = dep.as<double>(); b.direction = ddr.as<double>(); b.mode = mode; b.cross_section_type = cross_section_type; b.step = step; b.nucleon_mass = nucleon_mass;
    const double kg = 6.241509744511525e+36;  // NuRadioReco/utilities/units.py
    b.amu = 1.66e-27 * kg;                    // earth_attenuation.py:9
    b.simple_radius = 6357390 * 1.;           // :80-81
    b.simple_density = 2900 * kg / (1. * 1. * 1.);
    const double RE = 6.378140e6 * 1.;        // :110-112
    const double dens[3] = {14000.0, 3400.0, 2900.0};
    for (int k = 0; k < 3; k++) b.layer_density[k] = dens[k] * kg / (1. * 1. * 1.);
    b.layer_radii[0] = 3.46e6 * 1.; b.layer_radii[1] = RE - 4.0e4 * 1.; b.layer_radii[2] = RE;
    b.layer_theta[0] = M_PI - std::asin(b.layer_radii[1] / b.layer_radii[2]);
    b.layer_theta[1] = M_PI - std::asin(b.layer_radii[0] / b.layer_radii[2]);
    nrhip::EarthModelDev m{};
    if (chord) {
        m.n_layers = model->n_layers; m.earth_radius = model->earth_radius;
        for (int k = 0; k < model->n_layers; k++) {
            m.radii[k] = model->radii[k];
            for (int j = 0; j < 4; j++) m.coef[k][j] = model->coef[k][j];
        }
    }
    nrhip::launch_earth_weights(s, b, m, dw.as<double>(), chord ? dsd.as<double>() : nullptr);
    HIPCHK(hipGetLastError());
    if (weight) HIPCHK(hipMemcpyAsync(weight, dw.p, nn * 8, hipMemcpyDeviceToHost, s));
    if (chord && slant_depth) HIPCHK(hipMemcpyAsync(slant_depth, dsd.p, nn * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int nrhip_attenuation_length(nrhip_ctx* ctx, int64_t n, const double* z, const double* freq, double* L)
{
    if (!ctx) return fail_msg("nrhip_attenuation_length: ctx is NULL");
    if (n <= 0) return 0;
    if (ctx->att_model == NRHIP_ATT_GL3 && !ctx->gl3) return fail_msg("nrhip_attenuation_length: GL3 needs nrhip_ctx_set_gl3_table");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf dz, df, dl;
    HIPCHK(dz.alloc(n * 8));
    HIPCHK(df.alloc(n * 8));
    HIPCHK(dl.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(dz.p, z, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(df.p, freq, n * 8, hipMemcpyHostToDevice, ctx->stream));
    nrhip::launch_attenuation_length(ctx->stream, n, dz.as<double>(), df.as<double>(), ctx->att_model, dl.as<double>(),
                                     ctx->gl3, ctx->gl3_n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(L, dl.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
