// api.hip -- C-ABI layer of libnrhip.so (see include/nrhip.h for the contract).
#include "../../include/nrhip.h"
#include "nrhip_internal.h"
#include "ctx.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace nrhip {
void launch_ray_limits(hipStream_t stream, long n_rays, const double* x1, const double* x2, const double* C0,
                       const IceConst& m, double* zint);
void launch_attenuation_items(hipStream_t stream, long n_rays, const double* C0, const double* zint, int n_freq,
                              const double* freqs, int model, const IceConst& m, double* att, int* neval,
                              const int* ray_index, unsigned long long* eval_counter, const double* gl3, int gl3_n);
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
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->gl3) (void)hipFree(ctx->gl3);
    if (ctx->twiddle) (void)hipFree(ctx->twiddle);
    if (ctx->w16) (void)hipFree(ctx->w16);
    delete ctx;
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
                           C0_in ? dgiven.as<double>() : nullptr);
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
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf dx1, dx2, dC0, dz, df, da, dn;
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
                                    ctx->gl3_n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(att, da.p, n_rays * n_freq * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (neval) HIPCHK(hipMemcpyAsync(neval, dn.p, n_rays * n_freq * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nrhip_attenuation_length(nrhip_ctx* ctx, int64_t n, const double* z, const double* freq, double* L)
{
    if (!ctx) return fail_msg("nrhip_attenuation_length: ctx is NULL");
    if (n <= 0) return 0;
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
