// cull.hip -- station-level selection of event groups for arrays of stations (BASELINE configs 3-5).
//
// speedup.distance_cut (NuRadioMC/simulation/simulation.py:155-163) skips a shower for a channel when the vertex is farther from
// the antenna than the shower's cut.  By the triangle inequality a shower farther from the STATION centre than its cut plus the
// station's radius (largest |relative antenna position|) is skipped for every channel of that station, so an event group none
// of whose showers comes closer contributes nothing to the station: no ray, no candidate, no trigger.  (The reference has the
// same quick cut at :1503-1509 -- with its `continue` commented out; it is result-neutral.)  For a 200-station array a shower
// is in range of a few stations only: instead of offering all n x n_channels pairs to the ray tracer, the groups in range are
// gathered into a compact shower list per station, the hot path runs on that, and the triggered flags are OR-ed back.
#include "../../include/nrhip.h"
#include "ctx.h"
#include <vector>

using namespace nrhip;

namespace {

__global__ void cull_flag_kernel(long n_groups, const int* __restrict__ group_begin, const double* __restrict__ vertex,
                                 const double* __restrict__ max_distance, double cx, double cy, double cz, double radius,
                                 int* __restrict__ flag, int* __restrict__ size)
{
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    if (g == n_groups) { flag[g] = 0; size[g] = 0; return; }   // sentinel for the exclusive scans
    const long a = group_begin ? group_begin[g] : g, b = group_begin ? group_begin[g + 1] : g + 1;
    int any = 0;
    for (long i = a; i < b; i++) {
        const double dx = vertex[3 * i] - cx, dy = vertex[3 * i + 1] - cy, dz = vertex[3 * i + 2] - cz;
        // d <= cut + radius, with a relative margin for the rounding of the two square roots (the per-channel test decides)
        if (sqrt(dx * dx + dy * dy + dz * dz) <= (max_distance[i] + radius) * (1. + 1e-12) + 1e-9) any = 1;
    }
    flag[g] = any;
    size[g] = any ? (int)(b - a) : 0;
}

// the same lists for the groups whose flag is set in a device mask (the triggered groups of a survey: pass 2)
__global__ void mask_flag_kernel(long n_groups, const int* __restrict__ group_begin, const unsigned char* __restrict__ mask,
                                 int* __restrict__ flag, int* __restrict__ size)
{
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    if (g == n_groups) { flag[g] = 0; size[g] = 0; return; }
    const long a = group_begin ? group_begin[g] : g, b = group_begin ? group_begin[g + 1] : g + 1;
    const int any = mask[g] ? 1 : 0;
    flag[g] = any;
    size[g] = any ? (int)(b - a) : 0;
}

__global__ void cull_scatter_kernel(long n_groups, const int* __restrict__ flag, const int* __restrict__ offset,
                                    const int* __restrict__ shower_offset, int* __restrict__ keep_index,
                                    int* __restrict__ group_begin_out)
{
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    if (g == n_groups) { group_begin_out[offset[g]] = shower_offset[g]; return; }   // closing entry: total number of showers
    if (!flag[g]) return;
    keep_index[offset[g]] = (int)g;
    group_begin_out[offset[g]] = shower_offset[g];
}

template <class T, int W>
__device__ inline void copy_row(T* dst, const T* src, long o, long i)
{
    if (!dst || !src) return;
    for (int d = 0; d < W; d++) dst[W * o + d] = src[W * i + d];
}

__global__ void gather_showers_kernel(long n_keep, const int* __restrict__ keep_index, const int* __restrict__ group_begin,
                                      const int* __restrict__ group_begin_out, const double* vertex, const double* zenith,
                                      const double* azimuth, const double* energy, const int* shower_type, const double* k_L,
                                      const double* vertex_time, const double* max_distance, double* o_vertex, double* o_zenith,
                                      double* o_azimuth, double* o_energy, int* o_type, double* o_kL, double* o_vertex_time,
                                      double* o_max_distance, int* __restrict__ shower_index)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keep) return;
    const long g = keep_index[k];
    const long a = group_begin ? group_begin[g] : g, b = group_begin ? group_begin[g + 1] : g + 1;
    long o = group_begin_out[k];
    for (long i = a; i < b; i++, o++) {
        copy_row<double, 3>(o_vertex, vertex, o, i);
        copy_row<double, 1>(o_zenith, zenith, o, i);
        copy_row<double, 1>(o_azimuth, azimuth, o, i);
        copy_row<double, 1>(o_energy, energy, o, i);
        copy_row<int, 1>(o_type, shower_type, o, i);
        copy_row<double, 1>(o_kL, k_L, o, i);
        copy_row<double, 1>(o_vertex_time, vertex_time, o, i);
        copy_row<double, 1>(o_max_distance, max_distance, o, i);
        if (shower_index) shower_index[o] = (int)i;
    }
}

__global__ void mask_scatter_or_kernel(long n, const int* __restrict__ index, const unsigned char* __restrict__ src,
                                       unsigned char* __restrict__ dst)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && src[k]) dst[index[k]] = 1;
}

__global__ void index_to_i64_kernel(long n, const int* __restrict__ index, long long offset, long long* __restrict__ out)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = offset + (index ? (long long)index[k] : (long long)k);
}

__global__ void gather_i64_kernel(long n, const int* __restrict__ index, const long long* __restrict__ src, long long* __restrict__ out)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = src[index[k]];
}

inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int nrhip_cull_groups(nrhip_ctx* ctx, int64_t n_showers, int64_t n_groups, const int32_t* group_begin, const double* vertex,
                      const double* max_distance, const double centre[3], double radius, int32_t* keep_index,
                      int32_t* group_begin_out, int64_t* n_keep, int64_t* n_showers_out)
{
    if (!ctx || !vertex || !max_distance || !centre || !keep_index || !group_begin_out || !n_keep || !n_showers_out)
        return nrhip_fail_msg("nrhip_cull_groups: NULL argument");
    if (n_showers < 0 || n_groups < 0 || n_groups > n_showers || (!group_begin && n_groups != n_showers))
        return nrhip_fail_msg("nrhip_cull_groups: bad sizes");
    *n_keep = *n_showers_out = 0;
    if (n_groups == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const long n1 = n_groups + 1;
    // scratch of the context: flags, sizes, their scans
    HIPCHK(ctx->cull_ws.reserve(sizeof(int) * (size_t)(4 * n1 + 2 * scan_tiles(n1) + 16)));
    int* flag = ctx->cull_ws.as<int>();
    int *size = flag + n1, *off = size + n1, *soff = off + n1, *tmp = soff + n1, *tmp2 = tmp + scan_tiles(n1);
    hipLaunchKernelGGL(cull_flag_kernel, dim3(blocks(n1)), dim3(256), 0, s, (long)n_groups, group_begin, vertex, max_distance,
                       centre[0], centre[1], centre[2], radius, flag, size);
    launch_exclusive_scan(s, n1, flag, off, tmp);
    launch_exclusive_scan(s, n1, size, soff, tmp2);
    hipLaunchKernelGGL(cull_scatter_kernel, dim3(blocks(n1)), dim3(256), 0, s, (long)n_groups, flag, off, soff, keep_index,
                       group_begin_out);
    HIPCHK(hipGetLastError());
    int h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&h[0], off + n_groups, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&h[1], soff + n_groups, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_keep = h[0];
    *n_showers_out = h[1];
    return 0;
}

int nrhip_select_groups(nrhip_ctx* ctx, int64_t n_showers, int64_t n_groups, const int32_t* group_begin, const uint8_t* mask,
                        int32_t* keep_index, int32_t* group_begin_out, int64_t* n_keep, int64_t* n_showers_out)
{
    if (!ctx || !mask || !keep_index || !group_begin_out || !n_keep || !n_showers_out)
        return nrhip_fail_msg("nrhip_select_groups: NULL argument");
    if (n_showers < 0 || n_groups < 0 || n_groups > n_showers || (!group_begin && n_groups != n_showers))
        return nrhip_fail_msg("nrhip_select_groups: bad sizes");
    *n_keep = *n_showers_out = 0;
    if (n_groups == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const long n1 = n_groups + 1;
    HIPCHK(ctx->cull_ws.reserve(sizeof(int) * (size_t)(4 * n1 + 2 * scan_tiles(n1) + 16)));
    int* flag = ctx->cull_ws.as<int>();
    int *size = flag + n1, *off = size + n1, *soff = off + n1, *tmp = soff + n1, *tmp2 = tmp + scan_tiles(n1);
    hipLaunchKernelGGL(mask_flag_kernel, dim3(blocks(n1)), dim3(256), 0, s, (long)n_groups, group_begin, mask, flag, size);
    launch_exclusive_scan(s, n1, flag, off, tmp);
    launch_exclusive_scan(s, n1, size, soff, tmp2);
    hipLaunchKernelGGL(cull_scatter_kernel, dim3(blocks(n1)), dim3(256), 0, s, (long)n_groups, flag, off, soff, keep_index,
                       group_begin_out);
    HIPCHK(hipGetLastError());
    int h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&h[0], off + n_groups, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&h[1], soff + n_groups, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_keep = h[0];
    *n_showers_out = h[1];
    return 0;
}

int nrhip_gather_groups(nrhip_ctx* ctx, int64_t n_keep, const int32_t* keep_index, const int32_t* group_begin,
                        const int32_t* group_begin_out, const double* vertex, const double* zenith, const double* azimuth,
                        const double* energy, const int32_t* shower_type, const double* k_L, const double* vertex_time,
                        const double* max_distance, double* o_vertex, double* o_zenith, double* o_azimuth, double* o_energy,
                        int32_t* o_shower_type, double* o_k_L, double* o_vertex_time, double* o_max_distance,
                        int32_t* shower_index)
{
    if (!ctx || (n_keep > 0 && (!keep_index || !group_begin_out))) return nrhip_fail_msg("nrhip_gather_groups: NULL argument");
    if (n_keep <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gather_showers_kernel, dim3(blocks(n_keep)), dim3(256), 0, ctx->stream, (long)n_keep, keep_index,
                       group_begin, group_begin_out, vertex, zenith, azimuth, energy, shower_type, k_L, vertex_time, max_distance,
                       o_vertex, o_zenith, o_azimuth, o_energy, o_shower_type, o_k_L, o_vertex_time, o_max_distance, shower_index);
    HIPCHK(hipGetLastError());
    return 0;
}

int nrhip_index_to_i64(nrhip_ctx* ctx, int64_t n, const int32_t* index, int64_t offset, int64_t* out)
{
    if (!ctx || (n > 0 && !out)) return nrhip_fail_msg("nrhip_index_to_i64: NULL argument");
    if (n <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(index_to_i64_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, (long)n, index, (long long)offset, (long long*)out);
    HIPCHK(hipGetLastError());
    return 0;
}

int nrhip_gather_i64(nrhip_ctx* ctx, int64_t n, const int32_t* index, const int64_t* src, int64_t* out)
{
    if (!ctx || (n > 0 && (!index || !src || !out))) return nrhip_fail_msg("nrhip_gather_i64: NULL argument");
    if (n <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gather_i64_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, (long)n, index, (const long long*)src, (long long*)out);
    HIPCHK(hipGetLastError());
    return 0;
}

int nrhip_mask_scatter_or(nrhip_ctx* ctx, int64_t n, const int32_t* index, const uint8_t* src, uint8_t* dst)
{
    if (!ctx || (n > 0 && (!index || !src || !dst))) return nrhip_fail_msg("nrhip_mask_scatter_or: NULL argument");
    if (n <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(mask_scatter_or_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, (long)n, index, src, dst);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
