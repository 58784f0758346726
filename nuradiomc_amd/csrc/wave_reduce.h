// wave_reduce.h -- sums of several per-lane values over the 64 lanes of a wave without the LDS crossbar.
//
// `__shfl_xor` compiles to ds_bpermute_b32: one LDS instruction per 32-bit word and butterfly step, 6 steps per value (12 per
// double).  gfx950 has two cheaper tools: v_permlane32_swap / v_permlane16_swap exchange half-waves / rows of TWO registers in one
// VALU instruction, so one swap + one add folds two values into one register; below a row (16 lanes) the DPP modifiers of the
// adder itself (row_ror:8, row_ror:4, quad_perm) finish the sum: 8 floats = 6 swaps + 14 adds instead of 48 bpermutes + 48 adds.
//
// Layout after wave_fold4(x0, x1, x2, x3): every lane of row 0 (lanes 0..15) holds sum(x0), row 1 sum(x2), row 2 sum(x1), row 3
// sum(x3) -- wave_row_value<i>() hands value i back as a wave-uniform number (v_readlane_b32).
// The order of the additions differs from the butterfly's: callers are bounds with their rounding slack, or integers.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}

// lanes 0..31: a(lane) + a(lane + 32); lanes 32..63: b(lane - 32) + b(lane)
__device__ __forceinline__ float wave_fold32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ double wave_fold32(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// rows 0 and 2: a(row) + a(row + 1); rows 1 and 3: b(row - 1) + b(row)
__device__ __forceinline__ float wave_fold16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ double wave_fold16(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// the sum over the 16 lanes of the row, in every lane of it
__device__ __forceinline__ float wave_row_sum(float v)
{
    v += dpp_f32<0x128>(v);  // row_ror:8
    v += dpp_f32<0x124>(v);  // row_ror:4
    v += dpp_f32<0x4e>(v);   // quad_perm:[2,3,0,1]
    v += dpp_f32<0xb1>(v);   // quad_perm:[1,0,3,2]
    return v;
}
__device__ __forceinline__ double wave_row_sum(double v)
{
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x4e>(v);
    v += dpp_f64<0xb1>(v);
    return v;
}
// four sums in one register: row 0 = sum(x0), row 1 = sum(x2), row 2 = sum(x1), row 3 = sum(x3)
template <class T> __device__ __forceinline__ T wave_fold4(T x0, T x1, T x2, T x3)
{
    return wave_row_sum(wave_fold16(wave_fold32(x0, x1), wave_fold32(x2, x3)));
}
// value i (0..3) of a wave_fold4 register, wave-uniform
template <int I> __device__ __forceinline__ float wave_row_value(float v)
{
    constexpr int row = (I == 0) ? 0 : (I == 1) ? 2 : (I == 2) ? 1 : 3;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * row));
}
template <int I> __device__ __forceinline__ double wave_row_value(double v)
{
    constexpr int row = (I == 0) ? 0 : (I == 1) ? 2 : (I == 2) ? 1 : 3;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * row), __builtin_amdgcn_readlane(__double2loint(v), 16 * row));
}
// one value: the wave's sum in every lane
template <class T> __device__ __forceinline__ T wave_sum(T v)
{
    const T h = wave_fold32(v, v);
    return wave_row_sum(wave_fold16(h, h));
}
// integer sum of the wave, in every lane
__device__ __forceinline__ int wave_sum_i32(int v)
{
    const auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = a[0] + a[1];
    const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = b[0] + b[1];
    v += dpp_i32<0x128>(v);
    v += dpp_i32<0x124>(v);
    v += dpp_i32<0x4e>(v);
    v += dpp_i32<0xb1>(v);
    return v;
}
// lanes 32..63: the value of lane - 32 (lanes 0..31: 0)
__device__ __forceinline__ float wave_from_lower_half(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(0, __float_as_int(v), false, false);
    return __int_as_float(r[0]);
}
// the value of the lane below (lane 0: 0): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ float wave_from_lane_below(float v) { return dpp_f32<0x138>(v); }
__device__ __forceinline__ float wave_lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

}  // namespace nrhip
