// conv_fft.h -- the transform pair of channel_conv_kernel, built around WAVE-PRIVATE sub-transforms.
//
// The real 2M-point circular convolution of efieldToVoltageConverter (NuRadioReco/modules/efieldToVoltageConverter.py:223-345:
// rfft -> * antenna response * filter -> irfft) is a packed complex M-point transform pair in LDS (M = 8192 or 4096 complex128).
// Round 3 ran it as nine block-wide passes (4 forward, spectrum product, 4 inverse), every one ending in __syncthreads(): with one
// 128 KB block per CU all eight waves sat in the same phase (all reading LDS, then all computing, then all writing) and 57 % of the
// wave-cycles were parked on barriers.  Here:
//
//   * the M points are 8 (4, 2) blocks of 1024, one per wave.  Only the first pass of the forward transform (radix 8 / 4 across the
//     blocks) and the last of the inverse are block-wide; the 1024-point transforms in between (radix 16, then radix 8) belong to
//     ONE wave each, which synchronises with itself only -- the waves drift apart, one's LDS traffic under another's arithmetic;
//   * the last three stages of the forward transform, the real-transform split, the product with the response spectrum G, the merge
//     and the first three stages of the inverse are ONE pass (a thread owns the eight bins k0 + j M/8 and their mirror partners
//     M - k): seven passes over the LDS instead of nine, five block barriers instead of eleven;
//   * every pass is laid out so that consecutive lanes touch consecutive elements (16-byte LDS accesses, no bank conflicts): between
//     the wave-private passes the data is transposed inside the wave's block on the way (a pass writes the order the next one reads),
//     and the last wave-private pass leaves the spectrum in natural order of the bin index k0 -- so the response spectrum is read
//     from HBM / L2 in coalesced runs;
//   * twiddles come from small per-pass tables (cft: 15 x 64 + 7 x 8 entries behind w16, L1 resident, coalesced) holding exactly the
//     master table's values: the butterflies are those of a radix-2 transform, their grouping into passes does not change a bit of
//     the result (the 4096-point transform gives the same bits in the 256- and in the 512-thread kernel).
//   * the upper half of the packed input is zero by construction (L <= M real samples): the first stage of the forward transform
//     reads half of its inputs.
//
// Layout: complex element i of the buffer lives at conv_pad(i) = i + (i >> 10) -- one element of padding per 1024-block, so that
// lanes walking ACROSS the blocks (spectrum pass) fall on different banks.
#pragma once
#include "fft_device.h"

namespace nrhip {

// Nothing in this header is left to the optimiser's choice of what to fuse: the translation unit is compiled with
// -ffp-contract=fast-honor-pragmas, under which the two instantiations of channel_conv_kernel (256 and 512 threads) got different
// fusions of the same source lines and their traces differed in the last bit.  Contraction is off from here to the end of the
// header (and around the kernel and its ray functions in spectral.hip); the fused multiply-adds are written out (cmulx, cmulcx).
#pragma clang fp contract(off)

__device__ __forceinline__ double2 cmulx(double2 a, double2 b)    // a * b: two products, two fused multiply-adds
{
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cmulcx(double2 a, double2 b)   // a * conj(b)
{
    return make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y)));
}


// Tables (twiddles, response spectra) live in HBM / L2.  Through a plain pointer parameter of an out-of-line function -- or one that
// went through conv_opaque -- the compiler no longer knows that and issues FLAT loads, which count against BOTH memory counters: every
// wait for an LDS read (lgkmcnt) then also waits for the table loads in flight, and the passes came out as "request everything, wait
// for everything, compute".  gload() reads through an explicit global address: global_load, counted by vmcnt alone.
typedef double conv_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 gload(const double2* p)
{
    const conv_d2v v = *(const __attribute__((address_space(1))) conv_d2v*)p;
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ double gload(const double* p) { return *(const __attribute__((address_space(1))) double*)p; }
__device__ __forceinline__ int gload(const int* p) { return *(const __attribute__((address_space(1))) int*)p; }
__device__ __forceinline__ unsigned char gload(const unsigned char* p) { return *(const __attribute__((address_space(1))) unsigned char*)p; }
__device__ __forceinline__ void gstore(double2* p, const double2 v)
{
    conv_d2v w;
    w.x = v.x;
    w.y = v.y;
    *(__attribute__((address_space(1))) conv_d2v*)p = w;
}
__device__ __forceinline__ void gstore(double* p, double v) { *(__attribute__((address_space(1))) double*)p = v; }

__device__ __forceinline__ int conv_pad(int i) { return i + (i >> 10); }
// complex elements of the padded buffer: the M points + block padding, and behind the event's samples (M / 2 + padding) room for
// the wave-private ray transforms (8 or 4 blocks of 512 points, block stride 532 / 520: ray_blk_stride)
__host__ __device__ constexpr int conv_lds_elems(int M) { return M + (M >= 8192 ? 176 : 48); }
constexpr int CFT_T2 = 0;             // [15][64]: W_1024^(l + 64 s) s < 8 | W_512^(l + 64 s) s < 4 | W_256^(l + 64 s) s < 2 | W_128^l
constexpr int CFT_T3 = 15 * 64;       // [7][8]:   W_64^(c + 8 s) s < 4 | W_32^(c + 8 s) s < 2 | W_16^c
constexpr int CFT_SIZE = CFT_T3 + 7 * 8;

// LDS accesses of one wave complete in program order: a wave that hands data between its own lanes needs no s_barrier, only that
// the compiler keeps the order
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The twiddles of a thread depend on nothing but its index: left alone, the optimiser hoists every table load of every pass out of
// the kernel's event and channel loops and keeps ~200 registers of twiddles alive across them (spilled).  An opaque copy of the table
// pointer per call keeps the loads where they are used.
__device__ __forceinline__ const double2* conv_opaque(const double2* p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// ... and an opaque copy of the thread index keeps the (64-bit) addresses derived from it from being computed once at kernel
// start and parked in scratch
__device__ __forceinline__ int conv_opaque(int t)
{
    asm volatile("" : "+v"(t));
    return t;
}

// Block barrier for data that is handed over through LDS only: __syncthreads() also waits for every global store in flight (its
// release fence covers all address spaces) -- with the traces of a triggered event streaming out to HBM that is a microsecond per
// barrier.  What the threads of channel_conv_kernel hand to each other goes through LDS -- with ONE exception, the per-sample
// coincidence counts (global scratch): the two points where those change hands use __syncthreads() (spectral.hip).
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ void dif_bf(double2& a, double2& b, const double2 w)
{
    const double2 d = csub(a, b);
    a = cadd(a, b);
    b = cmulx(d, w);
}
__device__ __forceinline__ void dif_bf1(double2& a, double2& b)   // twiddle 1
{
    const double2 d = csub(a, b);
    a = cadd(a, b);
    b = d;
}
__device__ __forceinline__ void dit_bf(double2& a, double2& b, const double2 w)   // w: the forward twiddle (conjugated here)
{
    const double2 t = cmulcx(b, w);
    b = csub(a, t);
    a = cadd(a, t);
}
__device__ __forceinline__ void dit_bf1(double2& a, double2& b)
{
    const double2 t = b;
    b = csub(a, t);
    a = cadd(a, t);
}
// multiplications by the eighth roots of unity W_8^1 = (1 - i) / sqrt 2, W_8^2 = -i, W_8^3 = -(1 + i) / sqrt 2 and their conjugates
__device__ __forceinline__ double2 mul_w8_1(double2 a) { const double c = 0.70710678118654752440; return make_double2((a.x + a.y) * c, (a.y - a.x) * c); }
__device__ __forceinline__ double2 mul_w8_2(double2 a) { return make_double2(a.y, -a.x); }
__device__ __forceinline__ double2 mul_w8_3(double2 a) { const double c = 0.70710678118654752440; return make_double2((a.y - a.x) * c, -(a.x + a.y) * c); }
__device__ __forceinline__ double2 mul_w8_1c(double2 a) { const double c = 0.70710678118654752440; return make_double2((a.x - a.y) * c, (a.x + a.y) * c); }
__device__ __forceinline__ double2 mul_w8_2c(double2 a) { return make_double2(-a.y, a.x); }
__device__ __forceinline__ double2 mul_w8_3c(double2 a) { const double c = 0.70710678118654752440; return make_double2(-(a.x + a.y) * c, (a.x - a.y) * c); }

// the last three stages (spans 4, 2, 1) of a decimation-in-frequency transform on eight consecutive positions, in place
__device__ __forceinline__ void dif8_tail(double2 (&a)[8])
{
    { double2 d; d = csub(a[0], a[4]); a[0] = cadd(a[0], a[4]); a[4] = d; }
    { double2 d; d = csub(a[1], a[5]); a[1] = cadd(a[1], a[5]); a[5] = mul_w8_1(d); }
    { double2 d; d = csub(a[2], a[6]); a[2] = cadd(a[2], a[6]); a[6] = mul_w8_2(d); }
    { double2 d; d = csub(a[3], a[7]); a[3] = cadd(a[3], a[7]); a[7] = mul_w8_3(d); }
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
        { double2 d; d = csub(a[b], a[b + 2]); a[b] = cadd(a[b], a[b + 2]); a[b + 2] = d; }
        { double2 d; d = csub(a[b + 1], a[b + 3]); a[b + 1] = cadd(a[b + 1], a[b + 3]); a[b + 3] = mul_w8_2(d); }
    }
#pragma unroll
    for (int b = 0; b < 8; b += 2) dif_bf1(a[b], a[b + 1]);
}
// ... and the first three stages (spans 1, 2, 4) of the decimation-in-time inverse
__device__ __forceinline__ void dit8_head(double2 (&a)[8])
{
#pragma unroll
    for (int b = 0; b < 8; b += 2) dit_bf1(a[b], a[b + 1]);
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
        dit_bf1(a[b], a[b + 2]);
        { const double2 t = mul_w8_2c(a[b + 3]); a[b + 3] = csub(a[b + 1], t); a[b + 1] = cadd(a[b + 1], t); }
    }
    dit_bf1(a[0], a[4]);
    { const double2 t = mul_w8_1c(a[5]); a[5] = csub(a[1], t); a[1] = cadd(a[1], t); }
    { const double2 t = mul_w8_2c(a[6]); a[6] = csub(a[2], t); a[2] = cadd(a[2], t); }
    { const double2 t = mul_w8_3c(a[7]); a[7] = csub(a[3], t); a[3] = cadd(a[3], t); }
}

__device__ __forceinline__ int br3(int r) { return ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1); }
__device__ __forceinline__ int br4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r >> 3) & 1); }

// ---- wave-private passes on the wave's 1024-block (base = padded index of its first element) ------------------------------------
// forward, stages with spans 512 .. 64: lane l holds the elements l + 64 j; written transposed for the next pass
__device__ __forceinline__ void conv_p2_fwd(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[16], t[15];
#pragma unroll
    for (int s = 0; s < 15; s++) t[s] = gload(&cft[CFT_T2 + s * 64 + lane]);
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = zb[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < 8; j++) dif_bf(a[j], a[j + 8], t[j]);
#pragma unroll
    for (int b = 0; b < 16; b += 8)
#pragma unroll
        for (int j = 0; j < 4; j++) dif_bf(a[b + j], a[b + j + 4], t[8 + j]);
#pragma unroll
    for (int b = 0; b < 16; b += 4)
#pragma unroll
        for (int j = 0; j < 2; j++) dif_bf(a[b + j], a[b + j + 2], t[12 + j]);
#pragma unroll
    for (int b = 0; b < 16; b += 2) dif_bf(a[b], a[b + 1], t[14]);
    wave_lds_sync();
    const int wb = (lane >> 3) * 128 + (lane & 7);
#pragma unroll
    for (int j = 0; j < 16; j++) zb[wb + (j >> 3) * 64 + 8 * (j & 7)] = a[j];
    wave_lds_sync();
}
// inverse of conv_p2_fwd
__device__ __forceinline__ void conv_p2_inv(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[16], t[15];
#pragma unroll
    for (int s = 0; s < 15; s++) t[s] = gload(&cft[CFT_T2 + s * 64 + lane]);
    const int wb = (lane >> 3) * 128 + (lane & 7);
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = zb[wb + (j >> 3) * 64 + 8 * (j & 7)];
#pragma unroll
    for (int b = 0; b < 16; b += 2) dit_bf(a[b], a[b + 1], t[14]);
#pragma unroll
    for (int b = 0; b < 16; b += 4)
#pragma unroll
        for (int j = 0; j < 2; j++) dit_bf(a[b + j], a[b + j + 2], t[12 + j]);
#pragma unroll
    for (int b = 0; b < 16; b += 8)
#pragma unroll
        for (int j = 0; j < 4; j++) dit_bf(a[b + j], a[b + j + 4], t[8 + j]);
#pragma unroll
    for (int j = 0; j < 8; j++) dit_bf(a[j], a[j + 8], t[j]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 16; j++) zb[lane + 64 * j] = a[j];
}
// forward, stages with spans 32, 16, 8: lane (c = l & 7, jl = l >> 3) holds, for u = 0, 1, the eight elements h of (c, j = jl + 8 u);
// written at c + 8 (br4(j) + 16 br3(h)): the block then holds the bins k0 = (wave's residue) + NW (br4(j) + 16 br3(h)) in natural
// order of k0, eight consecutive positions c per bin group
__device__ __forceinline__ void conv_p3_fwd(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[2][8], t[7];
    const int c = lane & 7, jl = lane >> 3;
#pragma unroll
    for (int s = 0; s < 7; s++) t[s] = gload(&cft[CFT_T3 + s * 8 + c]);
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int h = 0; h < 8; h++) a[u][h] = zb[h * 128 + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int h = 0; h < 4; h++) dif_bf(a[u][h], a[u][h + 4], t[h]);
#pragma unroll
        for (int b = 0; b < 8; b += 4)
#pragma unroll
            for (int h = 0; h < 2; h++) dif_bf(a[u][b + h], a[u][b + h + 2], t[4 + h]);
#pragma unroll
        for (int b = 0; b < 8; b += 2) dif_bf(a[u][b], a[u][b + 1], t[6]);
    }
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int wb = c + 8 * br4(jl + 8 * u);
#pragma unroll
        for (int h = 0; h < 8; h++) zb[wb + 128 * br3(h)] = a[u][h];
    }
}
__device__ __forceinline__ void conv_p3_inv(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[2][8], t[7];
    const int c = lane & 7, jl = lane >> 3;
#pragma unroll
    for (int s = 0; s < 7; s++) t[s] = gload(&cft[CFT_T3 + s * 8 + c]);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int wb = c + 8 * br4(jl + 8 * u);
#pragma unroll
        for (int h = 0; h < 8; h++) a[u][h] = zb[wb + 128 * br3(h)];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int b = 0; b < 8; b += 2) dit_bf(a[u][b], a[u][b + 1], t[6]);
#pragma unroll
        for (int b = 0; b < 8; b += 4)
#pragma unroll
            for (int h = 0; h < 2; h++) dit_bf(a[u][b + h], a[u][b + h + 2], t[4 + h]);
#pragma unroll
        for (int h = 0; h < 4; h++) dit_bf(a[u][h], a[u][h + 4], t[h]);
    }
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int h = 0; h < 8; h++) zb[h * 128 + u * 64 + lane] = a[u][h];
    wave_lds_sync();
}

// ---- the rays' N / 2-point inverse transforms (N / 2 = 512, 1024, 2048), wave-private like the above --------------------------------
// A transform of nh points is NBk = nh / 512 blocks of 512, one wave each, eight points per lane.  The spectrum is built by the
// threads that do the first log2(NBk) stages across the blocks (channel_conv_kernel), then per wave: spans 256, 128, 64 (ray_p2),
// spans 32, 16, 8 (ray_p3), and the last three stages are done by the threads that place the samples on the event's grid.
// Decimation in frequency with the conjugate twiddles (inverse transform, unscaled): natural-order spectrum in, sample j at the
// position that the passes leave it.
__device__ __forceinline__ void dif_bf_c(double2& a, double2& b, const double2 w)   // (a - b) * conj(w)
{
    const double2 d = csub(a, b);
    a = cadd(a, b);
    b = cmulcx(d, w);
}
__host__ __device__ constexpr int ray_blk_stride(int nbk) { return nbk == 4 ? 532 : 520; }
// lane l holds l + 64 j (j < 8); written transposed: (l >> 3) 64 + (l & 7) + 8 j
__device__ __forceinline__ void ray_p2(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[8], t[7];
#pragma unroll
    for (int s = 0; s < 7; s++) t[s] = gload(&cft[CFT_T2 + (8 + s) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = zb[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < 4; j++) dif_bf_c(a[j], a[j + 4], t[j]);
#pragma unroll
    for (int b = 0; b < 8; b += 4)
#pragma unroll
        for (int j = 0; j < 2; j++) dif_bf_c(a[b + j], a[b + j + 2], t[4 + j]);
#pragma unroll
    for (int b = 0; b < 8; b += 2) dif_bf_c(a[b], a[b + 1], t[6]);
    wave_lds_sync();
    const int wb = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
    for (int j = 0; j < 8; j++) zb[wb + 8 * j] = a[j];
    wave_lds_sync();
}
// lane (c = l & 7, j = l >> 3) holds the eight elements h; written at 65 c + x, x = br3(j) + 8 br3(h): row c of the block then
// holds, in natural order of x, the groups of eight positions whose last three stages give the samples x' + (nh / 8) m
__device__ __forceinline__ void ray_p3(double2* zb, const double2* __restrict__ cft, int lane)
{
    double2 a[8], t[7];
    const int c = lane & 7, j = lane >> 3;
#pragma unroll
    for (int s = 0; s < 7; s++) t[s] = gload(&cft[CFT_T3 + s * 8 + c]);
#pragma unroll
    for (int h = 0; h < 8; h++) a[h] = zb[h * 64 + lane];
#pragma unroll
    for (int h = 0; h < 4; h++) dif_bf_c(a[h], a[h + 4], t[h]);
#pragma unroll
    for (int b = 0; b < 8; b += 4)
#pragma unroll
        for (int h = 0; h < 2; h++) dif_bf_c(a[b + h], a[b + h + 2], t[4 + h]);
#pragma unroll
    for (int b = 0; b < 8; b += 2) dif_bf_c(a[b], a[b + 1], t[6]);
    wave_lds_sync();
    const int wb = 65 * c + br3(j);
#pragma unroll
    for (int h = 0; h < 8; h++) zb[wb + 8 * br3(h)] = a[h];
}
// last three stages (spans 4, 2, 1) with the conjugate twiddles
__device__ __forceinline__ void dif8_tail_c(double2 (&a)[8])
{
    { double2 d; d = csub(a[0], a[4]); a[0] = cadd(a[0], a[4]); a[4] = d; }
    { double2 d; d = csub(a[1], a[5]); a[1] = cadd(a[1], a[5]); a[5] = mul_w8_1c(d); }
    { double2 d; d = csub(a[2], a[6]); a[2] = cadd(a[2], a[6]); a[6] = mul_w8_2c(d); }
    { double2 d; d = csub(a[3], a[7]); a[3] = cadd(a[3], a[7]); a[7] = mul_w8_3c(d); }
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
        { double2 d; d = csub(a[b], a[b + 2]); a[b] = cadd(a[b], a[b + 2]); a[b + 2] = d; }
        { double2 d; d = csub(a[b + 1], a[b + 3]); a[b + 1] = cadd(a[b + 1], a[b + 3]); a[b + 3] = mul_w8_2c(d); }
    }
#pragma unroll
    for (int b = 0; b < 8; b += 2) dif_bf1(a[b], a[b + 1]);
}

// ---- forward transform up to (not including) the last three stages ----------------------------------------------------------------
// z: natural order, only the lower half non-zero (the upper half is not read).  NT threads call; the M / 16 first ones work.
// Ends WITHOUT a block barrier (conv_mid starts with one).
// (conv_fwd / conv_mid / conv_inv are real function calls, not inlined: channel_conv_kernel keeps ~400 scalars alive (its argument
// structures), and inlined into it the passes came out with spill reloads and lane reads of spilled scalars in their inner code; a
// call gives each pass its own register allocation, at the price of one s_swappc.  The buffer is the kernel's dynamic LDS.)
// FULL: the upper half of the input holds data as well (the chirp convolutions of the trigger-ADC chain with more than M / 2 inputs).
// nv: the elements behind index nv are taken as zero (not read): the caller need not clear them.
// first pass (block-wide, spans M / 2 .. 1024) and the wave-private pass with spans 512 .. 64; ends before the last forward pass
template <int LOG2M, int NT, bool FULL>
__device__ __forceinline__ void conv_fwd_head(const double2* __restrict__ tw, const double2* __restrict__ cft, int nv)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    constexpr int M = 1 << LOG2M, NA = M / 16, LW = LOG2M - 10, NB = 1 << LW;   // NA active threads, NB blocks of 1024
    static_assert(LOG2M >= 11 && LOG2M <= 13, "2048, 4096 or 8192 points");
    static_assert(NA <= NT, "16 points per thread");
    const int t = threadIdx.x;
    if (NA == NT || t < NA) {
#pragma unroll
        for (int g = 0; g < 16 / NB; g++) {
            const int i0 = t + NA * g;
            double2 a[NB];
#pragma unroll
            for (int j = 0; j < (FULL ? NB : NB / 2); j++) a[j] = (j * 1024 + i0 < nv) ? z[j * 1025 + i0] : make_double2(0., 0.);
            // first stage (span M / 2): the partners are zero unless FULL
#pragma unroll
            for (int j = 0; j < NB / 2; j++) {
                if (FULL) dif_bf(a[j], a[j + NB / 2], gload(&tw[(i0 + 1024 * j) * (8 >> LW)]));
                else a[j + NB / 2] = cmulx(a[j], gload(&tw[(i0 + 1024 * j) * (8 >> LW)]));
            }
#pragma unroll
            for (int e = 1; e < LW; e++) {
                const int half = NB >> (e + 1);
#pragma unroll
                for (int b = 0; b < NB; b += 2 * half)
#pragma unroll
                    for (int j = 0; j < half; j++) dif_bf(a[b + j], a[b + j + half], gload(&tw[(i0 + 1024 * j) * ((8 >> LW) << e)]));
            }
#pragma unroll
            for (int j = 0; j < NB; j++) z[j * 1025 + i0] = a[j];
        }
    }
    lds_barrier();
    if (NA == NT || t < NA) conv_p2_fwd(z + (t >> 6) * 1025, cft, t & 63);
}
template <int LOG2M, int NT, bool FULL = false>
__device__ __noinline__ void conv_fwd(const double2* __restrict__ tw, const double2* __restrict__ cft, int nv)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    constexpr int NA = (1 << LOG2M) / 16;
    const int t = threadIdx.x;
    conv_fwd_head<LOG2M, NT, FULL>(tw, cft, nv);
    if (NA == NT || t < NA) conv_p3_fwd(z + (t >> 6) * 1025, cft, t & 63);
}

// real-transform split, product with the response spectrum, merge -- on the mirror pair (k, Mr - k) of the packed transform
// (the arithmetic of round 3's spectrum pass)
__device__ __forceinline__ void conv_pair_mul(double2& A, double2& B, const double2 Gk, const double2 Gm, const double2 wk)
{
    const double2 Bc = cconj(B);
    const double2 Ee = cadd(A, Bc), D = csub(A, Bc);
    const double2 O = make_double2(D.y, -D.x);
    const double2 wO = cmulx(wk, O);
    const double2 Yk = cmulx(cadd(Ee, wO), Gk);
    const double2 Ymc = cconj(cmulx(cconj(csub(Ee, wO)), Gm));
    const double2 E2 = cadd(Yk, Ymc);
    const double2 D2 = cmulcx(csub(Yk, Ymc), wk);
    A = make_double2(E2.x - D2.y, E2.y + D2.x);
    B = make_double2(E2.x + D2.y, D2.x - E2.y);
}

// ---- last three forward stages + spectrum product + first three inverse stages ------------------------------------------------------
// G: response spectrum on the 2 FFT_MAX-point grid (bin k of the 2 M-point one at G[gs k], gs = FFT_MAX / M), w16[k] = exp(-i pi k /
// FFT_MAX).  Thread k0 (1 <= k0 < M / 16) owns the bins k0 + j M/8 and their mirror partners (M/8 - k0) + j M/8; thread 0 the two
// groups that are their own mirrors (k0 = 0 and M / 16).  Starts and ends with a block barrier.
// the first batch of table values of conv_mid_impl (slots 0 and 1, the two w16 entries, thread 0's extra pair): 32 registers
struct ConvMidPre { double2 wkA, wkB, Gk[2], Gm[2], Gh, wh; };
template <int LOG2M, int NT>
__device__ __forceinline__ ConvMidPre conv_mid_request(const double2* __restrict__ G, const double2* __restrict__ w16)
{
    constexpr int M = 1 << LOG2M, NA = M / 16, K = M / 8, gs = FFT_MAX / M;
    const int t = threadIdx.x;
    const int kA = (t == 0) ? 0 : t, kB = (t == 0) ? K / 2 : K - t;
    ConvMidPre p;
    p.wkA = p.wkB = p.Gh = p.wh = make_double2(1., 0.);
#pragma unroll
    for (int r = 0; r < 2; r++) p.Gk[r] = p.Gm[r] = make_double2(0., 0.);
    if (NA == NT || t < NA) {
        p.wkA = gload(&w16[gs * kA]);   // thread 0: 1 and exp(-i pi / 16)
        p.wkB = gload(&w16[gs * kB]);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int m = br3(r);
            const int kn = (m < 4) ? kA + K * m : kB + K * (7 - m);
            p.Gk[r] = gload(&G[gs * kn]);
            p.Gm[r] = gload(&G[gs * (M - kn)]);
        }
        if (t == 0) { p.Gh = gload(&G[gs * (M / 2)]); p.wh = gload(&w16[gs * (M / 2)]); }   // (requested late, inside the slot loop: 58 saved registers)
    }
    return p;
}
template <int LOG2M, int NT>
__device__ __forceinline__ void conv_mid_impl(const double2* __restrict__ G, const double2* __restrict__ w16, const ConvMidPre& pre)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    constexpr int M = 1 << LOG2M, NA = M / 16, K = M / 8, LW = LOG2M - 10, gs = FFT_MAX / M;
    const int t = threadIdx.x;
    const bool act = NA == NT || t < NA;
    const int kA = (t == 0) ? 0 : t, kB = (t == 0) ? K / 2 : K - t;
    const bool sp = t == 0;
    // The table values of the pass depend on the thread index alone.  Round 6: they are requested two slots at a time, two products
    // ahead of their use -- the first two (`pre`, conv_mid_request) in front of the barrier and the LDS reads, the next two behind the
    // butterflies, then two in front of every second product -- so that every batch's L2 / HBM latency lies under arithmetic and at
    // most four slots' values (32 registers) are in flight.  (All sixteen up front: 22 callee-saved registers stored and reloaded
    // per call, 3.3 GB of scratch writes per launch, no faster.  Until round 5 both halves were requested behind the butterflies
    // with a full wait each: the latency was exposed twice per channel.)  Slot r reads the bins kn(r) and M - kn(r).
    double2 wkA = pre.wkA, wkB = pre.wkB, Gk[8], Gm[8], Gh = pre.Gh, wh = pre.wh;
#pragma unroll
    for (int r = 0; r < 2; r++) { Gk[r] = pre.Gk[r]; Gm[r] = pre.Gm[r]; }
    auto request = [&](int r) {
        const int m = br3(r);
        const int kn = (m < 4) ? kA + K * m : kB + K * (7 - m);
        Gk[r] = gload(&G[gs * kn]);
        Gm[r] = gload(&G[gs * (M - kn)]);
    };
    lds_barrier();
    if (act) {
        // block of bin group k0: the wave whose residue it is (bit-reversed), position inside: 8 (k0 >> LW)
        const int wA = (LW == 3) ? br3(kA & 7) : (LW == 2 ? (((kA & 1) << 1) | ((kA >> 1) & 1)) : (kA & 1));
        const int wB = (LW == 3) ? br3(kB & 7) : (LW == 2 ? (((kB & 1) << 1) | ((kB >> 1) & 1)) : (kB & 1));
        double2* pa = z + wA * 1025 + 8 * (kA >> LW);
        double2* pb = z + wB * 1025 + 8 * (kB >> LW);
        double2 A[8], B[8];
#pragma unroll
        for (int c = 0; c < 8; c++) { A[c] = pa[c]; B[c] = pb[c]; }
        dif8_tail(A);
        dif8_tail(B);
        __builtin_amdgcn_sched_barrier(0);
        request(2);   // (two slots at a time, two products ahead of their use: at most four slots' values -- 32 registers -- in flight)
        request(3);
        __builtin_amdgcn_sched_barrier(0);
        // Eight mirror pairs per thread.  Thread k0: slot r of A (bin kA + K br3(r)) with slot 7 - r of B, taken from the pair's lower
        // bin (w16 goes up to M / 2): kA + m K for m = br3(r) < 4, kB + (7 - m) K else; w16 at those bins from ONE table entry per
        // group (bin k0 + m K is exp(-i pi m / 8) further round the circle: gs K = FFT_MAX / 8).  Thread 0 owns the groups that are
        // their own mirrors -- group 0: bins m K, mirror (8 - m) K, m = 0 and 4 alone; group K / 2: slot r with slot 7 - r -- and
        // runs them through the SAME eight slots (operands and results dealt by selects: a branch of its own made wave 0, and
        // every wave at the barrier behind it, take twice the time), plus one ninth product for the bin that is left (M / 2).
        const double2 c16_1 = make_double2(0.92387953251128675613, -0.38268343236508977173);   // exp(-i pi / 8)
        const double2 c16_3 = make_double2(0.38268343236508977173, -0.92387953251128675613);   // exp(-3 i pi / 8)
        auto sel = [&](const double2 x, const double2 y) { return make_double2(sp ? x.x : y.x, sp ? x.y : y.y); };
        // In place, slot by slot (a slot reads and writes the same two elements, for either kind of thread; the selects keep the
        // element of the other kind).  General: slot r = (A[r], B[7 - r]), X the lower bin's.  Thread 0: m = br3(r) < 4: group 0's pair
        // (A[r], A[br3(8 - m)]), bin 0 alone in slot 0 (its partner a copy); m >= 4: group K / 2's pair m' = 7 - m: (B[br3(m')],
        // B[7 - br3(m')]).  Bin M / 2 (A[1]) is the one left over for thread 0: its own partner.
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int m = br3(r);
            if (r == 2 || r == 4) {
                __builtin_amdgcn_sched_barrier(0);
                request(r + 2);
                request(r + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            const double2 wb = (m < 4) ? wkA : wkB;
            const int mm = (m < 4) ? m : 7 - m;
            const double2 wk = (mm == 0) ? wb : (mm == 1 ? cmulx(wb, c16_1) : (mm == 2 ? mul_w8_1(wb) : cmulx(wb, c16_3)));
            if (m < 4) {
                const int s1 = (m == 0) ? 0 : br3(8 - m);          // thread 0's partner slot in A (slot 0: a copy of itself)
                double2 x = A[r], y = sel(A[s1], B[7 - r]);
                conv_pair_mul(x, y, Gk[r], Gm[r], wk);
                A[r] = x;
                B[7 - r] = sel(B[7 - r], y);
                if (m != 0) A[s1] = sel(y, A[s1]);
            } else {
                // thread 0: group K / 2's pair m' = 7 - m sits in (B[br3(m')], B[7 - br3(m')]) = (B[7 - r], B[r]): the lower bin's
                // element is the same register as for the others, the partner B[r] instead of A[r]
                double2 x = B[7 - r], y = sel(B[r], A[r]);
                conv_pair_mul(x, y, Gk[r], Gm[r], wk);
                B[7 - r] = x;
                A[r] = sel(A[r], y);
                B[r] = sel(y, B[r]);
            }
        }
        if (sp) {
            double2 c = A[1];
            conv_pair_mul(A[1], c, Gh, Gh, wh);
        }
        dit8_head(A);
        dit8_head(B);
#pragma unroll
        for (int c = 0; c < 8; c++) { pa[c] = A[c]; pb[c] = B[c]; }
    }
    lds_barrier();
}

template <int LOG2M, int NT>
__device__ __noinline__ void conv_mid(const double2* __restrict__ G, const double2* __restrict__ w16)
{
    const ConvMidPre pre = conv_mid_request<LOG2M, NT>(G, w16);
    conv_mid_impl<LOG2M, NT>(G, w16, pre);
}
// ---- the same pass for a plain complex convolution: last three forward stages, product with a spectrum given in NATURAL bin order
// (Bn[k], k < M), first three inverse stages.  After dif8_tail slot r of a thread's group k0 holds bin k0 + (M / 8) br3(r).
// BR: the spectrum table is in bit-reversed order (the tables of the block-wide transforms of fft_device.h): bin k0 + (M / 8) br3(r)
// then sits at 8 bitrev(k0) + r -- a thread's eight values are one contiguous 128-byte run.
template <int LOG2M, int NT, bool BR = false>
__device__ __noinline__ void conv_mid_plain(const double2* __restrict__ Bn)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    constexpr int M = 1 << LOG2M, NA = M / 16, K = M / 8, LW = LOG2M - 10;
    const int t = threadIdx.x;
    lds_barrier();
    if (NA == NT || t < NA) {
        const int kA = (t == 0) ? 0 : t, kB = (t == 0) ? K / 2 : K - t;
        const int wA = (LW == 3) ? br3(kA & 7) : (LW == 2 ? (((kA & 1) << 1) | ((kA >> 1) & 1)) : (kA & 1));
        const int wB = (LW == 3) ? br3(kB & 7) : (LW == 2 ? (((kB & 1) << 1) | ((kB >> 1) & 1)) : (kB & 1));
        double2* pa = z + wA * 1025 + 8 * (kA >> LW);
        double2* pb = z + wB * 1025 + 8 * (kB >> LW);
        double2 A[8], B[8], GA[8], GB[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (BR) {
                GA[r] = gload(&Bn[8 * (int)(__brev((unsigned)kA) >> (32 - (LOG2M - 3))) + r]);
                GB[r] = gload(&Bn[8 * (int)(__brev((unsigned)kB) >> (32 - (LOG2M - 3))) + r]);
            } else {
                GA[r] = gload(&Bn[kA + K * br3(r)]);
                GB[r] = gload(&Bn[kB + K * br3(r)]);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) { A[c] = pa[c]; B[c] = pb[c]; }
        dif8_tail(A);
        dif8_tail(B);
#pragma unroll
        for (int r = 0; r < 8; r++) { A[r] = cmulx(A[r], GA[r]); B[r] = cmulx(B[r], GB[r]); }
        dit8_head(A);
        dit8_head(B);
#pragma unroll
        for (int c = 0; c < 8; c++) { pa[c] = A[c]; pb[c] = B[c]; }
    }
    lds_barrier();
}

// ---- the rest of the inverse transform: natural order out, ends with a block barrier -------------------------------------------------
template <int LOG2M, int NT>
__device__ __noinline__ void conv_inv(const double2* __restrict__ tw, const double2* __restrict__ cft)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2* z = (double2*)smem;
    constexpr int M = 1 << LOG2M, NA = M / 16, LW = LOG2M - 10, NB = 1 << LW;
    const int t = threadIdx.x;
    if (NA == NT || t < NA) {
        double2* zb = z + (t >> 6) * 1025;
        conv_p3_inv(zb, cft, t & 63);
        conv_p2_inv(zb, cft, t & 63);
    }
    lds_barrier();
    if (NA == NT || t < NA) {
#pragma unroll
        for (int g = 0; g < 16 / NB; g++) {
            const int i0 = t + NA * g;
            double2 a[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) a[j] = z[j * 1025 + i0];
#pragma unroll
            for (int e = LW - 1; e >= 0; e--) {
                const int half = NB >> (e + 1);
#pragma unroll
                for (int b = 0; b < NB; b += 2 * half)
#pragma unroll
                    for (int j = 0; j < half; j++) dit_bf(a[b + j], a[b + j + half], gload(&tw[(i0 + 1024 * j) * ((8 >> LW) << e)]));
            }
#pragma unroll
            for (int j = 0; j < NB; j++) z[j * 1025 + i0] = a[j];
        }
    }
    lds_barrier();
}

#pragma clang fp contract(fast)   // (what -ffp-contract=fast-honor-pragmas gives the rest of the translation unit)

}  // namespace nrhip
