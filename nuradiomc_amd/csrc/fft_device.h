// fft_device.h -- block-cooperative complex128 FFTs and chirp-z transforms held entirely in LDS.
//
// MI355X (gfx950): 160 KB LDS per CU holds one 8192-point complex128 working set (128 KB), so a whole
// arbitrary-length DFT (Bluestein: chirp multiply -> FFT -> pointwise multiply -> inverse FFT -> chirp
// multiply) runs without touching HBM for intermediates.  The event-dependent trace length L of
// efieldToVoltageConverter (NuRadioReco/modules/efieldToVoltageConverter.py:147-169, e.g. 5296 =
// 2^4 * 331) is therefore never an FFT "plan" parameter: every L uses the same power-of-two kernels and
// only a per-L chirp table (built on device, fft tables.hip) differs.
//
// Radix-2 decimation-in-frequency forward (natural -> bit-reversed) paired with decimation-in-time
// inverse (bit-reversed -> natural): the pointwise product of a convolution happens in bit-reversed
// order on both operands, so no reordering pass exists.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

constexpr int FFT_LOG2_MAX = 13;
constexpr int FFT_MAX = 1 << FFT_LOG2_MAX;  // 8192 complex128 = 128 KB of LDS

__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ inline double2 cscale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }

// ---- K radix-2 stages fused in registers (2^K points per thread and pass) --------------------------------------------
// The butterflies, their order and their rounding are exactly those of K separate radix-2 stages; only the number of
// LDS round trips and barriers drops (13 stages: 4 passes instead of 13).  With one 128 KB working set per CU there is a
// single block to hide LDS latency with, so passes -- not flops -- are what an in-LDS FFT costs here.
#ifndef NRHIP_FFT_FUSE
#define NRHIP_FFT_FUSE 4
#endif

// Padded LDS layout of the 8192-point convolution buffer: element i lives at i + (i >> 5) + (i >> 7).  With 16-byte elements on
// 32 four-byte banks, the strides the passes use between the lanes of a wave -- 32 elements in the third fused pass (and its mirror
// in the inverse transform), 128 elements in the bit-reversed accesses of the spectrum product -- land on the same banks in the
// plain layout (4- and 64-way conflicts); the two shifts move them to 33.25 and 133 elements, which visit all banks.
__host__ __device__ __forceinline__ int fft_pad(int i) { return i + (i >> 5) + (i >> 7); }
template <bool PAD> __device__ __forceinline__ int fft_at(int i) { return PAD ? fft_pad(i) : i; }
constexpr int FFT_PADDED_MAX = FFT_MAX + FFT_MAX / 32 + FFT_MAX / 128;   // elements of a padded FFT_MAX-point buffer

// decimation in frequency, stages s .. s + K - 1 (spans M >> (s + 1) .. M >> (s + K))
template <int K, bool PAD = false>
__device__ inline void fft_dif_pass(double2* x, int M, int s, const double2* __restrict__ tw, bool inverse)
{
    constexpr int R = 1 << K;
    const int q = M >> (s + K);
    for (int t = threadIdx.x; t < (M >> K); t += blockDim.x) {
        const int pos = t & (q - 1);
        const int i0 = ((t - pos) << K) + pos;
        double2 a[R];
#pragma unroll
        for (int j = 0; j < R; j++) a[j] = x[fft_at<PAD>(i0 + j * q)];
#pragma unroll
        for (int e = 0; e < K; e++) {
            const int half = R >> (e + 1);
            const int tstride = (FFT_MAX / 2) / (half * q);
#pragma unroll
            for (int j = 0; j < half; j++) {
                double2 w = tw[(pos + j * q) * tstride];
                if (inverse) w.y = -w.y;
#pragma unroll
                for (int blk = 0; blk < R; blk += 2 * half) {
                    const double2 u = a[blk + j], v = a[blk + j + half];
                    a[blk + j] = cadd(u, v);
                    a[blk + j + half] = cmul(csub(u, v), w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) x[fft_at<PAD>(i0 + j * q)] = a[j];
    }
    __syncthreads();
}

// decimation in time, stages with spans q = 1 << s, 2q, .. (K of them)
template <int K, bool PAD = false>
__device__ inline void fft_dit_pass(double2* x, int M, int s, const double2* __restrict__ tw, bool inverse)
{
    constexpr int R = 1 << K;
    const int q = 1 << s;
    for (int t = threadIdx.x; t < (M >> K); t += blockDim.x) {
        const int pos = t & (q - 1);
        const int i0 = ((t - pos) << K) + pos;
        double2 a[R];
#pragma unroll
        for (int j = 0; j < R; j++) a[j] = x[fft_at<PAD>(i0 + j * q)];
#pragma unroll
        for (int e = 0; e < K; e++) {
            const int half = 1 << e;
            const int tstride = (FFT_MAX / 2) / (half * q);
#pragma unroll
            for (int j = 0; j < half; j++) {
                double2 w = tw[(pos + j * q) * tstride];
                if (inverse) w.y = -w.y;
#pragma unroll
                for (int blk = 0; blk < R; blk += 2 * half) {
                    const double2 u = a[blk + j], b = cmul(a[blk + j + half], w);
                    a[blk + j] = cadd(u, b);
                    a[blk + j + half] = csub(u, b);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) x[fft_at<PAD>(i0 + j * q)] = a[j];
    }
    __syncthreads();
}

template <int F, bool PAD = false>
__device__ inline void fft_dif_fused_k(double2* x, int log2m, const double2* __restrict__ tw, bool inverse)
{
    const int M = 1 << log2m;
    int s = 0;
    for (; log2m - s >= F; s += F) fft_dif_pass<F, PAD>(x, M, s, tw, inverse);
    const int rem = log2m - s;
    if (rem == 3) fft_dif_pass<3, PAD>(x, M, s, tw, inverse);
    else if (rem == 2) fft_dif_pass<2, PAD>(x, M, s, tw, inverse);
    else if (rem == 1) fft_dif_pass<1, PAD>(x, M, s, tw, inverse);
}

// 2^s0 independent transforms of 2^(log2m - s0) points each, stored one after the other in x: the stages s0 .. log2m - 1 of a
// 2^log2m-point decimation-in-frequency transform are exactly that (stage s works inside blocks of 2^(log2m - s) points), so a
// batch of short transforms keeps every thread of the block busy in each pass.  Each block ends bit-reversed in place.
template <int F, bool PAD = false>
__device__ inline void fft_dif_batched(double2* x, int log2m, int s0, const double2* __restrict__ tw, bool inverse)
{
    const int M = 1 << log2m;
    int s = s0;
    for (; log2m - s >= F; s += F) fft_dif_pass<F, PAD>(x, M, s, tw, inverse);
    const int rem = log2m - s;
    if (rem == 3) fft_dif_pass<3, PAD>(x, M, s, tw, inverse);
    else if (rem == 2) fft_dif_pass<2, PAD>(x, M, s, tw, inverse);
    else if (rem == 1) fft_dif_pass<1, PAD>(x, M, s, tw, inverse);
}

template <int F, bool PAD = false>
__device__ inline void fft_dit_fused_k(double2* x, int log2m, const double2* __restrict__ tw, bool inverse)
{
    const int M = 1 << log2m;
    const int rem = log2m % F;
    if (rem == 3) fft_dit_pass<3, PAD>(x, M, 0, tw, inverse);
    else if (rem == 2) fft_dit_pass<2, PAD>(x, M, 0, tw, inverse);
    else if (rem == 1) fft_dit_pass<1, PAD>(x, M, 0, tw, inverse);
    for (int s = rem; s < log2m; s += F) fft_dit_pass<F, PAD>(x, M, s, tw, inverse);
}

// tw[k] = exp(-2 pi i k / FFT_MAX), k < FFT_MAX / 2 (global memory, L1/L2 resident, built on the host in
// extended precision).  Twiddle of a sub-size M transform: W_M^p = tw[p * (FFT_MAX / M)].
// Decimation in frequency: natural-order input, bit-reversed output.  inverse -> conjugate twiddles (no 1/M).
__device__ inline void fft_dif_pairs(double2* x, int log2m, const double2* __restrict__ tw, bool inverse)
{
    const int M = 1 << log2m;
    int s = 0;
    // two radix-2 stages fused in registers (same operations, same rounding as two separate stages): half the LDS
    // traffic and half the barriers
    for (; s + 1 < log2m; s += 2) {
        const int q = M >> (s + 2);             // span of the second stage; the first stage has span 2q
        const int ts1 = (FFT_MAX / 2) / (2 * q), ts2 = (FFT_MAX / 2) / q;
        for (int t = threadIdx.x; t < M / 4; t += blockDim.x) {
            int pos = t & (q - 1);
            int i0 = ((t - pos) << 2) + pos, i1 = i0 + q, i2 = i1 + q, i3 = i2 + q;
            double2 a0 = x[i0], a1 = x[i1], a2 = x[i2], a3 = x[i3];
            double2 wa = tw[pos * ts1], wb = tw[(pos + q) * ts1], wc = tw[pos * ts2];
            if (inverse) { wa.y = -wa.y; wb.y = -wb.y; wc.y = -wc.y; }
            double2 b0 = cadd(a0, a2), b2 = cmul(csub(a0, a2), wa);
            double2 b1 = cadd(a1, a3), b3 = cmul(csub(a1, a3), wb);
            x[i0] = cadd(b0, b1);
            x[i1] = cmul(csub(b0, b1), wc);
            x[i2] = cadd(b2, b3);
            x[i3] = cmul(csub(b2, b3), wc);
        }
        __syncthreads();
    }
    for (; s < log2m; s++) {
        const int span = M >> (s + 1);
        const int tstride = (FFT_MAX / 2) / span;
        for (int t = threadIdx.x; t < M / 2; t += blockDim.x) {
            int pos = t & (span - 1);
            int i0 = ((t - pos) << 1) + pos, i1 = i0 + span;
            double2 a = x[i0], b = x[i1];
            double2 w = tw[pos * tstride];
            if (inverse) w.y = -w.y;
            x[i0] = cadd(a, b);
            x[i1] = cmul(csub(a, b), w);
        }
        __syncthreads();
    }
}

// Decimation in time: bit-reversed input, natural-order output.
__device__ inline void fft_dit_pairs(double2* x, int log2m, const double2* __restrict__ tw, bool inverse)
{
    const int M = 1 << log2m;
    int s = 0;
    if (log2m & 1) {  // odd number of stages: one plain radix-2 stage first (span 1)
        for (int t = threadIdx.x; t < M / 2; t += blockDim.x) {
            int i0 = t << 1, i1 = i0 + 1;
            double2 w = tw[0];
            if (inverse) w.y = -w.y;
            double2 a = x[i0], b = cmul(x[i1], w);
            x[i0] = cadd(a, b);
            x[i1] = csub(a, b);
        }
        __syncthreads();
        s = 1;
    }
    for (; s + 1 < log2m; s += 2) {  // two radix-2 stages (spans q and 2q) fused in registers
        const int q = 1 << s;
        const int ts1 = (FFT_MAX / 2) / q, ts2 = (FFT_MAX / 2) / (2 * q);
        for (int t = threadIdx.x; t < M / 4; t += blockDim.x) {
            int pos = t & (q - 1);
            int i0 = ((t - pos) << 2) + pos, i1 = i0 + q, i2 = i1 + q, i3 = i2 + q;
            double2 wa = tw[pos * ts1], wb = tw[pos * ts2], wc = tw[(pos + q) * ts2];
            if (inverse) { wa.y = -wa.y; wb.y = -wb.y; wc.y = -wc.y; }
            double2 a0 = x[i0], a1 = cmul(x[i1], wa), a2 = x[i2], a3 = cmul(x[i3], wa);
            double2 b0 = cadd(a0, a1), b1 = csub(a0, a1), b2 = cadd(a2, a3), b3 = csub(a2, a3);
            double2 c2 = cmul(b2, wb), c3 = cmul(b3, wc);
            x[i0] = cadd(b0, c2);
            x[i2] = csub(b0, c2);
            x[i1] = cadd(b1, c3);
            x[i3] = csub(b1, c3);
        }
        __syncthreads();
    }
}

// Compile-time-sized variants (M = 2^LOG2M points, NT threads per block): all quads of a pass are loaded
// (LDS data + twiddles from global) before any is computed, so the ~2 x 7 independent loads per thread overlap
// instead of serialising on L2 latency.  Same operations and rounding as fft_dif / fft_dit.
template <int LOG2M, int NT>
__device__ inline void fft_dif_t_pairs(double2* x, const double2* __restrict__ tw, bool inverse)
{
    constexpr int M = 1 << LOG2M;
    constexpr int QT = (M / 4 + NT - 1) / NT;  // quads per thread and pass
    int s = 0;
#pragma unroll 1
    for (; s + 1 < LOG2M; s += 2) {
        const int q = M >> (s + 2);
        const int ts1 = (FFT_MAX / 2) / (2 * q), ts2 = (FFT_MAX / 2) / q;
        double2 a[QT][4], w[QT][3];
        int idx[QT];
#pragma unroll
        for (int u = 0; u < QT; u++) {
            int t = threadIdx.x + u * NT;
            int pos = t & (q - 1);
            int i0 = ((t - pos) << 2) + pos;
            idx[u] = (M / 4 % NT == 0 || t < M / 4) ? i0 : -1;
            if (idx[u] >= 0) {
                w[u][0] = tw[pos * ts1]; w[u][1] = tw[(pos + q) * ts1]; w[u][2] = tw[pos * ts2];
                a[u][0] = x[i0]; a[u][1] = x[i0 + q]; a[u][2] = x[i0 + 2 * q]; a[u][3] = x[i0 + 3 * q];
            }
        }
#pragma unroll
        for (int u = 0; u < QT; u++) {
            if (idx[u] < 0) continue;
            double2 wa = w[u][0], wb = w[u][1], wc = w[u][2];
            if (inverse) { wa.y = -wa.y; wb.y = -wb.y; wc.y = -wc.y; }
            double2 b0 = cadd(a[u][0], a[u][2]), b2 = cmul(csub(a[u][0], a[u][2]), wa);
            double2 b1 = cadd(a[u][1], a[u][3]), b3 = cmul(csub(a[u][1], a[u][3]), wb);
            int i0 = idx[u];
            x[i0] = cadd(b0, b1);
            x[i0 + q] = cmul(csub(b0, b1), wc);
            x[i0 + 2 * q] = cadd(b2, b3);
            x[i0 + 3 * q] = cmul(csub(b2, b3), wc);
        }
        __syncthreads();
    }
    if (LOG2M & 1) {  // last single stage, span 1 (twiddle 1)
        constexpr int PT = (M / 2 + NT - 1) / NT;
#pragma unroll
        for (int u = 0; u < PT; u++) {
            int t = threadIdx.x + u * NT;
            if (M / 2 % NT == 0 || t < M / 2) {
                double2 a0 = x[2 * t], a1 = x[2 * t + 1];
                double2 w0 = tw[0];
                if (inverse) w0.y = -w0.y;
                x[2 * t] = cadd(a0, a1);
                x[2 * t + 1] = cmul(csub(a0, a1), w0);
            }
        }
        __syncthreads();
    }
}

template <int LOG2M, int NT>
__device__ inline void fft_dit_t_pairs(double2* x, const double2* __restrict__ tw, bool inverse)
{
    constexpr int M = 1 << LOG2M;
    constexpr int QT = (M / 4 + NT - 1) / NT;
    int s = 0;
    if (LOG2M & 1) {
        constexpr int PT = (M / 2 + NT - 1) / NT;
#pragma unroll
        for (int u = 0; u < PT; u++) {
            int t = threadIdx.x + u * NT;
            if (M / 2 % NT == 0 || t < M / 2) {
                double2 w0 = tw[0];
                if (inverse) w0.y = -w0.y;
                double2 a0 = x[2 * t], b = cmul(x[2 * t + 1], w0);
                x[2 * t] = cadd(a0, b);
                x[2 * t + 1] = csub(a0, b);
            }
        }
        __syncthreads();
        s = 1;
    }
#pragma unroll 1
    for (; s + 1 < LOG2M; s += 2) {
        const int q = 1 << s;
        const int ts1 = (FFT_MAX / 2) / q, ts2 = (FFT_MAX / 2) / (2 * q);
        double2 a[QT][4], w[QT][3];
        int idx[QT];
#pragma unroll
        for (int u = 0; u < QT; u++) {
            int t = threadIdx.x + u * NT;
            int pos = t & (q - 1);
            int i0 = ((t - pos) << 2) + pos;
            idx[u] = (M / 4 % NT == 0 || t < M / 4) ? i0 : -1;
            if (idx[u] >= 0) {
                w[u][0] = tw[pos * ts1]; w[u][1] = tw[pos * ts2]; w[u][2] = tw[(pos + q) * ts2];
                a[u][0] = x[i0]; a[u][1] = x[i0 + q]; a[u][2] = x[i0 + 2 * q]; a[u][3] = x[i0 + 3 * q];
            }
        }
#pragma unroll
        for (int u = 0; u < QT; u++) {
            if (idx[u] < 0) continue;
            double2 wa = w[u][0], wb = w[u][1], wc = w[u][2];
            if (inverse) { wa.y = -wa.y; wb.y = -wb.y; wc.y = -wc.y; }
            double2 a0 = a[u][0], a1 = cmul(a[u][1], wa), a2 = a[u][2], a3 = cmul(a[u][3], wa);
            double2 b0 = cadd(a0, a1), b1 = csub(a0, a1), b2 = cadd(a2, a3), b3 = csub(a2, a3);
            double2 c2 = cmul(b2, wb), c3 = cmul(b3, wc);
            int i0 = idx[u];
            x[i0] = cadd(b0, c2);
            x[i0 + 2 * q] = csub(b0, c2);
            x[i0 + q] = cadd(b1, c3);
            x[i0 + 3 * q] = csub(b1, c3);
        }
        __syncthreads();
    }
}

// Entry points.  Run-time sized transforms (the N / 2-point ray transforms, <= 4096 points) keep the pairwise passes: 2^4
// points per thread would leave most of a block idle there and its 64 extra VGPRs lower the occupancy of the kernels that
// only ever call these.  The compile-time sized ones (the 8192-point convolutions) fuse NRHIP_FFT_FUSE stages.
__device__ inline void fft_dif(double2* x, int log2m, const double2* __restrict__ tw, bool inverse) { fft_dif_pairs(x, log2m, tw, inverse); }
__device__ inline void fft_dit(double2* x, int log2m, const double2* __restrict__ tw, bool inverse) { fft_dit_pairs(x, log2m, tw, inverse); }
#if NRHIP_FFT_FUSE > 2
template <int LOG2M, int NT, bool PAD = false>
__device__ inline void fft_dif_t(double2* x, const double2* __restrict__ tw, bool inverse) { fft_dif_fused_k<NRHIP_FFT_FUSE, PAD>(x, LOG2M, tw, inverse); }
template <int LOG2M, int NT, bool PAD = false>
__device__ inline void fft_dit_t(double2* x, const double2* __restrict__ tw, bool inverse) { fft_dit_fused_k<NRHIP_FFT_FUSE, PAD>(x, LOG2M, tw, inverse); }
#else
template <int LOG2M, int NT>
__device__ inline void fft_dif_t(double2* x, const double2* __restrict__ tw, bool inverse) { fft_dif_t_pairs<LOG2M, NT>(x, tw, inverse); }
template <int LOG2M, int NT>
__device__ inline void fft_dit_t(double2* x, const double2* __restrict__ tw, bool inverse) { fft_dit_t_pairs<LOG2M, NT>(x, tw, inverse); }
#endif

// 8192-point convolution with a bit-reversed spectrum table, NT threads
template <int NT>
__device__ inline void czt_convolve_t(double2* x, const double2* __restrict__ Btab, const double2* __restrict__ tw)
{
    constexpr int M = FFT_MAX;
    fft_dif_t<FFT_LOG2_MAX, NT>(x, tw, false);
#pragma unroll
    for (int u = 0; u < M / NT; u++) {
        int i = threadIdx.x + u * NT;
        x[i] = cmul(x[i], Btab[i]);
    }
    __syncthreads();
    fft_dit_t<FFT_LOG2_MAX, NT>(x, tw, true);
}

__device__ inline int bitrev(int i, int log2m) { return (int)(__brev((unsigned)i) >> (32 - log2m)); }

// exp(sgn * i pi n^2 / Q) with the phase reduced exactly in integers (n^2 mod 2Q) before sincospi
// x mod m for integers 0 <= x < 2^52, 0 < m < 2^31 held in doubles: exact (the quotient estimate is off by at most one, every
// product and difference below is an integer under 2^53) -- the 64-bit integer remainder it replaces is ~150 instructions on the GPU
__device__ __forceinline__ double mod_exact(double x, double m)
{
    double q = floor(x / m);
    double r = fma(-q, m, x);
    if (r < 0.) r += m;
    if (r >= m) r -= m;
    return r;
}
__device__ inline double2 chirp(long long n, long long Q, double sgn)
{
    double s, c;
    if (n < (1ll << 26) && n > -(1ll << 26) && Q < (1ll << 30)) {   // n^2 < 2^52: the same integer remainder, in doubles
        const double nd = (double)n;
        sincospi(mod_exact(nd * nd, 2. * (double)Q) / (double)Q, &s, &c);
    } else {
        long long r = (n * n) % (2 * Q);
        sincospi((double)r / (double)Q, &s, &c);
    }
    return make_double2(c, sgn * s);
}

// Bluestein kernel table for out[k] = sum_{j < n_in} in[j] exp(sgn 2 pi i j k / Q'), k < n_out, where the
// exponent is written sgn * i pi (2 j k) / Q with Q = Q' :  2jk = j^2 + k^2 - (k - j)^2.
// b[n] = conj(chirp(n)) for n in [-(n_in - 1), n_out - 1] (index n mod M), zero elsewhere; returns FFT(b) in
// bit-reversed order in x (LDS).  Needs M >= n_in + n_out - 1.
__device__ inline void czt_build_table(double2* x, int log2m, int n_in, int n_out, long long Q, double sgn,
                                       const double2* __restrict__ tw)
{
    const int M = 1 << log2m;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        double2 v = make_double2(0., 0.);
        if (i < n_out) v = cconj(chirp(i, Q, sgn));
        else if (i > M - n_in) v = cconj(chirp((long long)(M - i), Q, sgn));
        x[i] = v;
    }
    __syncthreads();
    fft_dif(x, log2m, tw, false);
}

// x holds in[j] * chirp(j) for j < n_in and zeros up to M (caller fills, then __syncthreads()).
// On return x[k] = (1 / M-scaled) convolution; the caller multiplies by chirp(k) and 1/M when reading.
__device__ inline void czt_convolve(double2* x, int log2m, const double2* __restrict__ Btab,
                                    const double2* __restrict__ tw)
{
    const int M = 1 << log2m;
    fft_dif(x, log2m, tw, false);
    for (int i = threadIdx.x; i < M; i += blockDim.x) x[i] = cmul(x[i], Btab[i]);
    __syncthreads();
    fft_dit(x, log2m, tw, true);
}

// ---- transforms of length nh = N / 2 for ANY even trace length N (a station constant) -------------------------------------------
// The reference only asks for an even number of samples (NuRadioReco/framework/base_trace.py:117-121); its own example runs
// N = 1280.  nh a power of two: the radix-2 code above (result in bit-reversed order).  Otherwise Bluestein's algorithm on the
// next power of two P >= 2 nh - 1 with the two spectrum tables of the chirp kernel built once per station (result in natural
// order).  x must have room for max(nh, P) complex numbers.
struct NPlan {
    int nh, log2nh, log2p;   // log2nh >= 0: power of two; else log2p = log2 P
    int radix = 1, log2m = 0;   // radix > 1 (3, 5, 7): nh = radix * 2^log2m, one radix pass + radix transforms of 2^log2m points --
                                // for N / 2 above FFT_MAX / 2, where Bluestein's 2 nh - 1 points no longer fit the LDS (N = 10 240)
    const double2* wN;       // [nh + 1] exp(-2 pi i k / N)
    const double2* cw;       // [nh]     exp(-i pi n^2 / nh)
    const double2* Bf;       // [P] spectrum (bit-reversed) of the forward chirp kernel
    const double2* Bi;       // [P] of the inverse one
};
__host__ __device__ inline int nplan_points(const NPlan& p) { return (p.log2nh >= 0 || p.radix > 1) ? p.nh : (1 << p.log2p); }
// exp(-2 pi i k / N), k <= nh
__device__ __forceinline__ double2 nplan_w(const NPlan& p, int k, const double2* __restrict__ tw)
{
    if (p.log2nh >= 0) return (k == p.nh) ? make_double2(-1., 0.) : tw[k * (FFT_MAX / (2 * p.nh))];
    return p.wN[k];
}
// where element j of the transform sits after nplan_fft
__device__ __forceinline__ int nplan_idx(const NPlan& p, int j)
{
    if (p.log2nh >= 0) return bitrev(j, p.log2nh);
    if (p.radix > 1) return (j % p.radix) * (p.nh / p.radix) + bitrev(j / p.radix, p.log2m);   // block k1 = j mod r, bit-reversed inside
    return j;
}
// in-place DFT (inverse: conjugate kernel, unscaled) of x[0 .. nh) given in natural order; ends with a barrier
__device__ inline void nplan_fft(double2* x, const NPlan& p, const double2* __restrict__ tw, bool inverse)
{
    if (p.log2nh >= 0) {
        fft_dif(x, p.log2nh, tw, inverse);
        return;
    }
    if (p.radix > 1) {
        // decimation in frequency by the odd factor r: X[r k2 + k1] = sum_j2 ((sum_j1 x[j2 + m j1] w_r^(j1 k1)) w_nh^(j2 k1)) w_m^(j2 k2).
        // A thread takes the r inputs of its j2 (stride m), does the r-point DFT and the twiddles and stores y_k1[j2] at k1 m + j2 --
        // the same r places it read --; then r radix-2 transforms of m points, one per block k1 (bit-reversed inside: nplan_idx)
        const int r = p.radix, nh = p.nh, m = nh / r;
        double2 wr[7];
        for (int q = 0; q < r; q++) {
            double sn, cs;
            sincospi(2. * q / r, &sn, &cs);
            wr[q] = make_double2(cs, inverse ? sn : -sn);   // exp(-+ 2 pi i q / r)
        }
        for (int j2 = threadIdx.x; j2 < m; j2 += blockDim.x) {
            double2 in[7], out[7];
            for (int j1 = 0; j1 < r; j1++) in[j1] = x[j2 + m * j1];
            for (int k1 = 0; k1 < r; k1++) {
                double2 a = in[0];
                for (int j1 = 1; j1 < r; j1++) a = cadd(a, cmul(in[j1], wr[(j1 * k1) % r]));
                // w_nh^(j2 k1) = exp(-2 pi i 2 t / N), t = j2 k1 < nh: wN[2 t], or -wN[2 t - nh] beyond the table's half turn
                const int t2 = 2 * (j2 * k1);
                double2 w = (t2 <= nh) ? p.wN[t2] : cscale(p.wN[t2 - nh], -1.);
                if (inverse) w = cconj(w);
                out[k1] = cmul(a, w);
            }
            for (int k1 = 0; k1 < r; k1++) x[k1 * m + j2] = out[k1];
        }
        __syncthreads();
        for (int k1 = 0; k1 < r; k1++) fft_dif(x + k1 * m, p.log2m, tw, inverse);
        return;
    }
    const int P = 1 << p.log2p, nh = p.nh;
    for (int j = threadIdx.x; j < P; j += blockDim.x) {
        double2 v = make_double2(0., 0.);
        if (j < nh) {
            const double2 c = p.cw[j];
            v = cmul(x[j], inverse ? cconj(c) : c);
        }
        x[j] = v;
    }
    __syncthreads();
    fft_dif(x, p.log2p, tw, false);
    const double2* __restrict__ B = inverse ? p.Bi : p.Bf;
    for (int i = threadIdx.x; i < P; i += blockDim.x) x[i] = cmul(x[i], B[i]);
    __syncthreads();
    fft_dit(x, p.log2p, tw, true);
    const double sc = 1.0 / P;
    for (int k = threadIdx.x; k < nh; k += blockDim.x) {
        const double2 c = p.cw[k];
        x[k] = cscale(cmul(x[k], inverse ? cconj(c) : c), sc);
    }
    __syncthreads();
}

}  // namespace nrhip
