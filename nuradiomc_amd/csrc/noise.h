// noise.h -- thermal noise on the channel spectra (NuRadioReco/modules/channelGenericNoiseAdder.py: bandlimited_noise :66-160 with
// type = 'rayleigh', as simulation.apply_det_response calls it, simulation.py:594-606): per frequency bin of the event's L-sample
// trace an amplitude drawn from a Rayleigh distribution and a uniform phase, DC empty, the Nyquist bin real; added to the channel
// spectrum BEFORE the filter chain.  The reference draws from one sequential numpy stream in loop order, which no parallel
// implementation can reproduce; here every (event group, sub-event, channel, bin) has its own Philox4x32-10 counter, so the
// noise of an event does not depend on batching, chunking or the number of GPUs.  The parity tests restate the same generator in
// numpy -- their traces and the kernel's agree to rounding; against the reference the agreement is statistical.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

struct NoiseDev {
    int on;
    unsigned long long seed;
    const double* amplitude;       // [n_ch] "amplitude" argument of bandlimited_noise per channel (0: noiseless channel)
    const long long* group_id;     // [n_groups] or NULL: group_offset + index
    long long group_offset;
    const int* ev_group;           // sub-event -> group (split_event_time_diff) or NULL
    const int* ev_sub;             // sub-event index inside its group or NULL
};

__device__ inline void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// spectrum bin k (0 .. L/2) of the noise of one channel, in the convention of the channel spectra (time2freq): ampl e^{i phi} / fs
__device__ inline double2 noise_bin(const NoiseDev& nz, long long gid, int sub, int ch, int k, int L, double fs)
{
    const int m = L / 2;
    if (k <= 0 || k > m) return make_double2(0., 0.);
    unsigned x[4];
    philox4x32_10((unsigned)gid, (unsigned)((unsigned long long)gid >> 32), ((unsigned)sub << 16) | (unsigned)ch, (unsigned)k,
                  (unsigned)nz.seed, (unsigned)(nz.seed >> 32), x);
    const double u1 = ((double)(x[0] >> 5) * 67108864. + (double)(x[1] >> 6)) * (1. / 9007199254740992.);   // [0, 1), 53 bits
    const double u2 = ((double)(x[2] >> 5) * 67108864. + (double)(x[3] >> 6)) * (1. / 9007199254740992.);
    // fsigma = amplitude * (L / sqrt(n_active)) / sqrt 2 with n_active = L / 2 bins (DC excluded, Nyquist included)
    const double fsigma = nz.amplitude[ch] * ((double)L / sqrt((double)m)) / 1.4142135623730951;
    const double a = fsigma * sqrt(-2. * log(1. - u1)) / fs;
    if (k == m) return make_double2(a, 0.);   // add_random_phases leaves the Nyquist bin of an even-length trace real
    double sn, cs;
    sincospi(2. * u2, &sn, &cs);
    return make_double2(a * cs, a * sn);
}

}  // namespace nrhip
