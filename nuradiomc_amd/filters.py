"""Analog Butterworth responses as rational polynomials b(s) / a(s), evaluated at s = j f like
scipy.signal.freqs -- the form NuRadioReco's channelBandPassFilter applies
(NuRadioReco/utilities/signal_processing.py:279-292: signal.butter(order, Wn, analog=True) + signal.freqs).

The design follows the published recipe (Butterworth prototype poles on the unit circle, then the
low-pass -> low-pass / low-pass -> band-pass frequency transformations, then the expansion of the pole
product into polynomial coefficients) with numpy only.
"""
import numpy as np

MAX_POLY = 24


def butter_analog(order, passband):
    """(b, a), highest power first.  passband = (0, f_hi) -> low-pass at f_hi; (f_lo, f_hi) -> band-pass."""
    m = np.arange(-order + 1, order, 2)
    p = -np.exp(1j * np.pi * m / (2 * order))  # prototype poles, unit cutoff; no zeros; gain 1
    k = 1.0
    lo, hi = float(passband[0]), float(passband[1])
    if lo == 0:
        wo = hi
        p = wo * p
        k = k * wo ** order
        z = np.array([])
    else:
        bw = hi - lo
        wo = np.sqrt(lo * hi)
        p_lp = p * bw / 2
        p_lp = p_lp.astype(complex)
        p = np.concatenate((p_lp + np.sqrt(p_lp ** 2 - wo ** 2), p_lp - np.sqrt(p_lp ** 2 - wo ** 2)))
        z = np.zeros(order)
        k = k * bw ** order
    b = k * np.poly(z)
    a = np.real(np.poly(p))
    return np.atleast_1d(np.real(b)), np.atleast_1d(a)


def _transform(p, k, order, passband):
    """low-pass prototype (poles p, gain k, no zeros) -> low-pass at f_hi or band-pass (f_lo, f_hi); (b, a)"""
    lo, hi = float(passband[0]), float(passband[1])
    if lo == 0:
        wo = hi
        p = wo * p
        k = k * wo ** order
        z = np.array([])
    else:
        bw = hi - lo
        wo = np.sqrt(lo * hi)
        p_lp = (p * bw / 2).astype(complex)
        p = np.concatenate((p_lp + np.sqrt(p_lp ** 2 - wo ** 2), p_lp - np.sqrt(p_lp ** 2 - wo ** 2)))
        z = np.zeros(order)
        k = k * bw ** order
    return np.atleast_1d(np.real(k * np.poly(z))), np.atleast_1d(np.real(np.poly(p)))


def cheby1_analog(order, rp, passband):
    """Chebyshev type I (rp dB of pass-band ripple) like scipy.signal.cheby1(order, rp, Wn, analog=True)
    (signal_processing.py:303-309): prototype poles on the ellipse, gain 1 (odd order) or 1 / sqrt(1 + eps^2) (even)."""
    eps = np.sqrt(10 ** (0.1 * rp) - 1.0)
    mu = 1.0 / order * np.arcsinh(1 / eps)
    m = np.arange(-order + 1, order, 2)
    theta = np.pi * m / (2 * order)
    p = -np.sinh(mu + 1j * theta)
    k = np.prod(-p, axis=0).real
    if order % 2 == 0:
        k = k / np.sqrt(1 + eps * eps)
    return _transform(p, k, order, passband)


KIND_RATIONAL, KIND_ABS, KIND_RECTANGULAR, KIND_TABULATED, KIND_GAUSSIAN_TAPERED = 0, 1, 2, 3, 4


def hardware_response(frequencies, gain, phase, temperature=293.15, correction=None):
    """Filter stage for a measured amplifier / signal-chain response (NuRadioReco/detector/RNO_G/analog_components.py:10-104,
    applied by RNO_G/hardwareResponseIncorporator.get_filter(..., sim_to_data=True) :93-135): table of frequencies [GHz], linear
    gain and phase [rad] (unwrapped here with np.unwrap as the reference does), linear interpolation, 0 outside the table.
    correction: None, 'rno_surface' or 'iglu' -- the reference's empirical temperature dependence c0(T) + c1(T) f^5."""
    t = float(temperature) - 273.15
    if correction is None:
        c0, c1 = 1.0, 0.0
    elif correction == 'rno_surface':
        c0, c1 = 1.0377798029 - 0.00135258197 * t, 0.4788208019 - 0.01790064797 * t
    elif correction == 'iglu':
        c0, c1 = 1.1139014286 - 0.00004392995 * (t + 28.8331610295) ** 2, 0.6301058083 - 0.0208741539 * t
    else:
        raise NotImplementedError("temperature correction {} is not provided (rno_surface, iglu)".format(correction))
    return dict(type='tabulated', frequencies=np.asarray(frequencies, float), gain=np.asarray(gain, float),
                phase=np.unwrap(np.asarray(phase, float)), c0=c0, c1=c1)


def design(spec):
    """One stage of the filter chain -> (kind, b, a).  spec: (order, (f_lo, f_hi)) = Butterworth, or a dict with the
    reference's get_filter_response arguments: {'type': 'butter' | 'butterabs' | 'cheby1' | 'rectangular', 'passband':
    (f_lo, f_hi), 'order': n, 'rp': ripple [dB]} (signal_processing.py:237-333)."""
    if not isinstance(spec, dict):
        order, pb = spec
        return (KIND_RATIONAL,) + butter_analog(order, pb)
    typ, pb = spec.get('type', spec.get('filter_type', 'butter')), spec.get('passband')
    if typ == 'rectangular':
        return KIND_RECTANGULAR, np.array([float(pb[0]), float(pb[1])]), np.array([1.0])
    if typ == 'butter':
        return (KIND_RATIONAL,) + butter_analog(spec['order'], pb)
    if typ == 'butterabs':
        return (KIND_ABS,) + butter_analog(spec['order'], pb)
    if typ == 'cheby1':
        return (KIND_RATIONAL,) + cheby1_analog(spec['order'], spec['rp'], pb)
    if typ == 'gaussian_tapered':   # b = (f_lo, f_hi, roll_width); grid dependent (signal_processing.py:310-321)
        return KIND_GAUSSIAN_TAPERED, np.array([float(pb[0]), float(pb[1]), float(spec.get('roll_width', 0.0025))]), np.array([1.0])
    if typ == 'tabulated':          # b = (c0, c1), a = table [n, 3]
        tab = np.stack([np.asarray(spec['frequencies'], float), np.asarray(spec['gain'], float),
                        np.asarray(spec['phase'], float)], axis=1)
        if len(tab) < 2 or np.any(np.diff(tab[:, 0]) <= 0):
            raise ValueError("tabulated response: frequencies must increase strictly")
        return KIND_TABULATED, np.array([float(spec.get('c0', 1.)), float(spec.get('c1', 0.))]), np.ascontiguousarray(tab)
    raise NotImplementedError("filter type {} is not provided (butter, butterabs, cheby1, rectangular, gaussian_tapered, "
                              "tabulated)".format(typ))


def gaussian_tapered(freqs, passband, roll_width):
    """signal_processing.get_filter_response(..., 'gaussian_tapered') :310-321 on the grid freqs (numpy only)"""
    freqs = np.asarray(freqs, float)
    f = np.ones(freqs.shape)
    f[freqs < passband[0]] = 0.
    f[freqs > passband[1]] = 0.
    n = len(freqs)
    std = int(round(roll_width / (freqs[1] - freqs[0])))
    w = np.exp(-0.5 * ((np.arange(n) - (n - 1) / 2.) / std) ** 2)   # signal.windows.gaussian(n, std)
    full = np.convolve(f, w)                                        # mode 'same': the centred n entries of the full product
    c = (n - 1) // 2
    out = full[c:c + n]
    return out / np.max(out)


def tabulated(freqs, c, table):
    """scipy interp1d(kind='linear', bounds_error=False, fill_value=0) of gain and phase, gain * (c0 + c1 f^5)"""
    freqs = np.asarray(freqs, float)
    x, g, ph = table[:, 0], table[:, 1], table[:, 2]
    idx = np.clip(np.searchsorted(x, freqs), 1, len(x) - 1)
    lo, hi = idx - 1, idx
    inside = (freqs >= x[0]) & (freqs <= x[-1])
    gi = np.where(inside, (g[hi] - g[lo]) / (x[hi] - x[lo]) * (freqs - x[lo]) + g[lo], 0.)
    pi = np.where(inside, (ph[hi] - ph[lo]) / (x[hi] - x[lo]) * (freqs - x[lo]) + ph[lo], 0.)
    return (c[0] + c[1] * freqs ** 5) * gi * np.exp(1j * pi)


def response(freqs, filters):
    """Product of the stage responses (host-side evaluation, e.g. for Vrms).  Stages are (b, a) or (kind, b, a):
    rational polyval(b, j f) / polyval(a, j f) (0 for f <= 0), its modulus, or a rectangular pass band."""
    freqs = np.asarray(freqs, float)
    H = np.ones(freqs.shape, complex)
    mask = freqs > 0
    for stage in filters:
        kind, b, a = stage if len(stage) == 3 else (KIND_RATIONAL,) + tuple(stage)
        if kind == KIND_RECTANGULAR:
            H = H * np.where((b[0] <= freqs) & (freqs <= b[1]), 1., 0.)
            continue
        if kind == KIND_GAUSSIAN_TAPERED:
            H = H * gaussian_tapered(freqs, b[:2], b[2])
            continue
        if kind == KIND_TABULATED:
            H = H * tabulated(freqs, b, a)
            continue
        h = np.zeros(freqs.shape, complex)
        s = 1j * freqs[mask]
        h[mask] = np.polyval(b, s) / np.polyval(a, s)
        H = H * (np.abs(h) if kind == KIND_ABS else h)
    return H


def vrms_from_filters(sampling_rate, filters, noise_temperature=300.):
    """Noise RMS the reference derives for a filter chain (NuRadioMC/simulation/simulation.py:1301-1376):
    Vrms = sqrt(T * 50 Ohm * k_B * int |H|^2 df) on a 10000-point grid, and the efield-equivalent
    Vrms / max|H| / 1 m used by the candidate cut."""
    ohm = 1.602176462e-10        # NuRadioReco/utilities/units.py
    k_B = 8.617334187250093e-05  # NuRadioReco/utilities/constants.py
    ff = np.linspace(0, 0.5 * sampling_rate, 10000)
    H = np.abs(response(ff, filters))
    bandwidth = np.sum(0.5 * (H[1:] ** 2 + H[:-1] ** 2) * np.diff(ff))
    vrms = (noise_temperature * (50 * ohm) * bandwidth * k_B) ** 0.5
    return vrms, vrms / H.max() / 1.0


def firwin(numtaps, cutoff, pass_zero=True, fs=2.0):
    """scipy.signal.firwin for ONE cutoff with the Hamming window (numpy only): low pass (pass_zero=True) or high pass (False, odd
    numtaps): windowed ideal response, scaled to unit gain at 0 (low pass) or at Nyquist (high pass).  What
    signal_processing.upsampling_fir :224-226 and PhasedArrayBase.hilbert_envelope :351 design their filters with."""
    numtaps = int(numtaps)
    c = float(cutoff) / (0.5 * fs)
    if not 0. < c < 1.:
        raise ValueError("Invalid cutoff frequency: frequencies must be greater than 0 and less than fs/2.")
    if not pass_zero and numtaps % 2 == 0:
        raise ValueError("A filter with an even number of coefficients must have zero response at the Nyquist frequency.")
    left, right = (0., c) if pass_zero else (c, 1.)
    m = np.arange(0, numtaps) - 0.5 * (numtaps - 1)
    h = right * np.sinc(right * m) - left * np.sinc(left * m)
    # scipy.signal.windows.hamming(numtaps, sym=True) = general_cosine with a = (0.54, 0.46)
    fac = np.linspace(-np.pi, np.pi, numtaps)
    h = h * (0.54 * np.cos(0 * fac) + 0.46 * np.cos(1 * fac)) if numtaps > 1 else h
    scale_frequency = 0. if left == 0 else 1.
    h /= np.sum(h * np.cos(np.pi * m * scale_frequency))
    return h
