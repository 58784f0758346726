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


KIND_RATIONAL, KIND_ABS, KIND_RECTANGULAR = 0, 1, 2


def design(spec):
    """One stage of the filter chain -> (kind, b, a).  spec: (order, (f_lo, f_hi)) = Butterworth, or a dict with the
    reference's get_filter_response arguments: {'type': 'butter' | 'butterabs' | 'cheby1' | 'rectangular', 'passband':
    (f_lo, f_hi), 'order': n, 'rp': ripple [dB]} (signal_processing.py:237-333)."""
    if not isinstance(spec, dict):
        order, pb = spec
        return (KIND_RATIONAL,) + butter_analog(order, pb)
    typ, pb = spec.get('type', spec.get('filter_type', 'butter')), spec['passband']
    if typ == 'rectangular':
        return KIND_RECTANGULAR, np.array([float(pb[0]), float(pb[1])]), np.array([1.0])
    if typ == 'butter':
        return (KIND_RATIONAL,) + butter_analog(spec['order'], pb)
    if typ == 'butterabs':
        return (KIND_ABS,) + butter_analog(spec['order'], pb)
    if typ == 'cheby1':
        return (KIND_RATIONAL,) + cheby1_analog(spec['order'], spec['rp'], pb)
    raise NotImplementedError("filter type {} is not provided (butter, butterabs, cheby1, rectangular)".format(typ))


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
