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


def response(freqs, filters):
    """prod_i polyval(b_i, j f) / polyval(a_i, j f), 0 for f <= 0 (host-side evaluation, e.g. for Vrms)."""
    freqs = np.asarray(freqs, float)
    H = np.ones(freqs.shape, complex)
    mask = freqs > 0
    for b, a in filters:
        h = np.zeros(freqs.shape, complex)
        s = 1j * freqs[mask]
        h[mask] = np.polyval(b, s) / np.polyval(a, s)
        H = H * h
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
