"""Drop-in for NuRadioMC.SignalGen.askaryan.get_frequency_spectrum / get_time_trace
(NuRadioMC/SignalGen/askaryan.py:10-213) for the frequency-domain parametrisations
(NuRadioMC/SignalGen/parametrizations.py: ZHS1992, Alvarez2000, Alvarez2009), evaluated on the GPU.

The stateful random draw of the Alvarez2009 EM parameter k_L (parametrizations.py:90-91, :160-173) stays on the
host and reproduces the reference's stream: one np.random.RandomState(seed) per model, created at the first call.
"""
import numpy as np
from .context import Context

_random_generators = {}
_Alvarez2009_k_L = None
_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        _ctx = Context((1.78, 0.423, 77.), 'SP1', device=0)  # the ice model is irrelevant for the emission
    return _ctx


def get_parametrizations():
    return ['ZHS1992', 'Alvarez2000', 'Alvarez2009']


def _alvarez2009_kL_distribution(energy):
    log10_E_0 = np.log10(energy / 1.)
    sigma = 3.39e-2 + (0 if log10_E_0 < 14.99 else 2.25e-2) * (log10_E_0 - 14.99)
    mean = 1.52 + (5.59e-2 if log10_E_0 < 16.61 else 0.39) * (log10_E_0 - 16.61)
    return mean, sigma


def get_frequency_spectrum(energy, theta, N, dt, shower_type, n_index, R, model, full_output=False, **kwargs):
    global _Alvarez2009_k_L
    shower_type = shower_type.upper()
    if model not in get_parametrizations():
        raise NotImplementedError("model {} unknown".format(model))
    seed = kwargs.get('seed')
    if model not in _random_generators:
        _random_generators[model] = np.random.RandomState(seed)
    additional = {}
    k_L = kwargs.get('k_L')
    if model == 'Alvarez2009':
        if shower_type not in ('HAD', 'EM'):
            raise NotImplementedError("shower type {} is not implemented in Alvarez2009 model.".format(shower_type))
        if shower_type == 'EM' and k_L is None:
            mean, sigma = _alvarez2009_kL_distribution(energy)
            if kwargs.get('average_shower'):
                k_L = 10 ** mean
            elif kwargs.get('same_shower'):
                if _Alvarez2009_k_L is None:
                    raise AttributeError("the same shower was requested but the function hasn't been called before.")
                k_L = _Alvarez2009_k_L
            else:
                _Alvarez2009_k_L = 10 ** _random_generators[model].normal(mean, sigma)
                k_L = _Alvarez2009_k_L
        if shower_type == 'HAD':
            k_L = 31.25 * (energy / 1.e15) ** 3.01e-2
        additional = {'k_L': k_L}
    elif shower_type not in ('HAD', 'EM'):
        raise NotImplementedError("shower type {} not implemented in {} Askaryan module".format(shower_type, model))
    if energy == 0:
        spec = np.zeros(N // 2 + 1, complex)
    else:
        spec = _context().askaryan_spectrum_batch(energy, theta, N, dt, shower_type, n_index, R, model,
                                                  k_L=1.0 if k_L is None else k_L)[0]
    return (spec, additional) if full_output else spec


def get_time_trace(energy, theta, N, dt, shower_type, n_index, R, model, full_output=False, **kwargs):
    """time domain via the FFT convention of NuRadioReco/utilities/fft.py:92"""
    tmp = get_frequency_spectrum(energy, theta, N, dt, shower_type, n_index, R, model, full_output=full_output, **kwargs)
    spec = tmp[0] if full_output else tmp
    trace = np.fft.irfft(spec, n=N) / dt / 2 ** 0.5
    return (trace, tmp[1]) if full_output else trace
