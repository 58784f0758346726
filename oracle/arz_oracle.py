"""ctypes front-end of oracle/arz_oracle.c plus the host logic of the reference's ARZ class (TEST INFRASTRUCTURE ONLY,
never imported by nuradiomc_amd).

Parity status: PINNED against outputs of the reference's NuRadioMC/SignalGen/ARZ/ARZ.py and askaryan.py run in the build
container (tests/golden/ref_arz.npz, generator tests/golden/gen/gen_arz.py; tests/test_oracle_golden.py).
"""
import ctypes
import numpy as np
from . import raytrace_oracle as _rto

_dp = ctypes.POINTER(ctypes.c_double)
V_S = 1e9   # units.V * units.s in NuRadioReco's base units (ns, m, eV, e+)
# ARZ.__set_model_parameters (ARZ.py:394-434): (Af, freq_pos, freq_neg, exp_pos, exp_neg, t0_pos, t0_neg), em_fraction used?
MODEL_PARAMETERS = {
    'ARZ2019': {'EM': (-4.5e-14 * 0.88 * V_S, 2.87, 3.05, -3.00, -3.50, 0.057, 0.030),
                'HAD': (-3.2e-14 * V_S, 3.00, 2.92, -2.65, -3.21, 0.065, 0.043), 'em_factor': False},
    'ARZ2020': {'EM': (-4.445e-14 * V_S, 2.298, 2.616, -3.588, -4.043, 0.0348, 0.0203),
                'HAD': (-4.071e-14 * V_S, 2.338, 2.686, -3.320, -3.687, 0.0391, 0.0234), 'em_factor': True}}


def _lib():
    L = _rto.lib()
    if not getattr(L, '_arz_ready', False):
        L.orc_arz_vector_potential.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                               _dp, _dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, _dp]
        L.orc_arz_time_trace.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp,
                                         _dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, _dp]
        L._arz_ready = True
    return L


def vector_potential(shower_energy, theta, N, dt, profile_depth, profile_ce, parameters, shower_type='HAD', n_index=1.78,
                     distance=1., interp_factor=1., interp_factor2=100., shift_for_xmax=False, em_factor=1.):
    """ARZ.get_vector_potential (ARZ.py:36-275): [N + 1, 3]"""
    d = np.ascontiguousarray(profile_depth, float)
    c = np.ascontiguousarray(profile_ce, float)
    p = np.ascontiguousarray(parameters, float)
    vp = np.zeros((N + 1, 3))
    st = _lib().orc_arz_vector_potential(shower_energy, theta, N, dt, len(d), d.ctypes.data_as(_dp), c.ctypes.data_as(_dp),
                                         p.ctypes.data_as(_dp), int(shower_type == 'HAD'), n_index, distance, interp_factor,
                                         interp_factor2, int(shift_for_xmax), em_factor, vp.ctypes.data_as(_dp))
    if st != 0:
        raise NotImplementedError("length of indices is not 2 nor 4")
    return vp


def em_fraction(energy, version='ARZ2020'):
    """ARZ.em_fraction (:436-447)"""
    if not MODEL_PARAMETERS[version]['em_factor']:
        return 1
    epsilon = np.log10(energy / 1.)
    f_epsilon = -21.98905 - 2.32492 * epsilon
    f_epsilon += 0.019650 * epsilon ** 2 + 13.76152 * np.sqrt(epsilon)
    return f_epsilon


class ARZ:
    """The library handling of the reference's ARZ class (:318-392, :449-596): closest library energy, amplitude rescaled
    by E / E_library, profile number drawn with RandomState.randint unless given / reused."""

    def __init__(self, library, seed=1234, interp_factor=1, interp_factor2=100, arz_version='ARZ2020'):
        self._rng = np.random.RandomState(seed)
        self._library = library
        self._version = arz_version
        self._f1, self._f2 = interp_factor, interp_factor2
        self._random_numbers = {}

    def set_seed(self, seed):
        self._rng.seed(seed)

    def get_last_shower_profile_id(self):
        return self._random_numbers

    def get_time_trace(self, shower_energy, theta, N, dt, shower_type, n_index, R, shift_for_xmax=False, same_shower=False,
                       iN=None, maximum_angle=20 * np.pi / 180):
        energies = np.array([*self._library[shower_type]])
        iE = np.argmin(np.abs(energies - shower_energy))
        rescaling_factor = shower_energy / energies[iE]
        profiles = self._library[shower_type][energies[iE]]
        n_profiles = len(profiles['charge_excess'])
        if iN is None or np.isnan(iN):
            if same_shower and shower_type in self._random_numbers:
                iN = self._random_numbers[shower_type]
            else:
                iN = self._rng.randint(n_profiles)
                self._random_numbers[shower_type] = iN
        else:
            iN = int(iN)
            self._random_numbers[shower_type] = iN
        depth = np.ascontiguousarray(profiles['depth'], float)
        ce = np.ascontiguousarray(np.asarray(profiles['charge_excess'][iN], float) * rescaling_factor)
        if shower_type not in ('HAD', 'EM'):
            raise NotImplementedError("showers of type {} are not implemented. Use 'HAD', 'EM'".format(shower_type))
        par = np.ascontiguousarray(MODEL_PARAMETERS[self._version][shower_type], float)
        emf = em_fraction(shower_energy, self._version) if shower_type == 'HAD' else 1.
        trace = np.zeros((3, N))
        st = _lib().orc_arz_time_trace(shower_energy, theta, N, dt, len(depth), depth.ctypes.data_as(_dp),
                                       ce.ctypes.data_as(_dp), par.ctypes.data_as(_dp), int(shower_type == 'HAD'), n_index, R,
                                       self._f1, self._f2, int(shift_for_xmax), float(emf), maximum_angle,
                                       trace.ctypes.data_as(_dp))
        if st != 0:
            raise NotImplementedError("length of indices is not 2 nor 4")
        return trace


def askaryan_time_trace(arz, energy, theta, N, dt, shower_type, n_index, R, same_shower=False, iN=None):
    """askaryan.get_time_trace for the ARZ models (askaryan.py:118-126): the eTheta component and the profile number"""
    trace = arz.get_time_trace(energy, theta, N, dt, shower_type.upper(), n_index, R, same_shower=same_shower, iN=iN)[1]
    return trace, {'iN': arz.get_last_shower_profile_id()[shower_type.upper()]}


def askaryan_frequency_spectrum(arz, energy, theta, N, dt, shower_type, n_index, R, **kw):
    """askaryan.get_frequency_spectrum (:143-213): fft.time2freq of the trace = rfft / fs * sqrt(2)"""
    tr, add = askaryan_time_trace(arz, energy, theta, N, dt, shower_type, n_index, R, **kw)
    return np.fft.rfft(tr, axis=-1) / (1 / dt) * 2 ** 0.5, add
