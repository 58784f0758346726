"""Drop-in for NuRadioMC.SignalGen.ARZ.ARZ (ARZ.py:317-673): the ARZ2019 / ARZ2020 time-domain Askaryan model with the
vector-potential integral and the trace on the GPU (nrhip_arz_time_trace_batch).

Host logic as in the reference: the shower library ({type: {energy: {'depth', 'charge_excess': [...]}}}, a pickle file
or a dict) is searched for the closest energy, the profile is rescaled by E / E_library, the profile number is drawn with
np.random.RandomState(seed).randint unless `iN` is given or `same_shower` re-uses the previous one.  The reference fetches
library_v1.2.pkl from its data server when it is missing; this module never touches the network: pass `library=`.
`get_time_trace_batch` is the batched form (many showers / rays per call) the GPU wants.
"""
import os
import pickle
import ctypes
import numpy as np
from . import _lib as L

V_S = 1e9   # units.V * units.s in NuRadioReco's base units (ns, m, eV, e+)
RHO = 5.767155003928648e+39   # 0.924 g / cm^3 (ARZ.py:31)
# (Af, freq_pos, freq_neg, exp_pos, exp_neg, t0_pos, t0_neg) -- ARZ.__set_model_parameters (:394-434)
_PARAMETERS = {'ARZ2019': {'HAD': (-3.2e-14 * V_S, 3.00, 2.92, -2.65, -3.21, 0.065, 0.043),
                           'EM': (-4.5e-14 * 0.88 * V_S, 2.87, 3.05, -3.00, -3.50, 0.057, 0.030), 'em': False},
               'ARZ2020': {'HAD': (-4.071e-14 * V_S, 2.338, 2.686, -3.320, -3.687, 0.0391, 0.0234),
                           'EM': (-4.445e-14 * V_S, 2.298, 2.616, -3.588, -4.043, 0.0348, 0.0203), 'em': True}}
_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        from .context import Context
        _ctx = Context((1.78, 0.423, 77.), 'SP1', device=0)  # the ice model is irrelevant for the emission
    return _ctx


def thetaprime_to_theta(thetaprime, xmax, R_prime):
    """ARZ.py:279-296"""
    return np.arctan2(R_prime * np.sin(thetaprime), R_prime * np.cos(thetaprime) + xmax / RHO)


def theta_to_thetaprime(theta, xmax, R):
    """ARZ.py:299-315"""
    return np.arctan2(R * np.sin(theta), R * np.cos(theta) - xmax / RHO)


class ARZ:
    def __init__(self, seed=1234, interp_factor=1, interp_factor2=100, library=None, arz_version='ARZ2020', use_numba=True,
                 ctx=None):
        if arz_version not in _PARAMETERS:
            raise ValueError('ARZ version does not exist. Please choose ARZ2019 or ARZ2020.')
        if library is None:
            raise FileNotFoundError("no shower library given: the reference downloads library_v1.2.pkl from its data server, "
                                    "this module does not use the network -- pass library=<path to the pickle file or dict>")
        if isinstance(library, dict):
            self._library = library
        else:
            if not os.path.exists(library):
                raise FileNotFoundError("user specified shower library {} not found.".format(library))
            with open(library, 'rb') as fin:
                self._library = pickle.load(fin, encoding='latin1')
        self._random_generator = np.random.RandomState(seed)
        self._interp_factor, self._interp_factor2 = interp_factor, interp_factor2
        self._random_numbers = {}
        self._arz_version = arz_version
        self._ctx = ctx

    # ---- the reference's small methods ----------------------------------------------------------------------
    def em_fraction(self, energy):
        """energy fraction of the electromagnetic component of a hadronic shower (:436-447)"""
        if not _PARAMETERS[self._arz_version]['em']:
            return 1
        epsilon = np.log10(energy / 1.)
        f_epsilon = -21.98905 - 2.32492 * epsilon
        f_epsilon += 0.019650 * epsilon ** 2 + 13.76152 * np.sqrt(epsilon)
        return f_epsilon

    def set_seed(self, seed):
        self._random_generator.seed(seed)

    def set_interpolation_factor(self, interp_factor):
        self._interp_factor = interp_factor

    def set_interpolation_factor2(self, interp_factor):
        self._interp_factor2 = interp_factor

    def get_last_shower_profile_id(self):
        return self._random_numbers

    def get_shower_profile(self, shower_energy, shower_type, iN):
        energies = np.array([*self._library[shower_type]])
        iE = np.argmin(np.abs(energies - shower_energy))
        profiles = self._library[shower_type][energies[iE]]
        return profiles['depth'], profiles['charge_excess'][iN] * (shower_energy / energies[iE])

    def _pick(self, shower_energy, shower_type, same_shower, iN):
        """closest library energy, rescaling factor, profile number (:561-591)"""
        if shower_type == 'TAU':
            raise NotImplementedError("Tau showers are not yet implemented")
        if shower_type not in ('HAD', 'EM'):
            raise NotImplementedError("showers of type {} are not implemented. Use 'HAD', 'EM'".format(shower_type))
        energies = np.array([*self._library[shower_type]])
        iE = np.argmin(np.abs(energies - shower_energy))
        profiles = self._library[shower_type][energies[iE]]
        n_profiles = len(profiles['charge_excess'])
        if iN is None or np.isnan(iN):
            if same_shower and shower_type in self._random_numbers:
                iN = self._random_numbers[shower_type]
            else:
                iN = self._random_generator.randint(n_profiles)
                self._random_numbers[shower_type] = iN
        else:
            iN = int(iN)
            self._random_numbers[shower_type] = iN
        return profiles['depth'], profiles['charge_excess'][iN], shower_energy / energies[iE], iN

    # ---- GPU calls ------------------------------------------------------------------------------------------
    def _run(self, energy, theta, R, types, depth, ce_rows, index, rescale, N, dt, n_index, shift_for_xmax, maximum_angle,
             em_factor=None, return_vp=False):
        lib = L.load()
        ctx = self._ctx or _context()
        n = len(energy)
        depth = L.f64(depth)
        ce_rows = np.ascontiguousarray(ce_rows, float).reshape(-1, len(depth))
        if self._interp_factor != 1:   # resampling of the whole profile (:108-113)
            nd = int(self._interp_factor * len(depth))
            dense = np.linspace(min(depth), max(depth), nd)
            ce_rows = np.ascontiguousarray([np.interp(dense, depth, c) for c in ce_rows])
            depth = np.ascontiguousarray(dense)
        par = np.ascontiguousarray([_PARAMETERS[self._arz_version]['HAD'], _PARAMETERS[self._arz_version]['EM']], float)
        ty = np.array([0 if t == 'HAD' else 1 for t in types], np.int32)
        if em_factor is None:
            em_factor = [self.em_fraction(e) if t == 'HAD' else 1. for e, t in zip(energy, types)]
        emf = L.f64(em_factor)
        trace = np.zeros((n, 3, N))
        vp = np.zeros((n, N + 1, 2)) if return_vp else None
        status = lib.nrhip_arz_time_trace_batch(
            ctx._h, n, L.dptr(L.f64(energy)), L.dptr(L.f64(theta)), L.dptr(L.f64(R)), L.iptr(ty), L.dptr(emf),
            L.iptr(np.ascontiguousarray(index, np.int32)), L.dptr(L.f64(rescale)), len(ce_rows), len(depth), L.dptr(depth),
            L.dptr(ce_rows), L.dptr(par), int(N), float(dt), float(n_index), float(self._interp_factor2), int(bool(shift_for_xmax)),
            float(maximum_angle), L.dptr(trace), L.dptr(vp) if return_vp else ctypes.cast(None, L.c_double_p))
        if status != 0:
            msg = lib.nrhip_last_error().decode()
            if 'length of indices' in msg or 'not implemented' in msg:   # the reference's NotImplementedError cases (:207, :635-641)
                raise NotImplementedError(msg)
            raise L.NrhipError(msg)
        return (trace, vp) if return_vp else trace

    def get_time_trace(self, shower_energy, theta, N, dt, shower_type, n_index, R, shift_for_xmax=False, same_shower=False,
                       iN=None, output_mode='trace', maximum_angle=20 * np.pi / 180, profile_depth=None, profile_ce=None):
        """ARZ.get_time_trace (:500-673): on-sky traces [3, N] (eR, eTheta, ePhi)"""
        if profile_depth is None:
            profile_depth, ce, resc, iN = self._pick(shower_energy, shower_type, same_shower, iN)
        else:
            if profile_ce is None:
                raise ValueError("if profile_depth is provided, profile_ce must also be provided")
            if shower_type not in ('HAD', 'EM'):
                raise NotImplementedError("showers of type {} are not implemented. Use 'HAD', 'EM'".format(shower_type))
            ce, resc = profile_ce, 1.
        trace = self._run([shower_energy], [theta], [R], [shower_type], profile_depth, [ce], [0], [resc], N, dt, n_index,
                          shift_for_xmax, maximum_angle)[0]
        ce_scaled = np.asarray(ce, float) * resc
        if output_mode == 'full':
            return trace, profile_depth, ce_scaled
        if output_mode == 'Xmax':
            return trace, profile_depth[np.argmax(ce_scaled)] / RHO
        return trace

    def get_time_trace_batch(self, shower_energy, theta, N, dt, shower_type, n_index, R, iN, shift_for_xmax=False,
                             maximum_angle=20 * np.pi / 180):
        """Many (shower, ray) pairs in one call; iN[i] = profile number of shower i (draw them first, e.g. with
        `draw_profile_numbers`, so that the result does not depend on how the list is split).  All profiles of the library
        must share one depth grid (the reference's libraries do).  Returns [n, 3, N]."""
        shower_energy, theta, R = (np.atleast_1d(L.f64(v)) for v in (shower_energy, theta, R))
        n = len(shower_energy)
        if n == 0:
            return np.zeros((0, 3, N))
        types = [shower_type] * n if isinstance(shower_type, str) else [str(t).upper() for t in shower_type]
        rows, row_of, index, resc, depth = [], {}, np.zeros(n, np.int32), np.zeros(n), None
        for i in range(n):
            energies = np.array([*self._library[types[i]]])
            E_lib = energies[np.argmin(np.abs(energies - shower_energy[i]))]
            prof = self._library[types[i]][E_lib]
            if depth is None:
                depth = np.asarray(prof['depth'], float)
            elif len(prof['depth']) != len(depth) or np.any(np.asarray(prof['depth']) != depth):
                raise ValueError("the profiles of the library do not share one depth grid")
            key = (types[i], float(E_lib), int(iN[i]))
            if key not in row_of:
                row_of[key] = len(rows)
                rows.append(np.asarray(prof['charge_excess'][int(iN[i])], float))
            index[i], resc[i] = row_of[key], shower_energy[i] / E_lib
        return self._run(shower_energy, theta, R, types, depth, rows, index, resc, N, dt, n_index, shift_for_xmax, maximum_angle)

    def draw_profile_numbers(self, shower_energy, shower_type):
        """the random profile numbers the reference would draw for these showers, in order (one randint per shower)"""
        out = []
        for E, t in zip(np.atleast_1d(shower_energy), [shower_type] * len(np.atleast_1d(shower_energy))
                        if isinstance(shower_type, str) else shower_type):
            energies = np.array([*self._library[t]])
            n_prof = len(self._library[t][energies[np.argmin(np.abs(energies - E))]]['charge_excess'])
            out.append(self._random_generator.randint(n_prof))
            self._random_numbers[t] = out[-1]
        return np.array(out, np.int32)

    def get_vector_potential(self, shower_energy, theta, N, dt, profile_depth, profile_ce, shower_type='HAD', n_index=1.78,
                             distance=1., shift_for_xmax=False, em_factor=1.):
        """module-level get_vector_potential of the reference (:36-275) with this object's model parameters and
        interpolation factors: [N + 1, 3]"""
        _, vp = self._run([shower_energy], [theta], [distance], [shower_type], profile_depth, [profile_ce], [0], [1.], N, dt,
                          n_index, shift_for_xmax, np.inf, em_factor=[em_factor], return_vp=True)
        return np.stack([vp[0][:, 0], np.zeros(N + 1), vp[0][:, 1]], axis=1)
