"""Drop-in for the reference's analytic ray tracer plugin.

`ray_tracing` has the constructor and methods of NuRadioMC.SignalProp.analyticraytracing.ray_tracing
(NuRadioMC/SignalProp/analyticraytracing.py:1932-3052) / ray_tracing_base
(NuRadioMC/SignalProp/propagation_base_class.py), with the same argument meaning and error behaviour, and runs
every computation on the GPU through libnrhip.  `get_propagation_module('analytic')` mirrors
NuRadioMC/SignalProp/propagation.py:21-56 so that `simulation.py:1237` can take the class from here.

Per-pair calls are thin wrappers over the batch entry points; the batched path a simulation should use is
`Context.find_solutions_batch` / `Station.simulate_events`.
"""
import logging
import os
import numpy as np
from .context import Context

solution_types = {1: 'direct', 2: 'refracted', 3: 'reflected'}
solution_types_revert = {v: k for k, v in solution_types.items()}
available_modules = ['analytic']

_contexts = {}


def _context_for(medium, attenuation_model, device=0, gl3_table=None, ray_finder='true_roots'):
    key = (float(medium.n_ice), float(medium.delta_n), float(medium.z_0), attenuation_model, device, ray_finder)
    if key not in _contexts:
        if attenuation_model == 'GL3' and gl3_table is None:
            # the model is defined by a data file of the NuRadioMC installation the drop-in is used in
            import os
            import NuRadioMC.utilities.attenuation as _att
            gl3_table = os.path.join(os.path.dirname(_att.__file__), 'data', 'GL3_params.csv')
        _contexts[key] = Context(key[:3], attenuation_model, device=device, gl3_table=gl3_table, ray_finder=ray_finder)
    return _contexts[key]


birefringence_models = {}   # name -> three (knots, coefficients, 3) tuples; filled on demand from a NuRadioMC installation


def birefringence_model(name):
    """The depth splines of nx, ny, nz of a birefringence ice model (NuRadioMC/utilities/medium.py:103-108 loads
    utilities/birefringence_models/birefringence_<name>.npy, spline coefficients in FITPACK's (t, c, k) form).  Register
    tables of your own in `birefringence_models`; otherwise the data file of an installed NuRadioMC is read."""
    if name not in birefringence_models:
        try:
            import NuRadioMC.utilities.medium as _m
            path = os.path.join(os.path.dirname(os.path.realpath(_m.__file__)), 'birefringence_models',
                                'birefringence_' + name + '.npy')
            tck = np.load(path, allow_pickle=True)
        except Exception as e:
            raise FileNotFoundError("birefringence model {}: no table registered in nuradiomc_amd.propagation."
                                    "birefringence_models and no NuRadioMC installation to read it from ({})".format(name, e))
        birefringence_models[name] = [(np.asarray(t[0], float), np.asarray(t[1], float), int(t[2])) for t in tck]
    return birefringence_models[name]


def get_propagation_module(name=None):
    if name == 'analytic':
        return ray_tracing
    raise NotImplementedError("Module '{}' not implemented. Available modules: {}".format(name, str(available_modules)))


def _fresnel(zenith, n_2, n_1):
    """get_fresnel_r_p / get_fresnel_r_s (NuRadioReco/utilities/geometryUtilities.py:208-263)"""
    n = n_2 / n_1
    s = np.lib.scimath.sqrt(n ** 2 - np.sin(zenith) ** 2)
    r_p = np.conjugate((n ** 2 * np.cos(zenith) - s) / (n ** 2 * np.cos(zenith) + s))
    r_s = np.conjugate((np.cos(zenith) - s) / (np.cos(zenith) + s))
    return r_p, r_s

def analytic_ray_path(X1, X2, C0, n_ice, delta_n, z_0, n_points=1000):
    """ray_tracing.get_path (analyticraytracing.py:2148-2162 / :1239-1291): n_points positions along the solution with launch
    parameter C0, equally spaced in (mirrored) depth, from the LOWER of the two end points to the other one -- the order the
    reference returns.  Closed form of the exponential profile: y(z) = z_0 / sqrt(n_ice^2 C0^2 - 1) * ln(g / (2 sqrt(c (g^2 - b g
    + c)) - b g + 2 c)) + C1 with g = delta_n exp(z / z_0), b = 2 n_ice, c = n_ice^2 - C0^-2, mirrored at the turning depth
    (a reflection off the surface is a turning point at z = 0)."""
    X1, X2 = np.asarray(X1, float), np.asarray(X2, float)
    A, B = (X1, X2) if X2[2] >= X1[2] else (X2, X1)
    b, c = 2 * n_ice, n_ice ** 2 - C0 ** -2
    pref = z_0 / np.sqrt(n_ice ** 2 * C0 ** 2 - 1)

    def y_of(z):     # without C1, valid below the turning depth
        g = delta_n * np.exp(np.asarray(z, float) / z_0)
        return pref * np.log(g / (2 * np.sqrt(c) * np.sqrt(np.abs(g * g - b * g + c)) - b * g + 2 * c))
    g_turn = 0.5 * b - np.sqrt(0.25 * b * b - c)
    z_turn = min(np.log(g_turn / delta_n) * z_0, 0.)
    y_turn0 = y_of(z_turn)
    d = np.hypot(B[0] - A[0], B[1] - A[1])   # horizontal distance of the end point in the plane of the ray (start at y = 0)
    C1 = -(y_of(A[2]) if A[2] < z_turn else 2 * y_turn0 - y_of(2 * z_turn - A[2]))
    y_turn = y_turn0 + C1
    z_stop = B[2] if not (y_turn < d) else A[2] + abs(z_turn - A[2]) + abs(z_turn - B[2])
    z = np.linspace(A[2], z_stop, int(n_points))
    up = z < z_turn
    yy, zz = np.empty_like(z), np.empty_like(z)
    yy[up], zz[up] = y_of(z[up]) + C1, z[up]
    yy[~up], zz[~up] = 2 * y_turn - (y_of(2 * z_turn - z[~up]) + C1), 2 * z_turn - z[~up]
    phi = np.arctan2(B[1] - A[1], B[0] - A[0])
    return np.stack([A[0] + yy * np.cos(phi), A[1] + yy * np.sin(phi), zz], axis=1)


class ray_tracing:
    def __init__(self, medium, attenuation_model=None, log_level=logging.NOTSET, n_frequencies_integration=None,
                 n_reflections=None, config=None, detector=None, ray_tracing_2D_kwards={}, use_cpp=None,
                 compile_numba=None, device=0, ray_finder=None):
        """ray_finder: 'true_roots' (default) | 'reference' -- which solution finder find_solutions runs (include/nrhip.h,
        nrhip_ctx_set_ray_finder; also accepted as ray_tracing_2D_kwards['ray_finder']).  'reference' repeats the reference's
        procedure and acceptance test (analyticraytracing.py:1476-1547) and with them its occasionally shorter solution list."""
        self.__logger = logging.getLogger('nuradiomc_amd.ray_tracing')
        self.__logger.setLevel(log_level)
        for attr in ('n_ice', 'delta_n', 'z_0'):
            if not hasattr(medium, attr):
                raise TypeError("The analytic raytracer can only handle ice model of the type 'IceModelSimple'")
        if not medium.delta_n > 0:
            raise RuntimeError('Analytic raytracer does not work with a uniform ice model. '
                               'Abort.... ! Use direct raytracing or a non-uniform ice model instead.')
        self._medium = medium
        self._config = config
        # propagation_base_class.py:86-133: config values override constructor arguments
        self._n_frequencies_integration = None
        self._n_reflections = None
        self._attenuation_model = None
        if config is not None:
            prop = config['propagation']
            self._n_frequencies_integration = prop.get('n_freq')
            self._n_reflections = prop.get('n_reflections')
            self._attenuation_model = prop.get('attenuation_model')
        if self._n_frequencies_integration is None:
            self._n_frequencies_integration = n_frequencies_integration or 100
        if self._n_reflections is None:
            self._n_reflections = n_reflections or 0
        if self._attenuation_model is None:
            self._attenuation_model = attenuation_model or 'SP1'
        if self._n_reflections:
            if getattr(medium, 'reflection', None) is None:
                self.__logger.warning("Ray paths with bottom reflections requested but medium does not have any "
                                      "reflective layer, setting number of reflections to zero.")
                self._n_reflections = 0
            elif self._n_reflections > 4:
                raise NotImplementedError("more than 4 reflections off the bottom are not provided")
        self.set_config(config)
        self._detector = detector
        self._max_detector_frequency = None
        if detector is not None:  # propagation_base_class.py:66-80
            for station_id in detector.get_station_ids():
                ch0 = detector.get_channel_ids(station_id)[0]
                fs = detector.get_sampling_frequency(station_id, ch0)
                if self._max_detector_frequency is None or fs * .5 > self._max_detector_frequency:
                    self._max_detector_frequency = fs * .5
        self._ray_finder = ray_finder or dict(ray_tracing_2D_kwards or {}).get('ray_finder', 'true_roots')
        self._ctx = _context_for(medium, self._attenuation_model, device, ray_finder=self._ray_finder)
        self.use_cpp = False
        self.reset_solutions()

    # ---- state ---------------------------------------------------------------------------------------------
    def reset_solutions(self):
        self._X1 = None
        self._X2 = None
        self._results = None
        self._tab = None

    def set_start_and_end_point(self, x1, x2):
        self.reset_solutions()
        self._X1 = np.array(x1, dtype=float)
        self._X2 = np.array(x2, dtype=float)
        if self._n_reflections:  # propagation_base_class.py:156-161
            if self._X1[2] < self._medium.reflection or self._X2[2] < self._medium.reflection:
                self.__logger.error("start or stop point is below the reflective bottom layer at {:.1f}m".format(
                    self._medium.reflection))
                raise AttributeError("start or stop point is below the reflective bottom layer at {:.1f}m".format(
                    self._medium.reflection))

    def use_optional_function(self, function_name, *args, **kwargs):
        if hasattr(self, function_name):
            getattr(self, function_name)(*args, **kwargs)

    def set_solution(self, raytracing_results):
        """analyticraytracing.py:2092-2116: launch parameters read back from an output file instead of a new root search;
        the per-solution tables (vectors, path length, travel time, ...) are rebuilt on the GPU from those C0."""
        C0s = np.asarray(raytracing_results['ray_tracing_C0'], float).reshape(-1)
        if self._n_reflections:
            return self._set_solution_reflections(raytracing_results, C0s)
        given = np.full(2, np.nan)
        keep = C0s[~np.isnan(C0s)][:2]
        given[:len(keep)] = keep
        t = self._ctx.find_solutions_batch(self._X1[None], self._X2[None], given_C0=given[None])
        self._tab = {k: v[0] for k, v in t.items()}
        results = []
        j = 0
        for i in range(len(C0s)):
            if not np.isnan(C0s[i]):
                refl = raytracing_results['ray_tracing_reflection'][i] if 'ray_tracing_reflection' in raytracing_results else 0
                case = raytracing_results['ray_tracing_reflection_case'][i] if 'ray_tracing_reflection' in raytracing_results else 0
                results.append({'type': raytracing_results['ray_tracing_solution_type'][i], 'C0': C0s[i],
                                'C1': raytracing_results['ray_tracing_C1'][i], 'reflection': refl, 'reflection_case': case})
                j += 1
        self._results = results[:2]

    def _set_solution_reflections(self, raytracing_results, C0s):
        st = self.get_number_of_raytracing_solutions()
        keep = np.flatnonzero(~np.isnan(C0s))[:st]
        sol = dict(n_sol=np.array([len(keep)], np.int32), C0=np.full((1, st), np.nan),
                   reflection=np.zeros((1, st), np.int32), reflection_case=np.zeros((1, st), np.int32))
        has = 'ray_tracing_reflection' in raytracing_results
        for j, i in enumerate(keep):
            sol['C0'][0, j] = C0s[i]
            sol['reflection'][0, j] = raytracing_results['ray_tracing_reflection'][i] if has else 0
            sol['reflection_case'][0, j] = raytracing_results['ray_tracing_reflection_case'][i] if has else 0
        t = self._ctx.find_solutions_reflections_batch(self._X1[None], self._X2[None], self._n_reflections,
                                                       self._medium.reflection, solutions=sol)
        self._tab = {k: v[0] for k, v in t.items()}
        self._results = [{'type': raytracing_results['ray_tracing_solution_type'][i], 'C0': C0s[i],
                          'C1': raytracing_results['ray_tracing_C1'][i], 'reflection': int(sol['reflection'][0, j]),
                          'reflection_case': int(sol['reflection_case'][0, j])} for j, i in enumerate(keep)]

    def find_solutions(self):
        if self._X2[2] > 0 or self._X1[2] > 0:
            # the reference's Python solution finder reports no solution for a point in air: its objective returns the
            # "turning point below the target" penalty for every C0 there (analyticraytracing.py:1437-1449 with :247-253;
            # 0 of 60 random pairs in tests/golden/gen notes), and so do the kernels
            self.__logger.warning("can't find a solution for ice/air propagation")
        if self._n_reflections:  # :2118-2130: the plain call, then (i reflections, case 1 / 2)
            z_refl = self._medium.reflection
            t = self._ctx.find_solutions_reflections_batch(self._X1[None], self._X2[None], self._n_reflections, z_refl)
            self._tab = {k: v[0] for k, v in t.items()}
            self._results = [{'type': int(self._tab['type'][i]), 'C0': float(self._tab['C0'][i]),
                              'C1': float(self._tab['C1'][i]), 'reflection': int(self._tab['reflection'][i]),
                              'reflection_case': int(self._tab['reflection_case'][i])}
                             for i in range(int(self._tab['n_sol']))]
            return
        t = self._ctx.find_solutions_batch(self._X1[None], self._X2[None])
        self._tab = {k: v[0] for k, v in t.items()}
        n = int(self._tab['n_sol'])
        self._results = [{'type': int(self._tab['type'][i]), 'C0': float(self._tab['C0'][i]),
                          'C1': float(self._tab['C1'][i]), 'reflection': 0, 'reflection_case': 1} for i in range(n)]

    def has_solution(self):
        return len(self._results) > 0

    def get_number_of_solutions(self):
        return len(self._results)

    def get_results(self):
        return self._results

    def get_number_of_raytracing_solutions(self):
        return 2 + 4 * self._n_reflections

    def _check(self, iS):
        n = self.get_number_of_solutions()
        if iS >= n:
            self.__logger.error("solution number {:d} requested but only {:d} solutions exist".format(iS + 1, n))
            raise IndexError

    def get_solution_type(self, iS):
        self._check(iS)
        return int(self._tab['type'][iS])

    def get_launch_vector(self, iS):
        self._check(iS)
        return self._tab['launch'][iS].copy()

    def get_receive_vector(self, iS):
        self._check(iS)
        return self._tab['receive'][iS].copy()

    def get_reflection_angle(self, iS):
        """:2626 -> :1201-1237: the zenith angle of the reflection at the surface or None; with bottom reflections one entry
        per path segment (np.squeeze of the list, as in the reference)"""
        self._check(iS)
        a = self._tab['refl_angle'][iS]
        if self._n_reflections and self._results[iS]['reflection'] > 0:
            mask, nseg = int(self._tab['surface_mask'][iS]), int(self._tab['n_segments'][iS])
            return np.squeeze([float(a) if (mask >> j) & 1 else None for j in range(nseg)])
        return None if np.isnan(a) else float(a)

    def get_path_length(self, iS, analytic=True):
        self._check(iS)
        return float(self._tab['D'][iS])

    def get_travel_time(self, iS, analytic=True):
        self._check(iS)
        return float(self._tab['T'][iS])

    def get_path(self, iS, n_points=1000):
        """analyticraytracing.py:2148-2162 (plotting helper, host side): see analytic_ray_path.  Paths with reflections off
        the bottom are not drawn."""
        self._check(iS)
        r = self._results[iS]
        if r.get('reflection', 0):
            raise NotImplementedError("get_path: paths with reflections off the bottom are not drawn")
        return analytic_ray_path(self._X1, self._X2, float(r['C0']), self._medium.n_ice, self._medium.delta_n, self._medium.z_0,
                                 n_points)

    def get_attenuation(self, iS, frequency, max_detector_freq=None):
        """analyticraytracing.py:2744 -> :933-1089: coarse grid on the GPU, np.interp on the host, DC = 1"""
        from .station import attenuation_frequencies
        self._check(iS)
        frequency = np.asarray(frequency, float)
        freqs = attenuation_frequencies(frequency, self._n_frequencies_integration, max_detector_freq)
        out = np.ones_like(frequency)
        mask = frequency > 0
        r = self._results[iS]
        if r['reflection'] > 0:   # one factor per path segment, each interpolated before they are multiplied (:1078-1084)
            _, seg = self._ctx.attenuation_reflections_batch(self._X1[None], self._X2[None], [r['C0']], [r['reflection']],
                                                             [r['reflection_case']], self._medium.reflection, freqs,
                                                             return_segments=True)
            for coarse in seg[0]:
                if not np.any(np.isnan(coarse)):
                    out[mask] *= np.interp(frequency[mask], freqs, coarse)
            return out
        coarse = self._ctx.attenuation_batch(self._X1[None], self._X2[None], [r['C0']], freqs)[0]
        out[mask] = np.interp(frequency[mask], freqs, coarse)
        return out

    def get_focusing(self, iS, dz=-0.01, limit=2., analytic=False):
        """analyticraytracing.py:2778-2888, numerical branch (the reference falls back to it whenever its analytic
        formula fails): the ray to the receiver moved by dz is traced on the GPU as well; receiver and emitter in ice."""
        self._check(iS)
        rec_vec = -1.0 * self.get_receive_vector(iS)
        rec_ang = np.arccos(rec_vec[2] / np.sqrt(rec_vec[0] ** 2 + rec_vec[1] ** 2 + rec_vec[2] ** 2))
        lau_vec = self.get_launch_vector(iS)
        lau_ang = np.arccos(lau_vec[2] / np.sqrt(lau_vec[0] ** 2 + lau_vec[1] ** 2 + lau_vec[2] ** 2))
        vet_pos, rec_pos = self._X1, self._X2
        rec_pos1 = np.array([rec_pos[0], rec_pos[1], rec_pos[2] + dz])
        if self._n_reflections:
            t1 = self._ctx.find_solutions_reflections_batch(vet_pos[None], rec_pos1[None], self._n_reflections,
                                                            self._medium.reflection)
        else:
            t1 = self._ctx.find_solutions_batch(vet_pos[None], rec_pos1[None])
        if iS < int(t1['n_sol'][0]):
            lau_vec1 = t1['launch'][0][iS]
            lau_ang1 = np.arccos(lau_vec1[2] / np.sqrt(lau_vec1[0] ** 2 + lau_vec1[1] ** 2 + lau_vec1[2] ** 2))
            distance = self.get_path_length(iS)
            focusing = np.sqrt(distance / np.sin(rec_ang) * np.abs((lau_ang1 - lau_ang) / (rec_pos1[2] - rec_pos[2])))
            radius = np.linalg.norm(rec_pos - vet_pos)
            sin_theta = np.linalg.norm((rec_pos - vet_pos)[:-1]) / radius
            focusing *= np.sqrt((distance * np.sin(lau_ang)) / (radius * sin_theta))
        else:
            focusing = 1.0
            self.__logger.warning("too few ray tracing solutions, setting focusing factor to 1")
        if focusing > limit:
            focusing = limit
        n1 = self._medium.n_ice - self._medium.delta_n * np.exp(vet_pos[2] / self._medium.z_0)
        n2 = self._medium.n_ice - self._medium.delta_n * np.exp(rec_pos[2] / self._medium.z_0)
        return focusing * (n1 / n2) ** 0.5

    def get_output_parameters(self):
        return [{'name': 'ray_tracing_C0', 'ndim': 1}, {'name': 'ray_tracing_C1', 'ndim': 1},
                {'name': 'focusing_factor', 'ndim': 1}, {'name': 'ray_tracing_reflection', 'ndim': 1},
                {'name': 'ray_tracing_reflection_case', 'ndim': 1}, {'name': 'ray_tracing_solution_type', 'ndim': 1}]

    def get_raytracing_output(self, i_solution):
        self._check(i_solution)
        focusing = 1
        if self._config['propagation']['focusing']:
            focusing = self.get_focusing(i_solution, limit=float(self._config['propagation']['focusing_limit']))
        r = self._results[i_solution]
        return {'ray_tracing_C0': r['C0'], 'ray_tracing_C1': r['C1'], 'ray_tracing_reflection': r['reflection'],
                'ray_tracing_reflection_case': r['reflection_case'],
                'ray_tracing_solution_type': self.get_solution_type(i_solution), 'focusing_factor': focusing}

    def apply_propagation_effects(self, efield, i_solution):
        """analyticraytracing.py:2937-3033 for in-ice rays: attenuation and surface-reflection Fresnel factors.
        `efield` is any object with get_frequency_spectrum / get_frequencies / get_sampling_rate /
        set_frequency_spectrum (the reference's ElectricField)."""
        spec = efield.get_frequency_spectrum()
        prop = self._config['propagation']
        if prop['attenuate_ice']:
            max_freq = np.max(efield.get_frequencies()) if self._max_detector_frequency is None \
                else self._max_detector_frequency
            spec *= self.get_attenuation(i_solution, efield.get_frequencies(), max_freq)
        for zenith_reflection in np.atleast_1d(self.get_reflection_angle(i_solution)):  # one per surface reflection
            if zenith_reflection is None:
                continue
            n1 = self._medium.n_ice - self._medium.delta_n * np.exp(-0.01 / self._medium.z_0)
            r_theta, r_phi = _fresnel(zenith_reflection, n_2=1., n_1=n1)
            try:
                from NuRadioReco.framework.parameters import electricFieldParameters as efp
                efield[efp.reflection_coefficient_theta] = r_theta
                efield[efp.reflection_coefficient_phi] = r_phi
            except Exception:
                pass
            spec[1] *= r_theta
            spec[2] *= r_phi
        i_reflections = self._results[i_solution]['reflection']
        if i_reflections > 0:  # :3002-3009: every bottom reflection costs the layer's coefficient and shifts the phase
            reflection_coefficient = self._medium.reflection_coefficient ** i_reflections
            phase_shift = (i_reflections * self._medium.reflection_phase_shift) % (2 * np.pi)
            spec[1] *= reflection_coefficient * np.exp(1j * phase_shift)
            spec[2] *= reflection_coefficient * np.exp(1j * phase_shift)
        if prop.get('focusing'):  # analyticraytracing.py:3011-3016
            spec[1:] *= self.get_focusing(i_solution, limit=float(prop['focusing_limit']))
        if prop.get('birefringence'):  # :3018-3030
            if prop.get('birefringence_propagation', 'analytical') != 'analytical':
                raise NotImplementedError("birefringence_propagation '{}' needs the RadioPropa tracer and is not provided "
                                          "(analytical)".format(prop.get('birefringence_propagation')))
            spec = self.get_pulse_propagation_birefringence(spec, efield.get_sampling_rate(), i_solution,
                                                            bire_model=prop.get('birefringence_model', 'southpole_A'))
        efield.set_frequency_spectrum(spec, efield.get_sampling_rate())
        return efield

    def get_pulse_propagation_birefringence(self, pulse, samp_rate, i_solution, bire_model='southpole_A'):
        """:2369-2445 on the GPU: pulse = spectra (eR, eTheta, ePhi); eR is not touched"""
        self._check(i_solution)
        if self._results[i_solution]['reflection'] > 0:
            raise NotImplementedError("birefringence along paths with bottom reflections is not provided (analyticraytracing.py:2404-2411 walks "
                                      "acc - 1 steps of a path that get_path returns with (k + 1) acc - k points for k reflections: "
                                      "it stops part way, tests/golden/ref_bire_reflection_probe.txt)")
        pulse = np.array(pulse, dtype=complex)
        angle = self._config['propagation'].get('angle_to_iceflow')
        out = self._ctx.birefringence_batch(self._X1[None], self._X2[None], [self._results[i_solution]['C0']],
                                            [self.get_path_length(i_solution)], pulse[1:][None], samp_rate,
                                            birefringence_model(bire_model), angle_to_iceflow=angle)[0]
        pulse[1], pulse[2] = out[0], out[1]
        return pulse

    def get_config(self):
        return self._config

    def set_config(self, config):
        if config is None:
            self._config = {'propagation': {'attenuate_ice': True, 'focusing_limit': 2, 'focusing': False,
                                            'birefringence': False}}
        else:
            self._config = config
