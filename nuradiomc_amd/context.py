"""Context = one GPU + one ice model + one attenuation model (include/nrhip.h: nrhip_ctx)."""
import ctypes
import numpy as np
from . import _lib as L
from . import station as _station  # noqa: F401  (registers the station entry points)

#: NuRadioMC/utilities/attenuation.py:14
ATTENUATION_MODEL_TO_INT = {"SP1": 1, "GL1": 2, "MB1": 3, "GL2": 4, "GL3": 5}
MAXS = 2
RAY_FINDER_TO_INT = {'true_roots': 0, 'reference': 1}   # NRHIP_FINDER_*


CROSS_SECTION_TO_INT = {'ctw': 0, 'ghandi': 1, 'given': 2}   # 'given': `energy` carries the cross sections [m^2]
PROTON_MASS_KG = 1.67262192595e-27  # scipy.constants.m_p (CODATA 2022), what cross_sections.get_interaction_length uses


class _EarthModel(ctypes.Structure):  # nrhip_earth_model
    _fields_ = [('n_layers', ctypes.c_int32), ('reserved', ctypes.c_int32), ('earth_radius', ctypes.c_double),
                ('radii', ctypes.c_double * 16), ('coef', (ctypes.c_double * 4) * 16)]


class Context:
    """Owns a `nrhip_ctx`.  `ice` is (n_ice, delta_n, z_0) of n(z) = n_ice - delta_n exp(z / z_0)."""

    def __init__(self, ice, attenuation_model="SP1", device=0, gl3_table=None, ray_finder='true_roots'):
        """gl3_table (GL3 only): the depth table of the model, an [n, 3] array (depth, slope, offset) or the path of
        NuRadioMC/utilities/data/GL3_params.csv (comma separated) -- the model is defined by that file.
        ray_finder: 'true_roots' (default: every root of the path objective, a superset of the reference's list) or 'reference'
        (the reference's procedure and acceptance test to the letter, analyticraytracing.py:1476-1547; `set_ray_finder`)."""
        self._lib = L.load()
        if attenuation_model not in ATTENUATION_MODEL_TO_INT:
            raise NotImplementedError("attenuation model {} is not implemented".format(attenuation_model))
        self.ice = tuple(float(v) for v in ice)
        self.attenuation_model = attenuation_model
        self.device = device
        h = ctypes.c_void_p()
        L.check(self._lib.nrhip_ctx_create(device, *self.ice, ATTENUATION_MODEL_TO_INT[attenuation_model],
                                           ctypes.byref(h)))
        self._h = h
        if attenuation_model == 'GL3':
            if gl3_table is None:
                raise ValueError("attenuation model GL3 needs gl3_table (NuRadioMC/utilities/data/GL3_params.csv)")
            t = np.genfromtxt(gl3_table, delimiter=',') if isinstance(gl3_table, str) else np.asarray(gl3_table, float)
            d, sl, of = (np.ascontiguousarray(t[:, k]) for k in range(3))
            L.check(self._lib.nrhip_ctx_set_gl3_table(h, len(t), L.dptr(d), L.dptr(sl), L.dptr(of)))
        self.ray_finder = 'true_roots'
        if ray_finder != 'true_roots':
            self.set_ray_finder(ray_finder)

    def set_ray_finder(self, finder):
        """'true_roots' | 'reference' (nrhip_ctx_set_ray_finder) for every later call on this context."""
        if finder not in RAY_FINDER_TO_INT:
            raise ValueError("ray_finder must be one of {}".format(sorted(RAY_FINDER_TO_INT)))
        L.check(self._lib.nrhip_ctx_set_ray_finder(self._h, RAY_FINDER_TO_INT[finder]))
        self.ray_finder = finder

    def close(self):
        if getattr(self, '_h', None):
            for st in list(getattr(self, '_stations', ())):  # stations hold the context: destroy them first
                st.close()
            self._lib.nrhip_ctx_destroy(self._h)
            self._h = None

    def _register_station(self, station):
        import weakref
        if not hasattr(self, '_stations'):
            self._stations = weakref.WeakSet()
        self._stations.add(station)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device memory ---------------------------------------------------------------------------
    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        L.check(self._lib.nrhip_malloc(self._h, int(nbytes), ctypes.byref(p)))
        return p.value

    def free(self, dev_ptr):
        L.check(self._lib.nrhip_free(self._h, ctypes.c_void_p(dev_ptr)))

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        p = self.malloc(max(a.nbytes, 8))
        if a.nbytes:
            L.check(self._lib.nrhip_memcpy_h2d(self._h, ctypes.c_void_p(p), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        return p

    def copy_to_device(self, dev_ptr, a):
        """overwrite an existing device buffer with the host array a"""
        a = np.ascontiguousarray(a)
        if a.nbytes:
            L.check(self._lib.nrhip_memcpy_h2d(self._h, ctypes.c_void_p(dev_ptr), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

    def to_host(self, out, dev_ptr):
        if out.nbytes:
            L.check(self._lib.nrhip_memcpy_d2h(self._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dev_ptr),
                                               out.nbytes))
        return out

    def synchronize(self):
        L.check(self._lib.nrhip_synchronize(self._h))

    # ---- Askaryan --------------------------------------------------------------------------------
    def askaryan_spectrum_batch(self, energy, theta, N, dt, shower_type, n_index, R, model, k_L=None):
        """[n, N/2+1] complex spectra of askaryan.get_frequency_spectrum for arrays of showers / viewing angles."""
        from .station import ASKARYAN_TO_INT, SHOWER_TO_INT
        if model not in ('Alvarez2009', 'Alvarez2000', 'ZHS1992'):
            raise NotImplementedError("model {} unknown".format(model))
        energy, theta, n_index, R = np.broadcast_arrays(L.f64(energy), L.f64(theta), L.f64(n_index), L.f64(R))
        n = energy.size
        st = np.broadcast_to(shower_type, energy.shape).reshape(-1)
        sti = np.zeros(n, np.int32)
        for i, s in enumerate(st):
            s = str(s).upper()
            if s not in SHOWER_TO_INT:
                raise NotImplementedError("shower type {} is not implemented in {} model.".format(s, model))
            sti[i] = SHOWER_TO_INT[s]
        kL = np.ascontiguousarray(np.broadcast_to(1.0 if k_L is None else L.f64(k_L), energy.shape).reshape(-1), float)
        if model == 'Alvarez2009' and k_L is None and np.any(sti == 1):
            raise ValueError("Alvarez2009 EM showers need k_L (the random draw stays on the host)")
        out = np.zeros((n, N // 2 + 1), np.complex128)
        a = [np.ascontiguousarray(v.reshape(-1), float) for v in (energy, theta, n_index, R)]
        L.check(self._lib.nrhip_askaryan_spectrum_batch(self._h, n, L.dptr(a[0]), L.dptr(a[1]), L.iptr(sti), L.dptr(a[2]),
                                                        L.dptr(a[3]), L.dptr(kL), ASKARYAN_TO_INT[model], int(N), float(dt),
                                                        out.ctypes.data_as(L.c_double_p)))
        return out

    def debug_czt(self, x, n_out, Q, sgn):
        x = np.ascontiguousarray(x, np.complex128).reshape(-1, x.shape[-1])
        out = np.zeros((x.shape[0], n_out), np.complex128)
        L.check(self._lib.nrhip_debug_czt(self._h, x.shape[0], x.shape[1], n_out, int(Q), float(sgn),
                                          x.ctypes.data_as(L.c_double_p), out.ctypes.data_as(L.c_double_p)))
        return out

    def debug_wave_sums(self, x):
        """test hook of csrc/wave_reduce.h: x [n_waves, 8, 64] -> [n_waves, 273] (layout: include/nrhip.h)"""
        x = np.ascontiguousarray(x, float)
        assert x.ndim == 3 and x.shape[1:] == (8, 64)
        out = np.zeros((x.shape[0], 273))
        L.check(self._lib.nrhip_debug_wave_sums(self._h, x.shape[0], x.ctypes.data_as(L.c_double_p), out.ctypes.data_as(L.c_double_p)))
        return out

    # ---- ray tracing -----------------------------------------------------------------------------
    def find_solutions_batch(self, x1, x2, outer=False, given_C0=None):
        """All ray solutions between x1[i] and x2[i] (or every x1 with every x2 if `outer`).

        Returns a dict of [n_pairs, 2(,3)] arrays: n_sol, type, C0, C1, D (path length), T (travel time),
        launch, receive, refl_angle -- NaN / 0 padded like the reference's HDF5 station tables.
        given_C0 [n_pairs, 2] (NaN = none): no root finding, the tables of these launch parameters (set_solution).
        """
        x1 = L.f64(x1).reshape(-1, 3)
        x2 = L.f64(x2).reshape(-1, 3)
        if outer:
            n = len(x1) * len(x2)
            n_x2 = len(x2)
        else:
            if len(x1) != len(x2):
                raise ValueError("x1 and x2 must have the same number of rows")
            n = len(x1)
            n_x2 = 0
        o = dict(n_sol=np.zeros(n, np.int32), type=np.zeros((n, MAXS), np.int32))
        for k in ('C0', 'C1', 'D', 'T', 'refl_angle'):
            o[k] = np.full((n, MAXS), np.nan)
        for k in ('launch', 'receive'):
            o[k] = np.full((n, MAXS, 3), np.nan)
        outs = (L.iptr(o['n_sol']), L.iptr(o['type']), L.dptr(o['C0']), L.dptr(o['C1']), L.dptr(o['D']), L.dptr(o['T']),
                L.dptr(o['launch']), L.dptr(o['receive']), L.dptr(o['refl_angle']))
        if given_C0 is None:
            L.check(self._lib.nrhip_find_solutions_batch(self._h, n, L.dptr(x1), L.dptr(x2), n_x2, *outs))
        else:
            g = np.ascontiguousarray(np.broadcast_to(L.f64(given_C0).reshape(-1, MAXS), (n, MAXS)))
            L.check(self._lib.nrhip_ray_records_batch(self._h, n, L.dptr(x1), L.dptr(x2), n_x2, L.dptr(g), *outs))
        return o

    def find_solutions_reflections_batch(self, x1, x2, n_reflections, z_reflection, outer=False, solutions=None):
        """Ray solutions with up to n_reflections reflections off the bottom of an ice shelf at depth z_reflection (< 0)
        (ray_tracing(medium, n_reflections).find_solutions, analyticraytracing.py:2118-2130).

        Returns a dict of [n_pairs, 2 + 4 n_reflections(, 3)] arrays in the reference's order of solutions: n_sol, type,
        C0, C1, reflection, reflection_case, D, T, launch, receive, refl_angle (surface reflection, NaN = none),
        n_segments, surface_mask (bit j: path segment j reflects at the surface), n_surface (their number).  solutions = dict(n_sol, C0, reflection, reflection_case):
        no root finding (set_solution)."""
        x1 = L.f64(x1).reshape(-1, 3)
        x2 = L.f64(x2).reshape(-1, 3)
        if outer:
            n, n_x2 = len(x1) * len(x2), len(x2)
        else:
            if len(x1) != len(x2):
                raise ValueError("x1 and x2 must have the same number of rows")
            n, n_x2 = len(x1), 0
        st = 2 + 4 * int(n_reflections)
        o = dict(n_sol=np.zeros(n, np.int32))
        for k in ('type', 'reflection', 'reflection_case', 'n_segments', 'surface_mask'):
            o[k] = np.zeros((n, st), np.int32)
        for k in ('C0', 'C1', 'D', 'T', 'refl_angle'):
            o[k] = np.full((n, st), np.nan)
        for k in ('launch', 'receive'):
            o[k] = np.full((n, st, 3), np.nan)
        fn = self._lib.nrhip_find_solutions_reflections_batch
        if solutions is not None:
            fn = self._lib.nrhip_ray_records_reflections_batch
            o['n_sol'][:] = np.asarray(solutions['n_sol'], np.int32).reshape(n)
            o['C0'][:] = L.f64(solutions['C0']).reshape(n, st)
            o['reflection'][:] = np.asarray(solutions['reflection'], np.int32).reshape(n, st)
            o['reflection_case'][:] = np.asarray(solutions['reflection_case'], np.int32).reshape(n, st)
        L.check(fn(self._h, n, L.dptr(x1), L.dptr(x2), n_x2, int(n_reflections), float(z_reflection), L.iptr(o['n_sol']),
                   L.iptr(o['type']), L.dptr(o['C0']), L.dptr(o['C1']), L.iptr(o['reflection']), L.iptr(o['reflection_case']),
                   L.dptr(o['D']), L.dptr(o['T']), L.dptr(o['launch']), L.dptr(o['receive']), L.dptr(o['refl_angle']),
                   L.iptr(o['n_segments']), L.iptr(o['surface_mask'])))
        o['n_surface'] = np.array([bin(v).count('1') for v in o['surface_mask'].ravel()], np.int32).reshape(n, st)
        return o

    def attenuation_reflections_batch(self, x1, x2, C0, reflection, reflection_case, z_reflection, freqs,
                                      return_segments=False):
        """exp(-int ds / L_att) along paths with bottom reflections: the product over the path segments (and, on request,
        the factors of the segments [n_rays, max(reflection) + 1, n_freq], NaN = no such segment)."""
        x1 = L.f64(x1).reshape(-1, 3)
        x2 = L.f64(x2).reshape(-1, 3)
        C0 = L.f64(C0).reshape(-1)
        rf = np.ascontiguousarray(reflection, np.int32).reshape(-1)
        rc = np.ascontiguousarray(reflection_case, np.int32).reshape(-1)
        freqs = L.f64(freqs).reshape(-1)
        att = np.zeros((len(C0), len(freqs)))
        seg = np.zeros((len(C0), (int(rf.max()) if len(rf) else 0) + 1, len(freqs)))
        L.check(self._lib.nrhip_attenuation_reflections_batch(self._h, len(C0), L.dptr(x1), L.dptr(x2), L.dptr(C0), L.iptr(rf),
                                                              L.iptr(rc), float(z_reflection), len(freqs), L.dptr(freqs),
                                                              L.dptr(att), L.dptr(seg)))
        return (att, seg) if return_segments else att

    def birefringence_batch(self, x1, x2, C0, path_length, spectra, sampling_rate, tck, angle_to_iceflow=None, n_ref=1.78,
                            return_steps=False):
        """Birefringent propagation of the (eTheta, ePhi) spectra [n_rays, 2, n_f] along the rays (x1, x2, C0)
        (get_pulse_propagation_birefringence, analyticraytracing.py:2369-2445).  tck = three (knots, coefficients[, 3])
        tuples: the depth splines of nx, ny, nz of a birefringence ice model.  Returns the new spectra (and the step
        records [n_steps, 5] = (a, b, c, d, t_1 - t_0) of all rays, one after the other)."""
        x1 = L.f64(x1).reshape(-1, 3)
        x2 = L.f64(x2).reshape(-1, 3)
        C0 = L.f64(C0).reshape(-1)
        D = L.f64(path_length).reshape(-1)
        n = len(C0)
        spec = np.ascontiguousarray(spectra, dtype=np.complex128)
        spec = spec.reshape(n, 2, spec.shape[-1]).copy()
        if n == 0:
            return (spec, np.zeros((0, 5))) if return_steps else spec
        n_f = spec.shape[2]
        knots = np.ascontiguousarray(np.concatenate([np.asarray(t[0], float) for t in tck]))
        coeffs = np.ascontiguousarray(np.concatenate([np.asarray(t[1], float) for t in tck]))
        for t in tck:
            if len(t) > 2 and int(t[2]) != 3:
                raise ValueError("birefringence_batch: the depth splines must be cubic")
            if len(t[0]) != len(t[1]):
                raise ValueError("birefringence_batch: knots and coefficients of a spline must have the same length (FITPACK tck)")
        nk = np.array([len(t[0]) for t in tck], np.int32)
        n_steps = int(np.sum(np.maximum(D.astype(int) - 1, 0)))
        steps = np.zeros((max(n_steps, 1), 5))
        L.check(self._lib.nrhip_birefringence_batch(self._h, n, L.dptr(x1), L.dptr(x2), L.dptr(C0), L.dptr(D), L.iptr(nk),
                                                    L.dptr(knots), L.dptr(coeffs), float(n_ref),
                                                    float('nan') if angle_to_iceflow is None else float(angle_to_iceflow), n_f,
                                                    float(sampling_rate), spec.view(np.float64).ctypes.data_as(L.c_double_p),
                                                    L.dptr(steps)))
        return (spec, steps[:n_steps]) if return_steps else spec

    def earth_weights_batch(self, zenith, energy, flavor, mode, endpoint=None, direction=None, model=None, step=500.,
                            nucleon_mass=None, return_slant_depth=False, cross_section_type='ctw'):
        """Earth-absorption weights of n events (earth_attenuation.get_weight, NuRadioMC/utilities/earth_attenuation.py:12-60;
        cross_section_type 'ctw' or 'ghandi'; 'given': `energy` holds each event's total cross section in m^2).  mode: 0 'simple', 1 'core_mantle_crust_simple', 2 chord through the layered density
        `model` = (earth_radius, radii [n_layers], coefficients [n_layers, 4]) from `endpoint` [n, 3] towards `direction`
        [n, 3] (PREM.slant_depth :183-240)."""
        if cross_section_type not in CROSS_SECTION_TO_INT:
            raise NotImplementedError("Cross-section {} not defined on the device ('ctw', 'ghandi')".format(cross_section_type))
        zenith = L.f64(zenith).reshape(-1)
        energy = L.f64(energy).reshape(-1)
        n = len(zenith)
        flavor = np.ascontiguousarray(np.broadcast_to(np.asarray(flavor), (n,)), dtype=np.int32)
        weight = np.ones(n)
        slant = np.zeros(n)
        md = None
        if int(mode) == 2:
            endpoint = L.f64(endpoint).reshape(n, 3)
            direction = L.f64(direction).reshape(n, 3)
            R, radii, coef = model
            radii = np.asarray(radii, float).reshape(-1)
            coef = np.asarray(coef, float).reshape(len(radii), 4)
            if len(radii) > 16:
                raise ValueError("earth_weights_batch: at most 16 layers")
            md = _EarthModel()
            md.n_layers = len(radii)
            md.earth_radius = float(R)
            for k in range(len(radii)):
                md.radii[k] = radii[k]
                for j in range(4):
                    md.coef[k][j] = coef[k, j]
        if nucleon_mass is None:
            nucleon_mass = PROTON_MASS_KG * 6.241509744511525e+36
        L.check(self._lib.nrhip_earth_weights_batch(
            self._h, n, L.dptr(zenith), L.dptr(energy), L.iptr(flavor), None if md is None else L.dptr(endpoint),
            None if md is None else L.dptr(direction), int(mode), CROSS_SECTION_TO_INT[cross_section_type],
            None if md is None else ctypes.byref(md), float(step),
            float(nucleon_mass), L.dptr(weight), L.dptr(slant) if md is not None else None))
        return (weight, slant) if return_slant_depth else weight

    def attenuation_batch(self, x1, x2, C0, freqs, return_neval=False):
        """exp(-int ds / L_att) for rays (x1[r] -> x2[r], C0[r]) at the given (> 0) frequencies."""
        x1 = L.f64(x1).reshape(-1, 3)
        x2 = L.f64(x2).reshape(-1, 3)
        C0 = L.f64(C0).reshape(-1)
        freqs = L.f64(freqs).reshape(-1)
        n = len(C0)
        att = np.zeros((n, len(freqs)))
        nev = np.zeros((n, len(freqs)), np.int32)
        L.check(self._lib.nrhip_attenuation_batch(self._h, n, L.dptr(x1), L.dptr(x2), L.dptr(C0), len(freqs),
                                                  L.dptr(freqs), L.dptr(att), L.iptr(nev)))
        return (att, nev) if return_neval else att

    def attenuation_last_overflow(self):
        """rays of the last attenuation_batch call that the dense quadrature kernel left to the general one"""
        return int(self._lib.nrhip_attenuation_last_overflow(self._h))

    def attenuation_length(self, z, frequency):
        z, frequency = np.broadcast_arrays(L.f64(z), L.f64(frequency))
        z = np.ascontiguousarray(z)
        frequency = np.ascontiguousarray(frequency)
        out = np.zeros(z.shape)
        L.check(self._lib.nrhip_attenuation_length(self._h, z.size, L.dptr(z), L.dptr(frequency), L.dptr(out)))
        return out
