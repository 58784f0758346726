"""Context = one GPU + one ice model + one attenuation model (include/nrhip.h: nrhip_ctx)."""
import ctypes
import numpy as np
from . import _lib as L

#: NuRadioMC/utilities/attenuation.py:14
ATTENUATION_MODEL_TO_INT = {"SP1": 1, "GL1": 2, "MB1": 3, "GL2": 4}
MAXS = 2


class Context:
    """Owns a `nrhip_ctx`.  `ice` is (n_ice, delta_n, z_0) of n(z) = n_ice - delta_n exp(z / z_0)."""

    def __init__(self, ice, attenuation_model="SP1", device=0):
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

    def close(self):
        if getattr(self, '_h', None):
            self._lib.nrhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ray tracing -----------------------------------------------------------------------------
    def find_solutions_batch(self, x1, x2, outer=False):
        """All ray solutions between x1[i] and x2[i] (or every x1 with every x2 if `outer`).

        Returns a dict of [n_pairs, 2(,3)] arrays: n_sol, type, C0, C1, D (path length), T (travel time),
        launch, receive, refl_angle -- NaN / 0 padded like the reference's HDF5 station tables.
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
        L.check(self._lib.nrhip_find_solutions_batch(
            self._h, n, L.dptr(x1), L.dptr(x2), n_x2, L.iptr(o['n_sol']), L.iptr(o['type']), L.dptr(o['C0']),
            L.dptr(o['C1']), L.dptr(o['D']), L.dptr(o['T']), L.dptr(o['launch']), L.dptr(o['receive']),
            L.dptr(o['refl_angle'])))
        return o

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

    def attenuation_length(self, z, frequency):
        z, frequency = np.broadcast_arrays(L.f64(z), L.f64(frequency))
        z = np.ascontiguousarray(z)
        frequency = np.ascontiguousarray(frequency)
        out = np.zeros(z.shape)
        L.check(self._lib.nrhip_attenuation_length(self._h, z.size, L.dptr(z), L.dptr(frequency), L.dptr(out)))
        return out
