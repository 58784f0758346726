"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's per-event Earth-absorption weight.
Nothing of the product imports this file; tests, smoke() and bench.py's cpu_baseline leg use it as the checker.

Follows NuRadioMC/utilities/earth_attenuation.py (get_weight :12-60, get_simple_weight :63-86,
get_core_mantle_crust_weight :89-130, PREM.density :171-181, PREM.slant_depth :183-240, CoreMantleCrustModel :243-270)
and NuRadioMC/utilities/cross_sections.py ('ctw' parametrisation: param :64-120, get_nu_cross_section :232 / :301-311,
get_interaction_length :393-421), called once per event group by simulation.py:880-903.
Pinned against tests/golden/ref_earth_weights.npz (tests/golden/gen/gen_earth_weights.py: the reference run here).

Statements kept from the reference because results depend on them:
  * the chord starts at the vertex and runs TOWARDS the arrival direction (theta, phi say where the neutrino came from);
  * n_steps = int(distance / step) (+1 if distance % step), ts = linspace(0, 1, n_steps) -- the samples are "just
    under" 500 m apart and include both ends; the last sample sits on the surface, r = R up to rounding, and the density
    there is that of the outermost layer or 0 (np.piecewise default) depending on that rounding: the operations below
    are written out in the reference's order so that every r is the same double;
  * radii outside every layer (r >= earth_radius) have density 0;
  * np.trapz over (rho * distance, ts).
  * np.dot(endpoint, direction) and np.linalg.norm(direction) (= sqrt(dot(d, d))) go through the BLAS ddot, whose
    tail loop evaluates three elements as fma(a2, b2, fma(a1, b1, a0 * b0)) on x86-64 with FMA (OpenBLAS 0.3.29 here:
    20000 of 20000 random vectors); written out below as exactly that, so the oracle does not depend on the BLAS build.
    With a plain sum of products instead, 6 % of the golden events put the surface sample on the other side of r = R.
"""
import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library('m') or 'libm.so.6')
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3


def _dot3(a, b):
    return _libm.fma(float(a[2]), float(b[2]), _libm.fma(float(a[1]), float(b[1]), float(a[0]) * float(b[0])))

# NuRadioReco/utilities/units.py (m = ns = eV = 1)
KG = 6.241509744511525e+36
G = 6.241509744511525e+33
CM = 0.01
GEV = 1e9
AMU = 1.66e-27 * KG                      # earth_attenuation.py:9
PROTON_MASS_KG = 1.67262192595e-27       # scipy.constants.m_p of the scipy in this image (CODATA 2022)
STEP = 500.0

_RHO = G / CM ** 3

# (c0, c1, c2, c3): rho(x) = ((c0 + c1 x) + c2 x^2) + c3 x^3, x = r / earth_radius.  Preliminary reference Earth model,
# Dziewonski & Anderson 1981, as tabulated in earth_attenuation.py:153-169
PREM_RADIUS = 6.3710e6
PREM_RADII = (1.2215e6, 3.4800e6, 5.7010e6, 5.7710e6, 5.9710e6, 6.1510e6, 6.3466e6, 6.3560e6, 6.3680e6, PREM_RADIUS)
_PREM_GCM3 = ((13.0885, 0., -8.8381, 0.), (12.5815, -1.2638, -3.6426, -5.5281), (7.9565, -6.4761, 5.5283, -3.0807),
              (5.3197, -1.4836, 0., 0.), (11.2494, -8.0298, 0., 0.), (7.1089, -3.8045, 0., 0.), (2.691, 0.6924, 0., 0.),
              (2.9, 0., 0., 0.), (2.6, 0., 0., 0.), (1.02, 0., 0., 0.))
CMC_RADIUS = 6.378140e6
CMC_RADII = (float(np.sqrt(1.2e13)), CMC_RADIUS - 4e4, CMC_RADIUS)
_CMC_GCM3 = ((14., 0., 0., 0.), (3.4, 0., 0., 0.), (2.9, 0., 0., 0.))


def _coefficients(table):
    # "13.0885 * units.g / units.cm ** 3": (c * g) / cm^3, the sign of a subtracted term moved into the coefficient (exact)
    return np.array([[(abs(c) * G / CM ** 3) * (1. if c >= 0 else -1.) for c in row] for row in table])


def earth_model(name):
    """(earth_radius, radii [n_layers], coefficients [n_layers][4]) of 'PREM' or 'core_mantle_crust'"""
    if name == 'PREM':
        return PREM_RADIUS, np.array(PREM_RADII), _coefficients(_PREM_GCM3)
    if name == 'core_mantle_crust':
        return CMC_RADIUS, np.array(CMC_RADII), _coefficients(_CMC_GCM3)
    raise NotImplementedError(name)


def ctw_param(energy, inttype):
    """cross_sections.param (:64-120), Connolly, Thorne, Waters 2011"""
    c = {'cc': (-1.826, -17.31, -6.406, 1.431, -17.91), 'nc': (-1.826, -17.31, -6.448, 1.431, -18.61),
         'cc_bar': (-1.033, -15.95, -7.247, 1.569, -17.72), 'nc_bar': (-1.033, -15.95, -7.296, 1.569, -18.30)}[inttype]
    energy = np.asarray(energy, dtype=float)
    if np.any(energy < 1e4 * GEV):
        return np.nan * np.ones_like(energy)
    epsilon = np.log10(energy / GEV)
    l_eps = np.log(epsilon - c[0])
    crscn = c[1] + c[2] * l_eps + c[3] * l_eps ** 2 + c[4] / l_eps
    return np.power(10, crscn) * CM ** 2


def ctw_total(energy, flavor):
    """get_nu_cross_section(..., inttype='total', cross_section_type='ctw') (:301-311): nc + cc, antiparticles apart"""
    energy = np.atleast_1d(np.asarray(energy, dtype=float))
    flavor = np.broadcast_to(np.asarray(flavor), energy.shape)
    out = np.empty_like(energy)
    for i in range(len(energy)):
        e = energy[i:i + 1]
        out[i] = (ctw_param(e, 'nc') + ctw_param(e, 'cc'))[0] if flavor[i] >= 0 else (ctw_param(e, 'nc_bar') + ctw_param(e, 'cc_bar'))[0]
    return out


def ghandi(energy):
    """get_nu_cross_section(..., cross_section_type='ghandi') (:280-281)"""
    return 7.84e-36 * CM ** 2 * np.power(np.atleast_1d(np.asarray(energy, dtype=float)) / GEV, 0.363)


def total_cross_section(energy, flavor, cross_section_type='ctw'):
    if cross_section_type == 'ctw':
        return ctw_total(energy, flavor)
    if cross_section_type == 'ghandi':
        return ghandi(energy)
    if cross_section_type == 'given':   # tabulated models evaluated by the caller: `energy` carries the cross sections [m^2]
        return np.asarray(energy, float)
    raise NotImplementedError(cross_section_type)


def interaction_length_unit_density(energy, flavor, proton_mass_kg=PROTON_MASS_KG, cross_section_type='ctw'):
    """get_interaction_length(density=1.) (:393-421)"""
    return proton_mass_kg * KG / total_cross_section(energy, flavor, cross_section_type) / 1.


def density(r, model):
    """PREM.density (:171-181): lower <= r < upper picks the layer, 0 outside"""
    R, radii, coef = model
    r = np.asarray(r, dtype=float)
    x = r / R
    bounds = np.concatenate(([0.], radii))
    rho = np.zeros_like(x)
    for k in range(len(radii)):
        m = (bounds[k] <= r) & (r < bounds[k + 1])
        xm = x[m]
        rho[m] = ((coef[k, 0] + coef[k, 1] * xm) + coef[k, 2] * xm ** 2) + coef[k, 3] * xm ** 3
    return rho


def slant_depth(vertex, zenith, azimuth, model, step=STEP, return_samples=False):
    """PREM.slant_depth (:183-240) for the chord from `vertex` towards (zenith, azimuth)"""
    R = model[0]
    d = np.array([np.sin(zenith) * np.cos(azimuth), np.sin(zenith) * np.sin(azimuth), np.cos(zenith)])
    d = d / np.sqrt(_dot3(d, d))
    e = np.array([vertex[0], vertex[1], vertex[2] + R])
    dot = _dot3(e, d)
    disc = dot ** 2 - ((e[0] ** 2 + e[1] ** 2) + e[2] ** 2) + R ** 2
    if disc <= 0:
        return 0.
    distance = -dot + np.sqrt(disc)
    if distance <= 0:
        return 0.
    n_steps = int(distance / step)
    if distance % step:
        n_steps += 1
    if n_steps > 1:
        ts = np.arange(n_steps) * (1. / (n_steps - 1))   # np.linspace(0, 1, n): arange * step, last sample set to stop
        ts[-1] = 1.
    else:
        ts = np.zeros(1)
    xs = e[0] + ts * distance * d[0]
    ys = e[1] + ts * distance * d[1]
    zs = e[2] + ts * distance * d[2]
    rs = np.sqrt(xs ** 2 + ys ** 2 + zs ** 2)
    y = density(rs, model) * distance
    res = float(np.sum(np.diff(ts) * (y[1:] + y[:-1]) / 2.0))
    if return_samples:
        return res, rs, n_steps, distance
    return res


def get_weight(zenith, azimuth, energy, flavor, vertex, mode, step=STEP, proton_mass_kg=PROTON_MASS_KG, cross_section_type='ctw'):
    """get_weight (:12-60) for arrays of events, cross_section_type 'ctw' or 'ghandi'"""
    zenith, azimuth, energy = (np.atleast_1d(np.asarray(a, dtype=float)) for a in (zenith, azimuth, energy))
    flavor = np.atleast_1d(flavor)
    n = len(zenith)
    w = np.ones(n)
    if mode in ('None', None):
        return w
    if mode == 'simple':                                   # :63-86, flavors = 0 -> particle cross section
        sigma = total_cross_section(energy, np.zeros(n, dtype=int), cross_section_type)
        for i in range(n):
            if zenith[i] > 0.5 * np.pi:
                dd = -2 * (6357390 * 1.) * np.cos(zenith[i])
                w[i] = np.exp(-dd * sigma[i] * (2900 * KG / 1. ** 3) / AMU)
        return w
    if mode == 'core_mantle_crust_simple':                 # :89-130
        RE = 6.378140e6
        dens = np.array([14000.0, 3400.0, 2900.0]) * KG / 1. ** 3
        radii = np.array([3.46e6, RE - 4.0e4, RE])
        sigma = total_cross_section(energy, flavor, cross_section_type)
        for i in range(n):
            th, s = zenith[i], sigma[i]
            if th <= 0.5 * np.pi:
                continue
            if th <= np.pi - np.arcsin(radii[1] / radii[2]):
                d_outer = -2 * RE * np.cos(th)
                w[i] = np.exp(-d_outer * s * dens[2] / AMU)
            elif th <= np.pi - np.arcsin(radii[0] / radii[2]):
                d_middle = 2 * np.sqrt(radii[1] * radii[1] - radii[2] * radii[2] * np.sin(np.pi - th) * np.sin(np.pi - th))
                d_outer = -2 * RE * np.cos(th) - d_middle
                w[i] = np.exp(-d_outer * s * dens[2] / AMU - d_middle * s * dens[1] / AMU)
            else:
                d_inner = 2 * np.sqrt(radii[0] * radii[0] - radii[2] * radii[2] * np.sin(np.pi - th) * np.sin(np.pi - th))
                d_middle = 2 * np.sqrt(radii[1] * radii[1] - radii[2] * radii[2] * np.sin(np.pi - th) * np.sin(np.pi - th)) - d_inner
                d_outer = -2 * RE * np.cos(th) - d_middle - d_inner
                w[i] = np.exp(-d_outer * s * dens[2] / AMU - d_middle * s * dens[1] / AMU - d_inner * s * dens[0] / AMU)
        return w
    model = earth_model(mode)
    vertex = np.asarray(vertex, dtype=float).reshape(n, 3)
    L = interaction_length_unit_density(energy, flavor, proton_mass_kg, cross_section_type)
    for i in range(n):
        w[i] = np.exp(-slant_depth(vertex[i], zenith[i], azimuth[i], model, step) / L[i])
    return w
