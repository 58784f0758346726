"""Golden vectors for the per-event Earth-absorption weight, produced by the reference's
NuRadioMC/utilities/earth_attenuation.py: get_weight (:12-60) in its four modes -- 'simple' (:63-86),
'core_mantle_crust_simple' (:89-130), 'core_mantle_crust' and 'PREM' (slant depth of the chord from the vertex
towards the arrival direction, :183-240, 500 m trapezoid rule) -- with the 'ctw' cross sections the reference's default
configuration uses (cross_sections.py:64-120, :232-391) and the interaction length of :393-421.  The call site is
simulation.py:880-903 (one call per event group, scalar arguments).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_earth_weights.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.utilities import earth_attenuation, cross_sections  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
rng = np.random.default_rng(77)
n = 400
zenith = np.arccos(rng.uniform(-1., 1., n))
zenith[:6] = [0., 0.5 * np.pi, np.pi, 0.5 * np.pi + 1e-9, 2.0, 3.0]
azimuth = rng.uniform(0., 2 * np.pi, n)
energy = 10 ** rng.uniform(16., 20., n) * units.eV
flavor = rng.choice(np.array([12, -12, 14, -14, 16, -16]), n)
r = 4000. * np.sqrt(rng.uniform(0., 1., n))
a = rng.uniform(0., 2 * np.pi, n)
vertex = np.stack([r * np.cos(a), r * np.sin(a), -rng.uniform(1., 2700., n)], axis=1)

out = dict(zenith=zenith, azimuth=azimuth, energy=energy, flavor=flavor, vertex=vertex)
for mode in ['simple', 'core_mantle_crust_simple', 'core_mantle_crust', 'PREM']:
    w = np.zeros(n)
    for i in range(n):
        w[i] = earth_attenuation.get_weight(float(zenith[i]), float(energy[i]), int(flavor[i]), mode=mode,
                                            cross_section_type='ctw', vertex_position=vertex[i].copy(),
                                            phi_nu=float(azimuth[i]))
    out['weight_' + mode] = w
# 'ghandi' cross section (cross_sections.py:280-281: one power law for all flavours and interactions)
for mode in ['simple', 'core_mantle_crust_simple', 'core_mantle_crust', 'PREM']:
    out['weight_ghandi_' + mode] = np.array([
        earth_attenuation.get_weight(float(zenith[i]), float(energy[i]), int(flavor[i]), mode=mode, cross_section_type='ghandi',
                                     vertex_position=vertex[i].copy(), phi_nu=float(azimuth[i])) for i in range(n)])
# the ingredients, for pinning the restatement piece by piece
out['sigma_total'] = np.array([cross_sections.get_nu_cross_section(float(e), int(f), inttype='total', cross_section_type='ctw')
                               for e, f in zip(energy, flavor)])
out['L_int_unit_density'] = np.array([cross_sections.get_interaction_length(float(e), density=1., flavor=int(f), inttype='total',
                                                                           cross_section_type='ctw') for e, f in zip(energy, flavor)])
for name, model in [('core_mantle_crust', earth_attenuation.CoreMantleCrustModel()), ('PREM', earth_attenuation.PREM())]:
    sd = np.zeros(n)
    for i in range(n):
        d = np.array([np.sin(zenith[i]) * np.cos(azimuth[i]), np.sin(zenith[i]) * np.sin(azimuth[i]), np.cos(zenith[i])])
        sd[i] = model.slant_depth(vertex[i].copy(), d)
    out['slant_depth_' + name] = sd
np.savez_compressed(os.path.join(OUT, 'ref_earth_weights.npz'), **out)
for k, v in out.items():
    print(k, v.shape, v.dtype, float(np.nanmin(v)), float(np.nanmax(v)))
