"""Golden vectors of the 'hedis_bgr18' arithmetic (NuRadioMC/utilities/cross_sections.py:17-61, :276-299, :424-537) computed by the
reference in the build container:   PYTHONPATH=/root/reference:tests/golden/gen/shims python tests/golden/gen/gen_hedis.py

The reference's data file (BGR18_dsigma_dy_H2O.npz) is a download and is not here; the reference is therefore run on a SYNTHETIC
table of the same layout (tests/golden/bgr18_synthetic.npz, written by this script: 6 flavors x (nc, cc) x 15 energies x 30 y nodes,
a smooth made-up shape with a few exact zeros) by redirecting the one np.load / os.path.exists of that file name.  Outputs ->
tests/golden/ref_hedis.npz: the reference's integrate_pwpl on the table and on a few extra rows, and get_nu_cross_section for random
(energy, flavor, inttype)."""
import os
import numpy as np
from NuRadioMC.utilities import cross_sections as cs

here = os.path.dirname(__file__)
table = os.path.join(here, '..', 'bgr18_synthetic.npz')
rng = np.random.default_rng(18)
flavors = np.array([12, -12, 14, -14, 16, -16])
kinds = np.array(['NC', 'CC'])
e = np.logspace(13, 21, 15)                       # eV
y = np.concatenate([np.logspace(-5, -1, 18), np.linspace(0.15, 0.98, 12)])
sig0 = 1.5e-31 * (e / 1e18) ** 0.36               # cm^2 per molecule, made up (the order of the real values)
d = np.empty((6, 2, len(e), len(y)))
for i in range(6):
    for j in range(2):
        p = -0.3 - 0.05 * i - 0.1 * j + 0.02 * np.log10(e / 1e13)[:, None]
        d[i, j] = (1 + 1.7 * j) * (1 - 0.1 * (i % 2)) * sig0[:, None] * y[None, :] ** p * (1 - y[None, :]) ** (1.5 - 0.5 * j) \
            * np.exp(0.05 * rng.standard_normal((len(e), len(y))))
d[4, 1, 0, :3] = 0.                                # nu_tau CC below threshold: exact zeros at low y
d[5, 1, 0, -2:] = 0.
np.savez_compressed(table, dsigma_dy_ref=d, flavors_ref=flavors, nu_energies_ref=e, y_ref=y, ncccs_ref=kinds)

real_load, real_exists = np.load, os.path.exists
is_table = lambda p: str(p).endswith('BGR18_dsigma_dy_H2O.npz')
np.load = lambda p, *a, **k: real_load(table if is_table(p) else p, *a, **k)
os.path.exists = lambda p: True if is_table(p) else real_exists(p)

n = 300
energy = 10 ** rng.uniform(13, 21, n)
energy[:3] = [1e13, 1e21, e[7]]
flavor = rng.choice(flavors, n)
inttype = rng.choice(np.array(['cc', 'nc', 'total']), n)
sigma = cs.get_nu_cross_section(energy, flavor, inttype, 'hedis_bgr18')
sigma_scalar = cs.get_nu_cross_section(3e17, 14, 'total', 'hedis_bgr18')
np.load, os.path.exists = real_load, real_exists

full = cs.integrate_pwpl(d * 1e-4 / 18, y, low=0, high=1)
inner = cs.integrate_pwpl(d, y)
part = cs.integrate_pwpl(d[0, 0], y, low=1e-7, high=0.99)
# flat and 1/x rows: slope 0 and slope -1 (the reference's formula divides by slope + 1: its -1 row is not finite and is not kept)
x2 = np.linspace(1., 3., 9)
rows = np.stack([np.full(9, 2.), x2 ** 2, x2 ** -2.5])
rows_int = cs.integrate_pwpl(rows, x2)
np.savez_compressed(os.path.join(here, '..', 'ref_hedis.npz'), energy=energy, flavor=flavor, inttype=inttype, sigma=sigma,
                    sigma_scalar=sigma_scalar, full=full, inner=inner, part=part, x2=x2, rows=rows, rows_int=rows_int)
print(sigma[:4], sigma_scalar, rows_int, np.isfinite(full).all(), (full > 0).all())
