"""Golden vectors of the tabulated 'csms' cross section (NuRadioMC/utilities/cross_sections.py:123-229, :381-382) computed by the
reference in the build container:   PYTHONPATH=/root/reference:tests/golden/gen/shims python tests/golden/gen/gen_csms.py
-> tests/golden/ref_csms.npz (energy, flavor, is_cc, sigma per interaction type, and what inttype='total' returns)."""
import os
import numpy as np
from NuRadioMC.utilities import cross_sections as cs

rng = np.random.default_rng(5)
n = 400
energy = 10 ** rng.uniform(np.log10(50e9), np.log10(5e20), n)
energy[:4] = [50e9, 5e20, 1e18, 2e15]          # table ends, nodes
flavor = rng.choice([12, -12, 14, -14, 16, -16], n)
is_cc = rng.random(n) < 0.5
sigma = cs.get_nu_cross_section(energy, flavor, np.where(is_cc, 'cc', 'nc'), 'csms')
sigma_total = cs.get_nu_cross_section(energy, flavor, 'total', 'csms')
out = os.path.join(os.path.dirname(__file__), '..', 'ref_csms.npz')
np.savez_compressed(out, energy=energy, flavor=flavor, is_cc=is_cc, sigma=sigma, sigma_total=sigma_total)
print(out, sigma[:4], sigma_total[:4])
