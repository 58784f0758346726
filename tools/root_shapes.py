"""The two facts the solution finder without the hybr stage rests on (oracle/nrmc_oracle.c find_solutions_bracketed, csrc/raytrace.hip
raytrace_roots_fast_kernel), checked numerically: for receivers down to 10 z_0, as functions of the launch parameter log C0 above
x_lo (the ray that turns at the receiver's depth),
    u = x2.y - y(z2)                 rises monotonically,
    v = (2 y_turn - y(z2)) - x2.y    has exactly one maximum,
so that delta_y = min(u, v) is positive on at most ONE interval.  Random pairs in three ice models, 3000 grid points each.
    python tools/root_shapes.py [pairs per model, default 20000]      (CPU; test infrastructure: imports oracle/)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import raytrace_oracle as rto   # noqa: E402


def check(n_pairs=20000, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    total = dict(pairs=0, u_not_monotone=0, v_not_unimodal=0, more_than_one_interval=0)
    for ice in [(1.78, 0.423, 77.), (1.78, 0.51, 37.25), (1.78, 0.46, 34.5)]:
        n_ice, dn, z0 = ice
        for _ in range(n_pairs):
            rho = np.sqrt(rng.uniform(0, 5000. ** 2))
            z1, z2 = rng.uniform(-2700, -0.5), rng.uniform(-10 * z0, -0.5)
            if z2 < z1:
                z1, z2 = z2, z1
            if z2 < -10 * z0:
                continue
            x_lo = np.log(1 / (n_ice - dn * np.exp(z2 / z0)) - 1 / n_ice)
            x = np.concatenate([x_lo + np.linspace(3e-5, np.sqrt(3 - x_lo), 2800) ** 2, np.linspace(3, 100, 200)[1:]])
            u, v = rto.uv_grid(x, (0., z1), (rho, z2), ice)
            assert np.all(np.isfinite(u)) and np.all(np.isfinite(v))
            total['pairs'] += 1
            scale = max(np.abs(u).max(), np.abs(v).max())
            total['u_not_monotone'] += bool(np.any(np.diff(u) < -1e-9 * scale))
            dv = np.diff(v)
            sg = np.sign(dv[np.abs(dv) > 1e-9 * scale])
            total['v_not_unimodal'] += bool((np.diff(sg) != 0).sum() > 1)
            pos = np.minimum(u, v) > 0
            total['more_than_one_interval'] += bool((np.diff(pos.astype(int)) == 1).sum() + int(pos[0]) > 1)
    if verbose:
        print(total)
    return total


if __name__ == '__main__':
    t = check(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
    assert t['u_not_monotone'] == 0 and t['v_not_unimodal'] == 0 and t['more_than_one_interval'] == 0
