"""Host-side helpers of the drop-in classes that need no GPU."""
import numpy as np
from conftest import golden


def test_ray_path_vs_reference_get_path():
    """nuradiomc_amd.propagation.analytic_ray_path (what ray_tracing.get_path returns) against paths drawn by the reference's
    ray_tracing.get_path (analyticraytracing.py:2148-2162) for its own solutions -- both orders of the end points, direct,
    refracted and reflected rays (tests/golden/gen/gen_paths.py)."""
    from nuradiomc_amd.propagation import analytic_ray_path
    g = golden('ref_paths.npz')
    n = 0
    for i in range(len(g['x1'])):
        for iS in range(g['n_sol'][i]):
            ref = g['path'][i, iS]
            p = analytic_ray_path(g['x1'][i], g['x2'][i], g['C0'][i, iS], *g['ice'], n_points=ref.shape[0])
            assert p.shape == ref.shape and np.max(np.abs(p - ref)) < 1e-4      # metres, paths of ~1 km (observed 3e-6)
            lo, hi = (g['x1'][i], g['x2'][i]) if g['x2'][i][2] >= g['x1'][i][2] else (g['x2'][i], g['x1'][i])
            assert np.max(np.abs(p[0] - lo)) < 1e-6 and np.max(np.abs(p[-1] - hi)) < 1e-3   # ends on the receiver (C0 to 1e-7)
            n += 1
    assert n >= 30
