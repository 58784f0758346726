import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd
ctx = nuradiomc_amd.Context((1.78, 0.423, 77.))
rng = np.random.default_rng(0)
x = rng.normal(size=(4096, 2048)) + 1j * rng.normal(size=(4096, 2048))
for _ in range(3):
    ctx.debug_czt(x, 2648, 2648, -1.)
