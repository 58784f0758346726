"""ARZ on the GPU vs the oracle's C loop: n (shower, ray) pairs, N samples at 1 / dt (usage: arz_probe.py [n] [N] [dt])."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from nuradiomc_amd import arz
from oracle import arz_oracle
from test_oracle_golden import _arz_library
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ref_arz.npz'))
lib = _arz_library(g)
n, N, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rng = np.random.default_rng(1)
types = [['HAD', 'EM'][i % 2] for i in range(n)]
E = 10 ** rng.uniform(17., 19., n)
th = np.arccos(1 / 1.78) + rng.uniform(-15, 15, n) * np.pi / 180
R = 10 ** rng.uniform(2., 3.5, n)
a = arz.ARZ(seed=5, library=lib)
iN = a.draw_profile_numbers(E, types)
a.get_time_trace_batch(E[:10], th[:10], N, dt, types[:10], 1.78, R[:10], iN[:10])
t = time.time()
tr = a.get_time_trace_batch(E, th, N, dt, types, 1.78, R, iN)
t_gpu = time.time() - t
o = arz_oracle.ARZ(lib, seed=5)
m = min(n, 40)
t = time.time()
for i in range(m):
    ref = o.get_time_trace(E[i], th[i], N, dt, types[i], 1.78, R[i], iN=int(iN[i]))
t_cpu = (time.time() - t) / m
print('GPU: %d traces of %d samples in %.3f s (host buffers, PCIe included) = %.0f traces/s; oracle C loop %.1f ms/trace = %.0f traces/s/core; last max rel dev %.2e'
      % (n, N, t_gpu, n / t_gpu, t_cpu * 1e3, 1 / t_cpu, np.max(np.abs(tr[m - 1] - ref)) / max(np.max(np.abs(ref)), 1e-300)))
