"""Where attenuation_group_kernel spends its shader clocks: node records (frequency-independent part, 21 lanes of a ray) vs the
per-lane rule evaluation vs everything else (QUADPACK bookkeeping, divergence between the two rays of a wave).

    NRHIP_LIB_NAME=libnrhip_at.so ./build.sh -DNRHIP_ATT_TIMING
    python tools/att_phase_probe.py           # on the GPU box: bench.py's config 2, 3 steps
"""
import contextlib
import ctypes
import io
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
os.environ['NRHIP_LIB_NAME'] = 'libnrhip_at.so'
sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
import bench  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('attenuation stage ms', d['config']['stage_ms_avg_per_step']['attenuation'])
h = ctypes.CDLL(os.path.join(ROOT, 'nuradiomc_amd', 'lib', 'libnrhip_at.so'))
out = (ctypes.c_ulonglong * 12)()
assert h.nrhip_debug_att_clocks(out, 0) == 0
if out[10] and not os.environ.get('NRHIP_ATT_LEGACY'):   # attenuation_dense_kernel's phases
    tot = float(out[10])
    names = ['round: ballots, tree children', 'round: node records', 'round: rules', 'bisection set-up', 'list updates before the sort',
             'sort_errors', 'first estimate: bookkeeping', 'after the rules of a bisection (incl. lists, sort, extrapolation)',
             'final result, stores', 'ray parameters, tree roots']
    for i, nm in enumerate(names):
        print('%-70s %5.1f %%' % (nm, 100 * out[i] / tot))
    print('%-70s %5.1f %%' % ('   of which the epsilon table (copy to LDS, DQELG, copy back)', 100 * out[11] / tot))
    print('%-70s %5.1f %%' % ('unaccounted', 100 * (tot - out[0] - out[1] - out[2] - out[3] - out[6] - out[7] - out[8] - out[9]) / tot))
    raise SystemExit(0)
tot = float(out[2])
print('node records                      %5.1f %%' % (100 * out[0] / tot))
print('rules (per-lane finish)           %5.1f %%' % (100 * out[1] / tot))
print('bisection set-up (end points)     %5.1f %%' % (100 * out[3] / tot))
print('after the rules: list updates     %5.1f %%' % (100 * out[4] / tot))
print('                 sort_errors      %5.1f %%' % (100 * out[5] / tot))
print('                 extrapolation... %5.1f %%' % (100 * (out[6] - out[4] - out[5]) / tot))
print('first estimate: bookkeeping      %5.1f %%' % (100 * out[7] / tot))
print('after the loop: final result      %5.1f %%' % (100 * out[8] / tot))
print('exp(-integral), stores            %5.1f %%' % (100 * out[10] / tot))
print('whole quadrature call             %5.1f %%' % (100 * out[9] / tot))
print('unaccounted inside the call       %5.1f %%' % (100 * (out[9] - out[0] - out[1] - out[3] - out[6] - out[7] - out[8]) / tot))
