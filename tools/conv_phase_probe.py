"""Where channel_conv_kernel spends its shader clocks (fill / field transform / placement / 8192-point transforms / multiply).

    NRHIP_LIB_NAME=libnrhip_ct.so ./build.sh -DNRHIP_CONV_TIMING      # a variant library with the clock instrumentation
    python tools/conv_phase_probe.py                                   # on the GPU box: bench.py's config 2, 3 steps

The clocks are s_memtime differences of wave 0 of every block, summed over blocks and launches."""
import contextlib
import ctypes
import io
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
os.environ['NRHIP_LIB_NAME'] = 'libnrhip_ct.so'
sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-end-to-end'] + sys.argv[1:]   # e.g. --no-traces, --config 5
import bench  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], d['config']['stage_ms_avg_per_step'])
h = ctypes.CDLL(os.path.join(ROOT, 'nuradiomc_amd', 'lib', 'libnrhip_ct.so'))
out = (ctypes.c_ulonglong * 16)()
assert h.nrhip_debug_conv_clocks(out, 0) == 0
names = ['skip / control', 'zero S', 'amplitude fill', 'field transform', 'placement', 'zero pad', '8192-pt forward', 'x G', '8192-pt inverse',
         'maximum / flags']
tot = float(sum(out[:10])) + float(out[11]) + float(out[12])
for n, v in zip(names, out[:10]):
    print('%-16s %6.2f %%  %.3e clk' % (n, 100 * v / tot, v))
print('  output loop %.2f %% (maximum / flags above: what follows it)' % (100 * out[12] / tot))
print('  job list (of the time before the transforms) %.2f %%' % (100 * out[11] / (tot + out[11])))
print('channel transforms', d['config']['n_channel_transforms'], 'ray transforms', d['config']['n_ray_transforms'])
print('amp_bound tiles of 4 rays: %d, decided by the two-sided group sums: %d' % (out[13], out[14]))
