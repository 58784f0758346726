"""Golden vectors for the trigger primitives, produced by the reference's own functions.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_trigger.py

get_high_low_triggers / get_majority_logic (NuRadioReco/modules/trigger/highLowThreshold.py:13-150) and
get_threshold_triggers (trigger/simpleThreshold.py:14-29) on the channel traces of tests/golden/chain_N256.npz plus
band-limited noise, for several parameter sets.
"""
import os
import numpy as np
from NuRadioReco.modules.trigger.highLowThreshold import get_high_low_triggers, get_majority_logic
from NuRadioReco.modules.trigger.simpleThreshold import get_threshold_triggers

HERE = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(HERE, '..', 'chain_N256.npz'), allow_pickle=True)
vrms = float(g['vrms'])
rng = np.random.default_rng(5)
traces = [g['V_concat'][:, g['V_offsets'][i]:g['V_offsets'][i + 1]].copy() for i in range(len(g['V_events']))]
for _ in range(6):  # noise-like events of odd lengths
    n = int(rng.integers(300, 700)) * 2
    traces.append(np.cumsum(rng.normal(0, vrms, (5, n)), axis=1) * 0.2 + rng.normal(0, 1.5 * vrms, (5, n)))
params = [dict(kind='high_low', high=2.0, low=-2.0, hl_win=5., coinc=30., ncoinc=2),
          dict(kind='high_low', high=3.0, low=-3.0, hl_win=5., coinc=200., ncoinc=1),
          dict(kind='high_low', high=1.5, low=-2.5, hl_win=3., coinc=10., ncoinc=3),
          dict(kind='high_low', high=2.0, low=-2.0, hl_win=5., coinc=5000., ncoinc=2),   # window longer than the trace
          dict(kind='simple', thr=3.0, coinc=200., ncoinc=1),
          dict(kind='simple', thr=2.0, coinc=20., ncoinc=3)]
fs = 2.0
out = dict(vrms=vrms, fs=fs, n_traces=len(traces), params=np.array([repr(p) for p in params]))
for it, V in enumerate(traces):
    out['V_%d' % it] = V
    for ip, p in enumerate(params):
        if p['kind'] == 'high_low':
            flags = [get_high_low_triggers(v, p['high'] * vrms, p['low'] * vrms, p['hl_win'], 1. / fs) for v in V]
        else:
            flags = [get_threshold_triggers(v, p['thr'] * vrms) for v in V]
        out['flags_%d_%d' % (it, ip)] = np.array(flags)
        trig, bins, times = get_majority_logic([f.copy() for f in flags], p['ncoinc'], p['coinc'], 1. / fs)
        out['trig_%d_%d' % (it, ip)] = bool(trig)
        out['bins_%d_%d' % (it, ip)] = np.asarray(bins, np.int64)
np.savez_compressed(os.path.join(HERE, '..', 'ref_trigger.npz'), **out)
print('wrote', len(traces), 'events x', len(params), 'parameter sets;', sum(bool(out['trig_%d_%d' % (i, j)]) for i in range(len(traces)) for j in range(len(params))), 'triggered')
