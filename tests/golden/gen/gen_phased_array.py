"""Golden vectors for the phased-array trigger core, produced by the reference's NuRadioReco/modules/phasedarray/
phasedArrayBase.py: calculate_time_delays (:58-124, beam rolls from antenna depths, cable delays, phasing angles),
phase_signals (:183-215, sum of np.roll-ed channel traces per beam) and power_sum (:217-271, sliding power windows),
with the decision of phased_trigger (:455-496: any window above the threshold), no ADC digitisation, no upsampling.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_phased_array.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioReco.modules.phasedarray.phasedArrayBase import PhasedArrayBase  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


class Det:
    def __init__(self, pos, cable): self.pos, self.cable = pos, cable
    def get_relative_position(self, sid, ch): return self.pos[ch]
    def get_cable_delay(self, sid, ch): return self.cable[ch]


class Sta:
    def get_id(self): return 11


rng = np.random.default_rng(21)
cases = []
out = {}
for k, (n_ch, n_samples, fs, window, step, n_beams) in enumerate([(4, 512, 2.0, 32, 16, 11), (4, 1024, 0.5, 4, 2, 7), (8, 774, 2.4, 24, 8, 15)]):
    z = -100. + np.sort(rng.uniform(-8., 0., n_ch))[::-1]
    pos = np.stack([np.zeros(n_ch), np.zeros(n_ch), z], axis=1)
    cable = rng.uniform(0., 6., n_ch)
    chans = list(range(n_ch))
    angles = np.arcsin(np.linspace(np.sin(-60 * np.pi / 180), np.sin(60 * np.pi / 180), n_beams))
    pa = PhasedArrayBase()
    pa.begin()
    rolls = pa.calculate_time_delays(Sta(), Det(pos, cable), chans, phasing_angles=angles, ref_index=1.75, sampling_frequency=fs)
    roll_arr = np.array([[r[c] for c in chans] for r in rolls])
    ev = []
    for e in range(6):
        traces = {c: rng.normal(0, 1., n_samples) + (8. if e % 2 else 0.) * np.exp(-0.5 * ((np.arange(n_samples) - 200 - 3 * c) / 3.) ** 2)
                  for c in chans}
        phased = pa.phase_signals(traces, rolls)
        powers = [pa.power_sum(coh_sum=p, window=window, step=step)[0] for p in phased]
        ev.append((np.array([traces[c] for c in chans]), np.array(phased), np.array(powers)))
    out['traces_%d' % k] = np.array([x[0] for x in ev])
    out['power_%d' % k] = np.array([x[2] for x in ev])
    out['rolls_%d' % k] = roll_arr
    out['pos_%d' % k], out['cable_%d' % k], out['angles_%d' % k] = pos, cable, angles
    cases.append((n_ch, n_samples, fs, window, step, n_beams))
    print(k, roll_arr.min(), roll_arr.max(), out['power_%d' % k].max())
out['cases'] = np.array(cases, float)
out['ref_index'] = 1.75
np.savez_compressed(os.path.join(OUT, 'ref_phased_array.npz'), **out)
