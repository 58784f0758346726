"""How much of the attenuation work is spent on rays whose events can never become candidates?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import nuradiomc_amd
import bench
n = 200000
ctx = nuradiomc_amd.Context(bench.ICE, 'SP1')
st = nuradiomc_amd.Station(ctx, bench.CHANNELS, antenna='analytic_VPol', n_samples=4096, sampling_rate=2.0, n_freq=25)
v, z, a = bench.make_events(n, 10)
out = st.simulate_events(v, z, a, np.full(n, bench.ENERGY), np.zeros(n, np.int32), np.ones(n), askaryan_model='Alvarez2009')
stats = out['stats'] if isinstance(out, dict) and 'stats' in out else None
act = st.fetch('ray_active')[:-1].astype(bool)
mx = st.fetch('ray_max_efield'); bnd = st.fetch('ray_bound'); ev = st.fetch('ray_event'); D = st.fetch('ray_D')
att = st.fetch('ray_att').reshape(len(mx), -1)
cand = st.fetch('ev_candidate')
print('rays', len(mx), 'active', act.sum(), 'transformed (exact-att bound passed)', (mx[act] > 0).sum())
# events possible after the exact-attenuation bound: any ray with max_efield > 0 (transform needed)
ev_need = np.zeros(n, bool); ev_need[ev[act & (mx > 0)]] = True
print('events: active', len(np.unique(ev[act])), 'needing a transform', ev_need.sum(), 'candidates', cand.sum())
print('rays of events needing a transform', ev_need[ev].sum())
# how loose is the pre-bound: ratio bound_pre / bound_exact for active rays (exact bound = -mx where mx < 0)
neg = act & (mx < 0)
print('active rays rejected by the exact bound: median pre/exact ratio', np.median(bnd[neg] / -mx[neg]))
print('mean att (over freqs) of active rays: median', np.median(att[act].mean(axis=1)), ' D median', np.median(D[act]))
cut = 2.0 * st.vrms_efield
own = bnd * (1 + 1e-6) > cut
print('cut', cut, 'rays with own pre-bound > cut', own.sum(), ' of which active', (own & act).sum())
evc = cand.astype(bool)[ev]
print('phase-2 rays (in candidate events, own bound <= cut)', (evc & ~own).sum(), 'rays in candidate events', evc.sum())
exact_own = mx > cut
print('rays exceeding the cut for real', exact_own.sum())
