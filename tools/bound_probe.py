"""How loose is the depth-binned attenuation bound?  For the rays of the bench list that amp_bound_kernel cannot prune but whose
field the bound with the COMPUTED attenuation (efield_bound_kernel) then decides: ratio of the two bounds.   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nuradiomc_amd, bench
n = 200000
wl = bench.make_workload(2, n, 10)
ctx = nuradiomc_amd.Context(wl['ice'], wl['att_model'])
st = bench.build_array(ctx, wl)
ev = wl['events']
trig, stats = st.simulate_events(ev['vertex'], ev['zenith'], ev['azimuth'], ev['energy'], ev['shower_type'], ev['k_L'])
nr = stats['n_rays']
b = st.fetch('ray_bound')[:nr]; mx = st.fetch('ray_max_efield')[:nr]; act = st.fetch('ray_active')[:nr].astype(bool)
att = st.fetch('ray_att').reshape(nr, -1)
D = st.fetch('ray_D')[:nr]
cut = 2.0 * st.vrms_efield
print('rays', nr, 'active', act.sum(), 'own bound > cut', (b > cut).sum(), 'decided by the exact-attenuation bound (max_efield < 0 among active)', (act & (mx < 0)).sum())
sel = act & (mx < 0) & (b > cut)
r = b[sel] / -mx[sel]
print('ratio amp_bound / bound with computed attenuation: n %d, percentiles 5 25 50 75 95: %s' % (sel.sum(), np.percentile(r, [5, 25, 50, 75, 95]).round(3)))
tau = -np.log(att[sel][:, 6])
print('tau at coarse frequency 6 (%.3f GHz): percentiles %s' % (st.att_freq[6], np.percentile(tau, [5, 25, 50, 75, 95]).round(2)))
print('path length D percentiles', np.percentile(D[sel], [5, 50, 95]).round(0))
print('how far above the cut is the amp_bound for these:', np.percentile(b[sel] / cut, [5, 25, 50, 75, 95]).round(3))
typ = st.fetch('slot_type')[st.fetch('ray_slot2')[:nr]] if False else None
slot = st.fetch('ray_slot')[:nr]
styp = st.fetch('slot_type')[slot]
for t in (1, 2, 3):
    m = sel & (styp == t)
    if m.sum():
        rr = b[m] / -mx[m]
        print('type', t, 'n', m.sum(), 'ratio percentiles 5 50 95', np.percentile(rr, [5, 50, 95]).round(3), 'tau(0.25) median', np.median(-np.log(att[m][:, 6])).round(2))
h, e = np.histogram(r, bins=[1, 1.05, 1.1, 1.2, 1.4, 1.6, 1.8, 1.95, 2.05, 2.2, 2.5, 3, 10])
print('ratio histogram', list(zip(e[:-1].round(2), h)))
# the same ratio against what the margin alone explains
tau_all = -np.log(np.clip(att[sel], 1e-300, 1))
print('exp(0.05 tau) at 0.25 GHz percentiles', np.percentile(np.exp(0.05 * tau_all[:, 6]), [5, 50, 95]).round(3))
zen = st.fetch('ray_zenith')[:nr]
print('ratio ~2 population: type counts', np.unique(styp[sel][(r > 1.9)], return_counts=True), 'D median', np.median(D[sel][r > 1.9]))
pt, pp = st.fetch('ray_pol_theta')[:nr], st.fetch('ray_pol_phi')[:nr]
print('pol_theta, pol_phi of ratio~2 population (median abs)', np.median(np.abs(pt[sel][r > 1.9])), np.median(np.abs(pp[sel][r > 1.9])), ' others', np.median(np.abs(pt[sel][r < 1.2])), np.median(np.abs(pp[sel][r < 1.2])))
