"""Measured oracle <-> reference residuals per fixture (the project's parity numbers; CPU only, a minute):

    python tools/parity_residuals.py > profiles/r02_parity_residuals.txt

Ray tracing: the oracle's table against the reference's pure-Python path on the three survey geometries (tests/golden/raytrace_*.npz)
and against the reference's own golden files.  Chain: the oracle's whole spectral chain on the reference's launch parameters
against the reference's amplitudes / traces (tests/golden/chain_*.npz).  The tests assert bounds; this prints the maxima observed.
The GPU equals the oracle bit for bit in the ray tables and to <= 1e-9 in the traces (tests/test_gpu_*.py)."""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import golden, max_rel   # noqa: E402
from oracle import raytrace_oracle as orc   # noqa: E402
from oracle import spectral_oracle as so   # noqa: E402
import test_oracle_chain as tc   # noqa: E402

print('# ray tracing, oracle vs reference Python path (pairs with equal solution counts)')
print('fixture      pairs  count_mismatch  types  max_rel_C0  max_rel_D  frac(D>1e-6)  max_rel_T  max_abs_launch  max_abs_receive  max_abs_C1[m]')
for name in 'ABC':
    g = golden('raytrace_%s.npz' % name)
    o = orc.raytrace_batch(g['x1'], g['x2'], g['ice'])
    bad = o['n_sol'] != g['n_sol']
    ok = ~bad
    relD = np.abs(o['D'][ok] - g['D'][ok]) / np.abs(g['D'][ok])
    relT = np.abs(o['T'][ok] - g['T'][ok]) / np.abs(g['T'][ok])
    relD, relT = relD[np.isfinite(relD)], relT[np.isfinite(relT)]
    print('raytrace_%s %6d  %8.4f %%     %s  %.2e    %.2e   %.4f %%      %.2e   %.2e        %.2e         %.2e' % (
        name, len(bad), 100 * bad.mean(), 'equal' if np.array_equal(o['type'][ok], g['type'][ok]) else 'DIFFER',
        max_rel(o['C0'][ok], g['C0'][ok]), relD.max(), 100 * (relD > 1e-6).mean(), relT.max(),
        np.nanmax(np.abs(o['launch'][ok] - g['launch'][ok])), np.nanmax(np.abs(o['receive'][ok] - g['receive'][ok])),
        np.nanmax(np.abs(o['C1'][ok] - g['C1'][ok]))))

print()
print('# spectral chain on the reference\'s launch parameters, oracle vs reference (relative to the largest value of the trace / event)')
print('fixture      rays  max_rel_max_efield  max_rel_amp_per_ray  max_rel_spectrum  events  candidates  triggers  decisions  max_rel_maxV  max_rel_trace')
for name in ['N256', 'N256_hpol', 'N256_lpda', 'N256_tab', 'N4096', 'N256_hw']:
    g = golden('chain_%s.npz' % name)
    st = tc._station(g)
    ice = g['ice']
    filters = so.DEFAULT_FILTERS
    if 'hw_amp' in g:
        filters = tc.hw_filters(g)
    vrms, vrms_e = float(g['vrms']), float(g['vrms_efield'])
    full = {int(k): i for i, k in enumerate(g['full_ray_index'])}
    vev = {int(e): i for i, e in enumerate(g['V_events'])}
    n_events = len(g['vertex']) if name != 'N4096' else 40
    m_e = m_a = m_s = m_v = m_t = 0.
    n_r = n_c = n_t = n_ev = 0
    same = True
    for ev in range(n_events):
        rays, sel = tc._rays_with_reference_launch_parameters(g, ev, st, ice)
        if rays is None:
            continue
        k_L = None if np.isnan(g['ev_k_L'][ev]) else float(g['ev_k_L'][ev])
        if str(g['shower_type'][ev]) == 'EM' and k_L is None:
            continue
        o = so.simulate_event(g['vertex'][ev], g['zenith'][ev], g['azimuth'][ev], g['energy'][ev], str(g['shower_type'][ev]), k_L, st,
                              ice, vrms, vrms_e, rays=rays, filters=filters)
        n_ev += 1
        for r, k in zip(o['rays'], sel):
            m_e = max(m_e, abs(r['max_efield'] - g['ray_max_efield'][k]) / g['ray_max_efield'][k])
            m_a = max(m_a, abs(r['max_amp_ray'] - g['ray_max_amp_ray'][k]) / g['ray_max_amp_ray'][k])
            if int(k) in full:
                ref = g['full_spec'][full[int(k)]]
                m_s = max(m_s, np.max(np.abs(r['spec'][1:] - ref)) / np.max(np.abs(ref)))
            n_r += 1
        same = same and o['candidate'] == bool(g['ev_candidate'][ev]) and o['triggered'] == bool(g['ev_triggered'][ev])
        if o['candidate']:
            n_c += 1
            same = same and o['L'] == int(g['ev_L'][ev])
            mv = np.max(np.abs(o['V']), axis=1)
            m_v = max(m_v, np.max(np.abs(mv - g['ev_maxV'][ev])) / np.max(g['ev_maxV'][ev]))
            if ev in vev:
                i = vev[ev]
                ref = g['V_concat'][:, g['V_offsets'][i]:g['V_offsets'][i + 1]]
                m_t = max(m_t, np.max(np.abs(o['V'] - ref)) / np.max(np.abs(ref)))
        n_t += o['triggered']
    print('chain_%-9s %4d  %.2e            %.2e             %.2e          %4d   %4d        %4d      %s      %.2e      %.2e' % (
        name, n_r, m_e, m_a, m_s, n_ev, n_c, n_t, 'equal' if same else 'DIFFER', m_v, m_t))
