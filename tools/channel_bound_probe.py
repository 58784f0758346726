"""How tight is the channel prefilter?  (CPU only; the checker's spectral chain, no GPU.)

channel_prefilter_kernel bounds a ray's contribution to a channel trace by Cauchy-Schwarz, ||e||_2 ||g||_2.  This probe takes the rays
of bench.py's config-2 list that pass the candidate cut and compares, per ray, with the true maximum of the ray's voltage trace:
  * Cauchy-Schwarz                          ||e||_2 ||g||_2
  * Young                                   ||e||_1 ||g||_inf          (with the exact ||e||_1)
  * sum of magnitudes in the frequency domain   (1 / N) sum_k |E_k| |G_k|
Result (profiles/r06_channel_bound_probe.txt): medians 2.26 / 8.5 / 1.22 of the true maximum -- four in five channel transforms of
the trigger pass compute a channel that does not trigger; a rigorous form of the third bound on the event's L grid is the next
lever of the channel stage.

    python tools/channel_bound_probe.py [n_events]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from oracle import spectral_oracle as so  # noqa: E402  (diagnostic tool, not the product)

n_events = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
st = so.Station(bench.CHANNELS, n_samples=4096, fs=2.0)
v, z, a = bench.make_events(n_events, 1)
vrms, vrms_e = so.vrms_from_filters(2.0)
ff = np.fft.rfftfreq(st.n_samples, 1. / st.fs)
H = so.filter_response(ff, so.DEFAULT_FILTERS)
rows = []
for i in range(len(v)):
    for ef in so.sim_efields_for_event(v[i], z[i], a[i], 1e18, 'HAD', None, st, bench.ICE):
        tr = so.freq2time(ef['spec'], st.fs)
        if np.max(np.abs(tr)) < 2.0 * vrms_e:   # below the candidate cut
            continue
        Vt, Vp = so.antenna_response(st.antenna_of(ef['channel']), ff, ef['zenith'], ef['azimuth'], st.orientation[ef['channel']])
        G = Vt * H
        G[ff < 0.005] = 0
        g = so.freq2time(G, st.fs)
        e = tr[1]   # (vertical dipoles: the theta component carries the signal)
        y = np.real(np.fft.ifft(np.fft.fft(e) * np.fft.fft(g)))
        m = np.max(np.abs(y))
        rows.append((np.linalg.norm(e) * np.linalg.norm(g) / m, np.sum(np.abs(e)) * np.max(np.abs(g)) / m,
                     np.sum(np.abs(np.fft.fft(e) * np.fft.fft(g))) / len(e) / m))
    if len(rows) > 400:
        break
r = np.array(rows)
print('rays above the candidate cut: %d (of the first %d events of the config-2 list)' % (len(r), i + 1))
for name, col in (('Cauchy-Schwarz ||e||_2 ||g||_2', 0), ('Young ||e||_1 ||g||_inf', 1), ('sum_k |E_k| |G_k| / N', 2)):
    q = np.quantile(r[:, col], [0.1, 0.5, 0.9])
    print('%-32s / true maximum: 10 %% %.2f  median %.2f  90 %% %.2f' % (name, q[0], q[1], q[2]))
