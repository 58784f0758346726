"""BASELINE config 3 in the shape the build can run today: an RNO-G like array of 35 stations x 24 channels (7 x 5 grid, 1.25 km
spacing; per station a power string, two helper strings and 9 shallow LPDAs -- analytic antenna models, the measured ones are
downloads), greenland_simple ice + GL1 attenuation, Alvarez2009, 1e18 eV hadronic showers in a cylinder around the array,
speedup.distance_cut with the coefficients of the reference's example config, simple 3 Vrms threshold on any channel, 2048
samples at 2 GHz.  One Station object is moved through the array (Station.move_to); every chunk of the event list is uploaded
once, offered to every station, and the masks are OR-ed.
usage: config3_probe.py [n_events] [n_stations] [gen2 | deep4 | pa | pa_adc]
`gen2` = BASELINE config 5 in the same shape: up to 200 stations on a 1.24 km square grid, each the 5-channel dipole string of
config 2 at -100 .. -104 m (no Gen2 detector file exists in the reference), showers log-uniform in 1e16 .. 1e20 eV."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import nuradiomc_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_st = int(sys.argv[2]) if len(sys.argv) > 2 else 35
d = np.pi / 180
pos, ant, ori = [], [], []
for z in [-95., -96., -97., -98., -80., -60., -40.]:
    pos.append([0., 0., z]); ant.append('analytic_VPol'); ori.append([0., 0., 90 * d, 90 * d])
for z in (-94., -79.):
    pos.append([0., 0., z]); ant.append('analytic_HPol'); ori.append([0., 0., 90 * d, 90 * d])
for x, y in ((-20., 30.), (25., 28.)):
    for z in (-95., -94., -93.):
        pos.append([x, y, z]); ant.append('analytic_VPol' if z != -94. else 'analytic_HPol'); ori.append([0., 0., 90 * d, 90 * d])
for k in range(9):
    a = 2 * np.pi * k / 9
    pos.append([12 * np.cos(a), 12 * np.sin(a), -3.])
    ant.append('analytic_LPDA')
    ori.append([0., 0., 90 * d, (90 + 40 * k) * d] if k % 3 == 0 else [120 * d, a, 90 * d, a + 90 * d])
pos, ori = np.array(pos), np.array(ori)
gen2 = len(sys.argv) > 3 and sys.argv[3] == 'gen2'
centres = np.array([[1250. * (i - 3), 1250. * (j - 2), 0.] for i in range(7) for j in range(5)])[:n_st]
if gen2:
    side = int(np.ceil(np.sqrt(n_st)))
    centres = np.array([[1240. * (i - (side - 1) / 2), 1240. * (j - (side - 1) / 2), 0.] for i in range(side) for j in range(side)])[:n_st]
    pos = np.array([[0., 0., -100. - i] for i in range(5)])
    ant, ori = ['analytic_VPol'] * 5, np.tile([0., 0., 90 * d, 90 * d], (5, 1))
ctx = nuradiomc_amd.Context((1.78, 0.51, 37.25), 'GL1', device=0)
def make_station(c):
    return nuradiomc_amd.Station(ctx, pos + c, antenna=ant, orientation=ori, n_samples=2048, sampling_rate=2.0, att_bound_depth=3000.)
rng = np.random.default_rng(10)
rmax = (np.max(np.abs(centres[:, :2])) if gen2 else 1250. * 3.5) + 3000.
r, ph = np.sqrt(rng.uniform(0, rmax ** 2, n)), rng.uniform(0, 2 * np.pi, n)
vertex = np.stack([r * np.cos(ph), r * np.sin(ph), rng.uniform(-2700., -1., n)], axis=1)
zen, az = np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
energy = 10 ** rng.uniform(16., 20., n) if gen2 else np.full(n, 1e18)
coef = [-1.56434411e+02, 2.54131322e+01, -1.34932379e+00, 2.39984185e-02]   # config_default.yaml:20 speedup.distance_cut_coefficients
chunk = 250000   # events per call: the workspace of a call (ray records, per-ray tables) stays resident in the Station object
any_trig = np.zeros(n, bool)
tot = dict(n_pairs=0, n_rays=0, n_active_rays=0, n_candidate_events=0)
from nuradiomc_amd.station import distance_cut
s = make_station(centres[0])
trig_kw = {}
if len(sys.argv) > 3 and sys.argv[3] == 'deep4':     # threshold trigger on the four deep dipoles of the power string only
    s.set_trigger_channels([0, 1, 2, 3])
if len(sys.argv) > 3 and sys.argv[3] == 'pa':        # phased array on the four deep dipoles (11 beams, 16-sample windows)
    s.set_phased_array([0, 1, 2, 3], np.arcsin(np.linspace(np.sin(-60 * d), np.sin(60 * d), 11)), window=16, step=8)
    trig_kw = dict(trigger='phased_array', trigger_threshold=2.0 * (2 * s.vrms) ** 2)
if len(sys.argv) > 3 and sys.argv[3] == 'pa_adc':    # the same with the trigger ADC (472 MHz, 8 bit, counts) and 4x FFT up-sampling
    s.set_phased_array([0, 1, 2, 3], np.arcsin(np.linspace(np.sin(-60 * d), np.sin(60 * d), 11)), window=24, step=8, upsampling_factor=4,
                       adc=dict(sampling_frequency=0.472, n_bits=8, noise_count=5, output='counts'))
    trig_kw = dict(trigger='phased_array', trigger_threshold=2.0 * (2 * 5) ** 2)
s.simulate_events(vertex[:1000], zen[:1000], az[:1000], energy[:1000], 'HAD', distance_cut_coefficients=coef)
t0 = time.time()
per = np.zeros(len(centres))
for a in range(0, n, chunk):
    sl = slice(a, min(n, a + chunk))
    m = sl.stop - sl.start
    # the shower list of the chunk goes to the GPU once and serves every station; so does the distance cut (a property of
    # the showers); ONE station object is moved through the array (identical stations: only the positions differ)
    md = distance_cut(vertex[sl], energy[sl], None, coef)
    d_in = [ctx.to_device(np.ascontiguousarray(x)) for x in (vertex[sl], zen[sl], az[sl], energy[sl], np.zeros(m, np.int32), np.ones(m))]
    d_md, d_trig = ctx.to_device(md), ctx.malloc(m)
    trig = np.zeros(m, np.uint8)
    for i, c in enumerate(centres):
        t1 = time.time()
        s.move_to(pos + c)
        stats = s.simulate_events_dev(m, *d_in, d_trig, d_max_distance=d_md, **trig_kw)
        ctx.to_host(trig, d_trig)
        any_trig[sl] |= trig.astype(bool)
        for k in tot:
            tot[k] += stats[k]
        per[i] += time.time() - t1
    for p_ in d_in + [d_md, d_trig]:
        ctx.free(p_)
s.close()
dt = time.time() - t0
print(('config 5' if gen2 else 'config 3') + ' (synthetic array): %d events x %d stations x %d channels = %.3g pairs offered, %.3g rays after the distance cut; '
      '%.2f s wall (host arrays in, masks out) = %.0f events/s, %.3g pairs/s; %d events trigger somewhere; per station %.3f .. %.3f s'
      % (n, len(centres), len(pos), n * len(centres) * float(len(pos)), tot['n_rays'], dt, n / dt, n * len(centres) * float(len(pos)) / dt, any_trig.sum(), min(per), max(per)))
print('last station: candidates', stats['n_candidate_events'], 'max L', stats['max_length'], 'stage ms:', stats['stage_ms'])
