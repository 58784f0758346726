"""Golden vectors for the birefringent pulse propagation (analyticraytracing.py:2165-2445, medium_base.py:378-420).

1. `ref_BF`: the reference's own golden file NuRadioMC/test/SignalProp/reference_BF.npy (T07test_birefringence.py: 10
   vertices, seed 42, receiver at -150 m, delta pulse 50-300 MHz, southpole_2015 ice, birefringence model southpole_A)
   with the inputs T07 regenerates from its seed and the input trace it builds.
2. The same through the reference here (its numerics differ from the stored file at the 1e-4 level, T07:92-96), plus, per
   ray (the first three of each set), get_path_properties_birefringence (path, nx / ny / nz, N1 / N2, the two polarisation vectors, time delays) and
   the final spectra; a second set with `angle_to_iceflow` and model greenland_A on greenland_simple ice.
3. The spline coefficients (t, c, k) of the birefringence models used (data files of the reference,
   NuRadioMC/utilities/birefringence_models/birefringence_*.npy).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_birefringence.py
"""
import os
import sys
import logging
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402
import NuRadioReco.framework.electric_field  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
ref_dir = os.path.join(os.path.dirname(ray.__file__), '..', 'test', 'SignalProp')
model_dir = os.path.join(os.path.dirname(medium.__file__), 'birefringence_models')
out = dict(ref_BF=np.load(os.path.join(ref_dir, 'reference_BF.npy')))
for name in ('southpole_A', 'greenland_A'):
    tck = np.load(os.path.join(model_dir, 'birefringence_%s.npy' % name), allow_pickle=True)
    for j in range(3):
        out['tck_%s_%d_t' % (name, j)] = np.asarray(tck[j][0], float)
        out['tck_%s_%d_c' % (name, j)] = np.asarray(tck[j][1], float)
        assert int(tck[j][2]) == 3

# ---- T07's inputs
np.random.seed(42)
n_events = 10
rr = np.random.triangular(50., 3000., 3000., n_events)
phiphi = np.random.uniform(0, 2 * np.pi, n_events)
zz = np.random.uniform(0., -3000., n_events)
points = np.array([rr * np.cos(phiphi), rr * np.sin(phiphi), zz]).T
x_receiver = np.array([0., 0., -150.])
size, sr = 500, 2.0
delta = np.zeros(size)
delta[size // 2] = 1
ef = NuRadioReco.framework.electric_field.ElectricField([1], position=None, shower_id=None, ray_tracing_id=None)
ef.set_trace(delta, sr)
filt = ef.get_filtered_trace([50 * units.MHz, 300 * units.MHz], filter_type='rectangular')
filt = 1 / np.sqrt(2) * filt / max(filt)
zeros = np.zeros(size)
out.update(points=points, x_receiver=x_receiver, input_trace=filt, sampling_rate=sr)


def run(ice, ice_name, bire_model, pts, rec, angle, tag):
    config = {'propagation': dict(attenuate_ice=False, focusing_limit=2, focusing=False, birefringence=True,
                                  birefringence_model=bire_model, birefringence_propagation='analytical')}
    if angle is not None:
        config['propagation']['angle_to_iceflow'] = angle
    th, ph, recs = [filt], [filt], []
    for iX, x in enumerate(pts):
        r = ray.ray_tracing(ice, log_level=logging.ERROR, use_cpp=False, compile_numba=False)
        r.set_start_and_end_point(x, rec)
        r.find_solutions()
        r.set_config(config)
        for iS in range(r.get_number_of_solutions()):
            ef.set_trace(np.vstack((zeros, filt, filt)), sr)
            final = r.apply_propagation_effects(ef, iS)
            th.append(np.array(final.get_trace()[1]))
            ph.append(np.array(final.get_trace()[2]))
            pp = r.get_path_properties_birefringence(iS, bire_model=bire_model)
            k = len(recs)
            recs.append((iX, iS, r.get_results()[iS]['C0'], r.get_path_length(iS), len(pp['path'])))
            out['%s_spec_%d' % (tag, k)] = np.array(final.get_frequency_spectrum()[1:])
            for key, short in () if k >= 3 else (('path', 'path'), ('refractive_index_x', 'nx'), ('refractive_index_y', 'ny'),
                               ('refractive_index_z', 'nz'), ('nominal_refractive_index', 'n'),
                               ('first_refractive_index', 'N1'), ('second_refractive_index', 'N2'),
                               ('first_polarization_vector', 'P1'), ('second_polarization_vector', 'P2'),
                               ('first_time_delay', 'T1'), ('second_time_delay', 'T2')):
                out['%s_%s_%d' % (tag, short, k)] = np.asarray(pp[key], float)
            print(tag, iX, iS, recs[-1][3], np.abs(th[-1]).max(), np.abs(ph[-1]).max(), flush=True)
    out[tag + '_rays'] = np.array(recs, float)
    out[tag + '_traces'] = np.vstack((np.array(th), np.array(ph)))
    out[tag + '_ice'] = np.array([ice.n_ice, ice.delta_n, ice.z_0])
    out[tag + '_model'] = bire_model
    out[tag + '_angle'] = np.nan if angle is None else angle


run(medium.get_ice_model('southpole_2015'), 'southpole_2015', 'southpole_A', points, x_receiver, None, 'sp')
dev = np.abs(out['sp_traces'] - out['ref_BF'])
print('this run vs reference_BF.npy: max abs dev', dev.max(), '(T07 tolerance 2e-4)')
rng = np.random.default_rng(8)
pts2 = np.stack([rng.uniform(-1500, 1500, 4), rng.uniform(-1500, 1500, 4), rng.uniform(-2400, -300, 4)], axis=1)
run(medium.get_ice_model('greenland_simple'), 'greenland_simple', 'greenland_A', pts2, np.array([10., -20., -90.]), 35., 'gl')
np.savez_compressed(os.path.join(OUT, 'ref_birefringence.npz'), **out)
