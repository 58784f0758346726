"""Golden vectors for ray tracing with reflections off the bottom of the ice shelf (Moore's Bay).

1. `ref_C0`: the reference's own golden table NuRadioMC/test/SignalProp/reference_C0_MooresBay.pkl (data of
   T06unit_test_C0_mooresbay.py: 1000 vertices, seed 10, n_reflections = 2, up to 10 solutions per vertex), together
   with the inputs T06 regenerates from the seed.
2. What the reference's Python path (use_cpp=False) returns for the same vertices (`py_*`).  NB: the Python
   find_solutions misses nearly all reflection_case = 2 solutions (rays that start downwards): get_delta_y shifts x1[0] IN
   PLACE (analyticraytracing.py:226-229) and scipy hands the same array to every objective evaluation, so the shift
   accumulates.  The golden table (written by the C++ twin, whose objective works on a fresh copy, cpp:420-424) holds them.
3. The complete solution list per vertex: the C0 of the golden table, labelled (reflection, reflection_case) with the
   reference's own module-level get_delta_y on a fresh copy of x1 (|delta y| < 1 mm), handed to the Python tracer through
   set_solution; for the first `n_full` vertices path length / travel time (analytic), launch / receive vectors,
   reflection angles per path segment and the attenuation on a coarse grid; for the first `n_prop` vertices what
   apply_propagation_effects makes of a flat unit spectrum.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/golden/gen/shims:/tmp/refcopy python tests/golden/gen/gen_mooresbay.py
"""
import os
import sys
import logging
import pickle
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refharness as rh  # noqa: E402,F401
from NuRadioMC.SignalProp import analyticraytracing as ray  # noqa: E402
from NuRadioMC.utilities import medium  # noqa: E402
from NuRadioReco.utilities import units  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
REF = os.path.join(os.path.dirname(ray.__file__), '..', 'test', 'SignalProp', 'reference_C0_MooresBay.pkl')
with open(REF, 'rb') as fin:
    ref_C0 = np.array(pickle.load(fin, encoding='latin1'))

ice = medium.mooresbay_simple()
np.random.seed(10)  # T06unit_test_C0_mooresbay.py:14-27
n_events = int(1e3)
rr = np.random.triangular(50. * units.m, 3. * units.km, 3. * units.km, n_events)
phiphi = np.random.uniform(0, 2 * np.pi, n_events)
zz = np.random.uniform(0., -0.5 * units.km, n_events)
points = np.array([rr * np.cos(phiphi), rr * np.sin(phiphi), zz]).T
x_receiver = np.array([0., 0., -5.])

MAXS = 10
n_full = int(os.environ.get('N_FULL', 150))
fcoarse = np.linspace(0.05, 0.5, 4)   # = the grid __get_frequencies_for_attenuation builds from it (no re-interpolation)
r = ray.ray_tracing(ice, attenuation_model='MB1', n_reflections=2, log_level=logging.CRITICAL, use_cpp=False,
                    compile_numba=False)
# ---- 2. the Python path as it is
py_n_sol = np.zeros(n_events, np.int32)
py_C0 = np.full((n_events, MAXS), np.nan)
py_refl = np.zeros((n_events, MAXS), np.int32)
py_case = np.zeros((n_events, MAXS), np.int32)
for i, x in enumerate(points):
    r.set_start_and_end_point(x, x_receiver)
    r.find_solutions()
    py_n_sol[i] = r.get_number_of_solutions()
    for iS, res in enumerate(r.get_results()):
        py_C0[i, iS], py_refl[i, iS], py_case[i, iS] = res['C0'], res['reflection'], res['reflection_case']
print('python path: solutions', int(py_n_sol.sum()), 'with reflection_case 2:', int(np.sum(py_case == 2)))

# ---- 3. the complete list
n_sol = np.zeros(n_events, np.int32)
C0 = np.full((n_events, MAXS), np.nan)
C1 = np.full((n_events, MAXS), np.nan)
typ = np.zeros((n_events, MAXS), np.int32)
refl = np.zeros((n_events, MAXS), np.int32)
case = np.zeros((n_events, MAXS), np.int32)
resid = np.full((n_events, MAXS), np.nan)
D = np.full((n_full, MAXS), np.nan)
T = np.full((n_full, MAXS), np.nan)
launch = np.full((n_full, MAXS, 3), np.nan)
receive = np.full((n_full, MAXS, 3), np.nan)
refl_angle = np.full((n_full, MAXS, 3), np.nan)   # per path segment; NaN = no surface reflection in that segment
att = np.full((n_full, MAXS, len(fcoarse)), np.nan)
LABELS = [(0, 1), (1, 1), (1, 2), (2, 1), (2, 2)]
b = 2 * ice.n_ice
for i, x in enumerate(points):
    r.set_start_and_end_point(x, x_receiver)
    x1, x2 = np.array(r._x1, float), np.array(r._x2, float)
    row = ref_C0[i][ref_C0[i] != 0]
    n_sol[i] = len(row)
    k_label = 0
    for iS, c0 in enumerate(row):
        # labels appear in the order of the five find_solutions calls (analyticraytracing.py:2121-2125)
        best = None
        for k in range(k_label, len(LABELS)):
            rf, cs = LABELS[k]
            dy = float(np.squeeze(ray.get_delta_y(c0, x1.copy(), x2.copy(), ice.n_ice, b, ice.delta_n, ice.z_0, ice.reflection,
                                                  (-1.0, -1.0), rf, cs)))
            if abs(dy) < 1e-3:
                best = (k, dy)
                break
        assert best is not None, (i, iS, c0)
        k_label = best[0]
        refl[i, iS], case[i, iS] = LABELS[k_label]
        resid[i, iS] = best[1]
        C0[i, iS] = c0
        C1[i, iS] = r._r2d.get_C_1(x1, c0)
        typ[i, iS] = r._r2d.determine_solution_type(x1, x2, c0)
    if i < n_full and n_sol[i]:
        m = n_sol[i]
        r.set_solution({'ray_tracing_C0': C0[i, :m], 'ray_tracing_C1': C1[i, :m], 'ray_tracing_solution_type': typ[i, :m],
                        'ray_tracing_reflection': refl[i, :m], 'ray_tracing_reflection_case': case[i, :m]})
        for iS in range(m):
            D[i, iS] = r.get_path_length(iS, analytic=True)
            T[i, iS] = r.get_travel_time(iS, analytic=True)
            launch[i, iS] = r.get_launch_vector(iS)
            receive[i, iS] = r.get_receive_vector(iS)
            ra = np.atleast_1d(r.get_reflection_angle(iS))
            for k, a in enumerate(ra):
                refl_angle[i, iS, k] = np.nan if a is None else float(a)
            att[i, iS] = r.get_attenuation(iS, fcoarse, fcoarse[-1])
    if i % 100 == 0:
        print(i, n_sol[i], flush=True)

# ---- 4. apply_propagation_effects (attenuation per segment, Fresnel factors per surface reflection, coefficient and
#         phase shift per bottom reflection) on a flat unit spectrum, N = 256 at 2 GHz, n_freq = 25
import NuRadioReco.framework.electric_field  # noqa: E402
n_prop = int(os.environ.get('N_PROP', 40))
prop_spec = np.full((n_prop, MAXS, 2, 129), np.nan, complex)
r25 = ray.ray_tracing(ice, attenuation_model='MB1', n_reflections=2, n_frequencies_integration=25, log_level=logging.CRITICAL,
                      use_cpp=False, compile_numba=False)
for i in range(n_prop):
    m = n_sol[i]
    if not m:
        continue
    r25.set_start_and_end_point(points[i], x_receiver)
    r25.set_solution({'ray_tracing_C0': C0[i, :m], 'ray_tracing_C1': C1[i, :m], 'ray_tracing_solution_type': typ[i, :m],
                      'ray_tracing_reflection': refl[i, :m], 'ray_tracing_reflection_case': case[i, :m]})
    for iS in range(m):
        ef = NuRadioReco.framework.electric_field.ElectricField([0])
        ef.set_frequency_spectrum(np.ones((3, 129), complex), 2.0)
        out = r25.apply_propagation_effects(ef, iS).get_frequency_spectrum()
        prop_spec[i, iS, 0], prop_spec[i, iS, 1] = out[1], out[2]
print('apply_propagation_effects on', int(n_sol[:n_prop].sum()), 'solutions')

# the Python path's solutions are a subset of the table, with the same labels
for i in range(n_events):
    for k in range(py_n_sol[i]):
        hit = [j for j in range(n_sol[i]) if abs(C0[i, j] - py_C0[i, k]) <= 1e-6 * py_C0[i, k]
               and refl[i, j] == py_refl[i, k] and case[i, j] == py_case[i, k]]
        assert len(hit) == 1, (i, k)
print('table solutions per label', {lab: int(np.sum((refl == lab[0]) & (case == lab[1]) & ~np.isnan(C0))) for lab in LABELS})
np.savez_compressed(os.path.join(OUT, 'ref_mooresbay.npz'), ref_C0=ref_C0, points=points, x_receiver=x_receiver,
                    ice=np.array([ice.n_ice, ice.delta_n, ice.z_0]), z_reflection=ice.reflection,
                    reflection_coefficient=ice.reflection_coefficient, reflection_phase_shift=ice.reflection_phase_shift,
                    py_n_sol=py_n_sol, py_C0=py_C0, py_reflection=py_refl, py_reflection_case=py_case,
                    n_sol=n_sol, C0=C0, C1=C1, type=typ, reflection=refl, reflection_case=case, delta_y=resid, n_full=n_full,
                    D=D, T=T, launch=launch, receive=receive, refl_angle=refl_angle, fcoarse=fcoarse, att=att, att_model='MB1',
                    n_prop=n_prop, prop_spec=prop_spec)
print('solutions', int(n_sol.sum()), 'max per vertex', n_sol.max(), 'max |delta y| of the golden C0', np.nanmax(np.abs(resid)))
