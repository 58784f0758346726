"""Which side holds the mathematically right solution set where oracle and reference disagree?  ("next" item 6.)

The analytic ray tracer's solutions are the roots of delta_y(log C0) (analyticraytracing.py:204-272, :1357).  The reference finds
the first one by scipy.optimize.root on (delta_y)^2 -- a double root, stopped ~1e-7 away and kept only if (delta_y)^2 < 1e-7 --
so for a few pairs in a thousand it keeps or loses a root with the last bits of exp / log.  This script
settles those pairs with arithmetic instead of argument: for every pair on which the oracle (oracle/nrmc_oracle.c) and the
reference (the committed fixtures tests/golden/raytrace_[ABC].npz and chain_bench_N4096.npz, written by the reference itself)
report different solution counts, and for every ray whose path length / travel time differ by more than 1e-6, it restates
delta_y, the solution type and the analytic path length / travel time in 60-digit arithmetic (mpmath), finds ALL roots of delta_y
on the reference's search interval and prints which list is the true set and the true D / T.

    python tools/true_roots.py                 # the table of DESIGN.md section 2 (a minute on one core)
    python tools/true_roots.py --json out.json

CPU only; reads nothing but the committed fixtures; test infrastructure (imports oracle/)."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

import mpmath as mp   # noqa: E402

mp.mp.dps = 60
SPEED_OF_LIGHT = mp.mpf('0.299792458')   # m / ns (NuRadioReco.utilities.units)


class Ice:
    """n(z) = n_ice - delta_n exp(z / z_0) in mpmath numbers; the functions are the reference's (file:line in oracle/nrmc_oracle.c)"""

    def __init__(self, ice):
        self.n_ice, self.delta_n, self.z_0 = (mp.mpf(float(v)) for v in ice)

    def n(self, z):
        return self.n_ice - self.delta_n * mp.exp(z / self.z_0)

    def gamma(self, z):
        return self.delta_n * mp.exp(z / self.z_0)

    def get_y(self, gamma, C0, C1):                      # :105-125
        b = 2 * self.n_ice
        c = self.n_ice ** 2 - 1 / C0 ** 2
        root = abs(gamma ** 2 - gamma * b + c)
        logarg = gamma / (2 * mp.sqrt(c) * mp.sqrt(root) - b * gamma + 2 * c)
        return self.z_0 / mp.sqrt(self.n_ice ** 2 * C0 ** 2 - 1) * mp.log(logarg) + C1

    def turning(self, C0):                               # :133-158
        b = 2 * self.n_ice
        c = self.n_ice ** 2 - 1 / C0 ** 2
        g2 = b / 2 - mp.sqrt(b * b / 4 - c)
        z2 = mp.log(g2 / self.delta_n) * self.z_0
        if z2 > 0:
            return self.delta_n, mp.mpf(0)
        return g2, z2

    def y_mirror(self, z, C0, C1):                       # :160-184
        g_t, z_t = self.turning(C0)
        y_t = self.get_y(g_t, C0, C1)
        if z < z_t:
            return self.get_y(self.gamma(z), C0, C1)
        return 2 * y_t - self.get_y(self.gamma(2 * z_t - z), C0, C1)

    def C0(self, logC0):
        return mp.exp(logC0) + 1 / self.n_ice            # :99

    def delta_y(self, logC0, x1, x2):                    # :204-272 (no bottom reflection)
        """returns (value, continuous): the penalty branch (turning point below the receiver) is not a crossing of zero"""
        C0 = self.C0(logC0)
        C1 = x1[0] - self.y_mirror(x1[1], C0, 0)
        g_t, z_t = self.turning(C0)
        y_t = self.get_y(g_t, C0, C1)
        if z_t < x2[1]:
            dz, dy = z_t - x2[1], y_t - x2[0]
            return -(mp.sqrt(dz * dz + dy * dy) + 10 * abs(dz)), False
        y2 = self.get_y(self.gamma(x2[1]), C0, C1)
        if y_t > x2[0]:
            return x2[0] - y2, True
        return -(x2[0] - (2 * y_t - y2)), True

    def solution_type(self, C0, x1, x2):                 # :1365-1398
        C1 = x1[0] - self.y_mirror(x1[1], C0, 0)
        g_t, z_t = self.turning(C0)
        y_t = self.get_y(g_t, C0, C1)
        if x2[0] < y_t:
            return 1
        return 3 if z_t == 0 else 2

    def path_length_time(self, C0, x1, x2):              # :602-690, :692-783 (analytic, receiver in ice)
        typ = self.solution_type(C0, x1, x2)
        # launch angle: dy/dz at the start point (get_y_diff :306-355), sin / cos of arctan
        n1 = self.n(x1[1])
        dy = 1 / mp.sqrt(C0 ** 2 * n1 ** 2 - 1)
        sin_launch = dy / mp.sqrt(1 + dy * dy)
        beta = n1 * sin_launch
        alpha = self.n_ice ** 2 - beta ** 2
        zz = [x1[1], x2[1], mp.mpf(0)]
        if typ == 2:
            zz[2] = self.turning(C0)[1]
        s, ct = [], []
        for z in zz:
            nz = self.n(z)
            gam = max(mp.mpf(0), nz * nz - beta * beta)
            l1 = mp.sqrt(alpha * gam) + self.n_ice * nz - beta * beta
            l2 = mp.sqrt(gam) + nz
            sa = mp.sqrt(alpha)
            s.append(self.n_ice / sa * (z - self.z_0 * mp.log(l1)) + self.z_0 * mp.log(l2))
            ct.append(self.z_0 * (mp.sqrt(gam) - self.n_ice ** 2 / sa * mp.log(l1) + self.n_ice * mp.log(l2)) + self.n_ice ** 2 * z / sa)
        if typ == 1:
            return typ, s[1] - s[0], (ct[1] - ct[0]) / SPEED_OF_LIGHT
        return typ, 2 * s[2] - s[0] - s[1], (2 * ct[2] - ct[0] - ct[1]) / SPEED_OF_LIGHT


def geometry_2d(X1, X2):
    """set_start_and_end_point (:2057-2090): the deeper point first, rotated into the y-z plane"""
    X1, X2 = np.asarray(X1, float), np.asarray(X2, float)
    if X2[2] < X1[2]:
        X1, X2 = X2, X1
    d = X2 - X1
    rho = np.hypot(d[0], d[1])
    return (X1[0], X1[2]), (X1[0] + rho, X2[2])


_orc = None


def _delta_y_double(logC0, x1, x2, ice):
    """the oracle's double-precision objective: only used to bracket; every bracket is refined in mpmath"""
    global _orc
    if _orc is None:
        from oracle import raytrace_oracle as orc
        _orc = orc.lib()
        _orc.orc_delta_y.restype = ctypes.c_double
        _orc.orc_delta_y.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double)]
    a1, a2, ai = (np.ascontiguousarray(v, float) for v in (x1, x2, ice))
    dp = ctypes.POINTER(ctypes.c_double)
    return _orc.orc_delta_y(float(logC0), a1.ctypes.data_as(dp), a2.ctypes.data_as(dp), ai.ctypes.data_as(dp))


def true_roots(x1, x2, ice, lo=-40., hi=20., n_grid=24001):
    """All zeros of delta_y(log C0) on [lo, hi] (outside it C0 is 1 / n_ice + 4e-18 or above 5e8: no ray of a survey geometry),
    as mpmath numbers: (roots, notes).  A grid in double precision brackets sign changes and near-tangent minima; both are then
    refined in 60-digit arithmetic, so neither a pair of close roots nor a double root is taken for the other."""
    m = Ice(ice)
    X1 = (mp.mpf(float(x1[0])), mp.mpf(float(x1[1])))
    X2 = (mp.mpf(float(x2[0])), mp.mpf(float(x2[1])))
    f = lambda t: m.delta_y(mp.mpf(t), X1, X2)
    grid = np.linspace(lo, hi, n_grid)
    val = np.array([_delta_y_double(t, x1, x2, ice) for t in grid])
    roots, notes = [], []

    def bisect(a, b):
        fa, fb = f(a)[0], f(b)[0]
        a, b = mp.mpf(a), mp.mpf(b)
        for _ in range(220):
            c = (a + b) / 2
            fc, cont = f(c)
            if (fc > 0) == (fa > 0):
                a, fa = c, fc
            else:
                b, fb = c, fc
        c = (a + b) / 2
        fc, cont = f(c)
        return c, fc, cont

    ok = np.isfinite(val)
    for i in range(n_grid - 1):
        if not (ok[i] and ok[i + 1]):
            continue
        if (val[i] > 0) != (val[i + 1] > 0) or val[i] == 0:
            c, fc, cont = bisect(grid[i], grid[i + 1])
            # (delta_y is continuous where the penalty branch begins -- the turning point reaches the receiver, both branches
            # go to zero there --, so what decides is whether |delta_y| vanishes at the limit of the bisection, not the branch)
            if abs(fc) < mp.mpf(10) ** -25:
                roots.append(c)
                if not cont:
                    notes.append('root at log C0 = %.6f: the ray turns AT the receiver (delta_y -> 0 from the penalty branch)' % float(c))
            else:
                notes.append('sign change at log C0 = %.6f is a jump (|delta_y| -> %.3g m): no root' % (float(c), float(abs(fc))))
    # near-tangent minima of |delta_y| without a sign change on the grid: look between the neighbours in mpmath
    a = np.abs(val)
    for i in range(1, n_grid - 1):
        if ok[i - 1] and ok[i] and ok[i + 1] and a[i] <= a[i - 1] and a[i] <= a[i + 1] and a[i] < 0.5 and \
                (val[i - 1] > 0) == (val[i] > 0) == (val[i + 1] > 0):
            lo_, hi_ = mp.mpf(grid[i - 1]), mp.mpf(grid[i + 1])
            g = lambda t: abs(f(t)[0])
            for _ in range(200):   # golden-section search for the minimum of |delta_y|
                p, q = lo_ + (hi_ - lo_) * mp.mpf('0.381966'), lo_ + (hi_ - lo_) * mp.mpf('0.618034')
                if g(p) < g(q):
                    hi_ = q
                else:
                    lo_ = p
            t0 = (lo_ + hi_) / 2
            v0, cont = f(t0)
            if cont and (v0 > 0) != (val[i] > 0):   # it does cross: two close roots
                for (p, q) in ((mp.mpf(grid[i - 1]), t0), (t0, mp.mpf(grid[i + 1]))):
                    c, fc, cont2 = bisect(p, q)
                    if abs(fc) < mp.mpf(10) ** -25:
                        roots.append(c)
            elif abs(v0) < mp.mpf(10) ** -30:
                roots.append(t0)
                notes.append('double root at log C0 = %.9f' % float(t0))
            elif abs(v0) < 1e-3:
                notes.append('near miss at log C0 = %.6f: min |delta_y| = %.3g m (no root)' % (float(t0), float(abs(v0))))
    roots = sorted(set(roots))
    return m, X1, X2, roots, notes


def judge(x1_3d, x2_3d, ice, c0_oracle, c0_ref):
    """one pair: the true roots (as C0), and for each list whether it is the true set (1e-6 relative, the north_star tolerance)"""
    x1, x2 = geometry_2d(x1_3d, x2_3d)
    m, X1, X2, roots, notes = true_roots(x1, x2, ice)
    true_c0 = [m.C0(t) for t in roots]

    def matches(lst):
        lst = [float(v) for v in lst if np.isfinite(v)]
        if len(lst) != len(true_c0):
            return False
        return all(abs(mp.mpf(a) - b) <= mp.mpf('1e-6') * b for a, b in zip(sorted(lst), sorted(true_c0)))

    def subset(lst):
        lst = [float(v) for v in lst if np.isfinite(v)]
        return all(any(abs(mp.mpf(a) - b) <= mp.mpf('1e-6') * b for b in true_c0) for a in lst)
    out = dict(x1=[float(v) for v in x1], x2=[float(v) for v in x2], true_C0=[float(v) for v in true_c0],
               true_logC0=[float(t) for t in roots],
               n_true=len(true_c0), n_oracle=int(np.isfinite(c0_oracle).sum()), n_ref=int(np.isfinite(c0_ref).sum()),
               oracle_is_true_set=matches(c0_oracle), ref_is_true_set=matches(c0_ref),
               oracle_subset_of_true=subset(c0_oracle), ref_subset_of_true=subset(c0_ref), notes=notes, rays=[])

    def missing(lst):   # indices (in ascending log C0) of the true roots a list does not hold
        lst = [float(v) for v in lst if np.isfinite(v)]
        return [i for i, b in enumerate(true_c0) if not any(abs(mp.mpf(a) - b) <= mp.mpf('1e-6') * b for a in lst)]
    out['oracle_missing'], out['ref_missing'] = missing(c0_oracle), missing(c0_ref)
    for t in roots:
        C0 = m.C0(t)
        typ, D, T = m.path_length_time(C0, X1, X2)
        # conditioning of the first root as the reference finds it: slope of delta_y there, and the (delta_y)^2 the reference's
        # hybr iterate has 1e-7 (relative, in log C0) from the root -- kept only below 1e-7
        h = mp.mpf('1e-20')
        slope = (m.delta_y(t + h, X1, X2)[0] - m.delta_y(t - h, X1, X2)[0]) / (2 * h)
        out['rays'].append(dict(C0=float(C0), logC0=float(t), type=typ, D=float(D), T=float(T), slope_m_per_unit_logC0=float(slope),
                                dy2_at_1e7=float((slope * mp.mpf('1e-7') * max(abs(t), 1)) ** 2)))
    return out, (m, X1, X2, roots)


def collect():
    """the pairs to settle: [(label, X1, X2, ice, C0 oracle [2], C0 reference [2], D/T oracle, D/T reference)]"""
    from oracle import raytrace_oracle as orc
    import bench
    G = os.path.join(ROOT, 'tests', 'golden')
    cases = []
    for name in 'ABC':
        g = np.load(os.path.join(G, 'raytrace_%s.npz' % name))
        o = orc.raytrace_batch(g['x1'], g['x2'], g['ice'])
        for i in np.flatnonzero(o['n_sol'] != g['n_sol']):
            cases.append(('raytrace_%s pair %d: solution count' % (name, i), g['x1'][i], g['x2'][i], g['ice'], o, g, i))
        ok = o['n_sol'] == g['n_sol']
        relD = np.abs(o['D'] - g['D']) / np.abs(g['D'])
        relT = np.abs(o['T'] - g['T']) / np.abs(g['T'])
        far = ok & (np.nan_to_num(np.maximum(relD, relT), nan=0.).max(axis=1) > 1e-6)
        for i in np.flatnonzero(far):
            cases.append(('raytrace_%s pair %d: D / T differ by > 1e-6' % (name, i), g['x1'][i], g['x2'][i], g['ice'], o, g, i))
    # the bench list: events whose kept-ray count differs between oracle chain and reference -> the (event, channel) pairs behind it
    g = np.load(os.path.join(G, 'chain_bench_N4096.npz'))
    K = len(g['zenith'])
    x1 = np.repeat(g['vertex'][:K], len(bench.CHANNELS), axis=0)
    x2 = np.tile(bench.CHANNELS, (K, 1))
    o = orc.raytrace_batch(x1, x2, g['ice'])
    ref_c0 = np.full((K * 5, 2), np.nan)
    # the fixture keeps the rays that pass the delta_C cut: a pair's reference list is known where the oracle's rays of the pair
    # pass the cut too; build it from (event, channel, iS)
    ref_c0[g['ray_event'] * 5 + g['ray_channel'], g['ray_iS']] = g['ray_C0']
    ref_has = np.zeros(K * 5, int)
    np.add.at(ref_has, g['ray_event'] * 5 + g['ray_channel'], 1)
    return cases, (g, o, x1, x2, ref_c0, ref_has)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    ap.add_argument('--bench-events', type=int, default=24000)
    return run(ap.parse_args())


def run(args):
    cases, (g, o, x1, x2, ref_c0, ref_has) = collect()
    results = []
    print('# pairs on which oracle and reference disagree, settled in %d-digit arithmetic' % mp.mp.dps)
    print('# case | true roots | oracle | reference | who holds the true set')
    for label, X1, X2, ice, oo, gg, i in cases:
        r, ctx = judge(X1, X2, ice, oo['C0'][i], gg['C0'][i])
        r['case'] = label
        m, mx1, mx2, roots = ctx
        line = '%s | n_true = %d | oracle %d (%s) | reference %d (%s)' % (
            label, r['n_true'], r['n_oracle'], 'true set' if r['oracle_is_true_set'] else ('subset' if r['oracle_subset_of_true'] else 'NOT a subset'),
            r['n_ref'], 'true set' if r['ref_is_true_set'] else ('subset' if r['ref_subset_of_true'] else 'NOT a subset'))
        if 'D / T' in label:
            # relative errors of both sides against the true D / T of the matching root
            for s in range(int(oo['n_sol'][i])):
                tr = min(r['rays'], key=lambda q: abs(q['C0'] - oo['C0'][i, s]))
                line += ' | ray %d (type %d): D true %.9f m, oracle %+.2e, reference %+.2e; T: oracle %+.2e, reference %+.2e' % (
                    s, tr['type'], tr['D'], oo['D'][i, s] / tr['D'] - 1, gg['D'][i, s] / tr['D'] - 1, oo['T'][i, s] / tr['T'] - 1,
                    gg['T'][i, s] / tr['T'] - 1)
                r.setdefault('dt', []).append(dict(ray=s, D_true=tr['D'], T_true=tr['T'], rel_D_oracle=oo['D'][i, s] / tr['D'] - 1,
                                                   rel_D_ref=gg['D'][i, s] / tr['D'] - 1, rel_T_oracle=oo['T'][i, s] / tr['T'] - 1,
                                                   rel_T_ref=gg['T'][i, s] / tr['T'] - 1))
        for nt in r['notes']:
            line += ' | ' + nt
        print(line)
        results.append(r)
    # the bench list (first 24 000 events of bench.py's 1e6-event list through the reference): the fixture keeps the rays that pass
    # the delta_C cut, so a pair is marked where the oracle's kept rays of the pair and the reference's differ in number
    from oracle import spectral_oracle as so
    K = min(args.bench_events, len(g['zenith']))
    n_ch = len(x2) // len(g['zenith'])
    shower_dir = -np.array([so.spherical_to_cartesian(z, a) for z, a in zip(g['zenith'][:K], g['azimuth'][:K])])
    vz = g['vertex'][:K, 2]
    cher = np.arccos(1. / (g['ice'][0] - g['ice'][1] * np.exp(vz / g['ice'][2])))
    kept = np.zeros(K * n_ch, int)
    cut = float(g['delta_C_cut'])
    for k in range(K * n_ch):
        ev = k // n_ch
        ns = int(o['n_sol'][k])
        if ns == 0:
            continue
        dC = np.array([so.get_angle(shower_dir[ev], o['launch'][k, s]) for s in range(ns)]) - cher[ev]
        kept[k] = int((np.abs(dC) <= cut).sum())
    marked = np.flatnonzero(kept != ref_has[:K * n_ch])
    print('# bench list: %d events, %d (event, channel) pairs; kept-ray counts differ on %d pairs of %d events' % (
        K, K * n_ch, len(marked), len(set(marked // n_ch))))
    for k in marked:
        r, ctx = judge(x1[k], x2[k], g['ice'], o['C0'][k], ref_c0[k])
        r['case'] = 'bench event %d channel %d' % (k // n_ch, k % n_ch)
        r['ref_kept_rays'] = int(ref_has[k])
        r['oracle_kept_rays'] = int(kept[k])
        line = '%s | n_true = %d | oracle %d (%s), %d pass the delta_C cut | reference kept %d (%s)' % (
            r['case'], r['n_true'], r['n_oracle'], 'true set' if r['oracle_is_true_set'] else ('subset' if r['oracle_subset_of_true'] else 'NOT a subset'),
            kept[k], ref_has[k], 'subset of the true set' if r['ref_subset_of_true'] else 'NOT a subset')
        for q in r['rays']:
            line += ' | root log C0 = %.6f type %d: (delta_y)^2 1e-7 off the root = %.2e' % (q['logC0'], q['type'], q['dy2_at_1e7'])
        for nt in r['notes']:
            line += ' | ' + nt
        print(line)
        results.append(r)
    n_cases = len(results)
    print('# summary: %d cases; oracle holds the true set in %d, the reference in %d; oracle list is a subset of the true set in %d' % (
        n_cases, sum(r['oracle_is_true_set'] for r in results), sum(r['ref_is_true_set'] for r in results),
        sum(r['oracle_subset_of_true'] for r in results)))
    if args.json:
        json.dump(results, open(args.json, 'w'), indent=1)
    return results


if __name__ == '__main__':
    main()
