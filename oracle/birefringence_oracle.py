"""CPU restatement of the birefringent pulse propagation (TEST INFRASTRUCTURE ONLY, never imported by nuradiomc_amd).

Follows NuRadioMC/SignalProp/analyticraytracing.py: get_pulse_propagation_birefringence (:2369-2445),
get_effective_index_birefringence (:2165-2207), get_polarization_birefringence (:2243-2337), on_sky_birefringence
(:2339-2367), the ray path of get_path (:1239-1291, :2148-2163), NuRadioMC/utilities/medium_base.py:378-420 (the three
depth splines, evaluated here with an own de Boor recursion as FITPACK's splev does, extrapolating with the end
pieces) and NuRadioReco/framework/base_trace.py:246-276 (apply_time_shift).

Parity status: PINNED against the reference's golden file reference_BF.npy (T07test_birefringence.py) and against
per-step path properties / final spectra of the reference run in the build container (tests/golden/ref_birefringence.npz,
generator tests/golden/gen/gen_birefringence.py; tests/test_oracle_golden.py).
"""
import numpy as np

SPEED_OF_LIGHT = 0.299792458   # m / ns (scipy.constants.c * units.m / units.s)


def spline_eval(t, c, x, k=3):
    """B-spline sum_i c_i B_{i,k}(x) (scipy UnivariateSpline._from_tck(tck)(x), ext = 0: outside the knot range the end
    polynomial piece continues)"""
    t = np.asarray(t, float)
    n = len(t)
    l = int(np.searchsorted(t, x, side='right')) - 1
    l = min(max(l, k), n - k - 2)
    h = np.zeros(k + 1)
    h[0] = 1.
    for j in range(1, k + 1):
        hh = h[:j].copy()
        h[0] = 0.
        for i in range(j):
            li = l + 1 + i
            lj = li - j
            if t[li] == t[lj]:
                h[i + 1] = 0.
                continue
            f = hh[i] / (t[li] - t[lj])
            h[i] += f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return float(np.dot(c[l - k:l + 1], h))


def _get_y(gamma, C0, C1, ice):
    n_ice, delta_n, z_0 = ice
    b = 2 * n_ice
    c = n_ice ** 2 - C0 ** -2
    root = np.abs(gamma ** 2 - gamma * b + c)
    logargument = gamma / (2 * c ** 0.5 * root ** 0.5 - b * gamma + 2 * c)
    return z_0 * (n_ice ** 2 * C0 ** 2 - 1) ** -0.5 * np.log(logargument) + C1


def _turning_point(C0, ice):
    n_ice, delta_n, z_0 = ice
    b = 2 * n_ice
    c = n_ice ** 2 - C0 ** -2
    gamma2 = b * 0.5 - (0.25 * b ** 2 - c) ** 0.5
    z2 = np.log(gamma2 / delta_n) * z_0
    if z2 > 0:
        z2 = 0
        gamma2 = delta_n
    return gamma2, z2


def ray_path(X1, X2, C0, ice, n_points):
    """ray_tracing.get_path (:2148-2163) on top of ray_tracing_2D.get_path (:1239-1291), receiver in ice, no bottom
    reflections: n_points positions, uniform in the mirrored depth coordinate, from the lower to the higher end point"""
    n_ice, delta_n, z_0 = ice
    X1, X2 = np.array(X1, float), np.array(X2, float)
    if X2[2] < X1[2]:
        X1, X2 = X2, X1
    dX = X2 - X1
    dPhi = -np.arctan2(dX[1], dX[0])
    cph, sph = np.cos(dPhi), np.sin(dPhi)
    R = np.array([[cph, -sph, 0], [sph, cph, 0], [0, 0, 1]])
    X2r = R.dot(dX) + X1
    x1, x2 = np.array([X1[0], X1[2]]), np.array([X2r[0], X2r[2]])
    gamma = lambda z: delta_n * np.exp(z / z_0)
    g_turn, z_turn = _turning_point(C0, ice)

    def y_mirror(z, C1=0.):
        y_turn = _get_y(g_turn, C0, C1, ice)
        return _get_y(gamma(z), C0, C1, ice) if z < z_turn else 2 * y_turn - _get_y(gamma(2 * z_turn - z), C0, C1, ice)
    C1 = x1[0] - y_mirror(x1[1])
    y_turn = _get_y(g_turn, C0, C1, ice)
    zstop = x2[1]
    if y_turn < x2[0]:
        zstop = x1[1] + np.abs(z_turn - x1[1]) + np.abs(z_turn - x2[1])
    z = np.linspace(x1[1], zstop, n_points)
    mask = z < z_turn
    res, zs = np.zeros_like(z), np.zeros_like(z)
    zs[mask] = z[mask]
    res[mask] = _get_y(gamma(z[mask]), C0, C1, ice)
    res[~mask] = 2 * y_turn - _get_y(gamma(2 * z_turn - z[~mask]), C0, C1, ice)
    zs[~mask] = 2 * z_turn - z[~mask]
    path_2d = np.array([res, np.zeros_like(res), zs]).T
    dP = path_2d - np.array([X1[0], 0, X1[2]])
    return np.matmul(R.T, dP.T).T + X1


def effective_indices(direction, nx, ny, nz):
    """:2188-2207"""
    sx, sy, sz = direction
    A = ny ** 2 * nz ** 2 * (-1 + sx ** 2) + nx ** 2 * (nz ** 2 * (-1 + sy ** 2) + ny ** 2 * (-1 + sz ** 2))
    B = np.sqrt(4 * nx ** 2 * ny ** 2 * nz ** 2 * (nz ** 2 * (-1 + sx ** 2 + sy ** 2) + ny ** 2 * (-1 + sx ** 2 + sz ** 2)
                                                    + nx ** 2 * (-1 + sy ** 2 + sz ** 2)) + A ** 2)
    num = -2 * nx ** 2 * ny ** 2 * nz ** 2
    return np.sqrt(num / (A - B)), np.sqrt(num / (A + B))


def _on_sky(direction, p):
    """hp.cartesian_to_spherical + on_sky_birefringence (:2339-2367)"""
    x, y, z = direction
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    theta = 0. if r == 0 else np.arccos(z / r)
    phi = np.arctan2(y, x)
    if phi < 0:
        phi += 2 * np.pi
    T = np.array([[np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)],
                  [np.cos(theta) * np.cos(phi), np.cos(theta) * np.sin(phi), -np.sin(theta)],
                  [-np.sin(phi), np.cos(phi), 0]])
    return T.dot(p)


def polarizations(N1, N2, direction, nx, ny, nz):
    """get_polarization_birefringence (:2243-2337): on-sky (r, theta, phi) vectors of the two eigen-polarisations"""
    narrow, wide = 1e-9, 1e-10
    nn = np.array([nx, ny, nz])
    c1, c2 = np.any(np.abs(N1 - nn) <= narrow), np.any(np.abs(N2 - nn) <= narrow)

    def simple(n):
        p = np.array([direction[0] / (n ** 2 - nx ** 2), direction[1] / (n ** 2 - ny ** 2), direction[2] / (n ** 2 - nz ** 2)])
        return p / np.linalg.norm(p)
    if c1 or c2:
        if c1 and c2:
            return np.zeros(3), np.zeros(3)
        if abs(N1 - nx) <= wide:
            return (np.array([0, 0, 1.]) if direction[0] < 0 else np.array([0, 0, -1.])), np.array([0, 1., 0])
        if abs(N1 - ny) <= narrow:
            return (np.array([0, 0, 1.]) if direction[1] < 0 else np.array([0, 0, -1.])), np.array([0, 1., 0])
        if abs(N2 - ny) <= narrow:
            return np.array([0, 1., 0]), (np.array([0, 0, -1.]) if direction[1] < 0 else np.array([0, 0, 1.]))
        if abs(N2 - nz) <= wide:
            return np.array([0, 0, -1.]), (np.array([0, -1., 0]) if direction[2] < 0 else np.array([0, 1., 0]))
    return _on_sky(direction, simple(N1)), _on_sky(direction, simple(N2))


def path_steps(X1, X2, C0, D, ice, tck, angle_to_iceflow=None, n_ref=1.78):
    """Per 1 m step of the path (acc = int(D / m) points): the 2 x 2 matrix R of the theta / phi components of the two
    eigen-polarisations, the delay t_1 - t_0 of the second one, and what the reference's
    get_path_properties_birefringence lists (:2447-2522)."""
    n_ice, delta_n, z_0 = ice
    acc = int(D / 1.)
    path = ray_path(X1, X2, C0, ice, acc)
    if angle_to_iceflow is not None:
        a = angle_to_iceflow * np.pi / 180
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        path[:, :2] = rot.dot(path[:, :2].T).T
    out = dict(path=path[1:], nx=[], ny=[], nz=[], n=[], N1=[], N2=[], P1=[], P2=[], T1=[], T2=[])
    for i in range(acc - 1):
        z = path[i][2]
        n_nominal = n_ice - delta_n * np.exp(z / z_0) if z <= 0 else 1.   # IceModelSimple.get_index_of_refraction
        bx, by, bz = (spline_eval(tck[j][0], tck[j][1], -z) for j in range(3))
        nx, ny, nz = n_nominal + bx - n_ref, n_nominal + by - n_ref, n_nominal + bz - n_ref
        dD = path[i + 1] - path[i]
        len_diff = np.linalg.norm(dD)
        direction = dD / len_diff
        N1, N2 = effective_indices(direction, nx, ny, nz)
        P1, P2 = polarizations(N1, N2, direction, nx, ny, nz)
        for k, v in zip(('nx', 'ny', 'nz', 'n', 'N1', 'N2', 'P1', 'P2', 'T1', 'T2'),
                        (bx, by, bz, n_nominal, N1, N2, P1, P2, len_diff * N1 / SPEED_OF_LIGHT, len_diff * N2 / SPEED_OF_LIGHT)):
            out[k].append(v)
    return {k: np.array(v) for k, v in out.items()}


def propagate(spec_theta, spec_phi, sampling_rate, steps):
    """get_pulse_propagation_birefringence (:2416-2445): per step E <- R^T diag(1, time shift by t_1 - t_0) R E, the time
    shift as in BaseTrace.apply_time_shift (a shift within 1e-5 of a whole number of samples rolls the time trace)"""
    e = np.array([spec_theta, spec_phi], complex)
    n_f = e.shape[1]
    N = 2 * (n_f - 1)
    ff = np.fft.rfftfreq(N, 1. / sampling_rate)
    for i in range(len(steps['T1'])):
        a, b = steps['P1'][i][1:]
        c, d = steps['P2'][i][1:]
        if np.isclose(a * d - b * c, 0) or np.isnan([a, b, c, d]).any():
            continue
        dt = steps['T2'][i] - steps['T1'][i]
        b0 = a * e[0] + b * e[1]
        b1 = c * e[0] + d * e[1]
        x = dt * sampling_rate
        if abs(round(x) - x) < 1e-5:
            tr = np.roll(np.fft.irfft(b1, n=N), int(round(x)))   # the (-> time -> roll -> frequency) round trip of set_trace
            b1 = np.fft.rfft(tr)
        else:
            b1 = b1 * np.exp(-2.j * np.pi * dt * ff)
        e = np.array([a * b0 + c * b1, b * b0 + d * b1])
    return e
