"""radiotools.helper subset: spherical<->cartesian, angles (see package docstring)."""
import numpy as np


def spherical_to_cartesian(zenith, azimuth):
    sinZenith = np.sin(zenith)
    x = sinZenith * np.cos(azimuth)
    y = sinZenith * np.sin(azimuth)
    z = np.cos(zenith)
    if hasattr(zenith, '__len__') and hasattr(azimuth, '__len__'):
        return np.array(list(zip(x, y, z)))
    return np.array([x, y, z])


def get_normalized_angle(angle, degree=False, interval=np.deg2rad([0, 360])):
    import collections.abc
    if degree:
        interval = np.rad2deg(interval)
    delta = interval[1] - interval[0]
    if isinstance(angle, (collections.abc.Sequence, np.ndarray)):
        angle = np.array(angle, dtype=float)
        while np.sum(angle >= interval[1]):
            angle[angle >= interval[1]] -= delta
        while np.sum(angle < interval[0]):
            angle[angle < interval[0]] += delta
    else:
        while angle >= interval[1]:
            angle -= delta
        while angle < interval[0]:
            angle += delta
    return angle


def cartesian_to_spherical(x, y, z):
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    if hasattr(x, '__len__') and hasattr(y, '__len__') and hasattr(z, '__len__'):
        theta = np.zeros_like(x)
        theta[z / r < 1] = np.arccos(z[z / r < 1] / r[z / r < 1])
        theta[z / r >= 1] = 0
        phi = np.arctan2(y, x)
        phi = get_normalized_angle(phi)
        return theta, phi
    if z / r < 1:
        theta = np.arccos(z / r)
    else:
        theta = 0
    phi = np.arctan2(y, x)
    phi = get_normalized_angle(phi)
    return theta, phi


def get_angle(v1, v2):
    arccos = np.dot(v1, v2) / (np.linalg.norm(v1.T, axis=0) * np.linalg.norm(v2.T, axis=0))
    mask1 = arccos > 1
    mask2 = arccos < -1
    mask = np.logical_or(mask1, mask2)
    if hasattr(arccos, '__len__'):
        arccos[mask1] = 1
        arccos[mask2] = -1
    else:
        if mask1:
            arccos = 1
        elif mask2:
            arccos = -1
    return np.arccos(arccos)


def is_equal(a, b, rel_precision=1e-5):
    if (a + b) != 0:
        return (0.5 * abs(a - b) / (abs(a + b))) < rel_precision
    return a == 0


def get_rotation(v1, v2):
    raise NotImplementedError("not needed on the hot path")
