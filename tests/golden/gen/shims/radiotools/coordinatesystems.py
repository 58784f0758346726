"""radiotools.coordinatesystems subset: the on-sky (eR, eTheta, ePhi) basis.

The matrix is the one the reference itself spells out in
NuRadioMC/SignalProp/analyticraytracing.py:2363-2365.
"""
import numpy as np


class cstrafo:
    def __init__(self, zenith, azimuth, magnetic_field_vector=None, site=None):
        ct, st = np.cos(zenith), np.sin(zenith)
        cp, sp = np.cos(azimuth), np.sin(azimuth)
        e1 = np.array([st * cp, st * sp, ct])
        e2 = np.array([ct * cp, ct * sp, -st])
        e3 = np.array([-sp, cp, 0])
        self.__transformation_matrix_onsky = np.array([e1, e2, e3])
        self.__inverse_transformation_matrix_onsky = np.linalg.inv(self.__transformation_matrix_onsky)

    def transform_from_ground_to_onsky(self, positions):
        return np.dot(self.__transformation_matrix_onsky, positions)

    def transform_from_onsky_to_ground(self, positions):
        return np.dot(self.__inverse_transformation_matrix_onsky, positions)
