"""Host-side mirror of NuRadioMC/utilities/cross_sections.py for the tabulated neutrino-nucleon cross sections that the device
does not evaluate itself ('ctw' and 'ghandi' are device code, csrc/earth.hip).

'csms' (Cooper-Sarkar, Mertsch, Sarkar, JHEP 08 (2011) 042; cross_sections.py:123-229): the published table -- energy [GeV], CC and
NC cross section [pb] for neutrinos and antineutrinos -- interpolated linearly in the energy (scipy interp1d, bounds_error=True: a
ValueError outside 50 GeV ... 5e11 GeV).  As in the reference only the per-interaction values exist: inttype 'cc' / 'nc' (arrays);
for inttype='total' the reference's csms() matches no row and returns zeros (cross_sections.py:213-227), which get_weight turns into
weight 1 -- reproduced here, and said so, rather than silently summed.

'hedis_bgr18' (:283-299) integrates a differential cross section from a data file the reference downloads on first use
(BGR18_dsigma_dy_H2O.npz); it is not available offline.  A caller who has it passes the resulting values through
`earth_attenuation.get_weight(..., cross_section=sigma)` / NRHIP_XS_GIVEN.
"""
import numpy as np

GeV = 1e9            # NuRadioReco/utilities/units.py: eV = 1
picobarn = 1e-40     # 1e-12 * 1e-28 m^2

# energy [GeV], sigma_CC [pb], sigma_NC [pb]  (JHEP 08 (2011) 042, tables 1 and 2)
_E = (50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 50000, 100000, 200000, 500000, 1e6, 2e6, 5e6, 1e7, 2e7, 5e7, 1e8, 2e8,
      5e8, 1e9, 2e9, 5e9, 1e10, 2e10, 5e10, 1e11, 2e11, 5e11)
_NU_CC = (0.32, 0.65, 1.3, 3.2, 6.2, 12., 27., 47., 77., 140., 210., 310., 490., 690., 950., 1400., 1900., 2600., 3700., 4800.,
          6200., 8700., 11000., 14000., 19000., 24000., 30000., 39000., 48000., 59000., 75000.)
_NU_NC = (0.10, 0.20, 0.41, 1.0, 2.0, 3.8, 8.6, 15., 26., 49., 75., 110., 180., 260., 360., 540., 730., 980., 1400., 1900., 2400.,
          3400., 4400., 5600., 7600., 9600., 12000., 16000., 20000., 24000., 31000.)
_NUBAR_CC = (0.15, 0.33, 0.69, 1.8, 3.6, 7., 17., 31., 55., 110., 180., 270., 460., 660., 920., 1400., 1900., 2500., 3700., 4800.,
             6200., 8700., 11000., 14000., 19000., 24000., 30000., 39000., 48000., 59000., 75000.)
_NUBAR_NC = (0.05, 0.12, 0.24, 0.61, 1.20, 2.4, 5.8, 11., 19., 39., 64., 99., 170., 240., 350., 530., 730., 980., 1400., 1900.,
             2400., 3400., 4400., 5600., 7600., 9600., 12000., 16000., 20000., 24000., 31000.)


def _interp(energy, values):
    e = np.asarray(_E, float) * GeV
    energy = np.asarray(energy, float)
    if np.any(energy < e[0]) or np.any(energy > e[-1]):
        raise ValueError("A value in x_new is outside the interpolation range.")   # interp1d(bounds_error=True)
    return np.interp(energy, e, np.asarray(values, float) * picobarn)


def csms(energy, inttype, flavors):
    """cross_sections.csms (:123-229): per-event cross section [m^2] for arrays energy / inttype ('cc' | 'nc') / flavors"""
    energy = np.atleast_1d(np.asarray(energy, float))
    flavors = np.broadcast_to(np.asarray(flavors), energy.shape)
    inttype = np.broadcast_to(np.asarray(inttype), energy.shape)
    out = np.zeros_like(energy)
    for anti, cc, table in ((False, True, _NU_CC), (False, False, _NU_NC), (True, True, _NUBAR_CC), (True, False, _NUBAR_NC)):
        m = ((flavors < 0) == anti) & (inttype == ('cc' if cc else 'nc'))
        if np.any(m):
            out[m] = _interp(energy[m], table)
    return out


def get_nu_cross_section(energy, flavors, inttype='total', cross_section_type='csms'):
    """get_nu_cross_section (:232-391) for the tabulated models; 'ctw' / 'ghandi' are evaluated on the device"""
    if cross_section_type == 'csms':
        return csms(energy, inttype, flavors)
    if cross_section_type == 'hedis_bgr18':
        raise NotImplementedError("hedis_bgr18 needs the reference's BGR18_dsigma_dy_H2O.npz download: pass the values as cross_section=")
    raise NotImplementedError("Cross-section {} not defined".format(cross_section_type))
