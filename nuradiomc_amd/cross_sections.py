"""Host-side mirror of NuRadioMC/utilities/cross_sections.py for the tabulated neutrino-nucleon cross sections that the device
does not evaluate itself ('ctw' and 'ghandi' are device code, csrc/earth.hip).

'csms' (Cooper-Sarkar, Mertsch, Sarkar, JHEP 08 (2011) 042; cross_sections.py:123-229): the published table -- energy [GeV], CC and
NC cross section [pb] for neutrinos and antineutrinos -- interpolated linearly in the energy (scipy interp1d, bounds_error=True: a
ValueError outside 50 GeV ... 5e11 GeV).  As in the reference only the per-interaction values exist: inttype 'cc' / 'nc' (arrays);
for inttype='total' the reference's csms() matches no row and returns zeros (cross_sections.py:213-227), which get_weight turns into
weight 1 -- reproduced here, and said so, rather than silently summed.

'hedis_bgr18' (:17-61, :276-299; arXiv:2004.04756): the differential cross section d sigma / dy per H2O molecule [cm^2] on a
(flavor, nc/cc, energy, y) grid, read from the data file the reference downloads on first use (BGR18_dsigma_dy_H2O.npz: arrays
dsigma_dy_ref, flavors_ref, nu_energies_ref, y_ref, ncccs_ref).  The file is not part of this repository (no network here): the
caller names it with `set_bgr18_file(path)`, the environment variable NRHIP_BGR18_FILE, or puts it at
nuradiomc_amd/data/BGR18_dsigma_dy_H2O.npz; without it get_nu_cross_section('hedis_bgr18') raises FileNotFoundError.  The values
are integrated over y as a piecewise power law (integrate_pwpl :424-537, extended to y = 0 and y = 1), 'total' = nc + cc, and
interpolated in the energy linearly in log10 sigma (ValueError outside the table).  The arithmetic is pinned on the reference run
on a synthetic table of the file's layout (tests/golden/ref_hedis.npz), not on the data file itself.

Whatever the model, the values reach the device as per-event cross sections through
`earth_attenuation.get_weight(..., cross_section=sigma)` / NRHIP_XS_GIVEN.
"""
import os
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


def integrate_power_law(y, x, low=None, high=None, cumulative=False):
    """Integral over the last axis of y(x) taken as A_i x^b_i between neighbouring nodes (the reference's integrate_pwpl,
    cross_sections.py:424-537; cumulative=True is its full_output: (total, (running integral from the lower limit, its nodes))):
    per interval y_i x_i (r^(b+1) - 1) / (b + 1) with r = x_{i+1} / x_i and b = ln(y_{i+1} / y_i) / ln r; intervals with a zero at either end contribute nothing; `low` / `high` extend the first / last
    interval's power law beyond the nodes (low = 0 needs b > -1 there, else ValueError as in the reference)."""
    y = np.asarray(y, float)
    x = np.asarray(x, float)
    y0, y1 = y[..., :-1], y[..., 1:]
    x0, x1 = x[:-1], x[1:]
    dead = (y0 == 0) | (y1 == 0)
    lr = np.log(x1 / x0)
    with np.errstate(divide='ignore', invalid='ignore'):
        b1 = np.where(dead, 0., np.log(np.where(dead, 1., y1 / np.where(y0 == 0, 1., y0))) / lr) + 1.   # b + 1

    def piece(ya, xa, lnr, e):
        # ya xa (exp(e lnr) - 1) / e, with the limit ya xa lnr at e -> 0
        t = e * lnr
        small = np.abs(t) < 1e-12
        return ya * xa * np.where(small, lnr, np.expm1(t) / np.where(small, 1., e))

    parts = np.where(dead, 0., piece(y0, x0, lr, b1))
    if low is not None:
        if low < 0:
            raise ValueError("Cannot use power-law integration for negative values.")
        e = b1[..., 0]
        if low == 0:
            if np.any(e <= 0):
                raise ValueError("Cannot integrate to x=0 because min(slope) {} <= -1".format(np.min(e) - 1))
            ext = y[..., 0] * x[0] / e
        else:
            ext = -piece(y[..., 0], x[0], np.log(low / x[0]), e)
        parts = np.concatenate([np.where(dead[..., 0], 0., ext)[..., None], parts], axis=-1)
        x = np.concatenate([[low], x])
    if high is not None:
        ext = piece(y[..., -1], x[-1], np.log(high / x[-1]), b1[..., -1])
        parts = np.concatenate([parts, np.where(dead[..., -1], 0., ext)[..., None]], axis=-1)
        x = np.concatenate([x, [high]])
    total = np.sum(parts, axis=-1)
    if cumulative:
        run = np.cumsum(parts, axis=-1)
        return total, (np.concatenate([np.zeros(run.shape[:-1] + (1,)), run], axis=-1), x)
    return total


_bgr18_file = None
_bgr18_cache = {}


def set_bgr18_file(path):
    """names the BGR18_dsigma_dy_H2O.npz file 'hedis_bgr18' reads (the reference downloads it into NuRadioMC/utilities/data)"""
    global _bgr18_file
    _bgr18_file = path


def _bgr18_table():
    path = _bgr18_file or os.environ.get('NRHIP_BGR18_FILE') or \
        os.path.join(os.path.dirname(__file__), 'data', 'BGR18_dsigma_dy_H2O.npz')
    if path not in _bgr18_cache:
        if not os.path.exists(path):
            raise FileNotFoundError("hedis_bgr18 needs the reference's BGR18_dsigma_dy_H2O.npz ({} does not exist): name it with "
                                    "cross_sections.set_bgr18_file() or NRHIP_BGR18_FILE, or pass cross_section=".format(path))
        d = np.load(path)
        # per nucleon: the file holds cm^2 per H2O molecule, 18 nucleons (:28-31)
        dsdy, yy = d['dsigma_dy_ref'] * (1e-4 / 18.), np.asarray(d['y_ref'], float)
        sig, (run, y_ext) = integrate_power_law(dsdy, yy, low=0, high=1, cumulative=True)          # [flavor, nc/cc, energy(, y)]
        kinds = [str(k).lower() for k in d['ncccs_ref']] + ['total']
        _bgr18_cache.clear()
        _bgr18_cache[path] = dict(energy=np.asarray(d['nu_energies_ref'], float), flavors=np.asarray(d['flavors_ref']), kinds=kinds,
                                  sigma=np.concatenate([sig, sig[:, :1] + sig[:, 1:2]], axis=1),
                                  cdf=run / sig[..., None], y=y_ext)
    return _bgr18_cache[path]


def bgr18_inelasticity_cdf(flavor, nccc, i_energy):
    """(cdf, y) of the inelasticity at energy node i_energy for one flavor and 'cc' | 'nc': the running integral of d sigma / dy
    from y = 0, normalised (inelasticities._get_inverse_cdf_interpolation :99-105)"""
    t = _bgr18_table()
    fi = np.flatnonzero(t['flavors'] == flavor)
    if len(fi) != 1:
        raise ValueError("hedis_bgr18 has no neutrino flavor {}".format(flavor))
    return t['cdf'][fi[0], t['kinds'].index(str(nccc).lower()), i_energy], t['y']


def hedis_bgr18(energy, flavors, inttype='total'):
    """get_nu_cross_section(..., 'hedis_bgr18') (:276-299): cross section per nucleon [m^2] for arrays (or scalars) energy /
    flavors / inttype ('cc' | 'nc' | 'total'); a flavor that is not in the file (0, as mode 'simple' of get_weight passes) is a
    ValueError -- the reference fails on it too (an index of an empty argwhere)."""
    t = _bgr18_table()
    e_ref, flav_ref, kinds, sig = t['energy'], t['flavors'], t['kinds'], t['sigma']
    scalar = np.ndim(energy) == 0
    energy = np.atleast_1d(np.asarray(energy, float))
    if np.any(energy > e_ref[-1]):
        raise ValueError("Exceeding energy limit of BGR18 cross-section parameterization (E_lim = {:.2e} eV). "
                         "Please use a different cross-section model.".format(e_ref[-1]))
    if np.any(energy < e_ref[0]):
        raise ValueError("A value in x_new is below the interpolation range.")     # interp1d(bounds_error=True)
    flavors = np.broadcast_to(np.asarray(flavors), energy.shape)
    inttype = np.char.lower(np.broadcast_to(np.asarray(inttype, str), energy.shape))
    out = np.zeros_like(energy)
    for f in np.unique(flavors):
        fi = np.flatnonzero(flav_ref == f)
        if len(fi) != 1:
            raise ValueError("hedis_bgr18 has no neutrino flavor {}".format(f))
        for it in np.unique(inttype):
            if it not in kinds:
                raise ValueError("hedis_bgr18 has no interaction type {}".format(it))
            m = (flavors == f) & (inttype == it)
            out[m] = 10 ** np.interp(energy[m], e_ref, np.log10(sig[fi[0], kinds.index(it)]))
    return float(out[0]) if scalar else out


def get_nu_cross_section(energy, flavors, inttype='total', cross_section_type='csms'):
    """get_nu_cross_section (:232-391) for the tabulated models; 'ctw' / 'ghandi' are evaluated on the device"""
    if cross_section_type == 'csms':
        return csms(energy, inttype, flavors)
    if cross_section_type == 'hedis_bgr18':
        return hedis_bgr18(energy, flavors, inttype)
    raise NotImplementedError("Cross-section {} not defined".format(cross_section_type))
