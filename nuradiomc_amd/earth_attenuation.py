"""Host-side mirror of NuRadioMC/utilities/earth_attenuation.py for the per-event Earth-absorption weight
(get_weight :12-60, the models PREM :133-240 and CoreMantleCrustModel :243-270), evaluated on the GPU for whole event
lists at once (nrhip_earth_weights_batch).  simulation.py:880-903 calls get_weight once per event group with scalars;
here the same arguments may be arrays of n events and the scalars of the reference are the n = 1 case.

cross_section_type 'ctw' (the reference's config_default.yaml) and 'ghandi' are evaluated on the device; the tabulated
'csms' (the published table) and 'hedis_bgr18' (the reference's data file, named by cross_sections.set_bgr18_file / NRHIP_BGR18_FILE)
go through nuradiomc_amd/cross_sections.py and, like any values the caller computed itself (`cross_section=`), reach the device as
per-event cross sections (NRHIP_XS_GIVEN).
"""
import numpy as np
from .context import Context

_ctx = None


def _default_context():
    global _ctx
    if _ctx is None:
        _ctx = Context((1.78, 0.423, 77.), 'SP1', device=0)  # the ice model is irrelevant for the weights
    return _ctx


def _layers(table):
    # units.g = 6.2415e33, units.cm = 0.01 (NuRadioReco/utilities/units.py); "13.0885 * units.g / units.cm ** 3 - 8.8381 * units.g / units.cm ** 3 * x ** 2": every coefficient is (c * g) / cm^3
    return np.array([[np.copysign(abs(c) * 6.241509744511525e+33 / 0.01 ** 3, c) for c in row] for row in table])


class PREM:
    """Preliminary reference Earth model (Dziewonski & Anderson 1981): density polynomials in x = r / earth_radius per
    radius range, as earth_attenuation.PREM (:133-169) tabulates them."""
    earth_radius = 6.3710e6
    radii = (1.2215e6, 3.4800e6, 5.7010e6, 5.7710e6, 5.9710e6, 6.1510e6, 6.3466e6, 6.3560e6, 6.3680e6, earth_radius)
    polynomials_g_cm3 = ((13.0885, 0., -8.8381, 0.), (12.5815, -1.2638, -3.6426, -5.5281), (7.9565, -6.4761, 5.5283, -3.0807),
                         (5.3197, -1.4836, 0., 0.), (11.2494, -8.0298, 0., 0.), (7.1089, -3.8045, 0., 0.), (2.691, 0.6924, 0., 0.),
                         (2.9, 0., 0., 0.), (2.6, 0., 0., 0.), (1.02, 0., 0., 0.))

    def __init__(self, ctx=None):
        self._ctx = ctx

    def _context(self):
        return self._ctx if self._ctx is not None else _default_context()

    def _model(self):
        return self.earth_radius, np.array(self.radii, float), _layers(self.polynomials_g_cm3)

    def slant_depth(self, endpoint, direction, step=500.):
        """Column density of the chord from `endpoint` along `direction` to the surface (:183-240); [3] or [n, 3] inputs"""
        endpoint = np.asarray(endpoint, float)
        scalar = endpoint.ndim == 1
        endpoint = endpoint.reshape(-1, 3)
        direction = np.broadcast_to(np.asarray(direction, float).reshape(-1, 3), endpoint.shape)
        n = len(endpoint)
        _, sd = self._context().earth_weights_batch(np.zeros(n), np.full(n, 1e18), np.full(n, 12), 2, endpoint=endpoint,
                                                    direction=direction, model=self._model(), step=step,
                                                    return_slant_depth=True)
        return float(sd[0]) if scalar else sd


class CoreMantleCrustModel(PREM):
    """Three layers of constant density, parameters from ARASim (earth_attenuation.CoreMantleCrustModel :243-270)"""
    earth_radius = 6.378140e6
    radii = (float(np.sqrt(1.2e13)), earth_radius - 4e4, earth_radius)
    polynomials_g_cm3 = ((14., 0., 0., 0.), (3.4, 0., 0., 0.), (2.9, 0., 0., 0.))


def get_weight(theta_nu, pnu, flavors, mode='simple', cross_section_type='ctw', vertex_position=None, phi_nu=None, ctx=None,
               cross_section=None):
    """Earth-absorption weight (earth_attenuation.get_weight :12-60): scalars as the reference takes them, or arrays of n
    events (theta_nu, pnu, flavors, phi_nu [n]; vertex_position [n, 3]).  cross_section [n] (m^2): total cross sections the caller
    evaluated, used instead of cross_section_type."""
    if mode == "None":
        return 1.
    if mode not in ('simple', 'core_mantle_crust_simple', 'core_mantle_crust', 'PREM'):
        raise NotImplementedError('mode {} not supported'.format(mode))
    scalar = np.ndim(theta_nu) == 0
    theta = np.atleast_1d(np.asarray(theta_nu, float))
    n = len(theta)
    pnu = np.broadcast_to(np.asarray(pnu, float), (n,))
    flavors = np.broadcast_to(np.asarray(flavors), (n,))
    if cross_section is None and cross_section_type not in ('ctw', 'ghandi'):
        # get_interaction_length asks for inttype='total' (cross_sections.py:393-421): 'csms' has no such rows and gives 0 -> weight 1
        from . import cross_sections
        cross_section = cross_sections.get_nu_cross_section(pnu, 0 if mode == 'simple' else flavors, 'total', cross_section_type)
    if cross_section is not None:
        pnu = np.broadcast_to(np.asarray(cross_section, float), (n,))
        cross_section_type = 'given'
    ctx = ctx if ctx is not None else _default_context()
    if mode == 'simple':
        w = ctx.earth_weights_batch(theta, pnu, flavors, 0, cross_section_type=cross_section_type)
    elif mode == 'core_mantle_crust_simple':
        w = ctx.earth_weights_batch(theta, pnu, flavors, 1, cross_section_type=cross_section_type)
    else:
        earth = CoreMantleCrustModel(ctx) if mode == 'core_mantle_crust' else PREM(ctx)
        phi = np.broadcast_to(np.asarray(phi_nu, float), (n,))
        # hp.spherical_to_cartesian(theta_nu, phi_nu)
        direction = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=1)
        vertex = np.asarray(vertex_position, float).reshape(n, 3)
        w = ctx.earth_weights_batch(theta, pnu, flavors, 2, endpoint=vertex, direction=direction, model=earth._model(),
                                    cross_section_type=cross_section_type)
    return float(w[0]) if scalar else w
