"""Host-side sequencing of the reference's event loop that the batched device path cannot express per ray:

* the ORDER in which `simulation.run()` meets the showers -- event group -> station -> channel -> shower -> ray solution
  (NuRadioMC/simulation/simulation.py:1454-1600 with calculate_sim_efield :143-242) -- which fixes the sequence of the
  stateful random draws of the emission models: Alvarez2009's k_L for electromagnetic showers
  (NuRadioMC/SignalGen/parametrizations.py:90-91, :160-173: one np.random.RandomState(seed) per model, one normal() per
  shower at its first ray that survives the delta_C cut, stored in the shower and reused, simulation.py:221-242) and the
  ARZ profile number (ARZ.py:561-591: one randint() per shower);
* `split_event_time_diff` (simulation.py:906-947, group_into_events): the rays of one event group are split into
  sub-events wherever two consecutive signal start times are further apart than the configured gap.

Pure numpy; the device only supplies `shower_first_channel` (nrhip_sim_config.select_only).
"""
import numpy as np


def reference_draw_order(first_channel, group_begin=None):
    """Indices of the showers in the order in which the reference draws their random emission parameters.

    first_channel: int [n_stations, n_showers] (or [n_showers]) -- per station the first channel on which the shower has a ray
    that passes the cuts, -1 = none (Station.fetch('shower_first_channel') after a select_only pass).  A shower is drawn
    when the loops (group, station, channel, shower) reach its first such ray; showers without any are never drawn."""
    fc = np.atleast_2d(np.asarray(first_channel))
    n_st, n = fc.shape
    has = fc >= 0
    any_ = has.any(axis=0)
    first_st = np.argmax(has, axis=0)
    ch = fc[first_st, np.arange(n)]
    if group_begin is None:
        group = np.arange(n)
    else:
        gb = np.asarray(group_begin).reshape(-1)
        group = np.repeat(np.arange(len(gb) - 1), np.diff(gb))
    order = np.lexsort((np.arange(n), ch, first_st, group))
    return order[any_[order]]


def alvarez2009_k_L_distribution(energy):
    """(log10 mean, sigma) of k_L of an electromagnetic shower (parametrizations.py:141-158)"""
    log10_E_0 = np.log10(energy)
    sigma = 3.39e-2 + (0. if log10_E_0 < 14.99 else 2.25e-2) * (log10_E_0 - 14.99)
    mean = 1.52 + (5.59e-2 if log10_E_0 < 16.61 else 0.39) * (log10_E_0 - 16.61)
    return mean, sigma


def draw_k_L(k_L, energy, shower_type_codes, order, rng):
    """Fill the NaN entries of k_L for the EM showers (code 1) of `order`, one rng.normal per shower in that order
    (parametrizations.py:172).  Hadronic showers get the deterministic value the reference stores for them (:134-139)."""
    k_L = np.array(k_L, float)
    energy = np.broadcast_to(np.asarray(energy, float), k_L.shape)
    for i in order:
        if shower_type_codes[i] == 1:
            if np.isnan(k_L[i]):
                mean, sigma = alvarez2009_k_L_distribution(energy[i])
                k_L[i] = 10 ** rng.normal(mean, sigma)
        elif np.isnan(k_L[i]):
            k_L[i] = 31.25 * (energy[i] / 1.e15) ** 3.01e-2
    return k_L


def split_event_times(start_times, split_event_time_diff):
    """group_into_events (simulation.py:906-947): sort the signal start times of an event group, cut where the gap between
    two consecutive ones exceeds split_event_time_diff.  Returns the sub-event index (0, 1, ...) of every entry."""
    t = np.asarray(start_times, float)
    if len(t) == 0:
        return np.zeros(0, int)
    srt = np.argsort(t)
    cuts = np.concatenate([[0], np.cumsum(np.diff(t[srt]) > split_event_time_diff)])
    out = np.empty(len(t), int)
    out[srt] = cuts
    return out
