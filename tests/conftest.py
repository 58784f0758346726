import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of GPU time; runs only when asked for: -m 'gpu and slow' or NRHIP_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    """`slow` tests are opt-in: they run when the -m expression names them (or NRHIP_RUN_SLOW=1), never as part of a plain -m gpu"""
    if 'slow' in (config.getoption('-m') or '') or os.environ.get('NRHIP_RUN_SLOW', '0') not in ('', '0'):
        return
    skip = pytest.mark.skip(reason="slow: opt in with -m 'gpu and slow' or NRHIP_RUN_SLOW=1")
    for it in items:
        if 'slow' in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def max_rel(a, b):
    """max |a-b| / |b| over entries finite in both; NaN patterns must agree."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN padding differs"
    m = np.isfinite(a) & np.isfinite(b)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)))


@pytest.fixture(scope='session')
def gpu_ctx_factory():
    """Contexts on cuda:0 through the C ABI; fails loudly (no skip, no CPU fallback) if the HIP library
    or the GPU is missing."""
    import nuradiomc_amd
    made = []

    def make(ice, attenuation_model='SP1', **kw):
        c = nuradiomc_amd.Context(ice, attenuation_model, device=0, **kw)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
