"""The C-ABI library loads and exports every symbol include/nrhip.h declares (no GPU needed)."""
import ctypes
import os
import re
import pytest
from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'nrhip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nrhip_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_entry_points():
    names = _declared()
    assert 'nrhip_find_solutions_batch' in names and 'nrhip_attenuation_batch' in names
    assert len(names) >= 10


def test_library_exports_all_declared_symbols():
    so = os.path.join(ROOT, 'nuradiomc_amd', 'lib', 'libnrhip.so')
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(so)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_header():
    from nuradiomc_amd import _lib
    lib = _lib.load()
    for n in _declared():
        assert getattr(lib, n).argtypes is not None or n in ('nrhip_last_error', 'nrhip_device_count'), n


def test_no_oracle_in_product():
    """The product package must never import / load anything under oracle/."""
    pkg = os.path.join(ROOT, 'nuradiomc_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle' not in txt.lower().replace('no oracle', ''), os.path.join(dp, f)
