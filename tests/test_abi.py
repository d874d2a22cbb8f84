"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/sdcmi.h declares; the ctypes
prototype table covers them all; calls fail loudly without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sdcmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sdc_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    import __graft_entry__ as g

    g.build()
    from pysdc_amd import lib

    L = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/sdcmi.h but not exported'
        assert s in lib.PROTOTYPES, f'{s} has no ctypes prototype in pysdc_amd/lib.py'
    for s in lib.PROTOTYPES:
        assert s in syms, f'{s} bound in lib.py but not declared in include/sdcmi.h'


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pysdc_amd import lib
    from pysdc_amd.engine import SweepEngine
    from pysdc_amd.errors import EngineError

    assert lib.load().sdc_version() >= 100
    with pytest.raises(EngineError):
        SweepEngine((8, 8, 8), 3)
    # parameter validation happens before any device call
    from pysdc_amd.errors import ParameterError

    with pytest.raises(ParameterError):
        SweepEngine((7, 7), 3)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'pysdc_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
