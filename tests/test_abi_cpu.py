"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/salsa_hip.h declares, and its host
helpers (bin limits, compression matrix) agree with the reference goldens.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope='module')
def lib():
    from salsa_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'salsa_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(salsa_[a-z_]+)\s*\(', hdr))
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), 'libsalsa_hip.so does not export %s' % n
    from salsa_amd import _lib
    assert set(_lib.EXPORTS) == names
    assert lib.salsa_abi_version() == 2
    gru = open(os.path.join(ROOT, 'include', 'salsa_gru.h')).read()
    gru = re.sub(r'/\*.*?\*/', '', gru, flags=re.S)
    gnames = set(re.findall(r'\b(salsa_gru_[a-z_]+)\s*\(', gru))
    assert gnames == set(_lib.GRU_EXPORTS) and all(hasattr(lib, n) for n in gnames)
    nn = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'salsa_nn.h')).read(), flags=re.S)
    nnames = set(re.findall(r'\b(salsa_nn_[a-z0-9_]+)\s*\(', nn))
    assert nnames == set(_lib.NN_EXPORTS) and all(hasattr(lib, n) for n in nnames)


def test_host_helpers_match_reference(lib):
    from salsa_amd.extractor import bin_limits, compress_matrix
    meta, a = load_golden('g5_w_bins')
    for fs, n_fft, fmin, fmax, lo, up, cut in meta['bins']:
        assert bin_limits(fs, n_fft, fmin, fmax) == (lo, up, cut)
    for n_fft in (512, 256):
        for comp, tag in ((True, 'c'), (False, 'n')):
            assert np.array_equal(compress_matrix(n_fft, comp), a['W_%d_%s' % (n_fft, tag)])
    with pytest.raises(AssertionError):
        compress_matrix(1024)


def test_params_struct_layout_matches_header():
    from salsa_amd import _lib
    # 6 ints, double (8-aligned), 7 ints, pad -> 64 bytes with natural alignment
    assert C.sizeof(_lib.SalsaParams) == 80
    assert _lib.SalsaParams.cond_num.offset == 24
