"""CPU-side checks of the drop-in boundary: libgga_hip.so loads and exports every symbol
include/gga_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from gga_amd import _lib


def _declared_symbols():
    src = open(os.path.join(REPO, 'include', 'gga_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gga_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header_symbols():
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/gga_hip.h but not exported'
    assert set(names) == set(_lib.SIGNATURES), 'python binding and header disagree'


def test_abi_version_and_grid_size():
    L = _lib.lib()
    assert L.gga_abi_version() == _lib.ABI_VERSION
    from gga_amd.functional import voxel_grid_size, voxel_params
    assert voxel_grid_size(voxel_params([0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1], 5, 16000)) == [1408, 1600, 40]
    assert voxel_grid_size(voxel_params([0.16, 0.16, 4], [0, -39.68, -3, 69.12, 39.68, 1], 32, 16000)) == [432, 496, 1]


def test_argument_validation_without_gpu():
    # invalid arguments are rejected before any HIP call, with a message
    L = _lib.lib()
    rc = L.gga_voxel_mean(None, None, 10, 5, 4, 4, None, None)
    assert rc == -1 and b'null pointer' in L.gga_last_error()
    rc = L.gga_pillar_scatter_fwd(None, None, 0, None, 1, 6, 4, 4, 0, 0, 1, 1, None)
    assert rc == -1 and b'multiple of 4' in L.gga_last_error()


def test_product_refuses_cpu_tensors():
    import torch
    from gga_amd import functional as F
    with pytest.raises(RuntimeError, match='GPU only'):
        F.voxel_mean(torch.zeros(2, 5, 4), torch.ones(2, dtype=torch.int32), 4)


def test_product_never_imports_oracle():
    # the oracle is test infrastructure; the product tree must not reference it
    for root, _, files in os.walk(os.path.join(REPO, 'gga_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.cc', '.h')):
                txt = open(os.path.join(root, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'gga_oracle' not in txt, f
