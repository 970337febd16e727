"""The C-ABI library loads and exports every symbol the header declares (no GPU needed)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from nerf_amd import _lib
    declared = _declared_symbols()
    assert declared, "no declarations found"
    assert sorted(_lib.SIGNATURES) == declared


def test_library_exports_all_symbols():
    from nerf_amd import _lib
    handle = _lib.lib()  # raises if the library is missing or a symbol is absent
    assert handle.nerf_abi_version() == 4
    for name in _declared_symbols():
        assert hasattr(handle, name), name


def test_cpu_tensors_are_rejected():
    from nerf_amd import ops
    from nerf_amd._lib import NerfHipError
    with pytest.raises(NerfHipError):
        ops.encode_pe(torch.zeros(4, 3), 4)
    with pytest.raises(NerfHipError):
        ops.composite(torch.zeros(2, 4, 3), torch.zeros(2, 4), torch.zeros(2, 4))
