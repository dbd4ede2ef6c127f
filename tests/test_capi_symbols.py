"""CPU test: libsfx.so loads and exports every symbol include/sfx.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from smplifyx_amd import _capi
    lib = _capi.load()
    hdr = open(os.path.join(ROOT, "include", "sfx.h")).read()
    declared = set(re.findall(r"\b(sfx_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    assert lib.sfx_version().decode().startswith("sfx")


def test_model_create_fails_loudly_without_gpu():
    import torch
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smplifyx_amd import _capi, engine, synthetic
    with pytest.raises(_capi.SfxError):
        engine.DeviceModel(synthetic.make_synthetic_model(0))
