"""CPU test: the libraries load and export exactly what their headers declare (no compute) -- libsfx.so / include/sfx.h (the
product: one form of every step, no environment switch) and, when it has been built, libsfx_lab.so / include/sfx_lab.h."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "smplify-x-partial_amd")


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(sfx_[a-z_0-9]+)\s*\(", hdr))


def _exported(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    return set(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("sfx_"))


def test_library_exports_every_declared_symbol():
    from smplifyx_amd import _capi
    lib = _capi.load()
    declared = _declared("sfx.h")
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    assert lib.sfx_version().decode().startswith("sfx")


def test_product_library_is_not_the_laboratory():
    """libsfx.so exports include/sfx.h and nothing else, does not import getenv and holds no SFX_* switch name."""
    from smplifyx_amd import _capi
    path = os.path.join(PKG, "libsfx.so")
    assert _exported(path) == _declared("sfx.h") == set(_capi.SYMBOLS)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    blob = open(path, "rb").read()
    switches = set(re.findall(r"\b(SFX_[A-Z][A-Z_0-9]+)\b", open(os.path.join(ROOT, "include", "sfx_lab.h")).read())) - {"SFX_LAB", "SFX_LAB_H_", "SFX_LIB"}
    assert len(switches) > 10
    assert not [n for n in switches if n.encode() in blob]
    assert not (_declared("sfx_lab.h") & _exported(path))


def test_lab_library_exports_both_headers():
    from smplifyx_amd import _capi
    path = os.path.join(PKG, "libsfx_lab.so")
    if not os.path.exists(path):
        pytest.skip("libsfx_lab.so not built (SFX_LAB=1 bash smplify-x-partial_amd/csrc/build.sh)")
    lab = _declared("sfx_lab.h")
    assert lab == set(_capi.LAB_SYMBOLS), lab ^ set(_capi.LAB_SYMBOLS)
    assert _exported(path) == _declared("sfx.h") | lab
    ctypes.CDLL  # (loading is left to SFX_LIB=...: two copies of the library in one process would share nothing)


def test_model_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smplifyx_amd import _capi, engine, synthetic
    with pytest.raises(_capi.SfxError):
        engine.DeviceModel(synthetic.make_synthetic_model(0))
