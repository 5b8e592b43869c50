"""CPU-side checks of the drop-in boundary: libshotfpfh.so loads and exports every symbol that
include/shotfpfh.h declares (no compute calls -- there is no GPU here), and the product fails loudly,
rather than falling back to anything, when no GPU is present."""
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "shotfpfh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from shot_fpfh_amd import _ffi

    lib = _ffi.load()
    names = header_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in shotfpfh.h but not exported"
    assert sorted(_ffi.SIGNATURES) == names, "ctypes prototype table and header disagree"
    assert b"gfx950" in lib.sf_version()


def test_no_silent_fallback_without_gpu():
    import shot_fpfh_amd as s
    from shot_fpfh_amd import _ffi

    if _ffi.load().sf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(s.ShotFpfhError, match="no CPU fallback"):
        s.Engine()
    import numpy as np

    with pytest.raises(s.ShotFpfhError):
        s.compute_fpfh_descriptor(np.arange(3), np.random.rand(10, 3), np.random.rand(10, 3), 0.5, 5, verbose=False)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under shot_fpfh_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "shot_fpfh_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"
