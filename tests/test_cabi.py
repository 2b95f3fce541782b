"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import orb_slam3_detailed_comments_kor_amd as pkg
    L = pkg.lib()
    names = _declared()
    assert len(names) >= 26
    for n in names:
        assert hasattr(L, n), n
    from orb_slam3_detailed_comments_kor_amd import binding
    assert sorted(binding.EXPORTS) == names
    assert b"gfx950" in L.orbfe_version()


def test_keypoint_layout_matches_cv_keypoint():
    from orb_slam3_detailed_comments_kor_amd import KP_DTYPE
    assert KP_DTYPE.itemsize == 28
    assert [KP_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_no_cpu_fallback_and_argument_errors():
    import torch
    import orb_slam3_detailed_comments_kor_amd as pkg
    L = pkg.lib()
    h = ctypes.c_void_p()
    # bad parameters are rejected before any device is touched
    assert L.orbfe_create(ctypes.byref(h), 0, 1.2, 8, 20, 7, 0) == pkg.binding.ERR_ARGS
    assert L.orbfe_create(ctypes.byref(h), 1000, 1.0, 8, 20, 7, 0) == pkg.binding.ERR_ARGS
    assert L.orbfe_create(ctypes.byref(h), 1000, 1.2, 0, 20, 7, 0) == pkg.binding.ERR_ARGS
    if not torch.cuda.is_available():
        # the product never falls back to a CPU path
        with pytest.raises(pkg.OrbfeError) as e:
            pkg.ORBextractor(1000)
        assert e.value.code == pkg.binding.ERR_NODEV
        with pytest.raises(pkg.OrbfeError):
            pkg.hamming_pairs(np.zeros((2, 32), np.uint8), np.zeros((2, 32), np.uint8))
    # empty operands need no device
    assert pkg.hamming_pairs(np.zeros((0, 32), np.uint8), np.zeros((3, 32), np.uint8)).shape == (0, 3)


def test_product_does_not_import_the_oracle():
    pkgdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orb_oracle" not in txt and "oracle/" not in txt.replace("the oracle/", ""), os.path.join(dirpath, f)
