"""The adapters' ORBFE_HAVE_OPENCV branches and the OpenCV differential harness seen by a compiler: `g++ -fsyntax-only`
against a DECLARATION-ONLY mock of the few OpenCV types they touch (tests/opencv_mock, never linked or shipped).  This image
has no OpenCV; the branches used to be dead text."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "opencv_mock")


def _syntax(src, *extra):
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + MOCK, *extra, src], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-3000:]


def test_extractor_adapter_opencv_branches_parse():
    _syntax(os.path.join(MOCK, "use_adapters.cpp"))


def test_extractor_adapter_standin_branches_still_parse():
    # the same header without OpenCV on the include path (what the GPU tests compile)
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DORBFE_NO_OPENCV=1", "-x", "c++",
                          os.path.join(ROOT, "adapters", "ORBextractor.h")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-3000:]


def test_opencv_differential_harness_parses():
    _syntax(os.path.join(ROOT, "adapters", "diff_opencv.cpp"))


def test_cmake_snippet_skips_quietly_without_opencv(tmp_path):
    if shutil.which("cmake") is None:
        pytest.skip("no cmake")
    out = subprocess.run(["cmake", "-S", os.path.join(ROOT, "adapters"), "-B", str(tmp_path)], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "OpenCV not found" in out.stdout or "diff_opencv" in out.stdout


def test_vocabulary_adapter_parses_with_and_without_opencv():
    # adapters/ORBVocabulary.h: the stand-in branch (what tests/test_gpu_vocabulary_adapter.py compiles) and against the mock
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DORBFE_NO_OPENCV=1", "-x", "c++",
                          os.path.join(ROOT, "adapters", "ORBVocabulary.h")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    _syntax(os.path.join(MOCK, "use_vocabulary.cpp"))
