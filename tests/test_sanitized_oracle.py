"""the CPU oracle (test infrastructure, but the thing every parity claim rests on) and the host-side
adapter code under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: the GPU pool allows no sanitizer runs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.check_output(["g++", "-print-file-name=" + name], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan.so next to g++")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ORB_ORACLE_LIB=os.path.join(ROOT, "oracle", "liborb_oracle_asan.so"),
               # python itself leaks by design; abort on the first real finding
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_primitives.py"), os.path.join(ROOT, "tests", "test_oracle_octree.py"),
                          os.path.join(ROOT, "tests", "test_oracle_matcher.py"), os.path.join(ROOT, "tests", "test_golden.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in out.stdout and "runtime error:" not in out.stdout, tail
    assert " passed" in out.stdout


@pytest.mark.parametrize("src", ["test_adapter.cpp", "test_matcher_adapter.cpp"])
def test_adapter_host_code_builds_with_sanitizers(src, tmp_path):
    """The C++ callers written against adapters/ORBextractor.h / ORBmatcher.h compile and link with -fsanitize (they need
    a GPU to run; the GPU suite runs the plain builds)."""
    libdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    if not os.path.exists(os.path.join(libdir, "liborbfe.so")):
        pytest.skip("liborbfe.so not built")
    exe = str(tmp_path / "a.out")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-Wall",
                           "-I" + os.path.join(ROOT, "adapters"), os.path.join(ROOT, "adapters", src), "-o", exe,
                           "-L" + libdir, "-lorbfe", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)
