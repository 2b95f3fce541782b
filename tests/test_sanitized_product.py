"""SURVEY.md section 5 / VERDICT r04 #7: the PRODUCT's host code -- arenas, handle tables, the pinned-memory registry, trig-cache
file I/O, the shared-memory transport of orbfe_mc_*, every argument / no-device error path -- under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer, in the build container (no GPU: the GPU pool allows no sanitizer runs,
and the instrumentation is host-only, `-fno-gpu-sanitize`).  `make -C csrc asan | tsan` builds liborbfe_asan.so /
liborbfe_tsan.so beside the product library; they are loaded through ORBFE_LIB with the matching clang runtime preloaded and
never shipped."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _runtime(kind):
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.so" % kind)
    return hits[0] if hits else None


def _build(target):
    if not os.path.exists(CLANG) or _runtime(target) is None:
        pytest.skip("no clang %s runtime in this image" % target)
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-s", target])
    lib = os.path.join(PKG, "liborbfe_%s.so" % target)
    assert os.path.exists(lib)
    return lib


def _exerciser(tmp_path, target, lib):
    exe = str(tmp_path / ("threads_cabi_" + target))
    rt = _runtime(target)
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=" + ("thread" if target == "tsan" else "address,undefined"),
                           "-shared-libsan", os.path.join(ROOT, "tests", "san", "threads_cabi.cpp"), "-o", exe, "-L" + PKG,
                           "-l:" + os.path.basename(lib), "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath," + os.path.dirname(rt), "-pthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", ORBFE_TRIG_CACHE=str(tmp_path))
    out = subprocess.run([exe, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    return out


def _clean(out):
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail
    for needle in ("AddressSanitizer", "ThreadSanitizer", "LeakSanitizer", "runtime error:"):
        assert needle not in out.stdout, tail


def test_host_surface_from_four_threads_under_tsan(tmp_path):
    lib = _build("tsan")
    out = _exerciser(tmp_path, "tsan", lib)
    _clean(out)
    assert "threads_cabi: 0 failures" in out.stdout


def test_host_surface_from_four_threads_under_asan_and_ubsan(tmp_path):
    lib = _build("asan")
    out = _exerciser(tmp_path, "asan", lib)
    _clean(out)  # (with leak detection: every error path gives back what it took)
    assert "threads_cabi: 0 failures" in out.stdout


def test_cpu_reachable_tests_pass_under_asan_and_ubsan():
    """The CPU tests that drive the library without a GPU -- exports and argument errors, the trig cache's file checks (a
    flipped nibble anywhere, links, modes, truncation), the world-2 host transport of orbfe_mc_* across two processes, the
    shard / ring / job bookkeeping -- once more through the instrumented library."""
    lib = _build("asan")
    env = dict(os.environ, LD_PRELOAD=_runtime("asan"), ORBFE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",  # (python itself leaks by design)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_cabi.py"), os.path.join(ROOT, "tests", "test_trig_cache.py"),
                          os.path.join(ROOT, "tests", "test_multicam_gloo.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT)
    _clean(out)
    assert " passed" in out.stdout
