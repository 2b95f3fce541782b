"""The libm table's cache file is only trusted on strong evidence: regular file, no symbolic
link, owned by the caller, not writable by group / others, expected size and header, and a checksum over the WHOLE payload.
Host side only (no GPU): the library exports the very checks it runs before an upload (`orbfe_debug_trig_cache_check`); the
device-side twin of the checksum is exercised by tests/test_gpu_extractor.py::test_corrupted_trig_cache_is_rebuilt."""
import ctypes as C
import os
import stat

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    import orb_slam3_detailed_comments_kor_amd as pkg
    lib = pkg.lib()
    lib.orbfe_debug_trig_cache_payload_bytes.restype = C.c_size_t
    lib.orbfe_debug_trig_cache_write.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    lib.orbfe_debug_trig_cache_check.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    lib.orbfe_debug_trig_cache_path.argtypes = [C.c_char_p, C.c_int]
    return lib


def _check(L, path):
    why = C.c_char_p()
    r = L.orbfe_debug_trig_cache_check(path.encode(), C.byref(why))
    return r, (why.value or b"").decode()


@pytest.fixture(scope="module")
def good(L, tmp_path_factory):
    d = tmp_path_factory.mktemp("trigcache")
    n = L.orbfe_debug_trig_cache_payload_bytes()
    assert 60e6 < n < 70e6
    payload = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    path = str(d / "codes.bin")
    assert L.orbfe_debug_trig_cache_write(path.encode(), payload.ctypes.data_as(C.c_void_p), n) == 0
    return path, n


def test_default_path_is_per_user_and_names_the_libm(L, monkeypatch):
    buf = C.create_string_buffer(512)
    monkeypatch.delenv("ORBFE_TRIG_CACHE", raising=False)
    n = L.orbfe_debug_trig_cache_path(buf, 512)
    assert n > 0 and buf.value.decode().startswith("/dev/shm/orbfe_trigcodes_u%d_" % os.geteuid())
    monkeypatch.setenv("ORBFE_TRIG_CACHE", "0")
    assert L.orbfe_debug_trig_cache_path(buf, 512) == 0


def test_written_file_is_private_and_passes(L, good):
    path, n = good
    st = os.stat(path)
    assert stat.S_IMODE(st.st_mode) == 0o600 and st.st_size == n + 32
    assert _check(L, path) == (0, "")
    assert not [f for f in os.listdir(os.path.dirname(path)) if ".tmp" in f]


def test_one_flipped_nibble_anywhere_is_rejected(L, good, tmp_path):
    path, n = good
    rng = np.random.default_rng(2)
    bad = str(tmp_path / "flipped.bin")
    data = bytearray(open(path, "rb").read())
    for off in [32, 32 + n - 1, 32 + 17, 32 + int(rng.integers(0, n)), 32 + int(rng.integers(0, n))]:  # first / last byte, the
        data[off] ^= 0x10                                  # byte round 3's sampled sum skipped, two random ones
        with open(bad, "wb") as f:
            f.write(data)
        os.chmod(bad, 0o600)
        r, why = _check(L, bad)
        assert r < 0 and "checksum" in why, (off, why)
        data[off] ^= 0x10
    with open(bad, "wb") as f:  # (restored: passes again)
        f.write(data)
    assert _check(L, bad)[0] == 0


def test_ownership_mode_link_size_and_header_checks(L, good, tmp_path):
    path, n = good
    data = open(path, "rb").read()
    p = str(tmp_path / "c.bin")
    open(p, "wb").write(data)
    os.chmod(p, 0o620)
    assert "writable" in _check(L, p)[1]
    os.chmod(p, 0o602)
    assert "writable" in _check(L, p)[1]
    os.chmod(p, 0o600)
    assert _check(L, p)[0] == 0
    link = str(tmp_path / "link.bin")
    os.symlink(p, link)
    r, why = _check(L, link)
    assert r < 0 and "link" in why
    open(p, "wb").write(data[:-8])
    assert "size" in _check(L, p)[1]
    hdr = bytearray(data)
    hdr[16] ^= 1  # the libm fingerprint
    open(p, "wb").write(hdr)
    assert "libm" in _check(L, p)[1]
    hdr = bytearray(data)
    hdr[7] = ord("3")  # round 3's format
    open(p, "wb").write(hdr)
    assert "format" in _check(L, p)[1]
    assert _check(L, str(tmp_path / "absent.bin"))[0] < 0


def test_store_does_not_follow_a_planted_temporary_link(L, good, tmp_path):
    path, n = good
    victim = str(tmp_path / "victim.txt")
    open(victim, "w").write("untouched")
    target = str(tmp_path / "new.bin")
    os.symlink(victim, target + ".tmp%d" % os.getpid())
    payload = np.zeros(n, np.uint8)
    assert L.orbfe_debug_trig_cache_write(target.encode(), payload.ctypes.data_as(C.c_void_p), n) == 0
    assert open(victim).read() == "untouched" and not os.path.islink(target) and _check(L, target)[0] == 0
