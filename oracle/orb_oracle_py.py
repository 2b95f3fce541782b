"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

TRIG_LIBM, TRIG_CR = 0, 1


class _FV(C.Structure):
    _fields_ = [("nn", C.c_int), ("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        # ORB_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle asan`)
        path = os.environ.get("ORB_ORACLE_LIB") or os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orb_oracle_create.restype = C.c_void_p
        L.orb_oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orb_oracle_destroy.argtypes = [C.c_void_p]
        L.orb_oracle_set_gauss_taps.argtypes = [C.c_void_p, C.c_void_p]
        L.orb_oracle_set_trig_mode.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_set_atan_fma.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_fast_atan2_fma.restype = C.c_float
        L.orb_oracle_fast_atan2_fma.argtypes = [C.c_float, C.c_float]
        L.orb_oracle_extract.restype = C.c_int
        L.orb_oracle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orb_oracle_extract_many.restype = C.c_long
        L.orb_oracle_extract_many.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_double)]
        L.orb_oracle_get_scale_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orb_oracle_get_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
        L.orb_oracle_get_umax.argtypes = [C.c_void_p, C.c_void_p]
        for f in (L.orb_oracle_get_level, L.orb_oracle_get_blurred):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                          C.POINTER(C.c_size_t)]
        for f in (L.orb_oracle_get_candidates, L.orb_oracle_get_level_keypoints):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orb_oracle_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                               C.c_int, C.c_size_t]
        L.orb_oracle_border_reflect101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int]
        L.orb_oracle_fast.restype = C.c_int
        L.orb_oracle_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orb_oracle_fast_score_closed.restype = C.c_int
        L.orb_oracle_fast_score_closed.argtypes = [C.c_void_p, C.c_size_t]
        L.orb_oracle_fast_score_2loop.restype = C.c_int
        L.orb_oracle_fast_score_2loop.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.orb_oracle_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                                C.c_void_p]
        L.orb_oracle_fast_atan2.restype = C.c_float
        L.orb_oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orb_oracle_sincos_cr.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orb_oracle_distribute_octree.restype = C.c_int
        L.orb_oracle_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_int]
        L.orb_oracle_descriptor_distance.restype = C.c_int
        L.orb_oracle_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orb_oracle_hamming_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orb_oracle_bfknn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orb_oracle_three_maxima.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]
        L.orb_oracle_search_bow_kf_f.restype = C.c_int
        L.orb_oracle_search_bow_kf_f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_FV),
                                                 C.c_void_p, C.c_int, C.c_void_p, C.POINTER(_FV), C.c_int, C.c_float,
                                                 C.c_int, C.c_void_p]
        L.orb_oracle_search_bow_kf_kf.restype = C.c_int
        L.orb_oracle_search_bow_kf_kf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_FV),
                                                  C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.POINTER(_FV), C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orb_oracle_search_triangulation.restype = C.c_int
        L.orb_oracle_search_triangulation.argtypes = (
            [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_FV)] * 2
            + [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
        L.orb_oracle_compute_stereo_matches.restype = C.c_int
        L.orb_oracle_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                        C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                                        C.c_void_p, C.c_void_p]
        L.orb_oracle_search_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orb_oracle_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orb_oracle_vocab_transform.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
        L.orb_oracle_kb8_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fv(fv):
    node_ids, offsets, indices = fv
    node_ids = np.ascontiguousarray(node_ids, np.uint32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    indices = np.ascontiguousarray(indices, np.int32)
    s = _FV(len(node_ids), node_ids.ctypes.data, offsets.ctypes.data, indices.ctypes.data)
    s._keep = (node_ids, offsets, indices)
    return s


class Extractor:
    """Oracle ORBextractor (reference src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, trig=TRIG_LIBM, taps=None, native=False):
        self.L = (_native_for_extractor() if native else None) or lib()
        self.h = self.L.orb_oracle_create(nfeatures, scale, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad ORBextractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.L.orb_oracle_set_trig_mode(self.h, trig)
        if taps is not None:
            t = np.ascontiguousarray(taps, np.int32)
            self.L.orb_oracle_set_gauss_taps(self.h, _p(t))

    def set_atan_fma(self, on=True):
        self.L.orb_oracle_set_atan_fma(self.h, int(on))

    def set_fastpath(self, on=True):
        """Timing-only SIMD FAST prefilter + vector-friendly blur (identical results; never the checker of a parity test).
        Returns True when the build has the AVX2 prefilter."""
        self.L.orb_oracle_set_fastpath.restype = C.c_int
        self.L.orb_oracle_set_fastpath.argtypes = [C.c_void_p, C.c_int]
        return bool(self.L.orb_oracle_set_fastpath(self.h, int(on)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orb_oracle_destroy(self.h)
            self.h = None

    def extract(self, img, lap=(0, 0), cap=None):
        """Returns (monoIndex, kps[KP_DTYPE], desc[n,32])."""
        img = np.asarray(img)
        if img.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
        cap = cap or (self.nfeatures + 64 + 16 * self.nlevels)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        r = self.L.orb_oracle_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], lap[0], lap[1],
                                      _p(kps), _p(desc), cap, C.byref(n))
        if r < -1:
            raise RuntimeError("oracle extract failed: %d" % r)
        return r, kps[: n.value].copy(), desc[: n.value].copy()

    def scale_tables(self):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        self.L.orb_oracle_get_scale_tables(self.h, *[_p(o) for o in out])
        return out

    def features_per_level(self):
        o = np.zeros(self.nlevels, np.int32)
        self.L.orb_oracle_get_features_per_level(self.h, _p(o))
        return o

    def umax(self):
        o = np.zeros(16, np.int32)
        self.L.orb_oracle_get_umax(self.h, _p(o))
        return o

    def _img(self, fn, level):
        d, r, c, s = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t()
        if fn(self.h, level, C.byref(d), C.byref(r), C.byref(c), C.byref(s)) != 0:
            return None
        buf = (C.c_uint8 * (r.value * s.value)).from_address(d.value)
        a = np.frombuffer(buf, np.uint8).reshape(r.value, s.value)[:, : c.value]
        return a.copy()

    def level(self, level):
        """Padded level (rows+38, cols+38), i.e. the buffer behind mvImagePyramid[level]."""
        return self._img(self.L.orb_oracle_get_level, level)

    def blurred(self, level):
        return self._img(self.L.orb_oracle_get_blurred, level)

    def _kps(self, fn, level):
        p = C.c_void_p()
        n = fn(self.h, level, C.byref(p))
        if n <= 0:
            return np.zeros(0, KP_DTYPE)
        buf = (C.c_uint8 * (n * 28)).from_address(p.value)
        return np.frombuffer(buf, KP_DTYPE).copy()

    def candidates(self, level):
        return self._kps(self.L.orb_oracle_get_candidates, level)

    def level_keypoints(self, level):
        return self._kps(self.L.orb_oracle_get_level_keypoints, level)


_NATIVE = None
NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off -fno-fast-math"
PORTABLE_FLAGS = "-O2 -ffp-contract=off -fno-fast-math"


def lib_native():
    """The same source built with BASELINE.md section 3's flags (-O3 -march=native -ffp-contract=off) ON THIS HOST,
    for the timed CPU baseline only (a -march=native binary must not travel between machines, so it is rebuilt
    whenever it is asked for in a new process).  Returns None when the build fails."""
    global _NATIVE
    if _NATIVE is None:
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
            L = C.CDLL(os.path.join(_HERE, "liborb_oracle_native.so"))
            L.orb_oracle_extract_many.restype = C.c_long
            L.orb_oracle_extract_many.argtypes = lib().orb_oracle_extract_many.argtypes
            _NATIVE = L
        except Exception:
            _NATIVE = False
    return _NATIVE or None


def _native_for_extractor():
    """The -march=native build with the Extractor's entry points typed (timing-only uses: tests/test_oracle_fastpath.py, bench.py)."""
    L = lib_native()
    if L is None:
        return None
    if not getattr(L, "_typed", False):
        b = lib()
        for name in ("orb_oracle_create", "orb_oracle_destroy", "orb_oracle_extract", "orb_oracle_set_trig_mode", "orb_oracle_set_gauss_taps",
                     "orb_oracle_set_atan_fma", "orb_oracle_get_scale_tables", "orb_oracle_get_features_per_level", "orb_oracle_get_umax",
                     "orb_oracle_get_level", "orb_oracle_get_blurred", "orb_oracle_get_candidates", "orb_oracle_get_level_keypoints"):
            f, g = getattr(L, name), getattr(b, name)
            f.restype = g.restype
            if g.argtypes is not None:
                f.argtypes = g.argtypes
        L._typed = True
    return L


def extract_many(imgs, nthreads, reps, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, lap=(0, 0),
                 native=False):
    """Threaded CPU baseline: returns (total keypoints, seconds)."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    assert imgs.ndim == 3
    sec = C.c_double(0)
    L = (lib_native() if native else None) or lib()
    n = L.orb_oracle_extract_many(nthreads, reps, nfeatures, scale, nlevels, ini_th, min_th, _p(imgs),
                                  imgs.shape[0], imgs.shape[1], imgs.shape[2], lap[0], lap[1], C.byref(sec))
    return int(n), sec.value


STAGES = ("pyramid", "fast", "quadtree", "orientation", "blur", "descriptors")


def extract_many_stages(imgs, nthreads, reps, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, lap=(0, 0), native=False,
                        fastpath=False):
    """extract_many with the timing-only fast path selectable and the wall seconds per stage (summed over the threads).
    Returns (total keypoints, seconds, {stage: seconds})."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    assert imgs.ndim == 3
    sec = C.c_double(0)
    st = (C.c_double * 6)()
    L = (lib_native() if native else None) or lib()
    L.orb_oracle_extract_many2.restype = C.c_long
    L.orb_oracle_extract_many2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = L.orb_oracle_extract_many2(nthreads, reps, nfeatures, scale, nlevels, ini_th, min_th, _p(imgs), imgs.shape[0], imgs.shape[1],
                                   imgs.shape[2], lap[0], lap[1], int(bool(fastpath)), C.cast(C.byref(sec), C.c_void_p),
                                   C.cast(st, C.c_void_p))
    return int(n), sec.value, dict(zip(STAGES, list(st)))


def resize_linear(src, dh, dw):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orb_oracle_resize_linear(_p(src), src.shape[0], src.shape[1], src.strides[0], _p(dst), dh, dw, dst.strides[0])
    return dst


def border_reflect101(interior, border):
    h, w = interior.shape
    buf = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    buf[border:border + h, border:border + w] = interior
    lib().orb_oracle_border_reflect101(_p(buf), buf.shape[0], buf.shape[1], buf.strides[0], border)
    return buf


def fast(img, threshold, nms=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().orb_oracle_fast(_p(img), img.shape[0], img.shape[1], img.strides[0], threshold, int(nms), _p(out), cap)
    return out[:n].copy()


def fast_score_closed(patch7):
    p = np.ascontiguousarray(patch7, np.uint8)
    assert p.shape == (7, 7)
    return lib().orb_oracle_fast_score_closed(C.c_void_p(p.ctypes.data + 3 * 7 + 3), 7)


def fast_score_2loop(patch7, threshold):
    p = np.ascontiguousarray(patch7, np.uint8)
    assert p.shape == (7, 7)
    return lib().orb_oracle_fast_score_2loop(C.c_void_p(p.ctypes.data + 3 * 7 + 3), 7, threshold)


def gaussian_blur7(img, taps=None):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    t = None if taps is None else np.ascontiguousarray(taps, np.int32)
    lib().orb_oracle_gaussian_blur7(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(dst), dst.strides[0],
                                    None if t is None else _p(t))
    return dst


def fast_atan2_fma(y, x):
    return float(lib().orb_oracle_fast_atan2_fma(float(y), float(x)))


def fast_atan2(y, x):
    return float(lib().orb_oracle_fast_atan2(float(y), float(x)))


def sincos_cr(angle):
    s, c = C.c_float(), C.c_float()
    lib().orb_oracle_sincos_cr(float(angle), C.byref(s), C.byref(c))
    return s.value, c.value


def distribute_octree(cands, minX, maxX, minY, maxY, N):
    cands = np.ascontiguousarray(cands, KP_DTYPE)
    cap = max(len(cands), 1)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orb_oracle_distribute_octree(_p(cands), len(cands), minX, maxX, minY, maxY, N, _p(out), cap)
    return out[:n].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orb_oracle_descriptor_distance(_p(a), _p(b))


def hamming_matrix(A, B):
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
    B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    D = np.zeros((len(A), len(B)), np.uint16)
    lib().orb_oracle_hamming_matrix(_p(A), len(A), _p(B), len(B), _p(D))
    return D


def bfknn2(Q, T):
    Q = np.ascontiguousarray(Q, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(Q), 2), np.int32)
    dist = np.zeros((len(Q), 2), np.int32)
    lib().orb_oracle_bfknn2(_p(Q), len(Q), _p(T), len(T), _p(idx), _p(dist))
    return idx, dist


def three_maxima(counts):
    c = np.ascontiguousarray(counts, np.int32)
    a, b, d = C.c_int(), C.c_int(), C.c_int()
    lib().orb_oracle_three_maxima(_p(c), len(c), C.byref(a), C.byref(b), C.byref(d))
    return a.value, b.value, d.value


def search_bow_kf_f(descKF, maskKF, angKF, fvKF, descF, angF, fvF, Nleft=-1, nnratio=0.7, check_ori=True):
    descKF = np.ascontiguousarray(descKF, np.uint8).reshape(-1, 32)
    descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
    maskKF = np.ascontiguousarray(maskKF, np.uint8)
    angKF = np.ascontiguousarray(angKF, np.float32)
    angF = np.ascontiguousarray(angF, np.float32)
    a, b = _fv(fvKF), _fv(fvF)
    match = np.zeros(len(descF), np.int32)
    n = lib().orb_oracle_search_bow_kf_f(_p(descKF), len(descKF), _p(maskKF), _p(angKF), C.byref(a), _p(descF),
                                         len(descF), _p(angF), C.byref(b), Nleft, nnratio, int(check_ori), _p(match))
    return n, match


def search_bow_kf_kf(desc1, mask1, ang1, fv1, desc2, mask2, ang2, fv2, lim1=-1, lim2=-1, nnratio=0.9,
                     check_ori=True):
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    mask1 = np.ascontiguousarray(mask1, np.uint8)
    mask2 = np.ascontiguousarray(mask2, np.uint8)
    ang1 = np.ascontiguousarray(ang1, np.float32)
    ang2 = np.ascontiguousarray(ang2, np.float32)
    a, b = _fv(fv1), _fv(fv2)
    match = np.zeros(len(desc1), np.int32)
    n = lib().orb_oracle_search_bow_kf_kf(_p(desc1), len(desc1), _p(mask1), _p(ang1), C.byref(a), lim1, _p(desc2),
                                          len(desc2), _p(mask2), _p(ang2), C.byref(b), lim2, nnratio, int(check_ori),
                                          _p(match))
    return n, match


def search_triangulation(desc1, hasMP1, kp1xy, ang1, oct1, uR1, fv1, desc2, hasMP2, kp2xy, ang2, oct2, uR2, fv2, F12,
                         ep, scaleFactors2, levelSigma2_2, only_stereo=False, coarse=False, check_ori=True):
    def prep(desc, has, xy, ang, oc, ur):
        return (np.ascontiguousarray(desc, np.uint8).reshape(-1, 32), np.ascontiguousarray(has, np.uint8),
                np.ascontiguousarray(xy, np.float32).reshape(-1, 2), np.ascontiguousarray(ang, np.float32),
                np.ascontiguousarray(oc, np.int32), np.ascontiguousarray(ur, np.float32))

    d1, h1, x1, a1, o1, u1 = prep(desc1, hasMP1, kp1xy, ang1, oct1, uR1)
    d2, h2, x2, a2, o2, u2 = prep(desc2, hasMP2, kp2xy, ang2, oct2, uR2)
    f1, f2 = _fv(fv1), _fv(fv2)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf2 = np.ascontiguousarray(scaleFactors2, np.float32)
    ls2 = np.ascontiguousarray(levelSigma2_2, np.float32)
    pairs = np.zeros((max(len(d1), 1), 2), np.int32)
    n = lib().orb_oracle_search_triangulation(
        _p(d1), len(d1), _p(h1), _p(x1), _p(a1), _p(o1), _p(u1), C.byref(f1),
        _p(d2), len(d2), _p(h2), _p(x2), _p(a2), _p(o2), _p(u2), C.byref(f2),
        _p(F), float(ep[0]), float(ep[1]), _p(sf2), _p(ls2), int(only_stereo), int(coarse), int(check_ori), _p(pairs))
    return pairs[:n].copy()


def compute_stereo_matches(exL, exR, kpsL, descL, kpsR, descR, mb, mbf):
    """Frame::ComputeStereoMatches; exL/exR = oracle Extractors that just processed the two images."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    dL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    dR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    uR = np.zeros(len(kpsL), np.float32)
    dep = np.zeros(len(kpsL), np.float32)
    n = lib().orb_oracle_compute_stereo_matches(exL.h, exR.h, _p(kpsL), _p(dL), len(kpsL), _p(kpsR), _p(dR),
                                                len(kpsR), mb, mbf, _p(uR), _p(dep))
    return n, uR, dep


class _TriKb8Args(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("hasMP1", C.c_void_p), ("kp1xy", C.c_void_p), ("ang1", C.c_void_p),
                ("oct1", C.c_void_p), ("uRight1", C.c_void_p), ("fv1", C.POINTER(_FV)), ("Nleft1", C.c_int),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("hasMP2", C.c_void_p), ("kp2xy", C.c_void_p), ("ang2", C.c_void_p),
                ("oct2", C.c_void_p), ("uRight2", C.c_void_p), ("fv2", C.POINTER(_FV)), ("Nleft2", C.c_int),
                ("kb8_1L", C.c_void_p), ("kb8_1R", C.c_void_p), ("kb8_2L", C.c_void_p), ("kb8_2R", C.c_void_p),
                ("R12", C.c_void_p), ("t12", C.c_void_p), ("ep", C.c_float * 2),
                ("scaleFactors2", C.c_void_p), ("levelSigma2_1", C.c_void_p), ("levelSigma2_2", C.c_void_p),
                ("only_stereo", C.c_int), ("coarse", C.c_int), ("check_orientation", C.c_int)]


class _Tri3dArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("hasMP1", C.c_void_p), ("kp1xy", C.c_void_p), ("ang1", C.c_void_p),
                ("oct1", C.c_void_p), ("fv1", C.POINTER(_FV)), ("Nleft1", C.c_int),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("hasMP2", C.c_void_p), ("kp2xy", C.c_void_p), ("ang2", C.c_void_p),
                ("oct2", C.c_void_p), ("fv2", C.POINTER(_FV)), ("Nleft2", C.c_int),
                ("kb8_1L", C.c_void_p), ("kb8_1R", C.c_void_p), ("kb8_2L", C.c_void_p), ("kb8_2R", C.c_void_p),
                ("Tcw1L", C.c_void_p), ("Tcw1R", C.c_void_p), ("Tcw2L", C.c_void_p), ("Tcw2R", C.c_void_p),
                ("levelSigma2_1", C.c_void_p), ("levelSigma2_2", C.c_void_p), ("check_orientation", C.c_int)]


class _ProjArgs(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("n", C.c_int), ("kx", C.c_void_p), ("ky", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("uright", C.c_void_p), ("taken", C.c_void_p), ("Nleft", C.c_int),
                ("left_to_right", C.c_void_p), ("right_to_left", C.c_void_p),
                ("minX", C.c_float), ("minY", C.c_float), ("gridWInv", C.c_float), ("gridHInv", C.c_float),
                ("nq", C.c_int), ("qdesc", C.c_void_p), ("qx", C.c_void_p), ("qy", C.c_void_p), ("qr", C.c_void_p),
                ("qmin_level", C.c_void_p), ("qmax_level", C.c_void_p), ("qxr", C.c_void_p), ("qflags", C.c_void_p),
                ("qangle", C.c_void_p), ("qblocks", C.c_void_p),
                ("mode", C.c_int), ("nnratio", C.c_float), ("th_high", C.c_int), ("check_orientation", C.c_int),
                ("inv_level_sigma2", C.c_void_p), ("n_levels", C.c_int), ("chi2_gate", C.c_int)]


_PROJ_ARRAYS = [("desc", np.uint8), ("kx", np.float32), ("ky", np.float32), ("octave", np.int32), ("angle", np.float32),
                ("uright", np.float32), ("taken", np.uint8), ("left_to_right", np.int32), ("right_to_left", np.int32),
                ("qdesc", np.uint8), ("qx", np.float32), ("qy", np.float32), ("qr", np.float32),
                ("qmin_level", np.int32), ("qmax_level", np.int32), ("qxr", np.float32), ("qflags", np.uint8),
                ("qangle", np.float32), ("qblocks", np.uint8), ("inv_level_sigma2", np.float32)]


def _proj_args(pr):
    """dict with the fields of the projection-search argument struct -> (struct, keep-alive arrays, n, nq)."""
    keep = {}
    a = _ProjArgs()
    for name, dt in _PROJ_ARRAYS:
        v = pr.get(name)
        if v is None:
            setattr(a, name, None)
        else:
            keep[name] = np.ascontiguousarray(v, dt)
            setattr(a, name, keep[name].ctypes.data)
    a.n = len(keep["kx"])
    a.nq = len(keep["qx"])
    a.Nleft = int(pr.get("Nleft", -1))
    for name in ("minX", "minY", "gridWInv", "gridHInv", "nnratio"):
        setattr(a, name, float(pr[name]))
    a.mode = int(pr["mode"])
    a.th_high = int(pr.get("th_high", 100))
    a.check_orientation = int(pr.get("check_orientation", 0))
    a.chi2_gate = int(pr.get("chi2_gate", 0))
    a.n_levels = len(keep["inv_level_sigma2"]) if "inv_level_sigma2" in keep else 0
    return a, keep, a.n, a.nq


class _InitArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("octave1", C.c_void_p), ("angle1", C.c_void_p),
                ("prev_xy", C.c_void_p),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("kx2", C.c_void_p), ("ky2", C.c_void_p), ("octave2", C.c_void_p),
                ("angle2", C.c_void_p),
                ("minX", C.c_float), ("minY", C.c_float), ("gridWInv", C.c_float), ("gridHInv", C.c_float),
                ("window_size", C.c_int), ("nnratio", C.c_float), ("check_orientation", C.c_int)]


def search_initialization(pr):
    keep = [np.ascontiguousarray(pr[k], dt) for k, dt in (("desc1", np.uint8), ("octave1", np.int32), ("angle1", np.float32),
                                                           ("prev_xy", np.float32), ("desc2", np.uint8), ("kx2", np.float32),
                                                           ("ky2", np.float32), ("octave2", np.int32), ("angle2", np.float32))]
    d1, o1, a1, pv, d2, kx, ky, o2, a2 = keep
    a = _InitArgs(d1.ctypes.data, len(o1), o1.ctypes.data, a1.ctypes.data, pv.ctypes.data, d2.ctypes.data, len(kx),
                  kx.ctypes.data, ky.ctypes.data, o2.ctypes.data, a2.ctypes.data, float(pr["minX"]), float(pr["minY"]),
                  float(pr["gridWInv"]), float(pr["gridHInv"]), int(pr["window_size"]), float(pr["nnratio"]),
                  int(pr.get("check_orientation", 1)))
    m = np.full(max(len(o1), 1), -1, np.int32)
    L = lib()
    L.orb_oracle_search_initialization.restype = C.c_int
    L.orb_oracle_search_initialization.argtypes = [C.c_void_p, C.c_void_p]
    r = L.orb_oracle_search_initialization(C.byref(a), _p(m))
    return r, m[:len(o1)]


def search_triangulation_kb8(I, only_stereo=False, coarse=False, check_ori=True):
    keep = []

    def arr(v, dt):
        if v is None:
            return None
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data

    f1 = _fv(I["fv1"])
    f2 = _fv(I["fv2"])
    a = _TriKb8Args(arr(I["d1"], np.uint8), len(I["d1"]), arr(I["has1"], np.uint8), arr(I["kp1"], np.float32),
                    arr(I["a1"], np.float32), arr(I["oct1"], np.int32), arr(I.get("u1"), np.float32), C.pointer(f1),
                    int(I["Nleft1"]),
                    arr(I["d2"], np.uint8), len(I["d2"]), arr(I["has2"], np.uint8), arr(I["kp2"], np.float32),
                    arr(I["a2"], np.float32), arr(I["oct2"], np.int32), arr(I.get("u2"), np.float32), C.pointer(f2),
                    int(I["Nleft2"]),
                    arr(I["P1L"], np.float32), arr(I.get("P1R"), np.float32), arr(I["P2L"], np.float32),
                    arr(I.get("P2R"), np.float32), arr(I["R12"], np.float32), arr(I["t12"], np.float32),
                    (C.c_float * 2)(float(I["ep"][0]), float(I["ep"][1])), arr(I["sf"], np.float32),
                    arr(I["sig1"], np.float32), arr(I["sig2"], np.float32), int(only_stereo), int(coarse), int(check_ori))
    pairs = np.zeros((max(len(I["d1"]), 1), 2), np.int32)
    L = lib()
    L.orb_oracle_search_triangulation_kb8.restype = C.c_int
    L.orb_oracle_search_triangulation_kb8.argtypes = [C.c_void_p, C.c_void_p]
    n = L.orb_oracle_search_triangulation_kb8(C.byref(a), _p(pairs))
    return pairs[:n].copy()


def search_triangulation_3d(I, check_ori=True):
    """ORBmatcher::SearchForTriangulation(..., vMatchedPoints) src/ORBmatcher.cc:1452-1641: (pairs[n,2], points[n,3])."""
    keep = []

    def arr(v, dt):
        if v is None:
            return None
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data

    f1 = _fv(I["fv1"])
    f2 = _fv(I["fv2"])
    T = I["Tcw"]
    a = _Tri3dArgs(arr(I["d1"], np.uint8), len(I["d1"]), arr(I["has1"], np.uint8), arr(I["kp1"], np.float32),
                   arr(I["a1"], np.float32), arr(I["oct1"], np.int32), C.pointer(f1), int(I["Nleft1"]),
                   arr(I["d2"], np.uint8), len(I["d2"]), arr(I["has2"], np.uint8), arr(I["kp2"], np.float32),
                   arr(I["a2"], np.float32), arr(I["oct2"], np.int32), C.pointer(f2), int(I["Nleft2"]),
                   arr(I.get("P1L"), np.float32), arr(I.get("P1R"), np.float32), arr(I.get("P2L"), np.float32),
                   arr(I.get("P2R"), np.float32), arr(T[0], np.float32), arr(T[1], np.float32), arr(T[2], np.float32),
                   arr(T[3], np.float32), arr(I["sig1"], np.float32), arr(I["sig2"], np.float32), int(check_ori))
    n1 = max(len(I["d1"]), 1)
    pairs = np.zeros((n1, 2), np.int32)
    points = np.zeros((n1, 3), np.float32)
    L = lib()
    L.orb_oracle_search_triangulation_3d.restype = C.c_int
    L.orb_oracle_search_triangulation_3d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orb_oracle_search_triangulation_3d(C.byref(a), _p(pairs), _p(points))
    return pairs[:n].copy(), points[:n].copy()


def kb8_match_and_triangulate(P1, P2, kp1, kp2, Tcw1, Tcw2, sigma1, sigma2):
    """KannalaBrandt8::matchAndtriangulate per pair: (ok[n] bool, x3D[n,3])."""
    L = lib()
    L.orb_oracle_kb8_match_and_triangulate.restype = C.c_int
    L.orb_oracle_kb8_match_and_triangulate.argtypes = [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_void_p]
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    A = [np.ascontiguousarray(v, np.float32) for v in (P1, P2, Tcw1, Tcw2)]
    ok = np.zeros(len(kp1), bool)
    X = np.zeros((len(kp1), 3), np.float32)
    for i in range(len(kp1)):
        ok[i] = L.orb_oracle_kb8_match_and_triangulate(_p(A[0]), _p(A[1]), kp1[i].ctypes.data, kp2[i].ctypes.data, _p(A[2]),
                                                       _p(A[3]), float(sigma1[i]), float(sigma2[i]), X[i].ctypes.data) != 0
    return ok, X


def kb8_triangulate(P1, P2, kp1, kp2, R12, t12, sigma1, sigma2):
    """KannalaBrandt8::TriangulateMatches_ per pair: (z1 or -1, p3D[n,3])."""
    L = lib()
    L.orb_oracle_kb8_triangulate_matches.restype = C.c_float
    L.orb_oracle_kb8_triangulate_matches.argtypes = [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_void_p]
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    A = [np.ascontiguousarray(v, np.float32) for v in (P1, P2, R12, t12)]
    z = np.zeros(len(kp1), np.float32)
    X = np.zeros((len(kp1), 3), np.float32)
    for i in range(len(kp1)):
        z[i] = L.orb_oracle_kb8_triangulate_matches(_p(A[0]), _p(A[1]), kp1[i].ctypes.data, kp2[i].ctypes.data, _p(A[2]),
                                                    _p(A[3]), float(sigma1[i]), float(sigma2[i]), X[i].ctypes.data)
    return z, X


def stereo_fisheye_matches(descL, kpL, octL, descR, kpR, octR, P1, P2, Rlr, tlr, level_sigma2):
    dL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    dR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    kL = np.ascontiguousarray(kpL, np.float32).reshape(-1, 2)
    kR = np.ascontiguousarray(kpR, np.float32).reshape(-1, 2)
    oL = np.ascontiguousarray(octL, np.int32)
    oR = np.ascontiguousarray(octR, np.int32)
    A = [np.ascontiguousarray(v, np.float32) for v in (P1, P2, Rlr, tlr, level_sigma2)]
    nL, nR = len(dL), len(dR)
    l2r = np.zeros(max(nL, 1), np.int32)
    r2l = np.zeros(max(nR, 1), np.int32)
    dep = np.zeros(max(nL, 1), np.float32)
    X = np.zeros((max(nL, 1), 3), np.float32)
    L = lib()
    L.orb_oracle_stereo_fisheye_matches.restype = C.c_int
    L.orb_oracle_stereo_fisheye_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] * 2 + [C.c_void_p] * 9
    n = L.orb_oracle_stereo_fisheye_matches(_p(dL), _p(kL), _p(oL), nL, _p(dR), _p(kR), _p(oR), nR, _p(A[0]), _p(A[1]), _p(A[2]),
                                            _p(A[3]), _p(A[4]), _p(l2r), _p(r2l), _p(dep), _p(X))
    return n, l2r[:nL], r2l[:nR], dep[:nL], X[:nL]


def search_projection(problem):
    a, keep, n, nq = _proj_args(problem)
    qm = np.full(max(nq, 1), -1, np.int32)
    fm = np.full(max(n, 1), -1, np.int32)
    r = lib().orb_oracle_search_projection(C.byref(a), _p(qm), _p(fm))
    return r, qm[:nq], fm[:n]


def distinctive_descriptors(pool, offsets):
    pool = np.ascontiguousarray(pool, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    best = np.zeros(len(offsets) - 1, np.int32)
    lib().orb_oracle_distinctive_descriptors(_p(pool), _p(offsets), len(best), _p(best))
    return best


def vocab_transform(vocab, feats, levelsup=4):
    """vocab = dict(desc[nn,32] u8, child_off[nn+1] i32, child_ids i32, word[nn] i32, weight[nn] f64, L)."""
    feats = np.ascontiguousarray(feats, np.uint8).reshape(-1, 32)
    n = len(feats)
    w = np.zeros(n, np.int32)
    nid = np.zeros(n, np.int32)
    wt = np.zeros(n, np.float64)
    lib().orb_oracle_vocab_transform(len(vocab["word"]), _p(vocab["desc"]), _p(vocab["child_off"]),
                                     _p(vocab["child_ids"]), _p(vocab["word"]), _p(vocab["weight"]), int(vocab["L"]),
                                     _p(feats), n, levelsup, _p(w), _p(nid), _p(wt))
    return w, nid, wt


def compute_bow(vocab, feats, levelsup=4, weighting=0, scoring=0):
    """TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup): returns
    ((word_ids u32, values f64), (node_ids u32, offsets i32, indices i32))."""
    feats = np.ascontiguousarray(feats, np.uint8).reshape(-1, 32)
    n = len(feats)
    L = lib()
    L.orb_oracle_compute_bow.restype = C.c_int
    L.orb_oracle_compute_bow.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    ids = np.zeros(max(n, 1), np.uint32)
    vals = np.zeros(max(n, 1), np.float64)
    node_ids = np.zeros(max(n, 1), np.uint32)
    offsets = np.zeros(n + 1, np.int32)
    indices = np.zeros(max(n, 1), np.int32)
    nn = C.c_int(0)
    nw = L.orb_oracle_compute_bow(len(vocab["word"]), _p(vocab["desc"]), _p(vocab["child_off"]), _p(vocab["child_ids"]),
                                  _p(vocab["word"]), _p(vocab["weight"]), int(vocab["L"]), _p(feats), n, levelsup, weighting, scoring,
                                  _p(ids), _p(vals), _p(node_ids), _p(offsets), _p(indices), C.cast(C.byref(nn), C.c_void_p))
    k = nn.value
    return (ids[:nw].copy(), vals[:nw].copy()), (node_ids[:k].copy(), offsets[:k + 1].copy(), indices[:offsets[k]].copy())


def kb8_unproject(params8, uv):
    P = np.ascontiguousarray(params8, np.float32)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    rays = np.zeros((len(uv), 3), np.float32)
    lib().orb_oracle_kb8_unproject(_p(P), _p(uv), len(uv), _p(rays))
    return rays
