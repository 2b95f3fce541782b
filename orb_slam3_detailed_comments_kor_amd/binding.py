"""ctypes binding of liborbfe.so (include/orbfe.h).

`ORBextractor` mirrors the reference class (include/ORBextractor.h:43-107): same
constructor arguments, `__call__(image, lapping_area)` = operator(), scale getters and
the image pyramid.  Everything goes through the C ABI; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])

ERR_ARGS, ERR_NODEV, ERR_STATE, ERR_IMAGE_SMALL, ERR_IMAGE_LARGE, ERR_NFEATURES, MC_ERR_RCCL = -2, -3, -4, -5, -6, -7, -8
TRIG_LIBM, TRIG_CR, TRIG_LIBM_HOSTCHECK = 0, 1, 2
LANES_BATCH, MAX_LANES = 0, 4  # ORBFE_LANES_BATCH / ORBFE_MAX_LANES (include/orbfe.h)
STAGES = ("pyramid", "fast", "octree", "pack", "desc", "trigfix")


class OrbfeError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


class _FV(C.Structure):
    _fields_ = [("nn", C.c_int), ("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p)]


class _BowArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("mask1", C.c_void_p), ("angle1", C.c_void_p),
                ("fv1", _FV), ("limit1", C.c_int),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("mask2", C.c_void_p), ("angle2", C.c_void_p),
                ("fv2", _FV), ("limit2", C.c_int),
                ("Nleft", C.c_int), ("nnratio", C.c_float), ("check_orientation", C.c_int), ("variant", C.c_int)]


class _Vocab(C.Structure):
    _fields_ = [("nnodes", C.c_int), ("node_desc", C.c_void_p), ("child_off", C.c_void_p), ("child_ids", C.c_void_p),
                ("node_word", C.c_void_p), ("node_weight", C.c_void_p), ("L", C.c_int)]


class _TriArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("hasMP1", C.c_void_p), ("kp1_xy", C.c_void_p),
                ("angle1", C.c_void_p), ("octave1", C.c_void_p), ("uRight1", C.c_void_p), ("fv1", _FV),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("hasMP2", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("angle2", C.c_void_p), ("octave2", C.c_void_p), ("uRight2", C.c_void_p), ("fv2", _FV),
                ("F12", C.c_float * 9), ("ep", C.c_float * 2),
                ("scaleFactors2", C.c_void_p), ("levelSigma2_2", C.c_void_p), ("nlevels2", C.c_int),
                ("only_stereo", C.c_int), ("coarse", C.c_int), ("check_orientation", C.c_int)]


class _TriKb8Args(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("hasMP1", C.c_void_p), ("kp1_xy", C.c_void_p),
                ("angle1", C.c_void_p), ("octave1", C.c_void_p), ("uRight1", C.c_void_p), ("fv1", _FV), ("Nleft1", C.c_int),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("hasMP2", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("angle2", C.c_void_p), ("octave2", C.c_void_p), ("uRight2", C.c_void_p), ("fv2", _FV), ("Nleft2", C.c_int),
                ("kb8_1L", C.c_void_p), ("kb8_1R", C.c_void_p), ("kb8_2L", C.c_void_p), ("kb8_2R", C.c_void_p),
                ("R12", C.c_void_p), ("t12", C.c_void_p), ("ep", C.c_float * 2),
                ("scaleFactors2", C.c_void_p), ("levelSigma2_1", C.c_void_p), ("levelSigma2_2", C.c_void_p),
                ("nlevels1", C.c_int), ("nlevels2", C.c_int),
                ("only_stereo", C.c_int), ("coarse", C.c_int), ("check_orientation", C.c_int)]


class _Tri3dArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("hasMP1", C.c_void_p), ("kp1_xy", C.c_void_p),
                ("angle1", C.c_void_p), ("octave1", C.c_void_p), ("fv1", _FV), ("Nleft1", C.c_int),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("hasMP2", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("angle2", C.c_void_p), ("octave2", C.c_void_p), ("fv2", _FV), ("Nleft2", C.c_int),
                ("kb8_1L", C.c_void_p), ("kb8_1R", C.c_void_p), ("kb8_2L", C.c_void_p), ("kb8_2R", C.c_void_p),
                ("Tcw1L", C.c_void_p), ("Tcw1R", C.c_void_p), ("Tcw2L", C.c_void_p), ("Tcw2R", C.c_void_p),
                ("levelSigma2_1", C.c_void_p), ("levelSigma2_2", C.c_void_p), ("nlevels1", C.c_int), ("nlevels2", C.c_int),
                ("check_orientation", C.c_int)]


class _InitArgs(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int), ("octave1", C.c_void_p), ("angle1", C.c_void_p),
                ("prev_xy", C.c_void_p),
                ("desc2", C.c_void_p), ("n2", C.c_int), ("kx2", C.c_void_p), ("ky2", C.c_void_p), ("octave2", C.c_void_p),
                ("angle2", C.c_void_p),
                ("minX", C.c_float), ("minY", C.c_float), ("gridWInv", C.c_float), ("gridHInv", C.c_float),
                ("window_size", C.c_int), ("nnratio", C.c_float), ("check_orientation", C.c_int)]


def _init_args(pr):
    keep = [np.ascontiguousarray(pr[k], dt) for k, dt in (("desc1", np.uint8), ("octave1", np.int32), ("angle1", np.float32),
                                                           ("prev_xy", np.float32), ("desc2", np.uint8), ("kx2", np.float32),
                                                           ("ky2", np.float32), ("octave2", np.int32), ("angle2", np.float32))]
    d1, o1, a1, pv, d2, kx, ky, o2, a2 = keep
    a = _InitArgs(d1.ctypes.data, len(o1), o1.ctypes.data, a1.ctypes.data, pv.ctypes.data, d2.ctypes.data, len(kx),
                  kx.ctypes.data, ky.ctypes.data, o2.ctypes.data, a2.ctypes.data, float(pr["minX"]), float(pr["minY"]),
                  float(pr["gridWInv"]), float(pr["gridHInv"]), int(pr["window_size"]), float(pr["nnratio"]),
                  int(pr.get("check_orientation", 1)))
    return a, keep, len(o1)


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


class _McLayout(C.Structure):
    _fields_ = [("desc_bytes", C.c_size_t), ("count_off", C.c_size_t), ("slab_bytes", C.c_size_t)]


class _McView(C.Structure):
    _fields_ = [("gathered", C.c_void_p), ("slab", C.c_void_p), ("d_kps", C.c_void_p), ("d_mono", C.c_void_p),
                ("slab_bytes", C.c_size_t), ("batch", C.c_long)]


MC_RCCL, MC_HOST, MC_ID_BYTES = 0, 1, 128
MC_MAX_IN_FLIGHT = 3  # ORBFE_MC_MAX_IN_FLIGHT (include/orbfe_mc.h): batches a handle accepts before a wait is due


def _share_hip_runtime_with_torch():
    """One HIP runtime per process: torch bundles its own libamdhip64.so (same SONAME as the system
    one).  Loading it first makes the dynamic linker bind liborbfe.so to that copy, so device
    pointers, streams and events are interchangeable with torch's (two runtimes in one process
    cannot both own the GPU)."""
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; without it the system runtime is used
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib_path():
    # ORBFE_LIB: another build of the library (kernel variants side by side on one GPU box: tools/ab_build.sh)
    return os.environ.get("ORBFE_LIB") or os.path.join(_HERE, "liborbfe.so")


def lib():
    """Load liborbfe.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("liborbfe.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C orb_slam3_detailed_comments_kor_amd/csrc)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(path)
        L.orbfe_version.restype = C.c_char_p
        L.orbfe_error_string.restype = C.c_char_p
        L.orbfe_error_string.argtypes = [C.c_int]
        L.orbfe_create.restype = C.c_int
        L.orbfe_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orbfe_destroy.argtypes = [C.c_void_p]
        L.orbfe_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        # multi-GPU path (include/orbfe_mc.h)
        L.orbfe_mc_layout.argtypes = [C.c_int, C.c_int, C.POINTER(_McLayout)]
        L.orbfe_mc_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orbfe_mc_ring_pairs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_mc_job_offsets.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_mc_unique_id.argtypes = [C.c_int, C.c_void_p]
        L.orbfe_mc_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orbfe_mc_destroy.restype = None
        L.orbfe_mc_destroy.argtypes = [C.c_void_p]
        L.orbfe_mc_extract_exchange_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                       C.c_int, C.c_int]
        L.orbfe_mc_extract_exchange_wait.argtypes = [C.c_void_p, C.POINTER(_McView)]
        L.orbfe_mc_match_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_mc_match_outputs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.orbfe_mc_match_ring_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long]
        L.orbfe_mc_exchange_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.orbfe_set_gaussian_taps.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_keyframe_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        L.orbfe_keyframe_set_mask.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_keyframe_destroy.argtypes = [C.c_void_p]
        L.orbfe_keyframe_destroy.restype = None
        L.orbfe_search_bow_keyframes.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbfe_search_tri_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbfe_set_lanes.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_lanes_join.argtypes = [C.c_void_p]
        L.orbfe_set_lane_mode.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_set_lane_input_guard.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_lanes_record.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_set_trig_mode.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_max_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orbfe_set_kb8.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_set_ray_output.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_get_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbfe_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orbfe_extract_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_size_t,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_extract_batch_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_set_atan_fma.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_debug_blurred_patch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orbfe_vocab_load_text.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int)]
        L.orbfe_extract_batch_submit.argtypes = L.orbfe_extract_batch.argtypes
        L.orbfe_extract_batch_wait.argtypes = [C.c_void_p]
        L.orbfe_host_alloc.restype = C.c_void_p
        L.orbfe_host_alloc.argtypes = [C.c_size_t]
        L.orbfe_host_free.restype = None
        L.orbfe_host_free.argtypes = [C.c_void_p]
        L.orbfe_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.orbfe_set_auto_register.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_host_unregister.argtypes = [C.c_void_p]
        L.orbfe_extract_stereo_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                                C.c_void_p, C.c_void_p]
        L.orbfe_extract_stereo_pair.restype = C.c_int
        L.orbfe_extract_stereo_pair_submit.argtypes = L.orbfe_extract_stereo_pair.argtypes
        L.orbfe_extract_stereo_pair_wait.argtypes = [C.c_void_p]
        L.orbfe_compute_stereo_matches_resident.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                                            C.c_void_p, C.c_void_p, C.c_int]
        L.orbfe_hamming_pairs_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_bfknn2_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]
        L.orbfe_bfknn2_frames_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_matcher_sync.argtypes = [C.c_int]
        L.orbfe_get_device_outputs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orbfe_extract_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t,
                                                 C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p]
        L.orbfe_sync.argtypes = [C.c_void_p]
        L.orbfe_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orbfe_get_levels.argtypes = [C.c_void_p]
        L.orbfe_get_scale_factor.restype = C.c_float
        L.orbfe_get_scale_factor.argtypes = [C.c_void_p]
        L.orbfe_get_scale_tables.restype = None
        L.orbfe_get_scale_tables.argtypes = [C.c_void_p] * 5
        L.orbfe_get_features_per_level.restype = None
        L.orbfe_get_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
        L.orbfe_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.orbfe_profile_read.argtypes = [C.c_void_p, C.c_void_p]
        L.orbfe_debug_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbfe_debug_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbfe_debug_fixups.argtypes = [C.c_void_p]
        L.orbfe_debug_trig.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_hamming_pairs.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_bfknn2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbfe_search_bow.argtypes = [C.c_int, C.POINTER(_BowArgs), C.c_void_p]
        L.orbfe_search_tri.argtypes = [C.c_int, C.POINTER(_TriArgs), C.c_void_p]
        L.orbfe_search_bow_batch.argtypes = [C.c_int, C.c_int, C.POINTER(_BowArgs), C.POINTER(C.c_void_p), C.c_void_p]
        L.orbfe_kb8_unproject.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_matcher_last_kernel_ms.restype = C.c_float
        L.orbfe_search_projection.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbfe_search_projection_last_sweeps.argtypes = []
        L.orbfe_search_projection_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                    C.c_void_p]
        L.orbfe_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orbfe_vocab_upload.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(_Vocab)]
        L.orbfe_vocab_free.argtypes = [C.c_void_p]
        L.orbfe_vocab_set_types.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orbfe_vocab_get_types.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbfe_vocab_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


EXPORTS = ["orbfe_error_string", "orbfe_set_auto_register", "orbfe_version", "orbfe_create", "orbfe_destroy", "orbfe_set_stream", "orbfe_get_stream", "orbfe_set_gaussian_taps",
           "orbfe_mc_layout", "orbfe_mc_shard", "orbfe_mc_ring_pairs", "orbfe_mc_job_offsets", "orbfe_mc_unique_id",
           "orbfe_mc_create", "orbfe_mc_destroy", "orbfe_mc_extract_exchange_submit", "orbfe_mc_extract_exchange_wait",
           "orbfe_mc_match_ring", "orbfe_mc_match_outputs", "orbfe_mc_match_ring_async", "orbfe_mc_exchange_host",
           "orbfe_set_trig_mode", "orbfe_debug_trig", "orbfe_release_caches", "orbfe_search_tri_kb8", "orbfe_search_tri_3d", "orbfe_frame_create", "orbfe_search_projection_frame", "orbfe_search_projection_frames", "orbfe_frame_destroy", "orbfe_kb8_triangulate", "orbfe_search_initialization", "orbfe_stereo_fisheye_matches", "orbfe_set_kb8", "orbfe_set_ray_output", "orbfe_get_rays", "orbfe_max_keypoints", "orbfe_extract", "orbfe_extract_batch",
           "orbfe_extract_batch_device", "orbfe_sync", "orbfe_compute_stereo_matches", "orbfe_get_levels", "orbfe_get_scale_factor",
           "orbfe_get_scale_tables", "orbfe_get_features_per_level", "orbfe_get_level", "orbfe_profile_enable",
           "orbfe_profile_read", "orbfe_debug_candidates", "orbfe_debug_level_keypoints", "orbfe_debug_fixups",
           "orbfe_hamming_pairs", "orbfe_bfknn2", "orbfe_search_bow", "orbfe_search_tri", "orbfe_search_bow_batch", "orbfe_kb8_unproject",
           "orbfe_matcher_last_kernel_ms", "orbfe_matcher_time_kernels", "orbfe_search_projection", "orbfe_search_projection_last_sweeps", "orbfe_search_projection_batch",
           "orbfe_distinctive_descriptors", "orbfe_vocab_upload", "orbfe_vocab_free", "orbfe_vocab_transform",
           "orbfe_extract_batch_submit", "orbfe_extract_batch_wait", "orbfe_host_alloc", "orbfe_host_free",
           "orbfe_host_register", "orbfe_host_unregister", "orbfe_compute_stereo_matches_resident", "orbfe_extract_stereo_pair", "orbfe_extract_stereo_pair_submit", "orbfe_extract_stereo_pair_wait",
           "orbfe_hamming_pairs_device", "orbfe_bfknn2_device", "orbfe_bfknn2_frames_device", "orbfe_matcher_sync",
           "orbfe_get_device_outputs", "orbfe_extract_batch_sizes", "orbfe_set_atan_fma", "orbfe_debug_blurred_patch",
           "orbfe_vocab_load_text", "orbfe_debug_trig_cache_path", "orbfe_debug_trig_cache_payload_bytes",
           "orbfe_debug_trig_cache_write", "orbfe_debug_trig_cache_check", "orbfe_set_lanes", "orbfe_set_lane_mode", "orbfe_set_lane_input_guard", "orbfe_lanes_join", "orbfe_lanes_record",
           "orbfe_keyframe_create", "orbfe_keyframe_set_mask", "orbfe_keyframe_destroy", "orbfe_search_bow_keyframes",
           "orbfe_debug_handle_table_selftest", "orbfe_vocab_set_types", "orbfe_vocab_get_types", "orbfe_bow_create", "orbfe_bow_destroy", "orbfe_bow_set_lazy_norm", "orbfe_compute_bow", "orbfe_bow_fv", "orbfe_bow_host", "orbfe_bow_device",
           "orbfe_search_tri_batch"]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class PinnedBuffer:
    """Page-locked host memory from orbfe_host_alloc, viewed as numpy arrays (`array(shape, dtype, offset)`)."""

    def __init__(self, nbytes):
        self.L = lib()
        self.nbytes = int(nbytes)
        self.ptr = self.L.orbfe_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("orbfe_host_alloc(%d) failed" % nbytes)
        self._raw = (C.c_uint8 * self.nbytes).from_address(self.ptr)

    def array(self, shape, dtype=np.uint8, offset=0):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        assert offset + n <= self.nbytes
        return np.frombuffer(self._raw, dtype=dtype, count=int(np.prod(shape)), offset=offset).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self._raw = None
            self.L.orbfe_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


def host_register(arr):
    """Pin an existing numpy array (orbfe_host_register); returns the array.  Unpin with host_unregister."""
    _chk(lib().orbfe_host_register(arr.ctypes.data, arr.nbytes), "orbfe_host_register")
    return arr


def host_unregister(arr):
    _chk(lib().orbfe_host_unregister(arr.ctypes.data), "orbfe_host_unregister")


def _chk(r, what):
    if r < 0:
        raise OrbfeError(r, what)
    return r


class ORBextractor:
    """ORB_SLAM3::ORBextractor on one MI355X (reference include/ORBextractor.h:43-107)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0,
                 trig=TRIG_LIBM, taps=None):
        self.L = lib()
        h = C.c_void_p()
        _chk(self.L.orbfe_create(C.byref(h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device),
             "orbfe_create")
        self.h = h
        self.nfeatures, self.nlevels, self.device, self.trig = nfeatures, nlevels, device, trig
        _chk(self.L.orbfe_set_trig_mode(self.h, trig), "orbfe_set_trig_mode")
        if taps is not None:
            t = np.ascontiguousarray(taps, np.int32)
            _chk(self.L.orbfe_set_gaussian_taps(self.h, _p(t)), "orbfe_set_gaussian_taps")

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_kb8(self, params8):
        """Enable fused KannalaBrandt8::unproject (bearing rays per keypoint); None disables."""
        if params8 is None:
            _chk(self.L.orbfe_set_kb8(self.h, None), "orbfe_set_kb8")
        else:
            p = np.ascontiguousarray(params8, np.float32)
            assert p.shape == (8,)
            _chk(self.L.orbfe_set_kb8(self.h, _p(p)), "orbfe_set_kb8")

    def rays(self, n, cap, img_index=0):
        out = np.zeros((n, 3), np.float32)
        _chk(self.L.orbfe_get_rays(self.h, img_index, cap, _p(out), n), "orbfe_get_rays")
        return out

    # -- operator() ---------------------------------------------------------------
    def max_keypoints(self, rows, cols):
        return _chk(self.L.orbfe_max_keypoints(self.h, rows, cols), "orbfe_max_keypoints")

    def __call__(self, image, lapping_area=(0, 0)):
        """operator()(image, mask, keypoints, descriptors, vLappingArea) -> (monoIndex, keypoints, descriptors)."""
        image = np.asarray(image)
        if image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1  # CV_8UC1, :1076
        cap = self.max_keypoints(*image.shape)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        r = self.L.orbfe_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0],
                                 int(lapping_area[0]), int(lapping_area[1]), _p(kps), _p(desc), cap, C.byref(n))
        if r < -1:
            raise OrbfeError(r, "orbfe_extract")
        return r, kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, images, lapping_areas=None):
        """Batched operator() over same-sized images (host arrays). Returns list of (mono, kps, desc)."""
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        nimg = len(images)
        rows, cols = images[0].shape
        assert all(im.shape == (rows, cols) for im in images)
        cap = self.max_keypoints(rows, cols)
        kps = np.zeros((nimg, cap), KP_DTYPE)
        desc = np.zeros((nimg, cap, 32), np.uint8)
        n = np.zeros(nimg, np.int32)
        mono = np.zeros(nimg, np.int32)
        ptrs = (C.c_void_p * nimg)(*[im.ctypes.data for im in images])
        lap = None
        if lapping_areas is not None:
            lap = np.ascontiguousarray(lapping_areas, np.int32).reshape(nimg, 2)
        _chk(self.L.orbfe_extract_batch(self.h, nimg, ptrs, rows, cols, cols, None if lap is None else _p(lap),
                                        _p(kps), _p(desc), cap, _p(n), _p(mono)), "orbfe_extract_batch")
        return [(int(mono[i]), kps[i, : n[i]].copy(), desc[i, : n[i]].copy()) for i in range(nimg)]

    class Batch:
        """Caller-side arrays of one host-pointer batch; reusable across calls (pinned=True: in page-locked
        memory, the zero-staging path)."""

        def __init__(self, ex, nimg, rows, cols, pinned=False):
            self.nimg, self.rows, self.cols = nimg, rows, cols
            self.cap = ex.max_keypoints(rows, cols)
            nb_img, nb_k, nb_d = nimg * rows * cols, nimg * self.cap * 28, nimg * self.cap * 32
            if pinned:
                self.pin = PinnedBuffer(nb_img + nb_k + nb_d + 256)
                o1 = (nb_img + 63) // 64 * 64
                o2 = (o1 + nb_k + 63) // 64 * 64
                self.images = self.pin.array((nimg, rows, cols), np.uint8, 0)
                self.kps = self.pin.array((nimg, self.cap), KP_DTYPE, o1)
                self.desc = self.pin.array((nimg, self.cap, 32), np.uint8, o2)
            else:
                self.pin = None
                self.images = np.zeros((nimg, rows, cols), np.uint8)
                self.kps = np.zeros((nimg, self.cap), KP_DTYPE)
                self.desc = np.zeros((nimg, self.cap, 32), np.uint8)
            self.n = np.zeros(nimg, np.int32)
            self.mono = np.zeros(nimg, np.int32)
            self.lap = np.zeros((nimg, 2), np.int32)
            self.ptrs = (C.c_void_p * nimg)(*[self.images[i].ctypes.data for i in range(nimg)])

        def results(self):
            return [(int(self.mono[i]), self.kps[i, : self.n[i]].copy(), self.desc[i, : self.n[i]].copy())
                    for i in range(self.nimg)]

    def run_batch(self, b):
        """Blocking orbfe_extract_batch on a Batch's arrays (no allocation, no copies on the Python side)."""
        _chk(self.L.orbfe_extract_batch(self.h, b.nimg, b.ptrs, b.rows, b.cols, b.cols, _p(b.lap), _p(b.kps), _p(b.desc),
                                        b.cap, _p(b.n), _p(b.mono)), "orbfe_extract_batch")

    def submit_batch(self, b):
        """orbfe_extract_batch_submit: queue the batch (at most two in flight), results valid after wait_batch()."""
        _chk(self.L.orbfe_extract_batch_submit(self.h, b.nimg, b.ptrs, b.rows, b.cols, b.cols, _p(b.lap), _p(b.kps),
                                               _p(b.desc), b.cap, _p(b.n), _p(b.mono)), "orbfe_extract_batch_submit")

    def wait_batch(self):
        _chk(self.L.orbfe_extract_batch_wait(self.h), "orbfe_extract_batch_wait")

    def extract_batch_sizes(self, images, lapping_areas=None):
        """orbfe_extract_batch_sizes: images of different sizes in one call.  Returns list of (mono, kps, desc)."""
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        nimg = len(images)
        cap = max(self.max_keypoints(*im.shape) for im in images)
        rows = np.array([im.shape[0] for im in images], np.int32)
        cols = np.array([im.shape[1] for im in images], np.int32)
        strides = np.array([im.strides[0] for im in images], np.uint64)
        kps = np.zeros((nimg, cap), KP_DTYPE)
        desc = np.zeros((nimg, cap, 32), np.uint8)
        n = np.zeros(nimg, np.int32)
        mono = np.zeros(nimg, np.int32)
        ptrs = (C.c_void_p * nimg)(*[im.ctypes.data for im in images])
        lap = None if lapping_areas is None else np.ascontiguousarray(lapping_areas, np.int32).reshape(nimg, 2)
        _chk(self.L.orbfe_extract_batch_sizes(self.h, nimg, ptrs, _p(rows), _p(cols), _p(strides),
                                              None if lap is None else _p(lap), _p(kps), _p(desc), cap, _p(n), _p(mono)),
             "orbfe_extract_batch_sizes")
        return [(int(mono[i]), kps[i, : n[i]].copy(), desc[i, : n[i]].copy()) for i in range(nimg)]

    def set_atan_fma(self, on=True):
        _chk(self.L.orbfe_set_atan_fma(self.h, int(on)), "orbfe_set_atan_fma")

    def debug_blurred_patch(self, kp_index, img_index=0):
        out = np.zeros((37, 37), np.uint8)
        _chk(self.L.orbfe_debug_blurred_patch(self.h, img_index, kp_index, _p(out)), "orbfe_debug_blurred_patch")
        return out

    def extract_batch_device(self, d_imgs_ptr, nimg, rows, cols, pitch, img_stride, lap, d_kps_ptr, d_desc_ptr, cap,
                             d_n_ptr, d_mono_ptr):
        """Device-resident batched operator(): every pointer is a device address (int)."""
        return _chk(self.L.orbfe_extract_batch_device(self.h, nimg, d_imgs_ptr, rows, cols, pitch, img_stride,
                                                      int(lap[0]), int(lap[1]), d_kps_ptr, d_desc_ptr, cap, d_n_ptr,
                                                      d_mono_ptr), "orbfe_extract_batch_device")

    def device_outputs(self):
        """(d_kps, d_desc, d_n, cap, nimg): where the last extraction left its results in HBM (device addresses)."""
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap, nimg = C.c_int(), C.c_int()
        _chk(self.L.orbfe_get_device_outputs(self.h, C.byref(k), C.byref(d), C.byref(n), C.byref(cap), C.byref(nimg)),
             "orbfe_get_device_outputs")
        return k.value, d.value, n.value, cap.value, nimg.value

    def set_stream(self, stream_ptr):
        _chk(self.L.orbfe_set_stream(self.h, stream_ptr), "orbfe_set_stream")

    def sync(self):
        _chk(self.L.orbfe_sync(self.h), "orbfe_sync")

    def set_lanes(self, lanes, mode=None):
        """orbfe_set_lanes: up to `lanes` (1..4) device-pointer batches in flight on streams the context owns, whole batches
        round-robin.  See include/orbfe.h for the ordering rules."""
        _chk(self.L.orbfe_set_lanes(self.h, int(lanes)), "orbfe_set_lanes")
        if mode is not None:
            _chk(self.L.orbfe_set_lane_mode(self.h, int(mode)), "orbfe_set_lane_mode")

    def set_lane_input_guard(self, on):
        """orbfe_set_lane_input_guard: off = the caller never rewrites the images of a call that may still be in flight."""
        _chk(self.L.orbfe_set_lane_input_guard(self.h, 1 if on else 0), "orbfe_set_lane_input_guard")

    def lanes_join(self):
        """orbfe_lanes_join: the context's stream waits for every lane (no host wait)."""
        _chk(self.L.orbfe_lanes_join(self.h), "orbfe_lanes_join")

    # -- getters (include/ORBextractor.h:61-83) -------------------------------------
    def GetLevels(self):
        return self.L.orbfe_get_levels(self.h)

    def GetScaleFactor(self):
        return self.L.orbfe_get_scale_factor(self.h)

    def _tables(self):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        self.L.orbfe_get_scale_tables(self.h, *[_p(o) for o in out])
        return out

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        o = np.zeros(self.nlevels, np.int32)
        self.L.orbfe_get_features_per_level(self.h, _p(o))
        return o

    def image_pyramid_level(self, level, img_index=0):
        """Padded buffer behind mvImagePyramid[level] (level + 19-px REFLECT_101 frame)."""
        r, c = C.c_int(), C.c_int()
        _chk(self.L.orbfe_get_level(self.h, img_index, level, None, 0, C.byref(r), C.byref(c)), "orbfe_get_level")
        out = np.zeros((r.value, c.value), np.uint8)
        _chk(self.L.orbfe_get_level(self.h, img_index, level, _p(out), out.strides[0], C.byref(r), C.byref(c)),
             "orbfe_get_level")
        return out

    # -- profiling / debug taps ---------------------------------------------------
    def profile(self, on=True):
        _chk(self.L.orbfe_profile_enable(self.h, int(on)), "orbfe_profile_enable")

    def stage_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        _chk(self.L.orbfe_profile_read(self.h, _p(ms)), "orbfe_profile_read")
        return dict(zip(STAGES, ms.tolist()))

    @staticmethod
    def _unpack(a):
        return (a & 0xFFF).astype(np.int32), ((a >> 12) & 0xFFF).astype(np.int32), (a >> 24).astype(np.int32)

    def debug_candidates(self, level, img_index=0, cap=1 << 20):
        out = np.zeros(cap, np.uint32)
        n = _chk(self.L.orbfe_debug_candidates(self.h, img_index, level, _p(out), cap), "orbfe_debug_candidates")
        return self._unpack(out[:n])

    def debug_level_keypoints(self, level, img_index=0, cap=1 << 16):
        out = np.zeros(cap, np.uint32)
        n = _chk(self.L.orbfe_debug_level_keypoints(self.h, img_index, level, _p(out), cap),
                 "orbfe_debug_level_keypoints")
        return self._unpack(out[:n])

    def debug_fixups(self):
        return self.L.orbfe_debug_fixups(self.h)

    def debug_trig(self, angles_deg):
        """(table kind: 2 libm values / 1 libm codes / 0 none, cos, sin) the descriptor kernel uses for these angles."""
        ang = np.ascontiguousarray(angles_deg, np.float32)
        a = np.empty_like(ang)
        b = np.empty_like(ang)
        r = _chk(self.L.orbfe_debug_trig(self.h, ang.ctypes.data, len(ang), a.ctypes.data, b.ctypes.data),
                 "orbfe_debug_trig")
        return int(r), a, b


def compute_stereo_matches(exL, exR, kpsL, descL, kpsR, descR, mb, mbf):
    """Frame::ComputeStereoMatches (src/Frame.cc:797-967) on the pyramids held by the two extractors."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    dL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    dR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    uR = np.zeros(len(kpsL), np.float32)
    dep = np.zeros(len(kpsL), np.float32)
    n = _chk(lib().orbfe_compute_stereo_matches(exL.h, exR.h, _p(kpsL), _p(dL), len(kpsL), _p(kpsR), _p(dR), len(kpsR),
                                                mb, mbf, _p(uR), _p(dep)), "orbfe_compute_stereo_matches")
    return n, uR, dep


def compute_stereo_matches_resident(exL, exR, nL, mb, mbf, imgL=0, imgR=0):
    """The same on what the two extractors' last calls left on the device (no keypoint / descriptor upload)."""
    uR = np.zeros(max(nL, 1), np.float32)
    dep = np.zeros(max(nL, 1), np.float32)
    n = _chk(lib().orbfe_compute_stereo_matches_resident(exL.h, imgL, exR.h, imgR, mb, mbf, _p(uR), _p(dep), nL),
             "orbfe_compute_stereo_matches_resident")
    return n, uR[:nL], dep[:nL]


def extract_stereo_pair(ex, imgL, imgR, mb, mbf, lap=None):
    """orbfe_extract_stereo_pair: both images and Frame::ComputeStereoMatches in one call with one host wait.
    Returns (matches, (monoL, kpsL, descL), (monoR, kpsR, descR), uRight, depth)."""
    imgL = np.ascontiguousarray(imgL, np.uint8)
    imgR = np.ascontiguousarray(imgR, np.uint8)
    assert imgL.shape == imgR.shape and imgL.ndim == 2
    rows, cols = imgL.shape
    cap = ex.max_keypoints(rows, cols)
    kps = np.zeros((2, cap), KP_DTYPE)
    desc = np.zeros((2, cap, 32), np.uint8)
    n = np.zeros(2, np.int32)
    mono = np.zeros(2, np.int32)
    uR = np.zeros(cap, np.float32)
    dep = np.zeros(cap, np.float32)
    lap4 = None if lap is None else np.ascontiguousarray(lap, np.int32).reshape(4)
    m = _chk(lib().orbfe_extract_stereo_pair(ex.h, imgL.ctypes.data, imgR.ctypes.data, rows, cols, cols,
                                             None if lap4 is None else lap4.ctypes.data, kps.ctypes.data, desc.ctypes.data, cap,
                                             n.ctypes.data, mono.ctypes.data, mb, mbf, uR.ctypes.data, dep.ctypes.data),
             "orbfe_extract_stereo_pair")
    return (m, (int(mono[0]), kps[0, :n[0]].copy(), desc[0, :n[0]].copy()), (int(mono[1]), kps[1, :n[1]].copy(), desc[1, :n[1]].copy()),
            uR[:n[0]].copy(), dep[:n[0]].copy())


class StereoPairStream:
    """orbfe_extract_stereo_pair_submit / _wait: up to `lanes` stereo frames in flight on one context.  submit() queues a frame,
    wait() returns the OLDEST frame's result in the format of extract_stereo_pair()."""

    def __init__(self, ex, rows, cols):
        self.ex, self.rows, self.cols = ex, rows, cols
        self.cap = ex.max_keypoints(rows, cols)
        self.q = []

    def submit(self, imgL, imgR, mb, mbf, lap=None):
        imgL = np.ascontiguousarray(imgL, np.uint8)
        imgR = np.ascontiguousarray(imgR, np.uint8)
        assert imgL.shape == imgR.shape == (self.rows, self.cols)
        cap = self.cap
        b = dict(imgL=imgL, imgR=imgR, kps=np.zeros((2, cap), KP_DTYPE), desc=np.zeros((2, cap, 32), np.uint8),
                 n=np.zeros(2, np.int32), mono=np.zeros(2, np.int32), uR=np.zeros(cap, np.float32), dep=np.zeros(cap, np.float32),
                 lap=None if lap is None else np.ascontiguousarray(lap, np.int32).reshape(4))
        _chk(lib().orbfe_extract_stereo_pair_submit(self.ex.h, imgL.ctypes.data, imgR.ctypes.data, self.rows, self.cols, self.cols,
                                                    None if b["lap"] is None else b["lap"].ctypes.data, b["kps"].ctypes.data,
                                                    b["desc"].ctypes.data, cap, b["n"].ctypes.data, b["mono"].ctypes.data, mb, mbf,
                                                    b["uR"].ctypes.data, b["dep"].ctypes.data), "orbfe_extract_stereo_pair_submit")
        self.q.append(b)  # (the arrays belong to the library until the frame's wait returns)

    def wait(self):
        # (the record leaves the queue whatever the call returns -- the library has retired the frame either way, and a
        # record left behind would hand every later wait() the previous frame's arrays)
        if not self.q:
            raise OrbfeError(ERR_STATE, "StereoPairStream.wait without a frame in flight")
        b = self.q.pop(0)
        m = _chk(lib().orbfe_extract_stereo_pair_wait(self.ex.h), "orbfe_extract_stereo_pair_wait")
        n, mono, kps, desc = b["n"], b["mono"], b["kps"], b["desc"]
        return (m, (int(mono[0]), kps[0, :n[0]].copy(), desc[0, :n[0]].copy()), (int(mono[1]), kps[1, :n[1]].copy(), desc[1, :n[1]].copy()),
                b["uR"][:n[0]].copy(), b["dep"][:n[0]].copy())

    def close(self):
        """Collects every frame still in flight: their output arrays belong to the library until their wait returns, and this
        object is what keeps them alive -- dropping it with frames in flight would let the next wait on the context copy into
        freed memory."""
        while self.q:
            b = self.q.pop(0)
            if getattr(self.ex, "h", None):
                lib().orbfe_extract_stereo_pair_wait(self.ex.h)  # (an error code is all the same here: the frame is retired)
            del b

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------ matcher
FV_RESIDENT = -0x0B0F  # ORBFE_FV_RESIDENT (include/orbfe.h)


def _fv(fv):
    if isinstance(fv, _FV):  # (already in the C layout: what orbfe_bow_fv returned earlier, handed on as it is)
        return fv, fv
    if isinstance(fv, Bow):  # the FeatureVector of an orbfe_bow handle, read where orbfe_compute_bow left it (orbfe_bow_fv)
        s = _FV()
        _chk(lib().orbfe_bow_fv(fv.h, C.byref(s)), "orbfe_bow_fv")
        return s, fv
    node_ids, offsets, indices = fv
    node_ids = np.ascontiguousarray(node_ids, np.uint32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    indices = np.ascontiguousarray(indices, np.int32)
    s = _FV(len(node_ids), node_ids.ctypes.data, offsets.ctypes.data, indices.ctypes.data)
    return s, (node_ids, offsets, indices)


def release_caches(device=0):
    """Frees the per-process libm trig table of `device` (rebuilt on the next ORBFE_TRIG_LIBM extraction)."""
    _chk(lib().orbfe_release_caches(device), "orbfe_release_caches")


def matcher_last_kernel_ms():
    return float(lib().orbfe_matcher_last_kernel_ms())


def matcher_time_kernels(on=True):
    lib().orbfe_matcher_time_kernels(int(on))


def hamming_pairs(A, B, device=0):
    """ORBmatcher::DescriptorDistance over all pairs (src/ORBmatcher.cc:2591-2607)."""
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32)
    B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    D = np.zeros((len(A), len(B)), np.uint16)
    _chk(lib().orbfe_hamming_pairs(device, _p(A), len(A), _p(B), len(B), _p(D)), "orbfe_hamming_pairs")
    return D


def bfknn2(Q, T, device=0):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (src/Frame.cc:1137)."""
    Q = np.ascontiguousarray(Q, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(Q), 2), np.int32)
    dist = np.zeros((len(Q), 2), np.int32)
    _chk(lib().orbfe_bfknn2(device, _p(Q), len(Q), _p(T), len(T), _p(idx), _p(dist)), "orbfe_bfknn2")
    return idx, dist


KNN2_JOB_DTYPE = np.dtype([("q_desc", "<u8"), ("q_count", "<u8"), ("t_desc", "<u8"), ("t_count", "<u8")])  # orbfe_knn2_job


def hamming_pairs_device(dA, nA, dB, nB, dD, stream=None, device=0):
    """orbfe_hamming_pairs_device: device addresses (ints), asynchronous on `stream` (a hipStream_t value or None)."""
    _chk(lib().orbfe_hamming_pairs_device(device, stream, dA, nA, dB, nB, dD), "orbfe_hamming_pairs_device")


def bfknn2_device(dQ, nQ, dT, nT, d_idx, d_dist, stream=None, device=0):
    _chk(lib().orbfe_bfknn2_device(device, stream, dQ, nQ, dT, nT, d_idx, d_dist), "orbfe_bfknn2_device")


def bfknn2_frames_device(d_jobs, njobs, cap, d_idx, d_dist, stream=None, device=0):
    """Cross-camera knn-2: `njobs` orbfe_knn2_job records at device address d_jobs (KNN2_JOB_DTYPE)."""
    _chk(lib().orbfe_bfknn2_frames_device(device, stream, d_jobs, njobs, cap, d_idx, d_dist), "orbfe_bfknn2_frames_device")


def matcher_sync(device=0):
    _chk(lib().orbfe_matcher_sync(device), "orbfe_matcher_sync")


def _desc_arg(d):
    """Descriptor argument of a matcher call: a host array, or an (address, rows) pair naming device memory."""
    if isinstance(d, tuple):
        return int(d[0]), int(d[1]), None
    a = np.ascontiguousarray(d, np.uint8).reshape(-1, 32)
    return a.ctypes.data, len(a), a


def search_bow(desc1, mask1, ang1, fv1, desc2, mask2, ang2, fv2, variant, nnratio, check_ori=True, Nleft=-1,
               limit1=-1, limit2=-1, device=0):
    """ORBmatcher::SearchByBoW: variant 0 = (KeyFrame*, Frame&) :269-471, 1 = (KeyFrame*, KeyFrame*) :823-963."""
    p1, n1, k1d = _desc_arg(desc1)
    p2, n2, k2d = _desc_arg(desc2)
    m1 = np.ascontiguousarray(mask1, np.uint8)
    m2 = np.ascontiguousarray(mask2 if mask2 is not None else np.ones(n2), np.uint8)
    a1 = np.ascontiguousarray(ang1, np.float32)
    a2 = np.ascontiguousarray(ang2, np.float32)
    f1, k1 = _fv(fv1)
    f2, k2 = _fv(fv2)
    args = _BowArgs(p1, n1, m1.ctypes.data, a1.ctypes.data, f1, limit1, p2, n2,
                    m2.ctypes.data, a2.ctypes.data, f2, limit2, Nleft, nnratio, int(check_ori), variant)
    match = np.zeros(n2 if variant == 0 else n1, np.int32)
    n = _chk(lib().orbfe_search_bow(device, C.byref(args), _p(match)), "orbfe_search_bow")
    return n, match


def _bow_args(desc1, mask1, ang1, fv1, desc2, mask2, ang2, fv2, variant, nnratio, check_ori, Nleft, limit1, limit2):
    p1, n1, d1 = _desc_arg(desc1)
    p2, n2, d2 = _desc_arg(desc2)
    m1 = np.ascontiguousarray(mask1, np.uint8)
    m2 = np.ascontiguousarray(mask2 if mask2 is not None else np.ones(n2), np.uint8)
    a1 = np.ascontiguousarray(ang1, np.float32)
    a2 = np.ascontiguousarray(ang2, np.float32)
    f1, k1 = _fv(fv1)
    f2, k2 = _fv(fv2)
    args = _BowArgs(p1, n1, m1.ctypes.data, a1.ctypes.data, f1, limit1, p2, n2,
                    m2.ctypes.data, a2.ctypes.data, f2, limit2, Nleft, nnratio, int(check_ori), variant)
    return args, (d1, d2, m1, m2, a1, a2, k1, k2), (n2 if variant == 0 else n1)


def search_bow_batch(problems, device=0):
    """problems: list of dicts with the keyword arguments of search_bow(); one launch for all of them.
    Returns [(nmatches, match), ...]."""
    n = len(problems)
    arr = (_BowArgs * n)()
    keep, outs = [], []
    for i, pr in enumerate(problems):
        a, k, nout = _bow_args(pr["desc1"], pr["mask1"], pr["ang1"], pr["fv1"], pr["desc2"], pr.get("mask2"),
                               pr["ang2"], pr["fv2"], pr["variant"], pr["nnratio"], pr.get("check_ori", True),
                               pr.get("Nleft", -1), pr.get("limit1", -1), pr.get("limit2", -1))
        arr[i] = a
        keep.append(k)
        outs.append(np.zeros(nout, np.int32))
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    nm = np.zeros(n, np.int32)
    _chk(lib().orbfe_search_bow_batch(device, n, arr, ptrs, _p(nm)), "orbfe_search_bow_batch")
    return [(int(nm[i]), outs[i]) for i in range(n)]


def search_triangulation(desc1, hasMP1, kp1xy, ang1, oct1, uR1, fv1, desc2, hasMP2, kp2xy, ang2, oct2, uR2, fv2, F12,
                         ep, scaleFactors2, levelSigma2_2, only_stereo=False, coarse=False, check_ori=True, device=0):
    """ORBmatcher::SearchForTriangulation_ (src/ORBmatcher.cc:1208-1449), pinhole gate."""
    def prep(desc, has, xy, ang, oc, ur):
        return (np.ascontiguousarray(desc, np.uint8).reshape(-1, 32), np.ascontiguousarray(has, np.uint8),
                np.ascontiguousarray(xy, np.float32).reshape(-1, 2), np.ascontiguousarray(ang, np.float32),
                np.ascontiguousarray(oc, np.int32), np.ascontiguousarray(ur, np.float32))

    d1, h1, x1, a1, o1, u1 = prep(desc1, hasMP1, kp1xy, ang1, oct1, uR1)
    d2, h2, x2, a2, o2, u2 = prep(desc2, hasMP2, kp2xy, ang2, oct2, uR2)
    f1, k1 = _fv(fv1)
    f2, k2 = _fv(fv2)
    sf2 = np.ascontiguousarray(scaleFactors2, np.float32)
    ls2 = np.ascontiguousarray(levelSigma2_2, np.float32)
    F = (C.c_float * 9)(*[float(v) for v in np.asarray(F12, np.float32).reshape(9)])
    e = (C.c_float * 2)(float(ep[0]), float(ep[1]))
    args = _TriArgs(d1.ctypes.data, len(d1), h1.ctypes.data, x1.ctypes.data, a1.ctypes.data, o1.ctypes.data,
                    u1.ctypes.data, f1, d2.ctypes.data, len(d2), h2.ctypes.data, x2.ctypes.data, a2.ctypes.data,
                    o2.ctypes.data, u2.ctypes.data, f2, F, e, sf2.ctypes.data, ls2.ctypes.data, len(sf2),
                    int(only_stereo), int(coarse), int(check_ori))
    pairs = np.zeros((max(len(d1), 1), 2), np.int32)
    n = _chk(lib().orbfe_search_tri(device, C.byref(args), _p(pairs)), "orbfe_search_tri")
    return pairs[:n].copy()


class _KeyFrameArgs(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("n", C.c_int), ("mask", C.c_void_p), ("angle", C.c_void_p), ("kp_xy", C.c_void_p),
                ("octave", C.c_void_p), ("uRight", C.c_void_p), ("fv", _FV)]


class _TriPair(C.Structure):
    _fields_ = [("F12", C.c_float * 9), ("ep", C.c_float * 2), ("scaleFactors2", C.c_void_p), ("levelSigma2_2", C.c_void_p),
                ("nlevels2", C.c_int), ("only_stereo", C.c_int), ("coarse", C.c_int), ("check_orientation", C.c_int),
                ("hasMP2", C.c_void_p)]


class KeyFrameHandle:
    """orbfe_keyframe_*: a keyframe's descriptors, flags, angles, FeatureVector (and keypoints / octaves / mvuRight when
    given) resident on the device.  `desc` may be a device address (int)."""

    def __init__(self, desc, mask, angle, fv, kp_xy=None, octave=None, uRight=None, device=0):
        self.L = lib()
        self.device = device
        p, n, keep = _desc_arg(desc)
        m = np.ascontiguousarray(mask, np.uint8)
        a = None if angle is None else np.ascontiguousarray(angle, np.float32)
        f, kf = _fv(fv)
        xy = None if kp_xy is None else np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        oc = None if octave is None else np.ascontiguousarray(octave, np.int32)
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        args = _KeyFrameArgs(p, n, m.ctypes.data, None if a is None else a.ctypes.data, None if xy is None else xy.ctypes.data,
                             None if oc is None else oc.ctypes.data, None if ur is None else ur.ctypes.data, f)
        self.h = C.c_void_p()
        self.n = n
        _chk(self.L.orbfe_keyframe_create(C.byref(self.h), device, C.byref(args)), "orbfe_keyframe_create")

    def set_mask(self, mask):
        m = np.ascontiguousarray(mask, np.uint8)
        assert len(m) == self.n
        _chk(self.L.orbfe_keyframe_set_mask(self.h, _p(m)), "orbfe_keyframe_set_mask")

    def close(self):
        if self.h:
            self.L.orbfe_keyframe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_bow_keyframes(problems, device=0):
    """orbfe_search_bow_keyframes: problems as for search_bow_batch; `kf1` / `kf2` (KeyFrameHandle) replace set 1 / set 2,
    whose array arguments may then be omitted.  Returns [(nmatches, match), ...]."""
    n = len(problems)
    arr = (_BowArgs * n)()
    k1 = (C.c_void_p * n)()
    k2 = (C.c_void_p * n)()
    keep, outs = [], []
    for i, pr in enumerate(problems):
        h1, h2 = pr.get("kf1"), pr.get("kf2")
        dummy = np.zeros((1, 32), np.uint8)
        one = np.ones(1, np.uint8)
        efv = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
        a, k, nout = _bow_args(dummy if h1 else pr["desc1"], pr["mask1"] if (not h1 or pr.get("mask1") is not None) else one,
                               np.zeros(1) if h1 else pr["ang1"],
                               efv if h1 else pr["fv1"], dummy if h2 else pr["desc2"],
                               pr.get("mask2") if (not h2 or pr.get("mask2") is not None) else one,
                               np.zeros(1) if h2 else pr["ang2"], efv if h2 else pr["fv2"], pr["variant"], pr["nnratio"],
                               pr.get("check_ori", True), pr.get("Nleft", -1), pr.get("limit1", -1), pr.get("limit2", -1))
        if h1 and pr.get("mask1") is None:
            a.mask1 = None  # (NULL = the handle's own flags; a non-null pointer is read as this call's n1 flags)
        if h2 and pr.get("mask2") is None:
            a.mask2 = None
        arr[i] = a
        k1[i] = h1.h if h1 else None
        k2[i] = h2.h if h2 else None
        n1 = h1.n if h1 else a.n1
        n2 = h2.n if h2 else a.n2
        keep.append(k)
        outs.append(np.zeros(n2 if pr["variant"] == 0 else n1, np.int32))
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    nm = np.zeros(n, np.int32)
    _chk(lib().orbfe_search_bow_keyframes(device, n, k1, k2, arr, ptrs, _p(nm)), "orbfe_search_bow_keyframes")
    return [(int(nm[i]), outs[i]) for i in range(n)]


def _keep(keep, v, dt):
    a = np.ascontiguousarray(v, dt)
    keep.append(a)
    return a.ctypes.data


def search_tri_batch(kf1, neighbours, hasMP1=None):
    """orbfe_search_tri_batch: kf1 (KeyFrameHandle with keypoints) against neighbours = [dict(kf=handle, F12, ep, sf, sig,
    only_stereo, coarse, check_ori[, hasMP2])], one launch; hasMP1 / hasMP2 = this call's flags instead of the handles'.
    Returns a list of pairs[n, 2] arrays."""
    n = len(neighbours)
    arr = (_TriPair * n)()
    kfs = (C.c_void_p * n)()
    keep, outs = [], []
    for i, q in enumerate(neighbours):
        sf = np.ascontiguousarray(q["sf"], np.float32)
        sg = np.ascontiguousarray(q["sig"], np.float32)
        keep += [sf, sg]
        arr[i] = _TriPair((C.c_float * 9)(*[float(v) for v in np.asarray(q["F12"], np.float32).reshape(9)]),
                          (C.c_float * 2)(float(q["ep"][0]), float(q["ep"][1])), sf.ctypes.data, sg.ctypes.data, len(sf),
                          int(q.get("only_stereo", False)), int(q.get("coarse", False)), int(q.get("check_ori", True)),
                          None if q.get("hasMP2") is None else _keep(keep, q["hasMP2"], np.uint8))
        kfs[i] = q["kf"].h
        outs.append(np.zeros((max(kf1.n, 1), 2), np.int32))
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    npairs = np.zeros(max(n, 1), np.int32)
    h1 = None if hasMP1 is None else _keep(keep, hasMP1, np.uint8)
    _chk(lib().orbfe_search_tri_batch(kf1.h, h1, n, kfs, arr, ptrs, _p(npairs)), "orbfe_search_tri_batch")
    return [outs[i][:npairs[i]].copy() for i in range(n)]


def search_initialization(problem, device=0):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821): (nmatches, vnMatches12)."""
    a, keep, n1 = _init_args(problem)
    m = np.full(max(n1, 1), -1, np.int32)
    r = _chk(lib().orbfe_search_initialization(device, C.byref(a), _p(m)), "orbfe_search_initialization")
    return r, m[:n1]


def search_triangulation_kb8(I, only_stereo=False, coarse=False, check_ori=True, device=0):
    """SearchForTriangulation_ with the KannalaBrandt8 gate; I = dict of tests/matcher_inputs.tri_kb8_inputs."""
    keep = []

    def arr(v, dt, shape=None):
        if v is None:
            return None
        a = np.ascontiguousarray(v, dt)
        if shape:
            a = a.reshape(shape)
        keep.append(a)
        return a.ctypes.data

    f1, k1 = _fv(I["fv1"])
    f2, k2 = _fv(I["fv2"])
    sf = np.ascontiguousarray(I["sf"], np.float32)
    s1 = np.ascontiguousarray(I["sig1"], np.float32)
    s2 = np.ascontiguousarray(I["sig2"], np.float32)
    a = _TriKb8Args(arr(I["d1"], np.uint8), len(I["d1"]), arr(I["has1"], np.uint8), arr(I["kp1"], np.float32),
                    arr(I["a1"], np.float32), arr(I["oct1"], np.int32), arr(I.get("u1"), np.float32), f1, int(I["Nleft1"]),
                    arr(I["d2"], np.uint8), len(I["d2"]), arr(I["has2"], np.uint8), arr(I["kp2"], np.float32),
                    arr(I["a2"], np.float32), arr(I["oct2"], np.int32), arr(I.get("u2"), np.float32), f2, int(I["Nleft2"]),
                    arr(I["P1L"], np.float32), arr(I.get("P1R"), np.float32), arr(I["P2L"], np.float32),
                    arr(I.get("P2R"), np.float32), arr(I["R12"], np.float32), arr(I["t12"], np.float32),
                    (C.c_float * 2)(float(I["ep"][0]), float(I["ep"][1])), sf.ctypes.data, s1.ctypes.data, s2.ctypes.data,
                    len(s1), len(s2), int(only_stereo), int(coarse), int(check_ori))
    pairs = np.zeros((max(len(I["d1"]), 1), 2), np.int32)
    n = _chk(lib().orbfe_search_tri_kb8(device, C.byref(a), _p(pairs)), "orbfe_search_tri_kb8")
    return pairs[:n].copy()


def search_triangulation_3d(I, check_ori=True, device=0):
    """The SearchForTriangulation overload that returns the triangulated points (src/ORBmatcher.cc:1452-1641);
    I = dict of tests/matcher_inputs.tri3d_inputs.  Returns (pairs[n,2], points[n,3])."""
    keep = []

    def arr(v, dt):
        if v is None:
            return None
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data

    f1, k1 = _fv(I["fv1"])
    f2, k2 = _fv(I["fv2"])
    s1 = np.ascontiguousarray(I["sig1"], np.float32)
    s2 = np.ascontiguousarray(I["sig2"], np.float32)
    T = I["Tcw"]  # 1L, 1R, 2L, 2R
    a = _Tri3dArgs(arr(I["d1"], np.uint8), len(I["d1"]), arr(I["has1"], np.uint8), arr(I["kp1"], np.float32),
                   arr(I["a1"], np.float32), arr(I["oct1"], np.int32), f1, int(I["Nleft1"]),
                   arr(I["d2"], np.uint8), len(I["d2"]), arr(I["has2"], np.uint8), arr(I["kp2"], np.float32),
                   arr(I["a2"], np.float32), arr(I["oct2"], np.int32), f2, int(I["Nleft2"]),
                   arr(I.get("P1L"), np.float32), arr(I.get("P1R"), np.float32), arr(I.get("P2L"), np.float32),
                   arr(I.get("P2R"), np.float32), arr(T[0], np.float32), arr(T[1], np.float32), arr(T[2], np.float32),
                   arr(T[3], np.float32), s1.ctypes.data, s2.ctypes.data, len(s1), len(s2), int(check_ori))
    n1 = max(len(I["d1"]), 1)
    pairs = np.zeros((n1, 2), np.int32)
    points = np.zeros((n1, 3), np.float32)
    n = _chk(lib().orbfe_search_tri_3d(device, C.byref(a), _p(pairs), _p(points)), "orbfe_search_tri_3d")
    return pairs[:n].copy(), points[:n].copy()


def kb8_triangulate(P1, P2, kp1, kp2, R12, t12, sigma1, sigma2, device=0):
    """KannalaBrandt8::TriangulateMatches_ for explicit pairs: (z1 = depth in camera 1 or -1, p3D[n,3])."""
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
    kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
    n = len(kp1)
    A = [np.ascontiguousarray(v, np.float32) for v in (P1, P2, R12, t12, sigma1, sigma2)]
    z = np.zeros(max(n, 1), np.float32)
    X = np.zeros((max(n, 1), 3), np.float32)
    _chk(lib().orbfe_kb8_triangulate(device, _p(A[0]), _p(A[1]), _p(kp1), _p(kp2), _p(A[2]), _p(A[3]), _p(A[4]), _p(A[5]), n,
                                     _p(z), _p(X)), "orbfe_kb8_triangulate")
    return z[:n], X[:n]


def stereo_fisheye_matches(descL, kpL, octL, descR, kpR, octR, P1, P2, Rlr, tlr, level_sigma2, device=0):
    """Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1119-1159): (nMatches, leftToRight, rightToLeft, depth, p3D)."""
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
    n = _chk(lib().orbfe_stereo_fisheye_matches(device, _p(dL), _p(kL), _p(oL), nL, _p(dR), _p(kR), _p(oR), nR, _p(A[0]),
                                                _p(A[1]), _p(A[2]), _p(A[3]), _p(A[4]), len(A[4]), _p(l2r), _p(r2l), _p(dep),
                                                _p(X)), "orbfe_stereo_fisheye_matches")
    return n, l2r[:nL], r2l[:nR], dep[:nL], X[:nL]


def search_projection(problem, device=0):
    """Inner loops of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:44-197 mode 0; :2193-2419, :2421-2541
    mode 1); `problem` holds the fields of orbfe_proj_args.  Returns (nmatches, q_match, feat_match)."""
    a, keep, n, nq = _proj_args(problem)
    qm = np.full(max(nq, 1), -1, np.int32)
    fm = np.full(max(n, 1), -1, np.int32)
    r = _chk(lib().orbfe_search_projection(device, C.byref(a), _p(qm), _p(fm)), "orbfe_search_projection")
    return r, qm[:nq], fm[:n]


class ProjectionFrame:
    """orbfe_frame: the frame side of the projection searches (descriptors, keypoints, mvuRight, the grid) resident on
    the device.  `problem` supplies the frame fields of orbfe_proj_args; `desc_ptr` = (device pointer, n) uses
    descriptors that are already in HBM (ORBextractor.device_outputs())."""

    def __init__(self, problem, device=0, desc_ptr=None):
        pr = dict(problem)
        n = len(pr["kx"])
        for q in ("qdesc", "qx", "qy", "qr", "qmin_level", "qmax_level"):  # (no queries yet)
            pr.setdefault(q, np.zeros((0, 32), np.uint8) if q == "qdesc" else np.zeros(0, np.float32))
        pr.setdefault("mode", 0)
        pr.setdefault("nnratio", 0.8)
        a, keep, _, _ = _proj_args(pr)
        if desc_ptr is not None:
            a.desc = int(desc_ptr[0])
            assert int(desc_ptr[1]) == n
        self.n = n
        self.h = C.c_void_p()
        L = lib()
        L.orbfe_frame_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        L.orbfe_search_projection_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbfe_frame_destroy.argtypes = [C.c_void_p]
        L.orbfe_frame_destroy.restype = None
        _chk(L.orbfe_frame_create(C.byref(self.h), device, C.byref(a)), "orbfe_frame_create")

    def search(self, problem):
        """orbfe_search_projection_frame: only the queries, thresholds, `taken` and the partner tables of `problem` are read."""
        pr = dict(problem)
        for k in ("desc", "kx", "ky", "octave", "angle", "uright"):
            pr.pop(k, None)
        pr["kx"] = np.zeros(0, np.float32)  # (_proj_args takes n from it; the library takes it from the handle)
        a, keep, _, nq = _proj_args(pr)
        a.kx = None
        qm = np.full(max(nq, 1), -1, np.int32)
        fm = np.full(max(self.n, 1), -1, np.int32)
        r = _chk(lib().orbfe_search_projection_frame(self.h, C.byref(a), _p(qm), _p(fm)), "orbfe_search_projection_frame")
        return r, qm[:nq], fm[:self.n]

    @staticmethod
    def search_many(frames, problems):
        """orbfe_search_projection_frames: problems[k] against frames[k] (handles may repeat) in one call.  Returns a list of
        (nmatches, q_match, feat_match)."""
        count = len(problems)
        assert len(frames) == count
        if count == 0:
            return []
        args = (_ProjArgs * count)()
        keep, qms, fms, nqs = [], [], [], []
        for k, (fr, problem) in enumerate(zip(frames, problems)):
            pr = dict(problem)
            for key in ("desc", "kx", "ky", "octave", "angle", "uright"):
                pr.pop(key, None)
            pr["kx"] = np.zeros(0, np.float32)
            a, kp, _, nq = _proj_args(pr)
            a.kx = None
            args[k] = a
            keep.append(kp)
            nqs.append(nq)
            qms.append(np.full(max(nq, 1), -1, np.int32))
            fms.append(np.full(max(fr.n, 1), -1, np.int32))
        hs = (C.c_void_p * count)(*[fr.h for fr in frames])
        qp = (C.c_void_p * count)(*[q.ctypes.data for q in qms])
        fp = (C.c_void_p * count)(*[f.ctypes.data for f in fms])
        nm = np.zeros(count, np.int32)
        L = lib()
        L.orbfe_search_projection_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.orbfe_search_projection_frames(hs, args, count, qp, fp, _p(nm)), "orbfe_search_projection_frames")
        return [(int(nm[k]), qms[k][:nqs[k]], fms[k][:frames[k].n]) for k in range(count)]

    def close(self):
        if self.h:
            lib().orbfe_frame_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_projection_batch(problems, device=0):
    """`len(problems)` independent projection searches in one call (orbfe_search_projection_batch); returns a list
    of (nmatches, q_match, feat_match), one per problem."""
    cnt = len(problems)
    arr = (_ProjArgs * max(cnt, 1))()
    keeps, qms, fms = [], [], []
    for k, pr in enumerate(problems):
        a, keep, n, nq = _proj_args(pr)
        arr[k] = a
        keeps.append(keep)
        qms.append(np.full(max(nq, 1), -1, np.int32))
        fms.append(np.full(max(n, 1), -1, np.int32))
    qp = (C.c_void_p * max(cnt, 1))(*[q.ctypes.data for q in qms])
    fp = (C.c_void_p * max(cnt, 1))(*[f.ctypes.data for f in fms])
    nm = np.zeros(max(cnt, 1), np.int32)
    _chk(lib().orbfe_search_projection_batch(device, C.addressof(arr), cnt, qp, fp, _p(nm)), "orbfe_search_projection_batch")
    return [(int(nm[k]), qms[k][:arr[k].nq], fms[k][:arr[k].n]) for k in range(cnt)]


def search_projection_last_sweeps():
    return int(lib().orbfe_search_projection_last_sweeps())


def distinctive_descriptors(pool, offsets, device=0):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:355-420) for many map points at once."""
    pool = np.ascontiguousarray(pool, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    best = np.zeros(len(offsets) - 1, np.int32)
    _chk(lib().orbfe_distinctive_descriptors(device, _p(pool), _p(offsets), len(best), _p(best)),
         "orbfe_distinctive_descriptors")
    return best


class Vocabulary:
    """DBoW2 vocabulary tree resident on the device (TemplatedVocabulary.h); vocab = the CSR dict of
    synth.make_vocabulary / an ORBvoc.txt loader."""

    def __init__(self, vocab, device=0):
        self.L = lib()
        self._keep = {k: np.ascontiguousarray(vocab[k]) for k in ("desc", "child_off", "child_ids", "word", "weight")}
        k = self._keep
        v = _Vocab(len(k["word"]), k["desc"].ctypes.data, k["child_off"].ctypes.data, k["child_ids"].ctypes.data,
                   k["word"].ctypes.data, k["weight"].ctypes.data, int(vocab["L"]))
        h = C.c_void_p()
        _chk(self.L.orbfe_vocab_upload(C.byref(h), device, C.byref(v)), "orbfe_vocab_upload")
        self.h = h

    @classmethod
    def from_text_file(cls, path, device=0):
        """TemplatedVocabulary::loadFromTextFile (ORBvoc.txt format) straight onto the device."""
        self = cls.__new__(cls)
        self.L = lib()
        self._keep = {}
        h = C.c_void_p()
        k, L_, nw = C.c_int(), C.c_int(), C.c_int()
        _chk(self.L.orbfe_vocab_load_text(C.byref(h), device, os.fsencode(path), C.byref(k), C.byref(L_), C.byref(nw)),
             "orbfe_vocab_load_text")
        self.h, self.k, self.levels, self.nwords = h, k.value, L_.value, nw.value
        return self

    def set_types(self, weighting, scoring):
        """WeightingType (0 TF_IDF, 1 TF, 2 IDF, 3 BINARY) / ScoringType (0 L1_NORM .. 5 DOT_PRODUCT), BowVector.h:39-56."""
        _chk(self.L.orbfe_vocab_set_types(self.h, int(weighting), int(scoring)), "orbfe_vocab_set_types")

    def get_types(self):
        w, sc = C.c_int(), C.c_int()
        _chk(self.L.orbfe_vocab_get_types(self.h, C.byref(w), C.byref(sc)), "orbfe_vocab_get_types")
        return w.value, sc.value

    def transform(self, feats, levelsup=4):
        """Per feature: (word id, node id `levelsup` levels above the leaves, weight)."""
        feats = np.ascontiguousarray(feats, np.uint8).reshape(-1, 32)
        n = len(feats)
        w = np.zeros(n, np.int32)
        nid = np.zeros(n, np.int32)
        wt = np.zeros(n, np.float64)
        _chk(self.L.orbfe_vocab_transform(self.h, _p(feats), n, levelsup, _p(w), _p(nid), _p(wt)), "orbfe_vocab_transform")
        return w, nid, wt

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_vocab_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class _BowView(C.Structure):
    _fields_ = [("n_kept", C.c_int), ("nn", C.c_int), ("nw", C.c_int), ("max_node", C.c_int), ("node_ids", C.c_void_p),
                ("offsets", C.c_void_p), ("indices", C.c_void_p), ("word_ids", C.c_void_p), ("word_values", C.c_void_p),
                ("d_header", C.c_void_p)]


class Bow:
    """orbfe_bow_*: Frame::ComputeBoW / KeyFrame::ComputeBoW on the device (TemplatedVocabulary::transform with both maps;
    src/Frame.cc:724-731).  compute() is asynchronous; host() is the host copy on request; the object itself may be passed as
    the `fv` of a search (the vector is then read on the device)."""

    def __init__(self, vocabulary, cap):
        self.L = lib()
        self.vocabulary = vocabulary  # (keeps the tree alive)
        self.h = C.c_void_p()
        self.cap = int(cap)
        self.L.orbfe_bow_fv.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orbfe_bow_host.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orbfe_bow_device.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orbfe_bow_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.orbfe_bow_destroy.argtypes = [C.c_void_p]
        self.L.orbfe_bow_destroy.restype = None
        self.L.orbfe_compute_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _chk(self.L.orbfe_bow_create(C.byref(self.h), vocabulary.h, self.cap), "orbfe_bow_create")

    def compute(self, desc, levelsup=4):
        """desc: host array [n, 32] or an (address, rows) pair naming device memory."""
        p, n, keep = _desc_arg(desc)
        _chk(self.L.orbfe_compute_bow(self.h, C.c_void_p(p), n, levelsup), "orbfe_compute_bow")
        return self

    def set_lazy_norm(self, on=True):
        """orbfe_bow_set_lazy_norm: BowVector::normalize in the host view instead of in the kernel."""
        self.L.orbfe_bow_set_lazy_norm.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.orbfe_bow_set_lazy_norm(self.h, int(on)), "orbfe_bow_set_lazy_norm")

    def host(self):
        """((word_ids, values), (node_ids, offsets, indices)) -- copies of the handle's page-locked host view."""
        v = _BowView()
        _chk(self.L.orbfe_bow_host(self.h, C.byref(v)), "orbfe_bow_host")

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        bow = (arr(v.word_ids, v.nw, np.uint32), arr(v.word_values, v.nw, np.float64))
        offsets = arr(v.offsets, v.nn + 1, np.int32)
        fv = (arr(v.node_ids, v.nn, np.uint32), offsets, arr(v.indices, v.n_kept, np.int32))
        self.last_counts = (v.n_kept, v.nn, v.nw, v.max_node)
        return bow, fv

    def device(self):
        v = _BowView()
        _chk(self.L.orbfe_bow_device(self.h, C.byref(v)), "orbfe_bow_device")
        return v

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_bow_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bow_from_transform(word, node, weight):
    """BowVector / FeatureVector as TemplatedVocabulary::transform builds them (:1147-1160, TF_IDF + L1):
    returns (bow dict word -> normalised weight, CSR feature vector (node_ids, offsets, indices))."""
    bow = {}
    keep = weight > 0  # w > 0: not a stop word
    for w_id, w in zip(word[keep].tolist(), weight[keep].tolist()):
        bow[w_id] = bow.get(w_id, 0.0) + w
    norm = sum(abs(v) for v in (bow[k] for k in sorted(bow)))
    if norm > 0:
        bow = {k: bow[k] / norm for k in sorted(bow)}
    idx = np.nonzero(keep)[0].astype(np.int32)
    order = np.argsort(node[idx], kind="stable")
    nodes_sorted = node[idx][order]
    node_ids, counts = np.unique(nodes_sorted, return_counts=True)
    offsets = np.zeros(len(node_ids) + 1, np.int32)
    offsets[1:] = np.cumsum(counts)
    return bow, (node_ids.astype(np.uint32), offsets, idx[order])


def kb8_unproject(params8, uv, device=0):
    """KannalaBrandt8::unproject (src/CameraModels/KannalaBrandt8.cpp:96-123)."""
    P = np.ascontiguousarray(params8, np.float32)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    rays = np.zeros((len(uv), 3), np.float32)
    _chk(lib().orbfe_kb8_unproject(device, _p(P), _p(uv), len(uv), _p(rays)), "orbfe_kb8_unproject")
    return rays


# ---------------------------------------------------------------------------------------------------------------------
# Multi-GPU / multi-camera path (include/orbfe_mc.h): thin ctypes layer over the C ABI.
def mc_layout(frames_per_rank, cap):
    """(desc_bytes, count_off, slab_bytes) of one rank's slab."""
    lay = _McLayout()
    _chk(lib().orbfe_mc_layout(frames_per_rank, cap, C.byref(lay)), "orbfe_mc_layout")
    return int(lay.desc_bytes), int(lay.count_off), int(lay.slab_bytes)


def mc_shard(nframes, world, rank):
    first, count = C.c_int(), C.c_int()
    _chk(lib().orbfe_mc_shard(nframes, world, rank, C.byref(first), C.byref(count)), "orbfe_mc_shard")
    return first.value, count.value


def mc_ring_pairs(world, frames_per_rank, rank, hops=(1,)):
    hops = np.ascontiguousarray(hops, np.int32)
    pairs = np.zeros((len(hops) * frames_per_rank, 2), np.int32)
    n = _chk(lib().orbfe_mc_ring_pairs(world, frames_per_rank, rank, _p(hops), len(hops), _p(pairs)), "orbfe_mc_ring_pairs")
    return [(int(q), int(g)) for q, g in pairs[:n]]


def mc_job_offsets(frames_per_rank, cap, pairs):
    pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    off = np.zeros((len(pr), 4), np.int64)
    _chk(lib().orbfe_mc_job_offsets(frames_per_rank, cap, _p(pr), len(pr), _p(off)), "orbfe_mc_job_offsets")
    return off


def mc_unique_id(transport=MC_RCCL):
    buf = (C.c_uint8 * MC_ID_BYTES)()
    _chk(lib().orbfe_mc_unique_id(transport, buf), "orbfe_mc_unique_id")
    return bytes(buf)


class MultiCam:
    """orbfe_mc handle: this rank's part of the sharded extraction + the all-gather + the ring matching.  `extractor`
    None (host transport only) gives a host-memory handle for the bookkeeping tests."""

    def __init__(self, extractor, uid, rank, world, frames_per_rank, cap, transport=MC_RCCL):
        self.L = lib()
        self.h = C.c_void_p()
        self.ex = extractor
        self.rank, self.world, self.frames, self.cap, self.transport = rank, world, frames_per_rank, cap, transport
        self.desc_bytes, self.count_off, self.slab_bytes = mc_layout(frames_per_rank, cap)
        idbuf = (C.c_uint8 * MC_ID_BYTES).from_buffer_copy(uid.ljust(MC_ID_BYTES, b"\0")) if uid is not None else None
        _chk(self.L.orbfe_mc_create(C.byref(self.h), extractor.h if extractor is not None else None, idbuf, rank, world,
                                    frames_per_rank, cap, transport), "orbfe_mc_create")

    def close(self):
        if self.h:
            self.L.orbfe_mc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, d_imgs_ptr, rows, cols, pitch, img_stride, lap=(0, 0)):
        _chk(self.L.orbfe_mc_extract_exchange_submit(self.h, C.c_void_p(d_imgs_ptr), rows, cols, pitch, img_stride,
                                                     int(lap[0]), int(lap[1])), "orbfe_mc_extract_exchange_submit")

    def wait(self):
        v = _McView()
        _chk(self.L.orbfe_mc_extract_exchange_wait(self.h, C.byref(v)), "orbfe_mc_extract_exchange_wait")
        return v

    def match_ring(self, hops=(1,), download=True):
        hops = np.ascontiguousarray(hops, np.int32)
        npairs = len(hops) * self.frames
        if not download:
            return _chk(self.L.orbfe_mc_match_ring(self.h, _p(hops), len(hops), None, None), "orbfe_mc_match_ring")
        idx = np.empty((npairs, self.cap, 2), np.int32)
        dist = np.empty((npairs, self.cap, 2), np.int32)
        _chk(self.L.orbfe_mc_match_ring(self.h, _p(hops), len(hops), _p(idx), _p(dist)), "orbfe_mc_match_ring")
        return idx, dist

    def match_ring_async(self, batch, hops=(1,)):
        hops = np.ascontiguousarray(hops, np.int32)
        return _chk(self.L.orbfe_mc_match_ring_async(self.h, _p(hops), len(hops), int(batch)), "orbfe_mc_match_ring_async")

    def match_outputs(self):
        di, dd, n = C.c_void_p(), C.c_void_p(), C.c_int()
        _chk(self.L.orbfe_mc_match_outputs(self.h, C.byref(di), C.byref(dd), C.byref(n)), "orbfe_mc_match_outputs")
        return di.value, dd.value, n.value

    def exchange_host(self, slab):
        """Host-memory exchange (collective): returns a read-only numpy view [world, slab_bytes] valid until the next call."""
        slab = np.ascontiguousarray(slab, np.uint8)
        assert slab.size == self.slab_bytes
        g = C.c_void_p()
        _chk(self.L.orbfe_mc_exchange_host(self.h, _p(slab), C.byref(g)), "orbfe_mc_exchange_host")
        buf = (C.c_uint8 * (self.world * self.slab_bytes)).from_address(g.value)
        return np.frombuffer(buf, np.uint8).reshape(self.world, self.slab_bytes)
