"""MI355X-native ORB front-end for ORB-SLAM3 (extractor + Hamming matcher hot path).

The product is the HIP library `liborbfe.so` behind the C ABI in include/orbfe.h; this
package is the thin Python binding used by the tests and by bench.py.  There is no CPU
fallback: importing the binding without the built library, or calling it without a
HIP device, fails loudly.
"""
from . import synth  # noqa: F401
from .binding import (KP_DTYPE, ORBextractor, ProjectionFrame, OrbfeError, Vocabulary, Bow, bfknn2, bow_from_transform, distinctive_descriptors, search_projection, search_projection_batch, search_projection_last_sweeps, compute_stereo_matches, hamming_pairs, kb8_unproject, lib, lib_path,  # noqa: F401
                      search_bow, search_bow_batch, search_bow_keyframes, search_tri_batch, KeyFrameHandle, search_triangulation, search_triangulation_kb8, search_triangulation_3d, kb8_triangulate, search_initialization, stereo_fisheye_matches)
