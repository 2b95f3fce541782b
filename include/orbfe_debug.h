/*
 * orbfe_debug.h -- test and diagnostic entry points of liborbfe.so, kept apart from the drop-in boundary (include/orbfe.h).
 * Nothing here is part of what an ORB-SLAM3 integration binds: the parity tests read intermediate state of the last call
 * through these (candidates and quadtree output per level, the fused kernel's blurred patch, the trig table's values), and
 * the trig-cache file checks run without a device.
 */
#ifndef ORBFE_DEBUG_H
#define ORBFE_DEBUG_H

#include "orbfe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Stage taps for parity tests: state of the last call. Packed entry = x | y<<12 | response<<24
 * with x,y relative to (minBorderX, minBorderY) = (16,16) of the level. */
int orbfe_debug_candidates(orbfe_ctx*, int img_index, int level, uint32_t* out, int cap);
int orbfe_debug_level_keypoints(orbfe_ctx*, int img_index, int level, uint32_t* out, int cap);
/* The 37 x 37 bytes of GaussianBlur's output (src/ORBextractor.cc:1114-1115) around keypoint `kp_index` (output
 * order) of image `img` of the last call: the fused kernel's blurred patch, for a direct comparison. */
int orbfe_debug_blurred_patch(orbfe_ctx*, int img, int kp_index, uint8_t* out37x37);
int orbfe_debug_fixups(orbfe_ctx*); /* keypoints re-evaluated with host libm trig in the last call */
/* (cos, sin) the descriptor kernel uses for the given keypoint angles (degrees) in the context's trig mode;
 * returns 2 when the table of libm values was used, 1 for the compact code table, 0 for none (ORBFE_TRIG_CR, or
 * no table), < 0 on error. */
int orbfe_debug_trig(orbfe_ctx*, const float* angles_deg, int n, float* a_out, float* b_out);
/* The cache file of the libm table (65 MB of 4-bit codes; default directory /dev/shm, ORBFE_TRIG_CACHE=<dir> moves it, =0
 * disables it), host side only -- these need no device.  The file is trusted only when it is a regular file (symbolic links
 * are not followed) owned by the calling user, not writable by group or others, of the expected size, with the expected
 * magic / angle range / libm fingerprint and a matching checksum over its WHOLE payload (the library verifies that checksum
 * on the device after the upload; `_check` runs the same tests on the host and names the first one that fails).
 * `_path`: the file this process would use (returns its length, 0 when disabled).  `_write`: a file in the library's
 * format around `payload` (`_payload_bytes()` bytes), created exclusively with mode 0600 under a temporary name and
 * renamed. */
int orbfe_debug_trig_cache_path(char* out, int cap);
size_t orbfe_debug_trig_cache_payload_bytes(void);
int orbfe_debug_trig_cache_write(const char* path, const uint8_t* payload, size_t bytes);
int orbfe_debug_trig_cache_check(const char* path, const char** why);

/* Host-only exerciser of the table behind orbfe_keyframe / orbfe_frame use counts (deferred destroy, stale handles refused) for
 * the sanitizer builds, which have no device to create real handles on: returns the number of protocol violations (0). */
int orbfe_debug_handle_table_selftest(int threads, int slots, int rounds);

#ifdef __cplusplus
}
#endif

#endif
