/*
 * orbfe.h -- C ABI of the MI355X-native ORB front-end (liborbfe.so).
 *
 * Drop-in boundary for the hot path of ORB-SLAM3's feature front-end
 * (reference = Taeyoung96/ORB_SLAM3_detailed_comments_KOR, paths below are relative to it):
 *
 *   extractor : ORB_SLAM3::ORBextractor              include/ORBextractor.h:43-107
 *               ctor                                   src/ORBextractor.cc:408-468
 *               operator()                             src/ORBextractor.cc:1068-1150
 *               (called only by Frame::ExtractORB      src/Frame.cc:413-420)
 *   matcher   : ORBmatcher::DescriptorDistance         src/ORBmatcher.cc:2591-2607
 *               ORBmatcher::SearchByBoW (KF,F)         src/ORBmatcher.cc:269-471
 *               ORBmatcher::SearchByBoW (KF,KF)        src/ORBmatcher.cc:823-963
 *               ORBmatcher::SearchForTriangulation_    src/ORBmatcher.cc:1208-1449
 *               Frame::ComputeStereoFishEyeMatches     src/Frame.cc:1119-1159 (BFMatcher knn-2)
 *               KannalaBrandt8::unproject              src/CameraModels/KannalaBrandt8.cpp:96-123
 *
 * Plain C, plain pointers and sizes, no OpenCV / torch types.  adapters/ORBextractor.h wraps
 * these entry points back into the reference's C++ signatures.
 *
 * Conventions: return >= 0 on success, negative on error; nothing throws or aborts.
 *   -1                 empty image (same value ORBextractor::operator() returns, :1072-1073)
 *   ORBFE_ERR_ARGS     bad argument / capacity too small
 *   ORBFE_ERR_NODEV    no usable HIP device (the library never falls back to the CPU)
 *   ORBFE_ERR_IMAGE_SMALL / _LARGE / ORBFE_ERR_NFEATURES   the documented limits of the extractor (below)
 *   <= -1000           -(1000 + hipError_t)
 * Thread-safety: distinct contexts may be used concurrently; one context = one caller at a time
 * (same rule as an ORBextractor instance, which owns mvImagePyramid).
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_ERR_ARGS (-2)
#define ORBFE_ERR_NODEV (-3)
#define ORBFE_ERR_STATE (-4)
/* The extractor's documented limits, each with its own code (orbfe_max_keypoints and every extraction call report them;
 * orbfe_error_string has the sentence):
 *   ORBFE_ERR_IMAGE_SMALL  some pyramid level is narrower or lower than 32 + 35 px: the reference's cell grid has no cell
 *                          there and divides by zero (src/ORBextractor.cc:779-782)
 *   ORBFE_ERR_IMAGE_LARGE  an image side above 4096 px (the packed candidate format holds 12-bit coordinates)
 *   ORBFE_ERR_NFEATURES    nfeatures so large that an image would need more than 65535 keypoint slots (round 4: levels whose
 *                          quadtree tables exceed a workgroup's LDS -- nfeatures above ~7800 at 8 levels / 1.2, such as the
 *                          5 x nFeatures initialisation extractor of src/Tracking.cc:1157 -- run on a global-memory table) */
#define ORBFE_ERR_IMAGE_SMALL (-5)
#define ORBFE_ERR_IMAGE_LARGE (-6)
#define ORBFE_ERR_NFEATURES (-7)
/* One sentence for any code this library returns (static storage; never NULL). */
const char* orbfe_error_string(int code);

typedef struct orbfe_ctx orbfe_ctx; /* one per ORBextractor instance */

/* Bit-compatible with cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id), 28 bytes. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbfe_kp;

/* ---- extractor: replaces ORBextractor::ORBextractor (src/ORBextractor.cc:408-468) ---- */
int orbfe_create(orbfe_ctx** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int device /* HIP device ordinal */);
void orbfe_destroy(orbfe_ctx*);
/* Frees what the library keeps per process and device beyond the contexts: the libm trig table (1.03 GB or 65 MB,
 * see orbfe_set_trig_mode).  It is rebuilt by the next ORBFE_TRIG_LIBM extraction.  No extraction may be in flight
 * on that device. */
int orbfe_release_caches(int device);

/* Run on an existing hipStream_t (e.g. the caller's torch stream); NULL = context-owned stream. */
int orbfe_set_stream(orbfe_ctx*, void* hip_stream);
/* The stream (hipStream_t) the context's kernels run on and its device ordinal: what a caller orders its own work against. */
int orbfe_get_stream(orbfe_ctx*, void** hip_stream, int* device);
/* 7 fixed-point (8.8) Gaussian taps; default {18,34,48,56,48,34,18} = OpenCV >= 4.1.2. */
int orbfe_set_gaussian_taps(orbfe_ctx*, const int* taps7);
/* Rotation trig of the descriptor (src/ORBextractor.cc:110-111):
 *   ORBFE_TRIG_LIBM (default): bit-identical to this host's libm cosf/sinf.  The first extraction of a
 *       process evaluates libm for every float angle in [2^-7, 360] degrees (all cores, a fraction of a
 *       second) and keeps the values in HBM (1.03 GB, the descriptor kernel just looks its angle up); every
 *       later batch runs without any host involvement.  ORBFE_TRIG_TABLE=1 in the environment (or a failed
 *       allocation) selects the compact form instead: 65 MB of 4-bit codes saying how libm differs from the
 *       device's correctly rounded value, which the kernel then still evaluates; ORBFE_TRIG_TABLE=0, or a libm
 *       that cannot be tabulated, makes the mode behave like ORBFE_TRIG_LIBM_HOSTCHECK.
 *   ORBFE_TRIG_LIBM_HOSTCHECK: same results, no table: the device flags keypoints whose sampling grid could
 *       change under a 1-ulp difference of sin/cos, the host evaluates libm for those after the batch (one
 *       stream synchronisation per call) and they are re-evaluated on the device with the libm values.
 *   ORBFE_TRIG_CR: correctly rounded sin/cos only (host independent). */
#define ORBFE_TRIG_LIBM 0
#define ORBFE_TRIG_CR 1
#define ORBFE_TRIG_LIBM_HOSTCHECK 2
int orbfe_set_trig_mode(orbfe_ctx*, int mode);

/* cv::fastAtan2's polynomial (src/ORBextractor.cc:101): seven separately rounded operations (default, the generic
 * C++ path) or, with on = 1, its three inner Horner steps as fused multiply-adds -- what an OpenCV whose AVX2
 * translation unit was built with -mfma evaluates.  ORBFE_ATAN_FMA=1 in the environment sets the
 * default of new contexts. */
int orbfe_set_atan_fma(orbfe_ctx*, int on);

/* Fisheye rigs (KannalaBrandt8, src/CameraModels/KannalaBrandt8.cpp:96-123): when params8 = {fx,fy,cx,cy,k0..k3}
 * is set, the extractor also unprojects every keypoint to its bearing ray in the output-packing kernel
 * (no extra pass).  Rays are indexed like the keypoints: 3 floats each.  NULL disables.
 * Host API: read them with orbfe_get_rays(); device API: orbfe_set_ray_output() gives the device buffer
 * (nimg * cap_per_img * 3 floats) the next device calls write to (NULL = context-owned buffer). */
int orbfe_set_kb8(orbfe_ctx*, const float* params8);
int orbfe_set_ray_output(orbfe_ctx*, float* d_rays);
int orbfe_get_rays(orbfe_ctx*, int img_index, int cap_per_img, float* rays, int n);

/* Upper bound of keypoints one image can yield with this context's parameters for a rows x cols image
 * (sum over levels of max(N_l + 3, 4 * nIni_l)); `cap` arguments below must be >= this. */
int orbfe_max_keypoints(orbfe_ctx*, int rows, int cols);

/* Replaces ORBextractor::operator() (src/ORBextractor.cc:1068-1150).  Host pointers.
 * Returns monoIndex (>= 0), -1 for an empty image; *n_out = number of keypoints (rows of desc).
 * lap0/lap1 = vLappingArea[0..1].  kps/desc must hold `cap` entries (cap*28 and cap*32 bytes).
 * (Waiting: the blocking calls of a frame or two -- this one, orbfe_extract_batch with <= 2 images,
 * orbfe_extract_stereo_pair, orbfe_compute_stereo_matches_resident -- and the matcher calls whose results come back through
 * page-locked memory spin, for a bounded time, on a completion word the last kernel publishes there instead of blocking in
 * hipStreamSynchronize; ORBFE_SPIN=0 in the environment restores the latter.  INTEGRATION.md.) */
int orbfe_extract(orbfe_ctx*, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1,
                  orbfe_kp* kps, uint8_t* desc, int cap, int* n_out);

/* Batched operator(): nimg images of identical size, one call (multi-camera / multi-frame path).
 * kps/desc are nimg slabs of cap_per_img entries; n_out[i], mono_out[i] per image. */
int orbfe_extract_batch(orbfe_ctx*, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride,
                        const int* lap /* 2*nimg or NULL (= {0,0}) */, orbfe_kp* kps, uint8_t* desc, int cap_per_img,
                        int* n_out, int* mono_out);

/* Images of DIFFERENT sizes in one call (a rig with unequal cameras): rows[i] / cols[i] / strides[i] per image.  The
 * images are grouped by size and each group runs as one batch; the context keeps the tables of the sizes it has seen
 * (up to eight), so alternating sizes rebuild nothing.  cap_per_img >= orbfe_max_keypoints of every size.  Returns 0,
 * or the error of the first group that failed (its images get n_out = 0; an empty image makes its group return -1). */
int orbfe_extract_batch_sizes(orbfe_ctx*, int nimg, const uint8_t* const* imgs, const int* rows, const int* cols,
                              const size_t* strides, const int* lap /* 2*nimg or NULL */, orbfe_kp* kps, uint8_t* desc,
                              int cap_per_img, int* n_out, int* mono_out);

/* The same call split in two, for callers that keep the PCIe link and the GPU busy at once: submit queues the
 * transfer of the images, the kernels and the transfer of the results and returns; wait completes the OLDEST
 * submitted batch (its n_out / mono_out / kps / desc are valid afterwards, not before).  Up to two batches may be in
 * flight per context: the images of batch i+1 and the results of batch i-1 then cross the link while the kernels
 * of batch i run.  ORBFE_ERR_STATE when a third batch is submitted, or nothing is in flight.  The blocking calls
 * above may not be mixed with batches in flight.  All arrays (and the images) must stay valid until the wait. */
int orbfe_extract_batch_submit(orbfe_ctx*, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride,
                               const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out);
int orbfe_extract_batch_wait(orbfe_ctx*);

/* Page-locked host memory.  A DMA engine moves pinned memory at the full PCIe rate with one command; pageable
 * memory is staged (by this library, with a few host threads) at a fraction of it.  Every host-pointer entry point
 * of the extractor recognises images and output arrays that are pinned -- allocated here, registered here, or
 * pinned by someone else (hipHostMalloc, torch pin_memory) -- and takes the direct path for them; rows of the
 * output arrays beyond n_out[i] are then unspecified.  Registering pins an existing buffer (e.g. a camera driver's
 * ring); it is expensive (page-locking), so do it once per buffer, not per frame. */
void* orbfe_host_alloc(size_t bytes);
void orbfe_host_free(void*);
int orbfe_host_register(void* p, size_t bytes);
/* For callers that cannot be changed (Frame::ExtractORB hands over whatever cv::Mat it was given) but whose image buffers
 * are long-lived -- a camera driver's ring, a preallocated cv::Mat that every frame is copied or decoded into: with
 * on != 0 the host-pointer calls page-lock a PAGEABLE image buffer themselves the second time they see the same address and
 * size, and from then on it takes the DMA path like memory from orbfe_host_alloc (at most 16 registrations per context,
 * least recently used out, all released by orbfe_set_auto_register(ctx, 0) / orbfe_destroy).  Off by default: a buffer
 * that is freed while registered must be unregistered first, which only the owner can know -- switch it on when the
 * buffers outlive the context, as they do in the drivers above. */
int orbfe_set_auto_register(orbfe_ctx*, int on);
int orbfe_host_unregister(void* p);

/* Same, with every buffer already resident in device memory (no PCIe traffic).  Asynchronous on the
 * context's stream unless the trig mode needs the host (ORBFE_TRIG_LIBM_HOSTCHECK synchronises once per call).
 * d_imgs: nimg images, image i at d_imgs + i*img_stride_bytes, row pitch `pitch`. */
int orbfe_extract_batch_device(orbfe_ctx*, int nimg, const uint8_t* d_imgs, int rows, int cols, size_t pitch,
                               size_t img_stride_bytes, int lap0, int lap1, orbfe_kp* d_kps, uint8_t* d_desc,
                               int cap_per_img, int32_t* d_n_out, int32_t* d_mono_out);
/* Lanes (the reference's own concurrency on this path is two extractors on two threads, src/Frame.cc:119-122; a rig or a
 * multi-GPU shard has more).  orbfe_set_lanes(ctx, n) with n = 2..ORBFE_MAX_LANES (or ORBFE_LANES=n in the environment when the
 * context is created; default 1) lets the context keep up to n device-pointer batches in flight on streams it owns.
 *
 * Every orbfe_extract_batch_device call goes, WHOLE, to the next lane round-robin.
 * Each lane has its own intermediate buffers (pyramids, candidates, quadtree keys, status header), so the kernels of
 * DIFFERENT batches overlap: the latency-bound chain of a small batch (8 x 1280x720 is four kernels of 15-30 us each on a
 * nearly empty chip) runs beside the chains of its neighbours -- 0.084 ms per batch with one lane, 0.05 with two, 0.043-0.045
 * with three or four.  The context's stream carries no kernels in this mode, only ordering:
 *   - a lane waits for the point of the call on the context's stream: images written on that stream before the call are
 *     complete when the lane reads them;
 *   - the context's stream waits for the lane's pyramid kernel, the only reader of the caller's images: a caller may refill
 *     the SAME image buffer on the context's stream right after the call returns (stream order, exactly as with one lane);
 *   - OUTPUTS are not ordered on the context's stream until a join (below), and calls that are in flight together must be
 *     given different output arrays (n lanes: a ring of n output sets; the call n calls ago on the same lane has finished
 *     before the lane writes again -- a lane is a stream).
 * (Round 4's other form -- one call as two half-batches, ORBFE_LANES_SPLIT -- was measured slower and left with round 6.)
 *
 * Results are bit-identical with any number of lanes.  What changes is ORDERING: after a call the outputs are NOT yet ordered on the
 * context's stream.  They are after any of
 *   orbfe_lanes_join(ctx)            -- the context's stream waits for every lane (no host wait; a few us on the stream),
 *   orbfe_get_device_outputs(ctx, ...) -- joins, then marks the stream (matcher calls of this library order themselves after it),
 *   orbfe_sync(ctx)                  -- waits for all lanes on the host,
 * and every other entry point of this header that reads the context's results or buffers joins by itself (orbfe_get_level,
 * the stereo-matching calls, the host-pointer extract calls, orbfe_mc_*; "the last call" they refer to is the last call
 * whatever lane it ran on).  A caller that launches its OWN kernels on orbfe_get_stream()'s stream right behind
 * orbfe_extract_batch_device must call orbfe_lanes_join first -- which is why lanes are opt-in.  Host-pointer calls use one
 * lane (orbfe_extract_stereo_pair_submit keeps several stereo frames in flight by itself). */
#define ORBFE_MAX_LANES 4
#define ORBFE_LANES_BATCH 0
int orbfe_set_lanes(orbfe_ctx*, int lanes /* 1 .. ORBFE_MAX_LANES */);
int orbfe_set_lane_mode(orbfe_ctx*, int mode /* ORBFE_LANES_BATCH: the one mode left (any other value: ORBFE_ERR_ARGS) */);
int orbfe_lanes_join(orbfe_ctx*);
/* Batch lanes, the input guard (on by default): after every call the context's stream waits for the lane's pyramid kernel, so
 * that the caller may overwrite the call's images in stream order.  That wait chains the pyramid kernels of consecutive calls
 * behind each other through two event hops (8 x 1280x720, three lanes: 0.055 ms per batch with the guard, 0.045 without).  A
 * caller that never rewrites the images of a call that may still be in flight -- a ring of at least `lanes` + 1 input buffers,
 * or images that stay resident -- switches it off: the images then belong to the library, like the outputs, until a join, an
 * orbfe_sync, or until `lanes` later calls have been made and joined. */
int orbfe_set_lane_input_guard(orbfe_ctx*, int on);
/* For a consumer on a stream of its own that must not hold the context's stream back: records `hip_event` (a hipEvent_t)
 * behind the LAST call's work on the lane that holds it, when the context's stream has not been ordered after that lane, and
 * returns 1 (0: nothing pending, the event was not touched).  The consumer waits for this event AND for one it records on
 * orbfe_get_stream()'s stream (what orbfe_mc_extract_exchange_submit does for its collective). */
int orbfe_lanes_record(orbfe_ctx*, void* hip_event);
/* Waits for the context's stream and returns ORBFE_ERR_STATE when a kernel of the finished work raised the device
 * error word (a quadtree list overflow, which the bounds of SURVEY.md A.9 rule out): the asynchronous call above
 * cannot report it itself. */
int orbfe_sync(orbfe_ctx*);

/* Scale getters (include/ORBextractor.h:61-81) and mvImagePyramid (:83). */
int orbfe_get_levels(orbfe_ctx*);
float orbfe_get_scale_factor(orbfe_ctx*);
void orbfe_get_scale_tables(orbfe_ctx*, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2);
void orbfe_get_features_per_level(orbfe_ctx*, int* n_per_level);
/* Copies level `level` of image `img_index` of the last call, with its 19-px REFLECT_101 frame, to
 * dst (rows+38 by cols+38 of the level, row stride dst_stride).  Pass dst=NULL to query the size. */
int orbfe_get_level(orbfe_ctx*, int img_index, int level, uint8_t* dst, size_t dst_stride, int* rows, int* cols);

/* Per-stage device time measured with hipEvents on the context's stream.  Enabling resets the
 * statistics; with on = N >= 1 every N-th call then records one event set (ring of 256) -- the seven
 * event records cost ~25 us of a 64-frame batch, so a sampling period keeps them out of the throughput;
 * orbfe_profile_read() waits for the stream, writes the per-stage AVERAGE over the recorded calls and
 * returns their number. */
#define ORBFE_STAGE_PYRAMID 0
#define ORBFE_STAGE_FAST 1
#define ORBFE_STAGE_OCTREE 2
#define ORBFE_STAGE_PACK 3
#define ORBFE_STAGE_DESC 4
#define ORBFE_STAGE_TRIGFIX 5 /* host libm check of the flagged keypoints + fix-up launch (ORBFE_TRIG_LIBM_HOSTCHECK) */
#define ORBFE_STAGE_COUNT 6
int orbfe_profile_enable(orbfe_ctx*, int on);
int orbfe_profile_read(orbfe_ctx*, float* ms_per_stage /* ORBFE_STAGE_COUNT */);

/* (The stage taps of the parity tests and the trig-cache file checks -- orbfe_debug_* -- are declared in include/orbfe_debug.h.) */

/* Frame::ComputeStereoMatches (src/Frame.cc:797-967), rectified stereo.  `left` / `right` are the contexts
 * that just extracted the two images (same size and parameters, same device): their pyramids are read in
 * place for the 11x11 SAD refinement, so mvImagePyramid never leaves the device.  uRight / depth hold one
 * float per left keypoint (-1 = no match, like mvuRight / mvDepth).  Returns the number of matches. */
int orbfe_compute_stereo_matches(orbfe_ctx* left, orbfe_ctx* right, const orbfe_kp* kpsL, const uint8_t* descL, int nL,
                                 const orbfe_kp* kpsR, const uint8_t* descR, int nR, float mb, float mbf, float* uRight,
                                 float* depth);

/* The same on what the two contexts' LAST extraction left on the device (image imgL of the left context's batch
 * against image imgR of the right one): keypoints, descriptors, counts and both pyramids are read in place, so a
 * stereo frame costs two image uploads and one download of uRight / depth.  nL = entries of uRight / depth
 * (the left image's n_out). */
int orbfe_compute_stereo_matches_resident(orbfe_ctx* left, int imgL, orbfe_ctx* right, int imgR, float mb, float mbf,
                                          float* uRight, float* depth, int nL);

/* A rectified stereo frame in ONE call with ONE host wait: what Frame::Frame (stereo) does with two extractor threads
 * (src/Frame.cc:119-122) followed by ComputeStereoMatches (:797-967).  Both images go through the extractor as a batch of
 * two (kps / desc / n_out / mono_out as orbfe_extract_batch with nimg = 2: image 0 = left, 1 = right; lap = 4 ints or
 * NULL), the matching is queued behind the extraction on the same stream, and uRight / depth (cap_per_img floats each)
 * receive one value per LEFT keypoint (-1 = no match).  Returns the number of stereo matches, or a negative error. */
int orbfe_extract_stereo_pair(orbfe_ctx*, const uint8_t* imgL, const uint8_t* imgR, int rows, int cols, size_t stride,
                              const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out,
                              float mb, float mbf, float* uRight, float* depth);
/* Stereo frames IN FLIGHT (round 5): _submit queues the same work on the context's next lane (orbfe_set_lanes: its stream, its
 * pyramids, its pinned result slab and completion word) and returns; _wait completes the OLDEST submitted frame, fills the
 * arrays THAT _submit was given and returns its number of stereo matches.  Up to `lanes` frames are accepted before a _wait is
 * due (ORBFE_ERR_STATE beyond that); the images and every output array of a frame belong to the library until its _wait
 * returns.  A tracker that may run one frame behind -- or several trackers / cameras sharing a context -- gets 2-3 frames
 * per latency of one (0.10 ms per EuRoC frame with the blocking call, see DESIGN.md 7.5 for the in-flight rates).  Not to
 * be mixed with orbfe_extract_batch_submit while frames are in flight. */
int orbfe_extract_stereo_pair_submit(orbfe_ctx*, const uint8_t* imgL, const uint8_t* imgR, int rows, int cols, size_t stride,
                                     const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out,
                                     float mb, float mbf, float* uRight, float* depth);
int orbfe_extract_stereo_pair_wait(orbfe_ctx*);

/* ---- matcher ---- */
/* DescriptorDistance over all pairs: D[i*nB+j] = popcount(A_i xor B_j).  Host pointers. */
int orbfe_hamming_pairs(int device, const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D);
/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (src/Frame.cc:1137): idx/dist hold 2 entries per query,
 * ascending distance, ties -> lower train index; -1 when fewer than 1/2 train rows exist. */
int orbfe_bfknn2(int device, const uint8_t* Q, int nQ, const uint8_t* T, int nT, int32_t* idx, int32_t* dist);

/* Device-resident forms of the two calls above: every pointer is DEVICE memory (for instance the descriptor slab
 * an extractor context left in HBM, orbfe_get_device_outputs, or the slab an all-gather delivered), nothing is
 * copied and the call does not wait: the kernel is queued on `hip_stream` (NULL = the calling thread's matcher
 * stream; synchronise it with orbfe_matcher_sync). */
int orbfe_hamming_pairs_device(int device, void* hip_stream, const uint8_t* dA, int nA, const uint8_t* dB, int nB,
                               uint16_t* dD);
int orbfe_bfknn2_device(int device, void* hip_stream, const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int32_t* d_idx,
                        int32_t* d_dist);
/* Cross-camera matching (the consumer of the multi-GPU all-gather of descriptors): njobs independent
 * knn-2 problems in ONE launch.  A job names a query frame and a train frame by device pointers -- their descriptor
 * rows (32 B each) and the address of their keypoint count -- wherever they live: an extractor's resident output, this
 * rank's slab, the slab the all-gather delivered.  Results as orbfe_bfknn2 per job: d_idx / d_dist hold
 * njobs x cap x 2 entries (rows beyond a query frame's count are not written); cap >= every count. */
typedef struct {
    const uint8_t* q_desc; const int32_t* q_count;
    const uint8_t* t_desc; const int32_t* t_count;
} orbfe_knn2_job;
int orbfe_bfknn2_frames_device(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                               int32_t* d_idx, int32_t* d_dist);
/* Waits for the calling thread's matcher stream on `device`. */
int orbfe_matcher_sync(int device);
/* Where the LAST extraction of a context left its results in HBM (valid until the next call on that context):
 * nimg slabs of cap keypoints (28 B, orbfe_kp) / descriptors (32 B) and the per-image counts.  For the host-pointer
 * calls these are the context's own buffers; for orbfe_extract_batch_device the caller's. */
int orbfe_get_device_outputs(orbfe_ctx*, const orbfe_kp** d_kps, const uint8_t** d_desc, const int32_t** d_n, int* cap,
                             int* nimg);
/* ORDERING of resident pointers.  orbfe_extract_batch_device is asynchronous on the context's stream; the arrays above
 * may still be being written when this call returns them.  The call therefore marks that point on the context's stream,
 * and every matcher entry point of this library (the host-pointer forms that recognise device pointers, the *_device
 * forms, orbfe_frame_create, orbfe_search_bow_batch ...) that is later handed one of these pointers makes ITS stream wait
 * for the mark before it reads: extraction -> orbfe_get_device_outputs -> matcher call needs no orbfe_sync in between.
 * Call it again after every extraction (a new batch moves the mark).  A caller's OWN kernels or copies on the arrays
 * must be ordered by the caller: run them on the context's stream (orbfe_set_stream / orbfe_get_stream) or after
 * orbfe_sync.  Buffers the caller owns and never passed through this call (e.g. the d_desc given to
 * orbfe_extract_batch_device and then straight to a *_device matcher form on ANOTHER stream) are likewise the caller's
 * to order. */
/* In every matcher call below the DESCRIPTOR arrays (desc1 / desc2 / desc / qdesc / pool ...) may also be device
 * pointers: the call recognises them (hipPointerGetAttributes) and reads them in place instead of uploading. */

/* DBoW2::FeatureVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h:24-25) in CSR form. */
typedef struct {
    int nn;                   /* distinct nodes                      */
    const uint32_t* node_ids; /* ascending                           */
    const int32_t* offsets;   /* nn+1                                */
    const int32_t* indices;   /* offsets[nn] feature indices         */
} orbfe_fv;
/* nn == ORBFE_FV_RESIDENT: the vector lives in an orbfe_bow handle (orbfe_bow_fv below fills such an orbfe_fv); accepted by
 * every call that takes an orbfe_fv. */
#define ORBFE_FV_RESIDENT (-0x0B0F)

typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* mask1 /* 1 = good MapPoint */; const float* angle1;
    orbfe_fv fv1; int limit1 /* -1, or mvKeysUn.size() for the KF-KF variant */;
    const uint8_t* desc2; int n2; const uint8_t* mask2 /* KF-KF variant only */; const float* angle2;
    orbfe_fv fv2; int limit2;
    int Nleft;       /* KF-F variant: F.Nleft, -1 = monocular         */
    float nnratio;   /* mfNNratio                                     */
    int check_orientation;
    int variant;     /* 0 = SearchByBoW(KeyFrame*,Frame&), 1 = SearchByBoW(KeyFrame*,KeyFrame*) */
} orbfe_bow_args;
/* variant 0: match[n2] = index into set 1 (the KeyFrame) or -1; variant 1: match[n1] = index into set 2 or -1.
 * Returns nmatches. */
int orbfe_search_bow(int device, const orbfe_bow_args*, int32_t* match);
/* `count` independent SearchByBoW problems (relocalisation candidates, src/Tracking.cc:3784; covisible
 * keyframes of a loop candidate, src/LoopClosing.cc:725) in ONE upload / launch / download.
 * match[p] sized like the single call's output; nmatches[p] per problem.  Returns 0 or an error. */
int orbfe_search_bow_batch(int device, int count, const orbfe_bow_args* args, int32_t* const* match, int* nmatches);

/* ---- resident keyframes (round 4) ----
 * The BoW / triangulation searches run one CURRENT frame or keyframe against MANY keyframes that do not change between
 * calls: Tracking::Relocalization and LoopClosing search one frame against every candidate (src/Tracking.cc:3784,
 * src/LoopClosing.cc:725), LocalMapping::CreateNewMapPoints one keyframe against its 10-20 best covisible neighbours
 * (src/LocalMapping.cc:556-621).  The calls above re-pack and re-upload both sides every time.  A keyframe handle keeps one
 * side on the device: descriptors (`desc` may be a device pointer -- orbfe_get_device_outputs of the extractor that produced
 * the frame --, then nothing of it crosses PCIe), the per-feature flag (mask = "has a good MapPoint": SearchByBoW's
 * mask1 / mask2 and SearchForTriangulation_'s hasMP), angles, and the FeatureVector; with kp_xy / octave / uRight given it
 * can also be a side of the triangulation search.  Descriptors, keypoints and the FeatureVector of a KeyFrame never change
 * (ComputeBoW runs once); its MapPoints do: orbfe_keyframe_set_mask re-sends the n flag bytes when they differ from the last
 * ones.  A handle is read by the searches of any thread; create / set_mask / destroy are the owner's: set_mask rewrites the
 * handle's flags in place and must not run while a search of another thread that relies on them (one that passes no flags of
 * its own: mask1 / mask2 / hasMP == NULL) is in progress -- callers that search one keyframe from several threads send the flags
 * with every call instead, as adapters/ORBmatcher.h does.  orbfe_keyframe_destroy may be called at any time (round 6): a call of
 * another thread that was given the handle keeps it alive until it returns -- the last such call frees it --, and a call that is
 * handed a handle that has already been destroyed returns ORBFE_ERR_ARGS instead of touching it (every entry point takes a use
 * of its handles under a lock that knows which handles are alive).  Destroy does not wait for the device: a search has done all
 * its device reads before it returns.  The same holds for orbfe_frame / orbfe_frame_destroy and orbfe_bow / orbfe_bow_destroy. */
typedef struct orbfe_keyframe orbfe_keyframe;
typedef struct {
    const uint8_t* desc; int n;            /* 32-B rows; host or device pointer                                      */
    const uint8_t* mask;                   /* 1 = the feature has a (good) MapPoint                                  */
    const float* angle;                    /* mvKeysUn[i].angle, or NULL when no search checks the orientation       */
    const float* kp_xy;                    /* 2 floats per feature, or NULL: BoW searches only                        */
    const int32_t* octave; const float* uRight; /* required with kp_xy                                               */
    orbfe_fv fv;                           /* KeyFrame::mFeatVec                                                      */
} orbfe_keyframe_args;
int orbfe_keyframe_create(orbfe_keyframe** out, int device, const orbfe_keyframe_args*);
int orbfe_keyframe_set_mask(orbfe_keyframe*, const uint8_t* mask);
void orbfe_keyframe_destroy(orbfe_keyframe*);
/* orbfe_search_bow_batch with sets taken from handles: kf1[p] / kf2[p] non-null replaces set 1 / set 2 of args[p] (its
 * desc / n / angle / fv fields are then ignored; limit, Nleft, nnratio, check_orientation and variant still come from
 * args[p]).  The FLAGS of a set in a handle may still be given per call: args[p].mask1 (mask2, variant 1) non-null is used
 * instead of the handle's -- a KeyFrame's MapPoints change while Tracking, LocalMapping and LoopClosing search it from
 * three threads, so a caller like adapters/ORBmatcher.h sends the current flags with every call (n bytes) rather than
 * writing them into the shared handle.  kf1 / kf2 may be NULL (no handle on that side).  One upload of whatever is not
 * resident, one launch; small results are written by the kernel straight into pinned host memory (no download command). */
int orbfe_search_bow_keyframes(int device, int count, orbfe_keyframe* const* kf1, orbfe_keyframe* const* kf2,
                               const orbfe_bow_args* args, int32_t* const* match, int* nmatches);
/* SearchForTriangulation_ (src/ORBmatcher.cc:1208-1449, pinhole gate) of ONE keyframe against `count` neighbours in ONE
 * launch; both sides are handles created with kp_xy.  pair[p] holds what depends on the pair; pairs[p] (2 * n1 ints) and
 * npairs[p] are the outputs of the p-th orbfe_search_tri call.  Returns 0 or an error. */
typedef struct {
    float F12[9]; float ep[2];
    const float* scaleFactors2; const float* levelSigma2_2; int nlevels2;
    int only_stereo, coarse, check_orientation;
    const uint8_t* hasMP2; /* this call's has-MapPoint flags of the neighbour, or NULL = the handle's */
} orbfe_tri_pair;
int orbfe_search_tri_batch(orbfe_keyframe* kf1, const uint8_t* hasMP1 /* this call's flags of kf1, or NULL = the handle's */,
                           int count, orbfe_keyframe* const* kf2, const orbfe_tri_pair* pair, int32_t* const* pairs, int* npairs);

typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* hasMP1; const float* kp1_xy; const float* angle1;
    const int32_t* octave1; const float* uRight1; orbfe_fv fv1;
    const uint8_t* desc2; int n2; const uint8_t* hasMP2; const float* kp2_xy; const float* angle2;
    const int32_t* octave2; const float* uRight2; orbfe_fv fv2;
    float F12[9];    /* K1^-T [t12]x R12 K2^-1 as Pinhole::epipolarConstrain_ forms it (Pinhole.cpp:161-164) */
    float ep[2];     /* epipole in image 2 */
    const float* scaleFactors2; const float* levelSigma2_2; int nlevels2;
    int only_stereo, coarse, check_orientation;
} orbfe_tri_args;
/* SearchForTriangulation_ (monocular / rectified pinhole rig).  pairs[2*k], pairs[2*k+1]; returns npairs. */
int orbfe_search_tri(int device, const orbfe_tri_args*, int32_t* pairs /* 2*n1 */);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821, monocular bootstrap): every level-0 keypoint of
 * F1 searches a window of half-size window_size around vbPrevMatched[i1] in F2's grid (levels [0, 0]) for its best
 * and second-best Hamming distance; a feature of F2 already held with a distance <= the candidate's is skipped
 * (vMatchedDistance, :744) and a better match steals it (:765-772); TH_LOW, the ratio test and the orientation
 * histogram as in the reference.  matches12[n1] = index into F2 or -1 (vnMatches12); returns nmatches.  The caller
 * updates vbPrevMatched from the result (:813-816).  Frame side as orbfe_proj_args (grid parameters of F2). */
typedef struct {
    const uint8_t* desc1; int n1; const int32_t* octave1; const float* angle1;
    const float* prev_xy;                  /* vbPrevMatched, 2 floats per F1 keypoint */
    const uint8_t* desc2; int n2; const float* kx2; const float* ky2; const int32_t* octave2; const float* angle2;
    float minX, minY, gridWInv, gridHInv;  /* of F2 */
    int window_size; float nnratio; int check_orientation;
} orbfe_init_args;
int orbfe_search_initialization(int device, const orbfe_init_args*, int32_t* matches12);

/* SearchForTriangulation_ for KannalaBrandt8 cameras (src/ORBmatcher.cc:1208-1449 with the gate
 * KannalaBrandt8::epipolarConstrain_ = TriangulateMatches_ > 0.0001f, src/CameraModels/KannalaBrandt8.cpp:239-242,
 * :409-480): a monocular fisheye keyframe pair (Nleft1 == Nleft2 == -1) or a two-camera rig (features [0, Nleft)
 * from the left camera, the rest from the right one; kp*_xy = mvKeys ++ mvKeysRight; the four relative poses of
 * :1238-1248 in the order ll, lr, rl, rr, only the first without a rig; R12 row-major).  The gate is a
 * floating-point triangulation per candidate (Newton unprojection, 4x4 Jacobi SVD as cv::SVD::compute, atan2f /
 * cosf / sinf projection): its value agrees with the host's to ~1e-6 relative, so a decision can differ from
 * the reference's only for a candidate that close to one of its thresholds (parity by tolerance, not bit-exact). */
typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* hasMP1; const float* kp1_xy; const float* angle1;
    const int32_t* octave1; const float* uRight1 /* or NULL */; orbfe_fv fv1; int Nleft1;
    const uint8_t* desc2; int n2; const uint8_t* hasMP2; const float* kp2_xy; const float* angle2;
    const int32_t* octave2; const float* uRight2 /* or NULL */; orbfe_fv fv2; int Nleft2;
    const float* kb8_1L; const float* kb8_1R; const float* kb8_2L; const float* kb8_2R; /* fx,fy,cx,cy,k0..k3; R: rigs */
    const float* R12; const float* t12;   /* 4 x 9 and 4 x 3 (1 x 9, 1 x 3 without a rig) */
    float ep[2];                          /* epipole in image 2 (only read without a rig, :1325-1333) */
    const float* scaleFactors2; const float* levelSigma2_1; const float* levelSigma2_2; int nlevels1, nlevels2;
    int only_stereo, coarse, check_orientation;
} orbfe_tri_kb8_args;
int orbfe_search_tri_kb8(int device, const orbfe_tri_kb8_args*, int32_t* pairs /* 2*n1 */);
/* The SearchForTriangulation overload that also returns the triangulated points (src/ORBmatcher.cc:1452-1641;
 * declared in include/ORBmatcher.h:73-75, no caller in the reference).  Same walk over the shared vocabulary nodes
 * as SearchForTriangulation_, without the stereo and epipole gates (bOnlyStereo and F12 are not read by it); the
 * gate is pCamera1->matchAndtriangulate(kp1, kp2, pCamera2, Tcw1, Tcw2, sigma1, sigma2, x3D):
 * KannalaBrandt8's (src/CameraModels/KannalaBrandt8.cpp:244-335: parallax of the world rays, cv::Mat Triangulate
 * :498-512, both depths, both reprojection errors) -- parity by tolerance like orbfe_search_tri_kb8 -- while
 * Pinhole's returns false (include/CameraModels/Pinhole.h:88-91): kb8_1L == NULL (a pinhole first camera) yields 0
 * pairs.  Tcw* = rows 0..2 of KeyFrame::GetPose / GetRightPose (3x4 row-major); the ..R entries are read for
 * keyframes of a two-camera rig (Nleft != -1).  points[3*k..] is the world point of pairs[2*k..]. */
typedef struct {
    const uint8_t* desc1; int n1; const uint8_t* hasMP1; const float* kp1_xy; const float* angle1;
    const int32_t* octave1; orbfe_fv fv1; int Nleft1;
    const uint8_t* desc2; int n2; const uint8_t* hasMP2; const float* kp2_xy; const float* angle2;
    const int32_t* octave2; orbfe_fv fv2; int Nleft2;
    const float* kb8_1L; const float* kb8_1R; const float* kb8_2L; const float* kb8_2R; /* fx,fy,cx,cy,k0..k3 */
    const float* Tcw1L; const float* Tcw1R; const float* Tcw2L; const float* Tcw2R;     /* 12 floats each */
    const float* levelSigma2_1; const float* levelSigma2_2; int nlevels1, nlevels2;
    int check_orientation;
} orbfe_tri3d_args;
int orbfe_search_tri_3d(int device, const orbfe_tri3d_args*, int32_t* pairs /* 2*n1 */, float* points /* 3*n1 */);
/* KannalaBrandt8::TriangulateMatches(_) for n explicit keypoint pairs -- the gate above, and what
 * Frame::ComputeStereoFishEyeMatches evaluates for every knn match that passes the ratio test (src/Frame.cc:1146-1155,
 * KannalaBrandt8.cpp:337-407): z1[i] = depth in camera 1 or -1 (accepted when > 0.0001f), p3D[3*i..] = the
 * triangulated point (mvStereo3Dpoints; zeros for rejected pairs; p3D may be NULL). */
int orbfe_kb8_triangulate(int device, const float* params1, const float* params2, const float* kp1_xy, const float* kp2_xy,
                          const float* R12, const float* t12, const float* sigma1, const float* sigma2, int n, float* z1,
                          float* p3D);

/* Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1119-1159) in one call: knn-2 brute force between the
 * lapping-area descriptors (stereoDescLeft / stereoDescRight, i.e. the rows from monoLeft / monoRight on), Lowe
 * ratio 0.7, KannalaBrandt8::TriangulateMatches(mRlr, mtlr) of every survivor.  All arrays are the lapping-area
 * slices; the caller adds monoLeft / monoRight to the returned indices.  leftToRight[nL], rightToLeft[nR] = -1 or
 * the partner, depth[nL] = mvDepth (or -1), p3D[3*nL] = mvStereo3Dpoints; returns nMatches.  The triangulation is
 * parity-by-tolerance like orbfe_search_tri_kb8. */
int orbfe_stereo_fisheye_matches(int device, const uint8_t* descL, const float* kpL_xy, const int32_t* octL, int nL,
                                 const uint8_t* descR, const float* kpR_xy, const int32_t* octR, int nR,
                                 const float* params1, const float* params2, const float* Rlr, const float* tlr,
                                 const float* levelSigma2, int nlevels, int32_t* leftToRight, int32_t* rightToLeft,
                                 float* depth, float* p3D);

/* KannalaBrandt8::unproject for n pixels (params = fx,fy,cx,cy,k0..k3). rays = 3 floats per pixel. */
int orbfe_kb8_unproject(int device, const float* params8, const float* uv, int n, float* rays);

/* Inner loops of ORBmatcher::SearchByProjection over one frame's grid:
 *   mode 0  (Frame& F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints), src/ORBmatcher.cc:44-197
 *           (Tracking::SearchLocalPoints): best / second best with the same-level ratio test;
 *   mode 1  (Frame& CurrentFrame, const Frame& LastFrame, th, bMono) :2193-2419 and
 *           (Frame& CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) :2421-2541: best only, TH = th_high,
 *           rotation-histogram cull when check_orientation.  The same loop, without the cull, is
 *           (KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) :473-586 and its vpMatchedKF twin :588-704
 *           (taken = vpMatched non-null, levels [pred-1, pred], th_high = floor(TH_LOW * ratioHamming)); and
 *           with qblocks all zero (no query hides a feature from a later one) it is the candidate search of
 *           Fuse(KeyFrame*, vpMapPoints, th, bRight) :1643-1841 (chi2_gate = 1, th_high = TH_LOW),
 *           Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) :1843-1965 and both directions of SearchBySim3
 *           :1967-2191 (th_high = TH_HIGH); q_match[q] is then that loop's bestIdx and the caller keeps the
 *           object logic (Replace / AddObservation, the mutual-agreement check).
 * The caller keeps the object walk (isBad, projection, RadiusByViewingCos, mnTrackScaleLevel ...) and passes
 * one query per GetFeaturesInArea call, in the reference's loop order.  Frame side: the N features with the
 * keypoints GetFeaturesInArea reads (mvKeysUn when Nleft == -1, else mvKeys ++ mvKeysRight), the grid
 * parameters of src/Frame.cc:158-159, and taken[i] = F.mvpMapPoints[i] && Observations()>0 (:83-85; for the
 * relocalisation overload simply non-null, :2484).  The library rebuilds Frame::AssignFeaturesToGrid
 * (src/Frame.cc:380-410) on the device. */
typedef struct {
    const uint8_t* desc; int n;            /* F.mDescriptors, 32-B rows                                  */
    const float* kx; const float* ky;      /* keypoint coordinates                                        */
    const int32_t* octave;
    const float* angle;                    /* only read for mode 1 + check_orientation                    */
    const float* uright;                   /* F.mvuRight or NULL; gate of :87-92 / :2270-2276 (Nleft == -1) */
    const uint8_t* taken;                  /* or NULL = none                                              */
    int Nleft;                             /* F.Nleft (-1 unless two-camera fisheye rig)                  */
    const int32_t* left_to_right;          /* F.mvLeftToRightMatch (Nleft entries) or NULL; mode 0        */
    const int32_t* right_to_left;          /* F.mvRightToLeftMatch (n - Nleft entries) or NULL; mode 0    */
    float minX, minY, gridWInv, gridHInv;  /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    int nq;
    const uint8_t* qdesc;                  /* pMP->GetDescriptor(), 32 B per query                        */
    const float* qx; const float* qy;      /* projection                                                  */
    const float* qr;                       /* window half-size handed to GetFeaturesInArea                */
    const int32_t* qmin_level; const int32_t* qmax_level; /* its minLevel / maxLevel arguments           */
    const float* qxr;                      /* mTrackProjXR (:89) or uv.x - mbf*invzc (:2272); with uright */
    const uint8_t* qflags;                 /* or NULL. bit0: right-camera grid (bRight); bit1: right-camera
                                              search of the previous query's map point, skipped when that
                                              query was rejected by the ratio test (`continue`, :128);
                                              bit2: skipped when the previous query's GetFeaturesInArea came
                                              back empty (`if(vIndices2.empty()) continue;` :2255 stands
                                              before the right-camera block :2326 of the same point)      */
    const float* qangle;                   /* keypoint angle in the last frame / keyframe (mode 1)        */
    const uint8_t* qblocks;                /* or NULL = all. pMP->Observations()>0 of the query's point   */
    int mode; float nnratio; int th_high; int check_orientation;
    const float* inv_level_sigma2;         /* pKF->mvInvLevelSigma2, read when chi2_gate                   */
    int n_levels;                          /* its length (octaves are checked against it)                  */
    int chi2_gate;                         /* mode 1: Fuse's per-candidate reprojection test (:1773-1799):
                                              e2 = ex^2 + ey^2 (+ (qxr - uright)^2 when uright >= 0) with
                                              e2 * inv_level_sigma2[octave] > 5.99 (7.8) -> skipped; replaces
                                              the uright window gate of :2270-2276                          */
} orbfe_proj_args;
/* q_match[nq] = feature written by the query itself (before the orientation cull) or -1; feat_match[n] =
 * index of the query whose map point ends up in F.mvpMapPoints[i], -1 = entry left as it was (or culled).
 * Returns nmatches, or an error. */
int orbfe_search_projection(int device, const orbfe_proj_args*, int32_t* q_match, int32_t* feat_match);
int orbfe_search_projection_last_sweeps(void); /* sweeps the last call on this thread needed (diagnostic) */
/* `count` independent searches in one upload / three launches / one download (one workgroup column per search):
 * the searches of a multi-camera rig's frames against the local map (src/Tracking.cc:2927, one call per camera in
 * the reference), the keyframe-by-keyframe Fuse calls of LocalMapping::SearchInNeighbors (src/LocalMapping.cc
 * :803-870), or the frames of several trackers served by one GPU.  items[k] / q_match[k] / feat_match[k] are
 * the arguments of the k-th orbfe_search_projection call, nmatches[k] its return value.  Returns 0 or an error
 * (then no output is defined). */
int orbfe_search_projection_batch(int device, const orbfe_proj_args* items, int count, int32_t* const* q_match,
                                  int32_t* const* feat_match, int32_t* nmatches);

/* The frame side of the projection searches kept on the device between calls.  Tracking runs several of them against
 * the same Frame -- SearchByProjection(CurrentFrame, LastFrame, th, ...) and again with 2*th when too few matches came
 * back (src/Tracking.cc:2817-2827), then SearchLocalPoints (:2927) -- and every call re-uploads the frame's descriptors
 * and keypoints and rebuilds Frame::AssignFeaturesToGrid.  orbfe_frame_create uploads desc / kx / ky / octave
 * (+ angle, uright when given), Nleft and the grid parameters of an orbfe_proj_args once and builds the grid once; `desc`
 * may be a device pointer (orbfe_get_device_outputs of the extractor that produced the frame: nothing crosses PCIe
 * for it).  orbfe_search_projection_frame is orbfe_search_projection with the frame side taken from the handle: of the
 * argument only the queries, mode / thresholds, `taken` and the stereo-partner tables are read.  A handle is read-only
 * after creation and may serve several threads at once. */
typedef struct orbfe_frame orbfe_frame;
int orbfe_frame_create(orbfe_frame** out, int device, const orbfe_proj_args* frame_side);
int orbfe_search_projection_frame(orbfe_frame*, const orbfe_proj_args* queries, int32_t* q_match, int32_t* feat_match);
/* `count` searches against resident frame sides in ONE upload, three launches, one download (round 5): frames[k] is the handle
 * search k runs against -- entries may repeat (Tracking::Relocalization projects every candidate keyframe's points into the one
 * current frame, src/Tracking.cc:3846-3870) or differ (one keyframe's points fused into each neighbour that has a handle) --,
 * queries[k] its queries / taken / partner tables as for orbfe_search_projection_frame; outputs as orbfe_search_projection_batch. */
int orbfe_search_projection_frames(orbfe_frame* const* frames, const orbfe_proj_args* queries, int count, int32_t* const* q_match,
                                   int32_t* const* feat_match, int32_t* nmatches);
void orbfe_frame_destroy(orbfe_frame*);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:355-420) for npts map points in one launch: the
 * observation descriptors are pooled, point p owns rows offsets[p] .. offsets[p+1); best[p] = index (relative
 * to the point) of the descriptor with the least median distance to the others, -1 if the point has none. */
int orbfe_distinctive_descriptors(int device, const uint8_t* pool, const int32_t* offsets, int npts, int32_t* best);

/* DBoW2 vocabulary tree (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) in CSR form: children of node i are
 * child_ids[child_off[i] .. child_off[i+1]) in stored order, node 0 is the root; leaves carry word id + weight. */
typedef struct {
    int nnodes;
    const uint8_t* node_desc;   /* nnodes * 32                         */
    const int32_t* child_off;   /* nnodes + 1                          */
    const int32_t* child_ids;   /* child_off[nnodes]                   */
    const int32_t* node_word;   /* word id of a leaf, -1 otherwise     */
    const double* node_weight;  /* WordValue of a leaf (idf weight)    */
    int L;                      /* depth levels m_L                    */
} orbfe_vocab;
typedef struct orbfe_vocab_dev orbfe_vocab_dev; /* the tree resident on one device */
int orbfe_vocab_upload(orbfe_vocab_dev** out, int device, const orbfe_vocab* v);
/* TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1423): ORBvoc.txt-style text file ("k L scoring
 * weighting", then "parent isLeaf d0..d31 weight" per node) straight onto the device; children in file order, word
 * ids to the leaves in file order, as the reference builds them.  k / L / number of words are returned when asked. */
int orbfe_vocab_load_text(orbfe_vocab_dev** out, int device, const char* path, int* k_out, int* L_out, int* nwords_out);
void orbfe_vocab_free(orbfe_vocab_dev*);
/* TemplatedVocabulary::transform(feature, word_id, weight, &nid, levelsup) (:1217-1259) for n features in one
 * launch (Frame::ComputeBoW, src/Frame.cc:724-731, uses levelsup = 4).  The caller folds the per-feature
 * results into BowVector::addWeight / FeatureVector::addFeature in feature order (:1147-1160). */
int orbfe_vocab_transform(orbfe_vocab_dev*, const uint8_t* feats, int n, int levelsup, int32_t* word_id,
                          int32_t* node_id, double* weight);

/* WeightingType / ScoringType of the vocabulary (Thirdparty/DBoW2/DBoW2/BowVector.h:39-56: weighting 0 TF_IDF, 1 TF, 2 IDF,
 * 3 BINARY; scoring 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT).  orbfe_vocab_load_text takes
 * them from the file's first line ("k L scoring weighting", :1366-1368); an uploaded tree starts as TF_IDF / L1_NORM, what
 * Vocabulary/ORBvoc.txt ("10 6 0 0") is. */
int orbfe_vocab_set_types(orbfe_vocab_dev*, int weighting, int scoring);
int orbfe_vocab_get_types(orbfe_vocab_dev*, int* weighting, int* scoring);

/* Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:724-731, src/KeyFrame.cc:105-114) as a device-resident step:
 * TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (TemplatedVocabulary.h:1127-1192) -- the
 * per-feature descent AND the fold into the two maps (BowVector::addWeight / addIfNotExist / normalize, BowVector.cpp:34-86;
 * FeatureVector::addFeature, FeatureVector.cpp:31-45), bit-identical to the reference's maps for every weighting / scoring type.
 *   orbfe_bow_create(&bow, vocab, cap)       a handle for frames of up to `cap` features (<= 65535); reused frame after frame.
 *                                            The vocabulary must outlive it (orbfe_vocab_free after the last orbfe_bow_destroy).
 *   orbfe_compute_bow(bow, desc, n, levelsup) `desc`: host or DEVICE pointer (orbfe_get_device_outputs: nothing crosses PCIe).
 *                                            Asynchronous on the calling thread's matcher stream -- two kernels (the descent;
 *                                            rank + fold, whose last workgroup also writes the host copy into page-locked
 *                                            memory of the handle); returns when they are queued.  One caller per handle at a time.
 *   orbfe_bow_fv(bow, &fv)                   the FeatureVector as an orbfe_fv that names the handle: a SearchByBoW batch against
 *                                            keyframe handles reads it where it lies (no host round trip between ComputeBoW and
 *                                            the search); every other consumer waits for the host copy itself.  Valid until the
 *                                            next orbfe_compute_bow on the handle.
 *   orbfe_bow_host(bow, &view)               the host copy, on request: waits for the call's kernels; pointers into
 *                                            page-locked memory of the handle, valid until the next orbfe_compute_bow.
 *                                            BowVector = (word_ids[i], word_values[i]), i < nw, ascending id (std::map order);
 *                                            FeatureVector = CSR (node_ids ascending, offsets[nn + 1], indices).
 *   orbfe_bow_device(bow, &view)             the same arrays on the device (stream-ordered behind the call; counts in
 *                                            d_header[0..3] = kept features, nodes, words, features of the largest node).
 *   orbfe_bow_destroy(bow)                   at any time: deferred until no call that was given the handle (or a vector that
 *                                            names it) is in progress; afterwards the address is refused (ORBFE_ERR_ARGS) by
 *                                            every entry point, also inside an orbfe_fv -- looked up, never dereferenced.
 * A feature whose word has weight 0 ("stopped", :1157) enters neither vector.  When the tree's leaves are shallower than
 * L - levelsup the reference leaves the node id uninitialised; here it is 0 (the root), as in orbfe_vocab_transform. */
typedef struct orbfe_bow orbfe_bow;
typedef struct {
    int n_kept, nn, nw, max_node;
    const uint32_t* node_ids; const int32_t* offsets; const int32_t* indices; /* FeatureVector */
    const uint32_t* word_ids; const double* word_values;                      /* BowVector     */
    const int32_t* d_header;                                                  /* device: the four counts */
} orbfe_bow_view;
int orbfe_bow_create(orbfe_bow** out, orbfe_vocab_dev*, int cap);
void orbfe_bow_destroy(orbfe_bow*);
int orbfe_compute_bow(orbfe_bow*, const uint8_t* desc, int n, int levelsup);
int orbfe_bow_fv(orbfe_bow*, orbfe_fv* fv);
/* on != 0: BowVector::normalize (a sum over the words in ascending id order, one dependent double addition after the other: a
 * single lane's work, 7-15 us of the kernel for 450-1000 words) is left to orbfe_bow_host, which runs the same arithmetic on the
 * mirrored values in a microsecond; the word values on the DEVICE then stay un-normalised -- nothing on the device reads them
 * (the searches use the FeatureVector, KeyFrameDatabase scores on the host).  Default off: both vectors complete on the device. */
int orbfe_bow_set_lazy_norm(orbfe_bow*, int on);
int orbfe_bow_host(orbfe_bow*, orbfe_bow_view* view);
int orbfe_bow_device(orbfe_bow*, orbfe_bow_view* view);

/* Device time (ms, hipEvents) of the matcher kernel launched by the last matcher call of this thread. */
float orbfe_matcher_last_kernel_ms(void);
/* Kernel timing of the matcher calls of this thread (events + a synchronisation per call): off by default. */
void orbfe_matcher_time_kernels(int on);

const char* orbfe_version(void);

#ifdef __cplusplus
}
#endif
#endif
