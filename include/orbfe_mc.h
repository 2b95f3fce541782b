/*
 * orbfe_mc.h -- C ABI of the multi-GPU / multi-camera path (BASELINE configs[3]).
 *
 * What it replaces: the reference has no multi-GPU code; its multi-camera structure is one ORBextractor per camera driven
 * from one host thread each (src/Frame.cc:119-122, two threads per stereo frame; src/System.cc:112-130 builds one
 * Tracking per process).  north_star shards the independent image pyramids of a rig / a batch over the GPUs of one node
 * -- one process per GPU -- and exchanges the 256-bit descriptors with ONE all-gather per batch for cross-camera
 * matching.  These entry points are what a C++ host (the only kind the reference has) binds for that:
 *
 *     rank 0:  orbfe_mc_unique_id(id)            -> broadcast `id` to the other ranks by any means (MPI, a socket, a file)
 *     all:     orbfe_mc_create(&mc, ctx, id, rank, world, frames_per_rank, cap, ORBFE_MC_RCCL)
 *     loop:    orbfe_mc_extract_exchange_submit(mc, d_imgs, ...)   extraction of this rank's frames straight into the
 *                                                 slab + one ncclAllGather on a side stream; two batches may be in flight
 *              orbfe_mc_extract_exchange_wait(mc, &view)            the oldest batch's gathered buffer is final
 *              orbfe_mc_match_ring(mc, hops, nhops, idx, dist)      knn-2 of this rank's frames against their partner
 *                                                 cameras, read from the gathered buffer in place (one launch)
 *
 * Slab of one rank (one contiguous buffer, so the exchange is a single collective; fixed size because the extractor's
 * output per image is bounded by orbfe_max_keypoints):
 *     [ frames_per_rank * cap * 32 B descriptors | frames_per_rank * int32 keypoint counts | pad to 256 B ]
 * The gathered buffer is `world` slabs back to back, rank r's at r * slab_bytes.
 *
 * Transports: ORBFE_MC_RCCL = ncclAllGather over xGMI (librccl is loaded when the first such handle is created: the
 * single-GPU library carries no dependency on it); ORBFE_MC_HOST = the same exchange through a POSIX shared-memory
 * segment (named after the id), for ranks that share a device, for tests of the bookkeeping in a container without a
 * GPU (ctx == NULL: the slabs then live in host memory and only the *_host calls and the layout queries are available),
 * and for world == 1.
 *
 * Error convention as orbfe.h: 0 / >= 0 success, negative = ORBFE_ERR_*; never throws, never aborts.
 */
#ifndef ORBFE_MC_H
#define ORBFE_MC_H

#include "orbfe.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_MC_RCCL 0
#define ORBFE_MC_HOST 1
#define ORBFE_MC_ID_BYTES 128 /* == sizeof(ncclUniqueId) */
/* An RCCL call failed (ncclGetUniqueId / ncclCommInitRank at create: no handle is returned; ncclAllGather at submit: the batch is
 * NOT in flight, the slot stays free).  orbfe_error_string knows the code; ORBFE_VERBOSE=1 prints RCCL's own reason to stderr. */
#define ORBFE_MC_ERR_RCCL (-8)

typedef struct orbfe_mc orbfe_mc;

typedef struct {
    size_t desc_bytes;  /* frames_per_rank * cap * 32                                 */
    size_t count_off;   /* byte offset of the int32 counts inside a slab (== desc_bytes) */
    size_t slab_bytes;  /* one rank's contribution, multiple of 256                   */
} orbfe_mc_layout_t;

/* Pure host bookkeeping (no device needed; the multi-process CPU tests drive these). */
int orbfe_mc_layout(int frames_per_rank, int cap, orbfe_mc_layout_t* out);
/* Contiguous block partition of `nframes` over `world` ranks: first frame and count of `rank`. */
int orbfe_mc_shard(int nframes, int world, int rank, int* first, int* count);
/* Cameras of a rig in a ring (global frame g = rank * frames_per_rank + local index): every local frame is a query
 * against the frames hops[h] cameras further round the ring.  pairs = nhops * frames_per_rank entries of
 * (local query frame, global train frame), hop-major.  Returns the number of pairs. */
int orbfe_mc_ring_pairs(int world, int frames_per_rank, int rank, const int* hops, int nhops, int32_t* pairs /* 2 per pair */);
/* Byte offsets of the job record of each pair: (query descriptors, query count) inside this rank's slab and
 * (train descriptors, train count) inside the gathered buffer.  offsets = 4 int64 per pair. */
int orbfe_mc_job_offsets(int frames_per_rank, int cap, const int32_t* pairs, int npairs, int64_t* offsets);

/* A fresh exchange id (rank 0 calls it; the caller carries the 128 bytes to the other ranks).  For ORBFE_MC_RCCL it is an
 * ncclUniqueId; for ORBFE_MC_HOST a random name. */
int orbfe_mc_unique_id(int transport, void* id128);

/* ctx: the extractor context whose device and stream the exchange uses (NULL only with ORBFE_MC_HOST: host-memory slabs).
 * cap >= orbfe_max_keypoints of the image size the batches will have. */
int orbfe_mc_create(orbfe_mc** out, orbfe_ctx* ctx, const void* id128, int rank, int world, int frames_per_rank, int cap,
                    int transport);
void orbfe_mc_destroy(orbfe_mc*);

/* Extraction of this rank's `frames_per_rank` images (DEVICE pointer, as orbfe_extract_batch_device) into the next slab
 * and, ordered after it on a side stream, the all-gather.  Returns at once; at most ORBFE_MC_MAX_IN_FLIGHT (3; round 3: 2)
 * batches in flight (ORBFE_ERR_STATE otherwise): the handle owns ORBFE_MC_SLOTS (4) slab / gathered buffer pairs.  Keypoints of the batch stay in a device buffer of the handle (orbfe_mc_view_t::d_kps). */
int orbfe_mc_extract_exchange_submit(orbfe_mc*, const uint8_t* d_imgs, int rows, int cols, size_t pitch,
                                     size_t img_stride_bytes, int lap0, int lap1);

#define ORBFE_MC_SLOTS 4
#define ORBFE_MC_MAX_IN_FLIGHT 3
typedef struct {
    const uint8_t* gathered;   /* world * slab_bytes, device memory (host memory for a ctx == NULL handle) */
    const uint8_t* slab;       /* this rank's slab of the same batch                                       */
    const orbfe_kp* d_kps;     /* frames_per_rank * cap keypoints of this rank's frames (device), or NULL  */
    const int32_t* d_mono;     /* monoIndex per local frame (device), or NULL                              */
    size_t slab_bytes;
    long batch;                /* sequence number of the batch                                             */
} orbfe_mc_view_t;
/* Blocks until the OLDEST submitted batch's collective has finished and describes its buffers; they stay valid until
 * two more batches have been submitted.  Returns ORBFE_ERR_STATE when nothing is in flight, a negative error when the
 * collective or the extraction failed. */
int orbfe_mc_extract_exchange_wait(orbfe_mc*, orbfe_mc_view_t* view);

/* Cross-camera matching, sharded by query frame: knn-2 (cv::BFMatcher(NORM_HAMMING).knnMatch(k=2), src/Frame.cc:1137) of
 * each local frame of the batch last returned by _wait against its ring partners (orbfe_mc_ring_pairs with these hops),
 * read from the gathered buffer in place, one launch.  idx / dist: host arrays of npairs * cap * 2 int32 (rows beyond a
 * query frame's count are -1), or NULL to leave the results on the device (orbfe_mc_match_outputs).  Returns npairs. */
int orbfe_mc_match_ring(orbfe_mc*, const int* hops, int nhops, int32_t* idx, int32_t* dist);
int orbfe_mc_match_outputs(orbfe_mc*, const int32_t** d_idx, const int32_t** d_dist, int* npairs);
/* The launch alone, queued on the context's stream without any host wait (bench.py's steady-state loop). */
int orbfe_mc_match_ring_async(orbfe_mc*, const int* hops, int nhops, long batch);

/* Host-memory form of the exchange (ORBFE_MC_HOST; the only data path of a ctx == NULL handle): contributes `slab`
 * (slab_bytes of host memory laid out as above) and returns this process's view of all ranks' slabs, valid until the
 * next call.  Collective: every rank of the exchange must call it. */
int orbfe_mc_exchange_host(orbfe_mc*, const uint8_t* slab, const uint8_t** gathered);

#ifdef __cplusplus
}
#endif
#endif
