/*
 * orbfe_extractor.hip -- host side of the extractor behind the C ABI (include/orbfe.h).
 *
 * Mirrors ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:43-107):
 *   orbfe_create          <- ORBextractor::ORBextractor      src/ORBextractor.cc:408-468
 *   orbfe_extract*        <- ORBextractor::operator()        src/ORBextractor.cc:1068-1150
 *   orbfe_get_level       <- mvImagePyramid                  include/ORBextractor.h:83
 * There is no CPU fallback: without a HIP device every entry point fails with ORBFE_ERR_NODEV.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/orbfe.h"
#include "../../include/orbfe_debug.h"
#include "orbfe_geom.h"
#include "orbfe_order.h"
#include "orbfe_pageable.h"
#include "orbfe_sincos.h"

// single translation unit: the kernels are compiled together with their launch code
#include "orbfe_kernels.hip"

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return -(1000 + (int)_e);    \
    } while (0)

namespace {

inline int cv_round_f(float v) { return (int)lrintf(v); }
inline int cv_floor_d(double v)
{
    int i = (int)v;
    return i - (i > v);
}
inline int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// ceil(2^32 / d) for the kernels' fast_div (0 encodes d == 1): x / d == umulhi(x, m) while x * d < 2^32
inline uint32_t recip32(unsigned d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) return -(1000 + (int)e);
        n = count;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
        // (fine-grained explicitly: kernels write results here that the host reads behind the completion word while the
        // kernel is, for the runtime, still running; the default is the same on this runtime unless HIP_HOST_COHERENT=0)
        hipError_t e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocCoherent);
        coherent = e == hipSuccess;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault);
        }
        if (e != hipSuccess) return -(1000 + (int)e);
        n = count;
        return 0;
    }
    bool coherent = false;
    // the address a kernel uses to read this buffer in place (zero-copy over PCIe)
    T* dev() const
    {
        void* d = nullptr;
        if (!p || hipHostGetDevicePointer(&d, (void*)p, 0) != hipSuccess) return p;
        return (T*)d;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
};

} // namespace

// Everything that depends on the image SIZE (derived tables on the host and their device copies).  A context keeps
// the state of the sizes it has seen (up to eight) so that a rig with unequal cameras -- or one extractor fed
// different sizes in turn -- does not rebuild and re-upload the tables at every switch.
struct orbfe_geom_state {
    int rows = 0, cols = 0;
    int pyrTile = 0; // tile side of the fused pyramid kernel at the coarsest level: part of the state's key
    std::vector<OrbLevelGeom> lg;
    std::vector<OrbCellGeom> cg;
    size_t pyrStride = 0, candStride = 0, keyStride = 0, kpStride = 0;
    int nCells = 0, maxKp = 0, maxListCap = 0;
    size_t qtLdsBytes = 0;
    int qtKeyOff = 0, qtKeyCap = 0;
    std::vector<int> qtSmall, qtBig; // levels whose quadtree tables live in LDS / in the global scratch area (k_octree<true>)
    size_t qtBigLdsBytes = 0, qtScratchStride = 0;
    int qtBigKeyOff = 0, qtBigKeyCap = 0;
    int fastPitch = 0, fastRows = 0, fastThreads = 256;
    size_t fastLdsBytes = 0;
    std::vector<OrbFastCell> fc; // K-FAST's cell records
    DevBuf<OrbFastCell> d_fc;
    std::vector<uint2> fastPat; // phase A's per-thread constants, fastThreads entries per pattern (OrbFastCell::pad names the pattern)
    DevBuf<uint2> d_fastPat;
    DevBuf<OrbDescSlot> d_ds; // K-DESC's per-slot records (level geometry of every keypoint slot)
    int fastTileBytes = 0, fastBmWords = 0;
    DevBuf<OrbLevelGeom> d_lg;
    DevBuf<OrbCellGeom> d_cg;
    DevBuf<OrbResizeX> d_xtab;
    DevBuf<OrbResizeY> d_ytab;
    DevBuf<uint4> d_pyrRecs; // the fused pyramid kernel's per-tile records (OrbPyrTileHdr + staged x groups + y entries)
    int pyrRecBytes = 0;
    int pyrNtx = 0, pyrNty = 0, pyrBuf0 = 0, pyrBuf1 = 0, pyrStageX = 0, pyrStageY = 0;
    size_t pyrLdsBytes = 0;
    bool pyrWeightsOk = true; // all resize weights in [0, 2050] with a0+a1, b0+b1 <= 2050 (k_pyr_fused drops the clamp)
    bool pyrFused = true;
    void release_tables()
    {
        d_lg.release(); d_cg.release(); d_fc.release(); d_fastPat.release(); d_ds.release(); d_xtab.release(); d_ytab.release(); d_pyrRecs.release();
    }
};

struct orbfe_ctx : orbfe_geom_state {
    std::vector<orbfe_geom_state> geomCache; // the other sizes this context has been used with
    // parameters (reference include/ORBextractor.h:89-105)
    int nfeatures;
    double scaleFactor; // a double initialised from a float, :96
    int nlevels, iniThFAST, minThFAST;
    int device;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    int taps[7] = {18, 34, 48, 56, 48, 34, 18};
    int trigMode = ORBFE_TRIG_LIBM;
    int atanFma = 0; // orbfe_set_atan_fma / ORBFE_ATAN_FMA: fused Horner steps in fastAtan2
    bool lastHostTrigCheck = false;

    hipStream_t stream = nullptr;
    bool ownStream = false;

    int fastXcdGroup = 4;        // cells per XCD group when a batch does not give whole images to the XCDs
    bool fastByImage = true;     // whole images per XCD for batches that fill the eight XCDs evenly
    int xcdAffine = 1;           // whole images per XCD when the batch is a multiple of 8
    int lanes = 1;
    // Batch lanes (round 5; orbfe_set_lanes(ctx, 2..4)): WHOLE device-pointer batches are dealt
    // round-robin to streams the context owns, so that kernels of DIFFERENT batches overlap -- the regime below 16 frames per
    // call (the per-rank shard of BASELINE configs[3], a stereo pair, a single frame) is a chain of four latency-bound kernels,
    // and two half-batches of one small call (round 4's form, removed in round 6) only made every link of that chain smaller.
    // Measured with separate contexts first (profiles/r05_lanes_probe.txt: 8 x 1280x720, 0.084 ms per batch on one stream,
    // 0.052 / 0.045 / 0.043 with 2 / 3 / 4 contexts).  Each lane owns the per-batch device state (LaneBufs: the members of the same
    // names below, swapped into the context for the lane's turn, so every other entry point keeps reading "the last batch" from
    // the context's own members); the size-dependent tables, taps, pattern and lapping table are shared and read-only while
    // lanes are busy.  The context's stream carries no kernels in this mode, only the ordering: a lane waits for the point of
    // the call on it (inputs ready), it waits for the lane's K-PYR (inputs consumed: a caller may refill the images in stream
    // order, as with one lane), and for whole lanes only at the joins.
    struct LaneBufs {
        DevBuf<uint8_t> d_pyr;
        DevBuf<uint32_t> d_cand, d_keys, d_lvlKp, d_lvlPre;
        DevBuf<uint16_t> d_keyNode;
        DevBuf<int32_t> d_cellCount, d_lvlCount, d_destMap;
        DevBuf<int4> d_fix;
        DevBuf<int> d_qtScratch;
        PinBuf<int4> h_fix, h_fixAB;
        DevBuf<unsigned> d_done; // the completion word's counters and flag, K-STEREO's result arrays: one set per lane, so that the
        PinBuf<unsigned> h_done; // stereo frames orbfe_extract_stereo_pair_submit keeps in flight do not count into each other's words
        DevBuf<float> d_stereo;
        PinBuf<float> h_stereo;
        int capImgs = 0, capKp = 0;
    };
    struct Lane {
        LaneBufs parked;          // this lane's buffers while another lane's are in the context (empty while its own are)
        hipStream_t stream = nullptr;
        hipStream_t pairStream = nullptr; // stereo frames in flight (orbfe_extract_stereo_pair_submit): a stream of NORMAL priority per
                                          // lane -- frames are collected in submission order, and a frame on a low-priority queue
                                          // waits behind its younger neighbours (three in flight: 0.091 ms per frame with the batch
                                          // lanes' mixed priorities, 0.045 with equal ones)
        bool pendingPair = false; // pairStream holds work the context's stream has not been ordered after
        hipEvent_t evJoin = nullptr, evRead = nullptr;
        bool pending = false;     // holds work the context's stream has not been ordered after
        int imgs = 0;             // images of its last batch (orbfe_sync reads that many ... one status header)
    } lane[ORBFE_MAX_LANES];
    int curSet = 0;               // the lane whose buffers the context's members hold
    int laneNext = 0, laneLast = -1; // lane of the next whole-batch call / of the last one (-1: the last call ran on the context's stream)
    int pairLast = -1;               // lane of the newest stereo frame submitted with orbfe_extract_stereo_pair_submit
    int pairBlockingResult = 0;      // ... and, with one lane, what the blocking call inside _submit returned
    hipEvent_t evBatchFork = nullptr;
    bool inputGuard = true;       // orbfe_set_lane_input_guard / ORBFE_LANES_INPUT_GUARD=0: the context's stream does not wait for a lane's K-PYR
    bool exchangeHint = false;    // orbfe_mc_create: a collective stream of the highest priority waits for this context's lanes
    bool forkSkipIdle = true;     // ORBFE_LANES_FORK_ALWAYS=1 (A/B): record + wait even when the context's stream is idle
    hipStream_t runStream = nullptr; // run_device / ensure_capacity: the stream of the call being queued (nullptr: `stream`)
    hipEvent_t guardEv = nullptr;    // run_device records it behind K-PYR (the call's last reader of the caller's images)

    // device state
    int capImgs = 0, capKp = 0; // allocated batch size / per-image keypoint capacity
    DevBuf<uint8_t> d_pyr;
    DevBuf<uint32_t> d_cand, d_keys, d_lvlKp;
    DevBuf<uint16_t> d_keyNode;
    DevBuf<int32_t> d_cellCount, d_lvlCount, d_lap;
    DevBuf<int4> d_fix; // [0] = {count,0,0,0}, then one entry per flagged keypoint
    DevBuf<float> d_kb8, d_rays;
    bool kb8On = false;
    float kb8[8] = {0};
    float* userRays = nullptr; // device pointer supplied by orbfe_set_ray_output
    DevBuf<int32_t> d_destMap; // K-PACK's output slot per keypoint slot (fisheye rays only)
    DevBuf<uint32_t> d_lvlPre; // K-QT's partition word per keypoint slot
    DevBuf<int> d_qtScratch;   // node tables of the levels that do not fit the LDS (k_octree<true>)
    DevBuf<int> d_taps;
    bool tapsDirty = true;
    uint32_t tapWords[32] = {}; // host copy of what d_taps holds (stays alive while the upload is in flight)
    int lapDev0 = 0, lapDev1 = 0, lapDevCount = 0; // what d_lap currently holds (orbfe_extract_batch_device)
    DevBuf<float4> d_patternF;
    PinBuf<int4> h_fix, h_fixAB; // pinned: h_fixAB is read by the fix-up kernel directly (zero-copy)

    // Host-pointer path (orbfe_extract, orbfe_extract_batch, orbfe_extract_batch_submit / _wait): two slots, each
    // with its own device input and output buffers and pinned staging, so that the H2D of batch i+1 and the D2H of
    // batch i-1 run beside the kernels of batch i.
    struct HostSlot {
        DevBuf<uint8_t> d_img;  // the batch's images
        DevBuf<uint8_t> d_out;  // [meta: n[B] | mono[B] | err, 64-B padded][kps B*cap*28][desc B*cap*32]
        PinBuf<uint8_t> h_in;   // staging of images that are not in pinned memory
        PinBuf<uint8_t> h_out;  // meta always; the keypoint / descriptor slabs when the caller's arrays are pageable
        PinBuf<int32_t> h_lap;  // per-image lapping ranges, read in place by K-QT (no H2D command)
        int32_t* d_lapAlias = nullptr;
        hipEvent_t evIn = nullptr, evK = nullptr, evDone = nullptr;
        bool busy = false, pipelined = false, outPinned = false;
        unsigned doneSeq = 0; // != 0: K-DESC publishes this number in the context's completion word when the slab is complete
        int nimg = 0, cap = 0;
        size_t metaBytes = 0;
        orbfe_kp* kps = nullptr;
        uint8_t* desc = nullptr;
        int* n_out = nullptr;
        int* mono_out = nullptr;
        float* uRight = nullptr; // orbfe_extract_stereo_pair_submit: where _wait leaves Frame::ComputeStereoMatches' results
        float* depth = nullptr;
    } slot[ORBFE_MAX_LANES]; // [0], [1]: the two-slot pipeline of the host-pointer calls; [k]: lane k's stereo frame in flight
    long slotSubmitted = 0, slotRetired = 0; // FIFO over the two slots
    long pairSubmitted = 0, pairRetired = 0; // FIFO of orbfe_extract_stereo_pair_submit / _wait (slot = lane = sequence % lanes)
    hipStream_t sIn = nullptr, sOut = nullptr; // copy streams of the pipelined form
    std::unordered_map<const void*, bool> pinnedCache; // what hipPointerGetAttributes said about a caller pointer
    // orbfe_set_auto_register: pageable caller buffers that came back (same address, same size) are page-locked by the
    // library on their second sighting and from then on take the DMA path; at most 16 at a time, least recently used out
    bool zeroCopy = true; // ORBFE_ZEROCOPY=0: the latency path downloads its results with a copy command (A/B)
    bool uploadKernel = true; // ORBFE_UPLOAD_KERNEL=0: the latency path uploads its images with copy commands (A/B)
    int mirrorMaxImgs = 2; // ORBFE_MIRROR_MAX: blocking calls of up to this many images write their results to pinned memory from the kernels
    // completion word of those calls (OrbDone in orbfe_kernels.hip; ORBFE_SPIN=0: wait in hipStreamSynchronize as before)
    bool spinWait = true;
    DevBuf<unsigned> d_done;  // 65 counters
    PinBuf<unsigned> h_done;  // the flag word
    unsigned doneSeq = 0;     // last sequence number handed out
    int spinMisses = 0;       // consecutive waits in which the word never came (spin_settle): at 8 spinWait is switched off for this context
    unsigned spinLate = 0;    // the number a wait ran into its bound for (spin_done -> spin_settle)
    int lapInlineN = 0, lapInline[4] = {0, 0, 0, 0}; // host_submit -> run_device: the lapping ranges of a call of <= 2 images
    bool doneWant = false;    // host_submit -> run_device: the caller wants the word for this call
    unsigned doneGot = 0;     // run_device -> host_submit: the number K-DESC will publish, or 0
    bool autoRegister = false;
    struct AutoPin {
        const void* p;
        size_t n;
        bool registered;
        unsigned long used;
    };
    std::vector<AutoPin> autoPins;
    unsigned long autoClock = 0;
    // where the last call left its outputs on the device (orbfe_compute_stereo_matches_resident, orbfe_frame_*)
    const float* lastKps = nullptr;
    const uint8_t* lastDesc = nullptr;
    const int32_t* lastN = nullptr;
    int lastCap = 0;
    bool lastPacked = false; // the last batch ran K-PACK (d_destMap holds its output slots)
    DevBuf<float> d_stereo; // uRight | depth | sad of orbfe_compute_stereo_matches_resident
    PinBuf<float> h_stereo;
    DevBuf<uint8_t> d_stereoIo; // orbfe_compute_stereo_matches: keypoints and descriptors of both images | uRight | depth | sad
    PinBuf<uint8_t> h_stereoIo;
    PinBuf<uint8_t> h_level; // orbfe_get_level's staging (mvImagePyramid on request)
    hipEvent_t evStereo = nullptr;
    std::vector<int> stereoSads; // scratch of the outlier cut (kept: no allocation per frame)
    hipEvent_t evOutputs = nullptr; // recorded by orbfe_get_device_outputs: the point after which the resident outputs are final

    int lastImgs = 0;
    int lastFixups = 0;
    // profiling: a ring of event sets, one set per call, read (averaged) after the timed region
    static const int kProfSets = 256;
    bool profile = false;
    int profEvery = 1;   // record an event set on every profEvery-th call
    long profSeen = 0;   // calls since profiling was enabled
    bool recNow = false; // the current call records
    std::vector<hipEvent_t> ev; // kProfSets * (ORBFE_STAGE_COUNT + 1)
    bool evReady = false;
    long profCalls = 0;
    bool packSkipped[kProfSets] = {}; // event set k was recorded without a K-PACK launch (no event between K-QT and K-DESC)
};

namespace {

inline hipStream_t run_stream(const orbfe_ctx* c) { return c->runStream ? c->runStream : c->stream; }

// Orders the context's stream after whatever the lanes still have in flight (one-way waits: a few us on the stream).
int lane_join(orbfe_ctx* c)
{
    if (!c) return 0;
    for (int k = 0; k < ORBFE_MAX_LANES; k++) {
        orbfe_ctx::Lane& L = c->lane[k];
        if (L.pendingPair) {
            HIP_TRY(hipEventRecord(L.evJoin, L.pairStream));
            HIP_TRY(hipStreamWaitEvent(c->stream, L.evJoin, 0));
            L.pendingPair = false;
        }
        if (!L.pending) continue;
        HIP_TRY(hipEventRecord(L.evJoin, L.stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, L.evJoin, 0));
        L.pending = false;
    }
    return 0;
}
// ... and the host: before buffers are freed / tables replaced / a stream is given up
void lane_quiesce(orbfe_ctx* c)
{
    if (!c) return;
    for (int k = 0; k < ORBFE_MAX_LANES; k++) {
        if (c->lane[k].stream) (void)hipStreamSynchronize(c->lane[k].stream);
        if (c->lane[k].pairStream) (void)hipStreamSynchronize(c->lane[k].pairStream);
        c->lane[k].pending = c->lane[k].pendingPair = false;
    }
}
bool lanes_busy(const orbfe_ctx* c)
{
    bool any = false;
    for (int k = 0; k < ORBFE_MAX_LANES; k++) any = any || c->lane[k].pending || c->lane[k].pendingPair;
    return any;
}
// The per-batch buffers of the context <-> a parking place
void lane_swap_bufs(orbfe_ctx* c, orbfe_ctx::LaneBufs& b)
{
    std::swap(c->d_pyr, b.d_pyr);
    std::swap(c->d_cand, b.d_cand);
    std::swap(c->d_keys, b.d_keys);
    std::swap(c->d_lvlKp, b.d_lvlKp);
    std::swap(c->d_lvlPre, b.d_lvlPre);
    std::swap(c->d_keyNode, b.d_keyNode);
    std::swap(c->d_cellCount, b.d_cellCount);
    std::swap(c->d_lvlCount, b.d_lvlCount);
    std::swap(c->d_destMap, b.d_destMap);
    std::swap(c->d_fix, b.d_fix);
    std::swap(c->d_qtScratch, b.d_qtScratch);
    std::swap(c->h_fix, b.h_fix);
    std::swap(c->h_fixAB, b.h_fixAB);
    std::swap(c->d_done, b.d_done);
    std::swap(c->h_done, b.h_done);
    std::swap(c->d_stereo, b.d_stereo);
    std::swap(c->h_stereo, b.h_stereo);
    std::swap(c->capImgs, b.capImgs);
    std::swap(c->capKp, b.capKp);
}
// Lane k's buffers into the context (the current ones are parked with their lane)
void lane_select(orbfe_ctx* c, int k)
{
    if (k == c->curSet) return;
    lane_swap_bufs(c, c->lane[c->curSet].parked);
    lane_swap_bufs(c, c->lane[k].parked);
    c->curSet = k;
}
// (the per-image strides changed: every lane re-checks its buffers' sizes)
void lanes_invalidate_caps(orbfe_ctx* c)
{
    c->capImgs = 0;
    for (int k = 0; k < ORBFE_MAX_LANES; k++) c->lane[k].parked.capImgs = 0;
}
// Streams and events of the batch lanes.  Streams of ONE priority share four hardware queues (GPU_MAX_HW_QUEUES) with every other
// stream of that priority in the process, and two lanes in one queue serialise (three contexts on normal-priority streams
// beside torch's: 0.063 ms per 8 x 1280x720 batch against 0.045 with GPU_MAX_HW_QUEUES=8, profiles/r05_lanes_probe.txt).
// Priorities have queues of their own, so the lanes live in the LOWEST class, where nothing else of the process does
// (1 = lowest, 0 = normal, -1 = highest; measured on that workload: three lanes 0.0452 ms per batch
// all lowest, 0.0432 as normal / lowest / highest, 0.0620 all normal; four lanes 0.0436 all lowest, 0.0555 with a second
// normal one).  The mixed form is 4 % faster on an otherwise idle process but puts a lane into the class of orbfe_mc's
// collective stream (highest): behind a one-rank RCCL exchange three mixed lanes took 0.070 ms per batch, two took 0.055.
// Behind an exchange (orbfe_mc_create marks the context) the all-lowest form loses instead -- the collective's stream sits in a
// higher class with a wait for the lane that is not yet satisfied, and the low queues are served less while it does: 8 x
// 1280x720 with two lanes 0.092-0.098 ms per batch all lowest, 0.055 as normal / lowest; 64 x 752x480 0.186 against 0.178-0.180
// (profiles/r05_lanes_probe.txt) -- so a context that feeds orbfe_mc_* uses normal / lowest / normal / lowest.
int batch_lane_setup(orbfe_ctx* c)
{
    int prios[ORBFE_MAX_LANES] = {1, 1, 1, 1};
    if (c->exchangeHint) { // orbfe_mc_*: see the end of the comment above
        prios[0] = prios[2] = 0;
    }
    int least = 0, greatest = 0;
    const bool ranged = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    for (int k = 0; k < c->lanes && k < ORBFE_MAX_LANES; k++) {
        orbfe_ctx::Lane& L = c->lane[k];
        if (L.stream) continue;
        if (!(ranged && prios[k] != 0 &&
              hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, prios[k] < 0 ? greatest : least) == hipSuccess)) {
            (void)hipGetLastError();
            HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        }
        HIP_TRY(hipEventCreateWithFlags(&L.evJoin, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.evRead, hipEventDisableTiming));
    }
    if (!c->evBatchFork) HIP_TRY(hipEventCreateWithFlags(&c->evBatchFork, hipEventDisableTiming));
    return 0;
}

// ORBextractor ctor, reference src/ORBextractor.cc:408-444
void init_tables(orbfe_ctx* c)
{
    const int nl = c->nlevels;
    c->mvScaleFactor.resize(nl);
    c->mvLevelSigma2.resize(nl);
    c->mvScaleFactor[0] = 1.0f;
    c->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        c->mvScaleFactor[i] = (float)(c->mvScaleFactor[i - 1] * c->scaleFactor);
        c->mvLevelSigma2[i] = c->mvScaleFactor[i] * c->mvScaleFactor[i];
    }
    c->mvInvScaleFactor.resize(nl);
    c->mvInvLevelSigma2.resize(nl);
    for (int i = 0; i < nl; i++) {
        c->mvInvScaleFactor[i] = 1.0f / c->mvScaleFactor[i];
        c->mvInvLevelSigma2[i] = 1.0f / c->mvLevelSigma2[i];
    }
    c->mnFeaturesPerLevel.resize(nl);
    float factor = (float)(1.0f / c->scaleFactor);
    float nDesired = c->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        c->mnFeaturesPerLevel[l] = cv_round_f(nDesired);
        sum += c->mnFeaturesPerLevel[l];
        nDesired *= factor;
    }
    c->mnFeaturesPerLevel[nl - 1] = std::max(c->nfeatures - sum, 0);
}

// Level sizes (:1157), cell grid (:771-804), quadtree roots (:540-556), resize tables
int build_geometry(orbfe_ctx* c, int rows, int cols, std::vector<OrbResizeX>& xtab, std::vector<OrbResizeY>& ytab)
{
    const int nl = c->nlevels;
    c->lg.assign(nl, OrbLevelGeom());
    c->cg.clear();
    xtab.clear();
    ytab.clear();
    c->pyrWeightsOk = true;
    size_t off = 0;
    int slot = 0, kpBase = 0, keyBase = 0, maxKp = 0, maxLC = 0;
    for (int l = 0; l < nl; l++) {
        OrbLevelGeom& L = c->lg[l];
        const float inv = c->mvInvScaleFactor[l];
        L.w = cv_round_f((float)cols * inv);
        L.h = cv_round_f((float)rows * inv);
        if (L.w > ORBFE_MAX_DIM || L.h > ORBFE_MAX_DIM) return ORBFE_ERR_IMAGE_LARGE;
        L.pitch = (int)align_up((size_t)ORBFE_ROI_X0 + L.w + ORBFE_EDGE, 64);
        L.bufOff = (uint32_t)off;
        L.roiOff = (uint32_t)(off + (size_t)ORBFE_EDGE * L.pitch + ORBFE_ROI_X0);
        off = align_up(off + (size_t)(L.h + 2 * ORBFE_EDGE) * L.pitch, 256);
        if (off > 0xFFFFFFFFull) return ORBFE_ERR_IMAGE_LARGE;
        L.maxBX = L.w - ORBFE_EDGE + 3;
        L.maxBY = L.h - ORBFE_EDGE + 3;
        const float width = (float)(L.maxBX - ORBFE_MINB), height = (float)(L.maxBY - ORBFE_MINB);
        const float W = 35;
        if (!(width > 0) || !(height > 0)) return ORBFE_ERR_IMAGE_SMALL;
        L.nCols = (int)(width / W);
        L.nRows = (int)(height / W);
        if (L.nCols < 1 || L.nRows < 1) return ORBFE_ERR_IMAGE_SMALL; // the reference divides by zero here
        L.wCell = (int)std::ceil(width / L.nCols);
        L.hCell = (int)std::ceil(height / L.nRows);
        L.cellBase = (int)c->cg.size();
        int levelSlots = 0;
        for (int i = 0; i < L.nRows; i++) {
            const float iniY = (float)(ORBFE_MINB + i * L.hCell);
            float maxY = iniY + L.hCell + 6;
            if (iniY >= L.maxBY - 3) continue;
            if (maxY > L.maxBY) maxY = (float)L.maxBY;
            for (int j = 0; j < L.nCols; j++) {
                const float iniX = (float)(ORBFE_MINB + j * L.wCell);
                float maxX = iniX + L.wCell + 6;
                if (iniX >= L.maxBX - 6) continue;
                if (maxX > L.maxBX) maxX = (float)L.maxBX;
                OrbCellGeom g;
                g.level = (int16_t)l;
                g.iniX = (int16_t)iniX;
                g.iniY = (int16_t)iniY;
                g.cw = (int16_t)((int)maxX - (int)iniX);
                g.ch = (int16_t)((int)maxY - (int)iniY);
                g.offX = (int16_t)(j * L.wCell);
                g.offY = (int16_t)(i * L.hCell);
                {
                    const int ox = g.iniX & 3;
                    const int nd = (g.cw + ox + 3) >> 2;
                    const int txLo = 3 + ox, txHi = g.cw - 4 + ox;
                    const int ndz = std::max((txHi >> 2) - (txLo >> 2) + 1, 1);
                    const int zw1 = std::max(g.cw - 6, 1);
                    g.nd = (int16_t)nd;
                    auto recip = [](int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / (uint64_t)d); };
                    g.mNd = recip(nd);
                    g.mNdz = recip(ndz);
                    g.mZw = recip(zw1);
                    g.mZh = recip(std::max(g.ch - 6, 1));
                    g.pad2[0] = g.pad2[1] = 0;
                    g.roiOff = L.roiOff;
                    g.pitch = L.pitch;
                }
                if (g.cw > ORBFE_FAST_TILE - 1 || g.ch > ORBFE_FAST_TILE - 1) return ORBFE_ERR_ARGS;
                const int zw = std::max(g.cw - 6, 0), zh = std::max(g.ch - 6, 0);
                g.slotBase = slot;
                g.slotCap = std::max(((zw + 1) / 2) * ((zh + 1) / 2), 1);
                slot += g.slotCap;
                levelSlots += g.slotCap;
                c->cg.push_back(g);
            }
        }
        L.nCells = (int)c->cg.size() - L.cellBase;
        L.nFeat = c->mnFeaturesPerLevel[l];
        L.nIni = (int)std::round(static_cast<float>(L.maxBX - ORBFE_MINB) / (L.maxBY - ORBFE_MINB));
        L.hX = L.nIni > 0 ? static_cast<float>(L.maxBX - ORBFE_MINB) / L.nIni : 1.f;
        L.kpCap = std::max(L.nFeat + 3, 4 * std::max(L.nIni, 0)) + 1;
        L.kpBase = kpBase;
        kpBase += L.kpCap;
        L.keyBase = keyBase;
        L.keyCap = levelSlots;
        keyBase += levelSlots;
        L.listCap = L.kpCap + 3;
        maxLC = std::max(maxLC, L.listCap);
        maxKp += L.kpCap;
        L.scale = c->mvScaleFactor[l];
        L.size = (float)(int)(31 * c->mvScaleFactor[l]);
        L.xtabOff = (int)xtab.size();
        L.ytabOff = (int)ytab.size();
        if (l > 0) {
            const OrbLevelGeom& S = c->lg[l - 1];
            const double inv_x = (double)L.w / S.w, inv_y = (double)L.h / S.h;
            const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_d(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
                OrbResizeX t;
                t.sx = (uint16_t)sx;
                t.pad = (uint16_t)std::min(sx + 1, S.w - 1);
                t.a0 = sat_s16(cv_round_f((1.f - fx) * 2048));
                t.a1 = sat_s16(cv_round_f(fx * 2048));
                if (t.a0 < 0 || t.a1 < 0 || t.a0 + t.a1 > 2050) c->pyrWeightsOk = false;
                xtab.push_back(t);
                // k_pyr_fused: the source pixels of any 4 consecutive destination columns fit an 8-byte window
                if (dx >= 3 && (int)t.sx - (int)xtab[xtab.size() - 4].sx > 6) c->pyrWeightsOk = false;
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_d(fy);
                fy -= sy;
                OrbResizeY t;
                t.sy0 = (uint16_t)std::min(std::max(sy, 0), S.h - 1);
                t.sy1 = (uint16_t)std::min(std::max(sy + 1, 0), S.h - 1);
                t.b0 = sat_s16(cv_round_f((1.f - fy) * 2048));
                t.b1 = sat_s16(cv_round_f(fy * 2048));
                if (t.b0 < 0 || t.b1 < 0 || t.b0 + t.b1 > 2050) c->pyrWeightsOk = false;
                ytab.push_back(t);
            }
        }
    }
    c->pyrStride = off;
    c->candStride = (size_t)slot;
    c->keyStride = (size_t)keyBase;
    c->kpStride = (size_t)kpBase;
    c->nCells = (int)c->cg.size();
    c->maxKp = maxKp;
    c->maxListCap = maxLC;
    {
        int maxCw = 0, maxCh = 0, maxZone = 0;
        for (const OrbCellGeom& g : c->cg) {
            maxCw = std::max<int>(maxCw, g.cw);
            maxCh = std::max<int>(maxCh, g.ch);
            maxZone = std::max(maxZone, std::max(g.cw - 6, 0) * std::max(g.ch - 6, 0));
        }
        // Tile pitch in dwords: the widest staged row (its dwords + one: phase A reads d+1), rounded up to one of
        // the ODD pitches the kernel is instantiated for (an odd pitch spreads a dword column over all LDS banks).
        int maxNd = 1;
        for (const OrbCellGeom& g : c->cg) maxNd = std::max<int>(maxNd, g.nd);
        const int pd = maxNd + 1 <= 13 ? 13 : maxNd + 1 <= 17 ? 17 : 21;
        if (maxNd + 1 > 21) return ORBFE_ERR_ARGS; // cannot happen: cw <= 75 -> nd <= 20
        c->fastPitch = 4 * pd;
        c->fastRows = maxCh;
        // LDS: tile | score map | zone bitmap | its prefix sums | survivor queue (u16 per zone pixel)
        c->fastTileBytes = (int)(align_up((size_t)c->fastRows, 4) * c->fastPitch);
        c->fastBmWords = (int)align_up(((size_t)std::max(maxZone, 1) + 31) / 32, 4);
        // (the score map has four rows less than the tile: rows 2 .. rows-3)
        c->fastLdsBytes = align_up((size_t)2 * c->fastTileBytes - (size_t)4 * c->fastPitch + 8 * (size_t)c->fastBmWords +
                                       2 * (size_t)std::max(maxZone, 1), 16);
        int nt = maxZone <= 128 * 64 ? 128 : 256; // 128 measured fastest (198 us vs 257 @64, 234 @256; 64x 752x480)
        c->fc.clear();
        c->fastPat.clear();
        std::vector<std::pair<int, int>> patKeys; // (cw, ox) of the patterns built so far
        for (const OrbCellGeom& g : c->cg) {
            OrbFastCell f;
            const int ox = g.iniX & 3;
            const int txLo = 3 + ox, txHi = g.cw - 4 + ox;
            const int ndz = std::max((txHi >> 2) - (txLo >> 2) + 1, 1);
            f.gOff = g.roiOff + (uint32_t)g.iniY * (uint32_t)g.pitch + (uint32_t)(g.iniX - ox);
            f.pitch = (uint32_t)g.pitch;
            f.dims = (uint32_t)g.cw | ((uint32_t)g.ch << 8) | ((uint32_t)ox << 16) | ((uint32_t)ndz << 20);
            f.off = (uint32_t)(uint16_t)g.offX | ((uint32_t)(uint16_t)g.offY << 16);
            f.slotBase = (uint32_t)g.slotBase;
            f.slotCap = (uint32_t)g.slotCap;
            f.mNdz = g.mNdz;
            // phase A's per-thread constants depend on the cell's width and alignment only: thread t works on zone rows
            // t / ndz, t / ndz + rpp, ... (rpp = threads / ndz) of dword column (txLo >> 2) + t % ndz; of its four pixels those
            // inside the zone columns [txLo, txHi] count (bit 7 of each byte of the mask)
            const std::pair<int, int> key(g.cw, ox);
            size_t pi = 0;
            while (pi < patKeys.size() && patKeys[pi] != key) pi++;
            if (pi == patKeys.size()) {
                patKeys.push_back(key);
                const int rpp = nt / ndz, d0 = txLo >> 2, pd = c->fastPitch / 4;
                for (int t = 0; t < nt; t++) {
                    const int r0 = t / ndz, dz = t - r0 * ndz;
                    uint2 e = make_uint2(0xFFFFFFFFu, 0u);
                    if (r0 < rpp) {
                        const int d = d0 + dz, tx0 = 4 * d;
                        unsigned valid = 0xFu;
                        if (tx0 < txLo) valid &= 0xFu << (txLo - tx0);
                        if (tx0 + 3 > txHi) valid &= 0xFu >> (tx0 + 3 - txHi);
                        valid &= 0xFu;
                        e.x = (uint32_t)((r0 + 3) * pd + d) | ((uint32_t)r0 << 16);
                        e.y = ((valid & 1u) << 7) | ((valid & 2u) << 14) | ((valid & 4u) << 21) | ((valid & 8u) << 28);
                    }
                    c->fastPat.push_back(e);
                }
            }
            f.pad = (uint32_t)pi;
            c->fc.push_back(f);
        }
        c->fastThreads = nt;
    }
    // K-QT: a level's node tables take 24 ints per list entry.  Levels whose tables fit a workgroup's LDS (160 KB) run the
    // LDS instantiation; larger ones (round 4: nfeatures above ~7800, e.g. the 5 x nFeatures initialisation extractor of
    // src/Tracking.cc:1157 with KITTI's 2000) run k_octree<true> on a global scratch area instead of being refused.
    if (maxKp > 65535) return ORBFE_ERR_NFEATURES; // keypoint slots / list positions are packed in 16 bits
    c->qtSmall.clear();
    c->qtBig.clear();
    int maxLCsmall = 0, maxLCbig = 0;
    const int ldsNodeBudget = (160 * 1024 - 64 * (int)sizeof(int)) / (24 * (int)sizeof(int)); // list entries per workgroup
    int forceBig = -1;
    if (const char* e = getenv("ORBFE_QT_GLOBAL_FROM")) forceBig = atoi(e); // tests: list capacities >= this take the global path
    for (int l = 0; l < nl; l++) {
        const int LC = c->lg[l].listCap;
        if (LC > ldsNodeBudget || (forceBig >= 0 && LC >= forceBig)) {
            c->qtBig.push_back(l);
            maxLCbig = std::max(maxLCbig, LC);
        } else {
            c->qtSmall.push_back(l);
            maxLCsmall = std::max(maxLCsmall, LC);
        }
    }
    c->qtKeyOff = 64 + std::max(24 * maxLCsmall, 2048); // ints (the gather uses 2 x 1024 ints of the array area)
    c->qtLdsBytes = sizeof(int) * (size_t)c->qtKeyOff;
    // room for the key arrays (4 B key + 2 B node index each) while staying under 64 KB
    c->qtKeyCap = 0;
    if (c->qtLdsBytes + 6 * 1024 <= 64 * 1024) {
        c->qtKeyCap = (int)std::min<size_t>(4096, (64 * 1024 - c->qtLdsBytes) / 6) & ~63;
        c->qtLdsBytes += 6 * (size_t)c->qtKeyCap;
    }
    // the global instantiation: LDS holds the scalars, the gather's scan buffers and the key arrays only
    c->qtBigKeyOff = 64 + 2048;
    c->qtBigKeyCap = 4096;
    c->qtBigLdsBytes = sizeof(int) * (size_t)c->qtBigKeyOff + 6 * (size_t)c->qtBigKeyCap;
    c->qtScratchStride = align_up((size_t)24 * (size_t)std::max(maxLCbig, 1), 4);
    return 0;
}

int max_kp_for(orbfe_ctx* c, int rows, int cols)
{
    int total = 0;
    for (int l = 0; l < c->nlevels; l++) {
        const int w = cv_round_f((float)cols * c->mvInvScaleFactor[l]);
        const int h = cv_round_f((float)rows * c->mvInvScaleFactor[l]);
        const int bx = w - 32, by = h - 32;
        if (w > ORBFE_MAX_DIM || h > ORBFE_MAX_DIM) return ORBFE_ERR_IMAGE_LARGE;
        if (bx < 35 || by < 35) return ORBFE_ERR_IMAGE_SMALL;
        const int nIni = (int)std::round(static_cast<float>(bx) / by);
        total += std::max(c->mnFeaturesPerLevel[l] + 3, 4 * std::max(nIni, 0)) + 1;
    }
    return total;
}

// Tile ranges of the fused pyramid kernel along one axis (see OrbPyrRange).  lo0[t]/hi1[t] give, for
// destination index t of level l, the first / last source index of level l-1 it reads.
void build_pyr_ranges(int nlevels, const std::vector<int>& extent, const std::vector<std::vector<int>>& srcLo,
                      const std::vector<std::vector<int>>& srcHi, int ntiles, int T, std::vector<OrbPyrRange>& out,
                      int* maxNeed0, int* maxNeed1)
{
    const int top = nlevels - 1;
    std::vector<std::vector<int>> b(nlevels, std::vector<int>(ntiles + 1));
    for (int i = 0; i <= ntiles; i++) b[top][i] = std::min(i * T, extent[top]);
    b[top][ntiles] = extent[top];
    for (int l = top; l >= 1; l--) {
        for (int i = 0; i < ntiles; i++) b[l - 1][i] = b[l][i] < extent[l] ? srcLo[l][b[l][i]] : extent[l - 1];
        b[l - 1][0] = 0;
        b[l - 1][ntiles] = extent[l - 1];
    }
    out.assign((size_t)nlevels * ntiles, OrbPyrRange());
    for (int i = 0; i < ntiles; i++) {
        int need = b[top][i + 1];
        for (int l = top; l >= 0; l--) {
            OrbPyrRange r;
            r.lo = (int16_t)b[l][i];
            r.ownHi = (int16_t)b[l][i + 1];
            need = std::max(need, b[l][i + 1]);
            if (need < b[l][i]) need = b[l][i];
            r.needHi = (int16_t)need;
            r.pad = 0;
            out[(size_t)l * ntiles + i] = r;
            if (l == 0) *maxNeed0 = std::max(*maxNeed0, need - b[l][i]);
            if (l == 1) *maxNeed1 = std::max(*maxNeed1, need - b[l][i]);
            if (l >= 1) need = need > b[l][i] ? srcHi[l][need - 1] + 1 : b[l - 1][i]; // footprint in level l-1
        }
    }
}

// Tile side of the fused pyramid kernel at the coarsest level.  A big batch wants few, large tiles (less halo work:
// 24 is the fastest for 64 frames); a frame or two wants MANY workgroups, because then the kernel lasts as long as one
// workgroup's walk down the eight levels and the chip is nearly empty: 12 instead of 24 takes a single 752x480 frame's
// pyramid from 18.1 to 13.6 us (tile 16: 14.5, 20: 16.5, 32: 22.1).
int pyr_tile_for(int nimg)
{
    return nimg <= 4 ? 12 : ORBFE_PYR_TILE;
}

int ensure_geometry(orbfe_ctx* c, int rows, int cols, int nimg)
{
    const int tile = pyr_tile_for(nimg);
    if (rows == c->rows && cols == c->cols && tile == c->pyrTile && !c->lg.empty()) return 0;
    orbfe_geom_state& cur = *c;
    // a size this context has seen before: swap its tables back in (no rebuild, no upload)
    for (size_t i = 0; i < c->geomCache.size(); i++)
        if (c->geomCache[i].rows == rows && c->geomCache[i].cols == cols && c->geomCache[i].pyrTile == tile) {
            lane_quiesce(c); // (the second lane may still be working with the other size's tables and strides)
            std::swap(cur, c->geomCache[i]);
            if (c->geomCache[i].lg.empty()) c->geomCache.erase(c->geomCache.begin() + (long)i);
            if (c->qtLdsBytes > 64 * 1024) {
                HIP_TRY(hipFuncSetAttribute((const void*)k_octree<false, QT_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)c->qtLdsBytes));
            }
            lanes_invalidate_caps(c); // the per-image strides changed: re-check every buffer's size
            return 0;
        }
    if (!c->lg.empty()) { // keep the current size's tables for later
        if (c->geomCache.size() >= 8) {
            lane_quiesce(c);
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->geomCache.front().release_tables();
            c->geomCache.erase(c->geomCache.begin());
        }
        c->geomCache.push_back(cur);
        cur = orbfe_geom_state(); // (the copy above owns the device tables now)
    }
    std::vector<OrbResizeX> xtab;
    std::vector<OrbResizeY> ytab;
    c->rows = c->cols = 0;
    int r = build_geometry(c, rows, cols, xtab, ytab);
    if (r < 0) {
        c->lg.clear();
        return r;
    }
    if ((r = c->d_lg.ensure(c->lg.size())) < 0) return r;
    if ((r = c->d_cg.ensure(c->cg.size())) < 0) return r;
    if ((r = c->d_fc.ensure(c->fc.size())) < 0) return r;
    if ((r = c->d_fastPat.ensure(std::max<size_t>(c->fastPat.size(), 1))) < 0) return r;
    if ((r = c->d_ds.ensure(std::max<size_t>(c->kpStride, 1))) < 0) return r;
    if ((r = c->d_xtab.ensure(std::max<size_t>(xtab.size(), 1))) < 0) return r;
    if ((r = c->d_ytab.ensure(std::max<size_t>(ytab.size(), 1))) < 0) return r;
    lane_quiesce(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->d_lg.p, c->lg.data(), c->lg.size() * sizeof(OrbLevelGeom), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_cg.p, c->cg.data(), c->cg.size() * sizeof(OrbCellGeom), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_fc.p, c->fc.data(), c->fc.size() * sizeof(OrbFastCell), hipMemcpyHostToDevice));
    if (!c->fastPat.empty())
        HIP_TRY(hipMemcpy(c->d_fastPat.p, c->fastPat.data(), c->fastPat.size() * sizeof(uint2), hipMemcpyHostToDevice));
    {
        // K-DESC's slot records: slot kpBase + k of every level carries that level's geometry
        std::vector<OrbDescSlot> ds(c->kpStride);
        for (int l = 0; l < c->nlevels; l++) {
            const OrbLevelGeom& L = c->lg[l];
            for (int k = 0; k < L.kpCap; k++) {
                OrbDescSlot& d = ds[(size_t)L.kpBase + k];
                d.roiOff = L.roiOff;
                d.pitch = L.pitch;
                d.wh = (uint32_t)L.w | ((uint32_t)L.h << 16);
                d.lk = (uint32_t)l | ((uint32_t)k << 8);
                d.scale = L.scale;
                d.size = L.size;
                d.pad[0] = d.pad[1] = 0;
            }
        }
        if (!ds.empty()) HIP_TRY(hipMemcpy(c->d_ds.p, ds.data(), ds.size() * sizeof(OrbDescSlot), hipMemcpyHostToDevice));
    }
    if (!xtab.empty())
        HIP_TRY(hipMemcpy(c->d_xtab.p, xtab.data(), xtab.size() * sizeof(OrbResizeX), hipMemcpyHostToDevice));
    if (!ytab.empty())
        HIP_TRY(hipMemcpy(c->d_ytab.p, ytab.data(), ytab.size() * sizeof(OrbResizeY), hipMemcpyHostToDevice));
    {
        // fused pyramid: tile ranges per level along x and y
        const int nl = c->nlevels;
        std::vector<int> ex(nl), ey(nl);
        std::vector<std::vector<int>> xlo(nl), xhi(nl), ylo(nl), yhi(nl);
        for (int l = 0; l < nl; l++) {
            ex[l] = c->lg[l].w;
            ey[l] = c->lg[l].h;
            if (l == 0) continue;
            xlo[l].resize(ex[l]);
            xhi[l].resize(ex[l]);
            ylo[l].resize(ey[l]);
            yhi[l].resize(ey[l]);
            for (int t = 0; t < ex[l]; t++) {
                xlo[l][t] = xtab[c->lg[l].xtabOff + t].sx;
                xhi[l][t] = xtab[c->lg[l].xtabOff + t].pad;
            }
            for (int t = 0; t < ey[l]; t++) {
                ylo[l][t] = ytab[c->lg[l].ytabOff + t].sy0;
                yhi[l][t] = ytab[c->lg[l].ytabOff + t].sy1;
            }
        }
        const int T = tile;
        c->pyrNtx = (ex[nl - 1] + T - 1) / T;
        c->pyrNty = (ey[nl - 1] + T - 1) / T;
        std::vector<OrbPyrRange> prx, pry;
        int mx0 = 0, mx1 = 0, my0 = 0, my1 = 0;
        build_pyr_ranges(nl, ex, xlo, xhi, c->pyrNtx, T, prx, &mx0, &mx1);
        build_pyr_ranges(nl, ey, ylo, yhi, c->pyrNty, T, pry, &my0, &my1);
        // + 16 B slack: the interpolation reads 3 aligned dwords per source row from its first pixel
        c->pyrBuf0 = (int)align_up((size_t)((mx0 + 3) & ~3) * my0 + 16, 16);
        c->pyrBuf1 = (int)align_up((size_t)((mx1 + 3) & ~3) * std::max(my1, 1) + 16, 16);
        c->pyrStageX = c->pyrStageY = 0;
        for (int i = 0; i < c->pyrNtx; i++) {
            int sum = 0;
            for (int l = 1; l < nl; l++) // (x entries are staged per group of four destination columns)
                sum += (prx[(size_t)l * c->pyrNtx + i].needHi - prx[(size_t)l * c->pyrNtx + i].lo + 3) >> 2;
            c->pyrStageX = std::max(c->pyrStageX, sum);
        }
        for (int j = 0; j < c->pyrNty; j++) {
            int sum = 0;
            for (int l = 1; l < nl; l++) sum += pry[(size_t)l * c->pyrNty + j].needHi - pry[(size_t)l * c->pyrNty + j].lo;
            c->pyrStageY = std::max(c->pyrStageY, sum);
        }
        // The per-tile records (orbfe_geom.h): header | x groups (16 B of selectors + 16 B of weights + 4 B of source
        // position each) | y entries (8 B: 16-bit LDS row offsets -- a region stays below 64 KB, the whole allocation does --
        // and the two weights).  Everything the kernel used to derive per workgroup is folded here, once per image size.
        {
            const size_t SX = (size_t)c->pyrStageX, SY = (size_t)c->pyrStageY;
            const size_t recBytes = sizeof(OrbPyrTileHdr) + 32 * SX + 4 * align_up(SX, 4) + 8 * align_up(SY, 2);
            c->pyrRecBytes = (int)recBytes;
            const size_t ntiles = (size_t)c->pyrNtx * c->pyrNty;
            std::vector<uint8_t> recs(ntiles * recBytes, 0);
            for (int tj = 0; tj < c->pyrNty; tj++)
                for (int ti = 0; ti < c->pyrNtx; ti++) {
                    uint8_t* const rec = recs.data() + ((size_t)tj * c->pyrNtx + ti) * recBytes;
                    OrbPyrTileHdr H;
                    std::memset(&H, 0, sizeof H);
                    int xs = 0, ys = 0;
                    for (int l = 0; l < nl; l++) {
                        const OrbPyrRange X = prx[(size_t)l * c->pyrNtx + ti], Y = pry[(size_t)l * c->pyrNty + tj];
                        H.xlo[l] = X.lo;
                        H.xown[l] = X.ownHi;
                        H.xneed[l] = X.needHi;
                        H.ylo[l] = Y.lo;
                        H.yown[l] = Y.ownHi;
                        H.yneed[l] = Y.needHi;
                        H.roi[l] = (int32_t)c->lg[l].roiOff;
                        H.pitch[l] = c->lg[l].pitch;
                        const int ng = (X.needHi - X.lo + 3) >> 2; // groups of 4 columns per region row
                        H.recip[l] = ng > 1 ? (uint32_t)(((1ull << 32) + (unsigned)ng - 1) / (unsigned)ng) : 0u;
                        H.xo[l] = xs; // (levels 1 .. l-1 precede level l; level 0 stages nothing)
                        H.yo[l] = ys;
                        if (l >= 1) {
                            xs += ng;
                            ys += Y.needHi - Y.lo;
                        }
                    }
                    H.xo[nl] = xs;
                    H.yo[nl] = ys;
                    std::memcpy(rec, &H, sizeof H);
                    uint32_t* const xsel = reinterpret_cast<uint32_t*>(rec + sizeof H);
                    uint32_t* const xaw = xsel + 4 * SX;
                    uint32_t* const xbw = xaw + 4 * SX;
                    uint32_t* const ytw = xbw + align_up(SX, 4);
                    for (int l = 1; l < nl; l++) {
                        const int nW = H.xneed[l] - H.xlo[l], nH = H.yneed[l] - H.ylo[l];
                        const OrbResizeX* const tabx = xtab.data() + c->lg[l].xtabOff + H.xlo[l];
                        for (int g = 0; g < ((nW + 3) >> 2); g++) {
                            const size_t idx = (size_t)H.xo[l] + g;
                            OrbResizeX e[4];
                            for (int k = 0; k < 4; k++) e[k] = tabx[std::min(4 * g + k, nW - 1)];
                            for (int k = 0; k < 4; k++) {
                                const uint32_t o = (uint32_t)e[k].sx - (uint32_t)e[0].sx; // 0..6 (pyrWeightsOk / the scale check)
                                xsel[4 * idx + k] = 0x0C000C00u | o | ((o + 1u) << 16);
                                xaw[4 * idx + k] = (uint32_t)(uint16_t)e[k].a0 | ((uint32_t)(uint16_t)e[k].a1 << 16);
                            }
                            const uint32_t x0 = (uint32_t)((int)e[0].sx - H.xlo[l - 1]); // relative to the source region
                            xbw[idx] = (x0 & ~3u) | ((x0 & 3u) << 16);
                        }
                        const int sLoY = H.ylo[l - 1];
                        const int sPitch = (H.xneed[l - 1] - H.xlo[l - 1] + 3) & ~3; // LDS pitch of the source region
                        const OrbResizeY* const taby = ytab.data() + c->lg[l].ytabOff + H.ylo[l];
                        for (int rr = 0; rr < nH; rr++) {
                            const OrbResizeY e = taby[rr];
                            const size_t idx = (size_t)H.yo[l] + rr;
                            ytw[2 * idx] = (uint32_t)(((int)e.sy0 - sLoY) * sPitch) | ((uint32_t)(((int)e.sy1 - sLoY) * sPitch) << 16);
                            ytw[2 * idx + 1] = (uint32_t)(uint16_t)e.b0 | ((uint32_t)(uint16_t)e.b1 << 16);
                        }
                    }
                }
            if ((r = c->d_pyrRecs.ensure(recs.size() / 16 + 1)) < 0) return r;
            HIP_TRY(hipMemcpy(c->d_pyrRecs.p, recs.data(), recs.size(), hipMemcpyHostToDevice));
            c->pyrLdsBytes = (size_t)c->pyrBuf0 + c->pyrBuf1 + recBytes;
        }
        c->pyrFused = c->pyrLdsBytes <= 64 * 1024 && mx0 <= 1024 && mx1 <= 1024 && c->pyrWeightsOk;
    }
    if (c->qtLdsBytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_octree<false, QT_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)c->qtLdsBytes));
    }
    c->rows = rows;
    c->cols = cols;
    c->pyrTile = tile;
    lanes_invalidate_caps(c); // per-image strides changed: force re-allocation
    if (getenv("ORBFE_VERBOSE"))
        fprintf(stderr,
                "orbfe: %dx%d: pyramid %zu B/img, %d FAST cells; LDS per workgroup: k_pyr_fused %zu B (%dx%d tiles of "
                "%d px, fused=%d), k_fast_cells %zu B (%d threads), k_octree %zu B\n",
                cols, rows, c->pyrStride, c->nCells, c->pyrLdsBytes, c->pyrNtx, c->pyrNty, tile,
                (int)c->pyrFused, c->fastLdsBytes, c->fastThreads, c->qtLdsBytes);
    return 0;
}

int ensure_capacity(orbfe_ctx* c, int nimg, int capKp)
{
    if (nimg <= c->capImgs && capKp <= c->capKp) return 0;
    lane_quiesce(c); // (buffers are about to be replaced)
    const size_t B = (size_t)std::max(nimg, c->capImgs);
    const size_t K = (size_t)std::max(capKp, c->capKp);
    int r;
    // (a buffer that has to grow is freed first, and hipFree waits for the device by itself)
    if ((r = c->d_pyr.ensure(B * c->pyrStride + 256)) < 0) return r;
    if ((r = c->d_cand.ensure(B * c->candStride)) < 0) return r;
    if ((r = c->d_cellCount.ensure(B * c->nCells)) < 0) return r;
    if ((r = c->d_keys.ensure(B * c->keyStride)) < 0) return r;
    if ((r = c->d_keyNode.ensure(B * c->keyStride)) < 0) return r;
    if ((r = c->d_lvlKp.ensure(B * c->kpStride)) < 0) return r;
    if ((r = c->d_lvlPre.ensure(B * c->kpStride)) < 0) return r;
    {
        // rows of ORBFE_MAX_LEVELS counts per image, zeroed once: the levels an extractor does not have stay 0 and
        // K-DESC sums whole rows
        const size_t before = c->d_lvlCount.n;
        if ((r = c->d_lvlCount.ensure(B * ORBFE_MAX_LEVELS)) < 0) return r;
        if (c->d_lvlCount.n != before)
            HIP_TRY(hipMemsetAsync(c->d_lvlCount.p, 0, c->d_lvlCount.n * sizeof(int32_t), run_stream(c))); // (ordered before K-QT)
    }
    if (!c->qtBig.empty() && (r = c->d_qtScratch.ensure(B * c->qtBig.size() * c->qtScratchStride)) < 0) return r;
    {
        const int32_t* const before = c->d_lap.p; // (shared by the lanes: read-only while any of them is busy)
        if ((r = c->d_lap.ensure(B * 2)) < 0) return r;
        if (c->d_lap.p != before) c->lapDevCount = 0; // a new buffer
    }
    if ((r = c->d_destMap.ensure(B * std::max(K, c->kpStride))) < 0) return r;
    if ((r = c->d_fix.ensure(B * K + 1)) < 0) return r;
    if ((r = c->h_fix.ensure(B * K + 1)) < 0) return r;
    if ((r = c->h_fixAB.ensure(B * K + 1)) < 0) return r;
    if ((r = c->d_taps.ensure(32)) < 0) return r; // the seven taps, then (from word 8) K-DESC's eighteen packed tap words
    if ((r = c->d_kb8.ensure(8)) < 0) return r;
    if (c->kb8On && (r = c->d_rays.ensure(B * K * 3)) < 0) return r;
    if (!c->d_patternF.p) {
        if ((r = c->d_patternF.ensure(256)) < 0) return r;
        static const signed char pat[256][4] = ORBFE_PATTERN_31_INIT;
        std::vector<float4> pf(256);
        for (int i = 0; i < 256; i++) pf[i] = make_float4((float)pat[i][0], (float)pat[i][1], (float)pat[i][2], (float)pat[i][3]);
        HIP_TRY(hipMemcpy(c->d_patternF.p, pf.data(), 256 * sizeof(float4), hipMemcpyHostToDevice));
    }
    c->capImgs = (int)B;
    c->capKp = (int)K;
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// libm trig table (ORBFE_TRIG_LIBM without a host round trip per batch, see k_trig_codes).
// Built once per process and device: the host evaluates cosf/sinf -- the very calls of the reference,
// src/ORBextractor.cc:110-111, on this machine's libm -- for every float angle in [2^-7, 360] degrees
// (1.29e8 values, all cores), the device compares them with its correctly rounded values and keeps a
// 4-bit code per angle (65 MB of HBM).  Below 2^-7 degrees the argument is < 2^-12 rad, where
// cosf(x) == 1 and sinf(x) == x for the correctly rounded functions; that libm agrees is checked on a
// sample.  If any libm value is further than one bit pattern from the correctly rounded one, or the
// small-angle check fails, the table is not used and ORBFE_TRIG_LIBM falls back to the per-batch host
// check of the fragile keypoints.
struct TrigTable {
    uint8_t* d = nullptr;   // compact form: 4-bit codes, (U1 - U0 + 2) / 2 bytes
    float2* full = nullptr; // full form: libm's (cosf, sinf) per angle, 8 B x (U1 - U0 + 1) = 1.03 GB
    bool tried = false, ok = false;
};
struct TrigTabs { // what a launch gets: at most one of the two is set
    const uint8_t* codes;
    const float2* full;
};
std::mutex g_trigMutex;
TrigTable g_trig[16];

int host_threads()
{
    cpu_set_t set;
    int n = (int)std::thread::hardware_concurrency();
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, CPU_COUNT(&set));
    return std::max(1, std::min(n, 16));
}

void fill_libm(float2* out, uint32_t u0, uint32_t n)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f); // as the reference, :105
    const int T = host_threads();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([=]() {
            const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
            for (uint32_t i = lo; i < hi; i++) {
                uint32_t u = u0 + i;
                float deg;
                std::memcpy(&deg, &u, 4);
                volatile float ang = deg * factorPI; // one rounded float multiply, :110
                out[i] = make_float2(cosf(ang), sinf(ang));
            }
        });
    for (auto& x : th) x.join();
}

// cosf(x) == 1 and sinf(x) == x on a sample of x = angle * pi/180 with angle below the table
bool small_angles_ok()
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    for (uint32_t u = 0; u < ORBFE_TRIG_U0; u += 4099u) { // ~246k angles over all binades, odd stride
        float deg;
        std::memcpy(&deg, &u, 4);
        volatile float ang = deg * factorPI;
        const float x = ang;
        if (cosf(x) != 1.0f || sinf(x) != x) return false;
    }
    for (uint32_t u = ORBFE_TRIG_U0 - 70000u; u < ORBFE_TRIG_U0; u++) { // and every angle just below the table
        float deg;
        std::memcpy(&deg, &u, 4);
        volatile float ang = deg * factorPI;
        const float x = ang;
        if (cosf(x) != 1.0f || sinf(x) != x) return false;
    }
    return true;
}

// ---- the compact table's cache file.  Building the table means evaluating libm's cosf / sinf for 1.29e8 angles (all
// host cores, and every rank of a node would do it at once).  The 65 MB of codes are therefore kept in a file (default
// directory /dev/shm, ORBFE_TRIG_CACHE=<dir> or =0 for none) named after the user and a fingerprint of THIS host's libm, and
// the next process -- a rank of the same job, the next run -- reads them back instead: first call 60-90 ms -> ~15 ms.
// Round 4: the directory is world-writable, so the file is only trusted when it is a regular
// file (no symlink followed) that belongs to this user and that nobody else may write; it is created exclusively with mode
// 0600 under a per-user name; its WHOLE payload is checksummed -- on the device, where it has just been uploaded -- and a
// file that fails any of this is ignored and rebuilt; the bytes are read() into pinned staging, never mapped (a file
// truncated under a mapping would raise SIGBUS inside the copy).
struct TrigCacheHeader {
    char magic[8];       // "ORBFETC4"
    uint32_t u0, n;      // first angle (bit pattern) and count
    uint64_t libmPrint;  // FNV-1a over libm's results on a sample and the C library's version string
    uint64_t payloadSum; // trig_payload_sum of the code bytes (zero-padded to whole 8-byte words)
};
uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
// the checksum k_trig_checksum computes, on the host (used when a file is written from host memory and by the checker the
// CPU tests call); the tail of a payload that is not a multiple of 8 bytes counts as zero-padded
uint64_t trig_payload_sum(const uint8_t* p, size_t n, uint64_t firstWord = 0)
{
    uint64_t acc = 0;
    const size_t nw = n / 8;
    for (size_t i = 0; i < nw; i++) {
        uint64_t w;
        std::memcpy(&w, p + 8 * i, 8);
        acc += trig_mix64(w, firstWord + i);
    }
    if (n > nw * 8) {
        uint64_t w = 0;
        std::memcpy(&w, p + 8 * nw, n - nw * 8);
        acc += trig_mix64(w, firstWord + nw);
    }
    return acc;
}
extern "C" const char* gnu_get_libc_version(void);
uint64_t libm_fingerprint()
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    uint64_t h = 1469598103934665603ull;
    for (uint32_t u = ORBFE_TRIG_U0; u <= ORBFE_TRIG_U1; u += 30011u) { // ~4300 angles over the whole table
        float deg;
        std::memcpy(&deg, &u, 4);
        volatile float ang = deg * factorPI;
        const float v[2] = {cosf(ang), sinf(ang)};
        h = fnv1a(v, sizeof v, h);
    }
    const char* ver = gnu_get_libc_version(); // another C library build gets another file even if the sample agrees
    if (ver) h = fnv1a(ver, std::strlen(ver), h);
    return h;
}
std::string trig_cache_path(uint64_t print)
{
    const char* dir = getenv("ORBFE_TRIG_CACHE");
    if (dir && (!*dir || !std::strcmp(dir, "0"))) return std::string();
    char name[112];
    std::snprintf(name, sizeof name, "/orbfe_trigcodes_u%lu_%016llx.bin", (unsigned long)geteuid(), (unsigned long long)print);
    return std::string(dir ? dir : "/dev/shm") + name;
}
// Opens the cache file if it may be trusted and its header fits: a regular file (O_NOFOLLOW: no symlink), owned by this
// user, not writable by group or others, of exactly the expected size, with the right magic / range / libm fingerprint.
// Returns the descriptor positioned at the payload (the caller closes it), or -1; *hdOut receives the header.
int trig_cache_open(const std::string& path, uint64_t print, size_t bytes, TrigCacheHeader* hdOut, const char** why = nullptr)
{
    const char* dummy;
    if (!why) why = &dummy;
    *why = "no cache directory";
    if (path.empty()) return -1;
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC | O_NONBLOCK);
    *why = "cannot be opened (absent, or a symbolic link)";
    if (fd < 0) return -1;
    struct stat st;
    TrigCacheHeader hd;
    *why = nullptr;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) *why = "not a regular file";
    else if (st.st_uid != geteuid()) *why = "owned by another user";
    else if (st.st_mode & (S_IWGRP | S_IWOTH)) *why = "writable by group or others";
    else if ((size_t)st.st_size != sizeof(TrigCacheHeader) + bytes) *why = "wrong size";
    else if (read(fd, &hd, sizeof hd) != (ssize_t)sizeof hd) *why = "short read";
    else if (std::memcmp(hd.magic, "ORBFETC4", 8) || hd.u0 != ORBFE_TRIG_U0 || hd.n != ORBFE_TRIG_U1 - ORBFE_TRIG_U0 + 1u)
        *why = "another format or angle range";
    else if (hd.libmPrint != print) *why = "another libm";
    if (*why) {
        close(fd);
        return -1;
    }
    *hdOut = hd;
    return fd;
}
bool read_fully(int fd, uint8_t* dst, size_t n)
{
    while (n) {
        const ssize_t k = read(fd, dst, n);
        if (k <= 0) return false;
        dst += k;
        n -= (size_t)k;
    }
    return true;
}
// Cache file -> device table `d` (codeBytes, allocated with room for the zero padding to whole words), through two pinned
// staging buffers (read() of chunk i+1 beside the DMA of chunk i), then the full-payload checksum on the device.
bool trig_cache_load(const std::string& path, uint64_t print, size_t bytes, uint8_t* d, hipStream_t s)
{
    TrigCacheHeader hd;
    const int fd = trig_cache_open(path, print, bytes, &hd);
    if (fd < 0) return false;
    const size_t chunk = 8u << 20;
    uint8_t* h[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    unsigned long long* dSum = nullptr;
    bool ok = hipHostMalloc((void**)&h[0], chunk) == hipSuccess && hipHostMalloc((void**)&h[1], chunk) == hipSuccess &&
              hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess && hipMalloc((void**)&dSum, 8) == hipSuccess &&
              hipMemsetAsync(dSum, 0, 8, s) == hipSuccess;
    const size_t padded = (bytes + 7) & ~(size_t)7;
    if (ok && padded > bytes) ok = hipMemsetAsync(d + bytes, 0, padded - bytes, s) == hipSuccess;
    int k = 0;
    for (size_t off = 0; ok && off < bytes; off += chunk, k ^= 1) {
        const size_t n = std::min(chunk, bytes - off);
        if (off >= 2 * chunk) ok = hipEventSynchronize(ev[k]) == hipSuccess; // this buffer's previous DMA has finished
        ok = ok && read_fully(fd, h[k], n) && hipMemcpyAsync(d + off, h[k], n, hipMemcpyHostToDevice, s) == hipSuccess &&
             hipEventRecord(ev[k], s) == hipSuccess;
    }
    close(fd);
    unsigned long long sum = 0;
    if (ok) {
        hipLaunchKernelGGL(k_trig_checksum, dim3(2048), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(d),
                           (unsigned long long)(padded / 8), dSum);
        ok = hipMemcpyAsync(&sum, dSum, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess &&
             sum == hd.payloadSum;
    } else {
        (void)hipStreamSynchronize(s);
    }
    (void)hipGetLastError();
    for (int i = 0; i < 2; i++) {
        if (h[i]) (void)hipHostFree(h[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (dSum) (void)hipFree(dSum);
    if (!ok && getenv("ORBFE_VERBOSE")) fprintf(stderr, "orbfe: trig cache %s rejected (checksum or read error): rebuilding\n", path.c_str());
    return ok;
}
// Written under a temporary name that is created exclusively (O_EXCL | O_NOFOLLOW, mode 0600) and renamed: a reader sees all
// or nothing, a planted link or file under the temporary name makes the store fail instead of being followed.
bool trig_cache_store(const std::string& path, uint64_t print, const uint8_t* src, size_t bytes, uint64_t payloadSum)
{
    if (path.empty()) return false;
    char tmp[64];
    std::snprintf(tmp, sizeof tmp, ".tmp%ld", (long)getpid());
    const std::string t = path + tmp;
    (void)unlink(t.c_str()); // (a leftover of a crashed process with this pid)
    const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    TrigCacheHeader hd;
    std::memcpy(hd.magic, "ORBFETC4", 8);
    hd.u0 = ORBFE_TRIG_U0;
    hd.n = ORBFE_TRIG_U1 - ORBFE_TRIG_U0 + 1u;
    hd.libmPrint = print;
    hd.payloadSum = payloadSum;
    bool ok = fchmod(fd, 0600) == 0 && write(fd, &hd, sizeof hd) == (ssize_t)sizeof hd;
    for (size_t off = 0; ok && off < bytes;) {
        const ssize_t k = write(fd, src + off, std::min<size_t>(bytes - off, 16u << 20));
        ok = k > 0;
        off += ok ? (size_t)k : 0;
    }
    ok = close(fd) == 0 && ok;
    if (!ok || std::rename(t.c_str(), path.c_str()) != 0) {
        (void)unlink(t.c_str());
        return false;
    }
    return true;
}

// The libm table of this process for `device` (both pointers null: not available); never fails the caller.
// ORBFE_TRIG_TABLE = 0: none (per-batch host check), 1: compact 4-bit codes (65 MB; the device still evaluates the
// correctly rounded sin/cos and applies the code), 2: libm's values themselves (1.03 GB of the 288 GB; the descriptor
// kernel then needs no double-precision sin/cos at all), expanded ON THE DEVICE from the codes; falls back to the compact
// form when that allocation fails.  Unset: 2 for a process on its own, 1 when the process is one rank of several
// (WORLD_SIZE / LOCAL_WORLD_SIZE > 1: eight ranks then hold 8 x 65 MB instead of 8 x 1.03 GB, at 0.259 instead of 0.253 ms
// per 64-frame batch).  A libm that is ever further than one bit pattern from the correctly rounded value cannot be coded:
// then mode 2 uploads libm's values directly (the first design) and mode 1 is unavailable.
TrigTabs trig_table(int device, hipStream_t s)
{
    TrigTabs none{nullptr, nullptr};
    if (device < 0 || device >= 16) return none;
    std::lock_guard<std::mutex> lock(g_trigMutex);
    TrigTable& t = g_trig[device];
    if (t.tried) return t.ok ? TrigTabs{t.d, t.full} : none;
    t.tried = true;
    int mode = 2;
    for (const char* v : {"WORLD_SIZE", "LOCAL_WORLD_SIZE"})
        if (const char* e = getenv(v))
            if (atoi(e) > 1) mode = 1;
    if (const char* e = getenv("ORBFE_TRIG_TABLE")) mode = atoi(e);
    if (mode <= 0) return none;
    const auto tBuild0 = std::chrono::steady_clock::now();
    if (!small_angles_ok()) return none;
    const uint32_t N = ORBFE_TRIG_U1 - ORBFE_TRIG_U0 + 1u;
    const size_t codeBytes = (size_t)(N + 1) / 2;
    const uint32_t chunk = 1u << 22; // 4M angles = 32 MB of (cosf, sinf) per transfer
    float2* h = nullptr; // pinned staging, only when libm has to be evaluated
    // ---- the codes: from the cache file (mapped, sent as it lies), else from libm
    const uint64_t print = libm_fingerprint();
    const std::string cache = trig_cache_path(print);
    bool fine = hipMalloc((void**)&t.d, ((codeBytes + 7) & ~(size_t)7)) == hipSuccess; // (whole 8-byte words: the checksum kernel)
    bool coded = false;
    if (fine) coded = trig_cache_load(cache, print, codeBytes, t.d, s);
    if (!coded && hipHostMalloc((void**)&h, std::max((size_t)chunk * sizeof(float2), codeBytes)) != hipSuccess) {
        (void)hipGetLastError();
        if (t.d) (void)hipFree(t.d);
        t.d = nullptr;
        return none;
    }
    if (fine && !coded) {
        float2* dAB = nullptr;
        int32_t* dBad = nullptr;
        int32_t bad = 0;
        coded = hipMalloc((void**)&dAB, (size_t)chunk * sizeof(float2)) == hipSuccess &&
                hipMalloc((void**)&dBad, sizeof(int32_t)) == hipSuccess && hipMemsetAsync(dBad, 0, sizeof(int32_t), s) == hipSuccess;
        for (uint32_t i0 = 0; coded && i0 < N; i0 += chunk) { // chunk is even: whole table bytes per chunk
            const uint32_t n = std::min(chunk, N - i0);
            fill_libm(h, ORBFE_TRIG_U0 + i0, n);
            coded = hipMemcpyAsync(dAB, h, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, s) == hipSuccess;
            if (!coded) break;
            hipLaunchKernelGGL(k_trig_codes, dim3((n / 2 + 256) / 256), dim3(256), 0, s, dAB, ORBFE_TRIG_U0 + i0, n,
                               t.d + i0 / 2, dBad);
            coded = hipStreamSynchronize(s) == hipSuccess;
        }
        // a libm value further than one bit pattern from the correctly rounded one cannot be coded
        if (coded) coded = hipMemcpy(&bad, dBad, sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
        if (dAB) (void)hipFree(dAB);
        if (dBad) (void)hipFree(dBad);
        if (coded && !cache.empty() && hipMemcpy(h, t.d, codeBytes, hipMemcpyDeviceToHost) == hipSuccess)
            (void)trig_cache_store(cache, print, reinterpret_cast<const uint8_t*>(h), codeBytes,
                                   trig_payload_sum(reinterpret_cast<const uint8_t*>(h), codeBytes));
    }
    if (!coded && t.d) {
        (void)hipFree(t.d);
        t.d = nullptr;
    }
    (void)hipGetLastError();
    // ---- the full table
    fine = coded;
    if (mode >= 2) {
        if (hipMalloc((void**)&t.full, (size_t)N * sizeof(float2)) == hipSuccess) {
            bool filled;
            if (coded) { // expanded on the device from the codes
                hipLaunchKernelGGL(k_trig_expand, dim3((N + 255) / 256), dim3(256), 0, s, t.d, ORBFE_TRIG_U0, N, t.full);
                filled = hipStreamSynchronize(s) == hipSuccess;
            } else { // an uncodable libm: its values themselves, evaluated here and sent over
                filled = h != nullptr;
                for (uint32_t i0 = 0; filled && i0 < N; i0 += chunk) {
                    const uint32_t n = std::min(chunk, N - i0);
                    fill_libm(h, ORBFE_TRIG_U0 + i0, n);
                    filled = hipMemcpyAsync(t.full + i0, h, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, s) == hipSuccess &&
                             hipStreamSynchronize(s) == hipSuccess; // h is refilled next
                }
            }
            if (filled) {
                fine = true;
                if (t.d) (void)hipFree(t.d); // K-DESC reads the full table; the codes have done their work
                t.d = nullptr;
            } else {
                (void)hipFree(t.full);
                t.full = nullptr;
            }
        } else {
            (void)hipGetLastError(); // a failed 1-GB allocation is not an error of the caller: the compact form serves
            t.full = nullptr;
        }
    }
    if (h) (void)hipHostFree(h);
    t.ok = fine && (t.d || t.full);
    if (getenv("ORBFE_VERBOSE"))
        fprintf(stderr, "orbfe: libm trig table on device %d: %s, codes %s, %.1f ms\n", device,
                t.full ? "1.03 GB of values" : t.d ? "65 MB of codes" : "none",
                coded ? (cache.empty() ? "from libm (no cache)" : "from the cache file or libm") : "not codable",
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count());
    return t.ok ? TrigTabs{t.d, t.full} : none;
}

inline void rec(orbfe_ctx* c, int i)
{
    if (c->recNow)
        (void)hipEventRecord(c->ev[(size_t)(c->profCalls % orbfe_ctx::kProfSets) * (ORBFE_STAGE_COUNT + 1) + i],
                             c->stream);
}

// The whole pipeline on the context's stream.  All pointers are device pointers.
int run_device(orbfe_ctx* c, int nimg, const uint8_t* d_imgs, int rows, int cols, size_t pitch, size_t imgStride,
               const int32_t* d_lap, float* d_kps, uint8_t* d_desc, int capPerImg, int32_t* d_n, int32_t* d_mono,
               int32_t* d_errOut = nullptr /* K-DESC (or K-PACK) copies the batch's error word here (host-pointer path) */,
               uint8_t* mirror = nullptr /* pinned host slab [meta | keypoints | descriptors] the kernels write the results
                                            into as well (device-side address; latency path of a frame or two) */,
               size_t mirrorMetaBytes = 0, bool allowLanes = false /* the caller tolerates results that are only ordered on the
                                            context's stream after lane_join (orbfe_extract_batch_device) */)
{
    int r;
    const int lapInlineN = c->lapInlineN; // (a host-path call of <= 2 images: its lapping ranges go to K-QT by value)
    c->lapInlineN = 0;
    if ((r = ensure_geometry(c, rows, cols, nimg)) < 0) return r;
    if (capPerImg < c->maxKp || capPerImg > 65535 || nimg > 32767) return ORBFE_ERR_ARGS; // fix-list packing
    if ((r = ensure_capacity(c, nimg, capPerImg)) < 0) return r;
    hipStream_t s = run_stream(c); // (a batch lane's stream when orbfe_extract_batch_device deals the call to one)
    const int nl = c->nlevels;
    // ORBFE_TRIG_LIBM: with the libm table the device reproduces host cosf/sinf by itself; without it the
    // fragile keypoints are listed and checked on the host after the batch
    const TrigTabs trigTab = c->trigMode == ORBFE_TRIG_LIBM ? trig_table(c->device, s) : TrigTabs{nullptr, nullptr};
    const bool hostTrigCheck = c->trigMode != ORBFE_TRIG_CR && !trigTab.codes && !trigTab.full;
    if (c->tapsDirty && lanes_busy(c)) lane_quiesce(c); // (the shared tap words are about to change under the other lanes' K-DESC)
    if (c->tapsDirty) { // 28 bytes, but a separate command on the stream: only when they changed
        const uint32_t t0 = (uint32_t)c->taps[0], t1 = (uint32_t)c->taps[1], t2 = (uint32_t)c->taps[2], t3 = (uint32_t)c->taps[3],
                       t4 = (uint32_t)c->taps[4], t5 = (uint32_t)c->taps[5], t6 = (uint32_t)c->taps[6];
        uint32_t* const w = c->tapWords;
        for (int i = 0; i < 7; i++) w[i] = (uint32_t)c->taps[i];
        w[7] = 0;
        // horizontal pass: byte taps of output j of a group against the three aligned dwords of its row (shifted TAPS)
        w[8] = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);
        w[9] = t4 | (t5 << 8) | (t6 << 16);
        w[10] = (t0 << 8) | (t1 << 16) | (t2 << 24);
        w[11] = t3 | (t4 << 8) | (t5 << 16) | (t6 << 24);
        w[12] = (t0 << 16) | (t1 << 24);
        w[13] = t2 | (t3 << 8) | (t4 << 16) | (t5 << 24);
        w[14] = t6;
        w[15] = t0 << 24;
        w[16] = t1 | (t2 << 8) | (t3 << 16) | (t4 << 24);
        w[17] = t5 | (t6 << 8);
        // vertical pass: u16 tap pairs of the even / odd output row of a pair
        w[18] = t0 | (t1 << 16);
        w[19] = t2 | (t3 << 16);
        w[20] = t4 | (t5 << 16);
        w[21] = t6;
        w[22] = t0 << 16;
        w[23] = t1 | (t2 << 16);
        w[24] = t3 | (t4 << 16);
        w[25] = t5 | (t6 << 16);
        HIP_TRY(hipMemcpyAsync(c->d_taps.p, w, 26 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        c->tapsDirty = false;
    }
    if (c->kb8On) HIP_TRY(hipMemcpyAsync(c->d_kb8.p, c->kb8, 8 * sizeof(float), hipMemcpyHostToDevice, s));
    // 16-B header of the fix list = {fragile count, error flag, 0, 0}.  The fused pyramid kernel clears it
    // (first command of the batch on this stream); the other configurations need a memset command.
    int32_t* const d_hdr = reinterpret_cast<int32_t*>(c->d_fix.p);
    (void)allowLanes; // (round 4's split of one call into two half-batches left with round 6; lanes are whole batches per stream)
    const bool kernelClearsHdr = c->pyrFused;
    if (!kernelClearsHdr) HIP_TRY(hipMemsetAsync(c->d_fix.p, 0, sizeof(int4), s));
    c->recNow = !c->runStream && c->profile && c->evReady && c->profSeen % c->profEvery == 0; // (stage events: one stream only)
    rec(c, 0);
    const int nsub = 1;
    {
        const int k = 0, i0 = 0, ni = nimg;
        hipStream_t q = s;
        int32_t* const d_hdrK = d_hdr;
        // K-FAST's order: whole images per XCD when the batch fills the 8 XCDs evenly enough (<= 1/8 idle), else groups of
        // G neighbouring cells per XCD
        const int perXcd = (ni + 7) / 8;
        const bool byImage = c->fastXcdGroup <= 0 ? true : (c->xcdAffine && c->fastByImage && ni >= 8 && perXcd * 8 - ni <= ni / 8);
        // K-PYR
        if (c->pyrFused) {
            hipLaunchKernelGGL(k_pyr_fused, dim3((unsigned)c->pyrNtx, (unsigned)c->pyrNty, (unsigned)ni), dim3(256),
                               c->pyrLdsBytes, q, d_imgs,
                               pitch, imgStride, c->d_pyr.p, c->pyrStride, c->d_pyrRecs.p, c->pyrRecBytes, nl,
                               c->pyrNtx, c->pyrNty, c->pyrBuf0, c->pyrBuf1, c->pyrStageX,
                               cols, i0, kernelClearsHdr ? d_hdrK : nullptr, (c->xcdAffine && ni % 8 == 0) ? 1 : 0,
                               recip32((unsigned)(c->pyrNtx * c->pyrNty)), recip32((unsigned)c->pyrNtx));
        } else {
            const OrbLevelGeom& L0 = c->lg[0];
            dim3 grid((unsigned)((L0.w + 1023) / 1024), (unsigned)L0.h, (unsigned)nimg);
            hipLaunchKernelGGL(k_pyr_level0, grid, dim3(256), 0, q, d_imgs, pitch, imgStride, c->d_pyr.p, c->pyrStride,
                               L0);
            for (int l = 1; l < nl; l++) {
                const OrbLevelGeom& Ld = c->lg[l];
                dim3 g2((unsigned)((Ld.w + 1023) / 1024), (unsigned)Ld.h, (unsigned)nimg);
                hipLaunchKernelGGL(k_pyr_resize, g2, dim3(256), 0, q, c->d_pyr.p, c->pyrStride, c->lg[l - 1], Ld,
                                   c->d_xtab.p, c->d_ytab.p);
            }
        }
        if (nsub == 1) rec(c, 1);
        if (c->guardEv && k == nsub - 1) HIP_TRY(hipEventRecord(c->guardEv, q)); // (the caller's images have been read)
        // K-FAST
        {
            // the image group is the grid's y coordinate (no division in the kernel); cell groups per XCD are powers of two
            int gShift = -1;
            if (!byImage)
                for (gShift = 0; (2 << gShift) <= std::max(1, c->fastXcdGroup); gShift++) {}
            const int G2 = byImage ? 0 : 1 << gShift;
            const dim3 grid = byImage ? dim3((unsigned)(8 * c->nCells), (unsigned)perXcd)
                                      : dim3((unsigned)(((c->nCells + 8 * G2 - 1) / (8 * G2)) * 8 * G2), (unsigned)ni);
#define ORBFE_FAST_LAUNCH(NT, PD)                                                                                       \
    hipLaunchKernelGGL((k_fast_cells<NT, PD>), grid, dim3(NT), c->fastLdsBytes, q, c->d_pyr.p, c->pyrStride, c->d_fc.p,   \
                       c->d_cand.p, c->candStride, c->d_cellCount.p, c->nCells, c->iniThFAST, c->minThFAST,            \
                       c->fastTileBytes, c->fastBmWords, gShift, i0, ni, c->d_fastPat.p)
#define ORBFE_FAST_PD(NT)                                \
    do {                                                 \
        if (c->fastPitch == 52) ORBFE_FAST_LAUNCH(NT, 13); \
        else if (c->fastPitch == 68) ORBFE_FAST_LAUNCH(NT, 17); \
        else ORBFE_FAST_LAUNCH(NT, 21);                  \
    } while (0)
            if (c->fastThreads == 64) ORBFE_FAST_PD(64);
            else if (c->fastThreads == 128) ORBFE_FAST_PD(128);
            else ORBFE_FAST_PD(256);
#undef ORBFE_FAST_PD
#undef ORBFE_FAST_LAUNCH
        }
        if (nsub == 1) rec(c, 2);
        // K-QT
        {
            OrbQtLevels lv = {};
            for (size_t i = 0; i < c->qtSmall.size(); i++) lv.v[i] = c->qtSmall[i];
            OrbLapInline lapIn = {}; // (the host path of a frame or two hands the ranges over by value)
            if (lapInlineN > 0 && i0 == 0) {
                lapIn.n = lapInlineN;
                for (int i = 0; i < 4; i++) lapIn.v[i] = c->lapInline[i];
            }
            const unsigned nS = (unsigned)c->qtSmall.size(), nB = (unsigned)c->qtBig.size();
            if (nS)
                hipLaunchKernelGGL((k_octree<false, QT_THREADS>), ORBFE_QT_IMG_MAJOR ? dim3((unsigned)ni, nS) : dim3(nS, (unsigned)ni),
                                   dim3(QT_THREADS), c->qtLdsBytes, q, c->d_lg.p,
                                   c->d_cg.p, c->d_cand.p, c->candStride, c->d_cellCount.p, c->nCells, c->d_keys.p,
                                   c->d_keyNode.p, c->keyStride, c->d_lvlKp.p, c->kpStride, c->d_lvlCount.p, nl, d_hdrK + 1, i0,
                                   c->qtKeyOff, c->qtKeyCap, d_lap, c->d_lvlPre.p, lv, (int*)nullptr, (size_t)0, lapIn);
            if (nB) { // levels whose node tables exceed the LDS: same kernel on a global scratch area (i0-relative slices)
                OrbQtLevels lb = {};
                for (size_t i = 0; i < c->qtBig.size(); i++) lb.v[i] = c->qtBig[i];
                hipLaunchKernelGGL((k_octree<true, QT_THREADS>), ORBFE_QT_IMG_MAJOR ? dim3((unsigned)ni, nB) : dim3(nB, (unsigned)ni),
                                   dim3(QT_THREADS), c->qtBigLdsBytes, q, c->d_lg.p,
                                   c->d_cg.p, c->d_cand.p, c->candStride, c->d_cellCount.p, c->nCells, c->d_keys.p,
                                   c->d_keyNode.p, c->keyStride, c->d_lvlKp.p, c->kpStride, c->d_lvlCount.p, nl, d_hdrK + 1, i0,
                                   c->qtBigKeyOff, c->qtBigKeyCap, d_lap, c->d_lvlPre.p, lb,
                                   c->d_qtScratch.p + (size_t)i0 * c->qtBig.size() * c->qtScratchStride, c->qtScratchStride, lapIn);
            }
        }
        if (nsub == 1) rec(c, 3);
        // K-PACK: only when bearing rays are wanted (orbfe_set_kb8).  Otherwise K-QT has left the mono / stereo partition
        // of every level behind and K-DESC derives its output slots, the keypoint records, the counts and the error word
        // itself: four launches per batch
        const bool needPack = c->kb8On;
        int32_t* const mMeta = mirror ? reinterpret_cast<int32_t*>(mirror) : nullptr;
        OrbLapInline lapInPack = {};
        if (lapInlineN > 0 && i0 == 0) {
            lapInPack.n = lapInlineN;
            for (int i = 0; i < 4; i++) lapInPack.v[i] = c->lapInline[i];
        }
        if (needPack)
            hipLaunchKernelGGL(k_pack, dim3((unsigned)ni), dim3(PACK_THREADS), 0, q, c->d_lg.p, nl, c->d_lvlKp.p, c->kpStride,
                               c->d_lvlCount.p, d_lap, capPerImg, c->d_destMap.p, d_n, d_mono,
                               c->kb8On ? c->d_kb8.p : nullptr,
                               c->kb8On ? (c->userRays ? c->userRays : c->d_rays.p) : nullptr, i0,
                               k == 0 ? d_hdr + 1 : nullptr, k == 0 ? d_errOut : nullptr, hostTrigCheck ? mMeta : nullptr, nimg,
                               lapInPack); // (the mirror's meta words: k_mirror_out's unless the trig check keeps the old form)
        if (nsub == 1 && needPack) rec(c, 4); // (without K-PACK no event separates K-QT from K-DESC: a record costs ~3.5 us)
        if (c->recNow) c->packSkipped[c->profCalls % orbfe_ctx::kProfSets] = !(nsub == 1 && needPack);
        // K-DESC
        {
            int tapSum = 0;
            for (int i = 0; i < 7; i++) tapSum += c->taps[i];
            float* const mKps = mirror ? reinterpret_cast<float*>(mirror + mirrorMetaBytes) : nullptr;
            uint8_t* const mDesc = mirror ? mirror + mirrorMetaBytes + (size_t)nimg * capPerImg * 28 : nullptr;
            // whole images per XCD: (8 x workgroups per image, images / 8), image = workgroup id mod 8 + 8 y
            const bool descAffine = c->xcdAffine && ni % 8 == 0;
            const unsigned descWg = (unsigned)((c->maxKp + ORBFE_DESC_WPW * ORBFE_DESC_KPW - 1) / (ORBFE_DESC_WPW * ORBFE_DESC_KPW));
            const dim3 descGrid = descAffine ? dim3(8u * descWg, (unsigned)(ni / 8)) : dim3(descWg, (unsigned)ni);
            // Round 6: the results of a frame or two reach the page-locked slab through a copy kernel BEHIND K-DESC (k_mirror_out:
            // one workgroup, whole 16-byte rows of 64 lanes) instead of by K-DESC's own stores -- a thousand wavefronts each writing
            // 28 + 32 bytes across PCIe made the single frame's K-DESC 28 us against 8 resident (rocprofv3 of tools/hostbench,
            // profiles/r06_hostbench_kernel_stats.csv).  The copy kernel is then the call's last kernel and publishes the completion
            // word: every workgroup's stores have landed before it counts itself, the last one to count writes the flag.
            // (With the host-side trig check K-DESC keeps writing the mirror itself, as before: a fix-up launch follows it.)
            // With K-PACK (fisheye rays) in front the copy kernel takes [n | mono | err] from where K-PACK left them.
            const bool copyOut = mirror && !hostTrigCheck;
            OrbDone copyDone{nullptr, nullptr, 0u, 0u};
            if (c->doneWant) {
                c->doneWant = false;
                if (copyOut && c->spinWait && c->d_done.p && c->h_done.p && c->h_done.coherent) {
                    if (++c->doneSeq >= 0x80000000u) c->doneSeq = 1u; // (bit 31 of the word: "finished, not vouched for")
                    copyDone = OrbDone{c->d_done.p + 40, c->h_done.dev(), c->doneSeq, 1u}; // (a counter of its own among the 80)
                    c->doneGot = c->doneSeq; // (what the caller waits for)
                }
            }
            int32_t* const kMeta = (needPack || copyOut) ? nullptr : mMeta;
            float* const kKps = copyOut ? nullptr : mKps;
            uint8_t* const kDesc = copyOut ? nullptr : mDesc;
#define ORBFE_DESC_LAUNCH(M, SAT)                                                                                         \
    hipLaunchKernelGGL((k_orient_blur_desc<M, SAT>), descGrid,                                                            \
                       dim3(64 * ORBFE_DESC_WPW), 0, q,                                                                   \
                       c->d_pyr.p, c->pyrStride, c->d_ds.p, c->maxKp, c->d_lvlKp.p, c->kpStride, c->d_lvlCount.p, nl,     \
                       c->d_lvlPre.p, needPack ? c->d_destMap.p : nullptr, capPerImg, d_kps, d_desc, c->d_taps.p, c->d_patternF.p,       \
                       c->d_fix.p, 0, hostTrigCheck ? 1 : 0, i0, descAffine ? 1 : 0, trigTab.codes,                          \
                       trigTab.full, c->atanFma, nullptr, 0, d_n, d_mono, k == 0 ? d_hdr + 1 : nullptr,                    \
                       k == 0 ? d_errOut : nullptr, kMeta, nimg, kKps, kDesc)
            if (hostTrigCheck) { // (the listing of fragile keypoints is an instantiation of its own)
                if (tapSum > 256) ORBFE_DESC_LAUNCH(2, true);
                else ORBFE_DESC_LAUNCH(2, false);
            } else {
                if (tapSum > 256) ORBFE_DESC_LAUNCH(0, true);
                else ORBFE_DESC_LAUNCH(0, false);
            }
#undef ORBFE_DESC_LAUNCH
            if (copyOut)
                hipLaunchKernelGGL(k_mirror_out, dim3(ORBFE_MIRROR_WGS), dim3(256), 0, q, reinterpret_cast<const uint8_t*>(d_n), reinterpret_cast<const uint8_t*>(d_kps),
                                   d_desc, mirror, (unsigned)mirrorMetaBytes, nimg, capPerImg, copyDone);
        }
    }
    rec(c, 5);
    c->lastImgs = nimg;
    c->lastKps = d_kps;
    c->lastDesc = d_desc;
    c->lastN = d_n;
    c->lastCap = capPerImg;
    c->lastPacked = c->kb8On;
    c->lastFixups = 0;
    c->lastHostTrigCheck = hostTrigCheck;
    if (hostTrigCheck) {
        // Trig fix-up: the device used the correctly rounded sin/cos and listed the keypoints whose
        // sampling grid is within a rounding hair of changing.  Evaluate the host libm cosf/sinf
        // (what the reference calls, src/ORBextractor.cc:111) for those; where libm differs from the
        // correctly rounded value, re-run the descriptor on the device with libm's (a, b).
        // one D2H covers the header (count, error word) and the first 1023 entries (normally all of them)
        const size_t total = (size_t)c->capImgs * c->capKp;
        const size_t first = std::min<size_t>(1023, total);
        HIP_TRY(hipMemcpyAsync(c->h_fix.p, c->d_fix.p, (1 + first) * sizeof(int4), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (c->h_fix.p[0].y != 0) return ORBFE_ERR_STATE;
        const size_t nFrag = std::min<size_t>((size_t)c->h_fix.p[0].x, total);
        if (nFrag > first) {
            HIP_TRY(hipMemcpyAsync(c->h_fix.p + 1 + first, c->d_fix.p + 1 + first, (nFrag - first) * sizeof(int4),
                                   hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        int nFix = 0;
        for (size_t i = 0; i < nFrag; i++) {
            const int4 e = c->h_fix.p[1 + i]; // {img<<16|g, angle, device cos, device sin} as float bits
            float angDeg, ac, bc;
            std::memcpy(&angDeg, &e.y, 4);
            std::memcpy(&ac, &e.z, 4);
            std::memcpy(&bc, &e.w, 4);
            const float ang = angDeg * factorPI;
            const float a = cosf(ang), b = sinf(ang); // what the reference evaluates (src/ORBextractor.cc:111)
            if (a != ac || b != bc) {
                int4 o = e;
                std::memcpy(&o.z, &a, 4);
                std::memcpy(&o.w, &b, 4);
                c->h_fixAB.p[nFix++] = o;
            }
        }
        if (nFix > 0) // the kernel reads the pinned list in place
            hipLaunchKernelGGL((k_orient_blur_desc<1, true>), dim3((unsigned)((nFix + ORBFE_DESC_WPW - 1) / ORBFE_DESC_WPW)),
                               dim3(64 * ORBFE_DESC_WPW), 0, s, c->d_pyr.p,
                               c->pyrStride, c->d_ds.p, c->maxKp, c->d_lvlKp.p, c->kpStride, c->d_lvlCount.p, nl,
                               c->d_lvlPre.p, c->kb8On ? c->d_destMap.p : nullptr, capPerImg, d_kps, d_desc, c->d_taps.p,
                               c->d_patternF.p, c->h_fixAB.p, nFix, 0, 0, 0, nullptr, nullptr, c->atanFma, nullptr, 0, nullptr,
                               nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                               mirror ? mirror + mirrorMetaBytes + (size_t)nimg * capPerImg * 28 : nullptr);
        c->lastFixups = nFix;
    }
    rec(c, 6);
    if (c->recNow) c->profCalls++;
    if (c->profile) c->profSeen++;
    c->recNow = false;
    HIP_TRY(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// Host-pointer path: pinned memory registry, staging-copy pool, the two-slot pipeline.
//
// What a DMA engine can read or write directly is page-locked ("pinned") host memory; a transfer from or to
// pageable memory is staged by the runtime through an internal bounce buffer on the calling thread, at a fraction
// of the PCIe rate, and blocks.  The boundary therefore (i) recognises caller buffers that are pinned -- allocated
// with orbfe_host_alloc, registered with orbfe_host_register, or pinned by anyone else (hipHostMalloc, torch's
// pin_memory: asked from the runtime once per pointer and remembered) -- and moves them with one DMA command, and
// (ii) stages pageable buffers itself, through pinned memory the slot owns, with several host threads and in
// chunks, so that the DMA of one chunk runs while the next is staged.
struct PinRange {
    const uint8_t* p;
    size_t n;
    bool owned;
};
std::mutex g_pinMutex;
std::vector<PinRange> g_pins;

void pin_add(const void* p, size_t n, bool owned)
{
    std::lock_guard<std::mutex> lock(g_pinMutex);
    g_pins.push_back(PinRange{(const uint8_t*)p, n, owned});
}
bool pin_remove(const void* p)
{
    std::lock_guard<std::mutex> lock(g_pinMutex);
    for (size_t i = 0; i < g_pins.size(); i++)
        if (g_pins[i].p == (const uint8_t*)p) {
            g_pins.erase(g_pins.begin() + (long)i);
            return true;
        }
    return false;
}
bool pin_known(const void* p, size_t n)
{
    std::lock_guard<std::mutex> lock(g_pinMutex);
    for (const PinRange& r : g_pins)
        if ((const uint8_t*)p >= r.p && (const uint8_t*)p + n <= r.p + r.n) return true;
    return false;
}
// Is [p, p+n) page-locked?  Registry first; otherwise the runtime is asked once per pointer value and the answer is
// cached in the context (a camera driver reuses its buffers).  A stale answer is harmless: a copy from memory
// wrongly believed pinned is staged by the runtime, one wrongly believed pageable is staged here.
bool is_pinned(orbfe_ctx* c, const void* p, size_t n)
{
    if (pin_known(p, n)) return true;
    auto it = c->pinnedCache.find(p);
    if (it != c->pinnedCache.end()) return it->second;
    hipPointerAttribute_t a;
    bool pinned = false;
    if (hipPointerGetAttributes(&a, p) == hipSuccess) pinned = a.type == hipMemoryTypeHost;
    else (void)hipGetLastError(); // plain malloc'ed memory is "invalid value" for older runtimes: not an error here
    if (c->pinnedCache.size() > 8192) c->pinnedCache.clear();
    c->pinnedCache[p] = pinned;
    return pinned;
}

// orbfe_set_auto_register: a caller that hands over the same pageable buffer again (a camera driver's ring, a preallocated
// cv::Mat) gets it page-locked on the second sighting.  Buffers seen once stay pageable (registering costs more than one
// staged copy), at most 16 registrations live per context.
bool auto_pin(orbfe_ctx* c, const void* p, size_t n)
{
    if (!c->autoRegister || !p || !n) return false;
    c->autoClock++;
    auto unpin = [&](orbfe_ctx::AutoPin& o) { // (nothing may still be reading it: the range is published process-wide, so another
        // context -- the right extractor fed from the same driver ring -- may have a copy or an upload kernel in flight on it;
        // evictions are rare, so the whole device is waited for: ADVICE r03)
        (void)hipDeviceSynchronize();
        if (pin_remove(o.p)) (void)hipHostUnregister(const_cast<void*>(o.p));
        c->pinnedCache.erase(o.p);
        o.registered = false;
    };
    long idx = -1;
    for (size_t i = 0; i < c->autoPins.size(); i++)
        if (c->autoPins[i].p == p && c->autoPins[i].n == n) idx = (long)i;
    if (idx < 0) { // first sighting: remember it, evicting the least recently used entry of a full table
        if (c->autoPins.size() >= 16) {
            size_t lru = 0;
            for (size_t i = 1; i < c->autoPins.size(); i++)
                if (c->autoPins[i].used < c->autoPins[lru].used) lru = i;
            if (c->autoPins[lru].registered) unpin(c->autoPins[lru]);
            c->autoPins.erase(c->autoPins.begin() + (long)lru);
        }
        c->autoPins.push_back(orbfe_ctx::AutoPin{p, n, false, c->autoClock});
        return false;
    }
    c->autoPins[(size_t)idx].used = c->autoClock;
    if (c->autoPins[(size_t)idx].registered) return true;
    // second sighting.  Ranges of ours that overlap this one (single images of a buffer that now arrives whole, or the
    // reverse) are given up first: a range must not be registered twice.
    const uint8_t* lo = (const uint8_t*)p;
    for (long i = (long)c->autoPins.size() - 1; i >= 0; i--) {
        orbfe_ctx::AutoPin& o = c->autoPins[(size_t)i];
        if (i == idx || !((const uint8_t*)o.p < lo + n && lo < (const uint8_t*)o.p + o.n)) continue;
        if (o.registered) unpin(o);
        c->autoPins.erase(c->autoPins.begin() + i);
        if (i < idx) idx--;
    }
    if (hipHostRegister(const_cast<void*>(p), n, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError(); // (registered by its owner already, or not registrable: the staging path takes it)
        return false;
    }
    pin_add(p, n, false);
    c->pinnedCache.erase(p);
    c->autoPins[(size_t)idx].registered = true;
    return true;
}
void auto_pin_release(orbfe_ctx* c)
{
    bool any = false;
    for (const orbfe_ctx::AutoPin& a : c->autoPins) any = any || a.registered;
    if (any) (void)hipDeviceSynchronize(); // (other contexts may be reading the ranges: see auto_pin)
    for (orbfe_ctx::AutoPin& a : c->autoPins)
        if (a.registered && pin_remove(a.p)) (void)hipHostUnregister(const_cast<void*>(a.p));
    c->autoPins.clear();
}

// A few persistent host threads for the staging copies of pageable caller memory (memcpy from one thread moves
// ~10 GB/s, a batch of 64 frames is 23 MB in and 4 MB out).  run(n, f) executes f(0..n-1) on the workers and the
// caller and returns when all are done.  One run at a time; a second concurrent caller does its work inline.
class CopyPool {
public:
    static CopyPool& get()
    {
        static CopyPool* p = new CopyPool(); // never destroyed: the workers are detached and die with the process
        return *p;
    }
    void run(int n, const std::function<void(int)>& f)
    {
        if (n <= 0) return;
        std::unique_lock<std::mutex> only(runM_, std::try_to_lock);
        if (!only.owns_lock() || nWorkers_ == 0 || n == 1) {
            for (int i = 0; i < n; i++) f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &f;
            ntasks_ = n;
            next_.store(0);
            checkedIn_ = 0;
            gen_++;
        }
        cvWork_.notify_all();
        for (int i; (i = next_.fetch_add(1)) < n;) f(i);
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [this] { return checkedIn_ == nWorkers_; }); // no worker touches fn_ after this
        fn_ = nullptr;
    }

private:
    CopyPool()
    {
        int T = std::min(8, host_threads()) - 1;
        if (const char* e = getenv("ORBFE_COPY_THREADS")) T = std::min(15, std::max(0, atoi(e) - 1));
        for (int t = 0; t < T; t++) {
            std::thread([this] { worker(); }).detach();
            nWorkers_++;
        }
    }
    void worker()
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            int n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cvWork_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                f = fn_;
                n = ntasks_;
            }
            for (int i; (i = next_.fetch_add(1)) < n;) (*f)(i);
            {
                std::lock_guard<std::mutex> lk(m_);
                checkedIn_++;
            }
            cvDone_.notify_one();
        }
    }
    std::mutex runM_, m_;
    std::condition_variable cvWork_, cvDone_;
    const std::function<void(int)>* fn_ = nullptr;
    int ntasks_ = 0, nWorkers_ = 0, checkedIn_ = 0;
    unsigned long gen_ = 0;
    std::atomic<int> next_{0};
};

inline void copy_rows(uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, int rows)
{
    if (dpitch == width && spitch == width) {
        std::memcpy(dst, src, width * (size_t)rows);
        return;
    }
    for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * dpitch, src + (size_t)y * spitch, width);
}

int slot_prepare(orbfe_ctx* c, orbfe_ctx::HostSlot& sl, bool pipelined)
{
    if (!sl.evDone) {
        HIP_TRY(hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.evK, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
    }
    if (pipelined && !c->sIn) {
        HIP_TRY(hipStreamCreateWithFlags(&c->sIn, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->sOut, hipStreamNonBlocking));
    }
    return 0;
}

// Upload of one image out of page-locked host memory by a kernel (latency path).  Returns the device address the image
// starts at (the destination carries the source's offset inside a 16-byte block), or nullptr when the runtime has no
// device-side address for the host pointer (the caller then uses a copy command).
const uint8_t* upload_by_kernel(hipStream_t s, uint8_t* d_dst /* 256-B aligned, 32 bytes of slack */, const uint8_t* h_src,
                                size_t bytes, uint8_t* d_dst2 = nullptr, const uint8_t* h_src2 = nullptr /* a second image of the
                                same size in the same launch (both stream at once); must sit at the same offset inside its
                                16-byte block as the first, else nothing is launched and nullptr returned */)
{
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, const_cast<uint8_t*>(h_src), 0) != hipSuccess || !dv) {
        (void)hipGetLastError();
        return nullptr;
    }
    const size_t mis = (size_t)((uintptr_t)dv & 15u);
    const unsigned n16 = (unsigned)((mis + bytes + 15) / 16);
    OrbUploadSegs segs = {};
    segs.src[0] = reinterpret_cast<const orbfe_u4v*>((const uint8_t*)dv - mis);
    segs.dst[0] = reinterpret_cast<orbfe_u4v*>(d_dst);
    segs.n16[0] = n16;
    unsigned nseg = 1;
    if (h_src2) {
        void* dv2 = nullptr;
        if (hipHostGetDevicePointer(&dv2, const_cast<uint8_t*>(h_src2), 0) != hipSuccess || !dv2 ||
            (size_t)((uintptr_t)dv2 & 15u) != mis) {
            (void)hipGetLastError();
            return nullptr;
        }
        segs.src[1] = reinterpret_cast<const orbfe_u4v*>((const uint8_t*)dv2 - mis);
        segs.dst[1] = reinterpret_cast<orbfe_u4v*>(d_dst2);
        segs.n16[1] = n16;
        nseg = 2;
    }
    const unsigned grid = std::min(512u / nseg, (n16 + 255u) / 256u);
    hipLaunchKernelGGL(k_upload, dim3(grid, nseg), dim3(256), 0, s, segs);
    return d_dst + mis;
}

// Queue one host-pointer batch: H2D of the images, the four kernels, D2H of the results.  `pipelined` puts the
// copies on their own streams (ordered by events) so that they overlap the kernels of the neighbouring batches;
// the blocking calls keep everything on the context's stream (no event traffic on the latency path).
int host_submit_impl(orbfe_ctx* c, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride, const int* lap,
                     orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out, bool pipelined, bool spinOk,
                     int laneSlot = -1 /* >= 0: orbfe_extract_stereo_pair_submit -- slot and lane of that index, the lane's buffers
                                          are in the context and c->runStream is the lane's stream; no FIFO bookkeeping here */)
{
    if (!c || nimg < 1 || !imgs || !kps || !desc || !n_out) return ORBFE_ERR_ARGS;
    for (int i = 0; i < nimg; i++) {
        n_out[i] = 0;
        if (mono_out) mono_out[i] = 0;
    }
    if (rows <= 0 || cols <= 0) return -1;
    for (int i = 0; i < nimg; i++)
        if (!imgs[i]) return -1;
    if (stride < (size_t)cols) return ORBFE_ERR_ARGS;
    if (laneSlot < 0 && c->slotSubmitted - c->slotRetired >= 2) return ORBFE_ERR_STATE; // both slots in flight: wait for one first
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if (laneSlot < 0 && (r = lane_join(c)) < 0) return r;
    if ((r = ensure_geometry(c, rows, cols, nimg)) < 0) return r;
    if (cap_per_img < c->maxKp) return ORBFE_ERR_ARGS;
    if ((r = ensure_capacity(c, nimg, cap_per_img)) < 0) return r;
    orbfe_ctx::HostSlot& sl = c->slot[laneSlot >= 0 ? laneSlot : (int)(c->slotSubmitted & 1)];
    if ((r = slot_prepare(c, sl, pipelined)) < 0) return r;
    hipStream_t s = run_stream(c);
    hipStream_t sIn = pipelined ? c->sIn : s, sOut = pipelined ? c->sOut : s;
    const size_t Kc = (size_t)cap_per_img;
    const size_t imgBytes = (size_t)(rows - 1) * stride + (size_t)cols; // what may be read behind an image pointer

    // ---- lapping ranges: written into pinned memory K-QT reads in place
    if (sl.h_lap.n < (size_t)2 * nimg) {
        if ((r = sl.h_lap.ensure((size_t)2 * std::max(nimg, 64))) < 0) return r;
        sl.d_lapAlias = sl.h_lap.dev();
    }
    for (int i = 0; i < 2 * nimg; i++) sl.h_lap.p[i] = lap ? lap[i] : 0;

    // ---- images
    if (c->autoRegister) { // a batch whose images sit evenly spaced in one buffer is one range, else one range per image
        bool oneBuffer = nimg > 1;
        for (int i = 1; i < nimg && oneBuffer; i++) oneBuffer = imgs[i] - imgs[i - 1] == imgs[1] - imgs[0] && imgs[1] > imgs[0];
        if (oneBuffer && (size_t)(imgs[1] - imgs[0]) <= 2 * align_up(imgBytes, 256)) {
            if (!pin_known(imgs[0], (size_t)(nimg - 1) * (size_t)(imgs[1] - imgs[0]) + imgBytes))
                (void)auto_pin(c, imgs[0], (size_t)(nimg - 1) * (size_t)(imgs[1] - imgs[0]) + imgBytes);
        } else {
            for (int i = 0; i < nimg; i++)
                if (!pin_known(imgs[i], imgBytes)) (void)auto_pin(c, imgs[i], imgBytes);
        }
    }
    bool allPinned = true;
    for (int i = 0; i < nimg && allPinned; i++) allPinned = is_pinned(c, imgs[i], imgBytes);
    size_t devPitch, devStride;
    // Latency path (a frame or a stereo pair per blocking call -- how ORB-SLAM3 itself calls the extractor): the RESULTS need
    // no copy command.  K-DESC writes them into the slot's pinned slab as well as into HBM (posted writes over
    // PCIe, 60 KB per frame), which takes the download command and its ~10 us of latency off the end of the call: 0.089 ->
    // 0.080 ms per pinned frame, 0.177 -> 0.157 ms per stereo pair in one call.  (The same idea for the INPUT -- K-PYR
    // reading the image over PCIe where it lies in page-locked memory -- was measured and dropped: 0.089 -> 0.106 ms, reads
    // across the link stall the kernel far longer than the upload command costs.)
    // (not with sub-batches on several streams, ORBFE_STREAMS > 1: the mirrored error word is written by the first sub-batch's
    // K-DESC, before the other sub-batches' K-QT have run -- such a call takes the download command, which is queued behind
    // the join of all sub-batches)
    const bool mirrorOut = c->zeroCopy && !pipelined && nimg <= c->mirrorMaxImgs;
    // ... and the IMAGES of such a call come in through a kernel that reads the page-locked source in 16-byte pieces
    // (k_upload): no copy engine, hence no queue hand-over, between the host call and the first kernel.
    // (Measured and not kept: staging and uploading a pageable image band by band so that the upload of one band overlaps
    // the host copy of the next -- 2 / 3 / 4 bands: -1 / +2 / +10 us per frame, +7..13 us per stereo pair: the extra launches
    // cost the host more than the overlap returns.  Round 5, a pair of 1024 x 1024 images staged and uploaded image by image: the
    // same, 0.327-0.370 against 0.309-0.322 ms per fisheye stereo frame, tools/r05_c5d.sh.)
    const bool kernelIn = c->uploadKernel && !pipelined && nimg <= c->mirrorMaxImgs;
    const uint8_t* d_imgBase = nullptr; // where image 0 starts on the device (set by whichever upload ran)
    if (allPinned && kernelIn && stride <= 2 * (size_t)cols) {
        devPitch = stride;
        devStride = align_up(imgBytes + 32, 256);
        if ((r = sl.d_img.ensure((size_t)nimg * devStride + 256)) < 0) return r;
        if (nimg == 2) { // a stereo pair: one launch, both images stream at once
            d_imgBase = upload_by_kernel(s, sl.d_img.p, imgs[0], imgBytes, sl.d_img.p + devStride, imgs[1]);
        } else {
            for (int i = 0; i < nimg; i++) {
                const uint8_t* at = upload_by_kernel(s, sl.d_img.p + (size_t)i * devStride, imgs[i], imgBytes);
                const uint8_t* want = at ? at - (size_t)i * devStride : nullptr;
                if (!at || (i > 0 && want != d_imgBase)) { // no device alias, or sources at different offsets in their 16-B block
                    d_imgBase = nullptr;
                    break;
                }
                d_imgBase = want;
            }
        }
    }
    if (d_imgBase) {
        // (uploaded above)
    } else if (allPinned) {
        // DMA straight from the caller's memory.  Rows are copied with their padding ((rows-1)*stride + cols bytes,
        // one linear command per image, or ONE command for the whole batch when the images are evenly spaced in one
        // buffer); the kernels then read the image with the caller's pitch.
        bool even = stride <= 2 * (size_t)cols;
        ptrdiff_t D = nimg > 1 ? imgs[1] - imgs[0] : (ptrdiff_t)align_up(imgBytes, 256);
        for (int i = 1; i < nimg && even; i++) even = imgs[i] - imgs[i - 1] == D;
        // (the bytes between two images are read too: they must belong to the same pinned buffer -- known from the
        // registry, or because the images follow each other without a gap)
        // ONE registered range must cover them all (images page-locked one by one cannot be read by one command), or they
        // are pinned by somebody else's allocation and follow each other without a gap
        even = even && D >= (ptrdiff_t)imgBytes && (size_t)D <= 2 * align_up(imgBytes, 256) &&
               (pin_known(imgs[0], (size_t)(nimg - 1) * (size_t)D + imgBytes) ||
                ((size_t)D == (size_t)rows * stride && !pin_known(imgs[0], imgBytes)));
        if (stride <= 2 * (size_t)cols) {
            devPitch = stride;
            devStride = even ? (size_t)D : align_up(imgBytes, 256);
            if ((r = sl.d_img.ensure((size_t)nimg * devStride + 256)) < 0) return r;
            if (even && hipMemcpyAsync(sl.d_img.p, imgs[0], (size_t)(nimg - 1) * devStride + imgBytes, hipMemcpyHostToDevice,
                                       sIn) != hipSuccess) {
                (void)hipGetLastError(); // (not one allocation after all: image by image)
                even = false;
            }
            if (!even) {
                for (int i = 0; i < nimg; i++)
                    HIP_TRY(hipMemcpyAsync(sl.d_img.p + (size_t)i * devStride, imgs[i], imgBytes, hipMemcpyHostToDevice,
                                           sIn));
            }
        } else { // views with a large pitch: 2-D copies
            devPitch = align_up((size_t)cols, 64);
            devStride = devPitch * rows;
            if ((r = sl.d_img.ensure((size_t)nimg * devStride + 256)) < 0) return r;
            for (int i = 0; i < nimg; i++)
                HIP_TRY(hipMemcpy2DAsync(sl.d_img.p + (size_t)i * devStride, devPitch, imgs[i], stride, (size_t)cols,
                                         (size_t)rows, hipMemcpyHostToDevice, sIn));
        }
    } else {
        // pageable (or mixed) images: dense staging copy by the pool, one DMA command per chunk of images while the
        // next chunk is being staged
        devPitch = (size_t)cols;
        devStride = (size_t)cols * rows;
        if ((r = sl.d_img.ensure((size_t)nimg * devStride + 256)) < 0) return r;
        if ((r = sl.h_in.ensure((size_t)nimg * devStride)) < 0) return r;
        const int chunk = nimg * devStride <= (2u << 20) ? nimg : std::max(1, (int)((4u << 20) / devStride));
        for (int i0 = 0; i0 < nimg; i0 += chunk) {
            const int ni = std::min(chunk, nimg - i0);
            if ((size_t)ni * devStride < (1u << 20)) { // a frame or two: the pool's wake-up costs more than it saves
                for (int i = i0; i < i0 + ni; i++)
                    copy_rows(sl.h_in.p + (size_t)i * devStride, devPitch, imgs[i], stride, (size_t)cols, rows);
            } else {
                // split every image into row bands so that the work divides evenly whatever the chunk holds
                const int bands = std::max(1, std::min(rows, 16 / ni + 1));
                CopyPool::get().run(ni * bands, [&](int t) {
                    const int i = i0 + t / bands, b = t % bands;
                    const int y0 = (int)((long)rows * b / bands), y1 = (int)((long)rows * (b + 1) / bands);
                    copy_rows(sl.h_in.p + (size_t)i * devStride + (size_t)y0 * devPitch, devPitch,
                              imgs[i] + (size_t)y0 * stride, stride, (size_t)cols, y1 - y0);
                });
            }
            if (kernelIn && upload_by_kernel(s, sl.d_img.p + (((size_t)i0 * devStride) & ~(size_t)15),
                                             sl.h_in.p + (size_t)i0 * devStride, (size_t)ni * devStride) ==
                                sl.d_img.p + (size_t)i0 * devStride)
                continue; // (the staging buffer is 16-byte aligned: same offset inside a 16-byte block on both sides)
            HIP_TRY(hipMemcpyAsync(sl.d_img.p + (size_t)i0 * devStride, sl.h_in.p + (size_t)i0 * devStride,
                                   (size_t)ni * devStride, hipMemcpyHostToDevice, sIn));
        }
    }
    if (!d_imgBase) d_imgBase = sl.d_img.p;
    if (pipelined) {
        HIP_TRY(hipEventRecord(sl.evIn, sIn));
        HIP_TRY(hipStreamWaitEvent(s, sl.evIn, 0));
    }

    // ---- kernels
    sl.metaBytes = align_up(((size_t)2 * nimg + 1) * sizeof(int32_t), 64);
    const size_t kpsBytes = (size_t)nimg * Kc * 28, descBytes = (size_t)nimg * Kc * 32;
    if ((r = sl.d_out.ensure(sl.metaBytes + kpsBytes + descBytes)) < 0) return r;
    int32_t* d_meta = reinterpret_cast<int32_t*>(sl.d_out.p);
    float* d_kps = reinterpret_cast<float*>(sl.d_out.p + sl.metaBytes);
    uint8_t* d_desc = sl.d_out.p + sl.metaBytes + kpsBytes;
    uint8_t* mirror = nullptr;
    sl.doneSeq = 0;
    if (mirrorOut) {
        if ((r = sl.h_out.ensure(sl.metaBytes + kpsBytes + descBytes)) < 0) return r;
        mirror = sl.h_out.dev();
        // completion word (the caller's wait is the next thing that happens to this slot, and K-DESC the call's last kernel)
        if (spinOk && c->spinWait && sl.h_out.coherent) {
            if (!c->d_done.p) {
                if ((r = c->d_done.ensure(80)) < 0) return r;
                HIP_TRY(hipMemsetAsync(c->d_done.p, 0, 80 * sizeof(unsigned), s)); // (ordered with the kernels that count)
                if ((r = c->h_done.ensure(16)) < 0) return r;
                c->h_done.p[0] = 0u;
            }
            c->doneWant = true;
            c->doneGot = 0u;
        }
    }
    c->lapInlineN = nimg <= 2 ? nimg : 0; // (consumed and cleared by run_device)
    for (int i = 0; i < 2 * c->lapInlineN; i++) c->lapInline[i] = lap ? lap[i] : 0;
    r = run_device(c, nimg, d_imgBase, rows, cols, devPitch, devStride, sl.d_lapAlias, d_kps, d_desc,
                   cap_per_img, d_meta, d_meta + nimg, d_meta + 2 * nimg, mirror, sl.metaBytes);
    sl.doneSeq = c->doneGot;
    c->doneGot = 0u;
    c->doneWant = false;
    if (r < 0) return r;
    if (pipelined) {
        HIP_TRY(hipEventRecord(sl.evK, s));
        HIP_TRY(hipStreamWaitEvent(sOut, sl.evK, 0));
    }

    // ---- results: into the caller's arrays directly when those are pinned (the layouts are the same: nimg slabs of
    // cap entries), else ONE transfer of [meta | keypoints | descriptors] into the slot's pinned staging
    // (a frame or two: one command into the staging buffer and a 60-KB memcpy beat three DMA commands)
    sl.outPinned = !mirror && kpsBytes + descBytes >= (1u << 20) && is_pinned(c, kps, kpsBytes) && is_pinned(c, desc, descBytes);
    if (mirror) {
        // (the kernels have written the slot's pinned slab themselves: nothing to queue)
    } else if (sl.outPinned) {
        if ((r = sl.h_out.ensure(sl.metaBytes)) < 0) return r;
        HIP_TRY(hipMemcpyAsync(sl.h_out.p, sl.d_out.p, sl.metaBytes, hipMemcpyDeviceToHost, sOut));
        HIP_TRY(hipMemcpyAsync(kps, d_kps, kpsBytes, hipMemcpyDeviceToHost, sOut));
        HIP_TRY(hipMemcpyAsync(desc, d_desc, descBytes, hipMemcpyDeviceToHost, sOut));
    } else {
        if ((r = sl.h_out.ensure(sl.metaBytes + kpsBytes + descBytes)) < 0) return r;
        HIP_TRY(hipMemcpyAsync(sl.h_out.p, sl.d_out.p, sl.metaBytes + kpsBytes + descBytes, hipMemcpyDeviceToHost, sOut));
    }
    if (pipelined) HIP_TRY(hipEventRecord(sl.evDone, sOut));
    sl.busy = true;
    sl.pipelined = pipelined;
    sl.nimg = nimg;
    sl.cap = cap_per_img;
    sl.kps = kps;
    sl.desc = desc;
    sl.n_out = n_out;
    sl.mono_out = mono_out;
    if (laneSlot < 0) c->slotSubmitted++;
    return 0;
}

// A submit that fails after its first copy or kernel has been queued returns an error without marking the slot busy: the
// caller cannot wait for it and may free or reuse its buffers at once.  So nothing may still be in flight then: the three
// streams are drained before the error is handed back (DMA engines read the images, and write pinned result arrays, in
// place).
int host_submit(orbfe_ctx* c, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride, const int* lap,
                orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out, bool pipelined,
                bool spinOk = false /* the caller waits at once and queues nothing behind the extraction */)
{
    if (c && c->slotSubmitted - c->slotRetired >= 2) return ORBFE_ERR_STATE; // both slots in flight: nothing was queued
    if (c && c->pairSubmitted != c->pairRetired) return ORBFE_ERR_STATE;     // stereo frames in flight own the slots
    const int r = host_submit_impl(c, nimg, imgs, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, pipelined,
                                   spinOk);
    if (r <= -1000 && c) { // (a HIP error: something may have been queued; validation errors come before the first command)
        if (hipSetDevice(c->device) == hipSuccess) {
            if (c->sIn) (void)hipStreamSynchronize(c->sIn);
            (void)hipStreamSynchronize(c->stream);
            if (c->sOut) (void)hipStreamSynchronize(c->sOut);
        }
    }
    return r;
}

// Spin on the context's completion word for `seq` (bounded), true when it was seen.
static bool spin_done(orbfe_ctx* c, unsigned seq)
{
    c->spinLate = 0u;
    if (!seq || !c->h_done.p) return false;
    const volatile unsigned* f = c->h_done.p;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; it++) {
        const unsigned v = *f;
        if (v == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            c->spinMisses = 0;
            return true;
        }
        if (v == (seq | 0x80000000u)) return false; // the kernel finished but does not vouch for its stores' order: synchronise
        __builtin_ia32_pause();
        if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
    }
    // The word did not come within the bound (the caller now synchronises the stream).  Should a counter ever be left non-zero
    // -- a kernel that died half-way -- every later call would time out as well: clear them behind whatever is still queued.
    (void)hipMemsetAsync(c->d_done.p, 0, 80 * sizeof(unsigned), run_stream(c));
    c->spinLate = seq; // (spin_settle, once the caller has synchronised the stream, says whether this was a miss)
    return false;
}
// After the stream synchronisation that follows a spin_done() == false: a word that is there now was merely late (a long
// call, a kernel queued behind somebody else's work); one that is still missing counts, and eight of those in a row switch
// the word off for this context (lateness alone used to count).
static void spin_settle(orbfe_ctx* c)
{
    if (!c->spinLate || !c->h_done.p) return;
    const unsigned v = *(const volatile unsigned*)c->h_done.p;
    if (v == c->spinLate || v == (c->spinLate | 0x80000000u)) c->spinMisses = 0;
    else if (++c->spinMisses >= 8) c->spinWait = false; // (the word does not arrive on this platform: stop paying the bound)
    c->spinLate = 0u;
}

// Complete the oldest submitted batch: wait for its transfers, hand out the counts, and -- for pageable output
// arrays -- copy the rows each image produced out of the staging buffer.
int host_wait(orbfe_ctx* c, int laneSlot = -1 /* as in host_submit_impl */)
{
    if (!c) return ORBFE_ERR_ARGS;
    if (laneSlot < 0 && c->slotSubmitted == c->slotRetired) return ORBFE_ERR_STATE;
    orbfe_ctx::HostSlot& sl = c->slot[laneSlot >= 0 ? laneSlot : (int)(c->slotRetired & 1)];
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipSuccess;
    // (the completion word the call's last kernel publishes behind the results it has written: OrbDone)
    const bool seen = !sl.pipelined && spin_done(c, sl.doneSeq);
    if (!seen) e = sl.pipelined ? hipEventSynchronize(sl.evDone) : hipStreamSynchronize(run_stream(c));
    if (!seen && !sl.pipelined) spin_settle(c);
    sl.busy = false;
    if (laneSlot < 0) c->slotRetired++;
    if (e != hipSuccess) return -(1000 + (int)e);
    const int nimg = sl.nimg;
    const int32_t* meta = reinterpret_cast<const int32_t*>(sl.h_out.p);
    if (meta[2 * nimg] != 0) return ORBFE_ERR_STATE; // a device-side list overflowed (cannot happen)
    const size_t Kc = (size_t)sl.cap;
    size_t total = 0;
    // (the counts come out of the page-locked slab the kernels wrote: a count outside [0, cap] cannot be a result -- never copy
    // by it into the caller's arrays, whatever made it so)
    for (int i = 0; i < nimg; i++)
        if (meta[i] < 0 || (size_t)meta[i] > Kc || meta[nimg + i] < 0 || meta[nimg + i] > meta[i]) {
            for (int k = 0; k < nimg; k++) {
                sl.n_out[k] = 0;
                if (sl.mono_out) sl.mono_out[k] = 0;
            }
            return ORBFE_ERR_STATE;
        }
    for (int i = 0; i < nimg; i++) {
        sl.n_out[i] = meta[i];
        if (sl.mono_out) sl.mono_out[i] = meta[nimg + i];
        total += (size_t)std::max(meta[i], 0);
    }
    if (!sl.outPinned && total > 0) {
        const uint8_t* hk = sl.h_out.p + sl.metaBytes;
        const uint8_t* hd = hk + (size_t)nimg * Kc * 28;
        auto one = [&](int i) {
            const int n = meta[i];
            if (n <= 0) return;
            std::memcpy((uint8_t*)sl.kps + (size_t)i * Kc * 28, hk + (size_t)i * Kc * 28, (size_t)n * 28);
            std::memcpy(sl.desc + (size_t)i * Kc * 32, hd + (size_t)i * Kc * 32, (size_t)n * 32);
        };
        if (total * 60 < (1u << 20)) {
            for (int i = 0; i < nimg; i++) one(i);
        } else {
            CopyPool::get().run(nimg, one);
        }
    }
    return 0;
}

} // namespace

extern "C" {

const char* orbfe_version(void) { return "orbfe 0.1 (gfx950)"; }

const char* orbfe_error_string(int code)
{
    switch (code) {
    case -1: return "empty image";
    case ORBFE_ERR_ARGS: return "bad argument (null pointer, negative size, or an output capacity below orbfe_max_keypoints)";
    case ORBFE_ERR_NODEV: return "no usable HIP device (the library has no CPU path)";
    case ORBFE_ERR_STATE: return "call out of order (nothing in flight / too much in flight / no results yet), or a device-side list overflowed";
    case ORBFE_ERR_IMAGE_SMALL:
        return "image too small: some pyramid level is narrower or lower than 32 + 35 px, where the reference's cell grid has no cell";
    case ORBFE_ERR_IMAGE_LARGE: return "image too large: a side above 4096 px (candidates are packed with 12-bit coordinates)";
    case ORBFE_ERR_NFEATURES:
        return "nfeatures too large: more than 65535 keypoint slots per image";
    case -8: return "an RCCL call failed (orbfe_mc_*: ncclGetUniqueId / ncclCommInitRank / ncclAllGather); ORBFE_VERBOSE=1 prints RCCL's reason";
    default: break;
    }
    if (code >= 0) return "success";
    if (code <= -1000) return hipGetErrorString((hipError_t)(-code - 1000));
    return "unknown error code";
}

int orbfe_create(orbfe_ctx** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int device)
{
    if (!out) return ORBFE_ERR_ARGS;
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > ORBFE_MAX_LEVELS || !(scaleFactor > 1.0f) || iniThFAST < 0 ||
        minThFAST < 0 || iniThFAST > 255 || minThFAST > 255)
        return ORBFE_ERR_ARGS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return ORBFE_ERR_NODEV;
    HIP_TRY(hipSetDevice(device));
    orbfe_ctx* c = new orbfe_ctx();
    c->nfeatures = nfeatures;
    c->scaleFactor = scaleFactor;
    c->nlevels = nlevels;
    c->iniThFAST = iniThFAST;
    c->minThFAST = minThFAST;
    c->device = device;
    init_tables(c);
    if (const char* e = getenv("ORBFE_ATAN_FMA")) c->atanFma = atoi(e) != 0;
    if (const char* e = getenv("ORBFE_SPIN")) c->spinWait = atoi(e) != 0;
    if (const char* e = getenv("ORBFE_LANES")) c->lanes = std::min(ORBFE_MAX_LANES, std::max(1, atoi(e)));
    if (const char* e = getenv("ORBFE_LANES_INPUT_GUARD")) c->inputGuard = atoi(e) != 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ORBFE_ERR_NODEV;
    }
    c->ownStream = true;
    *out = c;
    return 0;
}

void orbfe_destroy(orbfe_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    lane_quiesce(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->sIn) (void)hipStreamSynchronize(c->sIn);   // submitted batches nobody waited for: their copies still
    if (c->sOut) (void)hipStreamSynchronize(c->sOut); // touch the slots' buffers (and the caller's arrays)
    c->d_pyr.release();
    c->d_cand.release(); c->d_keys.release(); c->d_lvlKp.release(); c->d_keyNode.release();
    c->d_cellCount.release(); c->d_lvlCount.release(); c->d_lap.release();
    c->d_fix.release(); c->d_kb8.release(); c->d_rays.release();
    c->d_destMap.release();
    c->d_lvlPre.release();
    c->d_qtScratch.release();
    c->release_tables();
    for (auto& g : c->geomCache) g.release_tables();
    c->d_taps.release(); c->d_patternF.release();
    c->h_fix.release(); c->h_fixAB.release(); c->d_stereo.release(); c->h_stereo.release(); c->d_done.release(); c->h_done.release();
    c->d_stereoIo.release(); c->h_stereoIo.release(); c->h_level.release();
    for (auto& sl : c->slot) {
        sl.d_img.release(); sl.d_out.release(); sl.h_in.release(); sl.h_out.release(); sl.h_lap.release();
        if (sl.evIn) (void)hipEventDestroy(sl.evIn);
        if (sl.evK) (void)hipEventDestroy(sl.evK);
        if (sl.evDone) (void)hipEventDestroy(sl.evDone);
    }
    auto_pin_release(c);
    if (c->evStereo) (void)hipEventDestroy(c->evStereo);
    if (c->evOutputs) {
        orbfe_producer_retire(c->evOutputs);
        (void)hipEventDestroy(c->evOutputs);
    }
    if (c->sIn) (void)hipStreamDestroy(c->sIn);
    if (c->sOut) (void)hipStreamDestroy(c->sOut);
    if (c->evReady)
        for (auto& e : c->ev) (void)hipEventDestroy(e);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    for (int k = 0; k < ORBFE_MAX_LANES; k++) {
        orbfe_ctx::Lane& L = c->lane[k];
        orbfe_ctx::LaneBufs& b = L.parked; // (the current lane's buffers are the context's own members, released above)
        b.d_pyr.release(); b.d_cand.release(); b.d_keys.release(); b.d_lvlKp.release(); b.d_lvlPre.release(); b.d_keyNode.release();
        b.d_cellCount.release(); b.d_lvlCount.release(); b.d_destMap.release(); b.d_fix.release(); b.d_qtScratch.release();
        b.h_fix.release(); b.h_fixAB.release(); b.d_done.release(); b.h_done.release(); b.d_stereo.release(); b.h_stereo.release();
        if (L.evJoin) (void)hipEventDestroy(L.evJoin);
        if (L.evRead) (void)hipEventDestroy(L.evRead);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.pairStream) (void)hipStreamDestroy(L.pairStream);
    }
    if (c->evBatchFork) (void)hipEventDestroy(c->evBatchFork);
    delete c;
}

int orbfe_release_caches(int device)
{
    if (device < 0 || device >= 16) return ORBFE_ERR_ARGS;
    std::lock_guard<std::mutex> lock(g_trigMutex);
    TrigTable& t = g_trig[device];
    if (t.d || t.full) {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipDeviceSynchronize()); // a kernel of some context may still be reading the table
        if (t.d) (void)hipFree(t.d);
        if (t.full) (void)hipFree(t.full);
    }
    t = TrigTable(); // built again by the next ORBFE_TRIG_LIBM extraction
    return 0;
}

int orbfe_get_stream(orbfe_ctx* c, void** hip_stream, int* device)
{
    if (!c) return ORBFE_ERR_ARGS;
    if (hip_stream) *hip_stream = (void*)c->stream;
    if (device) *device = c->device;
    return 0;
}

int orbfe_set_stream(orbfe_ctx* c, void* hip_stream)
{
    if (!c) return ORBFE_ERR_ARGS;
    (void)hipSetDevice(c->device);
    lane_quiesce(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    c->ownStream = false;
    c->stream = (hipStream_t)hip_stream;
    if (!hip_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->ownStream = true;
    }
    return 0;
}

int orbfe_set_gaussian_taps(orbfe_ctx* c, const int* t)
{
    if (!c || !t) return ORBFE_ERR_ARGS;
    int sum = 0;
    for (int i = 0; i < 7; i++) {
        if (t[i] < 0 || t[i] > 255) return ORBFE_ERR_ARGS; // taps are u8 operands of v_dot4_u32_u8
        sum += t[i];
    }
    if (sum > 257) return ORBFE_ERR_ARGS; // horizontal pass must fit 16 bits
    for (int i = 0; i < 7; i++) c->taps[i] = t[i];
    c->tapsDirty = true;
    return 0;
}

int orbfe_set_trig_mode(orbfe_ctx* c, int mode)
{
    if (!c || (mode != ORBFE_TRIG_LIBM && mode != ORBFE_TRIG_CR && mode != ORBFE_TRIG_LIBM_HOSTCHECK)) return ORBFE_ERR_ARGS;
    c->trigMode = mode;
    return 0;
}

int orbfe_set_atan_fma(orbfe_ctx* c, int on)
{
    if (!c) return ORBFE_ERR_ARGS;
    c->atanFma = on != 0;
    return 0;
}

int orbfe_set_kb8(orbfe_ctx* c, const float* params8)
{
    if (!c) return ORBFE_ERR_ARGS;
    c->kb8On = params8 != nullptr;
    if (params8) {
        for (int i = 0; i < 8; i++) c->kb8[i] = params8[i];
        c->capImgs = 0; // make ensure_capacity allocate the ray buffer
    }
    return 0;
}

int orbfe_set_ray_output(orbfe_ctx* c, float* d_rays)
{
    if (!c) return ORBFE_ERR_ARGS;
    c->userRays = d_rays;
    return 0;
}

int orbfe_get_rays(orbfe_ctx* c, int img_index, int cap_per_img, float* rays, int n)
{
    if (!c || !c->kb8On || c->userRays || !c->d_rays.p || img_index < 0 || img_index >= c->lastImgs || n < 0 ||
        n > cap_per_img || (n && !rays))
        return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    lane_quiesce(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) // (the caller's array may be pageable: through page-locked memory of this thread, orbfe_pageable.h)
        HIP_TRY(orbfe_pageable::down(rays, c->d_rays.p + (size_t)img_index * cap_per_img * 3, (size_t)n * 3 * sizeof(float), c->stream));
    return 0;
}

int orbfe_max_keypoints(orbfe_ctx* c, int rows, int cols)
{
    if (!c || rows <= 0 || cols <= 0) return ORBFE_ERR_ARGS;
    return max_kp_for(c, rows, cols);
}

int orbfe_sync(orbfe_ctx* c)
{
    if (!c) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    bool laneErr = false;
    for (int k = 0; k < ORBFE_MAX_LANES; k++) { // batch lanes: the status header of every lane that held work
        orbfe_ctx::Lane& L = c->lane[k];
        if (L.pairStream) HIP_TRY(hipStreamSynchronize(L.pairStream));
        L.pendingPair = false;
        if (!L.stream) continue;
        HIP_TRY(hipStreamSynchronize(L.stream));
        const DevBuf<int4>& fx = k == c->curSet ? c->d_fix : L.parked.d_fix;
        if (L.pending && fx.p && k != c->curSet) {
            int4 h = {0, 0, 0, 0};
            HIP_TRY(hipMemcpy(&h, fx.p, sizeof(int4), hipMemcpyDeviceToHost));
            laneErr = laneErr || h.y != 0;
        }
        L.pending = false;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (laneErr) return ORBFE_ERR_STATE;
    if (c->d_fix.p && c->lastImgs > 0) { // the error word of the last batch (k_octree raises it, nothing else does)
        int r = c->h_fix.ensure(2);
        if (r < 0) return r;
        HIP_TRY(hipMemcpy(c->h_fix.p, c->d_fix.p, sizeof(int4), hipMemcpyDeviceToHost));
        if (c->h_fix.p[0].y != 0) return ORBFE_ERR_STATE;
    }
    return 0;
}

int orbfe_extract_batch_device(orbfe_ctx* c, int nimg, const uint8_t* d_imgs, int rows, int cols, size_t pitch,
                               size_t img_stride_bytes, int lap0, int lap1, orbfe_kp* d_kps, uint8_t* d_desc,
                               int cap_per_img, int32_t* d_n_out, int32_t* d_mono_out)
{
    if (!c || nimg < 1 || !d_imgs || !d_kps || !d_desc || !d_n_out || !d_mono_out) return ORBFE_ERR_ARGS;
    if (rows <= 0 || cols <= 0) return -1;
    if (pitch < (size_t)cols) return ORBFE_ERR_ARGS;
    if (c->pairSubmitted != c->pairRetired) return ORBFE_ERR_STATE; // stereo frames in flight hold the lanes' buffers
    HIP_TRY(hipSetDevice(c->device));
    int r;
    // Batch lanes: this call goes, whole, to the next lane's stream with that lane's buffers
    const bool batchLanes = c->lanes >= 2 && !c->kb8On;
    if (batchLanes) {
        if ((r = batch_lane_setup(c)) < 0) return r;
        if (c->laneNext >= c->lanes) c->laneNext = 0;
        lane_select(c, c->laneNext);
    } else if (lanes_busy(c)) {
        if ((r = lane_join(c)) < 0) return r; // a one-stream call behind lane calls
    }
    if ((r = ensure_geometry(c, rows, cols, nimg)) < 0) return r;
    if ((r = ensure_capacity(c, nimg, std::max(cap_per_img, c->maxKp))) < 0) return r;
    // The per-image lapping table only changes when the caller changes (lap0, lap1) or grows the batch: upload
    // it then (with a synchronisation, the source is a stack buffer) and never again -- the steady state of this
    // entry point issues kernels only and does not block the host.
    if (c->lapDev0 != lap0 || c->lapDev1 != lap1 || c->lapDevCount < nimg) {
        lane_quiesce(c); // (the second lane's K-QT may still be reading the table)
        std::vector<int32_t> lap((size_t)nimg * 2);
        for (int i = 0; i < nimg; i++) {
            lap[2 * i] = lap0;
            lap[2 * i + 1] = lap1;
        }
        HIP_TRY(hipMemcpyAsync(c->d_lap.p, lap.data(), lap.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->lapDev0 = lap0;
        c->lapDev1 = lap1;
        c->lapDevCount = nimg;
    }
    if (batchLanes) {
        orbfe_ctx::Lane& L = c->lane[c->laneNext];
        // the lane starts no earlier than this point of the context's stream (inputs, tables); one-way.  An idle stream --
        // resident inputs, nothing queued by the caller -- orders nothing: the event record and the wait (two barrier packets,
        // two queues) are skipped then
        if (!c->forkSkipIdle || hipStreamQuery(c->stream) != hipSuccess) {
            (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
            HIP_TRY(hipEventRecord(c->evBatchFork, c->stream));
            HIP_TRY(hipStreamWaitEvent(L.stream, c->evBatchFork, 0));
        }
        c->runStream = L.stream;
        c->guardEv = c->inputGuard ? L.evRead : nullptr;
        r = run_device(c, nimg, d_imgs, rows, cols, pitch, img_stride_bytes, c->d_lap.p, (float*)d_kps, d_desc, cap_per_img,
                       d_n_out, d_mono_out);
        c->runStream = nullptr;
        c->guardEv = nullptr;
        L.pending = true; // (also after an error: something may have been queued)
        L.imgs = nimg;
        c->laneLast = c->laneNext;
        c->laneNext = (c->laneNext + 1) % c->lanes;
        if (r < 0) return r;
        // ... and whatever the caller queues on the context's stream after this call -- the next frame's upload into the same
        // image buffer -- comes after the lane's K-PYR, the only reader of the images (stream order, as with one lane)
        if (c->inputGuard) HIP_TRY(hipStreamWaitEvent(c->stream, L.evRead, 0));
        return 0;
    }
    c->laneLast = -1;
    return run_device(c, nimg, d_imgs, rows, cols, pitch, img_stride_bytes, c->d_lap.p, (float*)d_kps, d_desc,
                      cap_per_img, d_n_out, d_mono_out, nullptr, nullptr, 0, /*allowLanes=*/true);
}

int orbfe_set_lanes(orbfe_ctx* c, int lanes)
{
    if (!c || lanes < 1 || lanes > ORBFE_MAX_LANES) return ORBFE_ERR_ARGS;
    if (c->pairSubmitted != c->pairRetired) return ORBFE_ERR_STATE; // stereo frames in flight are indexed by the lane count
    HIP_TRY(hipSetDevice(c->device));
    int r = lane_join(c);
    if (r < 0) return r;
    c->lanes = lanes;
    c->laneNext = 0;
    return 0;
}

// (not in the public header: orbfe_mc_create tells the context that an exchange consumes its batches -- lane priorities above)
int orbfe_internal_exchange_hint(orbfe_ctx* c, int on)
{
    if (!c) return ORBFE_ERR_ARGS;
    c->exchangeHint = on != 0;
    return 0;
}

int orbfe_set_lane_input_guard(orbfe_ctx* c, int on)
{
    if (!c) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    int r = lane_join(c);
    if (r < 0) return r;
    c->inputGuard = on != 0;
    return 0;
}

int orbfe_set_lane_mode(orbfe_ctx* c, int mode)
{
    // (ORBFE_LANES_SPLIT -- round 4's two half-batches of one call -- was measured slower than whole batches per lane in round 5
    // and left with round 6; the entry point stays so that callers which select ORBFE_LANES_BATCH explicitly keep linking)
    if (!c || mode != ORBFE_LANES_BATCH) return ORBFE_ERR_ARGS;
    return 0;
}

int orbfe_lanes_join(orbfe_ctx* c)
{
    if (!c) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    return lane_join(c);
}

int orbfe_lanes_record(orbfe_ctx* c, void* hip_event)
{
    if (!c || !hip_event) return ORBFE_ERR_ARGS;
    if (c->laneLast >= 0) { // batch lanes: the last call's work is on its lane's stream
        if (!c->lane[c->laneLast].pending) return 0;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipEventRecord((hipEvent_t)hip_event, c->lane[c->laneLast].stream));
        return 1;
    }
    return 0;
}

// ---- pinned host memory ---------------------------------------------------------------------------
int orbfe_set_auto_register(orbfe_ctx* c, int on)
{
    if (!c) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    if (!on && c->autoRegister) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->sIn) HIP_TRY(hipStreamSynchronize(c->sIn));
        auto_pin_release(c);
    }
    c->autoRegister = on != 0;
    return 0;
}

int orbfe_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return ORBFE_ERR_ARGS;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    pin_add(p, bytes, false);
    return 0;
}
int orbfe_host_unregister(void* p)
{
    if (!p || !pin_remove(p)) return ORBFE_ERR_ARGS;
    HIP_TRY(hipHostUnregister(p));
    return 0;
}
void* orbfe_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    pin_add(p, bytes, true);
    return p;
}
void orbfe_host_free(void* p)
{
    if (!p) return;
    if (pin_remove(p)) (void)hipHostFree(p);
}

// ---- host-pointer path ------------------------------------------------------------------------------
int orbfe_extract_batch_submit(orbfe_ctx* c, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride,
                               const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out)
{
    return host_submit(c, nimg, imgs, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, true);
}

int orbfe_extract_batch_wait(orbfe_ctx* c) { return host_wait(c); }

int orbfe_extract_batch(orbfe_ctx* c, int nimg, const uint8_t* const* imgs, int rows, int cols, size_t stride,
                        const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out)
{
    if (c && c->slotSubmitted != c->slotRetired) return ORBFE_ERR_STATE; // submitted batches must be waited for first
    // A frame or two: everything on the context's stream (no event traffic on the latency path).  A real batch: the
    // copies go to the copy streams like the pipelined form -- measured: a 23-MB upload queued on the stream the
    // kernels run on takes 0.8 ms instead of the 0.41 ms the DMA engine needs on a stream of its own.
    // (Round 5: not for a frame or a PAIR however large -- two 1024 x 1024 fisheye images are exactly 2 MB and took the batch
    // form: copy engine 44 us + 16 us of hand-over instead of the upload kernel's 32, a download command instead of the mirror.
    // Measured both ways, profiles/r05_c5_stages.txt.)
    const int copyStreamsFrom = 3;
    const bool ownCopyStreams = rows > 0 && cols > 0 && nimg >= copyStreamsFrom &&
                                (size_t)nimg * (size_t)rows * (size_t)cols >= (2u << 20);
    const int r = host_submit(c, nimg, imgs, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, ownCopyStreams,
                              true);
    if (r < 0) return r;
    return host_wait(c);
}

// Images of different sizes in one call (a rig with unequal cameras): the images are grouped by size and every group
// runs as one batch on this context, whose per-size tables are kept (orbfe_geom_state), so nothing is rebuilt when
// the sizes alternate.  Outputs land in the caller's per-image slabs in the caller's order.
int orbfe_extract_batch_sizes(orbfe_ctx* c, int nimg, const uint8_t* const* imgs, const int* rows, const int* cols,
                              const size_t* strides, const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img,
                              int* n_out, int* mono_out)
{
    if (!c || nimg < 1 || !imgs || !rows || !cols || !strides || !kps || !desc || !n_out) return ORBFE_ERR_ARGS;
    if (c->slotSubmitted != c->slotRetired) return ORBFE_ERR_STATE;
    std::vector<int> order(nimg);
    for (int i = 0; i < nimg; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (rows[a] != rows[b]) return rows[a] < rows[b];
        if (cols[a] != cols[b]) return cols[a] < cols[b];
        return strides[a] < strides[b];
    });
    int worst = 0;
    for (int g0 = 0; g0 < nimg;) {
        int g1 = g0 + 1;
        while (g1 < nimg && rows[order[g1]] == rows[order[g0]] && cols[order[g1]] == cols[order[g0]] &&
               strides[order[g1]] == strides[order[g0]])
            g1++;
        const int ng = g1 - g0;
        std::vector<const uint8_t*> gi(ng);
        std::vector<int> glap(2 * (size_t)ng, 0), gn(ng, 0), gm(ng, 0);
        for (int k = 0; k < ng; k++) {
            gi[k] = imgs[order[g0 + k]];
            if (lap) {
                glap[2 * k] = lap[2 * order[g0 + k]];
                glap[2 * k + 1] = lap[2 * order[g0 + k] + 1];
            }
        }
        std::vector<orbfe_kp> gk((size_t)ng * cap_per_img);
        std::vector<uint8_t> gd((size_t)ng * cap_per_img * 32);
        const int r = orbfe_extract_batch(c, ng, gi.data(), rows[order[g0]], cols[order[g0]], strides[order[g0]], glap.data(),
                                          gk.data(), gd.data(), cap_per_img, gn.data(), gm.data());
        for (int k = 0; k < ng; k++) {
            const int i = order[g0 + k];
            n_out[i] = r < 0 ? 0 : gn[k];
            if (mono_out) mono_out[i] = r < 0 ? 0 : gm[k];
            if (r >= 0 && gn[k] > 0) {
                std::memcpy(kps + (size_t)i * cap_per_img, gk.data() + (size_t)k * cap_per_img, (size_t)gn[k] * sizeof(orbfe_kp));
                std::memcpy(desc + (size_t)i * cap_per_img * 32, gd.data() + (size_t)k * cap_per_img * 32, (size_t)gn[k] * 32);
            }
        }
        if (r < 0 && worst == 0) worst = r; // the first failing group's code; the other groups still ran
        g0 = g1;
    }
    return worst;
}

int orbfe_extract(orbfe_ctx* c, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1,
                  orbfe_kp* kps, uint8_t* desc, int cap, int* n_out)
{
    if (n_out) *n_out = 0;
    if (!c) return ORBFE_ERR_ARGS;
    if (!img || rows <= 0 || cols <= 0) return -1; // _image.empty(), :1072-1073
    int lap[2] = {lap0, lap1};
    int n = 0, mono = 0;
    const uint8_t* imgs[1] = {img};
    int r = orbfe_extract_batch(c, 1, imgs, rows, cols, stride, lap, kps, desc, cap, &n, &mono);
    if (r < 0) return r;
    if (n_out) *n_out = n;
    return mono;
}

int orbfe_get_device_outputs(orbfe_ctx* c, const orbfe_kp** d_kps, const uint8_t** d_desc, const int32_t** d_n, int* cap,
                             int* nimg)
{
    if (!c) return ORBFE_ERR_ARGS;
    if (!c->lastKps || c->lastImgs < 1) return ORBFE_ERR_STATE;
    // The arrays may still be being written (orbfe_extract_batch_device returns at once): mark the point on the producing
    // stream and publish the ranges, so that a matcher call that is handed one of these pointers orders itself after it
    // (orbfe_order.h).  The caller's own kernels must still be ordered by the caller (same stream, or orbfe_sync).
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rj = lane_join(c); // (two lanes: the mark below must lie behind BOTH halves of the batch)
        if (rj < 0) return rj;
    }
    if (!c->evOutputs) HIP_TRY(hipEventCreateWithFlags(&c->evOutputs, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->evOutputs, c->stream));
    orbfe_producer_publish(c->lastDesc, (size_t)c->lastImgs * c->lastCap * 32, c->evOutputs);
    orbfe_producer_publish(c->lastKps, (size_t)c->lastImgs * c->lastCap * sizeof(orbfe_kp), c->evOutputs);
    orbfe_producer_publish(c->lastN, (size_t)c->lastImgs * sizeof(int32_t), c->evOutputs);
    if (d_kps) *d_kps = reinterpret_cast<const orbfe_kp*>(c->lastKps);
    if (d_desc) *d_desc = c->lastDesc;
    if (d_n) *d_n = c->lastN;
    if (cap) *cap = c->lastCap;
    if (nimg) *nimg = c->lastImgs;
    return 0;
}

int orbfe_get_levels(orbfe_ctx* c) { return c ? c->nlevels : ORBFE_ERR_ARGS; }
float orbfe_get_scale_factor(orbfe_ctx* c) { return c ? (float)c->scaleFactor : 0.f; }
void orbfe_get_scale_tables(orbfe_ctx* c, float* sf, float* inv, float* s2, float* is2)
{
    if (!c) return;
    for (int i = 0; i < c->nlevels; i++) {
        if (sf) sf[i] = c->mvScaleFactor[i];
        if (inv) inv[i] = c->mvInvScaleFactor[i];
        if (s2) s2[i] = c->mvLevelSigma2[i];
        if (is2) is2[i] = c->mvInvLevelSigma2[i];
    }
}
void orbfe_get_features_per_level(orbfe_ctx* c, int* n)
{
    if (!c || !n) return;
    for (int i = 0; i < c->nlevels; i++) n[i] = c->mnFeaturesPerLevel[i];
}

int orbfe_get_level(orbfe_ctx* c, int img_index, int level, uint8_t* dst, size_t dst_stride, int* rows, int* cols)
{
    if (!c || c->lg.empty() || level < 0 || level >= c->nlevels || img_index < 0 || img_index >= c->lastImgs)
        return ORBFE_ERR_ARGS;
    const OrbLevelGeom& L = c->lg[level];
    const int W = L.w + 2 * ORBFE_EDGE, H = L.h + 2 * ORBFE_EDGE;
    if (rows) *rows = H;
    if (cols) *cols = W;
    if (!dst) return 0;
    if (dst_stride < (size_t)W) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    {
        const int rj = lane_join(c);
        if (rj < 0) return rj;
    }
    hipLaunchKernelGGL(k_border, dim3((unsigned)((W * H + 255) / 256)), dim3(256), 0, c->stream, c->d_pyr.p,
                       c->pyrStride, L, img_index);
    const uint8_t* src = c->d_pyr.p + (size_t)img_index * c->pyrStride + L.bufOff + (ORBFE_ROI_X0 - ORBFE_EDGE);
    // Through page-locked memory of the context, never a rect copy into the caller's (pageable) array.  A 2-D copy to pageable
    // memory makes the runtime pin the destination on the fly and write through that mapping; in round 6 this very call died
    // once with "Memory access fault by GPU ... on address <a host heap address>" inside such a copy (what is
    // known, and the hypothesis about the runtime's cache of pinned regions that would explain it -- not demonstrated).  A
    // linear copy into memory this context pinned itself, and a host row copy, take the runtime's path out of the picture.
    const size_t bytes = (size_t)(H - 1) * L.pitch + (size_t)W;
    int r = c->h_level.ensure(bytes);
    if (r < 0) return r;
    HIP_TRY(hipMemcpyAsync(c->h_level.p, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < H; y++) std::memcpy(dst + (size_t)y * dst_stride, c->h_level.p + (size_t)y * L.pitch, (size_t)W);
    return 0;
}

// Frame::ComputeStereoMatches (src/Frame.cc:797-967) on the pyramids the two extractors left on the device.
int orbfe_compute_stereo_matches(orbfe_ctx* left, orbfe_ctx* right, const orbfe_kp* kpsL, const uint8_t* descL, int nL,
                                 const orbfe_kp* kpsR, const uint8_t* descR, int nR, float mb, float mbf, float* uRight,
                                 float* depth)
{
    if (!left || !right || nL < 0 || nR < 0 || (nL && (!kpsL || !descL || !uRight || !depth)) ||
        (nR && (!kpsR || !descR)) || nR >= (1 << 20) || !(mb > 0))
        return ORBFE_ERR_ARGS;
    for (int i = 0; i < nL; i++) {
        uRight[i] = -1.0f;
        depth[i] = -1.0f;
    }
    if (nL == 0 || nR == 0) return 0;
    if (left->lg.empty() || right->lg.empty() || left->lastImgs < 1 || right->lastImgs < 1 ||
        left->device != right->device || left->rows != right->rows || left->cols != right->cols ||
        left->nlevels != right->nlevels || left->scaleFactor != right->scaleFactor)
        return ORBFE_ERR_STATE;
    HIP_TRY(hipSetDevice(left->device));
    {
        int rj = lane_join(left);
        if (rj >= 0) rj = lane_join(right);
        if (rj < 0) return rj;
    }
    // Inputs and outputs live in the left context's arenas (no allocation per call): everything is staged in pinned memory,
    // goes up in ONE transfer and comes back in ONE; the right extractor's kernels are ordered by an event, not by the host.
    const size_t oKL = 0, oKR = align_up(oKL + (size_t)nL * 28, 16), oDL = align_up(oKR + (size_t)nR * 28, 16),
                 oDR = align_up(oDL + (size_t)nL * 32, 16), oOut = align_up(oDR + (size_t)nR * 32, 16),
                 total = oOut + 3 * (size_t)nL * 4;
    int r;
    if ((r = left->d_stereoIo.ensure(total)) < 0) return r;
    if ((r = left->h_stereoIo.ensure(total)) < 0) return r;
    uint8_t* const hb = left->h_stereoIo.p;
    uint8_t* const db = left->d_stereoIo.p;
    std::memcpy(hb + oKL, kpsL, (size_t)nL * 28);
    std::memcpy(hb + oKR, kpsR, (size_t)nR * 28);
    std::memcpy(hb + oDL, descL, (size_t)nL * 32);
    std::memcpy(hb + oDR, descR, (size_t)nR * 32);
    hipStream_t s = left->stream;
    if (right->stream != s) {
        if (!left->evStereo) HIP_TRY(hipEventCreateWithFlags(&left->evStereo, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(left->evStereo, right->stream));
        HIP_TRY(hipStreamWaitEvent(s, left->evStereo, 0));
    }
    HIP_TRY(hipMemcpyAsync(db, hb, oOut, hipMemcpyHostToDevice, s));
    float* dU = reinterpret_cast<float*>(db + oOut);
    float* dD = dU + nL;
    int32_t* dS = reinterpret_cast<int32_t*>(dD + nL);
    hipLaunchKernelGGL(k_stereo_match, dim3((unsigned)((nL + 3) / 4)), dim3(256), 0, s, left->d_pyr.p, right->d_pyr.p,
                       left->d_lg.p, left->nlevels, reinterpret_cast<const float*>(db + oKL), db + oDL, nL,
                       reinterpret_cast<const float*>(db + oKR), db + oDR, nR, mb, mbf, dU, dD, dS, nullptr, nullptr);
    HIP_TRY(hipMemcpyAsync(hb + oOut, db + oOut, 3 * (size_t)nL * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(uRight, hb + oOut, (size_t)nL * 4);
    std::memcpy(depth, hb + oOut + (size_t)nL * 4, (size_t)nL * 4);
    const int32_t* sad = reinterpret_cast<const int32_t*>(hb + oOut + 2 * (size_t)nL * 4);
    // outlier cut (:952-966): matches whose SAD is >= 1.5*1.4*median are dropped
    std::vector<std::pair<int, int>> vDistIdx;
    for (int i = 0; i < nL; i++)
        if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
    if (vDistIdx.empty()) return 0;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}

// The same on what the two extractors' last calls left on the device: keypoints, descriptors, counts and both
// pyramids are read in place; only uRight / depth / SAD come back (one transfer).
// K-STEREO on what the two contexts' last extraction left on the device, queued on the left context's stream; the three
// result arrays go straight into the left context's pinned arena (or, ORBFE_ZEROCOPY=0, into a device arena + copy).
// *doneSeq (when asked for): the number K-STEREO will publish in left's completion word once its three arrays are in the pinned
// arena, or 0 (the caller then synchronises the stream)
static int stereo_resident_launch(orbfe_ctx* left, int imgL, orbfe_ctx* right, int imgR, float mb, float mbf,
                                  unsigned* doneSeq = nullptr)
{
    if (doneSeq) *doneSeq = 0u;
    if (left->lg.empty() || right->lg.empty() || !left->lastKps || !right->lastKps || imgL < 0 || imgR < 0 ||
        imgL >= left->lastImgs || imgR >= right->lastImgs || left->device != right->device ||
        left->rows != right->rows || left->cols != right->cols || left->nlevels != right->nlevels ||
        left->scaleFactor != right->scaleFactor || right->lastCap >= (1 << 20))
        return ORBFE_ERR_STATE;
    int r;
    if (!left->runStream && ((r = lane_join(left)) < 0 || (r = lane_join(right)) < 0)) return r; // (not for a frame on its lane)
    const size_t cap = (size_t)left->lastCap;
    if ((r = left->d_stereo.ensure(3 * cap)) < 0) return r;
    if ((r = left->h_stereo.ensure(3 * cap)) < 0) return r;
    hipStream_t s = run_stream(left);
    if (run_stream(right) != s) { // the right extractor's kernels must have finished: order the streams, not the host
        if (!left->evStereo) HIP_TRY(hipEventCreateWithFlags(&left->evStereo, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(left->evStereo, right->stream));
        HIP_TRY(hipStreamWaitEvent(s, left->evStereo, 0));
    }
    // the kernel writes its three result arrays (one value per left keypoint each) straight into the pinned host arena:
    // no download command between the kernel and the host's wait (ORBFE_ZEROCOPY=0: device arena + copy, for A/B)
    float* dU = left->zeroCopy ? left->h_stereo.dev() : left->d_stereo.p;
    float* dD = dU + cap;
    int32_t* dS = reinterpret_cast<int32_t*>(dD + cap);
    const int capL = left->lastCap, capR = right->lastCap;
    OrbDone done{nullptr, nullptr, 0u, 0u};
    if (doneSeq && left->zeroCopy && left->spinWait && left->h_stereo.coherent) {
        if (!left->d_done.p) {
            if ((r = left->d_done.ensure(80)) < 0) return r;
            HIP_TRY(hipMemsetAsync(left->d_done.p, 0, 80 * sizeof(unsigned), s));
            if ((r = left->h_done.ensure(16)) < 0) return r;
            left->h_done.p[0] = 0u;
        }
        if (left->h_done.coherent) {
            if (++left->doneSeq >= 0x80000000u) left->doneSeq = 1u;
            done = OrbDone{left->d_done.p + 16, left->h_done.dev(), left->doneSeq, (unsigned)((capL + 3) / 4)};
            *doneSeq = left->doneSeq;
        }
    }
    hipLaunchKernelGGL(k_stereo_match, dim3((unsigned)((capL + 3) / 4)), dim3(256), 0, s,
                       left->d_pyr.p + (size_t)imgL * left->pyrStride, right->d_pyr.p + (size_t)imgR * right->pyrStride,
                       left->d_lg.p, left->nlevels, left->lastKps + (size_t)imgL * capL * 7,
                       left->lastDesc + (size_t)imgL * capL * 32, capL, right->lastKps + (size_t)imgR * capR * 7,
                       right->lastDesc + (size_t)imgR * capR * 32, capR, mb, mbf, dU, dD, dS, left->lastN + imgL,
                       right->lastN + imgR, done);
    if (!left->zeroCopy) HIP_TRY(hipMemcpyAsync(left->h_stereo.p, dU, 3 * cap * sizeof(float), hipMemcpyDeviceToHost, s));
    return 0;
}
// ... and, once the stream has been waited for, the host part: the outlier cut of :952-966 over the nL results
static int stereo_resident_finish(orbfe_ctx* left, float* uRight, float* depth, int nL)
{
    const size_t cap = (size_t)left->lastCap;
    const float* hU = left->h_stereo.p;
    const float* hD = hU + cap;
    const int32_t* hS = reinterpret_cast<const int32_t*>(hD + cap);
    // sorted by SAD, the matches from the back down to the first one below 1.5*1.4*median go -- i.e. every match with
    // SAD >= that bound; the median is the element size/2 of the sorted list
    std::vector<int>& sads = left->stereoSads;
    sads.clear();
    for (int i = 0; i < nL; i++) {
        uRight[i] = hU[i];
        depth[i] = hD[i];
        if (hS[i] >= 0) sads.push_back(hS[i]);
    }
    if (sads.empty()) return 0;
    std::nth_element(sads.begin(), sads.begin() + (long)(sads.size() / 2), sads.end());
    const float thDist = 1.5f * 1.4f * (float)sads[sads.size() / 2];
    int kept = 0;
    for (int i = 0; i < nL; i++) {
        if (hS[i] < 0) continue;
        if ((float)hS[i] >= thDist) {
            uRight[i] = -1;
            depth[i] = -1;
        } else {
            kept++;
        }
    }
    return kept;
}

int orbfe_compute_stereo_matches_resident(orbfe_ctx* left, int imgL, orbfe_ctx* right, int imgR, float mb, float mbf,
                                          float* uRight, float* depth, int nL)
{
    if (!left || !right || nL < 0 || (nL && (!uRight || !depth)) || !(mb > 0)) return ORBFE_ERR_ARGS;
    if (nL > left->lastCap) return ORBFE_ERR_STATE;
    HIP_TRY(hipSetDevice(left->device));
    int r;
    unsigned seq = 0;
    if ((r = stereo_resident_launch(left, imgL, right, imgR, mb, mbf, &seq)) < 0) return r;
    if (!spin_done(left, seq)) {
        HIP_TRY(hipStreamSynchronize(left->stream));
        spin_settle(left);
    }
    return stereo_resident_finish(left, uRight, depth, nL);
}

// A rectified stereo frame in ONE call and ONE host wait: both images through orbfe_extract_batch's latency path and
// Frame::ComputeStereoMatches (K-STEREO) queued behind the extraction on the same stream -- what Frame::Frame (stereo,
// src/Frame.cc:119-122, :797-967) does with two threads, two extractors and a host loop.
int orbfe_extract_stereo_pair(orbfe_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int rows, int cols, size_t stride,
                              const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out,
                              float mb, float mbf, float* uRight, float* depth)
{
    if (!c || !imgL || !imgR || !kps || !desc || !n_out || !uRight || !depth || !(mb > 0)) return ORBFE_ERR_ARGS;
    if (c->slotSubmitted != c->slotRetired || c->pairSubmitted != c->pairRetired) return ORBFE_ERR_STATE; // submitted work must be waited for first
    const uint8_t* two[2] = {imgL, imgR};
    int r = host_submit(c, 2, two, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, false);
    if (r < 0) return r;
    // (the extraction is queued, c->last* describe its outputs: the matching goes behind it on the same stream)
    unsigned seq = 0;
    r = stereo_resident_launch(c, 0, c, 1, mb, mbf, &seq);
    c->slot[(c->slotSubmitted - 1) & 1].doneSeq = r < 0 ? 0u : seq; // (K-STEREO is the call's last kernel: its word ends the wait)
    const int w = host_wait(c); // always: the slot must be retired
    if (r < 0) return r;
    if (w < 0) return w;
    if (n_out[0] > cap_per_img) return ORBFE_ERR_STATE;
    return stereo_resident_finish(c, uRight, depth, n_out[0]);
}

// Stereo frames IN FLIGHT (round 5): the same work as orbfe_extract_stereo_pair, queued on the next batch lane -- its stream, its
// pyramids and quadtree buffers, its pinned result slab, its completion word -- and collected by _wait in submission order, so
// that the latency chain of one stereo frame (upload, four kernels on two images, K-STEREO: ~80 us on a nearly empty chip)
// runs beside its neighbours'.  Up to `lanes` frames (orbfe_set_lanes) are accepted before a _wait is due.
int orbfe_extract_stereo_pair_submit(orbfe_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int rows, int cols, size_t stride,
                                     const int* lap, orbfe_kp* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out,
                                     float mb, float mbf, float* uRight, float* depth)
{
    if (!c || !imgL || !imgR || !kps || !desc || !n_out || !uRight || !depth || !(mb > 0)) return ORBFE_ERR_ARGS;
    if (c->slotSubmitted != c->slotRetired) return ORBFE_ERR_STATE; // (the two-slot pipeline shares the slots)
    if (c->kb8On) return ORBFE_ERR_STATE;
    const int depthMax = std::max(1, std::min(c->lanes, ORBFE_MAX_LANES));
    if (c->pairSubmitted - c->pairRetired >= depthMax) return ORBFE_ERR_STATE; // every lane holds a frame: wait for one first
    if (depthMax == 1) { // a pipeline of one IS the blocking call (measured: a lone frame on a lane's stream is no faster)
        const int m = orbfe_extract_stereo_pair(c, imgL, imgR, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, mb, mbf,
                                                uRight, depth);
        if (m < 0) return m;
        c->pairBlockingResult = m;
        c->pairSubmitted++;
        return 0;
    }
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if (c->pairSubmitted == c->pairRetired && lanes_busy(c) && (r = lane_join(c)) < 0) return r; // (behind device-pointer calls)
    const int k = (int)(c->pairSubmitted % depthMax);
    orbfe_ctx::Lane& L = c->lane[k];
    if (!L.pairStream) HIP_TRY(hipStreamCreateWithFlags(&L.pairStream, hipStreamNonBlocking));
    if (!L.evJoin) HIP_TRY(hipEventCreateWithFlags(&L.evJoin, hipEventDisableTiming));
    lane_select(c, k);
    c->runStream = L.pairStream;
    const uint8_t* two[2] = {imgL, imgR};
    // No completion word here: with kernels of several queues on the chip the workgroups do not run on the XCDs their ids name,
    // every one of them releases for itself and the word is withheld (OrbDone) -- and a pipeline that is kept full gains nothing
    // from a wait that ends 5 us earlier.  _wait synchronises the frame's stream.
    const bool word = false;
    const int ks = k;
    r = host_submit_impl(c, 2, two, rows, cols, stride, lap, kps, desc, cap_per_img, n_out, mono_out, false, /*spinOk=*/word, ks);
    unsigned seq = 0;
    int rs = 0;
    if (r >= 0) {
        rs = stereo_resident_launch(c, 0, c, 1, mb, mbf, word ? &seq : nullptr);
        c->slot[ks].doneSeq = rs < 0 ? 0u : seq; // (K-STEREO is the frame's last kernel: its word ends the wait)
        c->slot[ks].uRight = uRight;
        c->slot[ks].depth = depth;
    }
    c->runStream = nullptr;
    L.pendingPair = true;
    c->pairLast = k;
    if (r <= -1000 || rs < 0) (void)hipStreamSynchronize(L.pairStream); // (something may be queued that reads the caller's images)
    if (r < 0) return r;
    if (rs < 0) {
        c->slot[ks].busy = false;
        return rs;
    }
    c->pairSubmitted++;
    return 0;
}

// Completes the OLDEST submitted stereo frame: returns its number of stereo matches (as orbfe_extract_stereo_pair does) with the
// arrays handed to that _submit filled.
int orbfe_extract_stereo_pair_wait(orbfe_ctx* c)
{
    if (!c) return ORBFE_ERR_ARGS;
    if (c->pairSubmitted == c->pairRetired) return ORBFE_ERR_STATE;
    HIP_TRY(hipSetDevice(c->device));
    const int depthMax = std::max(1, std::min(c->lanes, ORBFE_MAX_LANES));
    if (depthMax == 1) {
        c->pairRetired++;
        return c->pairBlockingResult;
    }
    const int k = (int)(c->pairRetired % depthMax);
    lane_select(c, k);
    const int ks = k;
    orbfe_ctx::HostSlot& sl = c->slot[ks];
    c->runStream = c->lane[k].pairStream;
    const int w = host_wait(c, ks);
    c->runStream = nullptr;
    c->pairRetired++;
    int m = w;
    if (w >= 0) {
        if (sl.n_out[0] > sl.cap) {
            m = ORBFE_ERR_STATE;
        } else { // (h_stereo is lane k's -- lane_select above --, laid out by the capacity of the frame's own submit)
            const int keepCap = c->lastCap;
            c->lastCap = sl.cap;
            m = stereo_resident_finish(c, sl.uRight, sl.depth, sl.n_out[0]);
            c->lastCap = keepCap;
        }
    }
    c->lane[k].pendingPair = false; // (the host has seen the frame's last kernel finish)
    if (c->pairLast >= 0) lane_select(c, c->pairLast); // "the last call" of the getters = the newest submitted frame
    return m;
}

int orbfe_profile_enable(orbfe_ctx* c, int on)
{
    if (!c) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    if (on && !c->evReady) {
        c->ev.resize((size_t)orbfe_ctx::kProfSets * (ORBFE_STAGE_COUNT + 1));
        for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
        c->evReady = true;
    }
    c->profile = on != 0;
    c->profEvery = on > 1 ? on : 1;
    if (on) c->profCalls = c->profSeen = 0;
    return 0;
}

int orbfe_profile_read(orbfe_ctx* c, float* ms)
{
    if (!c || !ms || !c->evReady || c->profCalls == 0) return ORBFE_ERR_STATE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int nset = (int)std::min<long>(c->profCalls, orbfe_ctx::kProfSets);
    double acc[ORBFE_STAGE_COUNT] = {0};
    for (int k = 0; k < nset; k++) {
        const hipEvent_t* e = &c->ev[(size_t)k * (ORBFE_STAGE_COUNT + 1)];
        for (int i = 0; i < ORBFE_STAGE_COUNT; i++) {
            float t = 0.f;
            if (c->packSkipped[k] && i == 3) continue; // no K-PACK in this set: the stage is empty ...
            HIP_TRY(hipEventElapsedTime(&t, e[(c->packSkipped[k] && i == 4) ? 3 : i], e[i + 1])); // ... and K-DESC starts at K-QT's end
            acc[i] += t;
        }
    }
    for (int i = 0; i < ORBFE_STAGE_COUNT; i++) ms[i] = (float)(acc[i] / nset);
    return nset;
}

int orbfe_debug_candidates(orbfe_ctx* c, int img, int level, uint32_t* out, int cap)
{
    if (!c || c->lg.empty() || level < 0 || level >= c->nlevels || img < 0 || img >= c->lastImgs) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    lane_quiesce(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    const OrbLevelGeom& L = c->lg[level];
    std::vector<int32_t> cnt(L.nCells);
    HIP_TRY(hipMemcpy(cnt.data(), c->d_cellCount.p + (size_t)img * c->nCells + L.cellBase, L.nCells * sizeof(int32_t),
                      hipMemcpyDeviceToHost));
    int n = 0;
    for (int ci = 0; ci < L.nCells; ci++) {
        const OrbCellGeom& g = c->cg[L.cellBase + ci];
        const int k = std::min(cnt[ci], std::max(cap - n, 0));
        if (k > 0)
            HIP_TRY(hipMemcpy(out + n, c->d_cand.p + (size_t)img * c->candStride + g.slotBase, (size_t)k * 4,
                              hipMemcpyDeviceToHost));
        n += cnt[ci];
    }
    return n;
}

int orbfe_debug_level_keypoints(orbfe_ctx* c, int img, int level, uint32_t* out, int cap)
{
    if (!c || c->lg.empty() || level < 0 || level >= c->nlevels || img < 0 || img >= c->lastImgs) return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    lane_quiesce(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, c->d_lvlCount.p + (size_t)img * ORBFE_MAX_LEVELS + level, sizeof(int32_t), hipMemcpyDeviceToHost));
    n &= 0xFFFF; // (high half: the level's count of lapping-range keypoints)
    const int k = std::min(n, cap);
    if (k > 0)
        HIP_TRY(hipMemcpy(out, c->d_lvlKp.p + (size_t)img * c->kpStride + c->lg[level].kpBase, (size_t)k * 4,
                          hipMemcpyDeviceToHost));
    return n;
}

int orbfe_debug_fixups(orbfe_ctx* c) { return c ? c->lastFixups : ORBFE_ERR_ARGS; }

#ifdef ORBFE_STEREO_TIMING
extern "C" int orbfe_debug_stereo_times(unsigned long long* out8)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stereoTimes), 8 * sizeof(unsigned long long)));
    static const unsigned long long zeros[8] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stereoTimes), zeros, sizeof(zeros)));
    return 0;
}
#endif

#ifdef ORBFE_QT_TIMING
extern "C" int orbfe_debug_qt_times(unsigned long long* out64)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_qtTimes), 64 * sizeof(unsigned long long)));
    static const unsigned long long zeros[64] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_qtTimes), zeros, sizeof(zeros))); // the next read sees one run only
    return 0;
}
#endif

#ifdef ORBFE_FAST_TIMING
extern "C" int orbfe_debug_fast_times(unsigned long long* out16)
{
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> all(4096 * 16);
    HIP_TRY(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_fastTimes), all.size() * sizeof(unsigned long long)));
    for (int k = 0; k < 16; k++) out16[k] = 0;
    for (size_t i = 0; i < all.size(); i++) out16[i & 15] += all[i];
    std::fill(all.begin(), all.end(), 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fastTimes), all.data(), all.size() * sizeof(unsigned long long)));
    return 0;
}
#endif

// GaussianBlur's output under one keypoint of the last call: re-runs the descriptor kernel for that image (same
// results) with the tap armed, so that the fused blur is compared with the oracle's blurred level directly.
int orbfe_debug_blurred_patch(orbfe_ctx* c, int img, int kp_index, uint8_t* out37x37)
{
    if (!c || c->lg.empty() || !out37x37 || img < 0 || img >= c->lastImgs || kp_index < 0 || kp_index >= c->lastCap ||
        !c->lastKps)
        return ORBFE_ERR_ARGS;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<uint8_t> d;
    int r = d.ensure(37 * 37);
    if (r < 0) return r;
    HIP_TRY(hipMemsetAsync(d.p, 0, 37 * 37, c->stream));
    const TrigTabs trigTab = c->trigMode == ORBFE_TRIG_LIBM ? trig_table(c->device, c->stream) : TrigTabs{nullptr, nullptr};
    int tapSum = 0;
    for (int i = 0; i < 7; i++) tapSum += c->taps[i];
    float* kps = const_cast<float*>(c->lastKps);
    uint8_t* desc = const_cast<uint8_t*>(c->lastDesc);
    const dim3 grid((unsigned)((c->maxKp + ORBFE_DESC_WPW * ORBFE_DESC_KPW - 1) / (ORBFE_DESC_WPW * ORBFE_DESC_KPW)), 1u);
    const int32_t* const dm = c->lastPacked ? c->d_destMap.p : nullptr;
#define ORBFE_DBG_LAUNCH(SAT)                                                                                              \
    hipLaunchKernelGGL((k_orient_blur_desc<0, SAT, true>), grid, dim3(64 * ORBFE_DESC_WPW), 0, c->stream, c->d_pyr.p, c->pyrStride, c->d_ds.p, \
                       c->maxKp, c->d_lvlKp.p, c->kpStride, c->d_lvlCount.p, c->nlevels, c->d_lvlPre.p, dm, c->lastCap, kps, desc,         \
                       c->d_taps.p, c->d_patternF.p, c->d_fix.p, 0, 0, img, 0, trigTab.codes, trigTab.full, c->atanFma, d.p, \
                       kp_index)
    if (tapSum > 256) ORBFE_DBG_LAUNCH(true);
    else ORBFE_DBG_LAUNCH(false);
#undef ORBFE_DBG_LAUNCH
    hipError_t e = hipMemcpyAsync(out37x37, d.p, 37 * 37, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

int orbfe_debug_trig(orbfe_ctx* c, const float* angles_deg, int n, float* a_out, float* b_out)
{
    if (!c || !angles_deg || !a_out || !b_out || n < 0) return ORBFE_ERR_ARGS;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const TrigTabs tab = c->trigMode == ORBFE_TRIG_LIBM ? trig_table(c->device, c->stream) : TrigTabs{nullptr, nullptr};
    DevBuf<float> d;
    int r = d.ensure((size_t)3 * n);
    if (r < 0) return r;
    // (a test hands over millions of angles in pageable arrays: orbfe_pageable.h)
    hipError_t e = orbfe_pageable::up(d.p, angles_deg, (size_t)n * sizeof(float), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_trig, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d.p, n, tab.codes, tab.full, d.p + n,
                           d.p + 2 * (size_t)n);
        e = orbfe_pageable::down(a_out, d.p + n, (size_t)n * sizeof(float), c->stream);
    }
    if (e == hipSuccess) e = orbfe_pageable::down(b_out, d.p + 2 * (size_t)n, (size_t)n * sizeof(float), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    if (e != hipSuccess) return -(1000 + (int)e);
    return tab.full ? 2 : tab.codes ? 1 : 0; /* 2: libm values, 1: libm codes, 0: no table */
}

// ---- the libm table's cache file, host side only (no device needed): what the CPU tests exercise
int orbfe_debug_trig_cache_path(char* out, int cap)
{
    if (!out || cap < 1) return ORBFE_ERR_ARGS;
    const std::string p = trig_cache_path(libm_fingerprint());
    if ((int)p.size() + 1 > cap) return ORBFE_ERR_ARGS;
    std::memcpy(out, p.c_str(), p.size() + 1);
    return (int)p.size();
}
size_t orbfe_debug_trig_cache_payload_bytes(void) { return ((size_t)(ORBFE_TRIG_U1 - ORBFE_TRIG_U0 + 1u) + 1) / 2; }
int orbfe_debug_trig_cache_write(const char* path, const uint8_t* payload, size_t bytes)
{
    if (!path || !payload || bytes != orbfe_debug_trig_cache_payload_bytes()) return ORBFE_ERR_ARGS;
    return trig_cache_store(path, libm_fingerprint(), payload, bytes, trig_payload_sum(payload, bytes)) ? 0 : ORBFE_ERR_STATE;
}
int orbfe_debug_trig_cache_check(const char* path, const char** why)
{
    static const char* const kSum = "payload checksum mismatch";
    static const char* const kRead = "short read";
    if (why) *why = nullptr;
    if (!path) return ORBFE_ERR_ARGS;
    const size_t bytes = orbfe_debug_trig_cache_payload_bytes();
    TrigCacheHeader hd;
    const char* w = nullptr;
    const int fd = trig_cache_open(path, libm_fingerprint(), bytes, &hd, &w);
    if (fd < 0) {
        if (why) *why = w;
        return ORBFE_ERR_STATE;
    }
    std::vector<uint8_t> buf(8u << 20); // (a multiple of 8: the word index carries over from chunk to chunk)
    uint64_t sum = 0;
    bool ok = true;
    for (size_t off = 0; ok && off < bytes; off += buf.size()) {
        const size_t n = std::min(buf.size(), bytes - off);
        ok = read_fully(fd, buf.data(), n);
        if (ok) sum += trig_payload_sum(buf.data(), n, off / 8);
    }
    close(fd);
    if (!ok || sum != hd.payloadSum) {
        if (why) *why = ok ? kSum : kRead;
        return ORBFE_ERR_STATE;
    }
    return 0;
}

} // extern "C"
