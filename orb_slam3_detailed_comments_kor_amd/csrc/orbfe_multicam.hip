/*
 * orbfe_multicam.hip -- the multi-GPU / multi-camera path behind the C ABI (include/orbfe_mc.h).
 *
 * One process per GPU.  Every rank extracts its own frames straight into a fixed-size slab (the extractor's device
 * entry point writes descriptors and counts where it is told to), ONE ncclAllGather per batch on a side stream gives
 * every rank every camera's descriptors, and the cross-camera matching (K-KNN2F, orbfe_bfknn2_frames_device) reads the
 * gathered buffer in place, sharded by query frame.  Two batches may be in flight: the all-gather of batch i runs
 * beside the extraction of batch i+1 (separate HIP streams, ordered by events only).  No reduction anywhere.
 *
 * RCCL is loaded with dlopen when the first ORBFE_MC_RCCL handle is created (librccl.so is half a gigabyte: a
 * single-GPU user of liborbfe.so never maps it).  The host transport does the same exchange through POSIX shared
 * memory: for ranks that share a device, for world == 1, and for the bookkeeping tests in a container without a GPU.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/orbfe_mc.h"
#include "orbfe_pageable.h"

#define MC_HIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return -(1000 + (int)_e);    \
    } while (0)

extern "C" int orbfe_get_stream(orbfe_ctx*, void** hip_stream, int* device);
extern "C" int orbfe_lanes_record(orbfe_ctx*, void* hip_event);
extern "C" int orbfe_internal_exchange_hint(orbfe_ctx*, int on);
extern "C" int orbfe_internal_bfknn2_frames(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                                            int32_t* d_idx, int32_t* d_dist, int flags /* 1: shares the chip, 2: fills the tails */);

// RCCL is resolved at run time (dlopen: the single-GPU library has no link dependency on it), so the few facts about its ABI
// this file relies on are mirrored by hand below.  Where the header exists at build time they are CHECKED:
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#include <type_traits>
static_assert(sizeof(ncclUniqueId) == ORBFE_MC_ID_BYTES, "ORBFE_MC_ID_BYTES must be sizeof(ncclUniqueId)");
static_assert((int)ncclUint8 == 1 && (int)ncclInt8 == 0, "kNcclUint8 mirrors ncclUint8");
static_assert((int)ncclSuccess == 0, "0 is ncclSuccess");
static_assert(std::is_same<decltype(&ncclAllGather),
                           ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value,
              "ncclAllGather's signature changed");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value,
              "ncclCommInitRank's signature changed");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId's signature changed");
#define ORBFE_RCCL_HEADER_CHECKED 1
#endif
#endif

namespace {

inline size_t align_up_mc(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- the handful of RCCL entry points the exchange needs, resolved at run time ----
struct Rccl {
    typedef struct ncclComm* comm_t;
    struct unique_id { char internal[ORBFE_MC_ID_BYTES]; };
    int (*GetUniqueId)(unique_id*) = nullptr;
    int (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /* ncclDataType_t */, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* so = nullptr;
    bool ok = false;
};
Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.so) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    });
    return r;
}
const int kNcclUint8 = 1; // ncclUint8 / ncclChar's unsigned twin in rccl.h (ncclInt8 = 0, ncclUint8 = 1)

// ---- host transport: one shared-memory segment per exchange id ----
// [ header: arrive counter, generation | world slabs, double buffered ]
struct ShmHeader {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    std::atomic<int> poisoned; // a rank gave up in a barrier: the arrive count is off for good, every later call fails
    int pad[12];
};
struct HostXfer {
    std::string name;
    int fd = -1;
    uint8_t* base = nullptr;
    size_t bytes = 0;
    int world = 1, rank = 0;
    size_t slabBytes = 0;
    long round = 0;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    uint8_t* slabs(int parity) const { return base + 256 + (size_t)parity * world * slabBytes; }
    int open_(const char* id, int rank_, int world_, size_t slabBytes_)
    {
        world = world_;
        rank = rank_;
        slabBytes = slabBytes_;
        bytes = 256 + 2 * (size_t)world * slabBytes;
        char nm[64];
        std::snprintf(nm, sizeof nm, "/orbfe_mc_%.40s", id);
        name = nm;
        fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return ORBFE_ERR_STATE;
        if (ftruncate(fd, (off_t)bytes) != 0) return ORBFE_ERR_STATE; // (a fresh segment is zero-filled: the header starts at 0)
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) return ORBFE_ERR_STATE;
        base = (uint8_t*)p;
        hdr()->attached.fetch_add(1);
        return 0;
    }
    // sense-reversing barrier over the ranks; gives up after `seconds` (a rank that died must not hang the others)
    int barrier(double seconds = 60.0)
    {
        ShmHeader* h = hdr();
        if (h->poisoned.load()) return ORBFE_ERR_STATE;
        const int gen = h->generation.load();
        if (h->arrived.fetch_add(1) + 1 == world) {
            h->arrived.store(0);
            h->generation.fetch_add(1);
            return 0;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (h->generation.load() == gen) {
            if (h->poisoned.load()) return ORBFE_ERR_STATE;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
                // this rank's arrival stays counted, so the segment cannot be used again: mark it, for every rank
                h->poisoned.store(1);
                return ORBFE_ERR_STATE;
            }
            std::this_thread::yield();
        }
        return 0;
    }
    int exchange(const uint8_t* slab, const uint8_t** gathered)
    {
        const int parity = (int)(round & 1);
        std::memcpy(slabs(parity) + (size_t)rank * slabBytes, slab, slabBytes);
        round++;
        int r = barrier(); // every rank's slab of this round is in place; the other parity is free to be overwritten
        if (r < 0) return r;
        *gathered = slabs(parity);
        return 0;
    }
    void close_()
    {
        if (base) {
            const bool last = hdr()->attached.fetch_sub(1) == 1;
            munmap(base, bytes);
            base = nullptr;
            if (last) shm_unlink(name.c_str());
        }
        if (fd >= 0) close(fd);
        fd = -1;
    }
};

} // namespace

struct orbfe_mc {
    orbfe_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t sCtx = nullptr;  // the extractor's stream
    hipStream_t sComm = nullptr; // the collective's stream
    int rank = 0, world = 1, frames = 0, cap = 0, transport = ORBFE_MC_HOST;
    orbfe_mc_layout_t lay{};
    Rccl::comm_t comm = nullptr;
    HostXfer host;
    struct Buf {
        uint8_t* slab = nullptr;     // device (host for a ctx == NULL handle)
        uint8_t* gathered = nullptr;
        hipEvent_t evProduced = nullptr, evProduced2 = nullptr, evGathered = nullptr;
        int status = 0;
        long batch = -1;
        orbfe_knn2_job* d_jobs = nullptr; // job table of the last hops used on this buffer
        int njobs = 0;
        std::vector<int> jobHops;
    } buf[ORBFE_MC_SLOTS];
    // (round 4: four slab pairs and three batches in flight instead of two and two.  With two, the host could submit batch
    // k + 1 only after the collective of batch k - 1; once a context runs its batches on two lanes the second lane finishes --
    // and the collective starts -- late in the step, the host was released late, and the GPU ran dry between batches:
    // 0.268 instead of 0.206 ms per step on the one-GPU rehearsal.)
    uint8_t* h_stage[ORBFE_MC_SLOTS] = {}; // pinned staging of the host transport per buffer: slab out | gathered in
    orbfe_kp* d_kps[ORBFE_MC_SLOTS] = {};
    int32_t* d_mono[ORBFE_MC_SLOTS] = {};
    long submitted = 0, retired = 0;
    int lastView = -1; // buffer index of the batch last returned by _wait
    int32_t *d_idx = nullptr, *d_dist = nullptr;
    int matchCap = 0, lastPairs = 0;
};

extern "C" {

int orbfe_mc_layout(int frames_per_rank, int cap, orbfe_mc_layout_t* out)
{
    if (!out || frames_per_rank <= 0 || cap <= 0) return ORBFE_ERR_ARGS;
    out->desc_bytes = (size_t)frames_per_rank * cap * 32;
    out->count_off = out->desc_bytes;
    out->slab_bytes = align_up_mc(out->desc_bytes + 4 * (size_t)frames_per_rank, 256); // every rank's rows stay aligned
    return 0;
}

int orbfe_mc_shard(int nframes, int world, int rank, int* first, int* count)
{
    if (nframes < 0 || world <= 0 || rank < 0 || rank >= world || !first || !count) return ORBFE_ERR_ARGS;
    const int base = nframes / world, rem = nframes % world;
    *first = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return 0;
}

int orbfe_mc_ring_pairs(int world, int frames_per_rank, int rank, const int* hops, int nhops, int32_t* pairs)
{
    if (world <= 0 || frames_per_rank <= 0 || rank < 0 || rank >= world || nhops < 0 || (nhops && !hops) || !pairs)
        return ORBFE_ERR_ARGS;
    const long total = (long)world * frames_per_rank;
    int k = 0;
    for (int h = 0; h < nhops; h++)
        for (int i = 0; i < frames_per_rank; i++, k++) {
            long g = ((long)rank * frames_per_rank + i + hops[h]) % total;
            if (g < 0) g += total;
            pairs[2 * k] = i;
            pairs[2 * k + 1] = (int32_t)g;
        }
    return k;
}

int orbfe_mc_job_offsets(int frames_per_rank, int cap, const int32_t* pairs, int npairs, int64_t* offsets)
{
    orbfe_mc_layout_t L;
    int r = orbfe_mc_layout(frames_per_rank, cap, &L);
    if (r < 0) return r;
    if (npairs < 0 || (npairs && (!pairs || !offsets))) return ORBFE_ERR_ARGS;
    for (int k = 0; k < npairs; k++) {
        const int qi = pairs[2 * k], g = pairs[2 * k + 1];
        if (qi < 0 || qi >= frames_per_rank || g < 0) return ORBFE_ERR_ARGS;
        const int rk = g / frames_per_rank, j = g % frames_per_rank;
        offsets[4 * k + 0] = (int64_t)qi * cap * 32;
        offsets[4 * k + 1] = (int64_t)L.count_off + 4 * (int64_t)qi;
        offsets[4 * k + 2] = (int64_t)rk * (int64_t)L.slab_bytes + (int64_t)j * cap * 32;
        offsets[4 * k + 3] = (int64_t)rk * (int64_t)L.slab_bytes + (int64_t)L.count_off + 4 * (int64_t)j;
    }
    return npairs;
}

int orbfe_mc_unique_id(int transport, void* id128)
{
    if (!id128) return ORBFE_ERR_ARGS;
    std::memset(id128, 0, ORBFE_MC_ID_BYTES);
    if (transport == ORBFE_MC_RCCL) {
        Rccl& R = rccl();
        if (!R.ok) return ORBFE_ERR_NODEV;
        Rccl::unique_id id;
        if (R.GetUniqueId(&id) != 0) return ORBFE_MC_ERR_RCCL;
        std::memcpy(id128, &id, ORBFE_MC_ID_BYTES);
        return 0;
    }
    if (transport != ORBFE_MC_HOST) return ORBFE_ERR_ARGS;
    // a name nobody else on this host uses: pid, time and a counter, printable
    static std::atomic<unsigned> seq{0};
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf((char*)id128, ORBFE_MC_ID_BYTES, "h%ld_%llx_%u", (long)getpid(), (unsigned long long)now, seq.fetch_add(1));
    return 0;
}

void orbfe_mc_destroy(orbfe_mc* m)
{
    if (!m) return;
    if (m->ctx) {
        (void)hipSetDevice(m->device);
        if (m->sComm) (void)hipStreamSynchronize(m->sComm);
        (void)hipStreamSynchronize(m->sCtx);
    }
    if (m->comm && rccl().ok) (void)rccl().CommDestroy(m->comm);
    m->host.close_();
    for (auto& b : m->buf) {
        if (m->ctx) {
            if (b.slab) (void)hipFree(b.slab);
            if (b.gathered) (void)hipFree(b.gathered);
            if (b.d_jobs) (void)hipFree(b.d_jobs);
            if (b.evProduced) (void)hipEventDestroy(b.evProduced);
            if (b.evProduced2) (void)hipEventDestroy(b.evProduced2);
            if (b.evGathered) (void)hipEventDestroy(b.evGathered);
        } else {
            std::free(b.slab);
        }
    }
    for (int k = 0; k < ORBFE_MC_SLOTS; k++) {
        if (m->d_kps[k]) (void)hipFree(m->d_kps[k]);
        if (m->d_mono[k]) (void)hipFree(m->d_mono[k]);
    }
    for (int k = 0; k < ORBFE_MC_SLOTS; k++)
        if (m->h_stage[k]) (void)hipHostFree(m->h_stage[k]);
    if (m->d_idx) (void)hipFree(m->d_idx);
    if (m->d_dist) (void)hipFree(m->d_dist);
    if (m->sComm) (void)hipStreamDestroy(m->sComm);
    delete m;
}

int orbfe_mc_create(orbfe_mc** out, orbfe_ctx* ctx, const void* id128, int rank, int world, int frames_per_rank, int cap,
                    int transport)
{
    if (!out) return ORBFE_ERR_ARGS;
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world || frames_per_rank <= 0 || cap <= 0) return ORBFE_ERR_ARGS;
    if (transport != ORBFE_MC_RCCL && transport != ORBFE_MC_HOST) return ORBFE_ERR_ARGS;
    if (!ctx && transport != ORBFE_MC_HOST) return ORBFE_ERR_ARGS;
    if (world > 1 && !id128) return ORBFE_ERR_ARGS;
    orbfe_mc* m = new (std::nothrow) orbfe_mc();
    if (!m) return ORBFE_ERR_STATE;
    m->ctx = ctx;
    m->rank = rank;
    m->world = world;
    m->frames = frames_per_rank;
    m->cap = cap;
    m->transport = transport;
    (void)orbfe_mc_layout(frames_per_rank, cap, &m->lay);
    int r = 0;
    auto fail = [&](int code) {
        orbfe_mc_destroy(m);
        return code;
    };
    if (ctx) {
        void* st = nullptr;
        if ((r = orbfe_get_stream(ctx, &st, &m->device)) < 0) return fail(r);
        m->sCtx = (hipStream_t)st;
        (void)orbfe_internal_exchange_hint(ctx, 1); // (lane streams created from here on take the priorities that suit an exchange)
        if (hipSetDevice(m->device) != hipSuccess) return fail(ORBFE_ERR_NODEV);
        {
            // The collective's stream gets the HIGHEST priority: priorities have hardware queues of their own, and the runtime
            // deals the streams of one priority round-robin to only four queues (GPU_MAX_HW_QUEUES) -- with the context's two
            // lanes, torch's streams and this one, the collective's event wait ended up in FRONT of the second lane's kernels in
            // a shared queue and stalled them (round 4, one-GPU rehearsal: 0.262 ms per step, 0.175 with eight queues).  The
            // all-gather is tiny and on the critical path of the other ranks, so priority is right for it anyway.
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
                (void)hipGetLastError();
                least = greatest = 0;
            }
            int prio = greatest;
            if (hipStreamCreateWithPriority(&m->sComm, hipStreamNonBlocking, prio) != hipSuccess) {
                (void)hipGetLastError();
                if (hipStreamCreateWithFlags(&m->sComm, hipStreamNonBlocking) != hipSuccess) return fail(ORBFE_ERR_STATE);
            }
        }
        for (int k = 0; k < ORBFE_MC_SLOTS; k++) {
            orbfe_mc::Buf& b = m->buf[k];
            if (hipMalloc((void**)&b.slab, m->lay.slab_bytes) != hipSuccess ||
                hipMalloc((void**)&b.gathered, (size_t)world * m->lay.slab_bytes) != hipSuccess ||
                hipMalloc((void**)&m->d_kps[k], (size_t)frames_per_rank * cap * sizeof(orbfe_kp)) != hipSuccess ||
                hipMalloc((void**)&m->d_mono[k], (size_t)frames_per_rank * sizeof(int32_t)) != hipSuccess ||
                hipEventCreateWithFlags(&b.evProduced, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&b.evGathered, hipEventDisableTiming) != hipSuccess)
                return fail(ORBFE_ERR_STATE);
            (void)hipMemset(b.slab, 0, m->lay.slab_bytes);
            (void)hipMemset(b.gathered, 0, (size_t)world * m->lay.slab_bytes);
        }
    } else {
        for (auto& b : m->buf) {
            b.slab = (uint8_t*)std::calloc(1, m->lay.slab_bytes);
            if (!b.slab) return fail(ORBFE_ERR_STATE);
        }
    }
    if (transport == ORBFE_MC_RCCL) {
        Rccl& R = rccl();
        if (!R.ok) return fail(ORBFE_ERR_NODEV);
        Rccl::unique_id id;
        if (id128) std::memcpy(&id, id128, ORBFE_MC_ID_BYTES);
        else if (R.GetUniqueId(&id) != 0) return fail(ORBFE_MC_ERR_RCCL); // world == 1 without an id
        const int e = R.CommInitRank(&m->comm, world, id, rank);
        if (e != 0) { // (the reason goes to stderr only when asked for: a library does not print on its caller's behalf)
            if (getenv("ORBFE_VERBOSE"))
                std::fprintf(stderr, "orbfe_mc: ncclCommInitRank failed: %s\n", R.GetErrorString ? R.GetErrorString(e) : "?");
            m->comm = nullptr;
            return fail(ORBFE_MC_ERR_RCCL);
        }
    } else if (world > 1 || !ctx) {
        char idz[ORBFE_MC_ID_BYTES + 1];
        std::memset(idz, 0, sizeof idz);
        if (id128) std::memcpy(idz, id128, ORBFE_MC_ID_BYTES);
        else std::snprintf(idz, sizeof idz, "solo%ld", (long)getpid());
        if ((r = m->host.open_(idz, rank, world, m->lay.slab_bytes)) < 0) return fail(r);
        for (int k = 0; ctx && k < ORBFE_MC_SLOTS; k++)
            if (hipHostMalloc((void**)&m->h_stage[k], (size_t)(world + 1) * m->lay.slab_bytes, hipHostMallocDefault) != hipSuccess)
                return fail(ORBFE_ERR_STATE);
        if ((r = m->host.barrier()) < 0) return fail(r); // everybody is attached before the first exchange
    }
    *out = m;
    return 0;
}

int orbfe_mc_exchange_host(orbfe_mc* m, const uint8_t* slab, const uint8_t** gathered)
{
    if (!m || !slab || !gathered || !m->host.base) return ORBFE_ERR_ARGS;
    return m->host.exchange(slab, gathered);
}

int orbfe_mc_extract_exchange_submit(orbfe_mc* m, const uint8_t* d_imgs, int rows, int cols, size_t pitch,
                                     size_t img_stride_bytes, int lap0, int lap1)
{
    if (!m || !m->ctx || !d_imgs) return ORBFE_ERR_ARGS;
    if (m->submitted - m->retired >= ORBFE_MC_MAX_IN_FLIGHT) return ORBFE_ERR_STATE;
    MC_HIP_TRY(hipSetDevice(m->device));
    {
        // (the extractor's stream is asked for at every submit -- orbfe_set_stream after orbfe_mc_create replaces it,
        // and a cached handle would then name a destroyed stream)
        void* st = nullptr;
        const int rs = orbfe_get_stream(m->ctx, &st, nullptr);
        if (rs < 0) return rs;
        if ((hipStream_t)st != m->sCtx) {
            if (m->submitted != m->retired) return ORBFE_ERR_STATE; // batches in flight were queued on the old stream
            m->sCtx = (hipStream_t)st;
        }
    }
    const int k = (int)(m->submitted % ORBFE_MC_SLOTS);
    orbfe_mc::Buf& b = m->buf[k];
    // the slab pair was last read by the collective (and the matcher) of ORBFE_MC_SLOTS batches ago: the extractor's stream waits
    // for that collective; the matcher runs on the extractor's stream itself
    if (b.batch >= 0) MC_HIP_TRY(hipStreamWaitEvent(m->sCtx, b.evGathered, 0));
    int r = orbfe_extract_batch_device(m->ctx, m->frames, d_imgs, rows, cols, pitch, img_stride_bytes, lap0, lap1, m->d_kps[k],
                                       b.slab, m->cap, reinterpret_cast<int32_t*>(b.slab + m->lay.count_off), m->d_mono[k]);
    if (r < 0) return r;
    MC_HIP_TRY(hipEventRecord(b.evProduced, m->sCtx));
    MC_HIP_TRY(hipStreamWaitEvent(m->sComm, b.evProduced, 0));
    {
        // a context with two lanes (orbfe_set_lanes): the second half of the batch is produced on the context's second stream;
        // the collective waits for that lane too, the extractor's stream is not held back
        if (!b.evProduced2) MC_HIP_TRY(hipEventCreateWithFlags(&b.evProduced2, hipEventDisableTiming));
        const int two = orbfe_lanes_record(m->ctx, (void*)b.evProduced2);
        if (two < 0) return two;
        if (two == 1) MC_HIP_TRY(hipStreamWaitEvent(m->sComm, b.evProduced2, 0));
    }
    b.status = 0;
    if (m->transport == ORBFE_MC_RCCL) {
        const int e = rccl().AllGather(b.slab, b.gathered, m->lay.slab_bytes, kNcclUint8, m->comm, m->sComm);
        if (e != 0) {
            // The collective was not queued: the batch is NOT in flight (nothing to wait for), the slot stays free, and the
            // caller gets the code at once -- the other ranks' matching calls will not complete, which is theirs to time out on.
            if (getenv("ORBFE_VERBOSE"))
                std::fprintf(stderr, "orbfe_mc: ncclAllGather failed: %s\n", rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
            return ORBFE_MC_ERR_RCCL;
        }
    } else if (m->world == 1) {
        MC_HIP_TRY(hipMemcpyAsync(b.gathered, b.slab, m->lay.slab_bytes, hipMemcpyDeviceToDevice, m->sComm));
    } else {
        // host transport: slab -> pinned -> shared memory, barrier, shared memory -> pinned -> gathered.  The host part
        // runs in _wait (it blocks on the other ranks); here only the download is queued.
        MC_HIP_TRY(hipMemcpyAsync(m->h_stage[k], b.slab, m->lay.slab_bytes, hipMemcpyDeviceToHost, m->sComm));
    }
    MC_HIP_TRY(hipEventRecord(b.evGathered, m->sComm));
    b.batch = m->submitted;
    m->submitted++;
    return 0;
}

int orbfe_mc_extract_exchange_wait(orbfe_mc* m, orbfe_mc_view_t* view)
{
    if (!m || !m->ctx || !view) return ORBFE_ERR_ARGS;
    if (m->submitted == m->retired) return ORBFE_ERR_STATE;
    MC_HIP_TRY(hipSetDevice(m->device));
    const int k = (int)(m->retired % ORBFE_MC_SLOTS);
    orbfe_mc::Buf& b = m->buf[k];
    MC_HIP_TRY(hipEventSynchronize(b.evGathered));
    if (m->transport == ORBFE_MC_HOST && m->world > 1) {
        const uint8_t* g = nullptr;
        int r = m->host.exchange(m->h_stage[k], &g);
        if (r < 0) return r;
        std::memcpy(m->h_stage[k] + m->lay.slab_bytes, g, (size_t)m->world * m->lay.slab_bytes);
        MC_HIP_TRY(hipMemcpyAsync(b.gathered, m->h_stage[k] + m->lay.slab_bytes, (size_t)m->world * m->lay.slab_bytes,
                                  hipMemcpyHostToDevice, m->sComm));
        MC_HIP_TRY(hipEventRecord(b.evGathered, m->sComm));
        MC_HIP_TRY(hipEventSynchronize(b.evGathered));
    }
    m->retired++;
    m->lastView = k;
    if (b.status < 0) return b.status;
    // The device error word (a quadtree list overflow, ruled out by the bounds of SURVEY.md A.9) can only be read by
    // waiting for the context's whole stream: done when nothing else is queued behind this batch -- with another batch
    // in flight the wait would cost exactly the overlap the two slab pairs exist for.
    if (m->submitted == m->retired) {
        const int r = orbfe_sync(m->ctx);
        if (r < 0) return r;
    }
    view->gathered = b.gathered;
    view->slab = b.slab;
    view->d_kps = m->d_kps[k];
    view->d_mono = m->d_mono[k];
    view->slab_bytes = m->lay.slab_bytes;
    view->batch = b.batch;
    return 0;
}

static int mc_jobs(orbfe_mc* m, orbfe_mc::Buf& b, const int* hops, int nhops)
{
    if (b.d_jobs && (int)b.jobHops.size() == nhops && std::equal(b.jobHops.begin(), b.jobHops.end(), hops)) return b.njobs;
    const int np = nhops * m->frames;
    std::vector<int32_t> pairs(2 * (size_t)np);
    std::vector<int64_t> off(4 * (size_t)np);
    int r = orbfe_mc_ring_pairs(m->world, m->frames, m->rank, hops, nhops, pairs.data());
    if (r < 0) return r;
    if ((r = orbfe_mc_job_offsets(m->frames, m->cap, pairs.data(), np, off.data())) < 0) return r;
    std::vector<orbfe_knn2_job> jobs((size_t)np);
    for (int k = 0; k < np; k++) {
        jobs[k].q_desc = b.slab + off[4 * k + 0];
        jobs[k].q_count = reinterpret_cast<const int32_t*>(b.slab + off[4 * k + 1]);
        jobs[k].t_desc = b.gathered + off[4 * k + 2];
        jobs[k].t_count = reinterpret_cast<const int32_t*>(b.gathered + off[4 * k + 3]);
    }
    if (b.d_jobs) (void)hipFree(b.d_jobs);
    b.d_jobs = nullptr;
    MC_HIP_TRY(hipMalloc((void**)&b.d_jobs, std::max<size_t>(1, jobs.size()) * sizeof(orbfe_knn2_job)));
    MC_HIP_TRY(hipMemcpy(b.d_jobs, jobs.data(), jobs.size() * sizeof(orbfe_knn2_job), hipMemcpyHostToDevice));
    b.jobHops.assign(hops, hops + nhops);
    b.njobs = np;
    if (np > m->matchCap) {
        if (m->d_idx) (void)hipFree(m->d_idx);
        if (m->d_dist) (void)hipFree(m->d_dist);
        m->d_idx = m->d_dist = nullptr;
        MC_HIP_TRY(hipMalloc((void**)&m->d_idx, (size_t)np * m->cap * 2 * sizeof(int32_t)));
        MC_HIP_TRY(hipMalloc((void**)&m->d_dist, (size_t)np * m->cap * 2 * sizeof(int32_t)));
        m->matchCap = np;
    }
    return np;
}

int orbfe_mc_match_ring_async(orbfe_mc* m, const int* hops, int nhops, long batch)
{
    if (!m || !m->ctx || nhops <= 0 || !hops) return ORBFE_ERR_ARGS;
    MC_HIP_TRY(hipSetDevice(m->device));
    int k = -1;
    for (int i = 0; i < ORBFE_MC_SLOTS; i++)
        if (m->buf[i].batch == batch) k = i;
    if (k < 0) return ORBFE_ERR_STATE;
    orbfe_mc::Buf& b = m->buf[k];
    const int np = mc_jobs(m, b, hops, nhops);
    if (np < 0) return np;
    // the launch is queued on the extractor's stream, after the collective that fills the gathered buffer
    MC_HIP_TRY(hipStreamWaitEvent(m->sCtx, b.evGathered, 0));
    // (with batches in flight the kernel shares the CUs with their extraction: it then asks for no more LDS than it uses; the rows
    // between a frame's count and cap are set to -1 by the kernel itself -- two 0.5-MB clearing commands per launch until round 5)
    int r = orbfe_internal_bfknn2_frames(m->device, m->sCtx, b.d_jobs, np, m->cap, m->d_idx, m->d_dist,
                                         (m->submitted != m->retired ? 1 : 0) | 2);
    if (r < 0) return r;
    m->lastPairs = np;
    return np;
}

int orbfe_mc_match_ring(orbfe_mc* m, const int* hops, int nhops, int32_t* idx, int32_t* dist)
{
    if (!m || !m->ctx || m->lastView < 0) return m && m->ctx ? ORBFE_ERR_STATE : ORBFE_ERR_ARGS;
    const int np = orbfe_mc_match_ring_async(m, hops, nhops, m->buf[m->lastView].batch);
    if (np < 0) return np;
    const size_t bytes = (size_t)np * m->cap * 2 * sizeof(int32_t);
    // (the caller's arrays may be pageable: in pieces through page-locked memory of this thread, orbfe_pageable.h)
    if (idx) MC_HIP_TRY(orbfe_pageable::down(idx, m->d_idx, bytes, m->sCtx));
    if (dist) MC_HIP_TRY(orbfe_pageable::down(dist, m->d_dist, bytes, m->sCtx));
    MC_HIP_TRY(hipStreamSynchronize(m->sCtx));
    return np;
}

int orbfe_mc_match_outputs(orbfe_mc* m, const int32_t** d_idx, const int32_t** d_dist, int* npairs)
{
    if (!m || !m->ctx) return ORBFE_ERR_ARGS;
    if (d_idx) *d_idx = m->d_idx;
    if (d_dist) *d_dist = m->d_dist;
    if (npairs) *npairs = m->lastPairs;
    return 0;
}

} // extern "C"
