// Threaded exerciser of the CPU-reachable host surface of liborbfe: built by tests/test_sanitized_product.py
// against the ThreadSanitizer / AddressSanitizer builds of the library (make -C csrc tsan | asan) and run WITHOUT a GPU.
// What it drives from four threads at once:
//   * the pinned-memory registry (orbfe_host_register / _unregister / orbfe_host_alloc / _free: process-wide table + mutex);
//     without a device the HIP calls inside fail, the registry's bookkeeping still runs,
//   * orbfe_create / orbfe_keyframe_create / matcher entry points' argument and no-device error paths (every one must hand
//     back an error code and release what it took),
//   * the shard / ring / job-offset / layout arithmetic of include/orbfe_mc.h (pure functions, shared nothing),
//   * orbfe_mc_create with the host transport for a world of ONE (shared-memory segment created, attached, torn down) -- the
//     world-2 exchange itself runs under the sanitizers in tests/test_multicam_gloo.py (two processes),
//   * orbfe_error_string and the trig-cache file checks (orbfe_debug_trig_cache_check on files written concurrently).
// Exit code 0 and no sanitizer report = pass.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"
#include "../../include/orbfe_debug.h"
#include "../../include/orbfe_mc.h"

static std::atomic<int> g_fail{0};
#define EXPECT(c)                                                              \
    do {                                                                       \
        if (!(c)) {                                                            \
            std::fprintf(stderr, "threads_cabi: %s failed (line %d)\n", #c, __LINE__); \
            g_fail++;                                                          \
        }                                                                      \
    } while (0)

static void worker(int t, const std::string& dir)
{
    std::vector<uint8_t> buf(1 << 16), desc(64 * 32, (uint8_t)t);
    for (int it = 0; it < 200; it++) {
        // pinned registry
        void* p = orbfe_host_alloc(4096 + 64 * (size_t)t);
        if (p) orbfe_host_free(p); // (no device: the allocation fails and nothing is registered)
        (void)orbfe_host_register(buf.data(), buf.size());
        (void)orbfe_host_unregister(buf.data());
        // argument / no-device error paths
        orbfe_ctx* c = nullptr;
        EXPECT(orbfe_create(&c, 0, 1.2f, 8, 20, 7, 0) == ORBFE_ERR_ARGS && c == nullptr);
        EXPECT(orbfe_create(&c, 1000, 1.2f, 8, 20, 7, 0) < 0 && c == nullptr); // ORBFE_ERR_NODEV here
        EXPECT(orbfe_create(nullptr, 1000, 1.2f, 8, 20, 7, 0) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_extract(nullptr, buf.data(), 16, 16, 16, 0, 0, nullptr, nullptr, 0, nullptr) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_set_lanes(nullptr, 2) == ORBFE_ERR_ARGS && orbfe_lanes_join(nullptr) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_extract_stereo_pair_wait(nullptr) == ORBFE_ERR_ARGS);
        orbfe_keyframe* K = nullptr;
        orbfe_keyframe_args a;
        std::memset(&a, 0, sizeof a);
        EXPECT(orbfe_keyframe_create(&K, 0, &a) == ORBFE_ERR_ARGS && K == nullptr);
        const uint32_t node[1] = {3};
        const int32_t off[2] = {0, 2}, ind[2] = {1, 99999}; // an index beyond n
        std::vector<uint8_t> mask(64, 1);
        a.desc = desc.data();
        a.n = 64;
        a.mask = mask.data();
        a.fv.nn = 1;
        a.fv.node_ids = node;
        a.fv.offsets = off;
        a.fv.indices = ind;
        EXPECT(orbfe_keyframe_create(&K, 0, &a) == ORBFE_ERR_ARGS && K == nullptr);
        uint16_t D[4];
        EXPECT(orbfe_hamming_pairs(0, desc.data(), 0, desc.data(), 2, D) >= 0); // empty operand: no device needed
        EXPECT(orbfe_hamming_pairs(0, desc.data(), 2, desc.data(), 2, D) < 0);  // needs one: an error code, not a crash
        EXPECT(std::strlen(orbfe_error_string(ORBFE_ERR_ARGS)) > 0 && std::strlen(orbfe_error_string(-12345)) > 0);
        // orbfe_mc arithmetic
        orbfe_mc_layout_t lay;
        EXPECT(orbfe_mc_layout(8, 1008, &lay) == 0 && lay.slab_bytes % 256 == 0 && lay.count_off >= (size_t)8 * 1008 * 32);
        for (int world = 1; world <= 8; world *= 2) {
            int covered = 0, next = 0;
            for (int r = 0; r < world; r++) {
                int first = -1, count = -1;
                EXPECT(orbfe_mc_shard(64 + t, world, r, &first, &count) == 0 && first == next && count >= 0);
                next = first + count;
                covered += count;
            }
            EXPECT(covered == 64 + t);
            std::vector<int32_t> pairs(2 * 8 * 3);
            const int hops[3] = {1, 2, 5};
            const int np = orbfe_mc_ring_pairs(world, 8, t % world, hops, 3, pairs.data());
            EXPECT(np == 8 * 3);
            for (int i = 0; i < np; i++) EXPECT(pairs[2 * i + 1] >= 0 && pairs[2 * i + 1] < world * 8);
        }
        EXPECT(orbfe_mc_layout(0, 1008, &lay) == ORBFE_ERR_ARGS && orbfe_mc_shard(8, 0, 0, nullptr, nullptr) == ORBFE_ERR_ARGS);
        // trig cache: a file of the right size but garbage must be rejected (and never crash the reader)
        if (it == 0 && t == 0) { // (65 MB of payload: once)
            const std::string path = dir + "/trig_" + std::to_string(t) + "_" + std::to_string(it) + ".bin";
            std::vector<uint8_t> payload(orbfe_debug_trig_cache_payload_bytes(), (uint8_t)(it + t));
            EXPECT(orbfe_debug_trig_cache_write(path.c_str(), payload.data(), payload.size()) == 0);
            const char* why = nullptr;
            EXPECT(orbfe_debug_trig_cache_check(path.c_str(), &why) == 0); // a well-formed file (header + checksum)
            FILE* f = std::fopen(path.c_str(), "r+b");
            if (f) {
                std::fseek(f, 4096 + 17 * t, SEEK_SET);
                const uint8_t x = (uint8_t)~payload[0];
                std::fwrite(&x, 1, 1, f);
                std::fclose(f);
                EXPECT(orbfe_debug_trig_cache_check(path.c_str(), &why) != 0 && why != nullptr); // one flipped byte
            }
            std::remove(path.c_str());
        }
    }
    // the shared-memory transport for a world of one: segment created, one exchange (the gathered view is the slab), destroyed
    char id[ORBFE_MC_ID_BYTES];
    std::memset(id, 0, sizeof id);
    if (orbfe_mc_unique_id(ORBFE_MC_HOST, id) == 0) {
        orbfe_mc* m = nullptr;
        const int r = orbfe_mc_create(&m, nullptr, id, 0, 1, 8, 1008, ORBFE_MC_HOST); // (no context: host-memory slabs)
        EXPECT(r == 0 && m != nullptr);
        if (m) {
            orbfe_mc_layout_t lay;
            EXPECT(orbfe_mc_layout(8, 1008, &lay) == 0);
            std::vector<uint8_t> slab(lay.slab_bytes, (uint8_t)(t + 1));
            const uint8_t* g = nullptr;
            EXPECT(orbfe_mc_exchange_host(m, slab.data(), &g) == 0 && g && std::memcmp(g, slab.data(), lay.slab_bytes) == 0);
            orbfe_mc_destroy(m);
        }
    }
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::vector<std::thread> ts;
    for (int t = 0; t < 4; t++) ts.emplace_back(worker, t, dir);
    for (auto& t : ts) t.join();
    // Round 6: the use counts behind orbfe_keyframe_destroy / orbfe_frame_destroy (a destroy under a search is deferred, a stale
    // handle is refused) -- the table itself, driven by three searcher threads against an owner that destroys and re-creates
    // eight handles two thousand times; and the orbfe_bow_* argument paths (no device: every call must return a code)
    EXPECT(orbfe_debug_handle_table_selftest(3, 8, 2000) == 0);
    {
        orbfe_bow* b = nullptr;
        EXPECT(orbfe_bow_create(&b, nullptr, 100) == ORBFE_ERR_ARGS && b == nullptr);
        EXPECT(orbfe_bow_create(nullptr, nullptr, 100) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_compute_bow(nullptr, nullptr, 0, 4) == ORBFE_ERR_ARGS);
        orbfe_fv fv;
        orbfe_bow_view v;
        EXPECT(orbfe_bow_fv(nullptr, &fv) == ORBFE_ERR_ARGS && orbfe_bow_host(nullptr, &v) == ORBFE_ERR_ARGS &&
               orbfe_bow_device(nullptr, &v) == ORBFE_ERR_ARGS);
        orbfe_bow_destroy(nullptr);
        EXPECT(orbfe_vocab_set_types(nullptr, 0, 0) == ORBFE_ERR_ARGS);
        // a stale / never-created handle address is refused by every entry point that takes one, not dereferenced
        alignas(64) static unsigned char fake[256];
        orbfe_keyframe* stale = reinterpret_cast<orbfe_keyframe*>(fake);
        uint8_t m[4] = {1, 1, 1, 1};
        EXPECT(orbfe_keyframe_set_mask(stale, m) == ORBFE_ERR_ARGS);
        orbfe_keyframe_destroy(stale);
        orbfe_frame_destroy(reinterpret_cast<orbfe_frame*>(fake));
        orbfe_proj_args pa;
        std::memset(&pa, 0, sizeof pa);
        int32_t q[1], f[1];
        EXPECT(orbfe_search_projection_frame(reinterpret_cast<orbfe_frame*>(fake), &pa, q, f) == ORBFE_ERR_ARGS);
        // ... and the same with an address whose memory has been FREED: under AddressSanitizer a single read of it is a report
        // (heap-use-after-free), so a silent run is the proof that handles are looked up, not dereferenced -- keyframe, frame
        // and (since the BoW handles moved into the same table) orbfe_bow entry points alike
        void* gone = std::malloc(1024);
        std::free(gone);
        orbfe_bow* sb = reinterpret_cast<orbfe_bow*>(gone);
        uint8_t d32[32] = {0};
        EXPECT(orbfe_compute_bow(sb, d32, 1, 4) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_bow_set_lazy_norm(sb, 1) == ORBFE_ERR_ARGS);
        EXPECT(orbfe_bow_host(sb, &v) == ORBFE_ERR_ARGS && orbfe_bow_device(sb, &v) == ORBFE_ERR_ARGS && orbfe_bow_fv(sb, &fv) == ORBFE_ERR_ARGS);
        orbfe_bow_destroy(sb);
        EXPECT(orbfe_keyframe_set_mask(reinterpret_cast<orbfe_keyframe*>(gone), m) == ORBFE_ERR_ARGS);
        orbfe_keyframe_destroy(reinterpret_cast<orbfe_keyframe*>(gone));
        orbfe_frame_destroy(reinterpret_cast<orbfe_frame*>(gone));
        EXPECT(orbfe_search_projection_frame(reinterpret_cast<orbfe_frame*>(gone), &pa, q, f) == ORBFE_ERR_ARGS);
        // a vector that NAMES such a handle (ORBFE_FV_RESIDENT) in the calls that take an orbfe_fv
        orbfe_fv named;
        named.nn = ORBFE_FV_RESIDENT;
        named.node_ids = reinterpret_cast<const uint32_t*>(gone);
        named.offsets = nullptr;
        named.indices = nullptr;
        orbfe_keyframe_args ka;
        std::memset(&ka, 0, sizeof ka);
        ka.n = 1;
        ka.desc = d32;
        ka.mask = m;
        ka.fv = named;
        orbfe_keyframe* kout = nullptr;
        EXPECT(orbfe_keyframe_create(&kout, 0, &ka) == ORBFE_ERR_ARGS && kout == nullptr);
    }
    std::printf("threads_cabi: %d failures\n", g_fail.load());
    return g_fail.load() ? 1 : 0;
}
