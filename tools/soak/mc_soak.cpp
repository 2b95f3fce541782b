// C++-only soak of the state the round-5 abort was seen in: a context with batch lanes under an orbfe_mc handle
// (RCCL, world 1 -- the process's n-th communicator -- or the host transport), a few batches in flight, ring matching with
// a download, handle and context destroyed, and THEN what the dying process was doing when the signal came: a fresh pageable
// host buffer of 1.44 MB copied to a fresh device allocation on the NULL stream (what torch's Tensor.cuda() issues).  No Python,
// no torch, no numpy.  A handler prints the C backtrace of whichever thread raises SIGABRT / SIGSEGV / SIGBUS.
//
// usage: mc_soak transport(0 = RCCL, 1 = host) seconds [max_iterations] [rows cols]
// Prints a progress line every 100 iterations (gpurun's hang detector wants output) and a summary; exit code 0 = no fault and
// every iteration's results equal the first iteration's.
#include <hip/hip_runtime.h>

#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <execinfo.h>
#include <unistd.h>

#include "../../include/orbfe_mc.h"

static void on_fatal(int sig)
{
    void* frames[64];
    const char* msg = sig == SIGABRT ? "\n*** mc_soak: SIGABRT, backtrace of the raising thread:\n" : "\n*** mc_soak: fatal signal, backtrace:\n";
    (void)!write(2, msg, strlen(msg));
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

#define CHECK(cond, what)                                                                     \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            std::fprintf(stderr, "iteration %ld: FAILED %s (line %d)\n", it, what, __LINE__); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

static void make_frame(unsigned char* p, int rows, int cols, unsigned seed)
{
    // rectangles of random grey on a gradient: corners for FAST at every level
    std::mt19937 g(seed);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) p[(size_t)y * cols + x] = (unsigned char)(64 + (x + y) % 64);
    for (int k = 0; k < 120; k++) {
        const int w = 6 + (int)(g() % 50), h = 6 + (int)(g() % 50);
        const int x0 = (int)(g() % (unsigned)(cols - w)), y0 = (int)(g() % (unsigned)(rows - h));
        const unsigned char v = (unsigned char)(g() % 256);
        for (int y = y0; y < y0 + h; y++) std::memset(p + (size_t)y * cols + x0, v, (size_t)w);
    }
}

int main(int argc, char** argv)
{
    long it = -1;
    if (argc < 3) {
        std::fprintf(stderr, "usage: mc_soak transport(0 rccl | 1 host) seconds [max_iterations] [rows cols]\n");
        return 2;
    }
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fatal;
    sigaction(SIGABRT, &sa, nullptr);
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
    const int transport = atoi(argv[1]);
    const double seconds = atof(argv[2]);
    const long maxIt = argc > 3 ? atol(argv[3]) : 1000000;
    const int rows = argc > 5 ? atoi(argv[4]) : 240, cols = argc > 5 ? atoi(argv[5]) : 376;
    const int frames = 16;
    const size_t fsz = (size_t)rows * cols;
    std::vector<unsigned char> imgs(frames * fsz);
    for (int i = 0; i < frames; i++) make_frame(imgs.data() + i * fsz, rows, cols, 1200u + i);
    CHECK(hipSetDevice(0) == hipSuccess, "hipSetDevice");
    unsigned char* d_img = nullptr;
    CHECK(hipMalloc((void**)&d_img, frames * fsz) == hipSuccess, "hipMalloc");
    CHECK(hipMemcpy(d_img, imgs.data(), frames * fsz, hipMemcpyHostToDevice) == hipSuccess, "upload");
    std::vector<int32_t> idx0, dist0;
    std::vector<unsigned char> slab0;
    std::mt19937 rng(7);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    double tCreate = 0, tMc = 0, tRun = 0, tDestroy = 0, tAfter = 0;
    for (it = 0; it < maxIt && elapsed() < seconds; it++) {
        auto ta = std::chrono::steady_clock::now();
        auto lap = [&](double& acc) {
            const auto tb = std::chrono::steady_clock::now();
            acc += std::chrono::duration<double>(tb - ta).count();
            ta = tb;
        };
        orbfe_ctx* ctx = nullptr;
        CHECK(orbfe_create(&ctx, 400, 1.2f, 8, 20, 7, 0) == 0, "orbfe_create");
        const int lanes = 2 + (int)(it % 3 == 2); // 2, 2, 3, ...
        CHECK(orbfe_set_lanes(ctx, lanes) == 0, "orbfe_set_lanes");
        if (it & 1) CHECK(orbfe_set_lane_input_guard(ctx, 0) == 0, "orbfe_set_lane_input_guard");
        const int cap = orbfe_max_keypoints(ctx, rows, cols);
        CHECK(cap > 0, "orbfe_max_keypoints");
        lap(tCreate);
        orbfe_mc* mc = nullptr;
        CHECK(orbfe_mc_create(&mc, ctx, nullptr, 0, 1, frames, cap, transport) == 0, "orbfe_mc_create");
        lap(tMc);
        orbfe_mc_layout_t lay;
        CHECK(orbfe_mc_layout(frames, cap, &lay) == 0, "orbfe_mc_layout");
        orbfe_mc_view_t v;
        int inflight = 0;
        for (int b = 0; b < 5; b++) {
            if (inflight == ORBFE_MC_MAX_IN_FLIGHT) {
                CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0, "wait");
                inflight--;
            }
            CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit");
            inflight++;
        }
        while (inflight) {
            CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0, "wait");
            inflight--;
        }
        // ring matching with a download into fresh pageable arrays (numpy's np.empty), the gathered slab through the null stream
        const int hops[2] = {1, 2};
        const size_t nres = (size_t)2 * frames * cap * 2;
        int32_t* idx = (int32_t*)std::malloc(nres * 4);
        int32_t* dist = (int32_t*)std::malloc(nres * 4);
        unsigned char* g = (unsigned char*)std::malloc(lay.slab_bytes);
        CHECK(idx && dist && g, "malloc");
        CHECK(orbfe_mc_match_ring(mc, hops, 2, idx, dist) == 2 * frames, "orbfe_mc_match_ring");
        CHECK(hipMemcpy(g, v.gathered, lay.slab_bytes, hipMemcpyDeviceToHost) == hipSuccess, "gathered slab down");
        if (it == 0) {
            idx0.assign(idx, idx + nres);
            dist0.assign(dist, dist + nres);
            slab0.assign(g, g + lay.slab_bytes);
            const int32_t* counts = reinterpret_cast<const int32_t*>(g + lay.count_off);
            long total = 0;
            for (int i = 0; i < frames; i++) total += counts[i];
            CHECK(total > 50 * frames, "the frames have keypoints");
            std::printf("transport %s: %ld keypoints per batch of %d frames, cap %d, slab %zu bytes\n", transport == ORBFE_MC_RCCL ? "rccl" : "host",
                        total, frames, cap, lay.slab_bytes);
        } else {
            const int32_t* counts = reinterpret_cast<const int32_t*>(g + lay.count_off);
            const int32_t* counts0 = reinterpret_cast<const int32_t*>(slab0.data() + lay.count_off);
            CHECK(std::memcmp(counts, counts0, 4 * frames) == 0, "counts equal the first iteration's");
            for (int i = 0; i < frames; i++)
                CHECK(std::memcmp(g + (size_t)i * cap * 32, slab0.data() + (size_t)i * cap * 32, (size_t)counts[i] * 32) == 0,
                      "descriptors equal the first iteration's");
            for (int k = 0; k < 2 * frames; k++) {
                const int n = counts[k % frames];
                CHECK(std::memcmp(idx + (size_t)k * cap * 2, idx0.data() + (size_t)k * cap * 2, (size_t)n * 8) == 0, "knn-2 indices equal");
                CHECK(std::memcmp(dist + (size_t)k * cap * 2, dist0.data() + (size_t)k * cap * 2, (size_t)n * 8) == 0, "knn-2 distances equal");
            }
        }
        std::free(idx);
        std::free(dist);
        std::free(g);
        lap(tRun);
        orbfe_mc_destroy(mc);
        orbfe_destroy(ctx);
        lap(tDestroy);
        // what the dying process did next: fresh pageable arrays, then Tensor.cuda() = device allocation + null-stream copy
        const unsigned r = rng();
        if (r % 64 == 0) usleep(1000000);         // the gap the round-5 process had (a second of numpy work)
        else if (r % 4 == 0) usleep(r % 20000);
        for (int s = 0; s < 2; s++) {
            unsigned char* h = (unsigned char*)std::malloc(frames * fsz);
            CHECK(h, "malloc");
            std::memcpy(h, imgs.data(), frames * fsz);
            unsigned char* d = nullptr;
            CHECK(hipMalloc((void**)&d, frames * fsz) == hipSuccess, "hipMalloc after destroy");
            CHECK(hipMemcpyAsync(d, h, frames * fsz, hipMemcpyHostToDevice, nullptr) == hipSuccess, "null-stream copy after destroy");
            CHECK(hipStreamSynchronize(nullptr) == hipSuccess, "null-stream sync after destroy");
            CHECK(hipFree(d) == hipSuccess, "hipFree");
            std::free(h);
        }
        lap(tAfter);
        if ((it + 1) % 100 == 0) {
            std::printf("%ld iterations, %.0f s (per iteration: create %.1f ms, mc_create %.1f, batches %.1f, destroy %.1f, after %.1f)\n", it + 1,
                        elapsed(), 1e3 * tCreate / (it + 1), 1e3 * tMc / (it + 1), 1e3 * tRun / (it + 1), 1e3 * tDestroy / (it + 1),
                        1e3 * tAfter / (it + 1));
            std::fflush(stdout);
        }
    }
    (void)hipFree(d_img);
    std::printf("mc_soak: %ld clean iterations (transport %s) in %.0f s\n", it, transport == ORBFE_MC_RCCL ? "rccl" : "host", elapsed());
    return 0;
}
