// C++ host of the multi-GPU path (include/orbfe_mc.h): what a multi-camera ORB-SLAM3 process per GPU would run.
// Spawns `world` ranks (fork before any HIP call), each of which extracts its shard of the frames straight into its
// slab, exchanges the slabs (ORBFE_MC_RCCL: ncclAllGather over xGMI, one GPU per rank; ORBFE_MC_HOST: shared memory, the
// ranks may share a device) with up to three batches in flight, matches its frames against the next camera of the ring, and
// checks everything against plain single-GPU calls of the same C ABI (orbfe_extract_batch, orbfe_bfknn2):
//   * rank r's slab inside the gathered buffer == the descriptors / counts a local extraction of r's frames gives,
//   * the ring matching == orbfe_bfknn2 on those descriptor sets.
// usage: test_multicam frames.raw rows cols nframes world transport(0 = RCCL, 1 = host) [nfeatures]
// Built with hipcc and run by tests/test_gpu_multicam.py (world 1 with RCCL, world 2 over shared memory on one GPU).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../include/orbfe_mc.h"

struct Shared {
    std::atomic<int> idReady;
    unsigned char id[ORBFE_MC_ID_BYTES];
};

#define CHECK(cond, what)                                                      \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::fprintf(stderr, "rank %d: FAILED %s (line %d)\n", rank, what, __LINE__); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static int run_rank(int rank, int world, int transport, const std::vector<unsigned char>& frames, int rows, int cols, int total,
                    int nfeatures, Shared* sh)
{
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, "hipGetDeviceCount");
    const int dev = transport == ORBFE_MC_RCCL ? rank % ndev : 0;
    CHECK(transport != ORBFE_MC_RCCL || world <= ndev, "RCCL needs one GPU per rank");
    const int per = total / world;
    orbfe_ctx* ctx = nullptr;
    CHECK(orbfe_create(&ctx, nfeatures, 1.2f, 8, 20, 7, dev) == 0, "orbfe_create");
    const int cap = orbfe_max_keypoints(ctx, rows, cols);
    CHECK(cap > 0, "orbfe_max_keypoints");
    // the exchange id: rank 0 makes it, the others read it from the shared page (any broadcast would do)
    if (world > 1 || transport == ORBFE_MC_HOST) {
        if (rank == 0) {
            CHECK(orbfe_mc_unique_id(transport, sh->id) == 0, "orbfe_mc_unique_id");
            sh->idReady.store(1);
        }
        while (sh->idReady.load() == 0) usleep(1000);
    }
    orbfe_mc* mc = nullptr;
    CHECK(orbfe_mc_create(&mc, ctx, (world > 1 || transport == ORBFE_MC_HOST) ? sh->id : nullptr, rank, world, per, cap,
                          transport) == 0, "orbfe_mc_create");
    orbfe_mc_layout_t lay;
    CHECK(orbfe_mc_layout(per, cap, &lay) == 0, "orbfe_mc_layout");
    const size_t fsz = (size_t)rows * cols;
    unsigned char* d_img = nullptr;
    CHECK(hipSetDevice(dev) == hipSuccess, "hipSetDevice");
    CHECK(hipMalloc((void**)&d_img, per * fsz) == hipSuccess, "hipMalloc");
    CHECK(hipMemcpy(d_img, frames.data() + (size_t)rank * per * fsz, per * fsz, hipMemcpyHostToDevice) == hipSuccess, "upload");
    // five batches through the four slab pairs, up to ORBFE_MC_MAX_IN_FLIGHT (3) in flight; a fourth one is refused
    orbfe_mc_view_t v;
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit 0");
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit 1");
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit 2");
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == ORBFE_ERR_STATE, "a fourth batch in flight is refused");
    CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0 && v.batch == 0, "wait 0");
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit 3");
    CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0 && v.batch == 1, "wait 1");
    CHECK(orbfe_mc_extract_exchange_submit(mc, d_img, rows, cols, cols, fsz, 0, 0) == 0, "submit 4 (reuses the slab pair of batch 0)");
    CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0 && v.batch == 2, "wait 2");
    CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0 && v.batch == 3, "wait 3");
    CHECK(orbfe_mc_extract_exchange_wait(mc, &v) == 0 && v.batch == 4, "wait 4");
    CHECK(v.slab_bytes == lay.slab_bytes, "slab size");
    std::vector<unsigned char> g((size_t)world * lay.slab_bytes);
    CHECK(hipMemcpy(g.data(), v.gathered, g.size(), hipMemcpyDeviceToHost) == hipSuccess, "download gathered");
    // ring matching before anything else touches the context
    const int hop = 1;
    std::vector<int32_t> idx((size_t)per * cap * 2), dist((size_t)per * cap * 2);
    CHECK(orbfe_mc_match_ring(mc, &hop, 1, idx.data(), dist.data()) == per, "orbfe_mc_match_ring");
    // reference: every rank's frames through the plain batched call on THIS GPU
    std::vector<orbfe_kp> kps((size_t)total * cap);
    std::vector<unsigned char> desc((size_t)total * cap * 32);
    std::vector<int> n(total), mono(total), lap(2 * total, 0);
    std::vector<const unsigned char*> ptr(total);
    for (int i = 0; i < total; i++) ptr[i] = frames.data() + (size_t)i * fsz;
    CHECK(orbfe_extract_batch(ctx, total, ptr.data(), rows, cols, cols, lap.data(), kps.data(), desc.data(), cap, n.data(),
                              mono.data()) == 0, "orbfe_extract_batch");
    long nkp = 0;
    for (int r = 0; r < world; r++)
        for (int j = 0; j < per; j++) {
            const int f = r * per + j;
            const unsigned char* slab = g.data() + (size_t)r * lay.slab_bytes;
            int cnt;
            std::memcpy(&cnt, slab + lay.count_off + 4 * (size_t)j, 4);
            CHECK(cnt == n[f] && cnt > 50, "gathered count");
            CHECK(std::memcmp(slab + (size_t)j * cap * 32, desc.data() + (size_t)f * cap * 32, (size_t)cnt * 32) == 0, "gathered descriptors");
            nkp += cnt;
        }
    std::vector<int32_t> pairs(2 * per);
    CHECK(orbfe_mc_ring_pairs(world, per, rank, &hop, 1, pairs.data()) == per, "orbfe_mc_ring_pairs");
    for (int k = 0; k < per; k++) {
        const int q = rank * per + pairs[2 * k], t = pairs[2 * k + 1];
        CHECK(t == (q + 1) % total, "ring partner");
        std::vector<int32_t> ri((size_t)n[q] * 2), rd((size_t)n[q] * 2);
        CHECK(orbfe_bfknn2(dev, desc.data() + (size_t)q * cap * 32, n[q], desc.data() + (size_t)t * cap * 32, n[t], ri.data(),
                           rd.data()) == 0, "orbfe_bfknn2");
        CHECK(std::memcmp(ri.data(), idx.data() + (size_t)k * cap * 2, ri.size() * 4) == 0, "ring matching indices");
        CHECK(std::memcmp(rd.data(), dist.data() + (size_t)k * cap * 2, rd.size() * 4) == 0, "ring matching distances");
        for (int i = n[q]; i < cap; i++) CHECK(idx[((size_t)k * cap + i) * 2] == -1, "rows past the count are -1");
    }
    std::printf("rank %d of %d ok: %d frames per rank, cap %d, %ld keypoints gathered, slab %zu B, transport %s\n", rank, world, per,
                cap, nkp, (size_t)lay.slab_bytes, transport == ORBFE_MC_RCCL ? "rccl" : "host");
    orbfe_mc_destroy(mc);
    (void)hipFree(d_img);
    orbfe_destroy(ctx);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s frames.raw rows cols nframes world transport [nfeatures]\n", argv[0]);
        return 2;
    }
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), total = atoi(argv[4]), world = atoi(argv[5]), transport = atoi(argv[6]);
    const int nfeatures = argc > 7 ? atoi(argv[7]) : 500;
    if (world < 1 || total % world != 0) return 2;
    std::vector<unsigned char> frames((size_t)total * rows * cols);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) return 3;
    std::fclose(f);
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) return 4;
    std::memset((void*)sh, 0, sizeof(Shared));
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        const pid_t p = fork(); // (no HIP call has been made yet: every rank initialises the runtime itself)
        if (p == 0) {
            const int rc = run_rank(r, world, transport, frames, rows, cols, total, nfeatures, sh);
            std::fflush(stdout);
            std::fflush(stderr);
            _exit(rc);
        }
        if (p < 0) return 5;
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++;
    }
    if (bad) {
        std::fprintf(stderr, "%d of %d ranks failed\n", bad, world);
        return 1;
    }
    std::printf("all %d ranks ok\n", world);
    return 0;
}
