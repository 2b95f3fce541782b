// Does a flag written to page-locked host memory arrive behind the results other workgroups wrote there?  (hipcc -O2
// --offload-arch=gfx950 -pthread -o tools/flag_order_probe tools/flag_order_probe.hip)
// Every iteration a kernel of W workgroups x 4 wavefronts scatters the iteration number over a page-locked array (4-byte
// stores, like a matcher kernel's match rows), the workgroups count themselves on the device and the last one writes the
// iteration number to a flag word behind a system-scope fence; the host spins on the flag and then looks at the array at once.
//   mode 0  every wavefront waits for its own store acknowledgements (s_waitcnt vmcnt(0)) before it counts
//   mode 1  ... and the wavefront that completes its WORKGROUP's count does a system-scope release before the workgroup counts
//   mode 2  a system-scope release by every wavefront
// with and without two host threads saturating the link in both directions.  Prints the iterations in which the host saw a
// stale word.  The library's completion word: csrc/orbfe_matcher.hip (mode 1), csrc/orbfe_kernels.hip (per XCD).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k(unsigned* __restrict__ res, int n, unsigned it, unsigned* ctr, unsigned* flag, int mode, unsigned nwg)
{
    __shared__ unsigned wgCnt;
    if (threadIdx.x == 0) wgCnt = 0u;
    __syncthreads();
    // scattered rows: thread t of the grid owns words t, t + T, ... (permuted so that neighbours in a wavefront are far apart)
    const unsigned T = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    for (unsigned i = t; i < (unsigned)n; i += T) res[(i * 97u) % (unsigned)n] = it;
    if (mode == 2) __threadfence_system();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned closes = 0u;
    if ((threadIdx.x & 63) == 0) closes = atomicAdd(&wgCnt, 1u) + 1u == 4u ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane(closes)) return;
    if (mode == 1) __threadfence_system();
    if ((threadIdx.x & 63) == 0 && atomicAdd(ctr, 1u) + 1u == nwg) {
        *ctr = 0u;
        __threadfence_system();
        *(volatile unsigned*)flag = it;
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 100000, n = 4099 /* (a prime: the permutation above is one) */, W = 150;
    unsigned *hRes, *hFlag, *dRes, *dFlag, *ctr;
    CK(hipHostMalloc(&hRes, n * 4 + 64, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostMalloc(&hFlag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&dRes, hRes, 0));
    CK(hipHostGetDevicePointer((void**)&dFlag, hFlag, 0));
    CK(hipMalloc(&ctr, 64));
    CK(hipMemset(ctr, 0, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::atomic<bool> stop{false};
    auto loader = [&](int dir) {
        void *h, *d;
        CK(hipHostMalloc(&h, 32 << 20, 0));
        CK(hipMalloc(&d, 32 << 20));
        hipStream_t ls;
        CK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
        while (!stop.load()) {
            if (dir) CK(hipMemcpyAsync(d, h, 32 << 20, hipMemcpyHostToDevice, ls));
            else CK(hipMemcpyAsync(h, d, 32 << 20, hipMemcpyDeviceToHost, ls));
            CK(hipStreamSynchronize(ls));
        }
    };
    unsigned seq = 0;
    for (int load = 0; load < 2; load++) {
        std::vector<std::thread> th;
        stop = false;
        if (load) {
            th.emplace_back(loader, 0);
            th.emplace_back(loader, 1);
        }
        for (int mode = 0; mode < 3; mode++) {
            long staleIters = 0, staleWords = 0, timeouts = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < iters; it++) {
                ++seq;
                hipLaunchKernelGGL(k, dim3(W), dim3(256), 0, s, dRes, n, seq, ctr, dFlag, mode, (unsigned)W);
                const volatile unsigned* f = hFlag;
                long spins = 0;
                while (*f != seq && ++spins < 200000000L) __builtin_ia32_pause();
                if (*f != seq) { timeouts++; CK(hipStreamSynchronize(s)); continue; }
                std::atomic_thread_fence(std::memory_order_acquire);
                long bad = 0;
                for (int i = 0; i < n; i++) bad += ((volatile unsigned*)hRes)[i] != seq;
                if (bad) { staleIters++; staleWords += bad; }
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("link %s, mode %d (%s): %ld of %d iterations saw stale words (%ld words), %ld timeouts, %.2f us per iteration\n", load ? "LOADED" : "idle",
                   mode, mode == 0 ? "acknowledgements only" : mode == 1 ? "release per workgroup" : "release per wavefront", staleIters, iters, staleWords,
                   timeouts, us);
            fflush(stdout);
        }
        stop = true;
        for (auto& t : th) t.join();
    }
    return 0;
}
