// latency_probe.hip -- what one "launch a kernel, wait for its result" round trip costs on this box, and how much of it is
// the end-of-kernel signal path: (a) empty kernel + hipStreamSynchronize, (b) the kernel writes a flag into page-locked host
// memory and the host spins on it, (c) the same two with a kernel that runs ~10 us.  Tuning only.
//   hipcc -O2 --offload-arch=gfx950 -o tools/latency_probe tools/latency_probe.hip && tools/latency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include <atomic>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(int spin, volatile unsigned* flag, unsigned seq, int* sink)
{
    int acc = 0;
    for (int i = 0; i < spin; i++) acc += __builtin_amdgcn_s_memtime() & 1; // ~ 40 ns per iteration
    if (acc == -1) *sink = acc;
    if (flag && threadIdx.x == 0) {
        __threadfence_system();
        *flag = seq;
    }
}

// the same round trip for a kernel that scatters `nw` 4-byte results first: into page-locked host memory (what the matcher's
// latency path does) or into device memory
__global__ void k_scatter(int spin, volatile unsigned* flag, unsigned seq, int* sink, unsigned* out, int nw, unsigned* ctr)
{
    int acc = 0;
    for (int i = 0; i < spin; i++) acc += __builtin_amdgcn_s_memtime() & 1;
    if (acc == -1) *sink = acc;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nw) out[(t * 37) % nw * 3] = (unsigned)t; // scattered, 12 bytes apart
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ctr, 1u) + 1u == gridDim.x) {
        *ctr = 0u;
        if (flag) {
            __threadfence_system();
            *flag = seq;
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* hflag;
    CK(hipHostMalloc((void**)&hflag, 64));
    unsigned* dflag;
    CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
    int* sink;
    CK(hipMalloc((void**)&sink, 4));
    *hflag = 0;
    unsigned seq = 0;
    for (int spin : {0, 64, 256}) {
        for (int mode = 0; mode < 3; mode++) { // 0: sync, 1: flag spin (then sync outside the timed part), 2: hipEventSynchronize
            std::vector<double> t;
            hipEvent_t ev;
            CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            for (int it = 0; it < 400; it++) {
                seq++;
                const double t0 = now_us();
                hipLaunchKernelGGL(k_work, dim3(25), dim3(256), 0, st, spin, mode == 1 ? dflag : nullptr, seq, sink);
                if (mode == 0) CK(hipStreamSynchronize(st));
                else if (mode == 1) {
                    while (*(volatile unsigned*)hflag != seq) { }
                } else {
                    CK(hipEventRecord(ev, st));
                    CK(hipEventSynchronize(ev));
                }
                const double t1 = now_us();
                if (mode == 1) CK(hipStreamSynchronize(st));
                if (it >= 50) t.push_back(t1 - t0);
            }
            std::sort(t.begin(), t.end());
            printf("spin %4d  %-22s p50 %6.2f us  p99 %6.2f us\n", spin, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "flag in pinned memory" : "event synchronize",
                   t[t.size() / 2], t[t.size() * 99 / 100]);
        }
    }
    {
        unsigned *hout, *dhout, *dout, *ctr;
        CK(hipHostMalloc((void**)&hout, 1 << 20));
        CK(hipHostGetDevicePointer((void**)&dhout, hout, 0));
        CK(hipMalloc((void**)&dout, 1 << 20));
        CK(hipMalloc((void**)&ctr, 64));
        CK(hipMemset(ctr, 0, 64));
        hipEvent_t ea, eb;
        CK(hipEventCreate(&ea));
        CK(hipEventCreate(&eb));
        for (int nw : {0, 1000, 4000}) {
            for (int where = 0; where < 2; where++) {
                for (int mode = 0; mode < 2; mode++) { // 0: flag, 1: events around the kernel
                    std::vector<double> t;
                    for (int it = 0; it < 300; it++) {
                        seq++;
                        const double t0 = now_us();
                        float ms = 0.f;
                        if (mode == 1) CK(hipEventRecord(ea, st));
                        hipLaunchKernelGGL(k_scatter, dim3(100), dim3(256), 0, st, 64, mode == 0 ? dflag : nullptr, seq, sink,
                                           where ? dout : dhout, nw, ctr);
                        if (mode == 0) {
                            while (*(volatile unsigned*)hflag != seq) { }
                        } else {
                            CK(hipEventRecord(eb, st));
                            CK(hipEventSynchronize(eb));
                            CK(hipEventElapsedTime(&ms, ea, eb));
                        }
                        const double t1 = now_us();
                        CK(hipStreamSynchronize(st));
                        if (it >= 50) t.push_back(mode == 0 ? t1 - t0 : ms * 1000.0);
                    }
                    std::sort(t.begin(), t.end());
                    printf("100 x 256 threads, %4d scattered 4-byte stores to %-6s  %-28s p50 %6.2f us\n", nw, where ? "device" : "host",
                           mode == 0 ? "launch -> flag seen" : "kernel by events", t[t.size() / 2]);
                }
            }
        }
    }
    return 0;
}
