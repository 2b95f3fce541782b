// Probe (round 4): what does it cost to overlap two dependent kernel chains on this box?
//  (1) two 50-us kernels of 64 workgroups each, back to back in ONE stream            -> serialised baseline
//  (2) the second launched with hipExtAnyOrderLaunch (no barrier bit)                 -> does gfx950 honour it?
//  (3) the two kernels on TWO streams, no dependency                                   -> free overlap
//  (4) chains of 4 short kernels per step on one stream vs fork/join over two streams  -> cost of cross-stream events
// build: hipcc -O2 --offload-arch=gfx950 -o overlap_probe tools/overlap_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long ticks, int* sink)
{
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    int* sink;
    hipMalloc(&sink, 4);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t e0, e1, ea, eb;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreateWithFlags(&ea, hipEventDisableTiming);
    hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    const unsigned long long T50 = 5000; // 50 us
    auto timeit = [&](const char* name, int reps, auto body) {
        for (int i = 0; i < 5; i++) body();
        hipDeviceSynchronize();
        const double t = now();
        for (int i = 0; i < reps; i++) body();
        hipDeviceSynchronize();
        printf("%-58s %8.2f us per iteration\n", name, 1e6 * (now() - t) / reps);
    };
    timeit("(1) two 50-us kernels, one stream", 200, [&] {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T50, sink);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T50, sink);
    });
    timeit("(2) second with hipExtAnyOrderLaunch, one stream", 200, [&] {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T50, sink);
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, T50, sink);
    });
    timeit("(3) two 50-us kernels, two streams", 200, [&] {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T50, sink);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, T50, sink);
    });
    const unsigned long long T10 = 1000; // 10 us
    timeit("(4a) chain of 4 x 10-us kernels, one stream", 500, [&] {
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T10, sink);
    });
    timeit("(4b) 2 + 2 kernels on two streams, fork + join events", 500, [&] {
        hipEventRecord(ea, s0);
        hipStreamWaitEvent(s1, ea, 0);
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T10, sink);
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, T10, sink);
        hipEventRecord(eb, s1);
        hipStreamWaitEvent(s0, eb, 0);
    });
    timeit("(4c) 2 + 2 kernels on two streams, join event only", 500, [&] {
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T10, sink);
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, T10, sink);
        hipEventRecord(eb, s1);
        hipStreamWaitEvent(s0, eb, 0);
    });
    timeit("(4d) 2 + 2 kernels on two streams, no events (free running)", 500, [&] {
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T10, sink);
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, T10, sink);
    });
    timeit("(4e) 4 kernels one stream, 2nd and 4th hipExtAnyOrderLaunch", 500, [&] {
        for (int k = 0; k < 4; k++) {
            if (k & 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, T10, sink);
            else hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, T10, sink);
        }
    });
    return 0;
}
