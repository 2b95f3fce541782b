// icache_probe.hip -- what a wavefront pays for instructions it executes for the first time: the same 4096 dependent
// v_add_u32 as straight-line code (16 KB) and as a loop over a 256-byte body, one wavefront, timed inside the kernel with
// s_memtime; each kernel launched several times in a row (is the instruction cache warm on the next launch?), and the
// straight-line form also after another large kernel has run in between.  Tuning only.
//   hipcc -O2 --offload-arch=gfx950 -o tools/icache_probe tools/icache_probe.hip && tools/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_straight(unsigned long long* out, unsigned* sink)
{
    unsigned x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(".rept 4096\n\tv_add_u32 %0, %0, 1\n\t.endr" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xFFFFFFFFu) *sink = x;
}
__global__ void k_straight2(unsigned long long* out, unsigned* sink) // (a second copy at another address)
{
    unsigned x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(".rept 4096\n\tv_sub_u32 %0, %0, 1\n\t.endr" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xFFFFFFFFu) *sink = x;
}
__global__ void k_loop(unsigned long long* out, unsigned* sink)
{
    unsigned x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; i++) asm volatile(".rept 64\n\tv_add_u32 %0, %0, 1\n\t.endr" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xFFFFFFFFu) *sink = x;
}

int main()
{
    unsigned long long* out;
    unsigned* sink;
    CK(hipMalloc((void**)&out, 256 * 8));
    CK(hipMalloc((void**)&sink, 4));
    unsigned long long h[256];
    auto report = [&](const char* name, int n) {
        hipDeviceSynchronize();
        hipMemcpy(h, out, n * 8, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull, mx = 0;
        for (int i = 0; i < n; i++) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
        printf("%-44s %3d wavefronts: %7.2f .. %7.2f us (s_memtime at 100 MHz)\n", name, n, mn * 0.01, mx * 0.01);
    };
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, 0, out, sink);
        report("loop 64 x 64 adds, launch", 1);
    }
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_straight, dim3(1), dim3(64), 0, 0, out, sink);
        report("straight 4096 adds (16 KB), launch", 1);
    }
    hipLaunchKernelGGL(k_straight2, dim3(1), dim3(64), 0, 0, out, sink);
    report("other straight kernel", 1);
    hipLaunchKernelGGL(k_straight, dim3(1), dim3(64), 0, 0, out, sink);
    report("straight again after the other one", 1);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_straight, dim3(100), dim3(64), 0, 0, out, sink);
        report("straight, 100 workgroups", 100);
    }
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_loop, dim3(100), dim3(64), 0, 0, out, sink);
        report("loop, 100 workgroups", 100);
    }
    return 0;
}
