// FETCH_SIZE calibration on known byte counts (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports
// some access widths).  Streams one 256 MiB buffer with 1, 4 and 16 bytes per lane; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./pmc_calib
// and compare FETCH_SIZE*1024 with 268435456 for each kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T>
__global__ void k_read(const T* p, size_t n, unsigned long long* out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        for (unsigned k = 0; k < sizeof(T); k++) acc += b[k];
    }
    if (acc == 0x7fffffffffffffffull) out[0] = acc;
}
int main()
{
    const size_t bytes = 256ull << 20;
    void* d;
    unsigned long long* o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 8);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read<unsigned char>, dim3(8192), dim3(256), 0, 0, (const unsigned char*)d, bytes, o);
        hipLaunchKernelGGL(k_read<unsigned int>, dim3(8192), dim3(256), 0, 0, (const unsigned int*)d, bytes / 4, o);
        hipLaunchKernelGGL(k_read<uint4>, dim3(8192), dim3(256), 0, 0, (const uint4*)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
