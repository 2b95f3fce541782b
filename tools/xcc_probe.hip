// WG -> XCD mapping probe (hipcc -O2 --offload-arch=gfx950 -o tools/xcc_probe tools/xcc_probe.hip): every workgroup records the
// XCC_ID hardware register; the host checks linear id % 8 == XCC_ID.  The extractor's completion word counts per XCD on that rule
// (and checks it in every workgroup): csrc/orbfe_kernels.hip, xcd_done.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out)
{
    if (threadIdx.x == 0) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
        out[L] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;
    }
}
int main()
{
    unsigned* d;
    hipMalloc(&d, 1 << 20);
    const dim3 grids[] = {dim3(1032, 1), dim3(1032, 2), dim3(75, 1), dim3(300, 1), dim3(7, 3), dim3(4096, 1), dim3(13, 5)};
    for (const dim3& g : grids)
        for (int rep = 0; rep < 3; rep++) {
            const unsigned T = g.x * g.y;
            hipMemset(d, 0xFF, T * 4);
            hipLaunchKernelGGL(k, g, dim3(rep == 1 ? 256 : 64), 0, 0, d);
            std::vector<unsigned> h(T);
            hipMemcpy(h.data(), d, T * 4, hipMemcpyDeviceToHost);
            unsigned bad = 0, first = 0;
            for (unsigned i = 0; i < T; i++)
                if (h[i] != (i & 7u)) { if (!bad) first = i; bad++; }
            printf("grid %ux%u rep %d: %u of %u workgroups off the round-robin (first %u: xcc %u)\n", g.x, g.y, rep, bad, T, first, bad ? h[first] : 0);
        }
    return 0;
}
