// tools/gmem_align.hip -- cost of misaligned global dword loads/stores on gfx950 (L2-resident data).
// Build + run:  hipcc -O2 --offload-arch=gfx950 -w -o /tmp/gmem_align tools/gmem_align.hip && /tmp/gmem_align
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_load(const uint8_t* src, uint32_t* out, int mis, int rowPitch, int lanesPerRow)
{
    // lane i reads 4 bytes at row (i / lanesPerRow) * rowPitch + (i % lanesPerRow) * 4 + mis: a tile-like pattern
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* p = src + (size_t)blockIdx.x * 65536 + wave * 16384 + (lane / lanesPerRow) * rowPitch +
                       (lane % lanesPerRow) * 4 + mis;
    uint32_t acc = 0;
    for (int it = 0; it < 64; it++) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_memcpy(&v[k], p + k * 8 * rowPitch % 8192 + (it & 7) * 64, 4);
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_store(uint8_t* dst, int mis, int rowPitch, int lanesPerRow)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* p = dst + (size_t)blockIdx.x * 65536 + wave * 16384 + (lane / lanesPerRow) * rowPitch + (lane % lanesPerRow) * 4 + mis;
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t v = it * 8 + k;
            __builtin_memcpy(p + k * 8 * rowPitch % 8192 + (it & 7) * 64, &v, 4);
        }
    }
}

int main()
{
    uint8_t* d;
    uint32_t* o;
    const int blocks = 1024;
    hipMalloc(&d, (size_t)blocks * 65536 + 65536);
    hipMalloc(&o, 4096);
    hipMemset(d, 1, (size_t)blocks * 65536 + 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int shapes[][2] = {{256, 64}, {832, 12}, {832, 11}};
    for (auto& sh : shapes)
        for (int mis = 0; mis < 4; mis++) {
            float ms;
            k_load<<<blocks, 256>>>(d, o, mis, sh[0], sh[1]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) k_load<<<blocks, 256>>>(d, o, mis, sh[0], sh[1]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * 4 * 64 * 8; // wave-level load instructions per launch
            printf("load  pitch %4d lanes/row %2d misalign %d : %7.3f ms  %6.1f ns per 1000 wave-loads  %7.1f GB/s useful\n", sh[0],
                   sh[1], mis, ms / 5, ms / 5 * 1e6 / instr * 1000, instr * 256 / (ms / 5 * 1e-3) / 1e9);
            k_store<<<blocks, 256>>>(d, mis, sh[0], sh[1]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) k_store<<<blocks, 256>>>(d, mis, sh[0], sh[1]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("store pitch %4d lanes/row %2d misalign %d : %7.3f ms  %6.1f ns per 1000 wave-stores %7.1f GB/s useful\n", sh[0],
                   sh[1], mis, ms / 5, ms / 5 * 1e6 / instr * 1000, instr * 256 / (ms / 5 * 1e-3) / 1e9);
        }
    return 0;
}
