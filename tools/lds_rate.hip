// tools/lds_rate.hip -- cost of LDS read flavours on gfx950 (cycles per wave64 instruction per CU),
// in particular unaligned 2-byte reads vs byte reads vs aligned dword reads.
// Build + run:  hipcc -O2 --offload-arch=gfx950 -w -o /tmp/lds_rate tools/lds_rate.hip && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 1024
#define UNROLL 8

// addr pattern: lane i reads base + i * strideBytes + misalign (+ k * 256 per unrolled read)
#define KERNEL(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int strideBytes, int misalign)   \
    {                                                                                          \
        __shared__ uint8_t lds[16384];                                                         \
        for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (uint8_t)i;                    \
        __syncthreads();                                                                       \
        uint32_t addr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * strideBytes + misalign; \
        uint32_t acc = 0;                                                                      \
        for (int it = 0; it < ITERS; it++) {                                                   \
            uint32_t v[UNROLL];                                                                \
            _Pragma("unroll") for (int k = 0; k < UNROLL; k++) { ASM; }                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
            _Pragma("unroll") for (int k = 0; k < UNROLL; k++) acc ^= v[k];                    \
        }                                                                                      \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                                        \
    }

KERNEL(k_u8, asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(k * 256)))
KERNEL(k_u16, asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(k * 256)))
KERNEL(k_b32, asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(k * 256)))
KERNEL(k_u8_pair, { v[k] = 0; asm volatile("ds_read_u8_d16 %0, %1 offset:%2\n\tds_read_u8_d16_hi %0, %1 offset:%3" : "+v"(v[k]) : "v"(addr), "n"(k * 256), "n"(k * 256 + 1)); })
KERNEL(k_b64, { uint64_t t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(k * 256)); v[k] = (uint32_t)t ^ (uint32_t)(t >> 32); })
KERNEL(k_read2_b32, { uint64_t t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(addr), "n"(k * 16), "n"(k * 16 + 1)); v[k] = (uint32_t)t ^ (uint32_t)(t >> 32); })

// f64 VALU issue cost (K-DESC evaluates sin/cos in double): same harness as tools/valu_rate.hip
#define F64K(name, ASM)                                                                        \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int s, int m)                    \
    {                                                                                          \
        double a[8], b = 1.0 + 1e-9 * (threadIdx.x + s), c = 1e-12 * m;                        \
        for (int k = 0; k < 8; k++) a[k] = 1.0 + k * 1e-3;                                     \
        for (int it = 0; it < ITERS; it++) {                                                   \
            _Pragma("unroll") for (int k = 0; k < 8; k++) { ASM; }                             \
        }                                                                                      \
        double t = 0;                                                                          \
        for (int k = 0; k < 8; k++) t += a[k];                                                 \
        if (t == 0.12345) out[threadIdx.x] = 1;                                                \
    }
F64K(k_fma64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
F64K(k_mul64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
F64K(k_add64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c)))

template <typename K>
static void run(K kern, const char* name, uint32_t* d_out, int stride, int mis)
{
    const int blocks = 256 * 2, threads = 256; // 2 workgroups (8 waves) per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d_out, stride, mis);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(d_out, stride, mis);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // wave-instructions per CU: 8 waves * ITERS * UNROLL
    const double wi = 8.0 * ITERS * UNROLL;
    printf("%-22s stride %2d B  misalign %d : %7.3f ms  %6.2f cycles per wave-instr per CU\n", name, stride, mis, ms,
           ms * 1e-3 * 2.4e9 / wi);
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 4096);
    const int strides[] = {4, 5, 8};
    for (int s : strides) {
        for (int mis = 0; mis < 4; mis++) {
            run(k_u8, "ds_read_u8", d, s, mis);
            run(k_u16, "ds_read_u16", d, s, mis);
            run(k_u8_pair, "ds_read_u8_d16 + _hi", d, s, mis);
            if (mis == 0 || s == 5 || s == 4) run(k_b32, "ds_read_b32", d, s, mis);
        }
    }
    run(k_b64, "ds_read_b64", d, 8, 0);
    run(k_b64, "ds_read_b64", d, 8, 4);
    run(k_read2_b32, "ds_read2_b32", d, 4, 0);
    run(k_read2_b32, "ds_read2_b32", d, 8, 4);
    run(k_fma64, "v_fma_f64 (x4 waves)", d, 0, 0);
    run(k_mul64, "v_mul_f64 (x4 waves)", d, 0, 0);
    run(k_add64, "v_add_f64 (x4 waves)", d, 0, 0);
    printf("(f64 rows: 8 waves per CU = 2 per SIMD, so cycles per wave-instr per SIMD = value x 4)\n");
    return 0;
}
