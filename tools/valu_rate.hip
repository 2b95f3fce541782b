// tools/valu_rate.hip -- issue cost of the integer VALU instructions the extractor kernels lean on,
// measured on the GPU box: cycles per wave64 instruction per SIMD, relative to v_add_u32.
// Build + run:  hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 2048
#define UNROLL 8

#define KERNEL(name, ASM)                                                              \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed)           \
    {                                                                                  \
        uint32_t a[UNROLL], b = seed + threadIdx.x, c = seed * 3 + 1;                  \
        for (int k = 0; k < UNROLL; k++) a[k] = threadIdx.x * (k + 1) + seed;          \
        for (int it = 0; it < ITERS; it++) {                                           \
            _Pragma("unroll") for (int k = 0; k < UNROLL; k++) { ASM; }                \
        }                                                                              \
        uint32_t s = 0;                                                                \
        for (int k = 0; k < UNROLL; k++) s ^= a[k];                                    \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                    \
    }

KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_mad_u64, { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(a[k]), "v"(b) : "vcc"); a[k] = (uint32_t)t; })
KERNEL(k_pk_max, asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_pk_add, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_alignbyte, asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_min3, asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_max3_i16, asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_dot4, asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_dot2, asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_bcnt, asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[k])))
KERNEL(k_sad_u8, asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc"))
KERNEL(k_cmp, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc"))
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k])))
KERNEL(k_add_u16_sdwa, asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_2" : "+v"(a[k]) : "v"(b)))
KERNEL(k_fma_f64, { double t; asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(t) : "v"((double)a[k])); a[k] = (uint32_t)(uint64_t)t; })


KERNEL(k_sub, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_or, asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k])))
KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b)))
KERNEL(k_min_i32, asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b)))
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_or3, asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_add_u16, asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_min_i16, asm volatile("v_min_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_max_u16, asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_min3_u32, asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_med3_i32, asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_min3_f32, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_fmac_f32, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_pk_sub_i16, asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_pk_mul_lo_u16, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_pk_mad_u16, asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_pk_lshr_b16, asm volatile("v_pk_lshrrev_b16 %0, 3, %0" : "+v"(a[k])))
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b)))
KERNEL(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_cvt_pk_u8, asm volatile("v_cvt_pk_u8_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_msad, asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_lerp, asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)))
KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[k]) : "v"(b) : "vcc"))
KERNEL(k_cmp_e64, asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21"))
KERNEL(k_ffbh, asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[k])))
KERNEL(k_sub_sdwa_b, asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a[k]) : "v"(b)))
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b)))
KERNEL(k_add_dpp, asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b)))
KERNEL(k_pk_add_f32, { uint64_t t = a[k]; asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(t)); a[k] = (uint32_t)t; })

template <typename K>
static double run(K kern, const char* name, uint32_t* d_out, double ref)
{
    // 256 CUs x 4 SIMDs x 4 waves per SIMD: enough independent waves to hide the dependent-issue latency
    const int blocks = 256 * 4, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(d_out, 1u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // wave-instructions per SIMD: (blocks*4 waves / 1024 SIMDs) * ITERS * UNROLL
    const double wi = (double)blocks * 4 / 1024.0 * ITERS * UNROLL;
    const double cyc = ms * 1e-3 * 2.4e9 / wi;
    printf("%-16s %8.3f ms  %6.2f cycles/wave-instr @2.4GHz  (%.2fx v_add)\n", name, ms, cyc, ref > 0 ? cyc / ref : 1.0);
    return cyc;
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 4096);
    double ref = run(k_add, "v_add_u32", d, 0);
#define R(k, n) run(k, n, d, ref)
    R(k_mul_lo, "v_mul_lo_u32");
    R(k_mul_hi, "v_mul_hi_u32");
    R(k_mul24, "v_mul_u32_u24");
    R(k_mad24, "v_mad_u32_u24");
    R(k_mad_u64, "v_mad_u64_u32");
    R(k_pk_max, "v_pk_max_i16");
    R(k_pk_add, "v_pk_add_u16");
    R(k_perm, "v_perm_b32");
    R(k_alignbyte, "v_alignbyte_b32");
    R(k_min3, "v_min3_i32");
    R(k_max3_i16, "v_max3_i16");
    R(k_dot4, "v_dot4_u32_u8");
    R(k_dot2, "v_dot2_u32_u16");
    R(k_bcnt, "v_bcnt_u32_b32");
    R(k_mbcnt, "v_mbcnt_lo");
    R(k_lshl_add, "v_lshl_add_u32");
    R(k_add3, "v_add3_u32");
    R(k_bfe, "v_bfe_u32");
    R(k_sad_u8, "v_sad_u8");
    R(k_cndmask, "v_cndmask_b32");
    R(k_cmp, "v_cmp_lt_u32");
    R(k_mul_f32, "v_mul_f32");
    R(k_cvt_f32_u32, "v_cvt_f32_u32");
    R(k_add_u16_sdwa, "v_add_u16_sdwa");
    R(k_fma_f64, "v_fma_f64");

    R(k_sub, "v_sub_u32"); R(k_and, "v_and_b32"); R(k_or, "v_or_b32"); R(k_xor, "v_xor_b32");
    R(k_lshl, "v_lshlrev_b32"); R(k_lshr, "v_lshrrev_b32"); R(k_min_i32, "v_min_i32"); R(k_max_u32, "v_max_u32");
    R(k_mov, "v_mov_b32"); R(k_and_or, "v_and_or_b32"); R(k_or3, "v_or3_b32"); R(k_add_u16, "v_add_u16");
    R(k_min_i16, "v_min_i16"); R(k_max_u16, "v_max_u16"); R(k_min3_u32, "v_min3_u32"); R(k_med3_i32, "v_med3_i32");
    R(k_min3_f32, "v_min3_f32"); R(k_min_f32, "v_min_f32"); R(k_fma_f32, "v_fma_f32"); R(k_fmac_f32, "v_fmac_f32");
    R(k_pk_sub_i16, "v_pk_sub_i16"); R(k_pk_mul_lo_u16, "v_pk_mul_lo_u16"); R(k_pk_mad_u16, "v_pk_mad_u16");
    R(k_pk_lshr_b16, "v_pk_lshrrev_b16"); R(k_lshl_or, "v_lshl_or_b32"); R(k_bitop3, "v_bitop3_b32");
    R(k_cvt_pk_u8, "v_cvt_pk_u8_f32"); R(k_msad, "v_msad_u8"); R(k_lerp, "v_lerp_u8"); R(k_add_co, "v_add_co_u32");
    R(k_cmp_e64, "v_cmp_lt_u32_e64"); R(k_ffbh, "v_ffbh_u32"); R(k_sub_sdwa_b, "v_sub_u16_sdwa(bytes)");
    R(k_mov_dpp, "v_mov_b32_dpp"); R(k_add_dpp, "v_add_u32_dpp"); R(k_pk_add_f32, "v_pk_add_f32");
    return 0;
}
