/*
 * orbfe_matcher.hip -- Hamming brute-force kernels of ORBmatcher behind the C ABI.
 *
 *   K-HAM     k_hamming_pairs   ORBmatcher::DescriptorDistance, all pairs   src/ORBmatcher.cc:2591-2607
 *   K-BFKNN2  k_bfknn2          cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)   src/Frame.cc:1137
 *   K-BOW     k_search_bow      SearchByBoW inner loops                     src/ORBmatcher.cc:297-433, 853-932
 *   K-TRI     k_search_tri      SearchForTriangulation_ inner loops         src/ORBmatcher.cc:1274-1437
 *   K-KB8     k_kb8_unproject   KannalaBrandt8::unproject                   src/CameraModels/KannalaBrandt8.cpp:96-123
 *
 * 256-bit descriptors are four 64-bit words; a distance is 4 x (xor + popcount).  Candidate
 * scans run across the 64 lanes of a wavefront and are reduced with wave shuffles on packed
 * (distance, position) keys -- integer/bitwise work, no MFMA.
 * The host keeps the reference's control flow around the loops: the merge-join over the two
 * FeatureVectors (std::map walk + lower_bound, :285-448) and the rotation-histogram cull
 * (ComputeThreeMaxima, :2545-2586) are O(N) host code.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/orbfe.h"
#include "orbfe_order.h"
#include "orbfe_pageable.h"
#include "orbfe_sincos.h"
#include "orbfe_kb8.h"

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return -(1000 + (int)_e); \
    } while (0)

#define INT_TRY(expr)            \
    do {                         \
        const int _r = (expr);   \
        if (_r < 0) return _r;   \
    } while (0)

namespace {

const int TH_LOW = 50;       // src/ORBmatcher.cc:37
const int HISTO_LENGTH = 30; // :38

struct Desc {
    unsigned long long w[4];
};

__device__ __forceinline__ Desc load_desc(const uint8_t* p)
{
    // rows are 32 bytes; cv::Mat rows (and ours) are at least 8-byte aligned
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
    Desc d;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const ulonglong2 a = q[0], b = q[1];
        d.w[0] = a.x;
        d.w[1] = a.y;
        d.w[2] = b.x;
        d.w[3] = b.y;
    } else {
        const unsigned* u = reinterpret_cast<const unsigned*>(p);
#pragma unroll
        for (int i = 0; i < 4; i++) d.w[i] = (unsigned long long)u[2 * i] | ((unsigned long long)u[2 * i + 1] << 32);
    }
    return d;
}
__device__ __forceinline__ int hamming(const Desc& a, const Desc& b)
{
    return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) +
           __popcll(a.w[3] ^ b.w[3]);
}
// Wave-wide reductions without the LDS (round 4): four DPP exchange steps inside the 16-lane rows -- partners xor 1, xor 2
// (quad permutes), 7 - i inside a group of 8 (row_half_mirror), 15 - i inside the row (row_mirror): at every step a lane
// meets a lane of a DISJOINT group that already agrees on its partial result -- then the four row results are read with
// v_readlane and merged on the scalar side.  A ds_bpermute round trip per step (what __shfl_xor compiles to) made the
// sequential row loop of K-BOW latency-bound: ~24 dependent LDS round trips per row.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_xchg(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    // (inline asm: left to itself the compiler emits a v_mov_b32_dpp and a v_min_u32 per step instead of the one
    // v_min_u32_dpp; the s_nop keeps the hazard distance between a VALU write and a DPP read of the same register)
    asm volatile("s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
// Sum over the 64 lanes (all active) with DPP and four v_readlane; the result is wave-uniform.  For counters: `atomicAdd(&word,
// perLaneValue)` on one address makes the compiler's atomic optimizer emit a scalar loop over the active lanes (~1 us per
// wavefront); one lane adding the wavefront's sum does not.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); // row_mirror: every lane holds its row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
// the two smallest keys of the wavefront (keys of different lanes are distinct, or the sentinel 0xFFFFFFFF): on return k0 <= k1
// hold them in every lane
__device__ __forceinline__ void wave_two_min(unsigned& k0, unsigned& k1)
{
    // as two plain minima (v_min_u32 with a DPP operand: four instructions each): the smallest key, then the smallest of
    // what is left when the lane that holds it puts its second key forward -- a third of the merge form's dependent chain,
    // which is what a row of K-BOW costs when its wavefront has the SIMD to itself
    const unsigned g0 = wave_min_u32(k0);
    const unsigned g1 = wave_min_u32(k0 == g0 ? k1 : k0);
    k0 = g0;
    k1 = g1;
}

#include "orbfe_matcher_knn2.hip"
#include "orbfe_matcher_bow.hip"
#include "orbfe_matcher_tri.hip"
#include "orbfe_matcher_proj.hip"
#include "orbfe_matcher_small.hip"
#include "orbfe_matcher_host.hip"
#include "orbfe_matcher_api_knn2.hip"
#include "orbfe_matcher_api_bow.hip"
#include "orbfe_matcher_api_tri.hip"
#include "orbfe_matcher_api_proj.hip"
#include "orbfe_matcher_api_vocab.hip"
} // extern "C"
#include "orbfe_matcher_bowvec.hip"
extern "C" {

int orbfe_kb8_unproject(int device, const float* P, const float* uv, int n, float* rays)
{
    if (n < 0 || !P || (n && (!uv || !rays))) return ORBFE_ERR_ARGS;
    if (n == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    float *dP, *dU, *dR;
    if ((r = s.up(&dP, P, 8)) < 0) return r;
    if ((r = s.up(&dU, uv, (size_t)n * 2)) < 0) return r;
    if ((r = s.up<float>(&dR, nullptr, (size_t)n * 3)) < 0) return r;
    {
        KernelTimer timer(s);
    hipLaunchKernelGGL(k_kb8_unproject, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_ms, dP, dU, n, dR);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(rays, dR, (size_t)n * 3 * sizeof(float)));
    INT_TRY(s.fetch());
    return 0;
}

} // extern "C"
