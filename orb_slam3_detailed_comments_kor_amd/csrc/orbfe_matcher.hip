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
#include <vector>

#include "../../include/orbfe.h"
#include "orbfe_order.h"
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

// ------------------------------------------------------------------ K-HAM
// 64x64 tile per workgroup: the 64 A rows sit in LDS (read as wave-wide broadcasts), every lane
// keeps one B row in registers; stores are 128-B rows of u16.
__global__ __launch_bounds__(256) void k_hamming_pairs(const uint8_t* __restrict__ A, int nA,
                                                       const uint8_t* __restrict__ B, int nB,
                                                       uint16_t* __restrict__ D)
{
    __shared__ unsigned long long sA[64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    {
        const int r = tid >> 2, wd = tid & 3;
        unsigned long long v = 0;
        if (i0 + r < nA) {
            const unsigned* u = reinterpret_cast<const unsigned*>(A + (size_t)(i0 + r) * 32 + wd * 8);
            v = (unsigned long long)u[0] | ((unsigned long long)u[1] << 32);
        }
        sA[r][wd] = v;
    }
    __syncthreads();
    const int j = j0 + lane;
    Desc b = {};
    if (j < nB) b = load_desc(B + (size_t)j * 32);
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
        const int r = wave * 16 + k;
        const int i = i0 + r;
        if (i >= nA) break;
        const int d = __popcll(sA[r][0] ^ b.w[0]) + __popcll(sA[r][1] ^ b.w[1]) + __popcll(sA[r][2] ^ b.w[2]) +
                      __popcll(sA[r][3] ^ b.w[3]);
        if (j < nB) D[(size_t)i * nB + j] = (uint16_t)d;
    }
}

// ---------------------------------------------------------------- K-BFKNN2
// One wavefront per query.  key = dist<<20 | trainIdx: the two smallest keys are exactly the
// sequential strict-'<' scan's best and second best (ties -> lower train index first).
__global__ __launch_bounds__(256) void k_bfknn2(const uint8_t* __restrict__ Q, int nQ, const uint8_t* __restrict__ T,
                                                int nT, int32_t* __restrict__ idx, int32_t* __restrict__ dist)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nQ) return;
    const Desc dq = load_desc(Q + (size_t)q * 32);
    unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    for (int t = lane; t < nT; t += 64) {
        const unsigned key = ((unsigned)hamming(dq, load_desc(T + (size_t)t * 32)) << 20) | (unsigned)t;
        if (key < k0) {
            k1 = k0;
            k0 = key;
        } else if (key < k1) {
            k1 = key;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o0 = (unsigned)__shfl_xor((int)k0, off), o1 = (unsigned)__shfl_xor((int)k1, off);
        // merge two sorted pairs, keep the two smallest
        const unsigned lo = min(k0, o0);
        const unsigned hi = max(k0, o0);
        k1 = min(hi, min(k1, o1));
        k0 = lo;
    }
    if (lane == 0) {
        idx[2 * q] = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 & 0xFFFFF);
        dist[2 * q] = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 >> 20);
        idx[2 * q + 1] = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 & 0xFFFFF);
        dist[2 * q + 1] = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 >> 20);
    }
}

// knn-2 of many (query frame, train frame) pairs in one launch -- the cross-camera matching that consumes the
// all-gathered descriptors.  A job names its two frames by device pointers (descriptor rows + count).
// One LANE per query (its descriptor stays in eight registers), the train descriptors are wave-uniform and arrive
// through the scalar cache; a workgroup's SPLIT wavefronts share the same 64 queries and take every SPLIT-th train
// row each, then merge their (best, second) pairs through LDS.  Per distance and lane: 8 xor + 8 popcount-accumulate
// + 4 for the running two smallest keys -- the VALU issue rate bounds it, not memory (a train row is fetched once
// per wavefront, for 64 distances).  Keys are distance<<20 | train index, so the two smallest keys are the
// sequential scan's best and second best with ties going to the lower train index.
// fillTail: the rows between a query frame's count and `cap` get -1 | -1 from the kernel (the caller would otherwise clear both
// output arrays in front of every launch: orbfe_mc_match_ring_async, two fill commands of 0.5 MB)
template <int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void k_bfknn2_frames(const orbfe_knn2_job* __restrict__ jobs, int cap,
                                                             int32_t* __restrict__ idx, int32_t* __restrict__ dist, int fillTail)
{
    __shared__ unsigned sk0[SPLIT][64], sk1[SPLIT][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int p = blockIdx.y;
    const orbfe_knn2_job J = jobs[p];
    const int nQ = min(J.q_count[0], cap), nT = min(J.t_count[0], cap);
    const int q0 = blockIdx.x * 64;
    if (fillTail && wave == 0 && q0 + lane >= nQ && q0 + lane < cap) {
        const size_t o = ((size_t)p * cap + q0 + lane) * 2;
        idx[o] = idx[o + 1] = dist[o] = dist[o + 1] = -1;
    }
    if (q0 >= nQ) return; // uniform
    const int q = q0 + lane;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (q < nQ) {
        const uint4* qp = reinterpret_cast<const uint4*>(J.q_desc + (size_t)q * 32);
        a = qp[0];
        b = qp[1];
    }
    // (a pointer read from memory is a generic one to the compiler; as a constant-address-space pointer with a
    // wave-uniform index the train rows become s_load_dwordx8 and feed the VALU straight from scalar registers)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef const v4u __attribute__((address_space(4))) * scalar_rows;
    const scalar_rows T = (scalar_rows)(uintptr_t)J.t_desc;
    unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
#pragma unroll 2
    for (int t = wave; t < nT; t += SPLIT) {
        const v4u u = T[2 * t], v = T[2 * t + 1]; // wave-uniform address: scalar loads
        unsigned d = __popc(a.x ^ u.x);
        d += __popc(a.y ^ u.y);
        d += __popc(a.z ^ u.z);
        d += __popc(a.w ^ u.w);
        d += __popc(b.x ^ v.x);
        d += __popc(b.y ^ v.y);
        d += __popc(b.z ^ v.z);
        d += __popc(b.w ^ v.w);
        const unsigned key = (d << 20) | (unsigned)t;
        k1 = min(k1, max(k0, key));
        k0 = min(k0, key);
    }
    if (SPLIT > 1) {
        sk0[wave][lane] = k0;
        sk1[wave][lane] = k1;
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < SPLIT; w++) {
            const unsigned o0 = sk0[w][lane], o1 = sk1[w][lane];
            k1 = min(min(k1, o1), max(k0, o0));
            k0 = min(k0, o0);
        }
    }
    if (q < nQ) {
        const size_t o = ((size_t)p * cap + q) * 2;
        idx[o] = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 & 0xFFFFF);
        dist[o] = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 >> 20);
        idx[o + 1] = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 & 0xFFFFF);
        dist[o + 1] = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 >> 20);
    }
}

// The same problem on the MATRIX pipe, exactly (round 5; VERDICT r04 #4b).  64 x (1000 x 1000) Hamming distances per launch is a
// dense contraction over k = 256: with a train descriptor's bits as a_k in {0, 32} and a query's as b_k in {-64 (bit set), +64},
//     sum_k a_k b_k = -2048 (n11 - n01) = 2048 (d - popcount(q))          d = popcount(q xor t), n_xy = #{k: q_k = x, t_k = y}
// -- products of +-2048 accumulate exactly in the i32 accumulator of v_mfma_i32_32x32x32_i8 -- and one more k-block carries the
// train's index t (a = t % 64, t / 64 against b = 1, 64), so the accumulator IS the key of the sequential scan, t + 2048 (d - |q|),
// shifted by a per-query constant: smaller distance first, lower train index on ties.  Nothing is added, shifted or packed in
// the epilogue: one v_med3_i32 + one v_min_i32 per distance keep the two smallest keys per lane, and the distance and the index
// come back out of the key at the very end (d = |q| + (key >> 11), t = key & 2047; hence counts <= 2048, larger frames take
// k_bfknn2_frames).  M = 32 trains (A: the job's packed rows come into LDS once, the workgroup expands 32 of them per step into
// one of three tile buffers that its four wavefronts share), N = 2 x 32 queries per wavefront (B, expanded once into registers).
// Per 32 trains a wavefront issues 18 MFMAs (2 tiles x (8 + 1) k-blocks) = 576 cycles of the matrix pipe for 2048 distances,
// against 2048 / 64 x 24 instructions x ~3.4 cycles = 2600 cycles on the vector pipe (k_bfknn2_frames: xor, popcount, key, two
// minima).  Two query tiles per wavefront because every A fragment is a kilobyte out of LDS: with one tile per wavefront the
// four SIMDs' MFMAs would ask for exactly the LDS' 128 bytes per clock (measured: 1115 cycles per step with eight wavefronts of
// 32 queries, 965 with four of 64 and no prefetch; profiles/r05_knn2_mfma.txt).  64 jobs x 1000 queries are 1024 wavefronts of 64
// queries -- one per SIMD --, so everything is pipelined by hand inside the wavefront: the A fragments of step s + 1 are read
// while the MFMAs of step s run, the minima of step s - 1 and the expansion of tile s + 2 issue in the MFMAs' shadow.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
#define KNN2M_ROW 272 /* LDS bytes per expanded train row: 256 + 16, so that the 64 lanes' ds_read_b128 of a k-block are conflict-free */
#define KNN2M_WAVES 4
#define KNN2M_THREADS (64 * KNN2M_WAVES)
#define KNN2M_QUERIES (64 * KNN2M_WAVES) /* per workgroup */
#define KNN2M_TILE (32 * KNN2M_ROW)
// 4 bits -> 4 bytes of {0, 32}: bit j lands on bit 5 + 8 j (the partial products never share a bit position: no carries)
__device__ __forceinline__ unsigned knn2m_expand4(unsigned nib) { return (nib * 0x4081020u) & 0x20202020u; }
__device__ __forceinline__ int32_t knn2m_med3(int32_t a, int32_t b, int32_t c)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#ifdef ORBFE_KNN2_TIMING // (tools/knn_times.py: s_memtime stamps of one wavefront per workgroup, summed over the grid)
__device__ unsigned long long g_knnTimes[16];
#define KNN2M_STAMP(k)                                                                                  \
    do { /* (into LDS: a global atomic here would be waited for by the next barrier, and 256 workgroups */  \
         /* hitting one address at the same moment take ~15 000 cycles to get through) */                   \
        if (tid == 0) sStamp_[k] = __builtin_amdgcn_s_memtime() - stamp0_;                                  \
    } while (0)
#else
#define KNN2M_STAMP(k) do {} while (0)
#endif
struct Knn2mAcc {
    v16i_t a0, a1;
};
struct Knn2mFrag {
    v4i_t k[8];
};
__global__ __launch_bounds__(KNN2M_THREADS) void k_bfknn2_frames_mfma(const orbfe_knn2_job* __restrict__ jobs, int cap,
                                                                      int32_t* __restrict__ idx, int32_t* __restrict__ dist,
                                                                      int fillTail /* as k_bfknn2_frames */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t knn2m_lds[];
    uint8_t* const sT = knn2m_lds;                                                 // three expanded tiles
    unsigned* const sPk = reinterpret_cast<unsigned*>(knn2m_lds + 3 * KNN2M_TILE); // the job's packed train rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The workgroups of a job (one per 256 queries) all stream the same train rows: workgroups are dealt to the XCDs round-robin
    // in linear-id order, so job = id % 8 + 8 (id / (8 nqb)) and query block = (id / 8) % nqb put them behind ONE L2
    // (speed only: nothing depends on where a workgroup runs).  Grids whose job count is no multiple of 8 keep the plain order.
    // The LAST query block of a job is mostly air when cap is no multiple of 256 (cap 1032 for nFeatures 1000: queries 1024 ..
    // 1031, normally beyond the frame's count): those workgroups come last in dispatch order, behind the full ones, so that a
    // grid of 4 + 1 blocks x 64 jobs still starts as 256 workgroups on 256 CUs (see the LDS request at the launch).
    const int nqb = (int)gridDim.x, njob = (int)gridDim.y;
    const int mainCols = (cap % KNN2M_QUERIES != 0 && nqb > 1) ? nqb - 1 : nqb;
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    int p, qb;
    if (L >= (unsigned)(mainCols * njob)) {
        p = (int)L - mainCols * njob;
        qb = mainCols;
    } else if ((njob & 7) == 0) {
        p = (int)(L & 7u) + 8 * (int)(L / (8u * (unsigned)mainCols));
        qb = (int)((L >> 3) % (unsigned)mainCols);
    } else {
        p = (int)(L / (unsigned)mainCols);
        qb = (int)(L % (unsigned)mainCols);
    }
#ifdef ORBFE_KNN2_TIMING
    __shared__ unsigned long long sStamp_[16];
    const unsigned long long stamp0_ = __builtin_amdgcn_s_memtime();
    const unsigned long long real0_ = __builtin_amdgcn_s_memrealtime();
    if (tid < 16) sStamp_[tid] = 0ull;
#endif
    const orbfe_knn2_job J = jobs[p];
    const int qwg = qb * KNN2M_QUERIES;
    const int n = lane & 31, h = lane >> 5;
    // (Rows are only read below the frames' counts: the call's contract is cap >= every count, not cap rows behind every
    // pointer, and a read past the end of somebody's allocation can fault.  Requesting the rows together with the counts --
    // one memory round trip less in front of the first MFMA -- saved 1 us of 20 when it was first tried, and nothing (17.7-18.0
    // against 17.9-18.0 us) when orbfe_mc, whose slabs do hold cap rows per frame, asked for it through a flag late in round 5.)
    const int nQ = min(J.q_count[0], cap), nT = min(J.t_count[0], cap);
    if (fillTail) {
        const int q = qwg + tid; // (KNN2M_THREADS == KNN2M_QUERIES: one row per thread)
        if (q >= nQ && q < cap) {
            const size_t o = ((size_t)p * cap + q) * 2;
            idx[o] = idx[o + 1] = dist[o] = dist[o + 1] = -1;
        }
    }
    if (qwg >= nQ) return; // uniform over the workgroup
    uint4 qlo[2], qhi[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int q = qwg + wave * 64 + 32 * u + n;
        const uint4* qp = reinterpret_cast<const uint4*>(J.q_desc + (size_t)min(q, nQ - 1) * 32);
        qlo[u] = qp[0];
        qhi[u] = qp[1];
    }
    {
        const uint4* const src = reinterpret_cast<const uint4*>(J.t_desc);
        uint4* const dst = reinterpret_cast<uint4*>(sPk);
        for (int i = tid; i < 2 * nT; i += KNN2M_THREADS) dst[i] = src[i];
    }
#ifdef ORBFE_KNN2_TIMING
    if (tid == 0) atomicAdd(&g_knnTimes[7], 1ull);
#endif
    // ---- queries: B operands of both tiles, expanded once (k-block kb, lane half h: bits 32 kb + 16 h .. + 15 of the descriptor)
    v4i_t B[2][9];
    int pq[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const unsigned w[8] = {qlo[u].x, qlo[u].y, qlo[u].z, qlo[u].w, qhi[u].x, qhi[u].y, qhi[u].z, qhi[u].w};
        int pc = 0;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            pc += __popc(w[kb]);
            const unsigned bits = (w[kb] >> (16 * h)) & 0xFFFFu;
            v4i_t b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned e = ((((bits >> (4 * j)) & 0xFu) * 0x204081u) & 0x01010101u) << 7; // 0x80 where the bit is set
                b[j] = (int)(e | 0x40404040u);                                                    // -64 (0xC0) / +64 (0x40)
            }
            B[u][kb] = b;
        }
        pq[u] = pc;
        B[u][8] = v4i_t{h == 0 ? 0x00004001 : 0, 0, 0, 0}; // the index block: b = 1, 64 against a = t % 64, t / 64
    }
    const int32_t kMax = 0x7FFFFFFF;
    int32_t k0[2] = {kMax, kMax}, k1[2] = {kMax, kMax};
    // ---- trains: the workgroup expands 32 rows per step into LDS (thread -> row tid / 8, dword tid % 8 -> 32 bytes of {0, 32})
    const int er = tid >> 3, ed = tid & 7;
    // (no branch around the read: the row is clamped into the frame and the value dropped, so that a step stays one basic
    // block and the scheduler can put the vector work between the MFMAs)
    auto fetch = [&](int t0) -> unsigned {
        const unsigned v = sPk[min(t0 + er, max(nT - 1, 0)) * 8 + ed];
        return (t0 + er < nT) ? v : 0u;
    };
    auto expand_store = [&](unsigned packed, int buf) {
        uint4 e0, e1;
        e0.x = knn2m_expand4(packed & 0xFu);
        e0.y = knn2m_expand4((packed >> 4) & 0xFu);
        e0.z = knn2m_expand4((packed >> 8) & 0xFu);
        e0.w = knn2m_expand4((packed >> 12) & 0xFu);
        e1.x = knn2m_expand4((packed >> 16) & 0xFu);
        e1.y = knn2m_expand4((packed >> 20) & 0xFu);
        e1.z = knn2m_expand4((packed >> 24) & 0xFu);
        e1.w = knn2m_expand4(packed >> 28);
        uint4* dst = reinterpret_cast<uint4*>(&sT[buf * KNN2M_TILE + er * KNN2M_ROW + ed * 32]);
        dst[0] = e0;
        dst[1] = e1;
    };
    auto load_frags = [&](Knn2mFrag& f, int buf) { // (all eight reads in flight together)
        const uint8_t* const arow = &sT[buf * KNN2M_TILE + n * KNN2M_ROW + 16 * h];
#pragma unroll
        for (int kb = 0; kb < 8; kb++) f.k[kb] = *reinterpret_cast<const v4i_t*>(arow + 32 * kb);
    };
    auto chain = [&](const Knn2mFrag& f, int t0) -> Knn2mAcc {
        const v16i_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        Knn2mAcc r;
        r.a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.k[0], B[0][0], z, 0, 0, 0);
        r.a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.k[0], B[1][0], z, 0, 0, 0);
#pragma unroll
        for (int kb = 1; kb < 8; kb++) {
            r.a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.k[kb], B[0][kb], r.a0, 0, 0, 0);
            r.a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.k[kb], B[1][kb], r.a1, 0, 0, 0);
        }
        const int t = t0 + n; // row n of the tile
        const v4i_t ai = v4i_t{h == 0 ? ((t & 63) | ((t >> 6) << 8)) : 0, 0, 0, 0};
        r.a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, B[0][8], r.a0, 0, 0, 0);
        r.a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, B[1][8], r.a1, 0, 0, 0);
        return r;
    };
    auto fold = [&](const Knn2mAcc& r) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            k1[0] = knn2m_med3(k0[0], k1[0], r.a0[i]); // (k0 <= k1: the median of the three is the new second)
            k0[0] = min(k0[0], r.a0[i]);
            k1[1] = knn2m_med3(k0[1], k1[1], r.a1[i]);
            k0[1] = min(k0[1], r.a1[i]);
        }
    };
    const int nsteps = (nT + 31) >> 5;
#ifdef ORBFE_KNN2_TIMING
    { // (the expansion is pure arithmetic and would otherwise sink past the stamp to its first use)
        int acc_ = 0;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int kb = 0; kb < 9; kb++) acc_ ^= B[u][kb][0] ^ B[u][kb][1] ^ B[u][kb][2] ^ B[u][kb][3];
        if (acc_ == 0x1234567) atomicAdd(&g_knnTimes[6], 1ull);
    }
#endif
    KNN2M_STAMP(1); // queries expanded, packed rows requested
    __syncthreads(); // the packed rows are in place
    KNN2M_STAMP(2);
    expand_store(fetch(0), 0);
    expand_store(fetch(32), 1);
    __syncthreads();
    KNN2M_STAMP(8); // tiles 0 and 1 expanded
    if (nsteps > 0) { // (nT >= 1)
        // One step: `fin` holds tile s's fragments; tile s + 1 is complete in LDS buffer (s + 1) % 3 (barrier passed) and is read
        // into `fout`; tile s + 2 is expanded into buffer (s + 2) % 3, which nobody has read since the barrier of step s - 1.
        int b1 = 1, b2 = 2; // (s + 1) % 3, (s + 2) % 3
        auto step = [&](int sidx, const Knn2mFrag& fin, Knn2mFrag& fout, const Knn2mAcc& prev, Knn2mAcc& out, bool foldPrev) {
            const unsigned nextPacked = fetch((sidx + 2) << 5);
            load_frags(fout, b1);
            out = chain(fin, sidx << 5);
            if (foldPrev) fold(prev); // (steps before the last are full tiles)
            expand_store(nextPacked, b2);
            const int t = b1 == 2 ? 0 : b1 + 1;
            b1 = b2;
            b2 = b2 == 2 ? 0 : b2 + 1;
            (void)t;
            __syncthreads();
        };
        Knn2mFrag f0, f1;
        Knn2mAcc x0, x1;
        load_frags(f0, 0);
        step(0, f0, f1, x1, x0, false);
        KNN2M_STAMP(3);
        int sdone = 1;
        for (; sdone + 1 < nsteps; sdone += 2) {
            step(sdone, f1, f0, x0, x1, true);
            step(sdone + 1, f0, f1, x1, x0, true);
        }
        if (sdone < nsteps) { // one more: the result ends up in x1
            step(sdone, f1, f0, x0, x1, true);
            x0 = x1;
        }
        KNN2M_STAMP(4); // the loop
        // the last tile may be partial: rows beyond the frame's count must not win (register i holds row (i & 3) + 8 (i >> 2) + 4 h)
        const int tl = (nsteps - 1) << 5;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool dead = tl + (i & 3) + 8 * (i >> 2) + 4 * h >= nT;
            x0.a0[i] = dead ? kMax : x0.a0[i];
            x0.a1[i] = dead ? kMax : x0.a1[i];
        }
        fold(x0);
    }
    // ---- the two lane halves of a column hold different rows: merge, then lanes of half 0 write their query's result
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int32_t o0 = __shfl_xor(k0[u], 32), o1 = __shfl_xor(k1[u], 32);
        const int32_t b1 = min(min(k1[u], o1), max(k0[u], o0)), b0 = min(k0[u], o0);
        const int q = qwg + wave * 64 + 32 * u + n;
        if (h == 0 && q < nQ) {
            const size_t o = ((size_t)p * cap + q) * 2;
            idx[o] = b0 == kMax ? -1 : (b0 & 2047);
            dist[o] = b0 == kMax ? -1 : pq[u] + (b0 >> 11);
            idx[o + 1] = b1 == kMax ? -1 : (b1 & 2047);
            dist[o + 1] = b1 == kMax ? -1 : pq[u] + (b1 >> 11);
        }
    }
    KNN2M_STAMP(5);
#ifdef ORBFE_KNN2_TIMING
    if (tid == 0) {
        sStamp_[11] = __builtin_amdgcn_s_memrealtime() - real0_;
        for (int k = 0; k < 16; k++)
            if (k != 7 && k != 6) atomicAdd(&g_knnTimes[k], sStamp_[k]);
    }
#endif
}
#ifdef ORBFE_KNN2_TIMING
extern "C" int orbfe_debug_knn_times(unsigned long long* out8)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_knnTimes), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    static const unsigned long long zeros[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_knnTimes), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------- K-BOW
struct BowNode {
    int off1, n1, off2, n2; // ranges in the (pooled) CSR index arrays of set 1 / set 2
    int prob;               // problem this node belongs to
};
// One (set 1, set 2) matching problem of a batch; *Base are row offsets into the pooled arrays.
struct BowProb {
    int d1Base, d2Base, outBase;
    int limit1, limit2, Nleft, variant;
    float nnratio;
    int tBase; // set 2's row in the pooled "taken" flags
    // Round 4: a set may live in a keyframe handle (orbfe_keyframe_create) instead of the pooled arrays of the call: then
    // these name its resident arrays (and the node offsets of that set are relative to its own index array); null = pooled.
    const uint8_t* rDesc1; const uint8_t* rMask1; const float* rAng1; const int32_t* rInd1;
    const uint8_t* rDesc2; const uint8_t* rMask2; const float* rAng2; const int32_t* rInd2;
    // Round 5: the FeatureVectors' node ids (ascending) and offsets on the device -- a handle's own arrays or the call's pool --
    // for launches that find the shared nodes themselves (k_search_bow with nodes == nullptr: workgroup (i, p) is node i of set 1
    // of problem p); i?Base = where the set's index array starts in the pool (0 for a handle)
    const uint32_t* node1; const int32_t* offs1; int nn1, i1Base;
    const uint32_t* node2; const int32_t* offs2; int nn2, i2Base;
    // Round 6: a FeatureVector that orbfe_compute_bow left on the device (orbfe_bow_fv): the host has never seen its node count,
    // the kernel reads it from the handle's header (null: nn1 / nn2 above)
    const int32_t* dnn1; const int32_t* dnn2;
};

// ComputeThreeMaxima (:2545-2586), the device twin of three_maxima() below
__device__ __forceinline__ void three_maxima_dev(const int* histo, int L, int* out3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    int ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        ind3 = -1;
    }
    out3[0] = ind1;
    out3[1] = ind2;
    out3[2] = ind3;
}

__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    // :391-396 -- factor is 1/HISTO_LENGTH (sic)
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
    if (bin == 30) bin = 0;
    return bin;
}

// Completion word of a latency-path call (round 4).  The kernels below write their (small) results into page-locked HOST memory
// themselves; the last workgroup to finish then writes the call's sequence number into a flag word next to them, and the host
// spins on that word instead of going through hipStreamSynchronize: the end-of-kernel cache release, the completion signal and
// the runtime's wait cost ~5 us of a 12-us launch + wait round trip on this box (tools/latency_probe.hip: 12.4 -> 7.5 us).
// A wavefront that is done waits for its own result stores to be acknowledged (the mirror is fine-grained host memory:
// uncached on the device, so there is nothing to write back) and counts itself in LDS; the last wavefront of a workgroup adds
// one to a device counter; the workgroup that brings the counter to `total` resets it for the next call (calls on one stream
// are ordered) and publishes the flag behind a system-scope fence.  Only for grids of a few hundred workgroups: the counter
// is one address (a 12 000-wavefront triangulation batch with a system fence and an atomic per wavefront took 0.36 ms
// instead of 0.13).
struct DoneSig {
    unsigned* ctr;  // device memory, zero between calls
    unsigned* flag; // the kernel's address of the page-locked flag word; nullptr: no completion word (the host synchronises)
    unsigned seq, total /* workgroups */, waves /* wavefronts of the whole grid that report */;
    // Where the results go.  A kernel whose wavefronts store a 4-byte match here and a byte there STRAIGHT into the pinned mirror
    // turns every one of them into a write transaction of its own across PCIe, and the flag word queues behind all of them: a
    // SearchByBoW with 500 matches kept the host waiting 13 us after its last wavefront had ended (tools/hostbench with a
    // -DORBFE_BOW_TIMING library: wavefronts done 13 us after the first one started, call 35 us).  So the results are
    // scattered into a block of DEVICE memory that is all ones (-1) between calls, and the wavefront that completes the count
    // copies the block to the mirror as whole 16-byte rows of 64 lanes (full-line writes), puts the all-ones back, and only
    // then publishes the flag.  (Large grids: k_copy_out does the same as a kernel of its own behind the main one.)
    uint4* outDev;    // the clean block (device memory), or nullptr: the kernel's output pointers are the final destination
    uint4* outMirror; // the kernel's address of the block's pinned mirror
    unsigned out16;   // 16-byte units
};
// Result stores into the pinned mirror must have LANDED in host memory before the flag does: the flag is written by another
// wavefront, possibly on another XCD, and travels to the host by a path of its own.  A wavefront's own acknowledgements
// (s_waitcnt vmcnt(0)) only say that its stores have reached its XCD's L2 -- measured fast, and found NOT sufficient: with
// three host threads loading the link a search now and then read a row of its mirror before the row's stores had arrived
// (tests/test_gpu_keyframes.py, three threads: 4 failures in 16 runs; none in 24 with the release below).  So: every wavefront
// waits for its own acknowledgements, and ONE wavefront per workgroup -- the one that completes the workgroup's count; all of a
// workgroup's wavefronts sit on one CU, hence behind one L2 -- does a system-scope release (write-back of that L2 and a wait for
// it: buffer_wbl2 sc0 sc1, s_waitcnt) before the workgroup counts itself.  (A release per WAVEFRONT costs SearchByBoW 2 us and
// a triangulation search 12: the write-backs of one XCD queue behind each other.)
__device__ __forceinline__ void own_stores_acknowledged()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void workgroup_stores_landed() // (by one wavefront, after every wavefront's own_stores_acknowledged)
{
    __threadfence_system();
}
// the wavefront that completed the count (all 64 lanes): results to the mirror, block clean again, flag
__device__ __forceinline__ void done_publish(const DoneSig& d)
{
    const int lane = threadIdx.x & 63;
    if (d.outDev) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the other workgroups' stores (released before they counted)
        const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (unsigned i = (unsigned)lane; i < d.out16; i += 64u) {
            const uint4 v = d.outDev[i];
            d.outMirror[i] = v;
            d.outDev[i] = ones;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (this wavefront's own stores: the fence below covers them)
    }
    if (lane == 0) {
        *d.ctr = 0u; // for the next call (calls on one stream are ordered)
        __threadfence_system();
        *(volatile unsigned*)d.flag = d.seq;
    }
}
// at the top of the kernel, before any wavefront can leave (every wavefront of the workgroup executes it)
__device__ __forceinline__ void done_begin(const DoneSig& d, unsigned* wgCnt)
{
    if (!d.flag) return; // (uniform)
    if (threadIdx.x == 0) *wgCnt = 0u;
    __syncthreads();
}
// every wavefront of a four-wavefront workgroup reports (the last workgroup may hold fewer reporting wavefronts)
__device__ __forceinline__ void wave_done(const DoneSig& d, unsigned* wgCnt)
{
    if (!d.flag) return; // (wave-uniform)
    // this wavefront's result stores: visible to the device (the clean block) / acknowledged (stores into the mirror itself)
    if (d.outDev) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else own_stores_acknowledged();
    unsigned closes = 0u, last = 0u;
    if ((threadIdx.x & 63) == 0) {
        const unsigned mine = min(4u, d.waves - 4u * blockIdx.x);
        closes = atomicAdd(wgCnt, 1u) + 1u == mine ? 1u : 0u;
    }
    if (!__builtin_amdgcn_readfirstlane(closes)) return;
    if (!d.outDev) workgroup_stores_landed();
    if ((threadIdx.x & 63) == 0) last = atomicAdd(d.ctr, 1u) + 1u == d.total ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last)) done_publish(d);
}
// ... and for a kernel in which ONE wavefront per workgroup reports (d.total = workgroups)
__device__ __forceinline__ void wg1_done(const DoneSig& d)
{
    if (!d.flag) return;
    if (d.outDev) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else workgroup_stores_landed(); // (the reporting wavefront is the workgroup's only writer, or stands behind its barrier)
    unsigned last = 0u;
    if ((threadIdx.x & 63) == 0) last = atomicAdd(d.ctr, 1u) + 1u == d.total ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last)) done_publish(d);
}
// The same copy as a kernel of its own, for grids too large to count on one address: queued behind the main kernel, a few
// workgroups copy the block to the mirror, clean it and count themselves; the last one publishes the flag (d.total = gridDim.x).
__global__ __launch_bounds__(256) void k_copy_out(const DoneSig d)
{
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < d.out16; i += gridDim.x * 256u) {
        const uint4 v = d.outDev[i];
        d.outMirror[i] = v;
        d.outDev[i] = ones;
    }
    if (!d.flag) return;
    own_stores_acknowledged();
    __syncthreads();
    if (threadIdx.x == 0) {
        workgroup_stores_landed();
        if (atomicAdd(d.ctr, 1u) + 1u == d.total) {
            *d.ctr = 0u;
            __threadfence_system();
            *(volatile unsigned*)d.flag = d.seq;
        }
    }
}

// One wavefront per vocabulary node shared by both feature vectors.  Every feature belongs to
// exactly one node, so nodes are independent; inside a node the rows of set 1 stay sequential
// (a match removes its set-2 feature from later rows, :324,:884,:911) while the candidates of a
// row are spread over the lanes.  variant 0: (KeyFrame*,Frame&), match2[idx2] = idx1;
// variant 1: (KeyFrame*,KeyFrame*), match1[idx1] = idx2.  bins[] gets the rotation bin per match.
#ifdef ORBFE_BOW_TIMING // tuning only (tools/ab_build.sh bowt "-DORBFE_BOW_TIMING"): where a node's first wavefront spends its time.
// Stamps stay in registers until the wavefront is done (an atomic or a store per stamp would sit in front of the kernel's own
// s_waitcnt and be measured as part of the next stage); then one record per node: 8 x 100-MHz ticks since the wavefront began.
__device__ unsigned long long g_bowTimes[16];     // max over the nodes of every stage's duration; [4..6]: counters
#define BT_BEGIN()                                                    \
    unsigned long long btS[6] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0}; \
    int btRounds = 0
#define BT(k) btS[(k) + 1] = (unsigned long long)wall_clock64()
#define BT_END()                                                                                              \
    do {                                                                                                      \
        if ((threadIdx.x & 63) == 0) {                                                                         \
            for (int k_ = 0; k_ < 5; k_++)                                                                     \
                if (btS[k_ + 1] && btS[k_]) atomicMax(&g_bowTimes[k_ == 4 ? 7 : k_], btS[k_ + 1] - btS[k_]);  \
            atomicAdd(&g_bowTimes[5], 1ull);                                                                   \
            atomicMin(&g_bowTimes[11], btS[0]);                                                                \
            atomicMax(&g_bowTimes[12], (unsigned long long)wall_clock64());                                    \
            atomicMax(&g_bowTimes[13], btS[0]);                                                                \
            atomicAdd(&g_bowTimes[6], (unsigned long long)btRounds);                                           \
        }                                                                                                      \
    } while (0)
#else
#define BT_BEGIN() do { } while (0)
#define BT(k) do { } while (0)
#define BT_END() do { } while (0)
#endif
__device__ __forceinline__ void bow_node(const BowNode N, const BowProb* __restrict__ probs,
                                         const uint8_t* __restrict__ descPool, const uint8_t* __restrict__ maskPool,
                                         const float* __restrict__ angPool, const int32_t* __restrict__ indPool,
                                         int32_t* __restrict__ matchPool, int8_t* __restrict__ binsPool,
                                         uint8_t* __restrict__ takenPool)
{
    const int lane = threadIdx.x & 63;
    const BowProb Pb = probs[N.prob];
    // (array by array: a set whose descriptors alone are resident -- an extractor's output slab -- pools the rest)
    const uint8_t* desc1 = Pb.rDesc1 ? Pb.rDesc1 : descPool + (size_t)Pb.d1Base * 32;
    const uint8_t* desc2 = Pb.rDesc2 ? Pb.rDesc2 : descPool + (size_t)Pb.d2Base * 32;
    const uint8_t* mask1 = Pb.rMask1 ? Pb.rMask1 : maskPool + Pb.d1Base;
    const uint8_t* mask2 = Pb.rMask2 ? Pb.rMask2 : maskPool + Pb.d2Base;
    const float* ang1 = Pb.rAng1 ? Pb.rAng1 : angPool + Pb.d1Base;
    const float* ang2 = Pb.rAng2 ? Pb.rAng2 : angPool + Pb.d2Base;
    const int32_t* ind1 = Pb.rInd1 ? Pb.rInd1 : indPool; // node offsets of a pooled set are already pooled
    const int32_t* ind2 = Pb.rInd2 ? Pb.rInd2 : indPool;
    int32_t* match = matchPool + Pb.outBase;
    int8_t* bins = binsPool + Pb.outBase;
    uint8_t* taken2 = takenPool + Pb.tBase;
    const int limit1 = Pb.limit1, limit2 = Pb.limit2, Nleft = Pb.Nleft, variant = Pb.variant;
    const float nnratio = Pb.nnratio;
    // "already matched" state of this node's set-2 features: only this wave touches them, so the
    // first 4096 candidates live in one register bit per (lane, step); the rest go through taken2[].
    unsigned long long takenMask = 0ull;
    // Everything the row loop needs is fetched ONCE, up front, with all loads in flight together: lane r holds row r
    // of the node (index, eligibility, descriptor), lane c holds candidate c (index, static eligibility, descriptor).
    // The sequential row loop then runs on registers (a row's descriptor is broadcast with v_readlane) -- it used to
    // chase index -> mask -> descriptor through global memory for every row and again for every candidate of every
    // row, five dependent round trips per row.  Rows / candidates beyond the first 64 of a node take the old path.
    int rIdx = 0, cIdx = 0;
    bool rOk = false, cOk = false;
    Desc rD = {}, cD = {};
    float rAng = 0.f, cAng = 0.f; // (round 4: the angles too, so that accepting a match needs no load behind the reduction)
    if (lane < N.n1) {
        rIdx = ind1[N.off1 + lane];
        rOk = !(variant == 1 && limit1 != -1 && rIdx >= limit1) && mask1[rIdx] != 0;
        rD = load_desc(desc1 + (size_t)rIdx * 32);
        rAng = ang1[rIdx];
    }
    if (lane < N.n2) {
        cIdx = ind2[N.off2 + lane];
        cOk = variant != 1 || (!(limit2 != -1 && cIdx >= limit2) && mask2[cIdx] != 0);
        cD = load_desc(desc2 + (size_t)cIdx * 32);
        cAng = ang2[cIdx];
    }
    // ---- Round 4: nodes of at most 64 x 64 (every node of a real FeatureVector) in two phases instead of one dependent chain
    // per row.  Phase 1: LANE r scans ALL candidates for ROW r by itself (the candidates' descriptors come as wave-uniform
    // broadcasts, v_readlane; no cross-lane reduction, iterations independent of each other) and keeps the two smallest keys
    // among the left-camera candidates and the smallest among the right-camera ones -- ignoring which candidates earlier rows
    // will have taken.  Phase 2: the rows in order, as the reference walks them: a row whose remembered keys name no taken
    // candidate is decided from them (the common case: its scan would have seen exactly these); a row that lost one of its
    // keys to an earlier row is scanned again across the lanes without the taken candidates.  The sequential part shrinks from
    // ~150 dependent instructions per row to the acceptance alone.
    if (N.n1 <= 64 && N.n2 <= 64) {
        unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, r0 = 0xFFFFFFFFu;
        for (int c = 0; c < N.n2; c++) { // (uniform)
            if (!__builtin_amdgcn_readlane((int)cOk, c)) continue;
            Desc d2;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cD.w[w] & 0xFFFFFFFFull), c);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cD.w[w] >> 32), c);
                d2.w[w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
            const unsigned key = ((unsigned)hamming(rD, d2) << 20) | (unsigned)c;
            const bool right = variant == 0 && Nleft != -1 && __builtin_amdgcn_readlane(cIdx, c) >= Nleft; // (uniform)
            if (!right) {
                if (key < k0) {
                    k1 = k0;
                    k0 = key;
                } else if (key < k1)
                    k1 = key;
            } else if (key < r0)
                r0 = key;
        }
        unsigned long long takenBits = 0ull; // candidates (positions in the node's list) matched so far
        for (int r = 0; r < N.n1; r++) {
            if (!__builtin_amdgcn_readlane((int)rOk, r)) continue;
            const int idx1 = __builtin_amdgcn_readlane(rIdx, r);
            const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rAng), r));
            unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)k0, r), b1 = (unsigned)__builtin_amdgcn_readlane((int)k1, r),
                     q0 = (unsigned)__builtin_amdgcn_readlane((int)r0, r);
            auto gone = [&](unsigned k) { return k != 0xFFFFFFFFu && ((takenBits >> (k & 63u)) & 1ull) != 0ull; };
            if (gone(b0) || gone(b1) || gone(q0)) { // (uniform) an earlier row took one of them: this row's scan again, without
                Desc d1;                            // the taken candidates, across the lanes (lane c = candidate c)
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] & 0xFFFFFFFFull), r);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] >> 32), r);
                    d1.w[w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
                }
                unsigned e0 = 0xFFFFFFFFu, e1 = 0xFFFFFFFFu, f0 = 0xFFFFFFFFu, f1 = 0xFFFFFFFFu;
                if (lane < N.n2 && cOk && !((takenBits >> lane) & 1ull)) {
                    const unsigned key = ((unsigned)hamming(d1, cD) << 20) | (unsigned)lane;
                    if (variant == 0 && Nleft != -1 && cIdx >= Nleft) f0 = key;
                    else e0 = key;
                }
                wave_two_min(e0, e1);
                wave_two_min(f0, f1);
                b0 = e0;
                b1 = e1;
                q0 = f0;
            }
            const int bestDist1 = b0 == 0xFFFFFFFFu ? 256 : (int)(b0 >> 20);
            const int bestDist2 = b1 == 0xFFFFFFFFu ? 256 : (int)(b1 >> 20);
            const int bestDist1R = q0 == 0xFFFFFFFFu ? 256 : (int)(q0 >> 20);
            const bool passTh = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW); // :373 vs :906
            if (passTh) {
                if ((float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
                    const int cpos = (int)(b0 & 0xFFFFF);
                    const int idx2 = __builtin_amdgcn_readlane(cIdx, cpos);
                    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cAng), cpos));
                    takenBits |= 1ull << cpos;
                    if (lane == 0) {
                        if (variant == 0) {
                            match[idx2] = idx1;
                            bins[idx2] = (int8_t)rot_bin(a1, a2);
                        } else {
                            match[idx1] = idx2;
                            bins[idx1] = (int8_t)rot_bin(a1, a2);
                        }
                    }
                }
                if (variant == 0 && bestDist1R <= TH_LOW) { // ratio test is "|| true" in the reference (:405)
                    const int cpos = (int)(q0 & 0xFFFFF);
                    const int idx2 = __builtin_amdgcn_readlane(cIdx, cpos);
                    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cAng), cpos));
                    takenBits |= 1ull << cpos;
                    if (lane == 0) {
                        match[idx2] = idx1;
                        bins[idx2] = (int8_t)rot_bin(a1, a2);
                    }
                }
            }
        }
        return;
    }
    for (int r = 0; r < N.n1; r++) {
        int idx1;
        float a1;
        Desc d1;
        if (r < 64) { // (uniform)
            if (!__builtin_amdgcn_readlane((int)rOk, r)) continue;
            idx1 = __builtin_amdgcn_readlane(rIdx, r);
            a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rAng), r));
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] & 0xFFFFFFFFull), r);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] >> 32), r);
                d1.w[w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
        } else {
            idx1 = ind1[N.off1 + r];
            if (variant == 1 && limit1 != -1 && idx1 >= limit1) continue;
            if (!mask1[idx1]) continue;
            d1 = load_desc(desc1 + (size_t)idx1 * 32);
            a1 = ang1[idx1];
        }
        // key = dist<<20 | position in the node's list (iteration order breaks ties)
        unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu;
        for (int c = lane; c < N.n2; c += 64) {
            const int step = c >> 6;
            const int idx2 = step == 0 ? cIdx : ind2[N.off2 + c];
            bool ok = step < 64 ? !((takenMask >> step) & 1ull) : !taken2[idx2];
            if (step == 0) ok = ok && cOk;
            else if (variant == 1) ok = ok && !(limit2 != -1 && idx2 >= limit2) && mask2[idx2];
            if (!ok) continue;
            const unsigned key = ((unsigned)hamming(d1, step == 0 ? cD : load_desc(desc2 + (size_t)idx2 * 32)) << 20) | (unsigned)c;
            const bool right = (variant == 0 && Nleft != -1 && idx2 >= Nleft);
            if (!right) {
                if (key < k0) {
                    k1 = k0;
                    k0 = key;
                } else if (key < k1)
                    k1 = key;
            } else {
                if (key < r0) {
                    r1 = r0;
                    r0 = key;
                } else if (key < r1)
                    r1 = key;
            }
        }
        wave_two_min(k0, k1);
        if (variant == 0 && Nleft != -1) wave_two_min(r0, r1); // (right-camera candidates only exist for a two-camera frame)
        // acceptance (wave-uniform values; lane 0 writes)
        const int bestDist1 = k0 == 0xFFFFFFFFu ? 256 : (int)(k0 >> 20);
        const int bestDist2 = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 20);
        const int bestDist1R = r0 == 0xFFFFFFFFu ? 256 : (int)(r0 >> 20);
        const bool passTh = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW); // :373 vs :906
        if (passTh) {
            // (a winner among the node's first 64 candidates is described by registers of lane cpos)
            if ((float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
                const int cpos = (int)(k0 & 0xFFFFF);
                const int idx2 = cpos < 64 ? __builtin_amdgcn_readlane(cIdx, cpos) : ind2[N.off2 + cpos];
                const float a2 = cpos < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cAng), cpos)) : ang2[idx2];
                if ((cpos & 63) == lane && (cpos >> 6) < 64) takenMask |= 1ull << (cpos >> 6);
                if (lane == 0) {
                    if ((cpos >> 6) >= 64) taken2[idx2] = 1;
                    if (variant == 0) {
                        match[idx2] = idx1;
                        bins[idx2] = (int8_t)rot_bin(a1, a2);
                    } else {
                        match[idx1] = idx2;
                        bins[idx1] = (int8_t)rot_bin(a1, a2);
                    }
                }
            }
            if (variant == 0 && bestDist1R <= TH_LOW) { // ratio test is "|| true" in the reference (:405)
                const int cpos = (int)(r0 & 0xFFFFF);
                const int idx2 = cpos < 64 ? __builtin_amdgcn_readlane(cIdx, cpos) : ind2[N.off2 + cpos];
                const float a2 = cpos < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cAng), cpos)) : ang2[idx2];
                if ((cpos & 63) == lane && (cpos >> 6) < 64) takenMask |= 1ull << (cpos >> 6);
                if (lane == 0) {
                    if ((cpos >> 6) >= 64) taken2[idx2] = 1;
                    match[idx2] = idx1;
                    bins[idx2] = (int8_t)rot_bin(a1, a2);
                }
            }
        }
        if (N.n2 > 4096) { // rare: make lane 0's taken2 writes visible to the wave before the next row
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// exclusive prefix OR over the 64 lanes (lane 0 gets 0): wave_shr:1, then the DPP scan steps of wave_incl_scan (row shifts
// inside the 16-lane rows, row broadcasts across them)
__device__ __forceinline__ unsigned wave_excl_or_u32(unsigned x)
{
    unsigned y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true); // wave_shr:1
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x111, 0xF, 0xF, true);          // row_shr:1
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x112, 0xF, 0xF, true);          // row_shr:2
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x114, 0xF, 0xF, true);          // row_shr:4
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x118, 0xF, 0xF, true);          // row_shr:8
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x142, 0xA, 0xF, true);          // row_bcast:15 into rows 1 and 3
    y |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)y, 0x143, 0xC, 0xF, true);          // row_bcast:31 into rows 2 and 3
    return y;
}
// sorted insertion of `key` into k[0] <= k[1] <= ... (keys are distinct or the sentinel): 2 N - 1 min / max
template <int N>
__device__ __forceinline__ void sorted_insert(unsigned (&k)[N], unsigned key)
{
    unsigned t = key;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const unsigned lo = min(k[i], t);
        t = max(k[i], t);
        k[i] = lo;
    }
}

// K-BOW, one WORKGROUP per shared vocabulary node (round 4, second form).  `bow_node` above -- one wavefront per node, rows
// decided one after the other -- spends 25 us on a 39 x 38 node (tools/hostbench with a -DORBFE_BOW_TIMING library: records 3,
// prefetch 5, scan 7, row decisions 13 us): a single wavefront pays the full latency of every dependent instruction, 430 cycles
// per candidate of the scan and 790 per row.  Nodes of at most 64 x 64 (every node of a real FeatureVector) now take this path:
//  * scan: the four wavefronts of the workgroup split the CANDIDATES; in each, lane r scans the wavefront's quarter for row r
//    and keeps the FOUR smallest keys among the left-camera candidates and the two smallest among the right-camera ones
//    (sorted insertion, 7 / 3 min-max per candidate); the quarters meet in LDS and wavefront 0 merges them;
//  * decisions without the row-by-row chain.  The reference walks the rows in order and removes a matched candidate from the
//    later rows (:324, :884, :911); row r's outcome is a function d(r, T_r) of the candidates taken before it, T_r = the union
//    of the earlier rows' outcomes.  Iterate ALL rows at once (lane = row): T = exclusive prefix OR of the outcomes across the
//    lanes (DPP), new outcome = d(r, T) from the stored keys -- until nothing changes.  Row r is final after r + 1 rounds at
//    the latest (induction over the rows), so the fixed point is unique and is the sequential result; real nodes settle in two
//    to four rounds of ~60 instructions instead of n1 rows of ~100;
//  * d(r, T) needs the best and second-best NON-taken left candidate and the best non-taken right one.  Four / two stored keys
//    decide that exactly unless so many of them are taken that an unseen candidate could matter (fewer than two free left keys
//    with more candidates than keys, and neither "nothing can pass the threshold" nor "the ratio test passes against any
//    unseen candidate" settles it): such a row is scanned again across the lanes (lane c = candidate c, descriptors from LDS)
//    without the candidates in its T, and keeps the exact keys for as long as its T stays that set.  (The first version sent
//    the whole node back to `bow_node` instead: with the hostbench frames some node of nearly every call did, and the call
//    stayed at 34 us -- found with ORBFE_BOW_STOP, the run-time cut after a stage or a number of rounds.)
//  * all accepted rows store their match at once (lane = row) instead of lane 0 row by row.
// Larger nodes: wavefront 0 runs `bow_node`.
__global__ __launch_bounds__(256) void k_search_bow(const BowNode* __restrict__ nodes, int nNodes,
                                                    const BowProb* __restrict__ probs,
                                                    const uint8_t* __restrict__ descPool,
                                                    const uint8_t* __restrict__ maskPool,
                                                    const float* __restrict__ angPool,
                                                    const int32_t* __restrict__ indPool,
                                                    int32_t* __restrict__ matchPool, int8_t* __restrict__ binsPool,
                                                    uint8_t* __restrict__ takenPool, const DoneSig done, const int stopAt)
{
    __shared__ unsigned partK[3][6][64]; // wavefronts 1..3: keys of their quarter, per row
    __shared__ int sIdx[64];             // candidate position -> feature index / angle (for the stores)
    __shared__ float sAng[64];
    __shared__ __attribute__((aligned(16))) unsigned long long sDesc[64][4]; // ... -> descriptor, eligibility (rescans)
    __shared__ uint8_t sOk[64];
    const int nd = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    BT_BEGIN();
    BowNode N;
    if (nodes) {
        if (nd >= nNodes) return;
        N = nodes[nd];
    } else {
        // The merge-join of the two FeatureVectors (the loop heads of :300-318 / :851-866) done here: this workgroup is node
        // blockIdx.x of set 1 of problem blockIdx.y; its partner in set 2 is the entry with the same id, found by all lanes at
        // once (ids are unique within a vector).  Every wavefront of the workgroup does the same look-up -- two dependent round
        // trips -- and leaves together when there is no partner; such a launch carries no completion count (bow_run).
        const int pi = (int)blockIdx.y;
        const BowProb* __restrict__ Q = probs + pi;
        const int nn1 = Q->dnn1 ? *Q->dnn1 : Q->nn1, nn2 = Q->dnn2 ? *Q->dnn2 : Q->nn2;
        if (nd >= nn1) return;
        const uint32_t* __restrict__ node2 = Q->node2;
        const int32_t* __restrict__ offs1 = Q->offs1;
        const uint32_t id = Q->node1[nd];
        const int o1 = offs1[nd], e1 = offs1[nd + 1];
        int j = -1;
        for (int base = 0; base < nn2 && j < 0; base += 64) { // (uniform)
            const int k = base + lane;
            const unsigned long long hit = __ballot(k < nn2 && node2[k] == id);
            if (hit) j = base + __ffsll((long long)hit) - 1;
        }
        if (j < 0) return;
        const int32_t* __restrict__ offs2 = Q->offs2;
        const int o2 = offs2[j], e2 = offs2[j + 1];
        N.off1 = Q->i1Base + o1;
        N.n1 = e1 - o1;
        N.off2 = Q->i2Base + o2;
        N.n2 = e2 - o2;
        N.prob = pi;
        if (N.n1 <= 0 || N.n2 <= 0) return;
    }
    if (!(N.n1 <= 64 && N.n2 <= 64)) { // (uniform over the workgroup)
        if (wave == 0) {
            bow_node(N, probs, descPool, maskPool, angPool, indPool, matchPool, binsPool, takenPool);
            wg1_done(done);
        }
        return;
    }
    const BowProb Pb = probs[N.prob];
#ifdef ORBFE_BOW_TIMING
    asm volatile("" ::"s"(Pb.outBase), "s"(N.n1));
    if (wave == 0) BT(0);
#endif
    if (stopAt == 1) { // tuning (ORBFE_BOW_STOP in the environment): the kernel cut short after its n-th stage; results are wrong
        asm volatile("" ::"s"(Pb.outBase), "s"(N.n1));
        if (wave == 0) wg1_done(done);
        return;
    }
    const uint8_t* desc1 = Pb.rDesc1 ? Pb.rDesc1 : descPool + (size_t)Pb.d1Base * 32;
    const uint8_t* desc2 = Pb.rDesc2 ? Pb.rDesc2 : descPool + (size_t)Pb.d2Base * 32;
    const uint8_t* mask1 = Pb.rMask1 ? Pb.rMask1 : maskPool + Pb.d1Base;
    const uint8_t* mask2 = Pb.rMask2 ? Pb.rMask2 : maskPool + Pb.d2Base;
    const float* ang1 = Pb.rAng1 ? Pb.rAng1 : angPool + Pb.d1Base;
    const float* ang2 = Pb.rAng2 ? Pb.rAng2 : angPool + Pb.d2Base;
    const int32_t* ind1 = Pb.rInd1 ? Pb.rInd1 : indPool;
    const int32_t* ind2 = Pb.rInd2 ? Pb.rInd2 : indPool;
    const int limit1 = Pb.limit1, limit2 = Pb.limit2, Nleft = Pb.Nleft, variant = Pb.variant;
    const float nnratio = Pb.nnratio;
    // lane r: row r (every wavefront); lane j: candidate c0 + j of this wavefront's quarter
    const int per = (N.n2 + 3) >> 2, c0 = wave * per, cn = max(0, min(per, N.n2 - c0));
    int rIdx = 0, cIdx = 0;
    bool rOk = false, cOk = false;
    Desc rD = {}, cD = {};
    float rAng = 0.f, cAng = 0.f;
    if (lane < N.n1) {
        rIdx = ind1[N.off1 + lane];
        rOk = !(variant == 1 && limit1 != -1 && rIdx >= limit1) && mask1[rIdx] != 0;
        rD = load_desc(desc1 + (size_t)rIdx * 32);
        rAng = ang1[rIdx];
    }
    if (lane < cn) {
        cIdx = ind2[N.off2 + c0 + lane];
        cOk = variant != 1 || (!(limit2 != -1 && cIdx >= limit2) && mask2[cIdx] != 0);
        cD = load_desc(desc2 + (size_t)cIdx * 32);
        cAng = ang2[cIdx];
        sIdx[c0 + lane] = cIdx;
        sAng[c0 + lane] = cAng;
        sOk[c0 + lane] = cOk ? 1 : 0;
#pragma unroll
        for (int w = 0; w < 4; w++) sDesc[c0 + lane][w] = cD.w[w];
    }
#ifdef ORBFE_BOW_TIMING
    asm volatile("" ::"v"(rD.w[0]), "v"(cD.w[0]), "v"(rAng), "v"(cAng), "v"(rOk), "v"(cOk));
    if (wave == 0) BT(1);
#endif
    if (stopAt == 2) {
        asm volatile("" ::"v"(rD.w[0]), "v"(cD.w[0]), "v"(rAng), "v"(cAng), "v"(rOk), "v"(cOk), "v"(rD.w[3]), "v"(cD.w[3]));
        if (wave == 0) wg1_done(done);
        return;
    }
    const unsigned INF = 0xFFFFFFFFu;
    unsigned kL[4] = {INF, INF, INF, INF}, kR[2] = {INF, INF};
    for (int j = 0; j < cn; j++) { // (uniform)
        if (!__builtin_amdgcn_readlane((int)cOk, j)) continue;
        Desc d2;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cD.w[w] & 0xFFFFFFFFull), j);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cD.w[w] >> 32), j);
            d2.w[w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
        const unsigned key = ((unsigned)hamming(rD, d2) << 20) | (unsigned)(c0 + j);
        const bool right = variant == 0 && Nleft != -1 && __builtin_amdgcn_readlane(cIdx, j) >= Nleft; // (uniform)
        if (!right) sorted_insert(kL, key);
        else sorted_insert(kR, key);
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) partK[wave - 1][i][lane] = kL[i];
        partK[wave - 1][4][lane] = kR[0];
        partK[wave - 1][5][lane] = kR[1];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; w++) {
#pragma unroll
        for (int i = 0; i < 4; i++) sorted_insert(kL, partK[w][i][lane]);
        sorted_insert(kR, partK[w][4][lane]);
        sorted_insert(kR, partK[w][5][lane]);
    }
#ifdef ORBFE_BOW_TIMING
    asm volatile("" ::"v"(kL[0]), "v"(kL[3]), "v"(kR[0]));
    BT(2);
#endif
    if (stopAt == 3) {
        asm volatile("" ::"v"(kL[0]), "v"(kL[1]), "v"(kL[2]), "v"(kL[3]), "v"(kR[0]), "v"(kR[1]));
        wg1_done(done);
        return;
    }
    // ---- decisions: fixed point of "outcome of row r given what the earlier rows take"
    // (for the rows the stored keys cannot decide: lane c = candidate c, all candidates, from LDS)
    Desc aD = {};
    bool aOk = false, aRight = false;
    if (lane < N.n2) {
        const ulonglong2* q = reinterpret_cast<const ulonglong2*>(&sDesc[lane][0]);
        const ulonglong2 u = q[0], v = q[1];
        aD.w[0] = u.x;
        aD.w[1] = u.y;
        aD.w[2] = v.x;
        aD.w[3] = v.y;
        aOk = sOk[lane] != 0;
        aRight = variant == 0 && Nleft != -1 && sIdx[lane] >= Nleft;
    }
    const bool active = lane < N.n1 && rOk;
    unsigned long long acc = 0ull; // this row's outcome as candidate bits (at most one left and one right candidate)
    int accL = -1, accR = -1;
    // a row's exact keys for one particular set of taken candidates (valid while its T is exactly that set)
    unsigned long long cT = 0ull;
    unsigned cE0 = INF, cE1 = INF, cF0 = INF;
    bool cValid = false;
    auto passes = [&](int d) { return variant == 0 ? d <= TH_LOW : d < TH_LOW; }; // :373 vs :906
    const int roundCap = stopAt > 10 ? stopAt - 10 : N.n1 + 2; // (tuning: ORBFE_BOW_STOP=10+k ends after k rounds)
    for (int round = 0; round < roundCap; round++) { // (settles within n1 + 1 rounds; normally 2-4)
        const unsigned long long T =
            ((unsigned long long)wave_excl_or_u32((unsigned)(acc >> 32)) << 32) | (unsigned long long)wave_excl_or_u32((unsigned)acc);
        auto isFree = [&](unsigned k) { return k != INF && ((T >> (k & 63u)) & 1ull) == 0ull; };
        int nL = -1, nR = -1;
        bool un = false;
        // the exact rule on keys that are known to be the best / second-best free left and the best free right candidate
        auto decide = [&](unsigned b0, unsigned b1, unsigned q0) {
            const int d1 = b0 == INF ? 256 : (int)(b0 >> 20), d2 = b1 == INF ? 256 : (int)(b1 >> 20),
                      dR = q0 == INF ? 256 : (int)(q0 >> 20);
            nL = -1;
            nR = -1;
            if (passes(d1)) {
                if ((float)d1 < __fmul_rn(nnratio, (float)d2)) nL = (int)(b0 & 63u);
                if (variant == 0 && dR <= TH_LOW) nR = (int)(q0 & 63u); // ratio test is "|| true" in the reference (:405)
            }
        };
        if (active) {
            if (cValid && cT == T) {
                decide(cE0, cE1, cF0);
            } else {
                unsigned b0 = INF, b1 = INF;
                int nb = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool f = isFree(kL[i]);
                    b1 = (f && nb == 1) ? kL[i] : b1;
                    b0 = (f && nb == 0) ? kL[i] : b0;
                    nb += f ? 1 : 0;
                }
                const bool moreL = kL[3] != INF; // the row may have left candidates beyond the four stored (all with keys > kL[3])
                const int dLast = (int)(kL[3] >> 20);
                const unsigned q0 = isFree(kR[0]) ? kR[0] : (isFree(kR[1]) ? kR[1] : INF);
                const bool moreR = kR[1] != INF && q0 == INF;
                const int d1 = b0 == INF ? 256 : (int)(b0 >> 20);
                if (nb == 0 && moreL) {
                    un = passes(dLast); // an unseen candidate (distance >= dLast) might pass
                } else if (passes(d1)) {
                    if (nb == 1 && moreL) { // the second-best is an unseen candidate: its distance is >= dLast
                        if ((float)d1 < __fmul_rn(nnratio, (float)dLast)) nL = (int)(b0 & 63u); // passes against any of them
                        else un = true;
                    } else {
                        decide(b0, b1, q0);
                    }
                    if (variant == 0 && !un) {
                        const int dR = q0 == INF ? 256 : (int)(q0 >> 20);
                        nR = dR <= TH_LOW ? (int)(q0 & 63u) : -1;
                        if (nR < 0 && moreR && (int)(kR[1] >> 20) <= TH_LOW) un = true;
                    }
                }
            }
        }
        // rows the stored keys do not decide: their scan again, across the lanes, without the candidates taken before them
        for (unsigned long long m = __ballot(un); m; m &= m - 1ull) { // (uniform)
            const int r = (int)__builtin_ctzll(m);
            const unsigned long long Tr = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(T >> 32), r) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)T, r);
            Desc d1;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] & 0xFFFFFFFFull), r);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rD.w[w] >> 32), r);
                d1.w[w] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
            unsigned e0 = INF, e1 = INF, f0 = INF;
            if (lane < N.n2 && aOk && !((Tr >> lane) & 1ull)) {
                const unsigned key = ((unsigned)hamming(d1, aD) << 20) | (unsigned)lane;
                if (aRight) f0 = key;
                else e0 = key;
            }
            wave_two_min(e0, e1);
            f0 = wave_min_u32(f0);
            if (lane == r) {
                cE0 = e0;
                cE1 = e1;
                cF0 = f0;
                cT = T;
                cValid = true;
                decide(e0, e1, f0);
            }
        }
        const unsigned long long nacc = (nL >= 0 ? 1ull << nL : 0ull) | (nR >= 0 ? 1ull << nR : 0ull);
        const bool changed = nacc != acc || nL != accL || nR != accR;
        acc = nacc;
        accL = nL;
        accR = nR;
#ifdef ORBFE_BOW_TIMING
        btRounds++;
#endif
        if (__ballot(changed) == 0ull) break;
    }
#ifdef ORBFE_BOW_TIMING
    asm volatile("" ::"v"(acc), "v"(accL), "v"(accR));
    BT(3); // the rounds
#endif
    if (stopAt == 4) {
        asm volatile("" ::"v"(acc), "v"(accL), "v"(accR));
        wg1_done(done);
        return;
    }
    int32_t* match = matchPool + Pb.outBase;
    int8_t* bins = binsPool + Pb.outBase;
    if (accL >= 0) {
        const int idx2 = sIdx[accL];
        const int8_t bin = (int8_t)rot_bin(rAng, sAng[accL]);
        if (variant == 0) {
            match[idx2] = rIdx;
            bins[idx2] = bin;
        } else {
            match[rIdx] = idx2;
            bins[rIdx] = bin;
        }
    }
    if (accR >= 0) { // (variant 0 only)
        const int idx2 = sIdx[accR];
        match[idx2] = rIdx;
        bins[idx2] = (int8_t)rot_bin(rAng, sAng[accR]);
    }
    BT(4); // stores issued
    wg1_done(done);
    BT_END();
}

// ------------------------------------------------------------------- K-TRI
struct TriRow {
    int idx1, off2, n2;
};

// The candidates of one row (lane = candidate, 64 per round): smallest distance, then the LAST position (:1323 rejects only
// dist > bestDist).  Round 4: everything a candidate needs is loaded at once -- index first, then flag, mvuRight, descriptor,
// keypoint and octave in flight together -- instead of one load behind each `continue` of the reference's loop (six dependent
// round trips per wavefront, each ~1 us in HBM and ~2 us when the arrays are read in place from pinned memory); the two level
// tables come from lanes 0..nlevels-1 with a lane shuffle instead of a seventh dependent load.
__device__ __forceinline__ unsigned tri_scan(const Desc& d1, bool bStereo1, float la, float lb, float lc, float den, int n2,
                                             const int32_t* __restrict__ ind2row, const uint8_t* __restrict__ desc2,
                                             const uint8_t* __restrict__ hasMP2, const float* __restrict__ kp2,
                                             const int32_t* __restrict__ oct2, const float* __restrict__ uR2,
                                             const float* __restrict__ sf2, const float* __restrict__ sig2, int nlevels2, float epx,
                                             float epy, int onlyStereo, int coarse)
{
    const int lane = threadIdx.x & 63;
    const float sfL = lane < nlevels2 ? sf2[lane] : 0.f, sgL = lane < nlevels2 ? sig2[lane] : 0.f;
    unsigned best = 0xFFFFFFFFu;
    for (int c0 = 0; c0 < n2; c0 += 64) { // (uniform)
        const int c = c0 + lane;
        const bool in = c < n2;
        const int idx2 = in ? ind2row[c] : 0;
        const uint8_t mp = in ? hasMP2[idx2] : (uint8_t)1;
        const float ur = in ? uR2[idx2] : -1.f;
        const Desc d2 = in ? load_desc(desc2 + (size_t)idx2 * 32) : d1;
        const float2 k2 = in ? *reinterpret_cast<const float2*>(kp2 + 2 * (size_t)idx2) : make_float2(0.f, 0.f);
        const int o2 = in ? oct2[idx2] : 0;
        const float sfo = __shfl(sfL, o2), sgo = __shfl(sgL, o2); // (whole wavefront: before any lane drops out)
        if (mp) continue;
        const bool bStereo2 = ur >= 0;
        if (onlyStereo && !bStereo2) continue;
        const int dist = hamming(d1, d2);
        if (dist > TH_LOW) continue;
        const float k2x = k2.x, k2y = k2.y;
        if (!bStereo1 && !bStereo2) {
            const float ex = __fsub_rn(epx, k2x), ey = __fsub_rn(epy, k2y);
            if (__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)) < __fmul_rn(100.f, sfo)) continue;
        }
        bool ok = coarse != 0;
        if (!ok && den != 0.f) {
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2x), __fmul_rn(lb, k2y)), lc);
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            ok = (double)dsqr < __dmul_rn(3.84, (double)sgo);
        }
        if (!ok) continue;
        best = min(best, ((unsigned)dist << 20) | (0xFFFFFu - (unsigned)c));
    }
    return wave_min_u32(best);
}

// One wavefront per unmatched keypoint of KF1 (vbMatched2 is never set in the reference, so rows
// are independent).  A candidate passes when dist <= TH_LOW, the epipole gate (:1332-1340) and
// Pinhole::epipolarConstrain_ (Pinhole.cpp:159-181) hold (or bCoarse); the sequential scan keeps
// the smallest distance and, among equals, the LAST candidate (:1323 rejects only dist > bestDist).
__global__ __launch_bounds__(256) void k_search_tri(const TriRow* __restrict__ rows, int nRows,
                                                    const uint8_t* __restrict__ desc1, const float* __restrict__ kp1,
                                                    const float* __restrict__ uR1, const uint8_t* __restrict__ desc2,
                                                    const uint8_t* __restrict__ hasMP2, const float* __restrict__ kp2,
                                                    const int32_t* __restrict__ oct2, const float* __restrict__ uR2,
                                                    const int32_t* __restrict__ ind2, const float* __restrict__ F12,
                                                    float epx, float epy, const float* __restrict__ sf2,
                                                    const float* __restrict__ sig2, int nlevels2, int onlyStereo, int coarse,
                                                    int32_t* __restrict__ match12, const DoneSig done)
{
    __shared__ unsigned wgCnt;
    done_begin(done, &wgCnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rix = blockIdx.x * 4 + wave;
    if (rix >= nRows) return;
    const TriRow R = rows[rix];
    const int idx1 = R.idx1;
    const Desc d1 = load_desc(desc1 + (size_t)idx1 * 32);
    const float k1x = kp1[2 * idx1], k1y = kp1[2 * idx1 + 1];
    const bool bStereo1 = uR1[idx1] >= 0;
    // epipolar line l = x1' F12 (separately rounded products and sums, no FMA)
    const float la = __fadd_rn(__fadd_rn(__fmul_rn(k1x, F12[0]), __fmul_rn(k1y, F12[3])), F12[6]);
    const float lb = __fadd_rn(__fadd_rn(__fmul_rn(k1x, F12[1]), __fmul_rn(k1y, F12[4])), F12[7]);
    const float lc = __fadd_rn(__fadd_rn(__fmul_rn(k1x, F12[2]), __fmul_rn(k1y, F12[5])), F12[8]);
    const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
    const unsigned best = tri_scan(d1, bStereo1, la, lb, lc, den, R.n2, ind2 + R.off2, desc2, hasMP2, kp2, oct2, uR2, sf2, sig2, nlevels2,
                                   epx, epy, onlyStereo, coarse);
    if (lane == 0) match12[idx1] = best == 0xFFFFFFFFu ? -1 : ind2[R.off2 + (int)(0xFFFFFu - (best & 0xFFFFFu))];
    wave_done(done, &wgCnt);
}

// K-TRI for ONE current keyframe against several neighbours in one launch (round 4; LocalMapping::CreateNewMapPoints
// runs SearchForTriangulation_ of the current keyframe against 10-20 covisible keyframes, src/LocalMapping.cc:556-621):
// the same row as above with the neighbour's arrays and pair geometry taken from a per-problem record.
struct TriProb {
    const uint8_t* desc2; const uint8_t* hasMP2; const float* kp2; const int32_t* oct2; const float* uR2; const int32_t* ind2;
    const float* sf2; const float* sig2;
    float F12[9];
    float epx, epy;
    int onlyStereo, coarse;
    int outBase; // this problem's match12 row in the pooled output
    int nlevels2; // entries of sf2 / sig2
    const float* ang2; // keypoint angles of the neighbour (k_tri_compact's rotation histogram)
    int checkOri, pad;
};
struct TriRowB {
    int idx1, off2, n2, prob;
};
__global__ __launch_bounds__(256) void k_search_tri_batch(const TriRowB* __restrict__ rows, int nRows,
                                                          const TriProb* __restrict__ probs,
                                                          const uint8_t* __restrict__ desc1, const float* __restrict__ kp1,
                                                          const float* __restrict__ uR1, int32_t* __restrict__ matchPool,
                                                          const DoneSig done)
{
    __shared__ unsigned wgCnt;
    done_begin(done, &wgCnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rix = blockIdx.x * 4 + wave;
    if (rix >= nRows) return;
    const TriRowB R = rows[rix];
    const TriProb& Q = probs[R.prob];
    const int idx1 = R.idx1;
    const Desc d1 = load_desc(desc1 + (size_t)idx1 * 32);
    const float k1x = kp1[2 * idx1], k1y = kp1[2 * idx1 + 1];
    const bool bStereo1 = uR1[idx1] >= 0;
    const float la = __fadd_rn(__fadd_rn(__fmul_rn(k1x, Q.F12[0]), __fmul_rn(k1y, Q.F12[3])), Q.F12[6]);
    const float lb = __fadd_rn(__fadd_rn(__fmul_rn(k1x, Q.F12[1]), __fmul_rn(k1y, Q.F12[4])), Q.F12[7]);
    const float lc = __fadd_rn(__fadd_rn(__fmul_rn(k1x, Q.F12[2]), __fmul_rn(k1y, Q.F12[5])), Q.F12[8]);
    const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
    const uint8_t* const desc2 = Q.desc2;
    const int32_t* const ind2 = Q.ind2;
    const unsigned best = tri_scan(d1, bStereo1, la, lb, lc, den, R.n2, ind2 + R.off2, desc2, Q.hasMP2, Q.kp2, Q.oct2, Q.uR2, Q.sf2, Q.sig2,
                                   Q.nlevels2, Q.epx, Q.epy, Q.onlyStereo, Q.coarse);
    if (lane == 0)
        matchPool[Q.outBase + idx1] = best == 0xFFFFFFFFu ? -1 : ind2[R.off2 + (int)(0xFFFFFu - (best & 0xFFFFFu))];
    wave_done(done, &wgCnt);
}

// What the host used to do with the batch's match rows (:1402-1446), per neighbour on the device: the matches of row p in index
// order, the rotation histogram over them, ComputeThreeMaxima, the cull, the surviving pairs compacted in order -- so that the
// host reads ~150 pairs per neighbour instead of walking 1200 row entries of freshly written pinned memory (20 of 80 us of a
// 20-neighbour call).  One workgroup per neighbour; a thread owns a contiguous stretch of the row, so a block prefix sum over
// the threads' counts gives the ordered positions.  The row block is the arena's clean block: entries go back to -1 as they
// are read.  The last workgroup publishes the call's completion word (few workgroups: the counter is cheap here).
__global__ __launch_bounds__(256) void k_tri_compact(int32_t* __restrict__ rowsBlk, int n1, const TriProb* __restrict__ probs,
                                                     const float* __restrict__ ang1, int32_t* __restrict__ outPairs /* count x 2 n1 */,
                                                     int32_t* __restrict__ outN /* count */, const DoneSig done)
{
    __shared__ int sHist[32], sInd[3], sWave[4];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TriProb& Q = probs[p];
    int32_t* const m12 = rowsBlk + (size_t)p * n1;
    int32_t* const out = outPairs + (size_t)p * 2 * n1;
    const int per = (n1 + 255) >> 8, i0 = min(n1, tid * per), i1 = min(n1, i0 + per);
    const bool check = Q.checkOri != 0;
    // a thread's stretch of the row is read ONCE (entries and, for matches, the rotation bin) when it fits eight registers --
    // rows of up to 2048 features --; longer rows walk global memory three times (the first form: 9.9 us per launch, each pass
    // a chain of dependent loads)
    constexpr int CAP = 8;
    const bool inRegs = per <= CAP; // (uniform)
    int mReg[CAP], bReg[CAP];
    if (tid < 32) sHist[tid] = 0;
    if (inRegs) {
#pragma unroll
        for (int k = 0; k < CAP; k++) {
            const int i = i0 + k;
            mReg[k] = i < i1 ? m12[i] : -1;
        }
#pragma unroll
        for (int k = 0; k < CAP; k++) bReg[k] = (check && mReg[k] >= 0) ? rot_bin(ang1[i0 + k], Q.ang2[mReg[k]]) : 0;
    }
    __syncthreads();
    if (check) {
        if (inRegs) {
#pragma unroll
            for (int k = 0; k < CAP; k++)
                if (mReg[k] >= 0) atomicAdd(&sHist[bReg[k]], 1);
        } else {
            for (int i = i0; i < i1; i++) {
                const int m = m12[i];
                if (m >= 0) atomicAdd(&sHist[rot_bin(ang1[i], Q.ang2[m])], 1);
            }
        }
    }
    __syncthreads();
    if (tid == 0) three_maxima_dev(sHist, 30, sInd);
    __syncthreads();
    const int ind1 = sInd[0], ind2 = sInd[1], ind3 = sInd[2];
    auto binKept = [&](int b) { return !check || b == ind1 || b == ind2 || b == ind3; };
    auto keeps = [&](int i, int m) { return m >= 0 && (!check || binKept(rot_bin(ang1[i], Q.ang2[m]))); };
    int kept = 0;
    if (inRegs) {
#pragma unroll
        for (int k = 0; k < CAP; k++) kept += (mReg[k] >= 0 && binKept(bReg[k])) ? 1 : 0;
    } else {
        for (int i = i0; i < i1; i++) kept += keeps(i, m12[i]) ? 1 : 0;
    }
    // exclusive prefix of `kept` over the 256 threads
    int inc = kept;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    int before = inc - kept, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int t = sWave[w];
        if (w < wave) before += t;
        total += t;
    }
    int pos = before;
    if (inRegs) {
#pragma unroll
        for (int k = 0; k < CAP; k++) {
            if (mReg[k] < 0) continue;
            if (binKept(bReg[k])) {
                out[2 * pos] = i0 + k;
                out[2 * pos + 1] = mReg[k];
                pos++;
            }
            m12[i0 + k] = -1; // (the clean block stays clean)
        }
    } else {
        for (int i = i0; i < i1; i++) {
            const int m = m12[i];
            if (m < 0) continue;
            if (keeps(i, m)) {
                out[2 * pos] = i;
                out[2 * pos + 1] = m;
                pos++;
            }
            m12[i] = -1;
        }
    }
    if (tid == 0) outN[p] = total;
    if (!done.flag) return;
    own_stores_acknowledged();
    __syncthreads();
    if (tid == 0) {
        workgroup_stores_landed();
        if (atomicAdd(done.ctr, 1u) + 1u == done.total) {
            *done.ctr = 0u;
            __threadfence_system();
            *(volatile unsigned*)done.flag = done.seq;
        }
    }
}

// K-TRI with the KannalaBrandt8 gate (fisheye monocular pairs and two-camera rigs): same row / candidate
// structure as k_search_tri; the gate of a candidate is KannalaBrandt8::epipolarConstrain_ = a full
// triangulation (unproject x2, 4x4 Jacobi SVD, project x2) per lane.  Float-library functions (atan2f,
// tanf, cosf, sinf, hypot) are evaluated through double on the device, so gate values agree with the host to
// ~1e-6 relative and decisions can differ only within that distance of a threshold.
struct TriKb8Dev {
    const TriRow* rows;
    int nRows;
    const uint8_t *desc1, *desc2, *hasMP2;
    const float *kp1, *kp2, *uR1, *uR2;
    const int32_t *oct1, *oct2, *ind2;
    int Nleft1, Nleft2, rig;
    float P[4][8];   // 1L, 1R, 2L, 2R
    float R12[4][9]; // ll, lr, rl, rr
    float t12[4][3];
    float epx, epy;
    const float *sf2, *sig1, *sig2;
    int onlyStereo, coarse;
    int32_t* match12;
};
__global__ __launch_bounds__(256) void k_search_tri_kb8(TriKb8Dev T)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rix = blockIdx.x * 4 + wave;
    if (rix >= T.nRows) return;
    const TriRow R = T.rows[rix];
    const int idx1 = R.idx1;
    const Desc d1 = load_desc(T.desc1 + (size_t)idx1 * 32);
    const float k1x = T.kp1[2 * idx1], k1y = T.kp1[2 * idx1 + 1];
    const bool bStereo1 = !T.rig && T.uR1 && T.uR1[idx1] >= 0;
    const bool bRight1 = !(T.Nleft1 == -1 || idx1 < T.Nleft1);
    const float sigma1 = T.sig1[T.oct1[idx1]];
    unsigned best = 0xFFFFFFFFu;
    for (int c = lane; c < R.n2; c += 64) {
        const int idx2 = T.ind2[R.off2 + c];
        if (T.hasMP2[idx2]) continue;
        const bool bStereo2 = !T.rig && T.uR2 && T.uR2[idx2] >= 0;
        if (T.onlyStereo && !bStereo2) continue;
        const int dist = hamming(d1, load_desc(T.desc2 + (size_t)idx2 * 32));
        if (dist > TH_LOW) continue;
        const float k2x = T.kp2[2 * idx2], k2y = T.kp2[2 * idx2 + 1];
        const int o2 = T.oct2[idx2];
        if (!bStereo1 && !bStereo2 && !T.rig) {
            const float ex = __fsub_rn(T.epx, k2x), ey = __fsub_rn(T.epy, k2y);
            if (__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)) < __fmul_rn(100.f, T.sf2[o2])) continue;
        }
        bool ok = T.coarse != 0;
        if (!ok) {
            const bool bRight2 = !(T.Nleft2 == -1 || idx2 < T.Nleft2);
            const int sel = T.rig ? (bRight1 ? 2 : 0) + (bRight2 ? 1 : 0) : 0; // ll, lr, rl, rr (:1342-1370)
            const float* P1 = T.P[(T.rig && bRight1) ? 1 : 0];
            const float* P2 = T.P[(T.rig && bRight2) ? 3 : 2];
            ok = orbfe_kb8_triangulate_dev(P1, P2, k1x, k1y, k2x, k2y, T.R12[sel], T.t12[sel], sigma1, T.sig2[o2]) > 0.0001f;
        }
        if (!ok) continue;
        best = min(best, ((unsigned)dist << 20) | (0xFFFFFu - (unsigned)c)); // smallest dist, then last position
    }
    best = wave_min_u32(best);
    if (lane == 0) T.match12[idx1] = best == 0xFFFFFFFFu ? -1 : T.ind2[R.off2 + (int)(0xFFFFFu - (best & 0xFFFFFu))];
}
// test hook: the gate value (z1 or -1) of explicit pairs
__global__ __launch_bounds__(256) void k_kb8_triangulate(const float* __restrict__ P1, const float* __restrict__ P2,
                                                         const float* __restrict__ kp1, const float* __restrict__ kp2,
                                                         const float* __restrict__ R12, const float* __restrict__ t12,
                                                         const float* __restrict__ sigma1, const float* __restrict__ sigma2,
                                                         int n, float* __restrict__ z1, float* __restrict__ p3D)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float X[3] = {0.f, 0.f, 0.f};
    z1[i] = orbfe_kb8_triangulate_dev(P1, P2, kp1[2 * i], kp1[2 * i + 1], kp2[2 * i], kp2[2 * i + 1], R12, t12, sigma1[i],
                                      sigma2[i], X);
    if (p3D) {
        p3D[3 * i] = X[0];
        p3D[3 * i + 1] = X[1];
        p3D[3 * i + 2] = X[2];
    }
}

// The SearchForTriangulation overload that returns the triangulated points (src/ORBmatcher.cc:1452-1641): rows as in
// k_search_tri_kb8, no stereo / epipole gates, the gate is KannalaBrandt8::matchAndtriangulate with the world poses
// of the two cameras a candidate pair belongs to; the winner's point goes to points[3 * idx1].
struct Tri3dDev {
    const TriRow* rows;
    int nRows;
    const uint8_t *desc1, *desc2, *hasMP2;
    const float *kp1, *kp2;
    const int32_t *oct1, *oct2, *ind2;
    int Nleft1, Nleft2;
    float P[4][8];  // 1L, 1R, 2L, 2R
    float T[4][12]; // their poses
    const float *sig1, *sig2;
    int32_t* match12;
    float* points;
};
__global__ __launch_bounds__(256) void k_search_tri_3d(Tri3dDev T)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rix = blockIdx.x * 4 + wave;
    if (rix >= T.nRows) return;
    const TriRow R = T.rows[rix];
    const int idx1 = R.idx1;
    const Desc d1 = load_desc(T.desc1 + (size_t)idx1 * 32);
    const float k1x = T.kp1[2 * idx1], k1y = T.kp1[2 * idx1 + 1];
    const int c1 = (T.Nleft1 == -1 || idx1 < T.Nleft1) ? 0 : 1;
    const float sigma1 = T.sig1[T.oct1[idx1]];
    unsigned best = 0xFFFFFFFFu;
    float bx = 0.f, by = 0.f, bz = 0.f;
    for (int c = lane; c < R.n2; c += 64) {
        const int idx2 = T.ind2[R.off2 + c];
        if (T.hasMP2[idx2]) continue;
        const int dist = hamming(d1, load_desc(T.desc2 + (size_t)idx2 * 32));
        if (dist > TH_LOW) continue;
        const unsigned key = ((unsigned)dist << 20) | (0xFFFFFu - (unsigned)c); // smallest dist, then last position
        if (key >= best) continue;                                              // (cannot win: skip its triangulation)
        const int c2 = (T.Nleft2 == -1 || idx2 < T.Nleft2) ? 2 : 3;
        float X[3];
        if (!orbfe_kb8_match_triangulate_dev(T.P[c1], T.P[c2], k1x, k1y, T.kp2[2 * idx2], T.kp2[2 * idx2 + 1], T.T[c1], T.T[c2],
                                             sigma1, T.sig2[T.oct2[idx2]], X))
            continue;
        best = key;
        bx = X[0];
        by = X[1];
        bz = X[2];
    }
    const unsigned win = wave_min_u32(best);
    if (win == 0xFFFFFFFFu) {
        if (lane == 0) T.match12[idx1] = -1;
        return;
    }
    if (best == win) { // keys are distinct: exactly one lane
        T.match12[idx1] = T.ind2[R.off2 + (int)(0xFFFFFu - (win & 0xFFFFFu))];
        T.points[3 * (size_t)idx1] = bx;
        T.points[3 * (size_t)idx1 + 1] = by;
        T.points[3 * (size_t)idx1 + 2] = bz;
    }
}

// Frame::ComputeStereoFishEyeMatches after the knn search (src/Frame.cc:1142-1157): Lowe ratio on the two
// nearest right descriptors, then KannalaBrandt8::TriangulateMatches of the survivor with its best neighbour.
__global__ __launch_bounds__(256) void k_fisheye_stereo(const int32_t* __restrict__ knnIdx, const int32_t* __restrict__ knnDist,
                                                        int nL, int nR, const float* __restrict__ kpL,
                                                        const float* __restrict__ kpR, const int32_t* __restrict__ octL,
                                                        const int32_t* __restrict__ octR, const float* __restrict__ P1,
                                                        const float* __restrict__ P2, const float* __restrict__ Rlr,
                                                        const float* __restrict__ tlr, const float* __restrict__ sigma2,
                                                        int32_t* __restrict__ l2r, float* __restrict__ depth,
                                                        float* __restrict__ p3D)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nL) return;
    int match = -1;
    float d = -1.f, X[3] = {0.f, 0.f, 0.f};
    if (nR >= 2 && (double)(float)knnDist[2 * q] < __dmul_rn((double)(float)knnDist[2 * q + 1], 0.7)) {
        const int t = knnIdx[2 * q];
        const float z = orbfe_kb8_triangulate_dev(P1, P2, kpL[2 * q], kpL[2 * q + 1], kpR[2 * t], kpR[2 * t + 1], Rlr, tlr,
                                                  sigma2[octL[q]], sigma2[octR[t]], X);
        if (z > 0.0001f) {
            match = t;
            d = z;
        } else {
            X[0] = X[1] = X[2] = 0.f;
        }
    }
    l2r[q] = match;
    depth[q] = d;
    p3D[3 * q] = X[0];
    p3D[3 * q + 1] = X[1];
    p3D[3 * q + 2] = X[2];
}

// ------------------------------------------------------------------ K-PROJ
// Inner loops of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:44-197, :2193-2419, :2421-2541): window
// query in the frame grid (Frame::GetFeaturesInArea, src/Frame.cc:643-708) + best / second-best Hamming
// distance + the reference's sequential occupancy rule (a feature that an earlier map point took is skipped
// by later ones, :83-85).  Three launches:
//   k_proj_grid        Frame::AssignFeaturesToGrid (src/Frame.cc:380-410) as a CSR, one workgroup;
//   k_proj_candidates  one wavefront per query: every candidate that passes the static tests (window, level,
//                      mvuRight gate, occupied on entry) gets a key  distance | visit order | feature, and the
//                      keys of a query are stored sorted -- the order in which the reference's `dist<bestDist`
//                      / `dist<bestDist2` chain ranks them;
//   k_proj_sweeps      the sequential rule as a fixpoint: query q sees feature f as taken when the least-index
//                      blocking writer of f in the previous sweep is < q; its best / second best are the
//                      first two untaken keys.  The result of q depends only on queries < q, so the unique
//                      fixpoint is the sequential result and sweep k fixes at least queries 0..k (2-5 sweeps
//                      in practice).  One workgroup, because a sweep ends in a grid-wide barrier.
struct ProjDev {
    const uint8_t* desc;
    const float *kx, *ky;
    const int32_t* octave;
    const float* uright;
    const uint8_t* taken;
    const int32_t *l2r, *r2l;
    int n, Nleft;
    float minX, minY, wInv, hInv;
    int nq;
    const uint8_t* qdesc;
    const float *qx, *qy, *qr, *qxr;
    const int32_t *qmin, *qmax;
    const uint8_t *qflags, *qblocks;
    int mode;
    float nnratio;
    int thHigh;
    const float* invSigma2; // per level, chi2 gate
    int chi2;
    int32_t* cellStart; // 2 * 3072 + 1
    int32_t* cellItems; // n
    int32_t* cellOf;    // n
    unsigned long long *rawKeys, *sortedKeys;
    int keyCap;
    int32_t *qStart, *qCount; // nq
    int32_t* qArea;     // nq or NULL: 1 = GetFeaturesInArea returned something (read by queries with flag bit 2)
    int32_t* minW;      // 2 * n
    int32_t* state;     // 2 * 3 * nq: choice, partner, rejected
    int32_t* qMatch;    // nq
    int32_t* featMatch; // n
    int32_t* status;    // nmatches, sweeps, keys needed
    int sweepLds;       // k_proj_sweeps keeps minW and state in its dynamic LDS
    // latency path (one search against a resident frame): k_proj_sweeps -- one workgroup -- copies status | qMatch | featMatch
    // (contiguous) into the pinned mirror and publishes the call's completion word (DoneSig; no counter: one workgroup)
    int32_t* mirror;
    int mirrorInts;
    unsigned* doneFlag;
    unsigned doneSeq;
    // Round 5: the one state the fixpoint of k_proj_sweeps does not represent -- map points with Observations() == 0 among
    // the queries TOGETHER with stereo-partner writes (src/ORBmatcher.cc:83-85, :117-121: the partner entry is overwritten
    // without looking at its occupant, so a non-blocking point can FREE a feature an earlier point had taken) -- walks the
    // queries in order instead (proj_inorder_body).  `taken` is then null for the candidates kernel (a feature that is
    // occupied on entry may become free) and the entry state travels in taken0.
    int inorder;
    const uint8_t* taken0;
    int resident; // the frame side and its grid come from an orbfe_frame handle: k_proj_grid_batch has nothing to build
};
// Every query owns PROJ_QUOTA key slots (its stretch starts at PROJ_QUOTA * q); a query with more candidates takes a stretch of
// the overflow region behind them, handed out by an atomic on status[2].  (Handing out EVERY stretch that way -- 300 wavefronts
// adding to one word and waiting for the old value -- cost each of them 4.5 of its 9.6 us, tools/hostbench with a
// -DORBFE_PROJ_TIMING library.)  keyCap counts both regions; the host adds PROJ_QUOTA * nq to status[2] when it sizes a retry.
#define PROJ_QUOTA 32
#define PROJ_GC 64
#define PROJ_GR 48
#define PROJ_CELLS (PROJ_GC * PROJ_GR)
#define PROJ_THREADS 1024
// key = dist(9) << 55 | cell sequence(12) << 43 | position in cell(19) << 24 | feature(24, only 19 used)
#define PROJ_MAXN (1 << 19)

__device__ __forceinline__ void proj_grid_body(const ProjDev& P)
{
    if (P.resident) { // (uniform; a batch that mixes resident and staged frame sides)
        if (threadIdx.x == 0) P.status[2] = 0;
        return;
    }
    // Round 4: the cell of a thread's first features stays in a register between the counting and the filling pass, and a
    // frame of up to PROJ_ITEMS_LDS features builds and orders its cell lists in LDS (one coalesced write at the end) -- the
    // first form filled and insertion-sorted them in global memory, behind its own stores: 11.6 us for one workgroup, most of
    // orbfe_frame_create.
    constexpr int PROJ_ITEMS_LDS = 4096, KEEP = 4;
    __shared__ int sCnt[2 * PROJ_CELLS];
    __shared__ int sItems[PROJ_ITEMS_LDS];
    __shared__ int sWave[PROJ_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n;
    const bool inLds = n <= PROJ_ITEMS_LDS; // (uniform)
    for (int c = tid; c < 2 * PROJ_CELLS; c += PROJ_THREADS) sCnt[c] = 0;
    if (tid == 0) P.status[2] = 0;
    __syncthreads();
    int cellReg[KEEP] = {-1, -1, -1, -1};
    auto cell_of = [&](int i) {
        const float fx = roundf(__fmul_rn(__fsub_rn(P.kx[i], P.minX), P.wInv));
        const float fy = roundf(__fmul_rn(__fsub_rn(P.ky[i], P.minY), P.hInv));
        int c = -1;
        if (fx >= 0.f && fx < (float)PROJ_GC && fy >= 0.f && fy < (float)PROJ_GR)
            c = (int)fx * PROJ_GR + (int)fy + ((P.Nleft != -1 && i >= P.Nleft) ? PROJ_CELLS : 0);
        return c;
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const int i = tid + k * PROJ_THREADS;
        if (i < n) {
            const int c = cell_of(i);
            cellReg[k] = c;
            if (c >= 0) atomicAdd(&sCnt[c], 1);
            P.cellOf[i] = c;
        }
    }
    for (int i = tid + KEEP * PROJ_THREADS; i < n; i += PROJ_THREADS) {
        const int c = cell_of(i);
        if (c >= 0) atomicAdd(&sCnt[c], 1);
        P.cellOf[i] = c;
    }
    __syncthreads();
    {
        const int per = 2 * PROJ_CELLS / PROJ_THREADS; // 6
        int loc[per], sum = 0;
#pragma unroll
        for (int k = 0; k < per; k++) {
            loc[k] = sum;
            sum += sCnt[tid * per + k];
        }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) sWave[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; w++) wbase += sWave[w];
        const int excl = wbase + inc - sum;
#pragma unroll
        for (int k = 0; k < per; k++) {
            const int st = excl + loc[k];
            P.cellStart[tid * per + k] = st;
            sCnt[tid * per + k] = st; // becomes the fill cursor
        }
        if (tid == PROJ_THREADS - 1) P.cellStart[2 * PROJ_CELLS] = excl + sum;
    }
    __syncthreads();
    auto put = [&](int i, int c) {
        if (c < 0) return;
        const int at = atomicAdd(&sCnt[c], 1), v = (P.Nleft != -1 && i >= P.Nleft) ? i - P.Nleft : i;
        if (inLds) sItems[at] = v;
        else P.cellItems[at] = v;
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const int i = tid + k * PROJ_THREADS;
        if (i < n) put(i, cellReg[k]);
    }
    for (int i = tid + KEEP * PROJ_THREADS; i < n; i += PROJ_THREADS) put(i, P.cellOf[i]);
    __syncthreads();
    // push_back order = ascending feature index.  A cell's list is [end of the cell before, its own fill cursor): the cursors
    // of consecutive cells meet
    for (int c = tid; c < 2 * PROJ_CELLS; c += PROJ_THREADS) {
        const int st = c ? sCnt[c - 1] : 0, en = sCnt[c];
        if (inLds) {
            for (int a = st + 1; a < en; a++) {
                const int v = sItems[a];
                int b = a - 1;
                while (b >= st && sItems[b] > v) {
                    sItems[b + 1] = sItems[b];
                    b--;
                }
                sItems[b + 1] = v;
            }
        } else {
            for (int a = st + 1; a < en; a++) {
                const int v = P.cellItems[a];
                int b = a - 1;
                while (b >= st && P.cellItems[b] > v) {
                    P.cellItems[b + 1] = P.cellItems[b];
                    b--;
                }
                P.cellItems[b + 1] = v;
            }
        }
    }
    if (inLds) {
        __syncthreads();
        for (int i = tid; i < n; i += PROJ_THREADS) P.cellItems[i] = sItems[i]; // (entries past the in-grid features are never read)
    }
}

// static tests of one candidate (everything except "taken by an earlier query"); g = feature index into the
// frame arrays, local = its index inside its camera's list (what the grid cells hold)
// Returns 0 = not in the area (GetFeaturesInArea would not return it), 1 = in the area but rejected by the loop over
// vIndices, 2 = a candidate.
__device__ __forceinline__ int proj_static_ok(const ProjDev& P, int g, int local, float x, float y, float r,
                                              int minLevel, int maxLevel, bool gate, float xr)
{
    const int oct = P.octave[g];
    if (oct < minLevel || (maxLevel >= 0 && oct > maxLevel)) return 0;
    const float kpx = P.kx[g], kpy = P.ky[g];
    if (!(fabsf(__fsub_rn(kpx, x)) < r && fabsf(__fsub_rn(kpy, y)) < r)) return 0;
    if (P.taken && P.taken[g]) return 1;
    if (P.chi2) {
        // Fuse (src/ORBmatcher.cc:1773-1799): mvuRight is read with the camera-local index (before :1801)
        const float ex = __fsub_rn(x, kpx), ey = __fsub_rn(y, kpy);
        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
        const float kpr = P.uright ? P.uright[local] : -1.f;
        double lim = 5.99;
        if (kpr >= 0.f) {
            const float er = __fsub_rn(xr, kpr);
            e2 = __fadd_rn(e2, __fmul_rn(er, er));
            lim = 7.8;
        }
        if ((double)__fmul_rn(e2, P.invSigma2[oct]) > lim) return 1;
    } else if (gate) {
        const float ur = P.uright[g];
        if (ur > 0.f && fabsf(__fsub_rn(xr, ur)) > r) return 1;
    }
    return 2;
}

// The window of a query is a run of grid columns, and inside a column the cells cy0..cy1 are neighbours in the
// CSR: the candidates of one column are ONE contiguous stretch of cellItems, and the reference's visit order
// (columns, rows, push_back order: Frame::GetFeaturesInArea :682-703) is the order of the concatenated stretches.
// Lanes first fetch the <= 64 stretch bounds, a prefix sum turns them into one flat candidate range, and every
// lane then handles candidates flat = lane, lane + 64, ...: all loads of a round are independent, and the
// dependent chain is bounds -> item -> keypoint -> descriptor whatever the window holds.  Keys are collected and
// rank-sorted in LDS (PROJ_KCAP per query); a window with more candidates takes the global two-pass path.
#define PROJ_KCAP 192
#ifdef ORBFE_PROJ_TIMING
__device__ unsigned long long g_projTimes[16]; // [1..8] the sweeps workgroup, [10..13] sums over [14] candidate wavefronts (100-MHz ticks)
#endif
__device__ __forceinline__ void proj_candidates_body(const ProjDev& P)
{
    __shared__ int sLo[4][64], sBase[4][65];
    __shared__ unsigned long long sKeys[4][PROJ_KCAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= P.nq) return;
#ifdef ORBFE_PROJ_TIMING
    unsigned long long ctS[6] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0};
#define CT(k) ctS[k] = (unsigned long long)wall_clock64()
#else
#define CT(k) do { } while (0)
#endif
    const int flags = P.qflags ? P.qflags[q] : 0;
    const bool bRight = flags & 1;
    const float x = P.qx[q], y = P.qy[q], r = P.qr[q];
    // (the query's descriptor and level range with its other fields -- on the latency path they all sit in pinned host memory,
    // a PCIe round trip each when they are asked for one after the other)
    const Desc dq = load_desc(P.qdesc + (size_t)q * 32);
    const int minLevel = P.qmin[q], maxLevel = P.qmax[q];
    const float fx0 = floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, P.minX), r), P.wInv));
    const float fx1 = ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, P.minX), r), P.wInv));
    const float fy0 = floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, P.minY), r), P.hInv));
    const float fy1 = ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, P.minY), r), P.hInv));
    int m = 0, base = 0;
    bool inArea = false; // (per lane) some feature of the window passed GetFeaturesInArea's own tests
    if (fx0 < (float)PROJ_GC && fx1 >= 0.f && fy0 < (float)PROJ_GR && fy1 >= 0.f) {
        const int cx0 = fx0 > 0.f ? (int)fx0 : 0, cx1 = fx1 < (float)(PROJ_GC - 1) ? (int)fx1 : PROJ_GC - 1;
        const int cy0 = fy0 > 0.f ? (int)fy0 : 0, cy1 = fy1 < (float)(PROJ_GR - 1) ? (int)fy1 : PROJ_GR - 1;
        const int ncy = cy1 - cy0 + 1, ncols = cx1 - cx0 + 1; // ncols <= PROJ_GC = 64
        const int fbase = bRight ? P.Nleft : 0, side = bRight ? PROJ_CELLS : 0;
        const bool gate = !bRight && P.Nleft == -1 && P.uright != nullptr;
        const float xr = (gate || (P.chi2 && P.qxr)) ? P.qxr[q] : 0.f;
        int lo = 0, cnt = 0;
        if (lane < ncols) {
            const int c0 = side + (cx0 + lane) * PROJ_GR + cy0;
            lo = P.cellStart[c0];
            cnt = P.cellStart[c0 + ncy] - lo;
        }
        int inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        const int T = __shfl(inc, 63);
        CT(1); // query fields + cell ranges
        sLo[wave][lane] = lo;
        sBase[wave][lane] = inc - cnt;
        if (lane == 0) sBase[wave][64] = T; // (entries >= ncols hold T as well: cnt = 0 there)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // pass over the flat range; keep = 0: count and collect in LDS; keep = 1 (only when the window overflowed
        // the LDS buffer): write to the reserved stretch of rawKeys
        auto enumerate = [&](bool toGlobal) -> int {
            int mm = 0;
            for (int i0 = 0; i0 < T; i0 += 64) {
                const int i = i0 + lane;
                bool ok = false;
                unsigned long long key = 0ull;
                if (i < T) {
                    int col = 0; // last column whose base <= i and which is not empty
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1)
                        if (col + step < ncols && sBase[wave][col + step] <= i) col += step;
                    const int local = P.cellItems[sLo[wave][col] + (i - sBase[wave][col])];
                    const int g = local + fbase;
                    const int verdict = proj_static_ok(P, g, local, x, y, r, minLevel, maxLevel, gate, xr);
                    inArea = inArea || verdict != 0;
                    ok = verdict == 2;
                    if (ok) {
                        const int dist = hamming(dq, load_desc(P.desc + (size_t)g * 32));
                        key = ((unsigned long long)dist << 55) | ((unsigned long long)i << 24) | (unsigned long long)g;
                    }
                }
                const unsigned long long mask = __ballot(ok);
                if (ok) {
                    const int pos = mm + __popcll(mask & ((1ull << lane) - 1ull));
                    if (toGlobal) P.rawKeys[base + pos] = key;
                    else if (pos < PROJ_KCAP) sKeys[wave][pos] = key;
                }
                mm += __popcll(mask);
            }
            return mm;
        };
        m = enumerate(false);
        CT(2); // candidates enumerated and scored
        if (m > 0) {
            if (m <= PROJ_QUOTA) {
                base = PROJ_QUOTA * q;
            } else {
                if (lane == 0) base = PROJ_QUOTA * P.nq + atomicAdd(&P.status[2], m);
                base = __shfl(base, 0);
            }
            CT(3); // key range reserved
            if (base + m > P.keyCap) {
                m = -1; // the host enlarges the key buffers and runs again
            } else if (m <= PROJ_KCAP) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int e = lane; e < m; e += 64) { // rank sort in LDS (keys are distinct)
                    const unsigned long long key = sKeys[wave][e];
                    int rank = 0;
                    for (int k = 0; k < m; k++) rank += sKeys[wave][k] < key;
                    P.sortedKeys[base + rank] = key;
                }
            } else {
                enumerate(true);
                __threadfence();
                for (int e = lane; e < m; e += 64) {
                    const unsigned long long key = P.rawKeys[base + e];
                    int rank = 0;
                    for (int k = 0; k < m; k++) rank += P.rawKeys[base + k] < key;
                    P.sortedKeys[base + rank] = key;
                }
            }
        }
    }
    const bool anyInArea = __ballot(inArea) != 0ull;
    if (lane == 0) {
        P.qStart[q] = base;
        P.qCount[q] = m;
        if (P.qArea) P.qArea[q] = anyInArea ? 1 : 0;
    }
    CT(4); // keys sorted and written
#ifdef ORBFE_PROJ_TIMING
    if (lane == 0 && ctS[1] && ctS[2] && ctS[4]) { // (sums over the wavefronts that went through every stage)
        for (int k = 1; k <= 4; k++) atomicAdd(&g_projTimes[9 + k], (ctS[k] ? ctS[k] : ctS[k - 1]) - ctS[0]);
        atomicAdd(&g_projTimes[14], 1ull);
    }
#endif
#undef CT
}

#ifdef ORBFE_PROJ_TIMING // tuning only (tools/ab_build.sh projt "-DORBFE_PROJ_TIMING"): stage times of K-PROJ's sweeps workgroup // 100-MHz ticks since the workgroup began: init, cache, sweeps, final, mirror; [8] = sweeps
#define PT_BEGIN() unsigned long long ptS[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0}
#define PT(k) ptS[k] = (unsigned long long)wall_clock64()
#define PT_END(nsweeps)                                                           \
    do {                                                                          \
        if (threadIdx.x == 0) {                                                   \
            for (int k_ = 1; k_ < 8; k_++) g_projTimes[k_] = ptS[k_] ? ptS[k_] - ptS[0] : 0ull; \
            g_projTimes[8] = (unsigned long long)(nsweeps);                       \
        }                                                                         \
    } while (0)
#else
#define PT_BEGIN() do { } while (0)
#define PT(k) do { } while (0)
#define PT_END(n) do { } while (0)
#endif
// The sequential walk itself (ProjDev::inorder): ONE wavefront takes the queries in the reference's order over the keys
// k_proj_candidates left sorted by (distance, visit order); the lanes look at 64 keys of a query at a time, the first two
// whose feature is not blocked RIGHT NOW are the loop's best and second best.  Per feature: blocked (the occupant has
// Observations() > 0) and the last writer -- F.mvpMapPoints as the reference mutates it.  ~1 us per query (dependent reads);
// only the state above pays it, every other search keeps the fixpoint kernel.
__device__ __forceinline__ void proj_inorder_body(const ProjDev& P)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = P.n, nq = P.nq;
    int32_t* const blocked = P.minW; // n entries (the sweeps' table, unused here)
    for (int i = tid; i < n; i += PROJ_THREADS) {
        blocked[i] = (P.taken0 && P.taken0[i]) ? 1 : 0;
        P.featMatch[i] = -1;
    }
    __threadfence_block();
    __syncthreads();
    if (tid >= 64) return; // (no barrier below)
    int cnt = 0, prevRejected = 0;
    for (int q = 0; q < nq; q++) {
        const int flags = P.qflags ? P.qflags[q] : 0;
        const bool bRight = flags & 1;
        const bool skip = q > 0 && (((flags & 2) && prevRejected) || ((flags & 4) && P.qArea[q - 1] == 0));
        prevRejected = 0;
        const int m = skip ? 0 : P.qCount[q];
        const unsigned long long* const K = P.sortedKeys + P.qStart[q];
        int g1 = -1, d1 = 256, g2 = -1, d2 = 256;
        bool done_ = false;
        for (int base = 0; base < m && !done_; base += 64) {
            const unsigned long long key = base + lane < m ? K[base + lane] : ~0ull;
            const int d = (int)(key >> 55);
            const int g = (int)(key & 0xFFFFFF);
            const bool live = d < 256; // (`dist<bestDist` with bestDist = 256 never accepts the others; keys are sorted)
            const bool cand = live && blocked[g] == 0;
            unsigned long long mask = __ballot(cand);
            const bool ended = __ballot(!live) != 0ull; // (the keys are sorted: nothing behind this round can be accepted)
            while (mask && !done_) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int gl = __shfl(g, l), dl = __shfl(d, l);
                if (g1 < 0) {
                    g1 = gl;
                    d1 = dl;
                    if (P.mode != 0) done_ = true;
                } else {
                    g2 = gl;
                    d2 = dl;
                    done_ = true;
                }
            }
            if (ended) done_ = true;
        }
        int choice = -1, partner = -1;
        if (g1 >= 0 && d1 <= P.thHigh) {
            bool ok = true;
            if (P.mode == 0) {
                const int lvl1 = P.octave[g1];
                const int bestLevel2 = g2 >= 0 ? P.octave[g2] : -1;
                if (lvl1 == bestLevel2 && (float)d1 > __fmul_rn(P.nnratio, (float)d2)) {
                    ok = false;
                    prevRejected = 1;
                }
            }
            if (ok) {
                choice = g1;
                if (P.mode == 0 && P.Nleft != -1) {
                    if (!bRight && P.l2r && P.l2r[g1] != -1) partner = P.l2r[g1] + P.Nleft;
                    if (bRight && P.r2l && P.r2l[g1 - P.Nleft] != -1) partner = P.r2l[g1 - P.Nleft];
                }
            }
        }
        if (lane == 0) {
            const int blocks = (!P.qblocks || P.qblocks[q]) ? 1 : 0;
            P.qMatch[q] = choice;
            if (choice >= 0) { // F.mvpMapPoints[bestIdx] = pMP
                blocked[choice] = blocks;
                P.featMatch[choice] = q;
            }
            if (partner >= 0) { // ... and the stereo partner's entry, whoever held it (:117-121)
                blocked[partner] = blocks;
                P.featMatch[partner] = q;
            }
        }
        cnt += (choice >= 0) + (partner >= 0);
        __threadfence_block(); // the next query's lanes read what lane 0 has just written
    }
    if (lane == 0) {
        P.status[0] = cnt;
        P.status[1] = 1;
    }
}

__device__ __forceinline__ void proj_sweeps_body(const ProjDev& P)
{
    if (P.inorder) { // (uniform)
        proj_inorder_body(P);
        return;
    }
    __shared__ int sChanged;
    PT_BEGIN();
    extern __shared__ int32_t projLds[]; // (2 n + 6 nq) ints when the host found that they fit, else nothing
    const int tid = threadIdx.x;
    const int n = P.n, nq = P.nq;
    // the per-feature "least blocking writer" tables and the per-query states of both sweep parities: every sweep
    // reads and rewrites all of them, so they live in LDS whenever the frame is small enough (the usual case)
    int32_t* const minW = P.sweepLds ? projLds : P.minW;
    int32_t* const state = P.sweepLds ? projLds + 2 * (size_t)n : P.state;
    for (int i = tid; i < 2 * n; i += PROJ_THREADS) minW[i] = 0x7fffffff;
    for (int i = tid; i < 2 * 3 * nq; i += PROJ_THREADS) state[i] = -2;
    __syncthreads();
    // Round 4: what a query reads in EVERY sweep is fetched once.  A sweep used to walk the query's sorted keys in global memory
    // until it met a feature no earlier query blocks -- one dependent load per key, then the octaves of the two survivors, the
    // query's flags (in pinned host memory on the latency path: a PCIe round trip per sweep) -- ~5 us per sweep for work that is
    // a handful of compares.  The first PROJ_CK keys of the thread's first query, the octaves of their features and the query's
    // flags now sit in registers; a sweep touches LDS only, and goes back to the key array only when all cached keys are blocked.
    PT(1); // init
    constexpr int PROJ_CK = 4;
    unsigned long long ck[PROJ_CK];
    int co[PROJ_CK];
    int cFlags = 0, cBlocks = 1, cM = 0;
    const unsigned long long* cK = nullptr;
    if (tid < nq) {
        cFlags = P.qflags ? P.qflags[tid] : 0;
        cBlocks = (!P.qblocks || P.qblocks[tid]) ? 1 : 0;
        cM = P.qCount[tid];
        cK = P.sortedKeys + P.qStart[tid];
#pragma unroll
        for (int k = 0; k < PROJ_CK; k++) ck[k] = k < cM ? cK[k] : ~0ull;
#pragma unroll
        for (int k = 0; k < PROJ_CK; k++) co[k] = (P.mode == 0 && k < cM && (int)(ck[k] >> 55) < 256) ? P.octave[(int)(ck[k] & 0xFFFFFF)] : -1;
    }
#ifdef ORBFE_PROJ_TIMING
    asm volatile("" ::"v"(ck[0]), "v"(ck[3]), "v"(co[0]), "v"(co[3]), "v"(cFlags));
#endif
    PT(2); // cache
    int sweep = 0, last = 0;
    for (; sweep < nq + 2; sweep++) {
        const int32_t* prevW = minW + (size_t)(sweep & 1) * n;
        int32_t* newW = minW + (size_t)((sweep + 1) & 1) * n;
        const int32_t* prevS = state + (size_t)(sweep & 1) * 3 * nq;
        int32_t* newS = state + (size_t)((sweep + 1) & 1) * 3 * nq;
        last = (sweep + 1) & 1;
        for (int i = tid; i < n; i += PROJ_THREADS) newW[i] = 0x7fffffff;
        if (tid == 0) sChanged = 0;
        __syncthreads();
        for (int q = tid; q < nq; q += PROJ_THREADS) {
            const bool mine = q == tid; // (the thread's first query: cached)
            const int flags = mine ? cFlags : (P.qflags ? P.qflags[q] : 0);
            const bool bRight = flags & 1;
            int choice = -1, partner = -1, rejected = 0;
            const bool skip = q > 0 && (((flags & 2) && prevS[3 * (q - 1) + 2] == 1) || ((flags & 4) && P.qArea[q - 1] == 0));
            const int m = skip ? 0 : (mine ? cM : P.qCount[q]);
            const unsigned long long* K = mine ? cK : P.sortedKeys + P.qStart[q];
            int g1 = -1, d1 = 256, g2 = -1, d2 = 256, o1 = -1, o2 = -1;
            bool done_ = false; // the walk over the keys is over (two survivors, or one in mode 1, or the distances ran out)
            auto visit = [&](unsigned long long key, int oct) { // one key of the walk (:90-131 / :2285-2296); oct: its feature's octave or -2 = not loaded
                const int d = (int)(key >> 55);
                if (d >= 256) { // `dist<bestDist` with bestDist = 256 never accepts these
                    done_ = true;
                    return;
                }
                const int g = (int)(key & 0xFFFFFF);
                if (prevW[g] < q) return;
                if (g1 < 0) {
                    g1 = g;
                    d1 = d;
                    o1 = oct;
                    if (P.mode != 0) done_ = true;
                } else {
                    g2 = g;
                    d2 = d;
                    o2 = oct;
                    done_ = true;
                }
            };
            int k = 0;
            if (mine) {
#pragma unroll
                for (int c = 0; c < PROJ_CK; c++)
                    if (!done_ && c < m) {
                        visit(ck[c], co[c]);
                        k = c + 1;
                    }
            }
            for (; !done_ && k < m; k++) visit(K[k], -2);
            if (g1 >= 0 && d1 <= P.thHigh) {
                bool ok = true;
                if (P.mode == 0) {
                    const int lvl1 = o1 != -2 ? o1 : P.octave[g1];
                    const int bestLevel2 = g2 >= 0 ? (o2 != -2 ? o2 : P.octave[g2]) : -1;
                    if (lvl1 == bestLevel2 && (float)d1 > __fmul_rn(P.nnratio, (float)d2)) {
                        ok = false;
                        rejected = 1;
                    }
                }
                if (ok) {
                    choice = g1;
                    if (P.mode == 0 && P.Nleft != -1) {
                        if (!bRight && P.l2r && P.l2r[g1] != -1) partner = P.l2r[g1] + P.Nleft;
                        if (bRight && P.r2l && P.r2l[g1 - P.Nleft] != -1) partner = P.r2l[g1 - P.Nleft];
                    }
                }
            }
            newS[3 * q] = choice;
            newS[3 * q + 1] = partner;
            newS[3 * q + 2] = rejected;
            if (choice != prevS[3 * q] || partner != prevS[3 * q + 1] || rejected != prevS[3 * q + 2]) sChanged = 1;
            if (mine ? cBlocks != 0 : (!P.qblocks || P.qblocks[q])) {
                if (choice >= 0) atomicMin(&newW[choice], q);
                if (partner >= 0) atomicMin(&newW[partner], q);
            }
        }
        __syncthreads();
        if (!sChanged) break;
        __syncthreads();
    }

    PT(3); // sweeps
    // ---- final state of F.mvpMapPoints: the last writer of every feature
    const int32_t* S = state + (size_t)last * 3 * nq;
    if (P.mirror && P.sweepLds) {
        // latency path: the feature table is built in LDS (the sweeps' minW buffer is free now) and goes straight to the pinned
        // mirror as plain stores -- no atomics in global memory, no copy out of it -- with the completion word behind it
        __shared__ int sCnt;
        int32_t* const fm = minW; // n entries
        for (int i = tid; i < n; i += PROJ_THREADS) fm[i] = -1;
        if (tid == 0) sCnt = 0;
        // (k_proj_candidates' count: the host checks it against the buffers.  Counting in a word of the arena that this kernel
        // puts back to zero, so that no memset has to run in front of the call, was measured 9 us SLOWER per call -- 0.0637
        // against 0.0546 ms, twice each on one box -- and is gone.)
        const int keysNeeded = tid == 0 ? P.status[2] : 0;
        __syncthreads();
        int cnt = 0;
        for (int q = tid; q < nq; q += PROJ_THREADS) {
            const int c = S[3 * q], p = S[3 * q + 1];
            P.mirror[4 + q] = c;
            if (c >= 0) {
                atomicMax(&fm[c], q);
                cnt++;
            }
            if (p >= 0) {
                atomicMax(&fm[p], q);
                cnt++;
            }
        }
        cnt = wave_sum_i32(cnt);
        if ((tid & 63) == 0 && cnt) atomicAdd(&sCnt, cnt);
        __syncthreads();
        PT(4); // final
        for (int i = tid; i < n; i += PROJ_THREADS) P.mirror[4 + nq + i] = fm[i];
        if (tid == 0) {
            P.mirror[0] = sCnt;
            P.mirror[1] = sweep + 1;
            P.mirror[2] = keysNeeded;
            P.mirror[3] = 0;
        }
        own_stores_acknowledged(); // (one workgroup: the release in front of the flag below is the workgroup's)
        __syncthreads();
        if (tid == 0 && P.doneFlag) {
            __threadfence_system();
            *(volatile unsigned*)P.doneFlag = P.doneSeq;
        }
    } else {
    for (int i = tid; i < n; i += PROJ_THREADS) P.featMatch[i] = -1;
    if (tid == 0) {
        P.status[0] = 0;
        P.status[1] = sweep + 1;
    }
    __syncthreads();
    int cnt = 0;
    for (int q = tid; q < nq; q += PROJ_THREADS) {
        const int c = S[3 * q], p = S[3 * q + 1];
        P.qMatch[q] = c;
        if (c >= 0) {
            atomicMax(&P.featMatch[c], q);
            cnt++;
        }
        if (p >= 0) {
            atomicMax(&P.featMatch[p], q);
            cnt++;
        }
    }
    cnt = wave_sum_i32(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&P.status[0], cnt);
    PT(4); // final
    if (P.mirror) { // (uniform) the sweeps' tables did not fit LDS: results built in device memory, then copied
        __threadfence();
        __syncthreads();
        for (int i = tid; i < P.mirrorInts; i += PROJ_THREADS) P.mirror[i] = P.status[i];
        own_stores_acknowledged(); // (one workgroup: the release in front of the flag below is the workgroup's)
        __syncthreads();
        if (tid == 0 && P.doneFlag) {
            __threadfence_system();
            *(volatile unsigned*)P.doneFlag = P.doneSeq;
        }
    }
    }
    PT(5); // mirror
    PT_END(sweep + 1);
}

// one problem per launch (argument by value) / one problem per blockIdx.y (orbfe_search_projection_batch)
__global__ __launch_bounds__(PROJ_THREADS) void k_proj_grid(ProjDev P) { proj_grid_body(P); }
__global__ __launch_bounds__(256) void k_proj_candidates(ProjDev P) { proj_candidates_body(P); }
__global__ __launch_bounds__(PROJ_THREADS) void k_proj_sweeps(ProjDev P) { proj_sweeps_body(P); }
__global__ __launch_bounds__(PROJ_THREADS) void k_proj_grid_batch(const ProjDev* __restrict__ Ps)
{
    const ProjDev P = Ps[blockIdx.y];
    proj_grid_body(P);
}
__global__ __launch_bounds__(256) void k_proj_candidates_batch(const ProjDev* __restrict__ Ps)
{
    const ProjDev P = Ps[blockIdx.y];
    proj_candidates_body(P);
}
__global__ __launch_bounds__(PROJ_THREADS) void k_proj_sweeps_batch(const ProjDev* __restrict__ Ps)
{
    const ProjDev P = Ps[blockIdx.y];
    proj_sweeps_body(P);
}

// ------------------------------------------------------------------ K-INIT
// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821) on K-PROJ's grid and sorted candidate keys.
// The sequential rule here is distance dependent: F2 feature i2 is skipped by query q when an EARLIER query
// holds it with a distance <= dist(q, i2) (vMatchedDistance, :744), and a better match steals it (:765-772).
// Same fixpoint scheme as k_proj_sweeps: every sweep, each accepted query claims its feature in a per-feature
// list; a query evaluates its sorted keys against the previous sweep's claims of queries with a smaller index.
// The result of q depends only on queries < q, so the fixpoint is unique and equals the sequential run.
struct InitDev {
    ProjDev P;
    float nnratio;
    int32_t* head;   // 2 * n: newest claimant of a feature, per sweep parity
    int32_t* next;   // 2 * nq: linked list through the claimants
    int32_t* choice; // 2 * nq: claimed feature or -1
    int32_t* cdist;  // 2 * nq: its distance
};
__global__ __launch_bounds__(PROJ_THREADS) void k_init_sweeps(InitDev I)
{
    __shared__ int sChanged;
    const ProjDev& P = I.P;
    const int tid = threadIdx.x, n = P.n, nq = P.nq;
    for (int i = tid; i < 2 * n; i += PROJ_THREADS) I.head[i] = -1;
    for (int i = tid; i < 2 * nq; i += PROJ_THREADS) {
        I.choice[i] = -1;
        I.cdist[i] = 0;
        I.next[i] = -1;
    }
    __syncthreads();
    int sweep = 0, last = 0;
    for (; sweep < nq + 2; sweep++) {
        const int pv = sweep & 1, cu = pv ^ 1;
        last = cu;
        const int32_t *headP = I.head + (size_t)pv * n, *nextP = I.next + (size_t)pv * nq;
        const int32_t *choiceP = I.choice + (size_t)pv * nq, *distP = I.cdist + (size_t)pv * nq;
        int32_t *headC = I.head + (size_t)cu * n, *nextC = I.next + (size_t)cu * nq;
        int32_t *choiceC = I.choice + (size_t)cu * nq, *distC = I.cdist + (size_t)cu * nq;
        for (int i = tid; i < n; i += PROJ_THREADS) headC[i] = -1;
        if (tid == 0) sChanged = 0;
        __syncthreads();
        for (int q = tid; q < nq; q += PROJ_THREADS) {
            const int m = P.qCount[q];
            const unsigned long long* K = P.sortedKeys + P.qStart[q];
            int g1 = -1, d1 = 0x7fffffff, d2 = 0x7fffffff;
            for (int k = 0; k < m; k++) {
                const unsigned long long key = K[k];
                const int d = (int)(key >> 55), g = (int)(key & 0xFFFFFF);
                bool blocked = false; // vMatchedDistance[i2] <= dist, as left by the queries before q (:744)
                for (int p = headP[g]; p >= 0; p = nextP[p])
                    if (p < q && distP[p] <= d) {
                        blocked = true;
                        break;
                    }
                if (blocked) continue;
                if (g1 < 0) {
                    g1 = g;
                    d1 = d;
                } else {
                    d2 = d;
                    break;
                }
            }
            int c = -1;
            if (g1 >= 0 && d1 <= TH_LOW && (float)d1 < __fmul_rn((float)d2, I.nnratio)) c = g1; // :760-763
            choiceC[q] = c;
            distC[q] = d1;
            if (c != choiceP[q] || (c >= 0 && d1 != distP[q])) sChanged = 1;
            if (c >= 0) nextC[q] = atomicExch(&headC[c], q);
        }
        __syncthreads();
        if (!sChanged) break;
        __syncthreads();
    }
    const int32_t* S = I.choice + (size_t)last * nq;
    for (int q = tid; q < nq; q += PROJ_THREADS) P.qMatch[q] = S[q];
    if (tid == 0) P.status[1] = sweep + 1;
}

// ------------------------------------------------------------------ K-DIST
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:387-419): among the N observation descriptors of
// a map point pick the one with the least median Hamming distance to all of them (self distance 0
// included, median = sorted[(int)(0.5*(N-1))], first minimum wins).  One wavefront per map point, one
// descriptor per lane; the k-th smallest distance of a row is found by bisection on the value
// (distances are 0..256), recomputing the popcounts instead of storing an N x N matrix.
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t* __restrict__ pool,
                                                     const int32_t* __restrict__ offsets, int npts,
                                                     int32_t* __restrict__ best)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;
    if (p >= npts) return;
    const int o = offsets[p], N = offsets[p + 1] - o;
    if (N <= 0) {
        if (lane == 0) best[p] = -1;
        return;
    }
    const uint8_t* D = pool + (size_t)o * 32;
    const int k = (int)(0.5 * (double)(N - 1));
    unsigned bestKey = 0xFFFFFFFFu; // median << 20 | index
    for (int i = lane; i < N; i += 64) {
        const Desc di = load_desc(D + (size_t)i * 32);
        int lo = 0, hi = 256; // smallest v with #{j : d_ij <= v} >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) cnt += hamming(di, load_desc(D + (size_t)j * 32)) <= mid;
            if (cnt >= k + 1) hi = mid;
            else lo = mid + 1;
        }
        bestKey = min(bestKey, ((unsigned)lo << 20) | (unsigned)i);
    }
    bestKey = wave_min_u32(bestKey);
    if (lane == 0) best[p] = (int)(bestKey & 0xFFFFFu);
}

// ------------------------------------------------------------------ K-VOC
// DBoW2 TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259): walk the
// vocabulary tree, at every level the child with the smallest Hamming distance (first minimum in stored
// order, strict '<').  16 lanes per feature: one child per lane, group-min over (distance<<8 | order).
__global__ __launch_bounds__(256) void k_vocab_transform(const uint8_t* __restrict__ nodeDesc,
                                                         const int32_t* __restrict__ childOff,
                                                         const int32_t* __restrict__ childIds,
                                                         const int32_t* __restrict__ nodeWord,
                                                         const double* __restrict__ nodeWeight, int L,
                                                         const uint8_t* __restrict__ feats, int n, int levelsup,
                                                         int32_t* __restrict__ wordOut, int32_t* __restrict__ nodeOut,
                                                         double* __restrict__ weightOut, const DoneSig doneSig)
{
    __shared__ unsigned wgCnt;
    done_begin(doneSig, &wgCnt);
    const int sub = threadIdx.x & 15;
    const int f = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = f < n;
    const Desc df = live ? load_desc(feats + (size_t)f * 32) : Desc{};
    const int nidLevel = L - levelsup;
    int nid = 0, finalId = 0, level = 0;
    bool done = !live;
    // all 16 lanes of a group follow the same path; groups of a wave may finish at different depths
    for (int guard = 0; guard < 64; guard++) {
        const int c0 = done ? 0 : childOff[finalId], c1 = done ? 0 : childOff[finalId + 1];
        if (c0 >= c1) done = true; // leaf
        if (__ballot(!done) == 0ull) break;
        unsigned best = 0xFFFFFFFFu;
        if (!done)
            for (int k = c0 + sub; k < c1; k += 16) {
                const int id = childIds[k];
                const unsigned d = (unsigned)hamming(df, load_desc(nodeDesc + (size_t)id * 32));
                best = min(best, (d << 20) | (unsigned)(k - c0));
            }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off, 16));
        if (!done) {
            finalId = childIds[c0 + (int)(best & 0xFFFFFu)];
            level++;
            if (level == nidLevel) nid = finalId;
        }
    }
    if (live && sub == 0) {
        wordOut[f] = nodeWord[finalId];
        weightOut[f] = nodeWeight[finalId];
        nodeOut[f] = nid;
    }
    wave_done(doneSig, &wgCnt);
}

// ------------------------------------------------------------------ K-KB8
__global__ __launch_bounds__(256) void k_kb8_unproject(const float* __restrict__ P, const float* __restrict__ uv,
                                                       int n, float* __restrict__ rays)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    orbfe_kb8_unproject_dev(P, uv[2 * i], uv[2 * i + 1], rays + 3 * i);
}

// ------------------------------------------------------------- host helpers
// Per-thread, per-device arena: matcher calls are tiny (tens of KB), so hipMalloc/hipFree per call
// would cost more than the kernels.  The arena is a bump allocator over one persistent device
// buffer; a call that outgrows it falls back to hipMalloc for the overflow and the arena is
// enlarged before the next call.
bool is_device_ptr(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // ordinary host memory is "invalid value" to some runtimes
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

struct Arena {
    int device = -1;
    uint8_t* base = nullptr;
    uint8_t* pin = nullptr; // pinned host mirror of the arena: inputs are staged here and go up in ONE transfer,
                            // outputs come down into it in ONE transfer
    uint8_t* pinDev = nullptr; // the address a KERNEL uses for `pin` (results written into the mirror by the kernel itself)
    bool pinCoherent = false;  // `pin` was allocated hipHostMallocCoherent
    size_t cap = 0, off = 0, want = 0;
    // The calling thread's own non-blocking stream on this device: matcher calls of the Tracking, LocalMapping and
    // LoopClosing threads neither serialise with each other nor synchronise with the legacy null stream (and through
    // it with every blocking stream of the process, e.g. torch's default stream).
    hipStream_t stream = nullptr;
    // completion word of the latency-path calls (DoneSig): device counter, page-locked flag, sequence number
    unsigned* doneCtr = nullptr;
    unsigned* doneFlag = nullptr;    // host address
    unsigned* doneFlagDev = nullptr; // the kernel's address of the same word
    unsigned doneSeq = 0;
    unsigned spinProbe = 0; // calls since the word was given up (done_words re-probes now and then)
    int spinMisses = 0; // consecutive waits in which the word did not arrive within the bound; at 8 the word is given up for
                        // this thread (a platform where the kernel's flag store does not reach the host while the kernel runs
                        // would otherwise cost every call the full bound)
    // the clean block: device memory that is all ones between calls (the kernels' scattered results; DoneSig)
    uint8_t* cleanDev = nullptr;
    size_t cleanCap = 0;
    bool cleanDirty = false;
    ~Arena()
    { // thread exit: give the scratch back (a thread that called the matcher once used to leak it)
        if (device < 0) return;
        if (hipSetDevice(device) != hipSuccess) return;
        if (stream) (void)hipStreamSynchronize(stream);
        if (base) (void)hipFree(base);
        if (pin) (void)hipHostFree(pin);
        if (doneCtr) (void)hipFree(doneCtr);
        if (cleanDev) (void)hipFree(cleanDev);
        if (doneFlag) (void)hipHostFree(doneFlag);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
const int kMaxDevices = 16;
thread_local Arena g_arena[kMaxDevices];
thread_local hipStream_t g_ms = nullptr; // stream of the matcher call in progress on this thread

// Device blocks of the resident handles (orbfe_keyframe_*, orbfe_frame_*): a Frame handle lives for a frame, a KeyFrame handle
// for as long as the adapter's table keeps it, and hipMalloc / hipFree cost 10-20 us each -- as much as the search the handle
// is made for.  Freed blocks wait here (per device, up to 64 of them) for the next handle of about their size.
struct BlockPool {
    std::mutex m;
    struct Blk {
        void* p;
        size_t cap;
    };
    std::vector<Blk> freeBlocks[kMaxDevices];
    void* get(int device, size_t bytes, size_t* cap)
    {
        const size_t want = (bytes + 0xFFFF) & ~(size_t)0xFFFF; // 64-KB classes
        {
            std::lock_guard<std::mutex> g(m);
            auto& v = freeBlocks[device];
            for (size_t i = 0; i < v.size(); i++)
                if (v[i].cap >= want && v[i].cap <= 2 * want) {
                    const Blk b = v[i];
                    v[i] = v.back();
                    v.pop_back();
                    *cap = b.cap;
                    return b.p;
                }
        }
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        *cap = want;
        return p;
    }
    void put(int device, void* p, size_t cap)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(m);
            auto& v = freeBlocks[device];
            if (v.size() < 64) {
                v.push_back(Blk{p, cap});
                return;
            }
        }
        (void)hipFree(p);
    }
};
BlockPool g_blockPool;

// The staged inputs of a call brought to the device by a KERNEL (16 bytes per thread out of the pinned mirror) which also puts
// the all-ones into the result region -- instead of a clearing command, a copy command and the ~10 us the queue spends between
// two commands of different engines (SearchByBoW x 64 with the nodes paired on the device: fill 6 + gap 10 + copy 9 + gap 10 in
// front of the kernel -> one launch of ~5 us).  Used when what is staged is small (Scratch::flush_by_kernel).
struct StageRuns {
    const uint4* src[4];
    uint4* dst[4];
    unsigned n16[4]; // 16-byte units per run (unused runs: 0)
    uint4* fill;
    unsigned fill16;
};
__global__ __launch_bounds__(256) void k_stage_in(const StageRuns R)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (unsigned i = t; i < R.n16[k]; i += step) R.dst[k][i] = R.src[k][i];
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (unsigned i = t; i < R.fill16; i += step) R.fill[i] = ones;
}

struct Scratch { // device allocations of one call
    Arena* ar = nullptr;
    // Latency path (round 4): a call whose staged inputs are a few KB hands the KERNEL the pinned mirror itself (device-side
    // address of the host memory) instead of copying it to the device first: one stream command less in front of the launch.
    // Set before the first up() / reserve(); needs the arena's mirror (else the call takes the copy path as before).
    bool inPlace = false;
    std::vector<void*> overflow;
    std::vector<std::pair<size_t, size_t>> staged; // (offset, bytes) runs waiting in the pinned mirror
    struct Down {
        void* host;
        const void* dev;
        size_t bytes;
    };
    std::vector<Down> downs; // results the caller wants back (see down / fetch)
    explicit Scratch(int device)
    {
        ar = &g_arena[device]; // select_device() has checked 0 <= device < kMaxDevices
        ar->device = device;
        if (!ar->stream && hipStreamCreateWithFlags(&ar->stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            ar->stream = nullptr; // the null stream still works
        }
        g_ms = ar->stream;
        if (ar->want > ar->cap) { // grow between calls
            if (ar->base) (void)hipFree(ar->base);
            if (ar->pin) (void)hipHostFree(ar->pin);
            ar->base = ar->pin = nullptr;
            ar->cap = 0;
            void* p = nullptr;
            const size_t want = std::max<size_t>(ar->want * 2, 1 << 20);
            if (hipMalloc(&p, want) == hipSuccess) {
                ar->base = (uint8_t*)p;
                ar->cap = want;
                void* h = nullptr;
                ar->pinDev = nullptr;
                // (explicitly fine-grained: kernels write results into it that the host reads while the kernel is, for the
                // runtime, still running -- DoneSig; without the flag the default allocation serves the same way)
                ar->pinCoherent = hipHostMalloc(&h, want, hipHostMallocCoherent) == hipSuccess;
                if (!ar->pinCoherent) (void)hipGetLastError();
                if (ar->pinCoherent || hipHostMalloc(&h, want) == hipSuccess) {
                    ar->pin = (uint8_t*)h;
                    void* dv = nullptr;
                    if (hipHostGetDevicePointer(&dv, h, 0) == hipSuccess) ar->pinDev = (uint8_t*)dv;
                    else (void)hipGetLastError();
                } else (void)hipGetLastError();
            }
        }
        ar->off = 0;
        ar->want = 0;
    }
    ~Scratch()
    {
        if (!overflow.empty()) (void)hipStreamSynchronize(g_ms);
        for (void* p : overflow) (void)hipFree(p);
    }
    template <class T>
    int up(T** out, const T* host, size_t n)
    {
        *out = nullptr;
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        void* p = nullptr;
        ar->want += bytes;
        bool inArena = false;
        size_t at = 0;
        if (ar->base && ar->off + bytes <= ar->cap) {
            at = ar->off;
            p = ar->base + ar->off;
            ar->off += bytes;
            inArena = true;
        } else {
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return -(1000 + (int)e);
            overflow.push_back(p);
        }
        if (host && n) {
            if (inArena && ar->pin) { // stage; adjacent uploads merge into one run
                std::memcpy(ar->pin + at, host, n * sizeof(T));
                if (inPlace && ar->pinDev) {
                    *out = (T*)(ar->pinDev + at); // (read where it lies)
                    return 0;
                }
                if (!staged.empty() && staged.back().first + staged.back().second == at) staged.back().second += bytes;
                else staged.emplace_back(at, bytes);
            } else {
                hipError_t e = hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, g_ms);
                if (e != hipSuccess) return -(1000 + (int)e);
                e = hipStreamSynchronize(g_ms); // `host` is the caller's (pageable) memory
                if (e != hipSuccess) return -(1000 + (int)e);
            }
        }
        *out = (T*)p;
        return 0;
    }
    // Arena space the caller fills itself: *stage points into the pinned mirror (the bytes go up with the other
    // staged inputs in flush()), so a pooled upload needs no intermediate copy.  Falls back to a temporary host
    // buffer when the arena is too small for this call (it is enlarged before the next one).
    std::vector<std::vector<uint8_t>> temps;
    struct LateUp {
        void* dev;
        size_t temp, bytes;
    };
    std::vector<LateUp> lateUps;
    template <class T>
    int reserve(T** dev, T** stage, size_t n)
    {
        *dev = nullptr;
        *stage = nullptr;
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        ar->want += bytes;
        if (ar->base && ar->pin && ar->off + bytes <= ar->cap) {
            const size_t at = ar->off;
            ar->off += bytes;
            *stage = (T*)(ar->pin + at);
            if (inPlace && ar->pinDev) {
                *dev = (T*)(ar->pinDev + at);
                return 0;
            }
            *dev = (T*)(ar->base + at);
            if (!staged.empty() && staged.back().first + staged.back().second == at) staged.back().second += bytes;
            else staged.emplace_back(at, bytes);
            return 0;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return -(1000 + (int)e);
        overflow.push_back(p);
        temps.emplace_back(bytes);
        lateUps.push_back(LateUp{p, temps.size() - 1, bytes});
        *dev = (T*)p;
        *stage = (T*)temps.back().data();
        return 0;
    }
    // Small results the KERNEL writes straight into the pinned mirror (posted writes over PCIe): no download command at
    // the end of the call, the caller reads *host after the stream synchronisation.  *dev is the kernel's address of that
    // host memory; the caller pre-fills *host (e.g. with -1) before the launch.  Fails (returns 1) when the arena has no
    // mirror or no room: the caller then takes the download path.
    template <class T>
    int mirror_out(T** dev, T** host, size_t n)
    {
        *dev = nullptr;
        *host = nullptr;
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        ar->want += bytes;
        if (!(ar->base && ar->pin && ar->pinDev && ar->off + bytes <= ar->cap)) return 1;
        const size_t at = ar->off;
        ar->off += bytes;
        *dev = (T*)(ar->pinDev + at);
        *host = (T*)(ar->pin + at);
        return 0;
    }
    // `bytes` of the arena's pinned mirror as plain staging (the caller copies from it itself): nullptr when it does not fit
    uint8_t* pin_scratch(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        ar->want += bytes;
        if (!(ar->base && ar->pin && ar->off + bytes <= ar->cap)) return nullptr;
        uint8_t* p = ar->pin + ar->off;
        ar->off += bytes;
        return p;
    }
    // Descriptor arrays may already live on the device (an extractor's resident output slab, a gathered slab):
    // then they are read in place.
    int up_desc(uint8_t** out, const uint8_t* hostOrDev, size_t n)
    {
        if (hostOrDev && n && is_device_ptr(hostOrDev)) {
            // (an extractor may still be writing it on its own stream: orbfe_order.h)
            const int w = orbfe_producer_wait(hostOrDev, g_ms);
            if (w < 0) return w;
            *out = const_cast<uint8_t*>(hostOrDev);
            return 0;
        }
        return up(out, hostOrDev, n);
    }
    // send the staged inputs (called before the first kernel of the call, by KernelScope)
    int flush()
    {
        for (const auto& r : staged) {
            hipError_t e = hipMemcpyAsync(ar->base + r.first, ar->pin + r.first, r.second, hipMemcpyHostToDevice, g_ms);
            if (e != hipSuccess) return -(1000 + (int)e);
        }
        staged.clear();
        for (const LateUp& u : lateUps) {
            hipError_t e = hipMemcpyAsync(u.dev, temps[u.temp].data(), u.bytes, hipMemcpyHostToDevice, g_ms);
            if (e != hipSuccess) return -(1000 + (int)e);
        }
        if (!lateUps.empty()) {
            hipError_t e = hipStreamSynchronize(g_ms); // pageable sources
            if (e != hipSuccess) return -(1000 + (int)e);
            lateUps.clear();
        }
        return 0;
    }
    // flush() as ONE kernel that also fills [fill, fill + fillBytes) with ones (k_stage_in); false: not applicable (no device
    // alias of the mirror, more than four runs, a lot of bytes, late uploads) -- nothing was queued, the caller takes flush()
    // and a clearing command
    bool flush_by_kernel(void* fill, size_t fillBytes)
    {
        if (!ar->pinDev || !ar->base || staged.size() > 4 || !lateUps.empty()) return false;
        size_t total = 0;
        for (const auto& r : staged) total += r.second;
        if (total > (256u << 10) || (fillBytes >> 4) > 0xFFFFFFFFull) return false;
        StageRuns R;
        std::memset(&R, 0, sizeof R);
        for (size_t k = 0; k < staged.size(); k++) {
            R.src[k] = reinterpret_cast<const uint4*>(ar->pinDev + staged[k].first);
            R.dst[k] = reinterpret_cast<uint4*>(ar->base + staged[k].first);
            R.n16[k] = (unsigned)(staged[k].second >> 4);
        }
        R.fill = reinterpret_cast<uint4*>(fill);
        R.fill16 = (unsigned)((fillBytes + 15) >> 4);
        const size_t units = std::max<size_t>(total >> 4, R.fill16);
        const unsigned wgs = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (units + 255) / 256));
        hipLaunchKernelGGL(k_stage_in, dim3(wgs), dim3(256), 0, g_ms, R);
        if (hipGetLastError() != hipSuccess) return false;
        staged.clear();
        return true;
    }
    // Results: down() names a device range the caller wants in `host`; fetch() brings all of them back with ONE
    // transfer of the arena stretch that covers them into the pinned mirror (the outputs of a call are neighbours in
    // the arena), one stream synchronisation, and a memcpy each -- instead of one blocking pageable copy per array.
    int down(void* host, const void* dev, size_t bytes)
    {
        if (bytes) downs.push_back(Down{host, dev, bytes});
        return 0;
    }
    int fetch()
    {
        bool inArena = ar->base && ar->pin && !downs.empty();
        size_t lo = ~(size_t)0, hi = 0, total = 0;
        for (const Down& d : downs) total += d.bytes;
        if (total > (1u << 20)) inArena = false; // a distance matrix: straight into the caller's memory, no second copy
        for (const Down& d : downs) {
            const uint8_t* p = (const uint8_t*)d.dev;
            if (!(ar->base && p >= ar->base && p + d.bytes <= ar->base + ar->cap)) inArena = false;
            else {
                lo = std::min(lo, (size_t)(p - ar->base));
                hi = std::max(hi, (size_t)(p - ar->base) + d.bytes);
            }
        }
        hipError_t e = hipSuccess;
        if (inArena) {
            e = hipMemcpyAsync(ar->pin + lo, ar->base + lo, hi - lo, hipMemcpyDeviceToHost, g_ms);
            if (e == hipSuccess) e = hipStreamSynchronize(g_ms);
            if (e == hipSuccess)
                for (const Down& d : downs) std::memcpy(d.host, ar->pin + ((const uint8_t*)d.dev - ar->base), d.bytes);
        } else {
            for (const Down& d : downs)
                if (e == hipSuccess) e = hipMemcpyAsync(d.host, d.dev, d.bytes, hipMemcpyDeviceToHost, g_ms);
            if (e == hipSuccess) e = hipStreamSynchronize(g_ms);
        }
        downs.clear();
        return e == hipSuccess ? 0 : -(1000 + (int)e);
    }
    // ---- results and completion of the latency-path calls (DoneSig above): out_block() says where the kernel puts its results
    // and where the host finds them, done_sig() hands the kernel the call's sequence number when the completion word may be
    // used, complete() spins on the word for a bounded time -- a call that takes longer gains nothing from spinning -- and
    // falls back to the stream synchronisation, which also surfaces a failed launch.  ORBFE_MATCHER_SPIN=0 (or ORBFE_SPIN=0)
    // switches the word off.  The flag word is allocated coherent like the mirror.
    struct OutBlock {
        uint8_t* dev = nullptr;   // the arena's clean block: device memory, all ones between calls; the kernel scatters into it
        uint8_t* host = nullptr;  // its pinned mirror (host address): complete when complete() returns
        uint4* mirrorDev = nullptr;
        size_t bytes = 0;
        bool direct = false;      // dev IS the mirror (the kernel's address of it): the kernel's stores cross PCIe themselves
    };
    // `bytes` of the arena's clean block: device memory that is all ones between calls (whoever scatters into it puts the ones
    // back when it reads the results).  0 / 1 = not available.
    int clean_dev(uint8_t** dev, size_t bytes)
    {
        if (ar->cleanCap < bytes || ar->cleanDirty) {
            if (ar->cleanCap < bytes) {
                if (ar->cleanDev) {
                    (void)hipStreamSynchronize(g_ms);
                    (void)hipFree(ar->cleanDev);
                    ar->cleanDev = nullptr;
                    ar->cleanCap = 0;
                }
                void* p = nullptr;
                const size_t want = std::max<size_t>(2 * bytes, 64u << 10);
                if (hipMalloc(&p, want) != hipSuccess) {
                    (void)hipGetLastError();
                    return 1;
                }
                ar->cleanDev = (uint8_t*)p;
                ar->cleanCap = want;
            }
            if (hipMemsetAsync(ar->cleanDev, 0xFF, ar->cleanCap, g_ms) != hipSuccess) { // (on the call's own stream)
                (void)hipGetLastError();
                return 1;
            }
            ar->cleanDirty = false;
        }
        *dev = ar->cleanDev;
        return 0;
    }
    // 0: *ob describes where the kernel puts `bytes` of results (all ones to begin with) and where the host finds them;
    // 1: not available (the caller downloads as before).  `wgs` = workgroups of the kernel: which of the two forms is used
    // (ORBFE_MATCHER_BLOCK: 0 = always straight into the mirror -- the default: on one box the three policies were within
    // 1 us of each other for the 12 000-row triangulation batch, and the block cost the small calls 2-9 us (the copy is
    // one wavefront's work behind the last workgroup) --, 2 = always through the clean block, 1 = the block for grids too
    // large for the in-kernel completion word)
    int out_block(OutBlock* ob, size_t bytes, unsigned wgs)
    {
        bytes = (bytes + 15) & ~(size_t)15;
        if (!ar->pinCoherent) return 1;
        static const int policy = [] {
            const char* e = getenv("ORBFE_MATCHER_BLOCK");
            return e ? atoi(e) : 0;
        }();
        if (policy == 0 || (policy == 1 && wgs <= 256u)) {
            uint8_t *md = nullptr, *mh = nullptr;
            if (mirror_out(&md, &mh, bytes) != 0) return 1;
            std::memset(mh, 0xFF, bytes);
            ob->dev = md;
            ob->host = mh;
            ob->mirrorDev = nullptr;
            ob->bytes = bytes;
            ob->direct = true;
            return 0;
        }
        uint8_t* cd = nullptr;
        if (clean_dev(&cd, bytes) != 0) return 1;
        uint8_t *md = nullptr, *mh = nullptr;
        if (mirror_out(&md, &mh, bytes) != 0) return 1;
        ob->dev = cd;
        ob->host = mh;
        ob->mirrorDev = reinterpret_cast<uint4*>(md);
        ob->bytes = bytes;
        return 0;
    }
    static bool spin_enabled()
    {
        static const bool enabled = [] {
            const char* e = getenv("ORBFE_MATCHER_SPIN");
            if (!e) e = getenv("ORBFE_SPIN"); // (the extractor's switch for the same mechanism)
            return !(e && e[0] == '0');
        }();
        return enabled;
    }
    bool done_words()
    {
        // (eight waits in a row in which the word NEVER arrived -- not merely late, see complete() -- switch it off for
        // this thread; one call in 256 still carries it, so a thread that lost it on a loaded GPU gets it back)
        if (ar->spinMisses >= 8 && (++ar->spinProbe & 255u) != 0u) return false;
        if (ar->doneCtr) return true;
        void *c = nullptr, *h = nullptr, *dv = nullptr;
        // (cleared on the call's own stream: the null stream is not ordered with a non-blocking one)
        if (hipMalloc(&c, 64) != hipSuccess || hipMemsetAsync(c, 0, 64, g_ms) != hipSuccess ||
            hipHostMalloc(&h, 64, hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (c) (void)hipFree(c);
            if (h) (void)hipHostFree(h);
            return false;
        }
        ar->doneCtr = (unsigned*)c;
        ar->doneFlag = (unsigned*)h;
        ar->doneFlagDev = (unsigned*)dv;
        *ar->doneFlag = 0u;
        return true;
    }
    // The main kernel's completion record.  With a flag (ctr / flag / seq set): the kernel counts its workgroups and the last
    // one copies the block and publishes -- small grids whose inputs are read in place (a copy command queued behind a kernel
    // the runtime still holds as running came out slower) and nobody timing the kernel.  Otherwise only the block's addresses
    // are filled in (the kernel scatters into it) and complete() queues k_copy_out behind the kernel.
    DoneSig done_sig(unsigned waves /* the kernel runs four per workgroup */, const OutBlock* ob, bool timed)
    {
        DoneSig d{nullptr, nullptr, 0u, (waves + 3u) / 4u, waves, nullptr, nullptr, 0u};
        if (!ob || !ob->dev) return d;
        if (!ob->direct) {
            d.outDev = reinterpret_cast<uint4*>(ob->dev);
            d.outMirror = ob->mirrorDev;
            d.out16 = (unsigned)(ob->bytes >> 4);
        }
        if (!spin_enabled() || timed || waves == 0 || d.total > 256u || !inPlace || !done_words()) return d;
        if (++ar->doneSeq == 0u) ar->doneSeq = 1u;
        d.ctr = ar->doneCtr;
        d.flag = ar->doneFlagDev;
        d.seq = ar->doneSeq;
        return d;
    }
    // a word without a block or a counter: for a kernel whose one workgroup writes the mirror itself (K-PROJ's sweeps)
    DoneSig flag_only()
    {
        DoneSig d{nullptr, nullptr, 0u, 1u, 1u, nullptr, nullptr, 0u};
        if (!spin_enabled() || !inPlace || !ar->pinCoherent || !done_words()) return d;
        if (++ar->doneSeq == 0u) ar->doneSeq = 1u;
        d.ctr = ar->doneCtr;
        d.flag = ar->doneFlagDev;
        d.seq = ar->doneSeq;
        return d;
    }
    // a word for a small final kernel of `wgs` workgroups that count themselves (no block: the kernel writes the mirror)
    DoneSig word_for(unsigned wgs)
    {
        DoneSig d{nullptr, nullptr, 0u, wgs, wgs, nullptr, nullptr, 0u};
        if (!spin_enabled() || !ar->pinCoherent || wgs > 256u || !done_words()) return d;
        if (++ar->doneSeq == 0u) ar->doneSeq = 1u;
        d.ctr = ar->doneCtr;
        d.flag = ar->doneFlagDev;
        d.seq = ar->doneSeq;
        return d;
    }
    // after the main kernel has been launched: the block's mirror is complete when this returns
    int complete(const DoneSig& d)
    {
        DoneSig w = d;
        if (!d.flag && d.outDev) { // the copy as a kernel of its own (it publishes the flag when the word may be used)
            const unsigned wgs = std::min(32u, std::max(1u, d.out16 / 1024u));
            w.total = wgs;
            if (spin_enabled() && done_words()) {
                if (++ar->doneSeq == 0u) ar->doneSeq = 1u;
                w.ctr = ar->doneCtr;
                w.flag = ar->doneFlagDev;
                w.seq = ar->doneSeq;
            }
            hipLaunchKernelGGL(k_copy_out, dim3(wgs), dim3(256), 0, g_ms, w);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                ar->cleanDirty = true;
                return -(1000 + (int)le);
            }
        }
        if (w.flag) {
            const volatile unsigned* f = ar->doneFlag;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; it++) {
                if (*f == w.seq) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    ar->spinMisses = 0;
                    return 0;
                }
                __builtin_ia32_pause();
                if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) break;
            }
        }
        hipError_t e = hipStreamSynchronize(g_ms);
        // a miss is a word that has still not arrived when the stream is idle; a word that came after the bound belongs to a long
        // call, a first call (code object load) or a kernel queued behind somebody else's work, and says nothing about the platform
        if (w.flag) ar->spinMisses = (*(const volatile unsigned*)ar->doneFlag == w.seq) ? 0 : ar->spinMisses + 1;
        // the word did not come within the bound: normally a long call (its counter is back at zero by now); should the counter
        // ever be left non-zero -- a kernel that died half-way -- every later call would time out, so it is cleared here
        if (e == hipSuccess && w.flag) e = hipMemsetAsync(ar->doneCtr, 0, 64, g_ms); // (all its words: K-PROJ keeps a counter there too)
        if (e != hipSuccess) ar->cleanDirty = true; // (the block may hold half a call's results)
        return e == hipSuccess ? 0 : -(1000 + (int)e);
    }
};

thread_local float g_lastKernelMs = -1.f;
thread_local int g_lastProjSweeps = 0;
thread_local bool g_timeKernels = false; // orbfe_matcher_time_kernels(): bench / tests only
// Brackets the kernel launches of one call: uploads the staged inputs first; with timing enabled also measures the
// launches with events (two event creations and a synchronisation per call, so off by default).
struct KernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    explicit KernelTimer(Scratch& s)
    {
        (void)s.flush();
        if (g_timeKernels) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, g_ms);
        }
    }
    ~KernelTimer()
    {
        if (!a) return;
        (void)hipEventRecord(b, g_ms);
        (void)hipEventSynchronize(b);
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess) g_lastKernelMs = ms;
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
};

int select_device(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return ORBFE_ERR_NODEV;
    if (device >= kMaxDevices) return ORBFE_ERR_ARGS; // one scratch arena per (thread, device ordinal < 16)
    HIP_TRY(hipSetDevice(device));
    return 0;
}

bool fv_ok(const orbfe_fv& f)
{
    if (f.nn == ORBFE_FV_RESIDENT) return f.node_ids != nullptr; // names an orbfe_bow handle (fv_resolve / bow_run)
    if (f.nn < 0) return false;
    if (f.nn > 0 && (!f.node_ids || !f.offsets)) return false;
    // A FeatureVector is a std::map<NodeId, ...> (Thirdparty/DBoW2/DBoW2/FeatureVector.h:27): its ids come strictly ascending
    // and therefore unique.  The merge-join, the binary searches and the in-kernel pairing (one ballot over 64 ids of set 2 per
    // step finds THE partner of a node) all rely on it, so it is checked, not assumed; ~100 ids per vector.
    for (int i = 1; i < f.nn; i++)
        if (f.node_ids[i] <= f.node_ids[i - 1]) return false;
    for (int i = 0; i < f.nn; i++)
        if (f.offsets[i] < 0 || f.offsets[i + 1] < f.offsets[i]) return false;
    return true;
}

// merge-join of two ascending node-id lists (std::map iteration + lower_bound, :285-448)
template <class F>
void for_each_shared_node(const orbfe_fv& a, const orbfe_fv& b, F f)
{
    int i = 0, j = 0;
    while (i < a.nn && j < b.nn) {
        if (a.node_ids[i] == b.node_ids[j]) {
            f(i, j);
            i++;
            j++;
        } else if (a.node_ids[i] < b.node_ids[j]) {
            i = (int)(std::lower_bound(a.node_ids + i, a.node_ids + a.nn, b.node_ids[j]) - a.node_ids);
        } else {
            j = (int)(std::lower_bound(b.node_ids + j, b.node_ids + b.nn, a.node_ids[i]) - b.node_ids);
        }
    }
}

// ComputeThreeMaxima, src/ORBmatcher.cc:2545-2586
void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

// The rotation-consistency cull (:450-468) on the device for a batch whose results come back by a download command (dozens of problems: the host's two
// passes over every problem's match array were 45 of a 64-candidate call's 180 us): a workgroup per problem builds the
// histogram of the rotation bins of its matches, takes the three maxima, clears the matches outside them in place and leaves
// the number kept in nm[problem].  cull_by_rotation() below is the statement it follows line by line.
struct BowCull {
    int outBase, n, check, pad;
};
// Mout / nmOut (round 5): the culled rows and the counts written to a second place as well -- the call's pinned mirror, whole
// rows of consecutive 4-byte stores per wavefront -- so that no download command (and no ~9 us of queue hand-over in front of it)
// follows the kernel; null: in place only, the caller downloads M and nm.
__global__ __launch_bounds__(256) void k_bow_cull(const BowCull* __restrict__ C, int32_t* __restrict__ M, const int8_t* __restrict__ B,
                                                  int32_t* __restrict__ nm, int32_t* __restrict__ Mout, int32_t* __restrict__ nmOut)
{
    __shared__ int sHist[32], sInd[3], sCnt;
    const BowCull c = C[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < 32) sHist[tid] = 0;
    if (tid == 0) sCnt = 0;
    __syncthreads();
    int32_t* const m = M + c.outBase;
    const int8_t* const b = B + c.outBase;
    int cnt = 0;
    for (int i = tid; i < c.n; i += 256)
        if (m[i] >= 0) {
            cnt++;
            const int bin = b[i];
            if (c.check && bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&sHist[bin], 1);
        }
    if (c.check) {
        __syncthreads();
        if (tid == 0) three_maxima_dev(sHist, HISTO_LENGTH, sInd);
        __syncthreads();
        const int ind1 = sInd[0], ind2 = sInd[1], ind3 = sInd[2];
        cnt = 0;
        for (int i = tid; i < c.n; i += 256) {
            int v = m[i];
            if (v >= 0) {
                const int bin = b[i];
                if (bin == ind1 || bin == ind2 || bin == ind3) cnt++;
                else m[i] = v = -1;
            }
            if (Mout) Mout[c.outBase + i] = v;
        }
    } else if (Mout) {
        for (int i = tid; i < c.n; i += 256) Mout[c.outBase + i] = m[i];
    }
    cnt = wave_sum_i32(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&sCnt, cnt);
    __syncthreads();
    if (tid == 0) {
        nm[blockIdx.x] = sCnt;
        if (nmOut) nmOut[blockIdx.x] = sCnt;
    }
}

// rotation-consistency cull (:450-468): returns the number of surviving matches
int cull_by_rotation(int32_t* match, const int8_t* bins, int n, bool check)
{
    int nmatches = 0;
    int histo[HISTO_LENGTH] = {0};
    for (int i = 0; i < n; i++)
        if (match[i] >= 0) {
            nmatches++;
            if (check && bins[i] >= 0 && bins[i] < HISTO_LENGTH) histo[bins[i]]++;
        }
    if (!check) return nmatches;
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < n; i++)
        if (match[i] >= 0) {
            const int b = bins[i];
            if (b == ind1 || b == ind2 || b == ind3) continue;
            match[i] = -1;
            nmatches--;
        }
    return nmatches;
}

} // namespace

extern "C" {

int orbfe_hamming_pairs(int device, const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || (nA && nB && !D)) return ORBFE_ERR_ARGS;
    if (nA == 0 || nB == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t *dA, *dB;
    uint16_t* dD;
    if ((r = s.up_desc(&dA, A, (size_t)nA * 32)) < 0) return r;
    if ((r = s.up_desc(&dB, B, (size_t)nB * 32)) < 0) return r;
    if ((r = s.up<uint16_t>(&dD, nullptr, (size_t)nA * nB)) < 0) return r;
    {
        KernelTimer timer(s);
    hipLaunchKernelGGL(k_hamming_pairs, dim3((unsigned)((nB + 63) / 64), (unsigned)((nA + 63) / 64)), dim3(256), 0, g_ms, dA,
                       nA, dB, nB, dD);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(D, dD, (size_t)nA * nB * sizeof(uint16_t)));
    INT_TRY(s.fetch());
    return 0;
}

int orbfe_bfknn2(int device, const uint8_t* Q, int nQ, const uint8_t* T, int nT, int32_t* idx, int32_t* dist)
{
    if (nQ < 0 || nT < 0 || (nQ && (!Q || !idx || !dist)) || (nT && !T) || nT >= (1 << 20)) return ORBFE_ERR_ARGS;
    if (nQ == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t *dQ, *dT;
    int32_t *dI, *dD;
    if ((r = s.up_desc(&dQ, Q, (size_t)nQ * 32)) < 0) return r;
    if ((r = s.up_desc(&dT, T, (size_t)nT * 32)) < 0) return r;
    if ((r = s.up<int32_t>(&dI, nullptr, (size_t)nQ * 2)) < 0) return r;
    if ((r = s.up<int32_t>(&dD, nullptr, (size_t)nQ * 2)) < 0) return r;
    {
        KernelTimer timer(s);
    hipLaunchKernelGGL(k_bfknn2, dim3((unsigned)((nQ + 3) / 4)), dim3(256), 0, g_ms, dQ, nQ, dT, nT, dI, dD);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(idx, dI, (size_t)nQ * 2 * sizeof(int32_t)));
    INT_TRY(s.down(dist, dD, (size_t)nQ * 2 * sizeof(int32_t)));
    INT_TRY(s.fetch());
    return 0;
}

// ---- device-resident forms: every pointer is device memory, nothing is copied, nothing is waited for ----
static hipStream_t matcher_stream(int device, void* hip_stream)
{
    if (hip_stream) return (hipStream_t)hip_stream;
    Scratch s(device); // makes sure the calling thread's matcher stream exists
    return g_ms;
}

int orbfe_hamming_pairs_device(int device, void* hip_stream, const uint8_t* dA, int nA, const uint8_t* dB, int nB,
                               uint16_t* dD)
{
    if (nA < 0 || nB < 0 || (nA && !dA) || (nB && !dB) || (nA && nB && !dD)) return ORBFE_ERR_ARGS;
    if (nA == 0 || nB == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    if (int w = orbfe_producer_wait(dA, st); w < 0) return w;
    if (int w = orbfe_producer_wait(dB, st); w < 0) return w;
    hipLaunchKernelGGL(k_hamming_pairs, dim3((unsigned)((nB + 63) / 64), (unsigned)((nA + 63) / 64)), dim3(256), 0, st, dA, nA, dB, nB,
                       dD);
    HIP_TRY(hipGetLastError());
    return 0;
}

int orbfe_bfknn2_device(int device, void* hip_stream, const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int32_t* d_idx,
                        int32_t* d_dist)
{
    if (nQ < 0 || nT < 0 || (nQ && (!dQ || !d_idx || !d_dist)) || (nT && !dT) || nT >= (1 << 20)) return ORBFE_ERR_ARGS;
    if (nQ == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    if (int w = orbfe_producer_wait(dQ, st); w < 0) return w;
    if (int w = orbfe_producer_wait(dT, st); w < 0) return w;
    hipLaunchKernelGGL(k_bfknn2, dim3((unsigned)((nQ + 3) / 4)), dim3(256), 0, st, dQ, nQ, dT, nT, d_idx, d_dist);
    HIP_TRY(hipGetLastError());
    return 0;
}

// flags: bit 0 = the caller knows of other kernels in flight on the device (orbfe_mc_match_ring_async with extractions queued);
// bit 1 = rows between a query frame's count and `cap` are written too, as -1 | -1
extern "C" int orbfe_internal_bfknn2_frames(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                                            int32_t* d_idx, int32_t* d_dist, int flags);

int orbfe_bfknn2_frames_device(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                               int32_t* d_idx, int32_t* d_dist)
{
    return orbfe_internal_bfknn2_frames(device, hip_stream, d_jobs, njobs, cap, d_idx, d_dist, 0);
}

int orbfe_internal_bfknn2_frames(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                                 int32_t* d_idx, int32_t* d_dist, int flags)
{
    const int shared = flags & 1, fillTail = (flags >> 1) & 1;
    if (njobs < 0 || cap < 1 || cap >= (1 << 20) || (njobs && (!d_jobs || !d_idx || !d_dist))) return ORBFE_ERR_ARGS;
    if (njobs == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    // the job records live on the device, so the frames they name cannot be looked up one by one: this stream waits for
    // every extraction whose outputs were handed out (orbfe_get_device_outputs) -- a few events, fired long ago as a rule
    if ((r = orbfe_producer_wait_all(st)) < 0) return r;
    // The matrix-pipe form (k_bfknn2_frames_mfma: exact, keys of 11 index bits) for frames of up to 2048 keypoints;
    // ORBFE_KNN2_MFMA=0 keeps the vector-pipe kernel (A/B, and the form for larger frames)
    static const bool mfma = [] {
        const char* e = getenv("ORBFE_KNN2_MFMA");
        return !(e && e[0] == '0');
    }();
    if (mfma && cap <= 2048) {
        size_t lds = 3 * KNN2M_TILE + (size_t)cap * 32; // three expanded tiles + the job's packed train rows (<= 90 KB)
        const dim3 mgrid((unsigned)((cap + KNN2M_QUERIES - 1) / KNN2M_QUERIES), (unsigned)njobs);
        // A grid that fits the chip once (64 jobs x 4 query blocks = 256 workgroups on 256 CUs) must not be packed two to a CU
        // with the rest of the chip idle -- the dispatcher does exactly that when two fit: 26.0 us per launch against 18.0
        // when each asks for more than half a CU's LDS (82 KB and more: 17.9-18.3 us; 81 KB still let two in).  Larger grids
        // keep their real size (two per CU then overlap), and so does a launch that shares the chip with extraction kernels
        // (`shared`): a workgroup that needs 96 KB waits longer for a CU there -- the cross-camera step of bench.py with two
        // extractions in flight: 0.196-0.199 ms padded, 0.190-0.193 not (three runs each).
        // (the last query block of every job does not count when it is the mostly empty one: it is dispatched last)
        const unsigned mainCols = (cap % KNN2M_QUERIES != 0 && mgrid.x > 1) ? mgrid.x - 1 : mgrid.x;
        static const bool pad = !(getenv("ORBFE_KNN2_PAD") && atoi(getenv("ORBFE_KNN2_PAD")) == 0);
        if (pad && !shared && (size_t)mainCols * mgrid.y <= 256) lds = std::max(lds, (size_t)96 * 1024);
        // (the attribute belongs to the CURRENT device's copy of the kernel: one high-water mark per device -- ADVICE r05)
        static std::atomic<size_t> ldsSet[kMaxDevices];
        if (lds > 64 * 1024 && ldsSet[device].load() < lds) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_bfknn2_frames_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ldsSet[device].store(lds);
        }
        hipLaunchKernelGGL(k_bfknn2_frames_mfma, mgrid, dim3(KNN2M_THREADS), lds, st, d_jobs, cap, d_idx, d_dist, fillTail);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid((unsigned)((cap + 63) / 64), (unsigned)njobs);
    // few workgroups: more wavefronts per workgroup share the 64 queries (and fill the chip)
    if ((long)grid.x * njobs >= 2048)
        hipLaunchKernelGGL(k_bfknn2_frames<4>, grid, dim3(256), 0, st, d_jobs, cap, d_idx, d_dist, fillTail);
    else
        hipLaunchKernelGGL(k_bfknn2_frames<8>, grid, dim3(512), 0, st, d_jobs, cap, d_idx, d_dist, fillTail);
    HIP_TRY(hipGetLastError());
    return 0;
}

int orbfe_matcher_sync(int device)
{
    int r;
    if ((r = select_device(device)) < 0) return r;
    if (g_arena[device].stream) HIP_TRY(hipStreamSynchronize(g_arena[device].stream));
    return 0;
}

// A keyframe's matching data kept on the device between calls (round 4): descriptors, the good-MapPoint /
// has-MapPoint flags, angles, keypoints, octaves, mvuRight and the FeatureVector's index array; host copies of what the host
// side of a search reads (the FeatureVector's node ids / offsets / indices for the merge-join, flags, angles, mvuRight).
// orbfe_bow (orbfe_matcher_bowvec.hip): a FeatureVector that lives on the device
struct orbfe_bow;
namespace {
struct BowResident {
    const uint32_t* nodeIds;
    const int32_t *offsets, *indices, *hdr; // hdr[1] = number of nodes
    hipEvent_t ready;                       // behind the kernels that wrote them
    int n, device;
};
int bow_resident(orbfe_bow*, BowResident*);   // takes a use of the handle (bow_release gives it back)
int bow_host_fv(orbfe_bow*, orbfe_fv* host);  // waits for the host copy; takes no use: valid until the next orbfe_compute_bow
void bow_release(orbfe_bow*);
// An orbfe_fv that names a handle, replaced by the handle's host copy (every consumer but the in-kernel pairing of bow_run)
int fv_resolve(orbfe_fv* f)
{
    if (f->nn != ORBFE_FV_RESIDENT) return 0;
    return bow_host_fv(reinterpret_cast<orbfe_bow*>(const_cast<uint32_t*>(f->node_ids)), f);
}
struct BowHold { // the uses bow_run took, given back on every way out
    std::vector<orbfe_bow*> v;
    ~BowHold()
    {
        for (orbfe_bow* b : v) bow_release(b);
    }
};
} // namespace

struct orbfe_keyframe {
    int device = 0, n = 0;
    uint8_t* block = nullptr; // one allocation: everything below points into it
    size_t blockCap = 0;
    uint8_t *desc = nullptr, *mask = nullptr;
    float *ang = nullptr, *kp = nullptr, *uR = nullptr;
    int32_t *oct = nullptr, *ind = nullptr;
    uint32_t* dNode = nullptr; // the FeatureVector's node ids and offsets (nn, nn + 1 entries) for launches that pair the nodes
    int32_t* dOffs = nullptr;  // of two vectors themselves (bow_run, round 5)
    int maxNode = 0;           // features of the largest node
    bool hasTri = false; // keypoints / octaves / mvuRight were given: usable as a side of SearchForTriangulation_
    std::vector<uint32_t> nodeIds;
    std::vector<int32_t> offsets, indices, hOct;
    std::vector<uint8_t> hMask;
    std::vector<float> hAng, hUR;
    int octMin = 0, octMax = -1; // range of hOct (the triangulation search checks it against the caller's level tables per call)
    orbfe_fv fv() const
    {
        orbfe_fv f;
        f.nn = (int)nodeIds.size();
        f.node_ids = nodeIds.data();
        f.offsets = offsets.data();
        f.indices = indices.data();
        return f;
    }
};

namespace {
// SearchByBoW over `count` problems; kf1 / kf2 (arrays or null, entries may be null) name sets that live in handles
// staged inputs up to this size are read by the kernel from the pinned staging in place (ORBFE_MATCHER_INPLACE_KB, default 128: 85 KB of a host-array SearchByBoW read in place took 0.037 instead of 0.045 ms)
static size_t inplace_limit()
{
    static const size_t v = [] {
        const char* e = getenv("ORBFE_MATCHER_INPLACE_KB");
        const long kb = e ? atol(e) : 128;
        return (size_t)(kb < 0 ? 0 : kb) << 10;
    }();
    return v;
}
// tuning: ORBFE_BOW_STOP=n cuts K-BOW short after its n-th stage (wrong results; where the time of a call goes)
static int bow_stop_at()
{
    static const int v = [] {
        const char* e = getenv("ORBFE_BOW_STOP");
        return e ? atoi(e) : 0;
    }();
    return v;
}
// tuning only (tools/ab_build.sh trace "-DORBFE_CALL_TRACE", ORBFE_CALL_TRACE=1 in the environment): where the host time of a
// matcher call goes, printed per call
#ifdef ORBFE_CALL_TRACE
#define PTR_BEGIN()                                   \
    auto tr0 = std::chrono::steady_clock::now();      \
    double trT[8] = {0};                              \
    int trK = 0
#define PTR() do { auto n_ = std::chrono::steady_clock::now(); trT[trK++] = std::chrono::duration<double, std::micro>(n_ - tr0).count(); tr0 = n_; } while (0)
#else
#define PTR_BEGIN() do { } while (0)
#define PTR() do { } while (0)
#endif
int bow_run(int device, int count, const orbfe_bow_args* args, orbfe_keyframe* const* kf1, orbfe_keyframe* const* kf2,
            int32_t* const* match, int* nmatches)
{
    if (count < 0 || (count && (!args || !match || !nmatches))) return ORBFE_ERR_ARGS;
    PTR_BEGIN();
    // pass 1: validate, lay the pools out, list the shared vocabulary nodes (merge-join of the two FeatureVectors)
    // (the node list of a 64-candidate call is 128 KB, its download 256 KB: as fresh vectors they are mmap'ed, faulted in and
    // unmapped by every call; the thread keeps them)
    static thread_local std::vector<BowNode> nodesKeep;
    static thread_local std::vector<int32_t> downKeep;
    std::vector<BowNode>& nodes = nodesKeep;
    nodes.clear();
    std::vector<BowProb> probs(count);
    std::vector<int> outN(count), i1Base(count, 0), i2Base(count, 0);
    std::vector<uint8_t> active(count, 0);
    std::vector<orbfe_bow_args> eff(count); // the arguments with the handles' host views filled in
    int rows = 0, outTotal = 0, takenRows = 0;
    size_t indTotal = 0, ovTotal = 0;
    bool needTakenDev = false;
    std::vector<long> ovOff1(count, -1), ovOff2(count, -1); // per-call flags of sets in handles: offsets into their own pool
    // A set that is not in a handle travels with the call -- once: the problems of a call usually share one side (the current
    // frame against every relocalisation candidate, src/Tracking.cc:3784; the current keyframe against its covisibles), and the
    // same arrays (same pointers, same sizes) are staged and uploaded a single time (64 candidates: 1.3 MB -> 41 KB).
    struct SeenSet {
        const void *desc, *mask, *ang, *ind, *offs, *ids;
        int n, nn, rowBase, indBase, nodeBase /* in the pooled node ids; offsets: nodeBase + index of the set */, maxNode;
    };
    std::vector<SeenSet> seen;
    // Round 6: a set whose FeatureVector is resident (orbfe_bow_fv).  With the nodes paired in the kernel the vector is read where
    // orbfe_compute_bow left it; otherwise the handle's host copy takes its place (a wait for a copy that was queued with it).
    std::vector<BowResident> res1(count), res2(count);
    std::vector<uint8_t> isRes1(count, 0), isRes2(count, 0), inKf1(count, 0), inKf2(count, 0);
    BowHold hold;
    std::vector<uint8_t> own1(count, 0), own2(count, 0); // this problem stages the set (first occurrence)
    std::vector<int> set1(count, -1), set2(count, -1);  // index into `seen` of a pooled side
    size_t nodeTotal = 0;
    // Round 5: a call whose results are downloaded anyway (more than 256 KB of them: the 64 candidates of a
    // relocalisation) leaves the merge-join of the FeatureVectors to the kernel -- 33 us of host time and a 128-KB node list per
    // call of 64; the node ids / offsets of sets that are not in handles travel in the pool (~1 KB per set).
    // (Calls whose results come back through the pinned mirror keep the host list: their launch counts its workgroups for the
    // completion word.)  ORBFE_BOW_DEVNODES=0: host lists always (A/B).
    static const int devNodesPolicy = [] {
        const char* e = getenv("ORBFE_BOW_DEVNODES");
        return e ? atoi(e) : 1;
    }();
    bool devNodes = false;
    if (devNodesPolicy != 0) {
        size_t outPre = 0;
        for (int p = 0; p < count; p++) {
            const orbfe_keyframe* K1 = kf1 ? kf1[p] : nullptr;
            const orbfe_keyframe* K2 = kf2 ? kf2[p] : nullptr;
            const int v = args[p].variant;
            const int n = v == 0 ? (K2 ? K2->n : args[p].n2) : (K1 ? K1->n : args[p].n1);
            outPre += (size_t)std::max(n, 0);
        }
        devNodes = outPre * 5 > (256u << 10) && count <= 65535; // (the same test as `mirrored` below; problems = grid rows)
    }
    int maxNN1 = 0;
    auto place_set = [&](const uint8_t* desc, int n, const uint8_t* mask, const float* ang, const orbfe_fv& fv, int& rowBase, int& indBase) -> int {
        for (size_t k = 0; k < seen.size(); k++) {
            const SeenSet& q = seen[k];
            if (q.desc == desc && q.n == n && q.mask == mask && q.ang == ang && q.ind == fv.indices && q.offs == fv.offsets && q.nn == fv.nn &&
                q.ids == fv.node_ids) {
                rowBase = q.rowBase;
                indBase = q.indBase;
                return -(int)k - 1; // (seen before)
            }
        }
        rowBase = rows;
        indBase = (int)indTotal;
        int mx = 0;
        for (int i = 0; i < fv.nn; i++) mx = std::max(mx, fv.offsets[i + 1] - fv.offsets[i]);
        seen.push_back(SeenSet{desc, mask, ang, fv.indices, fv.offsets, fv.node_ids, n, fv.nn, rowBase, indBase, (int)nodeTotal, mx});
        rows += n;
        indTotal += (size_t)(fv.nn ? fv.offsets[fv.nn] : 0);
        nodeTotal += (size_t)fv.nn;
        return (int)seen.size();
    };
    for (int p = 0; p < count; p++) {
        orbfe_bow_args& e = eff[p];
        e = args[p];
        const orbfe_keyframe* K1 = kf1 ? kf1[p] : nullptr;
        const orbfe_keyframe* K2 = kf2 ? kf2[p] : nullptr;
        if ((K1 && K1->device != device) || (K2 && K2->device != device)) return ORBFE_ERR_ARGS;
        // (a set in a handle: its arrays come from the handle; the flags alone may be given per call -- args[p].mask1 / mask2
        // non-null --, because a keyframe's MapPoints change while several threads search it: they then travel with the call
        // instead of being written into the shared handle)
        const uint8_t* ov1 = K1 ? args[p].mask1 : nullptr;
        const uint8_t* ov2 = K2 && args[p].variant == 1 ? args[p].mask2 : nullptr;
        if (K1) {
            e.desc1 = K1->desc; e.n1 = K1->n; e.mask1 = ov1 ? ov1 : K1->hMask.data();
            e.angle1 = K1->hAng.empty() ? nullptr : K1->hAng.data();
            e.fv1 = K1->fv();
        }
        if (K2) {
            e.desc2 = K2->desc; e.n2 = K2->n; e.mask2 = ov2 ? ov2 : K2->hMask.data();
            e.angle2 = K2->hAng.empty() ? nullptr : K2->hAng.data();
            e.fv2 = K2->fv();
        }
        for (int side = 0; side < 2; side++) {
            orbfe_fv& f = side ? e.fv2 : e.fv1;
            if ((side ? K2 : K1) || f.nn != ORBFE_FV_RESIDENT) continue;
            if (!f.node_ids) return ORBFE_ERR_ARGS;
            orbfe_bow* B = reinterpret_cast<orbfe_bow*>(const_cast<uint32_t*>(f.node_ids));
            if (!devNodes) { // host lists: the handle's host copy
                const int rr = bow_host_fv(B, &f);
                if (rr < 0) return rr;
                continue;
            }
            BowResident& R = side ? res2[p] : res1[p];
            const int rr = bow_resident(B, &R);
            if (rr < 0) return rr;
            hold.v.push_back(B);
            if (R.device != device || R.n != (side ? e.n2 : e.n1)) return ORBFE_ERR_ARGS; // (the vector indexes THIS set's features)
            (side ? isRes2 : isRes1)[p] = 1;
            f.nn = 0; // (for the pooled layout below: the set brings no node list and no index array of its own)
            f.node_ids = nullptr;
            f.offsets = f.indices = nullptr;
        }
        inKf1[p] = K1 ? 1 : 0;
        inKf2[p] = K2 ? 1 : 0;
        ovOff1[p] = ov1 ? (long)ovTotal : -1;
        if (ov1) ovTotal += ((size_t)K1->n + 63) & ~(size_t)63;
        ovOff2[p] = ov2 ? (long)ovTotal : -1;
        if (ov2) ovTotal += ((size_t)K2->n + 63) & ~(size_t)63;
        const orbfe_bow_args* a = &e;
        if (!match[p] || a->n1 < 0 || a->n2 < 0 || !fv_ok(a->fv1) || !fv_ok(a->fv2) ||
            (a->variant != 0 && a->variant != 1))
            return ORBFE_ERR_ARGS;
        const int nOut = a->variant == 0 ? a->n2 : a->n1;
        outN[p] = nOut;
        nmatches[p] = 0;
        BowProb& P = probs[p];
        std::memset(&P, 0, sizeof P);
        P.outBase = outTotal;
        P.tBase = takenRows;
        P.limit1 = a->limit1;
        P.limit2 = a->limit2;
        P.Nleft = a->Nleft;
        P.variant = a->variant;
        P.nnratio = a->nnratio;
        outTotal += nOut;
        takenRows += a->n2;
        if (a->n1 == 0 || a->n2 == 0) continue;
        if (!a->desc1 || !a->desc2 || !a->mask1 || (a->variant == 1 && !a->mask2)) return ORBFE_ERR_ARGS;
        if (a->check_orientation && (!a->angle1 || !a->angle2)) return ORBFE_ERR_ARGS;
        active[p] = 1;
        P.d1Base = P.d2Base = rows; // (not read for a set in a handle)
        if (K1) {
            P.rDesc1 = K1->desc; P.rMask1 = K1->mask; P.rAng1 = K1->ang; P.rInd1 = K1->ind;
        } else {
            if (is_device_ptr(a->desc1)) P.rDesc1 = a->desc1; // read where the extractor left them
            const int k = place_set(a->desc1, a->n1, a->mask1, a->angle1, a->fv1, P.d1Base, i1Base[p]);
            own1[p] = k > 0 ? 1 : 0;
            set1[p] = k > 0 ? k - 1 : -k - 1;
        }
        if (K2) {
            P.rDesc2 = K2->desc; P.rMask2 = K2->mask; P.rAng2 = K2->ang; P.rInd2 = K2->ind;
        } else {
            if (is_device_ptr(a->desc2)) P.rDesc2 = a->desc2;
            // (the flags of set 2 are all ones in variant 0: such a set and one with real flags are different sets)
            const int k = place_set(a->desc2, a->n2, a->variant == 1 ? a->mask2 : nullptr, a->angle2, a->fv2, P.d2Base, i2Base[p]);
            own2[p] = k > 0 ? 1 : 0;
            set2[p] = k > 0 ? k - 1 : -k - 1;
        }
        bool bad = false;
        const int b1 = i1Base[p], b2 = i2Base[p]; // (0 for a set in a handle: its offsets are relative to its own index array)
        if (devNodes) { // (the kernel pairs the nodes: what the host still checks is the size of set 2's largest node)
            // (a resident vector's largest node is not known here: every feature of the set at most)
            const int mx2 = K2 ? K2->maxNode : isRes2[p] ? a->n2 : seen[(size_t)set2[p]].maxNode;
            if (mx2 >= (1 << 20)) return ORBFE_ERR_ARGS;
            needTakenDev = needTakenDev || mx2 > 4096;
            P.nn1 = isRes1[p] ? a->n1 : a->fv1.nn; // (resident: an upper bound for the grid; the kernel reads the count, dnn1)
            P.nn2 = a->fv2.nn;
            P.i1Base = b1;
            P.i2Base = b2;
            if (K1) { P.node1 = K1->dNode; P.offs1 = K1->dOffs; }
            if (K2) { P.node2 = K2->dNode; P.offs2 = K2->dOffs; }
            maxNN1 = std::max(maxNN1, P.nn1);
            continue;
        }
        for_each_shared_node(a->fv1, a->fv2, [&](int i, int j) {
            BowNode n;
            n.off1 = b1 + a->fv1.offsets[i];
            n.n1 = a->fv1.offsets[i + 1] - a->fv1.offsets[i];
            n.off2 = b2 + a->fv2.offsets[j];
            n.n2 = a->fv2.offsets[j + 1] - a->fv2.offsets[j];
            n.prob = p;
            if (n.n2 >= (1 << 20)) bad = true;
            if (n.n1 > 0 && n.n2 > 0) nodes.push_back(n);
        });
        if (bad) return ORBFE_ERR_ARGS;
    }
    // (every array is written whole at the end; the calls that end here have no match anywhere)
    auto none = [&]() {
        for (int p = 0; p < count; p++)
            for (int i = 0; i < outN[p]; i++) match[p][i] = -1;
    };
    if (devNodes ? maxNN1 == 0 : nodes.empty()) {
        none();
        return 0;
    }
    PTR(); // pass 1
    bool needTaken = needTakenDev; // the "taken" flags in memory are only touched by nodes with more than 4096 candidates
    for (const BowNode& nd : nodes) needTaken = needTaken || nd.n2 > 4096;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    for (int p = 0; p < count; p++) { // resident vectors: this stream behind the kernels that wrote them (another thread's, maybe)
        if (isRes1[p]) HIP_TRY(hipStreamWaitEvent(g_ms, res1[p].ready, 0));
        if (isRes2[p] && !(isRes1[p] && res2[p].ready == res1[p].ready) && !(p > 0 && isRes2[p - 1] && res2[p - 1].ready == res2[p].ready))
            HIP_TRY(hipStreamWaitEvent(g_ms, res2[p].ready, 0));
    }
    // (what travels: node list, problem records, the pooled sets.  A search against resident keyframes sends ~15 KB: the kernel
    // reads that from the pinned staging in place)
    // (not when the kernel pairs the nodes: every workgroup then starts with two or three DEPENDENT reads of the problem record
    // and the node ids -- across PCIe that made the 64-candidate kernel 48 us instead of 25; such a call's staged inputs go up
    // through k_stage_in, below)
    s.inPlace = !devNodes && nodes.size() * sizeof(BowNode) + (size_t)rows * 37 + indTotal * 4 + ovTotal <= inplace_limit();
    BowNode* dN;
    BowProb* dP;
    uint8_t *dDesc, *dMask, *taken, *hDesc, *hMask;
    float *dAng, *hAng;
    int32_t *dInd, *dM, *hInd;
    int8_t* dB;
    if (devNodes) dN = nullptr;
    else if ((r = s.up(&dN, nodes.data(), nodes.size())) < 0) return r;
    if (devNodes && !seen.empty()) { // node ids and offsets of the pooled sets (set k: ids at nodeBase, offsets at nodeBase + k)
        uint32_t *dIds = nullptr, *hIds = nullptr;
        int32_t *dOf = nullptr, *hOf = nullptr;
        if ((r = s.reserve(&dIds, &hIds, nodeTotal)) < 0) return r;
        if ((r = s.reserve(&dOf, &hOf, nodeTotal + seen.size())) < 0) return r;
        for (size_t k = 0; k < seen.size(); k++) {
            const SeenSet& q = seen[k];
            if (q.nn) std::memcpy(hIds + q.nodeBase, q.ids, (size_t)q.nn * sizeof(uint32_t));
            if (q.nn) std::memcpy(hOf + q.nodeBase + k, q.offs, ((size_t)q.nn + 1) * sizeof(int32_t));
            else hOf[q.nodeBase + k] = 0;
        }
        for (int p = 0; p < count; p++) {
            if (!active[p]) continue;
            if (set1[p] >= 0) {
                probs[p].node1 = dIds + seen[(size_t)set1[p]].nodeBase;
                probs[p].offs1 = dOf + seen[(size_t)set1[p]].nodeBase + set1[p];
            }
            if (set2[p] >= 0) {
                probs[p].node2 = dIds + seen[(size_t)set2[p]].nodeBase;
                probs[p].offs2 = dOf + seen[(size_t)set2[p]].nodeBase + set2[p];
            }
        }
    }
    for (int p = 0; devNodes && p < count; p++) { // resident vectors: node ids, offsets, indices and the node count where they lie
        if (!active[p]) continue;
        if (isRes1[p]) {
            probs[p].node1 = res1[p].nodeIds; probs[p].offs1 = res1[p].offsets; probs[p].dnn1 = res1[p].hdr + 1;
            probs[p].rInd1 = res1[p].indices; probs[p].i1Base = 0;
        }
        if (isRes2[p]) {
            probs[p].node2 = res2[p].nodeIds; probs[p].offs2 = res2[p].offsets; probs[p].dnn2 = res2[p].hdr + 1;
            probs[p].rInd2 = res2[p].indices; probs[p].i2Base = 0;
        }
    }
    {
        uint8_t *dOv = nullptr, *hOv = nullptr;
        if (ovTotal) {
            if ((r = s.reserve(&dOv, &hOv, ovTotal)) < 0) return r;
            for (int p = 0; p < count; p++) {
                if (ovOff1[p] >= 0) {
                    std::memcpy(hOv + ovOff1[p], eff[p].mask1, (size_t)eff[p].n1);
                    probs[p].rMask1 = dOv + ovOff1[p];
                }
                if (ovOff2[p] >= 0) {
                    std::memcpy(hOv + ovOff2[p], eff[p].mask2, (size_t)eff[p].n2);
                    probs[p].rMask2 = dOv + ovOff2[p];
                }
            }
        }
    }
    if ((r = s.up(&dP, probs.data(), probs.size())) < 0) return r;
    if ((r = s.reserve(&dDesc, &hDesc, (size_t)rows * 32)) < 0) return r;
    if ((r = s.reserve(&dMask, &hMask, (size_t)rows)) < 0) return r;
    if ((r = s.reserve(&dAng, &hAng, (size_t)rows)) < 0) return r;
    if ((r = s.reserve(&dInd, &hInd, indTotal)) < 0) return r;
    // results: written by the kernel into the pinned mirror when they are small (no download command), else downloaded
    int32_t* hM = nullptr;
    int8_t* hB = nullptr;
    Scratch::OutBlock ob;
    const size_t mBytes = ((size_t)outTotal * sizeof(int32_t) + 15) & ~(size_t)15;
    const bool mirrored = !devNodes && (size_t)outTotal * 5 <= (256u << 10) &&
                          s.out_block(&ob, mBytes + (size_t)outTotal, (unsigned)nodes.size()) == 0;
    if (mirrored) { // (the kernel scatters into the clean device block; its mirror arrives whole: DoneSig)
        dM = reinterpret_cast<int32_t*>(ob.dev);
        dB = reinterpret_cast<int8_t*>(ob.dev + mBytes);
        hM = reinterpret_cast<int32_t*>(ob.host);
        hB = reinterpret_cast<int8_t*>(ob.host + mBytes);
    } else {
        // [matches | kept per problem | bins]: one clearing command, one download (matches + counts; k_bow_cull consumes the bins)
        if ((r = s.up<int32_t>(&dM, nullptr, (size_t)outTotal + (size_t)count + ((size_t)outTotal + 3) / 4)) < 0) return r;
        dB = reinterpret_cast<int8_t*>(dM + outTotal + count);
    }
    if ((r = s.up<uint8_t>(&taken, nullptr, (size_t)takenRows)) < 0) return r;
    // pass 2: every problem's arrays go straight into the pinned mirror of the pools (one copy, no intermediate
    // vectors); descriptor sets that already live on the device are copied device-to-device after the upload; sets in
    // handles are read where they are
    struct D2D {
        size_t off;
        const uint8_t* src;
        size_t bytes;
    };
    std::vector<D2D> d2d;
    for (int p = 0; p < count; p++) {
        if (!active[p]) continue;
        const orbfe_bow_args* a = &eff[p];
        const bool R1 = inKf1[p] != 0, R2 = inKf2[p] != 0; // the whole set lives in a keyframe handle
        const size_t r1 = (size_t)probs[p].d1Base, r2 = (size_t)probs[p].d2Base;
        if (!R1 && own1[p]) {
            if (is_device_ptr(a->desc1)) { // (read in place: BowProb::rDesc1)
                if (int w = orbfe_producer_wait(a->desc1, g_ms); w < 0) return w;
            }
            else std::memcpy(hDesc + r1 * 32, a->desc1, (size_t)a->n1 * 32);
            std::memcpy(hMask + r1, a->mask1, (size_t)a->n1);
            if (a->angle1) std::memcpy(hAng + r1, a->angle1, (size_t)a->n1 * sizeof(float));
            else std::memset(hAng + r1, 0, (size_t)a->n1 * sizeof(float));
            if (a->fv1.nn) std::memcpy(hInd + i1Base[p], a->fv1.indices, (size_t)a->fv1.offsets[a->fv1.nn] * sizeof(int32_t));
        }
        if (!R2 && own2[p]) {
            if (is_device_ptr(a->desc2)) {
                if (int w = orbfe_producer_wait(a->desc2, g_ms); w < 0) return w;
            }
            else std::memcpy(hDesc + r2 * 32, a->desc2, (size_t)a->n2 * 32);
            if (a->variant == 1) std::memcpy(hMask + r2, a->mask2, (size_t)a->n2);
            else std::memset(hMask + r2, 1, (size_t)a->n2);
            if (a->angle2) std::memcpy(hAng + r2, a->angle2, (size_t)a->n2 * sizeof(float));
            else std::memset(hAng + r2, 0, (size_t)a->n2 * sizeof(float));
            if (a->fv2.nn) std::memcpy(hInd + i2Base[p], a->fv2.indices, (size_t)a->fv2.offsets[a->fv2.nn] * sizeof(int32_t));
        }
    }
    BowCull* dC = nullptr;
    if (!mirrored) {
        std::vector<BowCull> cu((size_t)count);
        for (int p = 0; p < count; p++) cu[(size_t)p] = BowCull{probs[p].outBase, outN[p], args[p].check_orientation != 0 ? 1 : 0, 0};
        if ((r = s.up(&dC, cu.data(), cu.size())) < 0) return r;
        const size_t clearBytes = ((size_t)outTotal + (size_t)count) * sizeof(int32_t) + (size_t)outTotal;
        // (the region is 256-byte aligned and rounded: the kernel's whole 16-byte units stay inside it)
        if (!(devNodes && s.flush_by_kernel(dM, clearBytes))) HIP_TRY(hipMemsetAsync(dM, 0xFF, clearBytes, g_ms));
    }
    if (needTaken) HIP_TRY(hipMemsetAsync(taken, 0, (size_t)takenRows, g_ms));
    const DoneSig done = s.done_sig(4u * (unsigned)nodes.size() /* (a workgroup per node) */, mirrored ? &ob : nullptr, g_timeKernels);
    PTR(); // staging
    {
        KernelTimer timer(s); // (sends the staged pools)
        for (const D2D& c : d2d)
            HIP_TRY(hipMemcpyAsync(dDesc + c.off, c.src, c.bytes, hipMemcpyDeviceToDevice, g_ms));
        if (devNodes) // (workgroup (i, p): node i of set 1 of problem p; no completion count: `done` carries no flag here)
            hipLaunchKernelGGL(k_search_bow, dim3((unsigned)maxNN1, (unsigned)count), dim3(256), 0, g_ms, (const BowNode*)nullptr, 0,
                               dP, dDesc, dMask, dAng, dInd, dM, dB, taken, done, bow_stop_at());
        else
        hipLaunchKernelGGL(k_search_bow, dim3((unsigned)nodes.size()), dim3(256), 0, g_ms, dN, (int)nodes.size(),
                           dP, dDesc, dMask, dAng, dInd, dM, dB, taken, done, bow_stop_at());
    }
    HIP_TRY(hipGetLastError());
    int32_t *dMir = nullptr, *hMir = nullptr; // the culled rows + counts in the pinned mirror (written by k_bow_cull)
    if (!mirrored) {
        if (devNodes && s.mirror_out(&dMir, &hMir, (size_t)outTotal + (size_t)count) != 0) dMir = hMir = nullptr;
        hipLaunchKernelGGL(k_bow_cull, dim3((unsigned)count), dim3(256), 0, g_ms, dC, dM, dB, dM + outTotal, dMir,
                           dMir ? dMir + outTotal : nullptr);
        HIP_TRY(hipGetLastError());
    }
    PTR(); // launch
    std::vector<int32_t>& m = downKeep;
    const int32_t* pm;
    const int8_t* pb = nullptr;
    if (mirrored) { // the results arrive in the pinned mirror: wait, read
        INT_TRY(s.complete(done));
        pm = hM;
        pb = hB;
    } else if (hMir) { // written by k_bow_cull: complete when the stream is
        HIP_TRY(hipStreamSynchronize(g_ms));
        pm = hMir;
    } else {
        // (reading the download where it lands in the pinned mirror instead of copying it out first was tried: the copy is a
        // streaming pass, the cull loop on freshly DMA-written lines is not -- 206 against 188 us for wait + tail of a 64-problem call)
        if (m.size() < (size_t)outTotal + (size_t)count) m.resize((size_t)outTotal + (size_t)count);
        INT_TRY(s.down(m.data(), dM, ((size_t)outTotal + (size_t)count) * sizeof(int32_t)));
        INT_TRY(s.fetch());
        pm = m.data();
    }
    PTR(); // wait
    for (int p = 0; p < count; p++) {
        std::memcpy(match[p], pm + probs[p].outBase, (size_t)outN[p] * sizeof(int32_t));
        if (pb) {
            nmatches[p] = cull_by_rotation(match[p], pb + probs[p].outBase, outN[p], args[p].check_orientation != 0);
        } else { // (culled and counted by k_bow_cull)
            const int kept = pm[(size_t)outTotal + (size_t)p];
            if (kept < 0 || kept > outN[p]) return ORBFE_ERR_STATE;
            nmatches[p] = kept;
        }
    }
    PTR();
#ifdef ORBFE_CALL_TRACE
    if (getenv("ORBFE_CALL_TRACE")) fprintf(stderr, "bow_run count=%d: pass1 %.1f stage %.1f launch %.1f sync %.1f tail %.1f us\n", count, trT[0], trT[1], trT[2], trT[3], trT[4]);
#endif
    return 0;
}
} // namespace

// Batched SearchByBoW: `count` independent (set 1, set 2) problems -- e.g. the relocalisation
// candidates of Tracking::Relocalization (src/Tracking.cc:3784, one call per candidate KF) or the
// covisible keyframes of LoopClosing (src/LoopClosing.cc:725) -- pooled into ONE upload, ONE launch
// (one wavefront per shared vocabulary node of any problem) and ONE download.
int orbfe_search_bow_batch(int device, int count, const orbfe_bow_args* args, int32_t* const* match, int* nmatches)
{
    return bow_run(device, count, args, nullptr, nullptr, match, nmatches);
}

int orbfe_search_bow(int device, const orbfe_bow_args* a, int32_t* match)
{
    if (!a || !match) return ORBFE_ERR_ARGS;
    int n = 0;
    int32_t* mp[1] = {match};
    const int r = orbfe_search_bow_batch(device, 1, a, mp, &n);
    return r < 0 ? r : n;
}

int orbfe_keyframe_create(orbfe_keyframe** out, int device, const orbfe_keyframe_args* a0)
{
    if (!out) return ORBFE_ERR_ARGS;
    *out = nullptr;
    orbfe_keyframe_args aLocal;
    const orbfe_keyframe_args* a = a0;
    if (a0 && a0->fv.nn == ORBFE_FV_RESIDENT) { // the vector of an orbfe_bow handle: its host copy (the handle keeps host views)
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv); rr < 0) return rr;
        a = &aLocal;
    }
    if (!a || a->n < 1 || a->n >= (1 << 20) || !a->desc || !a->mask || !fv_ok(a->fv)) return ORBFE_ERR_ARGS;
    const bool tri = a->kp_xy != nullptr;
    if (tri && (!a->octave || !a->uRight)) return ORBFE_ERR_ARGS;
    const size_t n = (size_t)a->n, ni = a->fv.nn ? (size_t)a->fv.offsets[a->fv.nn] : 0;
    for (size_t i = 0; i < ni; i++)
        if (a->fv.indices[i] < 0 || a->fv.indices[i] >= a->n) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oDesc = 0, oMask = oDesc + al(n * 32), oAng = oMask + al(n), oKp = oAng + al(n * 4), oUr = oKp + al(n * 8),
                 oOct = oUr + al(n * 4), oInd = oOct + al(n * 4), oNode = oInd + al(std::max<size_t>(ni, 1) * 4),
                 oOffs = oNode + al(std::max<size_t>((size_t)a->fv.nn, 1) * 4), total = oOffs + al(((size_t)a->fv.nn + 1) * 4);
    size_t blkCap = 0;
    void* blk = g_blockPool.get(device, total, &blkCap);
    if (!blk) return -(1000 + (int)hipErrorOutOfMemory);
    orbfe_keyframe* K = new orbfe_keyframe();
    K->device = device;
    K->n = a->n;
    K->block = (uint8_t*)blk;
    K->blockCap = blkCap;
    K->desc = K->block + oDesc;
    K->mask = K->block + oMask;
    K->ang = (float*)(K->block + oAng);
    K->kp = (float*)(K->block + oKp);
    K->uR = (float*)(K->block + oUr);
    K->oct = (int32_t*)(K->block + oOct);
    K->ind = (int32_t*)(K->block + oInd);
    K->dNode = (uint32_t*)(K->block + oNode);
    K->dOffs = (int32_t*)(K->block + oOffs);
    K->hasTri = tri;
    K->nodeIds.assign(a->fv.node_ids, a->fv.node_ids + a->fv.nn);
    K->offsets.assign(a->fv.offsets, a->fv.offsets + a->fv.nn + (a->fv.nn ? 1 : 0));
    if (K->offsets.empty()) K->offsets.push_back(0);
    for (int i = 0; i < a->fv.nn; i++) K->maxNode = std::max(K->maxNode, K->offsets[(size_t)i + 1] - K->offsets[(size_t)i]);
    K->indices.assign(a->fv.indices, a->fv.indices + ni);
    K->hMask.assign(a->mask, a->mask + n);
    if (a->angle) K->hAng.assign(a->angle, a->angle + n);
    if (tri) {
        K->hUR.assign(a->uRight, a->uRight + n);
        K->hOct.assign(a->octave, a->octave + n);
        if (n > 0) {
            const auto mm = std::minmax_element(K->hOct.begin(), K->hOct.end());
            K->octMin = *mm.first;
            K->octMax = *mm.second;
        }
    }
    Scratch s(device); // (this thread's matcher stream)
    const bool descResident = is_device_ptr(a->desc);
    if (descResident) {
        if (int w = orbfe_producer_wait(a->desc, g_ms); w < 0) {
            g_blockPool.put(device, blk, blkCap); // (this path used to leak the handle and its block)
            delete K;
            return w;
        }
    }
    // the whole block staged in this thread's pinned arena in the block's own layout, then ONE upload (seven pageable copies,
    // each staged and waited for by the runtime, were most of the 38 us this call took)
    hipError_t e = hipSuccess;
    const size_t first = descResident ? oMask : 0; // (resident descriptors: copied on the device)
    uint8_t* st = s.pin_scratch(total - first);
    if (st) {
        uint8_t* const b = st - first; // so that b + o* addresses the staged copy of block + o*
        if (!descResident) std::memcpy(b + oDesc, a->desc, n * 32);
        std::memcpy(b + oMask, a->mask, n);
        if (a->angle) std::memcpy(b + oAng, a->angle, n * 4);
        else std::memset(b + oAng, 0, n * 4);
        if (tri) {
            std::memcpy(b + oKp, a->kp_xy, n * 8);
            std::memcpy(b + oUr, a->uRight, n * 4);
            std::memcpy(b + oOct, a->octave, n * 4);
        }
        if (ni) std::memcpy(b + oInd, a->fv.indices, ni * 4);
        if (a->fv.nn) std::memcpy(b + oNode, K->nodeIds.data(), (size_t)a->fv.nn * 4);
        std::memcpy(b + oOffs, K->offsets.data(), ((size_t)a->fv.nn + 1) * 4);
        if (descResident) e = hipMemcpyAsync(K->desc, a->desc, n * 32, hipMemcpyDeviceToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(K->block + first, st, total - first, hipMemcpyHostToDevice, g_ms);
    } else { // (the arena is too small this once: array by array)
        e = hipMemcpyAsync(K->desc, a->desc, n * 32, descResident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(K->mask, a->mask, n, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && a->angle) e = hipMemcpyAsync(K->ang, a->angle, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && !a->angle) e = hipMemsetAsync(K->ang, 0, n * 4, g_ms);
        if (e == hipSuccess && tri) e = hipMemcpyAsync(K->kp, a->kp_xy, n * 8, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && tri) e = hipMemcpyAsync(K->uR, a->uRight, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && tri) e = hipMemcpyAsync(K->oct, a->octave, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && ni) e = hipMemcpyAsync(K->ind, a->fv.indices, ni * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && a->fv.nn) e = hipMemcpyAsync(K->dNode, K->nodeIds.data(), (size_t)a->fv.nn * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(K->dOffs, K->offsets.data(), ((size_t)a->fv.nn + 1) * 4, hipMemcpyHostToDevice, g_ms);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_ms); // the caller's arrays are free again; the handle is complete
    if (e != hipSuccess) {
        g_blockPool.put(device, blk, blkCap);
        delete K;
        return -(1000 + (int)e);
    }
    *out = K;
    return 0;
}

int orbfe_keyframe_set_mask(orbfe_keyframe* K, const uint8_t* mask)
{
    if (!K || !mask) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(K->device)) < 0) return r;
    if (std::memcmp(K->hMask.data(), mask, (size_t)K->n) == 0) return 0; // unchanged since the last call: nothing to send
    K->hMask.assign(mask, mask + K->n);
    Scratch s(K->device);
    // (ordered on this thread's matcher stream, which is the stream this thread's searches run on; the source is the
    // handle's own host copy, which lives until the next set_mask: wait here so that a second update cannot overtake it)
    HIP_TRY(hipMemcpyAsync(K->mask, K->hMask.data(), (size_t)K->n, hipMemcpyHostToDevice, g_ms));
    HIP_TRY(hipStreamSynchronize(g_ms));
    return 0;
}

void orbfe_keyframe_destroy(orbfe_keyframe* K)
{
    if (!K) return;
    // (no hipDeviceSynchronize here -- it drained the extractor's batches in flight and every other thread's
    // searches whenever the adapter's table evicted a keyframe.  The contract is the one of orbfe_frame_destroy: no call that
    // was given this handle is still running -- the adapter's reference count sees to it --, and every search has done all its
    // device reads before it returns, with the completion word as without it.)
    g_blockPool.put(K->device, K->block, K->blockCap);
    delete K;
}

int orbfe_search_bow_keyframes(int device, int count, orbfe_keyframe* const* kf1, orbfe_keyframe* const* kf2,
                               const orbfe_bow_args* args, int32_t* const* match, int* nmatches)
{
    return bow_run(device, count, args, kf1, kf2, match, nmatches);
}

// SearchForTriangulation_ of ONE keyframe against `count` neighbours (src/LocalMapping.cc:556-621), all sides resident:
// one upload of the row lists and pair records, ONE launch, one download.
int orbfe_search_tri_batch(orbfe_keyframe* K1, const uint8_t* hasMP1, int count, orbfe_keyframe* const* kf2,
                           const orbfe_tri_pair* pair, int32_t* const* pairs, int* npairs)
{
    if (!K1 || count < 0 || (count && (!kf2 || !pair || !pairs || !npairs)) || !K1->hasTri) return ORBFE_ERR_ARGS;
    PTR_BEGIN();
    const uint8_t* const has1 = hasMP1 ? hasMP1 : K1->hMask.data(); // (per call when given: see orbfe_search_bow_keyframes)
    const int device = K1->device, n1 = K1->n;
    std::vector<TriRowB> rows;
    std::vector<TriProb> probs(count);
    const orbfe_fv f1 = K1->fv();
    std::vector<int> rowOff[2], rowIdx[2]; // per bOnlyStereo: CSR of K1's rows (features without a MapPoint) by node
    size_t tabFloats = 0;
    for (int p = 0; p < count; p++) {
        const orbfe_keyframe* K2 = kf2[p];
        const orbfe_tri_pair& q = pair[p];
        if (!K2 || !K2->hasTri || K2->device != device || !pairs[p] || !q.scaleFactors2 || !q.levelSigma2_2 || q.nlevels2 < 1)
            return ORBFE_ERR_ARGS;
        if (q.check_orientation && (K1->hAng.empty() || K2->hAng.empty())) return ORBFE_ERR_ARGS;
        if (K2->n > 0 && (K2->octMin < 0 || K2->octMax >= q.nlevels2)) return ORBFE_ERR_ARGS; // (range kept by the handle)
        npairs[p] = 0;
        tabFloats += 2 * (size_t)q.nlevels2;
        const orbfe_fv f2 = K2->fv();
        bool bad = false;
        // (the rows of a node of K1 are the same for every neighbour with the same bOnlyStereo: listed once per call)
        const int so = q.only_stereo ? 1 : 0;
        if (rowOff[so].empty()) {
            rowOff[so].assign((size_t)f1.nn + 1, 0);
            rowIdx[so].reserve((size_t)n1);
            for (int i = 0; i < f1.nn; i++) {
                for (int k = f1.offsets[i]; k < f1.offsets[i + 1]; k++) {
                    const int idx1 = f1.indices[k];
                    if (has1[idx1]) continue;                              // :1279-1282
                    if (so && !(K1->hUR[idx1] >= 0)) continue;             // :1286-1288
                    rowIdx[so].push_back(idx1);
                }
                rowOff[so][(size_t)i + 1] = (int)rowIdx[so].size();
            }
            rows.reserve(rows.size() + rowIdx[so].size() * (size_t)(count - p));
        }
        for_each_shared_node(f1, f2, [&](int i, int j) {
            const int off2 = f2.offsets[j], n2 = f2.offsets[j + 1] - off2;
            if (n2 >= (1 << 20)) bad = true;
            if (n2 > 0)
                for (int k = rowOff[so][(size_t)i]; k < rowOff[so][(size_t)i + 1]; k++) rows.push_back(TriRowB{rowIdx[so][(size_t)k], off2, n2, p});
        });
        if (bad) return ORBFE_ERR_ARGS;
    }
    if (rows.empty()) return 0;
    PTR(); // rows
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    s.inPlace = rows.size() * sizeof(TriRowB) + (size_t)count * (sizeof(TriProb) + 64) <= inplace_limit(); // (rows + pair records only)
    TriRowB* dR;
    TriProb *dP, *hP;
    float *dTab, *hTab;
    int32_t* dM;
    if ((r = s.up(&dR, rows.data(), rows.size())) < 0) return r;
    size_t ovTotal = 0;
    for (int p = 0; p < count; p++)
        if (pair[p].hasMP2) ovTotal += ((size_t)kf2[p]->n + 63) & ~(size_t)63;
    uint8_t *dOv = nullptr, *hOv = nullptr;
    if (ovTotal && (r = s.reserve(&dOv, &hOv, ovTotal)) < 0) return r;
    if ((r = s.reserve(&dTab, &hTab, tabFloats)) < 0) return r;
    if ((r = s.reserve(&dP, &hP, (size_t)count)) < 0) return r;
    int32_t* hMir = nullptr;
    Scratch::OutBlock ob;
    // Form A (the usual one): the match rows stay on the device (the arena's clean block), k_tri_compact turns every
    // neighbour's row into its final pair list -- index order, rotation cull -- in the pinned mirror, and the host copies
    // those.  ORBFE_TRI_COMPACT=0: form B, the rows themselves come back and the host does it (A/B, and for batches whose
    // pair lists would not fit the mirror).
    static const bool compactOn = [] {
        const char* e = getenv("ORBFE_TRI_COMPACT");
        return !(e && e[0] == '0');
    }();
    uint8_t* cleanRows = nullptr;
    int32_t *dPairs = nullptr, *hPairs = nullptr, *dNp = nullptr, *hNp = nullptr;
    // (a single neighbour: its row is a microsecond of host work, the second kernel costs seven -- 0.018 against 0.025 ms)
    const bool compact = compactOn && count >= 4 && !g_timeKernels && (size_t)count * n1 * 8 <= (256u << 10) &&
                         s.clean_dev(&cleanRows, (size_t)count * n1 * sizeof(int32_t)) == 0 &&
                         s.mirror_out(&dPairs, &hPairs, (size_t)count * n1 * 2) == 0 && s.mirror_out(&dNp, &hNp, (size_t)count) == 0;
    const bool mirrored = !compact && (size_t)count * n1 * 4 <= (256u << 10) &&
                          s.out_block(&ob, (size_t)count * n1 * sizeof(int32_t), (unsigned)((rows.size() + 3) / 4)) == 0;
    if (compact) {
        dM = reinterpret_cast<int32_t*>(cleanRows);
    } else if (mirrored) {
        dM = reinterpret_cast<int32_t*>(ob.dev);
        hMir = reinterpret_cast<int32_t*>(ob.host);
    } else if ((r = s.up<int32_t>(&dM, nullptr, (size_t)count * n1)) < 0) return r;
    size_t tOff = 0, ovAt = 0;
    for (int p = 0; p < count; p++) {
        const orbfe_keyframe* K2 = kf2[p];
        const orbfe_tri_pair& q = pair[p];
        TriProb& Q = hP[p];
        Q.desc2 = K2->desc; Q.hasMP2 = K2->mask; Q.kp2 = K2->kp; Q.oct2 = K2->oct; Q.uR2 = K2->uR; Q.ind2 = K2->ind;
        if (q.hasMP2) { // this call's flags of the neighbour
            std::memcpy(hOv + ovAt, q.hasMP2, (size_t)K2->n);
            Q.hasMP2 = dOv + ovAt;
            ovAt += ((size_t)K2->n + 63) & ~(size_t)63;
        }
        std::memcpy(hTab + tOff, q.scaleFactors2, (size_t)q.nlevels2 * sizeof(float));
        std::memcpy(hTab + tOff + q.nlevels2, q.levelSigma2_2, (size_t)q.nlevels2 * sizeof(float));
        Q.sf2 = dTab + tOff;
        Q.sig2 = dTab + tOff + q.nlevels2;
        tOff += 2 * (size_t)q.nlevels2;
        std::memcpy(Q.F12, q.F12, sizeof Q.F12);
        Q.epx = q.ep[0];
        Q.epy = q.ep[1];
        Q.onlyStereo = q.only_stereo;
        Q.coarse = q.coarse;
        Q.outBase = p * n1;
        Q.nlevels2 = q.nlevels2;
        Q.ang2 = K2->ang;
        Q.checkOri = q.check_orientation ? 1 : 0;
        Q.pad = 0;
    }
    if (!mirrored && !compact) HIP_TRY(hipMemsetAsync(dM, 0xFF, (size_t)count * n1 * sizeof(int32_t), g_ms));
    const DoneSig done = s.done_sig((unsigned)rows.size(), mirrored ? &ob : nullptr, g_timeKernels);
    PTR(); // staging
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_search_tri_batch, dim3((unsigned)((rows.size() + 3) / 4)), dim3(256), 0, g_ms, dR, (int)rows.size(), dP,
                           K1->desc, K1->kp, K1->uR, dM, done);
    }
    HIP_TRY(hipGetLastError());
    if (compact) {
        DoneSig w = s.word_for((unsigned)count); // (the compaction's workgroups count themselves: one per neighbour)
        hipLaunchKernelGGL(k_tri_compact, dim3((unsigned)count), dim3(256), 0, g_ms, dM, n1, dP, K1->ang, dPairs, dNp, w);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            s.ar->cleanDirty = true;
            return -(1000 + (int)le);
        }
        PTR(); // launch
        INT_TRY(s.complete(w));
        PTR(); // wait
        for (int p = 0; p < count; p++) {
            const int np = hNp[p];
            if (np < 0 || np > n1) return ORBFE_ERR_STATE;
            std::memcpy(pairs[p], hPairs + (size_t)p * 2 * n1, (size_t)np * 2 * sizeof(int32_t));
            npairs[p] = np;
        }
        PTR();
#ifdef ORBFE_CALL_TRACE
        if (getenv("ORBFE_CALL_TRACE"))
            fprintf(stderr, "tri_batch (compact) count=%d rows=%zu: rows %.1f stage %.1f launch %.1f wait %.1f tail %.1f us\n", count, rows.size(),
                    trT[0], trT[1], trT[2], trT[3], trT[4]);
#endif
        return 0;
    }
    std::vector<int32_t> m;
    int32_t* mAll;
    PTR(); // launch
    if (mirrored) {
        INT_TRY(s.complete(done));
        PTR(); // wait
        mAll = hMir;
    } else {
        m.resize((size_t)count * n1);
        INT_TRY(s.down(m.data(), dM, m.size() * sizeof(int32_t)));
        INT_TRY(s.fetch());
        mAll = m.data();
    }
    // one pass over a neighbour's row collects its matches in index order (:1441-1446); the rotation histogram and its cull
    // (:1402-1438) then run over those alone
    std::vector<int8_t> bins;
    for (int p = 0; p < count; p++) {
        const int32_t* m12 = mAll + (size_t)p * n1;
        const orbfe_keyframe* K2 = kf2[p];
        int32_t* out = pairs[p];
        int np = 0;
        for (int i = 0; i < n1; i++) {
            const int32_t m = m12[i];
            if (m < 0) continue;
            out[2 * np] = i;
            out[2 * np + 1] = m;
            np++;
        }
        if (pair[p].check_orientation && np > 0) {
            bins.resize((size_t)np);
            int histo[HISTO_LENGTH] = {0};
            for (int k = 0; k < np; k++) {
                float rot = K1->hAng[out[2 * k]] - K2->hAng[out[2 * k + 1]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                bins[(size_t)k] = (int8_t)bin;
                if (bin >= 0 && bin < HISTO_LENGTH) histo[bin]++;
            }
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
            int kept = 0;
            for (int k = 0; k < np; k++) {
                const int b = bins[(size_t)k];
                if (!(b == ind1 || b == ind2 || b == ind3)) continue;
                out[2 * kept] = out[2 * k];
                out[2 * kept + 1] = out[2 * k + 1];
                kept++;
            }
            np = kept;
        }
        npairs[p] = np;
    }
    PTR();
#ifdef ORBFE_CALL_TRACE
    if (getenv("ORBFE_CALL_TRACE") && mirrored)
        fprintf(stderr, "tri_batch count=%d rows=%zu: rows %.1f stage %.1f launch %.1f wait %.1f tail %.1f us\n", count, rows.size(), trT[0], trT[1],
                trT[2], trT[3], trT[4]);
#endif
    return 0;
}

int orbfe_search_tri(int device, const orbfe_tri_args* a0, int32_t* pairs)
{
    orbfe_tri_args aLocal;
    const orbfe_tri_args* a = a0;
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2); rr < 0) return rr;
        a = &aLocal;
    }
    if (!a || !pairs || a->n1 < 0 || a->n2 < 0 || !fv_ok(a->fv1) || !fv_ok(a->fv2)) return ORBFE_ERR_ARGS;
    if (a->n1 == 0 || a->n2 == 0) return 0;
    if (!a->desc1 || !a->desc2 || !a->hasMP1 || !a->hasMP2 || !a->kp1_xy || !a->kp2_xy || !a->octave2 ||
        !a->uRight1 || !a->uRight2 || !a->scaleFactors2 || !a->levelSigma2_2 || a->nlevels2 < 1)
        return ORBFE_ERR_ARGS;
    if (a->check_orientation && (!a->angle1 || !a->angle2)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n2; i++)
        if (a->octave2[i] < 0 || a->octave2[i] >= a->nlevels2) return ORBFE_ERR_ARGS;
    std::vector<TriRow> rows;
    for_each_shared_node(a->fv1, a->fv2, [&](int i, int j) {
        const int off2 = a->fv2.offsets[j], n2 = a->fv2.offsets[j + 1] - off2;
        for (int k = a->fv1.offsets[i]; k < a->fv1.offsets[i + 1]; k++) {
            const int idx1 = a->fv1.indices[k];
            if (a->hasMP1[idx1]) continue;                             // :1279-1282
            if (a->only_stereo && !(a->uRight1[idx1] >= 0)) continue;   // :1286-1288
            if (n2 > 0) rows.push_back(TriRow{idx1, off2, n2});
        }
    });
    if (rows.empty()) return 0;
    for (const TriRow& t : rows)
        if (t.n2 >= (1 << 20)) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    TriRow* dR;
    uint8_t *d1, *d2, *h2;
    float *k1, *k2, *u1, *u2, *dF, *sf, *sg;
    int32_t *o2, *i2, *dM;
    // (latency path as in bow_run: two keyframes of ~1200 features stage ~110 KB, which the kernel reads in place; the
    // matches come back through the pinned mirror and its completion word)
    s.inPlace = rows.size() * sizeof(TriRow) + (size_t)a->n1 * 44 + (size_t)a->n2 * 53 + (size_t)a->fv2.offsets[a->fv2.nn] * 4 +
                    (size_t)a->nlevels2 * 8 + 4096 <= inplace_limit();
    if ((r = s.up(&dR, rows.data(), rows.size())) < 0) return r;
    if ((r = s.up_desc(&d1, a->desc1, (size_t)a->n1 * 32)) < 0) return r;
    if ((r = s.up_desc(&d2, a->desc2, (size_t)a->n2 * 32)) < 0) return r;
    if ((r = s.up(&h2, a->hasMP2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&k1, a->kp1_xy, (size_t)a->n1 * 2)) < 0) return r;
    if ((r = s.up(&k2, a->kp2_xy, (size_t)a->n2 * 2)) < 0) return r;
    if ((r = s.up(&u1, a->uRight1, (size_t)a->n1)) < 0) return r;
    if ((r = s.up(&u2, a->uRight2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&dF, a->F12, 9)) < 0) return r;
    if ((r = s.up(&sf, a->scaleFactors2, (size_t)a->nlevels2)) < 0) return r;
    if ((r = s.up(&sg, a->levelSigma2_2, (size_t)a->nlevels2)) < 0) return r;
    if ((r = s.up(&o2, a->octave2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&i2, a->fv2.indices, (size_t)a->fv2.offsets[a->fv2.nn])) < 0) return r;
    int32_t* hM = nullptr;
    Scratch::OutBlock ob;
    const bool mirrored = (size_t)a->n1 * 4 <= (256u << 10) && s.out_block(&ob, (size_t)a->n1 * sizeof(int32_t), (unsigned)((rows.size() + 3) / 4)) == 0;
    if (mirrored) {
        dM = reinterpret_cast<int32_t*>(ob.dev);
        hM = reinterpret_cast<int32_t*>(ob.host);
    } else {
        if ((r = s.up<int32_t>(&dM, nullptr, (size_t)a->n1)) < 0) return r;
        HIP_TRY(hipMemsetAsync(dM, 0xFF, (size_t)a->n1 * sizeof(int32_t), g_ms));
    }
    const DoneSig done = s.done_sig((unsigned)rows.size(), mirrored ? &ob : nullptr, g_timeKernels);
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_search_tri, dim3((unsigned)((rows.size() + 3) / 4)), dim3(256), 0, g_ms, dR, (int)rows.size(), d1,
                           k1, u1, d2, h2, k2, o2, u2, i2, dF, a->ep[0], a->ep[1], sf, sg, a->nlevels2, a->only_stereo, a->coarse, dM, done);
    }
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> m12(a->n1);
    if (mirrored) {
        INT_TRY(s.complete(done));
        std::memcpy(m12.data(), hM, (size_t)a->n1 * sizeof(int32_t));
    } else {
        INT_TRY(s.down(m12.data(), dM, (size_t)a->n1 * sizeof(int32_t)));
        INT_TRY(s.fetch());
    }
    std::vector<int8_t> bins(a->n1, -1);
    if (a->check_orientation) {
        for (int i = 0; i < a->n1; i++)
            if (m12[i] >= 0) {
                float rot = a->angle1[i] - a->angle2[m12[i]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                bins[i] = (int8_t)bin;
            }
    }
    cull_by_rotation(m12.data(), bins.data(), a->n1, a->check_orientation != 0);
    int np = 0;
    for (int i = 0; i < a->n1; i++) { // :1441-1446
        if (m12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = m12[i];
        np++;
    }
    return np;
}

int orbfe_stereo_fisheye_matches(int device, const uint8_t* descL, const float* kpL_xy, const int32_t* octL, int nL,
                                 const uint8_t* descR, const float* kpR_xy, const int32_t* octR, int nR,
                                 const float* params1, const float* params2, const float* Rlr, const float* tlr,
                                 const float* levelSigma2, int nlevels, int32_t* leftToRight, int32_t* rightToLeft,
                                 float* depth, float* p3D)
{
    if (nL < 0 || nR < 0 || nR >= (1 << 20) || nlevels < 1 || !params1 || !params2 || !Rlr || !tlr || !levelSigma2)
        return ORBFE_ERR_ARGS;
    if (nL && (!descL || !kpL_xy || !octL || !leftToRight || !depth || !p3D)) return ORBFE_ERR_ARGS;
    if (nR && (!descR || !kpR_xy || !octR || !rightToLeft)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < nL; i++)
        if (octL[i] < 0 || octL[i] >= nlevels) return ORBFE_ERR_ARGS;
    for (int i = 0; i < nR; i++)
        if (octR[i] < 0 || octR[i] >= nlevels) return ORBFE_ERR_ARGS;
    for (int i = 0; i < nR; i++) rightToLeft[i] = -1;
    for (int i = 0; i < nL; i++) {
        leftToRight[i] = -1;
        depth[i] = -1.0f;
        p3D[3 * i] = p3D[3 * i + 1] = p3D[3 * i + 2] = 0.f;
    }
    if (nL == 0 || nR == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t *dQ, *dT;
    int32_t *dI, *dD, *dOL, *dOR, *dL2R;
    float *dKL, *dKR, *dP1, *dP2, *dR, *dt, *dSig, *dDepth, *dX;
    // Round 5 (C5 taken apart, profiles/r05_c5_stages.txt): with both descriptor sets resident what travels is 12 bytes per
    // keypoint -- read by the triangulation kernel where it lies in the pinned staging (every thread reads its own few words once,
    // under an 85-us kernel) instead of a copy command and the queue's hand-over in front of the first kernel.  (Not with host
    // descriptors: the knn kernel reads every train row once per query block.)
    s.inPlace = is_device_ptr(descL) && is_device_ptr(descR) && ((size_t)nL + (size_t)nR) * 12 + 4096 <= inplace_limit();
    if ((r = s.up_desc(&dQ, descL, (size_t)nL * 32)) < 0) return r;
    if ((r = s.up_desc(&dT, descR, (size_t)nR * 32)) < 0) return r;
    if ((r = s.up(&dKL, kpL_xy, (size_t)nL * 2)) < 0) return r;
    if ((r = s.up(&dKR, kpR_xy, (size_t)nR * 2)) < 0) return r;
    if ((r = s.up(&dOL, octL, (size_t)nL)) < 0) return r;
    if ((r = s.up(&dOR, octR, (size_t)nR)) < 0) return r;
    if ((r = s.up(&dP1, params1, 8)) < 0) return r;
    if ((r = s.up(&dP2, params2, 8)) < 0) return r;
    if ((r = s.up(&dR, Rlr, 9)) < 0) return r;
    if ((r = s.up(&dt, tlr, 3)) < 0) return r;
    if ((r = s.up(&dSig, levelSigma2, (size_t)nlevels)) < 0) return r;
    if ((r = s.up<int32_t>(&dI, nullptr, (size_t)nL * 2)) < 0) return r;
    if ((r = s.up<int32_t>(&dD, nullptr, (size_t)nL * 2)) < 0) return r;
    // ... and the results (20 bytes per left keypoint) are stored by the kernel's threads into the pinned mirror themselves: the
    // threads end at very different times (Jacobi sweeps), so all but the last one's stores cross the link under the kernel, and
    // no download command follows it
    uint8_t *dMir = nullptr, *hMir = nullptr;
    const size_t oDepth = ((size_t)nL * 4 + 255) & ~(size_t)255, oX = 2 * oDepth;
    if (s.mirror_out(&dMir, &hMir, oX + (size_t)nL * 12) == 0) {
        dL2R = reinterpret_cast<int32_t*>(dMir);
        dDepth = reinterpret_cast<float*>(dMir + oDepth);
        dX = reinterpret_cast<float*>(dMir + oX);
    } else {
        dMir = hMir = nullptr;
        if ((r = s.up<int32_t>(&dL2R, nullptr, (size_t)nL)) < 0) return r;
        if ((r = s.up<float>(&dDepth, nullptr, (size_t)nL)) < 0) return r;
        if ((r = s.up<float>(&dX, nullptr, (size_t)nL * 3)) < 0) return r;
    }
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_bfknn2, dim3((unsigned)((nL + 3) / 4)), dim3(256), 0, g_ms, dQ, nL, dT, nR, dI, dD);
        hipLaunchKernelGGL(k_fisheye_stereo, dim3((unsigned)((nL + 255) / 256)), dim3(256), 0, g_ms, dI, dD, nL, nR, dKL, dKR, dOL,
                           dOR, dP1, dP2, dR, dt, dSig, dL2R, dDepth, dX);
    }
    HIP_TRY(hipGetLastError());
    if (hMir) {
        HIP_TRY(hipStreamSynchronize(g_ms));
        std::memcpy(leftToRight, hMir, (size_t)nL * sizeof(int32_t));
        std::memcpy(depth, hMir + oDepth, (size_t)nL * sizeof(float));
        std::memcpy(p3D, hMir + oX, (size_t)nL * 3 * sizeof(float));
    } else {
        INT_TRY(s.down(leftToRight, dL2R, (size_t)nL * sizeof(int32_t)));
        INT_TRY(s.down(depth, dDepth, (size_t)nL * sizeof(float)));
        INT_TRY(s.down(p3D, dX, (size_t)nL * 3 * sizeof(float)));
        INT_TRY(s.fetch());
    }
    int nMatches = 0;
    for (int q = 0; q < nL; q++) // mvRightToLeftMatch: the last left keypoint that chose a right one keeps it (:1150)
        if (leftToRight[q] >= 0) {
            rightToLeft[leftToRight[q]] = q;
            nMatches++;
        }
    return nMatches;
}

int orbfe_search_initialization(int device, const orbfe_init_args* a, int32_t* matches12)
{
    if (!a || !matches12 || a->n1 < 0 || a->n2 < 0 || a->window_size < 0) return ORBFE_ERR_ARGS;
    if (a->n1 && (!a->desc1 || !a->octave1 || !a->prev_xy)) return ORBFE_ERR_ARGS;
    if (a->n2 && (!a->desc2 || !a->kx2 || !a->ky2 || !a->octave2)) return ORBFE_ERR_ARGS;
    if (a->check_orientation && a->n1 && a->n2 && (!a->angle1 || !a->angle2)) return ORBFE_ERR_ARGS;
    if (a->n2 >= PROJ_MAXN) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n1; i++) matches12[i] = -1;
    if (a->n1 == 0 || a->n2 == 0) return 0;
    // one query per level-0 keypoint of F1, in index order (:721-724)
    std::vector<int32_t> qidx;
    for (int i = 0; i < a->n1; i++)
        if (!(a->octave1[i] > 0)) qidx.push_back(i);
    const size_t n = (size_t)a->n2, nq = qidx.size();
    if (nq == 0) return 0;
    std::vector<uint8_t> qdesc(nq * 32);
    std::vector<float> qx(nq), qy(nq), qr(nq, (float)a->window_size);
    std::vector<int32_t> qlev(nq);
    for (size_t q = 0; q < nq; q++) {
        std::memcpy(&qdesc[q * 32], a->desc1 + (size_t)qidx[q] * 32, 32);
        qx[q] = a->prev_xy[2 * qidx[q]];
        qy[q] = a->prev_xy[2 * qidx[q] + 1];
        qlev[q] = a->octave1[qidx[q]]; // GetFeaturesInArea(..., level1, level1)
    }
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    InitDev I{};
    ProjDev& P = I.P;
    uint8_t *dDesc, *dQdesc;
    float *dKx, *dKy, *dQx, *dQy, *dQr;
    int32_t *dOct, *dQlev;
    if ((r = s.up_desc(&dDesc, a->desc2, n * 32)) < 0) return r;
    if ((r = s.up(&dKx, a->kx2, n)) < 0) return r;
    if ((r = s.up(&dKy, a->ky2, n)) < 0) return r;
    if ((r = s.up(&dOct, a->octave2, n)) < 0) return r;
    if ((r = s.up(&dQdesc, qdesc.data(), nq * 32)) < 0) return r;
    if ((r = s.up(&dQx, qx.data(), nq)) < 0) return r;
    if ((r = s.up(&dQy, qy.data(), nq)) < 0) return r;
    if ((r = s.up(&dQr, qr.data(), nq)) < 0) return r;
    if ((r = s.up(&dQlev, qlev.data(), nq)) < 0) return r;
    if ((r = s.up<int32_t>(&P.cellStart, nullptr, 2 * PROJ_CELLS + 1)) < 0) return r;
    if ((r = s.up<int32_t>(&P.cellItems, nullptr, n)) < 0) return r;
    if ((r = s.up<int32_t>(&P.cellOf, nullptr, n)) < 0) return r;
    if ((r = s.up<int32_t>(&P.qStart, nullptr, nq)) < 0) return r;
    if ((r = s.up<int32_t>(&P.qCount, nullptr, nq)) < 0) return r;
    if ((r = s.up<int32_t>(&I.head, nullptr, 2 * n)) < 0) return r;
    if ((r = s.up<int32_t>(&I.next, nullptr, 2 * nq)) < 0) return r;
    if ((r = s.up<int32_t>(&I.choice, nullptr, 2 * nq)) < 0) return r;
    if ((r = s.up<int32_t>(&I.cdist, nullptr, 2 * nq)) < 0) return r;
    int32_t* dOut;
    if ((r = s.up<int32_t>(&dOut, nullptr, 4 + nq)) < 0) return r;
    P.status = dOut;
    P.qMatch = dOut + 4;
    size_t keyCap = PROJ_QUOTA * nq + std::max<size_t>(PROJ_QUOTA * nq, 1 << 15); // (the queries' own stretches + overflow)
    if ((r = s.up<unsigned long long>(&P.rawKeys, nullptr, keyCap)) < 0) return r;
    if ((r = s.up<unsigned long long>(&P.sortedKeys, nullptr, keyCap)) < 0) return r;
    P.keyCap = (int)keyCap;
    P.desc = dDesc; P.kx = dKx; P.ky = dKy; P.octave = dOct; P.n = a->n2; P.Nleft = -1;
    P.minX = a->minX; P.minY = a->minY; P.wInv = a->gridWInv; P.hInv = a->gridHInv;
    P.nq = (int)nq; P.qdesc = dQdesc; P.qx = dQx; P.qy = dQy; P.qr = dQr; P.qmin = dQlev; P.qmax = dQlev;
    P.mode = 1;
    I.nnratio = a->nnratio;
    std::vector<int32_t> out(4 + nq);
    for (int attempt = 0;; attempt++) {
        {
            KernelTimer timer(s);
            hipLaunchKernelGGL(k_proj_grid, dim3(1), dim3(PROJ_THREADS), 0, g_ms, P);
            hipLaunchKernelGGL(k_proj_candidates, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, g_ms, P);
            hipLaunchKernelGGL(k_init_sweeps, dim3(1), dim3(PROJ_THREADS), 0, g_ms, I);
        }
        HIP_TRY(hipGetLastError());
        INT_TRY(s.down(out.data(), dOut, out.size() * 4));
        INT_TRY(s.fetch());
        if (out[2] >= 0 && PROJ_QUOTA * nq + (size_t)out[2] <= keyCap) break;
        if (attempt > 0 || out[2] < 0) return ORBFE_ERR_STATE;
        keyCap = PROJ_QUOTA * nq + (size_t)out[2];
        if ((r = s.up<unsigned long long>(&P.rawKeys, nullptr, keyCap)) < 0) return r;
        if ((r = s.up<unsigned long long>(&P.sortedKeys, nullptr, keyCap)) < 0) return r;
        P.keyCap = (int)keyCap;
    }
    g_lastProjSweeps = out[1];
    // the bookkeeping of :765-789 over the per-query choices, in query order
    std::vector<int32_t> vnMatches21(n, -1);
    std::vector<int8_t> bins(a->n1, -1);
    int nmatches = 0;
    for (size_t q = 0; q < nq; q++) {
        const int f = out[4 + q];
        if (f < 0) continue;
        const int i1 = qidx[q];
        if (vnMatches21[f] >= 0) {
            matches12[vnMatches21[f]] = -1;
            nmatches--;
        }
        matches12[i1] = f;
        vnMatches21[f] = i1;
        nmatches++;
        if (a->check_orientation) {
            float rot = a->angle1[i1] - a->angle2[f];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            bins[i1] = (int8_t)bin;
        }
    }
    if (a->check_orientation) { // :791-811: the histogram counts every accepted query, also those robbed later
        int histo[HISTO_LENGTH] = {0};
        for (int i = 0; i < a->n1; i++)
            if (bins[i] >= 0 && bins[i] < HISTO_LENGTH) histo[bins[i]]++;
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < a->n1; i++)
            if (bins[i] >= 0 && bins[i] != ind1 && bins[i] != ind2 && bins[i] != ind3 && matches12[i] >= 0) {
                matches12[i] = -1;
                nmatches--;
            }
    }
    return nmatches;
}

int orbfe_search_tri_kb8(int device, const orbfe_tri_kb8_args* a0, int32_t* pairs)
{
    orbfe_tri_kb8_args aLocal;
    const orbfe_tri_kb8_args* a = a0;
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2); rr < 0) return rr;
        a = &aLocal;
    }
    if (!a || !pairs || a->n1 < 0 || a->n2 < 0 || !fv_ok(a->fv1) || !fv_ok(a->fv2)) return ORBFE_ERR_ARGS;
    if (a->n1 == 0 || a->n2 == 0) return 0;
    if (!a->desc1 || !a->desc2 || !a->hasMP1 || !a->hasMP2 || !a->kp1_xy || !a->kp2_xy || !a->octave1 || !a->octave2 ||
        !a->scaleFactors2 || !a->levelSigma2_1 || !a->levelSigma2_2 || a->nlevels1 < 1 || a->nlevels2 < 1 || !a->kb8_1L ||
        !a->kb8_2L || !a->R12 || !a->t12)
        return ORBFE_ERR_ARGS;
    const bool rig = a->Nleft1 != -1 && a->Nleft2 != -1;
    if ((a->Nleft1 == -1) != (a->Nleft2 == -1)) return ORBFE_ERR_ARGS; // the reference dereferences both second cameras
    if (rig && (!a->kb8_1R || !a->kb8_2R || a->Nleft1 < 0 || a->Nleft1 > a->n1 || a->Nleft2 < 0 || a->Nleft2 > a->n2))
        return ORBFE_ERR_ARGS;
    if (a->check_orientation && (!a->angle1 || !a->angle2)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n1; i++)
        if (a->octave1[i] < 0 || a->octave1[i] >= a->nlevels1) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n2; i++)
        if (a->octave2[i] < 0 || a->octave2[i] >= a->nlevels2) return ORBFE_ERR_ARGS;
    std::vector<TriRow> rows;
    for_each_shared_node(a->fv1, a->fv2, [&](int i, int j) {
        const int off2 = a->fv2.offsets[j], n2 = a->fv2.offsets[j + 1] - off2;
        for (int k = a->fv1.offsets[i]; k < a->fv1.offsets[i + 1]; k++) {
            const int idx1 = a->fv1.indices[k];
            if (a->hasMP1[idx1]) continue;
            const bool bStereo1 = !rig && a->uRight1 && a->uRight1[idx1] >= 0;
            if (a->only_stereo && !bStereo1) continue;
            if (n2 > 0) rows.push_back(TriRow{idx1, off2, n2});
        }
    });
    if (rows.empty()) return 0;
    for (const TriRow& t : rows)
        if (t.n2 >= (1 << 20)) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    TriKb8Dev T{};
    TriRow* dR;
    uint8_t *d1, *d2, *h2;
    float *k1, *k2, *u1 = nullptr, *u2 = nullptr, *sf, *sg1, *sg2;
    int32_t *o1, *o2, *i2, *dM;
    if ((r = s.up(&dR, rows.data(), rows.size())) < 0) return r;
    if ((r = s.up_desc(&d1, a->desc1, (size_t)a->n1 * 32)) < 0) return r;
    if ((r = s.up_desc(&d2, a->desc2, (size_t)a->n2 * 32)) < 0) return r;
    if ((r = s.up(&h2, a->hasMP2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&k1, a->kp1_xy, (size_t)a->n1 * 2)) < 0) return r;
    if ((r = s.up(&k2, a->kp2_xy, (size_t)a->n2 * 2)) < 0) return r;
    if (a->uRight1 && (r = s.up(&u1, a->uRight1, (size_t)a->n1)) < 0) return r;
    if (a->uRight2 && (r = s.up(&u2, a->uRight2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&sf, a->scaleFactors2, (size_t)a->nlevels2)) < 0) return r;
    if ((r = s.up(&sg1, a->levelSigma2_1, (size_t)a->nlevels1)) < 0) return r;
    if ((r = s.up(&sg2, a->levelSigma2_2, (size_t)a->nlevels2)) < 0) return r;
    if ((r = s.up(&o1, a->octave1, (size_t)a->n1)) < 0) return r;
    if ((r = s.up(&o2, a->octave2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&i2, a->fv2.indices, (size_t)a->fv2.offsets[a->fv2.nn])) < 0) return r;
    if ((r = s.up<int32_t>(&dM, nullptr, (size_t)a->n1)) < 0) return r;
    HIP_TRY(hipMemsetAsync(dM, 0xFF, (size_t)a->n1 * sizeof(int32_t), g_ms));
    T.rows = dR; T.nRows = (int)rows.size(); T.desc1 = d1; T.desc2 = d2; T.hasMP2 = h2; T.kp1 = k1; T.kp2 = k2;
    T.uR1 = u1; T.uR2 = u2; T.oct1 = o1; T.oct2 = o2; T.ind2 = i2; T.Nleft1 = a->Nleft1; T.Nleft2 = a->Nleft2;
    T.rig = rig ? 1 : 0;
    const float* Ps[4] = {a->kb8_1L, rig ? a->kb8_1R : a->kb8_1L, a->kb8_2L, rig ? a->kb8_2R : a->kb8_2L};
    for (int c = 0; c < 4; c++) std::memcpy(T.P[c], Ps[c], 8 * sizeof(float));
    const int nposes = rig ? 4 : 1;
    for (int c = 0; c < 4; c++) {
        std::memcpy(T.R12[c], a->R12 + 9 * (c < nposes ? c : 0), 9 * sizeof(float));
        std::memcpy(T.t12[c], a->t12 + 3 * (c < nposes ? c : 0), 3 * sizeof(float));
    }
    T.epx = a->ep[0]; T.epy = a->ep[1]; T.sf2 = sf; T.sig1 = sg1; T.sig2 = sg2;
    T.onlyStereo = a->only_stereo; T.coarse = a->coarse; T.match12 = dM;
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_search_tri_kb8, dim3((unsigned)((rows.size() + 3) / 4)), dim3(256), 0, g_ms, T);
    }
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> m12(a->n1);
    INT_TRY(s.down(m12.data(), dM, (size_t)a->n1 * sizeof(int32_t)));
    INT_TRY(s.fetch());
    std::vector<int8_t> bins(a->n1, -1);
    if (a->check_orientation) {
        for (int i = 0; i < a->n1; i++)
            if (m12[i] >= 0) {
                float rot = a->angle1[i] - a->angle2[m12[i]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                bins[i] = (int8_t)bin;
            }
    }
    cull_by_rotation(m12.data(), bins.data(), a->n1, a->check_orientation != 0);
    int np = 0;
    for (int i = 0; i < a->n1; i++) {
        if (m12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = m12[i];
        np++;
    }
    return np;
}

int orbfe_search_tri_3d(int device, const orbfe_tri3d_args* a0, int32_t* pairs, float* points)
{
    orbfe_tri3d_args aLocal;
    const orbfe_tri3d_args* a = a0;
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2); rr < 0) return rr;
        a = &aLocal;
    }
    if (!a || !pairs || !points || a->n1 < 0 || a->n2 < 0 || !fv_ok(a->fv1) || !fv_ok(a->fv2)) return ORBFE_ERR_ARGS;
    if (a->n1 == 0 || a->n2 == 0) return 0;
    if (!a->desc1 || !a->desc2 || !a->hasMP1 || !a->hasMP2 || !a->kp1_xy || !a->kp2_xy || !a->octave1 || !a->octave2 ||
        !a->levelSigma2_1 || !a->levelSigma2_2 || a->nlevels1 < 1 || a->nlevels2 < 1)
        return ORBFE_ERR_ARGS;
    if (!a->kb8_1L) return 0; // Pinhole::matchAndtriangulate returns false (include/CameraModels/Pinhole.h:88-91)
    if (!a->kb8_2L || !a->Tcw1L || !a->Tcw2L) return ORBFE_ERR_ARGS;
    if (a->Nleft1 != -1 && (a->Nleft1 < 0 || a->Nleft1 > a->n1 || !a->kb8_1R || !a->Tcw1R)) return ORBFE_ERR_ARGS;
    if (a->Nleft2 != -1 && (a->Nleft2 < 0 || a->Nleft2 > a->n2 || !a->kb8_2R || !a->Tcw2R)) return ORBFE_ERR_ARGS;
    if (a->check_orientation && (!a->angle1 || !a->angle2)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n1; i++)
        if (a->octave1[i] < 0 || a->octave1[i] >= a->nlevels1) return ORBFE_ERR_ARGS;
    for (int i = 0; i < a->n2; i++)
        if (a->octave2[i] < 0 || a->octave2[i] >= a->nlevels2) return ORBFE_ERR_ARGS;
    std::vector<TriRow> rows;
    for_each_shared_node(a->fv1, a->fv2, [&](int i, int j) {
        const int off2 = a->fv2.offsets[j], n2 = a->fv2.offsets[j + 1] - off2;
        for (int k = a->fv1.offsets[i]; k < a->fv1.offsets[i + 1]; k++) {
            const int idx1 = a->fv1.indices[k];
            if (a->hasMP1[idx1]) continue;
            if (n2 > 0) rows.push_back(TriRow{idx1, off2, n2});
        }
    });
    if (rows.empty()) return 0;
    for (const TriRow& t : rows)
        if (t.n2 >= (1 << 20)) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    Tri3dDev T{};
    TriRow* dR;
    uint8_t *d1, *d2, *h2;
    float *k1, *k2, *sg1, *sg2, *dX;
    int32_t *o1, *o2, *i2, *dM;
    if ((r = s.up(&dR, rows.data(), rows.size())) < 0) return r;
    if ((r = s.up_desc(&d1, a->desc1, (size_t)a->n1 * 32)) < 0) return r;
    if ((r = s.up_desc(&d2, a->desc2, (size_t)a->n2 * 32)) < 0) return r;
    if ((r = s.up(&h2, a->hasMP2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&k1, a->kp1_xy, (size_t)a->n1 * 2)) < 0) return r;
    if ((r = s.up(&k2, a->kp2_xy, (size_t)a->n2 * 2)) < 0) return r;
    if ((r = s.up(&sg1, a->levelSigma2_1, (size_t)a->nlevels1)) < 0) return r;
    if ((r = s.up(&sg2, a->levelSigma2_2, (size_t)a->nlevels2)) < 0) return r;
    if ((r = s.up(&o1, a->octave1, (size_t)a->n1)) < 0) return r;
    if ((r = s.up(&o2, a->octave2, (size_t)a->n2)) < 0) return r;
    if ((r = s.up(&i2, a->fv2.indices, (size_t)a->fv2.offsets[a->fv2.nn])) < 0) return r;
    if ((r = s.up<int32_t>(&dM, nullptr, (size_t)a->n1)) < 0) return r;
    if ((r = s.up<float>(&dX, nullptr, (size_t)a->n1 * 3)) < 0) return r;
    HIP_TRY(hipMemsetAsync(dM, 0xFF, (size_t)a->n1 * sizeof(int32_t), g_ms));
    T.rows = dR; T.nRows = (int)rows.size(); T.desc1 = d1; T.desc2 = d2; T.hasMP2 = h2; T.kp1 = k1; T.kp2 = k2;
    T.oct1 = o1; T.oct2 = o2; T.ind2 = i2; T.Nleft1 = a->Nleft1; T.Nleft2 = a->Nleft2;
    const float* Ps[4] = {a->kb8_1L, a->Nleft1 != -1 ? a->kb8_1R : a->kb8_1L, a->kb8_2L, a->Nleft2 != -1 ? a->kb8_2R : a->kb8_2L};
    const float* Ts[4] = {a->Tcw1L, a->Nleft1 != -1 ? a->Tcw1R : a->Tcw1L, a->Tcw2L, a->Nleft2 != -1 ? a->Tcw2R : a->Tcw2L};
    for (int c = 0; c < 4; c++) {
        std::memcpy(T.P[c], Ps[c], 8 * sizeof(float));
        std::memcpy(T.T[c], Ts[c], 12 * sizeof(float));
    }
    T.sig1 = sg1; T.sig2 = sg2; T.match12 = dM; T.points = dX;
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_search_tri_3d, dim3((unsigned)((rows.size() + 3) / 4)), dim3(256), 0, g_ms, T);
    }
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> m12(a->n1);
    std::vector<float> X((size_t)a->n1 * 3);
    INT_TRY(s.down(m12.data(), dM, (size_t)a->n1 * sizeof(int32_t)));
    INT_TRY(s.down(X.data(), dX, X.size() * sizeof(float)));
    INT_TRY(s.fetch());
    std::vector<int8_t> bins(a->n1, -1);
    if (a->check_orientation) {
        for (int i = 0; i < a->n1; i++)
            if (m12[i] >= 0) {
                float rot = a->angle1[i] - a->angle2[m12[i]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                bins[i] = (int8_t)bin;
            }
    }
    cull_by_rotation(m12.data(), bins.data(), a->n1, a->check_orientation != 0);
    int np = 0;
    for (int i = 0; i < a->n1; i++) {
        if (m12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = m12[i];
        for (int k = 0; k < 3; k++) points[3 * np + k] = X[3 * (size_t)i + k];
        np++;
    }
    return np;
}

int orbfe_kb8_triangulate(int device, const float* params1, const float* params2, const float* kp1_xy, const float* kp2_xy,
                          const float* R12, const float* t12, const float* sigma1, const float* sigma2, int n, float* z1,
                          float* p3D)
{
    if (!params1 || !params2 || !kp1_xy || !kp2_xy || !R12 || !t12 || !sigma1 || !sigma2 || !z1 || n < 0) return ORBFE_ERR_ARGS;
    if (n == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    float *dP1, *dP2, *dK1, *dK2, *dR, *dT, *dS1, *dS2, *dZ;
    if ((r = s.up(&dP1, params1, 8)) < 0) return r;
    if ((r = s.up(&dP2, params2, 8)) < 0) return r;
    if ((r = s.up(&dK1, kp1_xy, (size_t)2 * n)) < 0) return r;
    if ((r = s.up(&dK2, kp2_xy, (size_t)2 * n)) < 0) return r;
    if ((r = s.up(&dR, R12, 9)) < 0) return r;
    if ((r = s.up(&dT, t12, 3)) < 0) return r;
    if ((r = s.up(&dS1, sigma1, (size_t)n)) < 0) return r;
    if ((r = s.up(&dS2, sigma2, (size_t)n)) < 0) return r;
    if ((r = s.up<float>(&dZ, nullptr, (size_t)n)) < 0) return r;
    float* dX = nullptr;
    if (p3D && (r = s.up<float>(&dX, nullptr, (size_t)3 * n)) < 0) return r;
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_kb8_triangulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_ms, dP1, dP2, dK1, dK2, dR, dT,
                           dS1, dS2, n, dZ, dX);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(z1, dZ, (size_t)n * sizeof(float)));
    if (p3D) INT_TRY(s.down(p3D, dX, (size_t)3 * n * sizeof(float)));
    INT_TRY(s.fetch());
    return 0;
}

namespace {
// argument checks of one projection search (everything the kernels rely on)
int proj_validate(const orbfe_proj_args* a, const int32_t* q_match, const int32_t* feat_match)
{
    if (!a || a->n < 0 || a->nq < 0 || (a->mode != 0 && a->mode != 1)) return ORBFE_ERR_ARGS;
    if (a->n && (!a->desc || !a->kx || !a->ky || !a->octave || !feat_match)) return ORBFE_ERR_ARGS;
    if (a->nq && (!a->qdesc || !a->qx || !a->qy || !a->qr || !a->qmin_level || !a->qmax_level || !q_match))
        return ORBFE_ERR_ARGS;
    if (a->Nleft != -1 && (a->Nleft < 0 || a->Nleft > a->n)) return ORBFE_ERR_ARGS;
    if (a->n >= PROJ_MAXN || a->nq >= (1 << 28)) return ORBFE_ERR_ARGS;
    const bool orient = a->mode == 1 && a->check_orientation;
    if (orient && a->nq && (!a->angle || !a->qangle)) return ORBFE_ERR_ARGS;
    if (a->Nleft == -1 && a->uright && a->nq && !a->qxr) return ORBFE_ERR_ARGS;
    if (a->chi2_gate) { // Fuse's reprojection test: sigma table indexed by the candidates' octaves
        if (a->mode != 1 || !a->inv_level_sigma2 || a->n_levels < 1 || (a->uright && a->nq && !a->qxr)) return ORBFE_ERR_ARGS;
        for (int i = 0; i < a->n; i++)
            if (a->octave[i] < 0 || a->octave[i] >= a->n_levels) return ORBFE_ERR_ARGS;
    }
    for (int q = 0; q < a->nq; q++) {
        const int f = a->qflags ? a->qflags[q] : 0;
        if ((f & 1) && a->Nleft == -1) return ORBFE_ERR_ARGS; // there is no right grid
        // bit 2 refers to the query before, which must be an unconditional one
        if ((f & 4) && (q == 0 || (a->qflags[q - 1] & 6))) return ORBFE_ERR_ARGS;
        // (a non-blocking map point -- Observations() == 0 -- that overwrites a stereo partner can free a taken feature again,
        // :117-121: that search walks its queries in order, proj_needs_inorder / proj_inorder_body; refused until round 5)
    }
    if (a->Nleft != -1 && a->mode == 0 && a->n && a->nq) {
        for (int i = 0; a->left_to_right && i < a->Nleft; i++)
            if (a->left_to_right[i] < -1 || a->left_to_right[i] >= a->n - a->Nleft) return ORBFE_ERR_ARGS;
        for (int i = 0; a->right_to_left && i < a->n - a->Nleft; i++)
            if (a->right_to_left[i] < -1 || a->right_to_left[i] >= a->Nleft) return ORBFE_ERR_ARGS;
    }
    return 0;
}

bool proj_needs_inorder(const orbfe_proj_args* a)
{
    if (!(a->qblocks && a->mode == 0 && a->Nleft != -1 && (a->left_to_right || a->right_to_left))) return false;
    for (int q = 0; q < a->nq; q++)
        if (!a->qblocks[q]) return true;
    return false;
}

struct ProjJob {
    ProjDev P{};
    size_t keyCap = 0, outOff = 0; // outputs of the job: status[4] | qMatch[nq] | featMatch[n] at dOut + outOff
    size_t sweepBytes = 0;
};

} // namespace

// Frame side of the projection searches kept on the device between calls (orbfe_frame_create): the feature arrays and
// the grid of Frame::AssignFeaturesToGrid, built once; host copies of what the host tail of a search reads.
struct orbfe_frame {
    int device = 0, n = 0, Nleft = -1;
    float minX = 0, minY = 0, wInv = 0, hInv = 0;
    uint8_t* block = nullptr; // one allocation: everything below points into it
    size_t blockCap = 0;
    uint8_t* desc = nullptr;
    float *kx = nullptr, *ky = nullptr, *uright = nullptr;
    int32_t *octave = nullptr, *cellStart = nullptr, *cellItems = nullptr, *cellOf = nullptr, *status = nullptr;
    std::vector<int32_t> hOctave;
    std::vector<float> hAngle;
};

namespace {
// inputs and work arrays of one search on the device (outputs are assigned by the caller: one block per call).  Two phases,
// so that a batch stages the inputs of ALL its searches next to each other (one run of the pinned mirror = one upload
// command for the batch instead of one per search) and the work arrays after them: phase 0 = inputs, phase 1 = the rest.
// `prev` / `prevJ`: the search staged just before this one in the same call.  The searches of a batch usually share a side --
// one keyframe's map points fused into every neighbour (the same query descriptors), or every neighbour's points into the one
// keyframe (the same frame side), src/LocalMapping.cc:803-870 -- and a read-only array that comes with the same pointer and
// size as its predecessor's is staged and uploaded once.
int proj_stage(Scratch& s, const orbfe_proj_args* a, ProjJob& J, const orbfe_frame* F, int phase, const orbfe_proj_args* prev = nullptr,
               const ProjJob* prevJ = nullptr)
{
    int r;
    ProjDev& P = J.P;
    const size_t n = (size_t)a->n, nq = (size_t)a->nq;
    if (phase == 0) {
        const bool sameN = prev && prev->n == a->n, sameQ = prev && prev->nq == a->nq;
        uint8_t *dDesc = nullptr, *dTaken = nullptr, *dQdesc, *dQflags = nullptr, *dQblocks = nullptr;
        float *dKx = nullptr, *dKy = nullptr, *dUr = nullptr, *dQx, *dQy, *dQr, *dQxr = nullptr;
        int32_t *dOct = nullptr, *dL2r = nullptr, *dR2l = nullptr, *dQmin, *dQmax;
        if (F) { // the frame's arrays and grid are resident
            dDesc = F->desc; dKx = F->kx; dKy = F->ky; dOct = F->octave;
            if (F->uright && (a->Nleft == -1 || a->chi2_gate)) dUr = F->uright;
        } else {
            if (sameN && prev->desc == a->desc) dDesc = const_cast<uint8_t*>(prevJ->P.desc);
            else if ((r = s.up_desc(&dDesc, a->desc, n * 32)) < 0) return r;
            if (sameN && prev->kx == a->kx) dKx = const_cast<float*>(prevJ->P.kx);
            else if ((r = s.up(&dKx, a->kx, n)) < 0) return r;
            if (sameN && prev->ky == a->ky) dKy = const_cast<float*>(prevJ->P.ky);
            else if ((r = s.up(&dKy, a->ky, n)) < 0) return r;
            if (sameN && prev->octave == a->octave) dOct = const_cast<int32_t*>(prevJ->P.octave);
            else if ((r = s.up(&dOct, a->octave, n)) < 0) return r;
            if (a->uright && (a->Nleft == -1 || a->chi2_gate)) {
                if (sameN && prev->uright == a->uright && prevJ->P.uright) dUr = const_cast<float*>(prevJ->P.uright);
                else if ((r = s.up(&dUr, a->uright, n)) < 0) return r;
            }
        }
        float* dInvSigma2 = nullptr;
        if (a->chi2_gate && (r = s.up(&dInvSigma2, a->inv_level_sigma2, (size_t)a->n_levels)) < 0) return r;
        if (a->taken && (r = s.up(&dTaken, a->taken, n)) < 0) return r;
        if (a->Nleft != -1 && a->mode == 0) {
            if (a->left_to_right && (r = s.up(&dL2r, a->left_to_right, (size_t)a->Nleft)) < 0) return r;
            if (a->right_to_left && (r = s.up(&dR2l, a->right_to_left, n - (size_t)a->Nleft)) < 0) return r;
        }
        if (sameQ && prev->qdesc == a->qdesc) dQdesc = const_cast<uint8_t*>(prevJ->P.qdesc);
        else if ((r = s.up_desc(&dQdesc, a->qdesc, nq * 32)) < 0) return r;
        if ((r = s.up(&dQx, a->qx, nq)) < 0) return r;
        if ((r = s.up(&dQy, a->qy, nq)) < 0) return r;
        if ((r = s.up(&dQr, a->qr, nq)) < 0) return r;
        if (dUr && (r = s.up(&dQxr, a->qxr, nq)) < 0) return r;
        if ((r = s.up(&dQmin, a->qmin_level, nq)) < 0) return r;
        if ((r = s.up(&dQmax, a->qmax_level, nq)) < 0) return r;
        if (a->qflags && (r = s.up(&dQflags, a->qflags, nq)) < 0) return r;
        if (a->qblocks && (r = s.up(&dQblocks, a->qblocks, nq)) < 0) return r;
        P.desc = dDesc; P.kx = dKx; P.ky = dKy; P.octave = dOct; P.uright = dUr; P.taken = dTaken;
        P.l2r = dL2r; P.r2l = dR2l; P.n = a->n; P.Nleft = a->Nleft;
        P.minX = a->minX; P.minY = a->minY; P.wInv = a->gridWInv; P.hInv = a->gridHInv;
        P.nq = a->nq; P.qdesc = dQdesc; P.qx = dQx; P.qy = dQy; P.qr = dQr; P.qxr = dQxr;
        P.qmin = dQmin; P.qmax = dQmax; P.qflags = dQflags; P.qblocks = dQblocks;
        P.mode = a->mode; P.nnratio = a->nnratio; P.thHigh = a->th_high;
        P.invSigma2 = dInvSigma2; P.chi2 = a->chi2_gate ? 1 : 0;
        P.inorder = proj_needs_inorder(a) ? 1 : 0;
        P.taken0 = nullptr;
        if (P.inorder) { // (the entry state is applied by the walk, not by the candidates' static test)
            P.taken0 = P.taken;
            P.taken = nullptr;
        }
        return 0;
    }
    P.resident = F ? 1 : 0;
    if (F) {
        P.cellStart = F->cellStart; P.cellItems = F->cellItems; P.cellOf = F->cellOf;
    } else {
        if ((r = s.up<int32_t>(&P.cellStart, nullptr, 2 * PROJ_CELLS + 1)) < 0) return r;
        if ((r = s.up<int32_t>(&P.cellItems, nullptr, n)) < 0) return r;
        if ((r = s.up<int32_t>(&P.cellOf, nullptr, n)) < 0) return r;
    }
    if ((r = s.up<int32_t>(&P.minW, nullptr, 2 * n)) < 0) return r;
    if ((r = s.up<int32_t>(&P.state, nullptr, 6 * nq)) < 0) return r;
    if ((r = s.up<int32_t>(&P.qStart, nullptr, nq)) < 0) return r;
    if ((r = s.up<int32_t>(&P.qCount, nullptr, nq)) < 0) return r;
    P.qArea = nullptr;
    for (size_t q = 0; a->qflags && q < nq && !P.qArea; q++)
        if ((a->qflags[q] & 4) && (r = s.up<int32_t>(&P.qArea, nullptr, nq)) < 0) return r;
    J.keyCap = PROJ_QUOTA * nq + std::max<size_t>(PROJ_QUOTA * nq, 1 << 15); // (the queries' own stretches + overflow)
    if ((r = s.up<unsigned long long>(&P.rawKeys, nullptr, J.keyCap)) < 0) return r;
    if ((r = s.up<unsigned long long>(&P.sortedKeys, nullptr, J.keyCap)) < 0) return r;
    P.keyCap = (int)J.keyCap;
    J.sweepBytes = (2 * n + 6 * nq) * sizeof(int32_t);
    P.sweepLds = J.sweepBytes <= 60 * 1024 ? 1 : 0;
    if (!P.sweepLds) J.sweepBytes = 0;
    return 0;
}

// host tail of one search: outputs from the downloaded block, then the rotation histogram of :2307-2323 /
// :2397-2416 over the matches in query order
int proj_finish(const orbfe_proj_args* a, const int32_t* out, int32_t* q_match, int32_t* feat_match)
{
    const size_t n = (size_t)a->n, nq = (size_t)a->nq;
    int nmatches = out[0];
    std::copy(out + 4, out + 4 + nq, q_match);
    std::copy(out + 4 + nq, out + 4 + nq + n, feat_match);
    if (a->mode == 1 && a->check_orientation) {
        std::vector<int8_t> bins(nq, -1);
        int histo[HISTO_LENGTH] = {0};
        for (size_t q = 0; q < nq; q++)
            if (q_match[q] >= 0) {
                float rot = a->qangle[q] - a->angle[q_match[q]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                bins[q] = (int8_t)bin;
                if (bin >= 0 && bin < HISTO_LENGTH) histo[bin]++;
            }
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (size_t q = 0; q < nq; q++)
            if (q_match[q] >= 0 && bins[q] != ind1 && bins[q] != ind2 && bins[q] != ind3) {
                feat_match[q_match[q]] = -1;
                nmatches--;
            }
    }
    return nmatches;
}
} // namespace

namespace {
// `frame`: the one search (count == 1) runs against a resident frame: its arrays and grid are not staged or rebuilt
int proj_run(int device, const orbfe_proj_args* items, int count, int32_t* const* q_match, int32_t* const* feat_match,
             int32_t* nmatches, const orbfe_frame* const* frames /* per search, or NULL; entries may be NULL and may repeat */)
{
    if (count < 0 || (count && (!items || !q_match || !feat_match || !nmatches))) return ORBFE_ERR_ARGS;
    const orbfe_frame* const frame = (frames && count == 1) ? frames[0] : nullptr; // the latency path of ONE resident search
    PTR_BEGIN();
    int r;
    for (int k = 0; k < count; k++)
        if ((r = proj_validate(&items[k], q_match[k], feat_match[k])) < 0) return r;
    // searches with nothing to do are answered here; the others become device jobs
    std::vector<int> live;
    for (int k = 0; k < count; k++) {
        const orbfe_proj_args* a = &items[k];
        nmatches[k] = 0;
        if (a->n > 0 && a->nq > 0) {
            live.push_back(k); // (proj_finish writes both arrays whole)
            continue;
        }
        for (int i = 0; i < a->n; i++) feat_match[k][i] = -1;
        for (int q = 0; q < a->nq; q++) q_match[k][q] = -1;
    }
    if (live.empty()) return 0;
    PTR(); // validate + prefill
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    std::vector<ProjJob> jobs(live.size());
    // latency path: ONE search against a resident frame (what Tracking issues per frame).  Only the queries travel: the kernels
    // read them from the pinned staging in place, the last kernel copies the results into the pinned mirror and publishes the
    // completion word -- no upload command, no download command, no stream synchronisation
    const bool latency = frame && live.size() == 1;
    if (latency) s.inPlace = (size_t)items[live[0]].nq * 72 + (size_t)items[live[0]].n * 9 + 4096 <= inplace_limit();
    size_t outInts = 0, sweepBytes = 0;
    unsigned maxBlocks = 1;
    for (size_t j = 0; j < jobs.size(); j++) // (the inputs of all searches first: one upload for the batch)
        if ((r = proj_stage(s, &items[live[j]], jobs[j], frames ? frames[live[j]] : nullptr, 0, j ? &items[live[j - 1]] : nullptr,
                            j ? &jobs[j - 1] : nullptr)) < 0)
            return r;
    for (size_t j = 0; j < jobs.size(); j++) {
        const orbfe_proj_args* a = &items[live[j]];
        if ((r = proj_stage(s, a, jobs[j], frames ? frames[live[j]] : nullptr, 1)) < 0) return r;
        jobs[j].outOff = outInts;
        outInts += 4 + (size_t)a->nq + (size_t)a->n;
        sweepBytes = std::max(sweepBytes, jobs[j].sweepBytes);
        maxBlocks = std::max(maxBlocks, (unsigned)((a->nq + 3) / 4));
    }
    int32_t* dOut;
    if ((r = s.up<int32_t>(&dOut, nullptr, outInts)) < 0) return r;
    for (size_t j = 0; j < jobs.size(); j++) {
        const orbfe_proj_args* a = &items[live[j]];
        jobs[j].P.status = dOut + jobs[j].outOff;
        jobs[j].P.qMatch = jobs[j].P.status + 4;
        jobs[j].P.featMatch = jobs[j].P.qMatch + a->nq;
    }
    int32_t *dMir = nullptr, *hMir = nullptr;
    bool anyInorder = false;
    for (const ProjJob& J : jobs) anyInorder = anyInorder || J.P.inorder;
    const bool mirrored = latency && s.inPlace && !g_timeKernels && !anyInorder && outInts * 4 <= (256u << 10) &&
                          s.mirror_out(&dMir, &hMir, outInts) == 0;
    // (kept by the thread: as a fresh vector the download buffer of a 64-search call -- 600 KB -- is mapped, zeroed, faulted in
    // and unmapped by every call)
    static thread_local std::vector<int32_t> outKeep;
    std::vector<int32_t>& outv = outKeep;
    if (!mirrored && outv.size() < outInts) outv.resize(outInts);
    const int32_t* out = mirrored ? hMir : outv.data();
    std::vector<ProjDev> hostP(jobs.size());
    PTR(); // staging
    for (int attempt = 0;; attempt++) {
        ProjDev* dP = nullptr;
        if (jobs.size() > 1) {
            for (size_t j = 0; j < jobs.size(); j++) hostP[j] = jobs[j].P;
            if ((r = s.up(&dP, hostP.data(), hostP.size())) < 0) return r;
        }
        DoneSig done{nullptr, nullptr, 0u, 0u, 0u, nullptr, nullptr, 0u};
        if (mirrored) {
            done = s.flag_only(); // (one workgroup writes the mirror and publishes: neither block nor counter)
            jobs[0].P.mirror = dMir;
            jobs[0].P.mirrorInts = (int)outInts;
            jobs[0].P.doneFlag = done.flag;
            jobs[0].P.doneSeq = done.seq;
        }
        {
            KernelTimer timer(s);
            if (jobs.size() == 1) {
                const ProjDev& P = jobs[0].P;
                if (frame) HIP_TRY(hipMemsetAsync(P.status, 0, 4 * sizeof(int32_t), g_ms)); // (what k_proj_grid resets)
                else hipLaunchKernelGGL(k_proj_grid, dim3(1), dim3(PROJ_THREADS), 0, g_ms, P);
                hipLaunchKernelGGL(k_proj_candidates, dim3(maxBlocks), dim3(256), 0, g_ms, P);
                hipLaunchKernelGGL(k_proj_sweeps, dim3(1), dim3(PROJ_THREADS), sweepBytes, g_ms, P);
            } else {
                const unsigned nj = (unsigned)jobs.size();
                hipLaunchKernelGGL(k_proj_grid_batch, dim3(1, nj), dim3(PROJ_THREADS), 0, g_ms, dP);
                hipLaunchKernelGGL(k_proj_candidates_batch, dim3(maxBlocks, nj), dim3(256), 0, g_ms, dP);
                hipLaunchKernelGGL(k_proj_sweeps_batch, dim3(1, nj), dim3(PROJ_THREADS), sweepBytes, g_ms, dP);
            }
        }
        HIP_TRY(hipGetLastError());
        PTR(); // flush + launches
        if (mirrored) INT_TRY(s.complete(done));
        else {
            INT_TRY(s.down(outv.data(), dOut, outInts * 4));
            INT_TRY(s.fetch());
        }
        PTR(); // fetch
        // more candidate keys than a job's buffers hold: the kernel reported how many it needs; run again
        bool again = false;
        for (ProjJob& J : jobs) {
            if (out[J.outOff + 2] < 0) return ORBFE_ERR_STATE;
            const size_t need = (size_t)PROJ_QUOTA * (size_t)J.P.nq + (size_t)out[J.outOff + 2]; // (own stretches + overflow asked for)
            if (need > J.keyCap) {
                if (attempt > 0) return ORBFE_ERR_STATE;
                J.keyCap = need;
                if ((r = s.up<unsigned long long>(&J.P.rawKeys, nullptr, J.keyCap)) < 0) return r;
                if ((r = s.up<unsigned long long>(&J.P.sortedKeys, nullptr, J.keyCap)) < 0) return r;
                J.P.keyCap = (int)J.keyCap;
                again = true;
            }
        }
        if (!again) break;
    }
    for (size_t j = 0; j < jobs.size(); j++) {
        const int k = live[j];
        g_lastProjSweeps = out[jobs[j].outOff + 1];
        nmatches[k] = proj_finish(&items[k], out + jobs[j].outOff, q_match[k], feat_match[k]);
    }
    PTR();
#ifdef ORBFE_CALL_TRACE
    if (getenv("ORBFE_CALL_TRACE")) fprintf(stderr, "proj_run count=%d: validate %.1f stage %.1f launch %.1f fetch %.1f finish %.1f us\n", count, trT[0], trT[1], trT[2], trT[3], trT[4]);
#endif
    return 0;
}

} // namespace

int orbfe_search_projection_batch(int device, const orbfe_proj_args* items, int count, int32_t* const* q_match,
                                  int32_t* const* feat_match, int32_t* nmatches)
{
    return proj_run(device, items, count, q_match, feat_match, nmatches, nullptr);
}

int orbfe_frame_create(orbfe_frame** out, int device, const orbfe_proj_args* a)
{
    if (!out) return ORBFE_ERR_ARGS;
    *out = nullptr;
    if (!a || a->n < 1 || a->n >= PROJ_MAXN || !a->desc || !a->kx || !a->ky || !a->octave) return ORBFE_ERR_ARGS;
    if (a->Nleft != -1 && (a->Nleft < 0 || a->Nleft > a->n)) return ORBFE_ERR_ARGS;
    if (is_device_ptr(a->kx) || is_device_ptr(a->octave)) return ORBFE_ERR_ARGS; // (only the descriptors may be resident already)
    int r;
    if ((r = select_device(device)) < 0) return r;
    const size_t n = (size_t)a->n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oDesc = 0, oKx = oDesc + al(n * 32), oKy = oKx + al(n * 4), oOct = oKy + al(n * 4), oUr = oOct + al(n * 4),
                 oCs = oUr + al(n * 4), oCi = oCs + al((2 * PROJ_CELLS + 1) * 4), oCo = oCi + al(n * 4), oSt = oCo + al(n * 4),
                 total = oSt + 256;
    size_t blkCap = 0;
    void* blk = g_blockPool.get(device, total, &blkCap);
    if (!blk) return -(1000 + (int)hipErrorOutOfMemory);
    orbfe_frame* F = new orbfe_frame();
    F->device = device; F->n = a->n; F->Nleft = a->Nleft;
    F->minX = a->minX; F->minY = a->minY; F->wInv = a->gridWInv; F->hInv = a->gridHInv;
    F->block = (uint8_t*)blk;
    F->blockCap = blkCap;
    F->desc = F->block + oDesc;
    F->kx = (float*)(F->block + oKx); F->ky = (float*)(F->block + oKy);
    F->octave = (int32_t*)(F->block + oOct);
    F->uright = a->uright ? (float*)(F->block + oUr) : nullptr;
    F->cellStart = (int32_t*)(F->block + oCs); F->cellItems = (int32_t*)(F->block + oCi); F->cellOf = (int32_t*)(F->block + oCo);
    F->status = (int32_t*)(F->block + oSt);
    F->hOctave.assign(a->octave, a->octave + n);
    if (a->angle) F->hAngle.assign(a->angle, a->angle + n);
    Scratch s(device); // (this thread's matcher stream)
    const bool descResident = is_device_ptr(a->desc);
    if (descResident) {
        if (int w = orbfe_producer_wait(a->desc, g_ms); w < 0) return w;
    }
    // (staged in the block's own layout and sent as ONE upload, like orbfe_keyframe_create)
    hipError_t e = hipSuccess;
    const size_t first = descResident ? oKx : 0, upTo = oCs; // desc | kx | ky | octave | uright lie in front of the grid arrays
    uint8_t* st = s.pin_scratch(upTo - first);
    if (st) {
        uint8_t* const b = st - first;
        if (!descResident) std::memcpy(b + oDesc, a->desc, n * 32);
        std::memcpy(b + oKx, a->kx, n * 4);
        std::memcpy(b + oKy, a->ky, n * 4);
        std::memcpy(b + oOct, a->octave, n * 4);
        if (F->uright) std::memcpy(b + oUr, a->uright, n * 4);
        if (descResident) e = hipMemcpyAsync(F->desc, a->desc, n * 32, hipMemcpyDeviceToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(F->block + first, st, upTo - first, hipMemcpyHostToDevice, g_ms);
    } else {
        e = hipMemcpyAsync(F->desc, a->desc, n * 32, descResident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(F->kx, a->kx, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(F->ky, a->ky, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess) e = hipMemcpyAsync(F->octave, a->octave, n * 4, hipMemcpyHostToDevice, g_ms);
        if (e == hipSuccess && F->uright) e = hipMemcpyAsync(F->uright, a->uright, n * 4, hipMemcpyHostToDevice, g_ms);
    }
    if (e == hipSuccess) {
        ProjDev P{};
        P.kx = F->kx; P.ky = F->ky; P.n = F->n; P.Nleft = F->Nleft;
        P.minX = F->minX; P.minY = F->minY; P.wInv = F->wInv; P.hInv = F->hInv;
        P.cellStart = F->cellStart; P.cellItems = F->cellItems; P.cellOf = F->cellOf; P.status = F->status;
        hipLaunchKernelGGL(k_proj_grid, dim3(1), dim3(PROJ_THREADS), 0, g_ms, P);
        e = hipGetLastError();
    }
    // the handle is complete (and may serve other threads) when the grid is built.  (Waiting on a completion word published by
    // the grid kernel instead was tried: 0.036 against 0.029 ms -- the next creation's upload command then queues behind a
    // kernel the runtime still holds as running, the case the search calls avoid by reading their inputs in place.)
    if (e == hipSuccess) e = hipStreamSynchronize(g_ms);
    if (e != hipSuccess) {
        g_blockPool.put(device, blk, blkCap);
        delete F;
        return -(1000 + (int)e);
    }
    *out = F;
    return 0;
}

void orbfe_frame_destroy(orbfe_frame* F)
{
    if (!F) return;
    // (the block goes back to the pool: the handle is no longer in use by any thread -- its searches have returned --, and a
    // search's kernel has done all its reads before the search returns, with the completion word as without it)
    g_blockPool.put(F->device, F->block, F->blockCap);
    delete F;
}

int orbfe_search_projection_frame(orbfe_frame* F, const orbfe_proj_args* a, int32_t* q_match, int32_t* feat_match)
{
    if (!F || !a) return ORBFE_ERR_ARGS;
    orbfe_proj_args b = *a; // the frame side comes from the handle; taken / stereo partners / queries from the caller
    b.n = F->n; b.Nleft = F->Nleft;
    b.desc = F->desc; b.kx = F->kx; b.ky = F->ky;
    b.octave = F->hOctave.data();
    b.angle = F->hAngle.empty() ? nullptr : F->hAngle.data();
    b.uright = F->uright;
    b.minX = F->minX; b.minY = F->minY; b.gridWInv = F->wInv; b.gridHInv = F->hInv;
    int32_t nm = 0;
    int32_t* qm[1] = {q_match};
    int32_t* fm[1] = {feat_match};
    const orbfe_frame* fr[1] = {F};
    const int r = proj_run(F->device, &b, 1, qm, fm, &nm, fr);
    return r < 0 ? r : (int)nm;
}

// Many searches against RESIDENT frame sides in one upload, three launches, one download (round 5): the
// candidate keyframes' map points against the one current frame of a relocalisation (src/Tracking.cc:3846-3870: every entry
// names the same handle), or one keyframe's points fused into every neighbour that has a handle.  Only the queries, `taken` and
// the partner tables travel; no grid is rebuilt.
int orbfe_search_projection_frames(orbfe_frame* const* frames, const orbfe_proj_args* queries, int count, int32_t* const* q_match,
                                   int32_t* const* feat_match, int32_t* nmatches)
{
    if (count < 0 || (count && (!frames || !queries || !q_match || !feat_match || !nmatches))) return ORBFE_ERR_ARGS;
    if (count == 0) return 0;
    std::vector<orbfe_proj_args> b(queries, queries + count);
    for (int k = 0; k < count; k++) {
        const orbfe_frame* F = frames[k];
        if (!F || F->device != frames[0]->device) return ORBFE_ERR_ARGS;
        b[k].n = F->n; b[k].Nleft = F->Nleft;
        b[k].desc = F->desc; b[k].kx = F->kx; b[k].ky = F->ky;
        b[k].octave = F->hOctave.data();
        b[k].angle = F->hAngle.empty() ? nullptr : F->hAngle.data();
        b[k].uright = F->uright;
        b[k].minX = F->minX; b[k].minY = F->minY; b[k].gridWInv = F->wInv; b[k].gridHInv = F->hInv;
    }
    return proj_run(frames[0]->device, b.data(), count, q_match, feat_match, nmatches, frames);
}

int orbfe_search_projection(int device, const orbfe_proj_args* a, int32_t* q_match, int32_t* feat_match)
{
    int32_t nm = 0;
    int32_t* qm[1] = {q_match};
    int32_t* fm[1] = {feat_match};
    const int r = orbfe_search_projection_batch(device, a, a ? 1 : 0, qm, fm, &nm);
    if (!a) return ORBFE_ERR_ARGS;
    return r < 0 ? r : (int)nm;
}

int orbfe_search_projection_last_sweeps(void) { return g_lastProjSweeps; }

int orbfe_distinctive_descriptors(int device, const uint8_t* pool, const int32_t* offsets, int npts, int32_t* best)
{
    if (npts < 0 || (npts && (!offsets || !best))) return ORBFE_ERR_ARGS;
    if (npts == 0) return 0;
    const int total = offsets[npts];
    if (offsets[0] != 0 || total < 0 || (total && !pool)) return ORBFE_ERR_ARGS;
    for (int p = 0; p < npts; p++)
        if (offsets[p + 1] < offsets[p] || offsets[p + 1] - offsets[p] >= (1 << 20)) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t* dP;
    int32_t *dO, *dB;
    if ((r = s.up_desc(&dP, pool, (size_t)total * 32)) < 0) return r;
    if ((r = s.up(&dO, offsets, (size_t)npts + 1)) < 0) return r;
    if ((r = s.up<int32_t>(&dB, nullptr, (size_t)npts)) < 0) return r;
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_distinctive, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, g_ms, dP, dO, npts, dB);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(best, dB, (size_t)npts * 4));
    INT_TRY(s.fetch());
    return 0;
}

float orbfe_matcher_last_kernel_ms(void) { return g_lastKernelMs; }
void orbfe_matcher_time_kernels(int on) { g_timeKernels = on != 0; }
#ifdef ORBFE_KB8_TIMING
extern "C" int orbfe_debug_kb8_times(unsigned long long* out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_kb8Times), sizeof(g_kb8Times)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ORBFE_PROJ_TIMING
extern "C" int orbfe_debug_proj_times(unsigned long long* out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_projTimes), sizeof(g_projTimes)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ORBFE_BOW_TIMING
extern "C" int orbfe_debug_bow_times(unsigned long long* out8 /* 16 */, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bowTimes), sizeof(g_bowTimes)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        z[11] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bowTimes), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

struct orbfe_vocab_dev {
    int device, nnodes, L;
    int weighting = 0, scoring = 0; // WeightingType / ScoringType (BowVector.h:39-56); ORBvoc.txt: TF_IDF, L1_NORM
    uint8_t* desc;
    int32_t *childOff, *childIds, *word;
    double* weight;
};

int orbfe_vocab_upload(orbfe_vocab_dev** out, int device, const orbfe_vocab* v)
{
    if (!out || !v || v->nnodes < 1 || !v->node_desc || !v->child_off || !v->node_word || !v->node_weight || v->L < 1)
        return ORBFE_ERR_ARGS;
    *out = nullptr;
    const int nchild = v->child_off[v->nnodes];
    if (nchild < 0 || (nchild && !v->child_ids)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < v->nnodes; i++)
        if (v->child_off[i] > v->child_off[i + 1] || v->child_off[i + 1] - v->child_off[i] >= (1 << 20)) return ORBFE_ERR_ARGS;
    for (int i = 0; i < nchild; i++)
        if (v->child_ids[i] <= 0 || v->child_ids[i] >= v->nnodes) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(device)) < 0) return r;
    orbfe_vocab_dev* d = new orbfe_vocab_dev();
    d->device = device;
    d->nnodes = v->nnodes;
    d->L = v->L;
    bool ok = hipMalloc((void**)&d->desc, (size_t)v->nnodes * 32) == hipSuccess &&
              hipMalloc((void**)&d->childOff, (size_t)(v->nnodes + 1) * 4) == hipSuccess &&
              hipMalloc((void**)&d->childIds, (size_t)std::max(nchild, 1) * 4) == hipSuccess &&
              hipMalloc((void**)&d->word, (size_t)v->nnodes * 4) == hipSuccess &&
              hipMalloc((void**)&d->weight, (size_t)v->nnodes * 8) == hipSuccess;
    ok = ok && hipMemcpy(d->desc, v->node_desc, (size_t)v->nnodes * 32, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d->childOff, v->child_off, (size_t)(v->nnodes + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (nchild == 0 || hipMemcpy(d->childIds, v->child_ids, (size_t)nchild * 4, hipMemcpyHostToDevice) == hipSuccess) &&
         hipMemcpy(d->word, v->node_word, (size_t)v->nnodes * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d->weight, v->node_weight, (size_t)v->nnodes * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        orbfe_vocab_free(d);
        return ORBFE_ERR_NODEV;
    }
    *out = d;
    return 0;
}

// TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423): first line
// "k L scoring weighting", then one line per node in id order (ids from 1; node 0 is the root): "parent isLeaf
// d0 ... d31 weight".  Children keep their file order, word ids are handed out to the leaves in file order -- as the
// reference builds m_nodes / m_words.  The tree goes straight to the device.
int orbfe_vocab_load_text(orbfe_vocab_dev** out, int device, const char* path, int* k_out, int* L_out, int* nwords_out)
{
    if (!out || !path) return ORBFE_ERR_ARGS;
    *out = nullptr;
    FILE* f = fopen(path, "r");
    if (!f) return ORBFE_ERR_ARGS;
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 ||
        n2 > 3) { // the reference's own sanity test (:1357-1361)
        fclose(f);
        return ORBFE_ERR_ARGS;
    }
    std::vector<uint8_t> desc(32, 0);
    std::vector<int32_t> parent(1, -1), word(1, -1);
    std::vector<double> weight(1, 0.0);
    int nwords = 0;
    for (;;) {
        int pid = 0, leaf = 0;
        if (fscanf(f, "%d %d", &pid, &leaf) != 2) break; // end of file (or a trailing blank line)
        const int nid = (int)parent.size();
        uint8_t d[32];
        bool ok = pid >= 0 && pid < nid;
        for (int i = 0; i < 32 && ok; i++) {
            int v = 0;
            ok = fscanf(f, "%d", &v) == 1 && v >= 0 && v <= 255;
            d[i] = (uint8_t)v;
        }
        double w = 0;
        ok = ok && fscanf(f, "%lf", &w) == 1;
        if (!ok) {
            fclose(f);
            return ORBFE_ERR_ARGS;
        }
        parent.push_back(pid);
        desc.insert(desc.end(), d, d + 32);
        weight.push_back(w);
        word.push_back(leaf > 0 ? nwords++ : -1);
    }
    fclose(f);
    const int nn = (int)parent.size();
    if (nn < 2) return ORBFE_ERR_ARGS;
    // children lists in file order -> CSR
    std::vector<int32_t> childOff(nn + 1, 0), childIds(nn - 1), fill(nn, 0);
    for (int i = 1; i < nn; i++) childOff[parent[i] + 1]++;
    for (int i = 0; i < nn; i++) childOff[i + 1] += childOff[i];
    for (int i = 1; i < nn; i++) childIds[childOff[parent[i]] + fill[parent[i]]++] = i;
    orbfe_vocab v;
    v.nnodes = nn;
    v.node_desc = desc.data();
    v.child_off = childOff.data();
    v.child_ids = childIds.data();
    v.node_word = word.data();
    v.node_weight = weight.data();
    v.L = L;
    if (k_out) *k_out = k;
    if (L_out) *L_out = L;
    if (nwords_out) *nwords_out = nwords;
    const int r = orbfe_vocab_upload(out, device, &v);
    if (r == 0) { // the header's "scoring weighting" (:1366-1368: m_scoring = n1, m_weighting = n2)
        (*out)->scoring = n1;
        (*out)->weighting = n2;
    }
    return r;
}

void orbfe_vocab_free(orbfe_vocab_dev* d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->desc);
    (void)hipFree(d->childOff);
    (void)hipFree(d->childIds);
    (void)hipFree(d->word);
    (void)hipFree(d->weight);
    delete d;
}

int orbfe_vocab_transform(orbfe_vocab_dev* d, const uint8_t* feats, int n, int levelsup, int32_t* word_id,
                          int32_t* node_id, double* weight)
{
    if (!d || n < 0 || (n && (!feats || !word_id || !node_id || !weight))) return ORBFE_ERR_ARGS;
    if (n == 0) return 0;
    int r;
    if ((r = select_device(d->device)) < 0) return r;
    Scratch s(d->device);
    uint8_t* dF;
    int32_t *dW, *dN;
    double* dWt;
    // latency path (Frame::ComputeBoW of one frame): descriptors read in place, the three result arrays written into the
    // pinned mirror by the kernel, the completion word instead of a download and a stream synchronisation
    s.inPlace = (size_t)n * 32 <= inplace_limit();
    const unsigned wgs = (unsigned)((n * 16 + 255) / 256);
    if ((r = s.up_desc(&dF, feats, (size_t)n * 32)) < 0) return r;
    Scratch::OutBlock ob;
    const size_t iBytes = ((size_t)n * 8 + 15) & ~(size_t)15; // word ids | node ids, then the weights (8-byte aligned)
    const bool mirrored = (size_t)n * 16 <= (256u << 10) && s.out_block(&ob, iBytes + (size_t)n * 8, wgs) == 0;
    if (mirrored) {
        dW = reinterpret_cast<int32_t*>(ob.dev);
        dN = dW + n;
        dWt = reinterpret_cast<double*>(ob.dev + iBytes);
    } else {
        if ((r = s.up<int32_t>(&dW, nullptr, (size_t)n)) < 0) return r;
        if ((r = s.up<int32_t>(&dN, nullptr, (size_t)n)) < 0) return r;
        if ((r = s.up<double>(&dWt, nullptr, (size_t)n)) < 0) return r;
    }
    const DoneSig done = s.done_sig(4u * wgs, mirrored ? &ob : nullptr, g_timeKernels);
    {
        KernelTimer timer(s);
        hipLaunchKernelGGL(k_vocab_transform, dim3(wgs), dim3(256), 0, g_ms, d->desc, d->childOff, d->childIds, d->word, d->weight, d->L,
                           dF, n, levelsup, dW, dN, dWt, done);
    }
    HIP_TRY(hipGetLastError());
    if (mirrored) {
        INT_TRY(s.complete(done));
        std::memcpy(word_id, ob.host, (size_t)n * 4);
        std::memcpy(node_id, ob.host + (size_t)n * 4, (size_t)n * 4);
        std::memcpy(weight, ob.host + iBytes, (size_t)n * 8);
        return 0;
    }
    INT_TRY(s.down(word_id, dW, (size_t)n * 4));
    INT_TRY(s.down(node_id, dN, (size_t)n * 4));
    INT_TRY(s.down(weight, dWt, (size_t)n * 8));
    INT_TRY(s.fetch());
    return 0;
}

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
