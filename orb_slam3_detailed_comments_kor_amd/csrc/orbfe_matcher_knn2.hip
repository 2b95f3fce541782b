// orbfe_matcher_knn2.hip -- K-HAM, K-BFKNN2 / K-KNN2F (vector pipe and MFMA): kernels.
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
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

