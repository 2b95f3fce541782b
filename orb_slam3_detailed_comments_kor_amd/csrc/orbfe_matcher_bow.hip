// orbfe_matcher_bow.hip -- K-BOW (SearchByBoW), the completion word (DoneSig), k_bow_cull: kernels.
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
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
    // (Where the results go: straight into the pinned mirror, the kernel's own stores.  Round 4 also had a form that scattered them
    // into a clean device block which the last wavefront copied out; measured within a microsecond of this one for large grids and
    // 2-9 us slower for small calls, it left with round 6's pruning -- DESIGN_HISTORY.md 7.4.)
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
    // this wavefront's result stores (into the mirror itself): acknowledged
    own_stores_acknowledged();
    unsigned closes = 0u, last = 0u;
    if ((threadIdx.x & 63) == 0) {
        const unsigned mine = min(4u, d.waves - 4u * blockIdx.x);
        closes = atomicAdd(wgCnt, 1u) + 1u == mine ? 1u : 0u;
    }
    if (!__builtin_amdgcn_readfirstlane(closes)) return;
    workgroup_stores_landed();
    if ((threadIdx.x & 63) == 0) last = atomicAdd(d.ctr, 1u) + 1u == d.total ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last)) done_publish(d);
}
// ... and for a kernel in which ONE wavefront per workgroup reports (d.total = workgroups)
__device__ __forceinline__ void wg1_done(const DoneSig& d)
{
    if (!d.flag) return;
    workgroup_stores_landed(); // (the reporting wavefront is the workgroup's only writer, or stands behind its barrier)
    unsigned last = 0u;
    if ((threadIdx.x & 63) == 0) last = atomicAdd(d.ctr, 1u) + 1u == d.total ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last)) done_publish(d);
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
                                                    uint8_t* __restrict__ takenPool, const DoneSig done)
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
    const int roundCap = N.n1 + 2;
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

