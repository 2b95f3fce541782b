/*
 * orbfe_matcher_bowvec.hip -- Frame::ComputeBoW / KeyFrame::ComputeBoW as a device-resident step (round 6; SURVEY.md 8f rank 3,
 * VERDICT r05 missing #2).  Part of the matcher's translation unit (included by orbfe_matcher.hip behind K-VOC and the
 * vocabulary handle): it uses the calling thread's matcher stream, k_vocab_transform and struct orbfe_vocab_dev.
 *
 * Reference: TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1192), BowVector::addWeight / addIfNotExist / normalize
 * (BowVector.cpp:34-86), FeatureVector::addFeature (FeatureVector.cpp:31-45); callers src/Frame.cc:724-731,
 * src/KeyFrame.cc:105-114 (levelsup = 4).
 *
 * What the reference builds, one feature after the other, are two ordered maps:
 *   BowVector      word id -> sum of the word's weight over the features that fell into it, in feature order, then L1 (L2)
 *                  normalised over the words in ascending id order;
 *   FeatureVector  node id -> the indices of its features in ascending order.
 * Both are a SORT of the per-feature results of K-VOC by (id, feature index) followed by a segmentation, and that is how they
 * are built here -- two launches on the caller's matcher stream, no host involvement, no copy command:
 *   K-VOC      (k_vocab_transform)  word, node `levelsup` levels above the leaves, weight per feature;
 *   K-BOWRANK  (k_bow_rank_fold, every wavefront)  one wavefront per kept feature (weight > 0: "not stopped", :1157): its rank among the kept
 *                                   features by (node, index) and by (word, index), counted across the lanes -- n / 64 steps per
 *                                   wavefront, n wavefronts: the whole sort is a few microseconds for the 1000-2000 features of a
 *                                   frame and needs neither LDS nor a size limit; rank = final position, so the FeatureVector's
 *                                   index array is complete after this launch;
 *   K-BOWFOLD  (k_bow_rank_fold, the LAST workgroup to finish)  segment heads of both sorted id lists (a prefix sum), node ids / offsets, word
 *                                   ids, and the word values in the reference's OWN arithmetic: the c-fold sequential double sum
 *                                   w + w + ... of addWeight (not c * w: the roundings differ), the sequential sum of |value| in
 *                                   ascending word order of normalize (one lane, a dependent chain of ~1000 v_add_f64: 4 us),
 *                                   IEEE division.  Bit-identical to the maps of the reference for every weighting / scoring
 *                                   type (tests/test_gpu_bow.py).
 * The results stay on the device in exactly the arrays the searches read (orbfe_fv layout: node ids, offsets, indices) and are
 * mirrored into page-locked host memory by the kernel's own stores; orbfe_bow_host waits for the kernel -- the
 * "host copy on request".  A SearchByBoW against keyframe handles takes the FeatureVector from the handle itself
 * (orbfe_bow_fv), without the host ever seeing it: extract -> ComputeBoW -> SearchByBoW x 64 runs without a host round trip
 * between its stages (bow_run, "resident FeatureVector").
 */

// ---------------------------------------------------------------- K-BOWRANK + K-BOWFOLD (one launch)
// exclusive prefix sum of one int per thread over a 256-thread workgroup; *total = the sum (same in every thread)
__device__ __forceinline__ int bow_block_scan(int v, int* sWave /* 4 ints */, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int s = sWave[w];
        if (w < wave) before += s;
        all += s;
    }
    __syncthreads(); // (sWave is reused by the next scan)
    *total = all;
    return before + incl - v;
}

// agent-scope relaxed stores / loads (global_store / global_load with sc1: coherent across the XCDs' L2s without a fence)
__device__ __forceinline__ void bow_store_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bow_store_agent64(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t bow_load_agent(const uint32_t* p)
{
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long bow_load_agent64(const unsigned long long* p)
{
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef ORBFE_BOWVEC_TIMING // tuning only (make EXTRA=-DORBFE_BOWVEC_TIMING): where k_bow_rank_fold spends its time, 100-MHz ticks
__device__ unsigned long long g_bowvecTimes[16]; // [0] first wavefront start (min), [1] last counter increment (max), [2..] fold stamps
#define BV_STAMP(k)                                              \
    do {                                                         \
        if (threadIdx.x == 0) g_bowvecTimes[k] = wall_clock64(); \
    } while (0)
#else
#define BV_STAMP(k) do { } while (0)
#endif
#define BOW_LDS 2048 /* kept features up to which the fold works in LDS: sorted ids and weights, segment tables, values (56 KB) */
struct BowFoldArgs {
    const uint2* keys;     // K-VOC: (node, word) per feature, (~0, ~0) for a stopped one
    const double* weight;  // K-VOC: the word's weight per feature
    int n;
    uint32_t* sortedNode; // [m] node id by rank
    uint32_t* sortedWord; // [m] word id by rank
    double* sortedWt;     // [m] the word's weight by rank
    int32_t* hdr;         // [8]: kept features m, nodes nn, words nw, features of the largest node; [7]: the launch's block counter
    uint32_t* nodeIds;    // [cap]
    int32_t* offsets;     // [cap + 1]
    int32_t* indices;     // [cap]
    uint32_t* wordIds;    // [cap]
    double* values;       // [cap]
    int32_t* headPos;     // [cap + 1] scratch: first rank of every word (frames of more than BOW_LDS kept features)
    uint8_t* mirror;      // the kernel's address of the results' page-locked mirror (same layout as the block's result run)
    uint32_t oHdr, oNode, oOffs, oInd, oWid, oVal; // byte offsets of the result arrays inside the run / the mirror
    int addWeight;        // TF_IDF / TF: BowVector::addWeight; IDF / BINARY: addIfNotExist
    int must;             // the scoring object normalises (all but DOT_PRODUCT)
    int normL2;           // ... with the L2 norm (L2_NORM)
    int lazyNorm;         // leave BowVector::normalize to the host view (orbfe_bow_set_lazy_norm): the values stay un-normalised
};

// Grid: one wavefront per feature (four per workgroup).  Part 1, every wavefront: the feature's rank among the kept features
// by (node, index) and by (word, index), counted across the lanes -- the keys of 256 features per step, all loads of a step
// independent; rank = final position.  Part 2, the LAST workgroup to finish (a counter in the handle's header): segment heads
// of both sorted lists, node ids / offsets, word ids, and the word values in the reference's own arithmetic.  Every result is
// stored twice -- device array and page-locked mirror -- so that no copy command follows the kernel (a copy engine command
// behind a kernel costs 10-15 us of queue hand-over on this chip, more than the kernel).
__global__ __launch_bounds__(256) void k_bow_rank_fold(const BowFoldArgs A)
{
    __shared__ int sWave[4];
    __shared__ int sMax, sLast;
    __shared__ double sNorm;
    __shared__ double sVal[BOW_LDS], sWt[BOW_LDS];
    __shared__ uint32_t sNode[BOW_LDS], sWord[BOW_LDS];
    __shared__ uint16_t sHead[BOW_LDS + 2], sOff[BOW_LDS + 2];
    const int t = (int)threadIdx.x, lane = t & 63;
    const int n = A.n;
#ifdef ORBFE_BOWVEC_TIMING
    if (t == 0) atomicMin(&g_bowvecTimes[0], (unsigned long long)wall_clock64());
#endif
    {
        const int i = (int)blockIdx.x * 4 + (t >> 6);
        const uint2 ki = i < n ? A.keys[i] : make_uint2(~0u, ~0u);
        if (ki.x != ~0u || ki.y != ~0u) { // (wave-uniform) a stopped word (weight 0) enters neither vector (:1157)
            int rn = 0, rw = 0;
            for (int base = 0; base < n; base += 256) { // (uniform)
                uint2 kj[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = base + 64 * u + lane;
                    kj[u] = j < n ? A.keys[j] : make_uint2(~0u, ~0u);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = base + 64 * u + lane;
                    // (a stopped j holds ~0: never below a kept key; equal to it only for a stopped i, which is not here)
                    rn += (kj[u].x < ki.x || (kj[u].x == ki.x && j < i)) ? 1 : 0;
                    rw += (kj[u].y < ki.y || (kj[u].y == ki.y && j < i)) ? 1 : 0;
                }
            }
            rn = wave_sum_i32(rn);
            rw = wave_sum_i32(rw);
            if (lane == 0) {
                // (agent-scope stores: written through to where every XCD sees them.  An agent-scope RELEASE FENCE per workgroup
                // instead -- the textbook form -- writes back the whole L2 of the workgroup's XCD, 252 times for a frame of 1000
                // features, and the write-backs of one XCD queue behind each other: the kernel took 37 us that way, round 6.)
                bow_store_agent(&A.sortedNode[rn], ki.x);
                bow_store_agent(reinterpret_cast<uint32_t*>(&A.indices[rn]), (uint32_t)i);
                bow_store_agent(&A.sortedWord[rw], ki.y);
                bow_store_agent64(reinterpret_cast<unsigned long long*>(&A.sortedWt[rw]), (unsigned long long)__double_as_longlong(A.weight[i]));
            }
        }
    }
    // ---- the last workgroup folds: every wavefront's stores above are acknowledged before its workgroup counts itself, and the
    // folding workgroup reads them with agent-scope loads (no cache of its own between it and them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef ORBFE_BOWVEC_TIMING
    if (t == 0) atomicMax(&g_bowvecTimes[1], (unsigned long long)wall_clock64());
#endif
    if (t == 0) sLast = atomicAdd(reinterpret_cast<unsigned*>(A.hdr + 7), 1u) + 1u == gridDim.x ? 1 : 0;
    __syncthreads();
    if (!sLast) return;
    BV_STAMP(2);
    if (t == 0) {
        A.hdr[7] = 0; // (for the next call: calls on one handle are ordered)
        sMax = 0;
    }
    constexpr int NT = 256;
    int cnt = 0;
    for (int j = t; j < n; j += NT) {
        const uint2 k = A.keys[j];
        cnt += (k.x != ~0u || k.y != ~0u) ? 1 : 0;
    }
    int m = 0;
    (void)bow_block_scan(cnt, sWave, &m);
    BV_STAMP(3);
    // Up to BOW_LDS kept features (every frame ORB-SLAM3 makes: nFeatures 1000-2000) the sorted lists come into LDS in ONE
    // coalesced pass and everything below walks LDS; a larger frame walks the handle's device arrays (same code, `inLds` false).
    const bool inLds = m <= BOW_LDS; // (uniform)
    if (inLds) {
        for (int r = t; r < m; r += NT) {
            sNode[r] = bow_load_agent(&A.sortedNode[r]);
            sWord[r] = bow_load_agent(&A.sortedWord[r]);
            sWt[r] = __longlong_as_double((long long)bow_load_agent64(reinterpret_cast<const unsigned long long*>(&A.sortedWt[r])));
        }
        __syncthreads();
    }
    BV_STAMP(4);
    auto nodeAt = [&](int r) { return inLds ? sNode[r] : bow_load_agent(&A.sortedNode[r]); };
    auto wordAt = [&](int r) { return inLds ? sWord[r] : bow_load_agent(&A.sortedWord[r]); };
    // segment heads of both lists: thread t owns ranks [r0, r1)
    const int C = (m + NT - 1) / NT, r0 = min(m, t * C), r1 = min(m, r0 + C);
    int hn = 0, hw = 0;
    for (int r = r0; r < r1; r++) {
        hn += (r == 0 || nodeAt(r) != nodeAt(r - 1)) ? 1 : 0;
        hw += (r == 0 || wordAt(r) != wordAt(r - 1)) ? 1 : 0;
    }
    int nn = 0, nw = 0;
    int sn = bow_block_scan(hn, sWave, &nn);
    int sw = bow_block_scan(hw, sWave, &nw);
    BV_STAMP(5);
    uint32_t* const mNode = reinterpret_cast<uint32_t*>(A.mirror + A.oNode);
    int32_t* const mOffs = reinterpret_cast<int32_t*>(A.mirror + A.oOffs);
    uint32_t* const mWid = reinterpret_cast<uint32_t*>(A.mirror + A.oWid);
    double* const mVal = reinterpret_cast<double*>(A.mirror + A.oVal);
    for (int r = r0; r < r1; r++) {
        const uint32_t nd = nodeAt(r), wd = wordAt(r);
        // (LDS form: the tables stay in LDS until the end of the kernel -- every store to the device arrays or, worse, across
        // PCIe to the mirror would be waited for by the next barrier, an L2 / PCIe round trip per phase: stamps, 26 -> 18 us)
        if (r == 0 || nd != nodeAt(r - 1)) {
            if (inLds) sOff[sn] = (uint16_t)r;
            else {
                A.nodeIds[sn] = nd;
                A.offsets[sn] = r;
                mNode[sn] = nd;
                mOffs[sn] = r;
            }
            sn++;
        }
        if (r == 0 || wd != wordAt(r - 1)) {
            if (inLds) sHead[sw] = (uint16_t)r;
            else {
                A.wordIds[sw] = wd;
                mWid[sw] = wd;
                A.headPos[sw] = r;
            }
            sw++;
        }
    }
    if (t == 0) {
        if (inLds) {
            sOff[nn] = (uint16_t)m;
            sHead[nw] = (uint16_t)m;
        } else {
            A.offsets[nn] = m;
            mOffs[nn] = m;
            A.headPos[nw] = m;
        }
    }
    __syncthreads(); // (the workgroup's LDS and global stores above are visible to its threads below)
    BV_STAMP(6);
    // word values: BowVector::addWeight adds the word's weight once per feature, in feature order -- c sequential additions.
    // They stay in LDS until they are final: the normalisation below is ONE lane adding them up in order, and every trip to
    // L2 in that chain costs more than the addition.
    for (int s = t; s < nw; s += NT) {
        const int p = inLds ? (int)sHead[s] : A.headPos[s], c = (inLds ? (int)sHead[s + 1] : A.headPos[s + 1]) - p;
        const double w = inLds ? sWt[p] : __longlong_as_double((long long)bow_load_agent64(reinterpret_cast<const unsigned long long*>(&A.sortedWt[p])));
        double v = w;
        if (A.addWeight)
            for (int k = 1; k < c; k++) v = __dadd_rn(v, w);
        if (inLds) sVal[s] = v;
        else A.values[s] = v;
    }
    int mx = 0;
    for (int s = t; s < nn; s += NT) mx = max(mx, inLds ? (int)sOff[s + 1] - (int)sOff[s] : A.offsets[s + 1] - A.offsets[s]);
    if (mx) atomicMax(&sMax, mx);
    __syncthreads();
    BV_STAMP(7);
    double scale = 1.0; // every value is divided by this at the end (1: left as it is)
    if (A.addWeight && !A.must && nw > 0) scale = (double)nw; // "unnecessary when normalizing" (:1164-1170): value / number of words
    if (A.must && !A.lazyNorm) { // BowVector::normalize (BowVector.cpp:62-86): the sum runs over the map in ascending id order, one term at a time
        if (t == 0) {
            double norm = 0.0;
            int s = 0;
            if (inLds) {
                // (v_add_f64 issues at 32 cycles per wavefront instruction on this chip whatever the number of active lanes
                // (profiles/r01_valu_rate.txt): the chain of nw dependent additions is ~16 ns per word -- 7.4 us for 444 words --
                // and no arrangement of the reads changes that; orbfe_bow_set_lazy_norm leaves this sum to the host)
                for (; s + 8 <= nw; s += 8) { // (eight reads in flight, then the dependent additions)
                    double v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = sVal[s + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) norm = __dadd_rn(norm, A.normL2 ? __dmul_rn(v[k], v[k]) : fabs(v[k]));
                }
                for (; s < nw; s++) norm = __dadd_rn(norm, A.normL2 ? __dmul_rn(sVal[s], sVal[s]) : fabs(sVal[s]));
            } else {
                for (; s + 8 <= nw; s += 8) {
                    double v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = A.values[s + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) norm = __dadd_rn(norm, A.normL2 ? __dmul_rn(v[k], v[k]) : fabs(v[k]));
                }
                for (; s < nw; s++) {
                    const double v = A.values[s];
                    norm = __dadd_rn(norm, A.normL2 ? __dmul_rn(v, v) : fabs(v));
                }
            }
            if (A.normL2) norm = __dsqrt_rn(norm);
            sNorm = norm;
        }
        __syncthreads();
        scale = sNorm > 0.0 ? sNorm : 1.0; // (norm > 0.0 or the values stay, BowVector.cpp:80)
    }
    BV_STAMP(8);
    const bool divide = (A.must ? (!A.lazyNorm && sNorm > 0.0) : (A.addWeight && nw > 0));
    for (int s = t; s < nw; s += NT) { // final value: device array and mirror (each thread finishes the words it made)
        double v = inLds ? sVal[s] : A.values[s];
        if (divide) v = __ddiv_rn(v, scale);
        A.values[s] = v;
        mVal[s] = v;
        if (inLds) {
            const uint32_t wd = sWord[sHead[s]];
            A.wordIds[s] = wd;
            mWid[s] = wd;
        }
    }
    if (inLds) { // the FeatureVector's tables, out of LDS (nn + 1 offsets)
        for (int s2 = t; s2 <= nn; s2 += NT) {
            const int off = (int)sOff[s2];
            A.offsets[s2] = off;
            mOffs[s2] = off;
            if (s2 < nn) {
                const uint32_t nd = sNode[off];
                A.nodeIds[s2] = nd;
                mNode[s2] = nd;
            }
        }
    }
    {
        // the index array, mirrored in whole rows of 64 lanes (a 4-byte store per rank from the ranking wavefronts would cross
        // PCIe as a transaction each)
        int32_t* const mInd = reinterpret_cast<int32_t*>(A.mirror + A.oInd);
        for (int r = t; r < m; r += NT) mInd[r] = (int32_t)bow_load_agent(reinterpret_cast<const uint32_t*>(&A.indices[r]));
    }
    BV_STAMP(9);
    if (t == 0) {
        int32_t* const mh = reinterpret_cast<int32_t*>(A.mirror + A.oHdr);
        A.hdr[0] = m;
        A.hdr[1] = nn;
        A.hdr[2] = nw;
        A.hdr[3] = sMax;
        mh[0] = m;
        mh[1] = nn;
        mh[2] = nw;
        mh[3] = sMax;
    }
}

// ---------------------------------------------------------------- the handle
struct orbfe_bow {
    orbfe_vocab_dev* vocab = nullptr;
    int device = 0, cap = 0;
    int n = 0, levelsup = 0; // of the last orbfe_compute_bow
    bool computed = false, pending = false;
    bool lazyNorm = false;   // orbfe_bow_set_lazy_norm: BowVector::normalize runs in the host view, not in the kernel
    bool hostNormDue = false; // ... and has not run yet for the last call
    // ONE device block.  Results first, in one run (mirrored to the host by one copy): header | node ids | offsets | indices |
    // word ids | word values; then the scratch of the three kernels and a descriptor buffer for host-side callers.
    uint8_t* block = nullptr;
    size_t outBytes = 0;
    int32_t* hdr = nullptr;
    uint32_t* nodeIds = nullptr;
    int32_t* offsets = nullptr;
    int32_t* indices = nullptr;
    uint32_t* wordIds = nullptr;
    double* values = nullptr;
    int32_t *word = nullptr, *node = nullptr, *headPos = nullptr;
    uint2* keys = nullptr; // (node, word) per feature: what K-BOWRANK sorts by
    double *weight = nullptr, *sortedWt = nullptr;
    uint32_t *sortedNode = nullptr, *sortedWord = nullptr;
    uint8_t* dDesc = nullptr;
    const uint8_t* lastDesc = nullptr; // device address of the descriptors of the last call (the caller's, or dDesc)
    uint8_t* hOut = nullptr;           // pinned mirror of the results (written by the kernel itself)
    uint8_t* hOutDev = nullptr;        // the kernel's address of it
    hipStream_t stream = nullptr;      // the stream of the last call (a consumer on the same stream needs no event wait)
    uint8_t* hDesc = nullptr;          // pinned staging of host descriptors
    hipEvent_t ev = nullptr;           // behind the mirror copy of the last call
    std::mutex hostMu; // the host view's one-time work (the wait for the kernel, the lazy normalisation) when searches of several
                       // threads name the same vector
    // (lifetime: the process-wide handle table, g_handles -- every entry point takes a use by LOOK-UP before it reads the
    // object, so a destroyed handle is refused without being dereferenced and a destroy under a search is deferred to its return)
    size_t off(const void* p) const { return (size_t)((const uint8_t*)p - block); }
};

namespace {
void bow_free(orbfe_bow* b)
{
    (void)hipSetDevice(b->device);
    if (b->ev) {
        (void)hipEventSynchronize(b->ev); // kernels of the last call may still be writing the block
        (void)hipEventDestroy(b->ev);
    }
    if (b->block) (void)hipFree(b->block);
    if (b->hOut) (void)hipHostFree(b->hOut);
    if (b->hDesc) (void)hipHostFree(b->hDesc);
    delete b;
}
void bow_free_v(void* p) { bow_free(static_cast<orbfe_bow*>(p)); }
int bow_resident(orbfe_bow* b, BowResident* R) // (under a use of the handle)
{
    if (!b->computed) return ORBFE_ERR_STATE;
    R->nodeIds = b->nodeIds;
    R->offsets = b->offsets;
    R->indices = b->indices;
    R->hdr = b->hdr;
    R->ready = b->ev;
    R->stream = b->stream;
    R->n = b->n;
    R->device = b->device;
    return 0;
}
int bow_host_view(orbfe_bow* b, orbfe_bow_view* v);
int bow_host_fv(orbfe_bow* b, orbfe_fv* host) // (under a use of the handle)
{
    if (!b) return ORBFE_ERR_ARGS;
    orbfe_bow_view v;
    const int r = bow_host_view(b, &v);
    if (r < 0) return r;
    host->nn = v.nn;
    host->node_ids = v.node_ids;
    host->offsets = v.offsets;
    host->indices = v.indices;
    return 0;
}
// host view of the last call's results (waits for the mirror copy)
int bow_host_view(orbfe_bow* b, orbfe_bow_view* v)
{
    std::lock_guard<std::mutex> hostLock(b->hostMu);
    if (!b->computed) return ORBFE_ERR_STATE;
    if (b->pending) {
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipEventSynchronize(b->ev));
        b->pending = false;
    }
    const int32_t* h = reinterpret_cast<const int32_t*>(b->hOut);
    if (h[0] < 0 || h[0] > b->n || h[1] < 0 || h[1] > h[0] || h[2] < 0 || h[2] > h[0]) return ORBFE_ERR_STATE;
    if (b->hostNormDue) { // BowVector::normalize (BowVector.cpp:62-86) on the mirrored values: ascending word id, one term at a time
        b->hostNormDue = false;
        double* val = reinterpret_cast<double*>(b->hOut + b->off(b->values));
        const int nw = h[2];
        const bool l2 = b->vocab->scoring == 1;
        double norm = 0.0;
        if (!l2)
            for (int i = 0; i < nw; i++) norm += std::fabs(val[i]);
        else {
            for (int i = 0; i < nw; i++) norm += val[i] * val[i];
            norm = std::sqrt(norm);
        }
        if (norm > 0.0)
            for (int i = 0; i < nw; i++) val[i] /= norm;
    }
    v->n_kept = h[0];
    v->nn = h[1];
    v->nw = h[2];
    v->max_node = h[3];
    v->node_ids = reinterpret_cast<const uint32_t*>(b->hOut + b->off(b->nodeIds));
    v->offsets = reinterpret_cast<const int32_t*>(b->hOut + b->off(b->offsets));
    v->indices = reinterpret_cast<const int32_t*>(b->hOut + b->off(b->indices));
    v->word_ids = reinterpret_cast<const uint32_t*>(b->hOut + b->off(b->wordIds));
    v->word_values = reinterpret_cast<const double*>(b->hOut + b->off(b->values));
    v->d_header = b->hdr;
    return 0;
}
} // namespace

#ifdef ORBFE_BOWVEC_TIMING
extern "C" int orbfe_debug_bowvec_times(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bowvecTimes), sizeof(g_bowvecTimes)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        z[0] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bowvecTimes), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" {

int orbfe_vocab_set_types(orbfe_vocab_dev* d, int weighting, int scoring)
{
    if (!d || weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5) return ORBFE_ERR_ARGS;
    d->weighting = weighting;
    d->scoring = scoring;
    return 0;
}

int orbfe_vocab_get_types(orbfe_vocab_dev* d, int* weighting, int* scoring)
{
    if (!d) return ORBFE_ERR_ARGS;
    if (weighting) *weighting = d->weighting;
    if (scoring) *scoring = d->scoring;
    return 0;
}

int orbfe_bow_create(orbfe_bow** out, orbfe_vocab_dev* vocab, int cap)
{
    if (!out) return ORBFE_ERR_ARGS;
    *out = nullptr;
    if (!vocab || cap < 1 || cap > 65535) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(vocab->device)) < 0) return r;
    orbfe_bow* b = new (std::nothrow) orbfe_bow();
    if (!b) return ORBFE_ERR_STATE;
    b->vocab = vocab;
    b->device = vocab->device;
    b->cap = cap;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t c = (size_t)cap;
    size_t o = 0;
    const size_t oHdr = o; o += al(64);
    const size_t oNode = o; o += al(c * 4);
    const size_t oOffs = o; o += al((c + 1) * 4);
    const size_t oInd = o; o += al(c * 4);
    const size_t oWid = o; o += al(c * 4);
    const size_t oVal = o; o += al(c * 8);
    b->outBytes = o;
    const size_t oWord = o; o += al(c * 4);
    const size_t oNd = o; o += al(c * 4);
    const size_t oWt = o; o += al(c * 8);
    const size_t oSn = o; o += al(c * 4);
    const size_t oSw = o; o += al(c * 4);
    const size_t oSwt = o; o += al(c * 8);
    const size_t oHead = o; o += al((c + 1) * 4);
    const size_t oKeys = o; o += al(c * 8);
    const size_t oDesc = o; o += al(c * 32);
    bool ok = hipMalloc((void**)&b->block, o) == hipSuccess && hipHostMalloc((void**)&b->hOut, b->outBytes) == hipSuccess &&
              hipHostMalloc((void**)&b->hDesc, c * 32) == hipSuccess &&
              hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (b->ev) (void)hipEventDestroy(b->ev);
        b->ev = nullptr;
        bow_free(b);
        return -(1000 + (int)hipErrorOutOfMemory);
    }
    {
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, b->hOut, 0) != hipSuccess || !dv) {
            (void)hipGetLastError();
            bow_free(b);
            return ORBFE_ERR_STATE;
        }
        b->hOutDev = (uint8_t*)dv;
    }
    if (hipMemset(b->block, 0, 64) != hipSuccess) { // (the header with the launch's block counter)
        bow_free(b);
        return ORBFE_ERR_STATE;
    }
    uint8_t* B = b->block;
    b->hdr = (int32_t*)(B + oHdr);
    b->nodeIds = (uint32_t*)(B + oNode);
    b->offsets = (int32_t*)(B + oOffs);
    b->indices = (int32_t*)(B + oInd);
    b->wordIds = (uint32_t*)(B + oWid);
    b->values = (double*)(B + oVal);
    b->word = (int32_t*)(B + oWord);
    b->node = (int32_t*)(B + oNd);
    b->weight = (double*)(B + oWt);
    b->sortedNode = (uint32_t*)(B + oSn);
    b->sortedWord = (uint32_t*)(B + oSw);
    b->sortedWt = (double*)(B + oSwt);
    b->headPos = (int32_t*)(B + oHead);
    b->keys = (uint2*)(B + oKeys);
    b->dDesc = B + oDesc;
    std::memset(b->hOut, 0, 32);
    g_handles.add(b, bow_free_v);
    *out = b;
    return 0;
}

void orbfe_bow_destroy(orbfe_bow* b)
{
    if (!b) return;
    if (g_handles.destroy(b, bow_free_v)) bow_free(b); // (else: freed by the last call that still holds it, or not a live BoW handle)
}

/* Frame::ComputeBoW: mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) on n descriptors (host or device
 * pointer).  Asynchronous on the calling thread's matcher stream: returns when the kernels and the mirror copy are queued. */
int orbfe_compute_bow(orbfe_bow* b, const uint8_t* desc, int n, int levelsup)
{
    HandleUses uses;
    if (!b || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS; // (destroyed, or not a BoW handle)
    if (n < 0 || n > b->cap || (n && !desc)) return ORBFE_ERR_ARGS;
    int r;
    if ((r = select_device(b->device)) < 0) return r;
    Scratch s(b->device); // (this thread's matcher stream: g_ms)
    if (b->pending) { // the previous call's copy still owns the mirror (and hDesc)
        HIP_TRY(hipEventSynchronize(b->ev));
        b->pending = false;
    }
    b->n = n;
    b->levelsup = levelsup;
    const uint8_t* dF = b->dDesc;
    if (n && is_device_ptr(desc)) {
        if (int w = orbfe_producer_wait(desc, g_ms); w < 0) return w; // an extractor may still be writing them (orbfe_order.h)
        dF = desc;
    } else if (n) {
        std::memcpy(b->hDesc, desc, (size_t)n * 32); // (the caller's array is free when the call returns)
        HIP_TRY(hipMemcpyAsync(b->dDesc, b->hDesc, (size_t)n * 32, hipMemcpyHostToDevice, g_ms));
    }
    b->lastDesc = dF;
    const orbfe_vocab_dev* d = b->vocab;
    if (n) {
        const DoneSig none = {};
        hipLaunchKernelGGL(k_vocab_transform, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, g_ms, d->desc, d->childOff, d->childIds,
                           d->word, d->weight, d->L, dF, n, levelsup, b->word, b->node, b->weight, none, b->keys);
    }
    BowFoldArgs A;
    A.keys = b->keys;
    A.weight = b->weight;
    A.n = n;
    A.sortedNode = b->sortedNode;
    A.sortedWord = b->sortedWord;
    A.sortedWt = b->sortedWt;
    A.hdr = b->hdr;
    A.nodeIds = b->nodeIds;
    A.offsets = b->offsets;
    A.indices = b->indices;
    A.wordIds = b->wordIds;
    A.values = b->values;
    A.headPos = b->headPos;
    A.mirror = b->hOutDev;
    A.oHdr = (uint32_t)b->off(b->hdr);
    A.oNode = (uint32_t)b->off(b->nodeIds);
    A.oOffs = (uint32_t)b->off(b->offsets);
    A.oInd = (uint32_t)b->off(b->indices);
    A.oWid = (uint32_t)b->off(b->wordIds);
    A.oVal = (uint32_t)b->off(b->values);
    A.addWeight = d->weighting == 0 || d->weighting == 1; // TF_IDF || TF (:1145)
    A.must = d->scoring != 5;                             // every scoring object but DotProductScoring (ScoringObject.h:73-89)
    A.normL2 = d->scoring == 1;
    A.lazyNorm = b->lazyNorm ? 1 : 0;
    b->hostNormDue = b->lazyNorm && A.must;
    // (n == 0: one workgroup, which is the last one and writes the four zero counts)
    hipLaunchKernelGGL(k_bow_rank_fold, dim3((unsigned)std::max((n + 3) / 4, 1)), dim3(256), 0, g_ms, A);
    HIP_TRY(hipGetLastError());
    // (no copy command: the kernel has written the results' mirror itself; the event marks both)
    HIP_TRY(hipEventRecord(b->ev, g_ms));
    b->stream = g_ms;
    b->pending = true;
    b->computed = true;
    return 0;
}

/* on != 0: BowVector::normalize is left to orbfe_bow_host (the sum over the words is a chain of dependent double additions, one
 * lane's work: 7-15 us of the kernel for 450-1000 words, a microsecond on the host); the values on the DEVICE then stay
 * un-normalised (nothing on the device reads them: the searches use the FeatureVector).  Default off: both vectors complete on
 * the device. */
int orbfe_bow_set_lazy_norm(orbfe_bow* b, int on)
{
    HandleUses uses;
    if (!b || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS;
    b->lazyNorm = on != 0;
    return 0;
}

int orbfe_bow_host(orbfe_bow* b, orbfe_bow_view* view)
{
    HandleUses uses;
    if (!b || !view || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS;
    return bow_host_view(b, view);
}

int orbfe_bow_device(orbfe_bow* b, orbfe_bow_view* view)
{
    HandleUses uses;
    if (!b || !view || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS;
    if (!b->computed) return ORBFE_ERR_STATE;
    view->n_kept = view->nn = view->nw = view->max_node = -1; // (on the device: d_header[0..3])
    view->node_ids = b->nodeIds;
    view->offsets = b->offsets;
    view->indices = b->indices;
    view->word_ids = b->wordIds;
    view->word_values = b->values;
    view->d_header = b->hdr;
    return 0;
}

/* The FeatureVector of the last orbfe_compute_bow as an orbfe_fv that names the HANDLE (nn = ORBFE_FV_RESIDENT): accepted
 * wherever a search or orbfe_keyframe_create takes an orbfe_fv.  A SearchByBoW that pairs the nodes on the device reads the
 * handle's arrays where they lie (no host copy is ever made); every other consumer asks for the host view first. */
int orbfe_bow_fv(orbfe_bow* b, orbfe_fv* fv)
{
    HandleUses uses;
    if (!b || !fv || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS;
    if (!b->computed) return ORBFE_ERR_STATE;
    fv->nn = ORBFE_FV_RESIDENT;
    fv->node_ids = reinterpret_cast<const uint32_t*>(b);
    fv->offsets = nullptr;
    fv->indices = nullptr;
    return 0;
}

} // extern "C"
