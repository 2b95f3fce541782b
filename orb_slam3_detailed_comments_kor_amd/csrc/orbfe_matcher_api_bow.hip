// orbfe_matcher_api_bow.hip -- entry points: keyframe handles, SearchByBoW (bow_run).
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
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
    hipStream_t stream;                     // ... which ran on this stream
    int n, device;
};
void bow_free_v(void*);                        // what frees an orbfe_bow = its kind in the handle table (g_handles)
int bow_resident(orbfe_bow*, BowResident*);   // (the caller holds a use of the handle: HandleUses::take(b, bow_free_v))
int bow_host_fv(orbfe_bow*, orbfe_fv* host);  // waits for the host copy (same: under a use); valid until the next orbfe_compute_bow
// An orbfe_fv that names a handle, replaced by the handle's host copy (every consumer but the in-kernel pairing of bow_run).
// The use is the caller's: the host copy lives in the handle's page-locked memory and is read until the call returns.
int fv_resolve(orbfe_fv* f, HandleUses& uses)
{
    if (f->nn != ORBFE_FV_RESIDENT) return 0;
    orbfe_bow* b = reinterpret_cast<orbfe_bow*>(const_cast<uint32_t*>(f->node_ids));
    if (!b || !uses.take(b, bow_free_v)) return ORBFE_ERR_ARGS; // (destroyed, or not a BoW handle at all)
    return bow_host_fv(b, f);
}
} // namespace

namespace {
void keyframe_free(void* h);
}
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
// staged inputs up to this size are read by the kernel from the pinned staging in place (128 KB: 85 KB of a host-array SearchByBoW read in place took 0.037 instead of 0.045 ms)
static size_t inplace_limit() { return (size_t)128 << 10; }
// tuning only (tools/ab_build.sh trace "-DORBFE_CALL_TRACE"): where the host time of a
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
    HandleUses uses; // (a handle another thread destroys meanwhile lives until this call returns; a dead one is refused)
    for (int p = 0; p < count; p++) {
        if (kf1 && kf1[p] && !uses.take(kf1[p], keyframe_free)) return ORBFE_ERR_ARGS;
        if (kf2 && kf2[p] && !uses.take(kf2[p], keyframe_free)) return ORBFE_ERR_ARGS;
    }
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
    std::vector<uint8_t> own1(count, 0), own2(count, 0); // this problem stages the set (first occurrence)
    std::vector<int> set1(count, -1), set2(count, -1);  // index into `seen` of a pooled side
    size_t nodeTotal = 0;
    // Round 5: a call whose results are downloaded anyway (more than 256 KB of them: the 64 candidates of a
    // relocalisation) leaves the merge-join of the FeatureVectors to the kernel -- 33 us of host time and a 128-KB node list per
    // call of 64; the node ids / offsets of sets that are not in handles travel in the pool (~1 KB per set).
    // (Calls whose results come back through the pinned mirror keep the host list: their launch counts its workgroups for the
    // completion word.)
    bool devNodes = false;
    {
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
            if (!uses.take(B, bow_free_v)) return ORBFE_ERR_ARGS; // (destroyed, or no BoW handle; held until this call returns)
            if (!devNodes) { // host lists: the handle's host copy
                const int rr = bow_host_fv(B, &f);
                if (rr < 0) return rr;
                continue;
            }
            BowResident& R = side ? res2[p] : res1[p];
            const int rr = bow_resident(B, &R);
            if (rr < 0) return rr;
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
        // (a vector computed on THIS thread's stream is ordered by the stream itself)
        if (isRes1[p] && res1[p].stream != g_ms) HIP_TRY(hipStreamWaitEvent(g_ms, res1[p].ready, 0));
        if (isRes2[p] && res2[p].stream != g_ms && !(isRes1[p] && res2[p].ready == res1[p].ready) &&
            !(p > 0 && isRes2[p - 1] && res2[p - 1].ready == res2[p].ready))
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
                               dP, dDesc, dMask, dAng, dInd, dM, dB, taken, done);
        else
        hipLaunchKernelGGL(k_search_bow, dim3((unsigned)nodes.size()), dim3(256), 0, g_ms, dN, (int)nodes.size(),
                           dP, dDesc, dMask, dAng, dInd, dM, dB, taken, done);
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
    fprintf(stderr, "bow_run count=%d: pass1 %.1f stage %.1f launch %.1f sync %.1f tail %.1f us\n", count, trT[0], trT[1], trT[2], trT[3], trT[4]);
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
    HandleUses fvUses; // (a vector that names an orbfe_bow handle: the handle is held until this call returns)
    if (a0 && a0->fv.nn == ORBFE_FV_RESIDENT) { // the vector of an orbfe_bow handle: its host copy (the handle keeps host views)
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv, fvUses); rr < 0) return rr;
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
        using orbfe_pageable::up; // (the caller's arrays may be pageable: through page-locked memory of this thread)
        e = descResident ? hipMemcpyAsync(K->desc, a->desc, n * 32, hipMemcpyDeviceToDevice, g_ms) : up(K->desc, a->desc, n * 32, g_ms);
        if (e == hipSuccess) e = up(K->mask, a->mask, n, g_ms);
        if (e == hipSuccess && a->angle) e = up(K->ang, a->angle, n * 4, g_ms);
        if (e == hipSuccess && !a->angle) e = hipMemsetAsync(K->ang, 0, n * 4, g_ms);
        if (e == hipSuccess && tri) e = up(K->kp, a->kp_xy, n * 8, g_ms);
        if (e == hipSuccess && tri) e = up(K->uR, a->uRight, n * 4, g_ms);
        if (e == hipSuccess && tri) e = up(K->oct, a->octave, n * 4, g_ms);
        if (e == hipSuccess && ni) e = up(K->ind, a->fv.indices, ni * 4, g_ms);
        if (e == hipSuccess && a->fv.nn) e = up(K->dNode, K->nodeIds.data(), (size_t)a->fv.nn * 4, g_ms);
        if (e == hipSuccess) e = up(K->dOffs, K->offsets.data(), ((size_t)a->fv.nn + 1) * 4, g_ms);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_ms); // the caller's arrays are free again; the handle is complete
    if (e != hipSuccess) {
        g_blockPool.put(device, blk, blkCap);
        delete K;
        return -(1000 + (int)e);
    }
    g_handles.add(K, keyframe_free);
    *out = K;
    return 0;
}

int orbfe_keyframe_set_mask(orbfe_keyframe* K, const uint8_t* mask)
{
    if (!K || !mask) return ORBFE_ERR_ARGS;
    HandleUses uses;
    if (!uses.take(K, keyframe_free)) return ORBFE_ERR_ARGS; // (destroyed)
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

namespace {
void keyframe_free(void* h)
{
    orbfe_keyframe* K = static_cast<orbfe_keyframe*>(h);
    // (no device synchronisation: every search has done all its device reads before it gives its use back)
    g_blockPool.put(K->device, K->block, K->blockCap);
    delete K;
}
} // namespace

void orbfe_keyframe_destroy(orbfe_keyframe* K)
{
    if (!K) return;
    // (no hipDeviceSynchronize here -- it drained the extractor's batches in flight and every other thread's
    // searches whenever the adapter's table evicted a keyframe.  Round 6: a search of another thread that still holds the
    // handle keeps it alive -- the last use frees it, g_handles.)
    if (g_handles.destroy(K, keyframe_free)) keyframe_free(K);
}

int orbfe_search_bow_keyframes(int device, int count, orbfe_keyframe* const* kf1, orbfe_keyframe* const* kf2,
                               const orbfe_bow_args* args, int32_t* const* match, int* nmatches)
{
    return bow_run(device, count, args, kf1, kf2, match, nmatches);
}

