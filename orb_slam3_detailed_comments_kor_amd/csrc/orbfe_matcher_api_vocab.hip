// orbfe_matcher_api_vocab.hip -- entry points: kernel timing, timing taps, DBoW2 vocabulary (upload, text loader, transform).
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
// Host-only exerciser of the handle table (g_handles / HandleUses: who is alive, who holds a use, who frees) for the sanitizer
// builds, which run without a device and therefore cannot create a real handle (tests/san/threads_cabi.cpp, under TSan and ASan):
// `threads` searcher threads take and give back uses of `slots` fake handles while the calling thread destroys and re-creates
// them `rounds` times.  Checked: a handle is freed exactly once per creation, never while a use is out, and a use is never
// granted on a handle after its destroy.  Returns 0, or the number of violations.
int orbfe_debug_handle_table_selftest(int threads, int slots, int rounds)
{
    if (threads < 1 || slots < 1 || rounds < 1 || slots > 64 || threads > 64) return ORBFE_ERR_ARGS;
    struct Fake {
        std::atomic<int> usesOut{0}, freed{0}, alive{0};
    };
    static Fake pool[64 * 2]; // (two generations per slot so that a freed object is not re-added while a searcher may still name it)
    static std::atomic<int> violations;
    violations.store(0);
    std::atomic<bool> stop{false};
    std::atomic<Fake*> cur[64];
    auto freeFn = [](void* h) {
        Fake* f = static_cast<Fake*>(h);
        if (f->usesOut.load() != 0) violations.fetch_add(1); // freed while a use is out
        if (f->freed.fetch_add(1) != 0) violations.fetch_add(1); // freed twice
    };
    for (int s = 0; s < slots; s++) {
        Fake* f = &pool[2 * s];
        f->usesOut.store(0);
        f->freed.store(0);
        f->alive.store(1);
        g_handles.add(f, +freeFn);
        cur[s].store(f);
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            unsigned x = 12345u + 977u * (unsigned)t;
            while (!stop.load()) {
                HandleUses uses;
                for (int k = 0; k < 3; k++) {
                    x = x * 1664525u + 1013904223u;
                    Fake* f = cur[(x >> 8) % (unsigned)slots].load();
                    if (uses.take(f, +freeFn)) {
                        if (f->alive.load() == 0 && f->freed.load() != 0) violations.fetch_add(1); // a use granted on a freed handle
                    }
                }
                for (const auto& e : uses.held) static_cast<Fake*>(const_cast<void*>(e.first))->usesOut.fetch_add(1);
                for (const auto& e : uses.held) static_cast<Fake*>(const_cast<void*>(e.first))->usesOut.fetch_sub(1);
            }
        });
    for (int r = 0; r < rounds; r++)
        for (int s = 0; s < slots; s++) {
            Fake* old = cur[s].load();
            Fake* nw = old == &pool[2 * s] ? &pool[2 * s + 1] : &pool[2 * s];
            // (the other generation was destroyed a whole round ago; wait until its deferred free, if any, has happened)
            while (nw->alive.load() == 0 && nw->freed.load() == 0 && r > 0) std::this_thread::yield();
            nw->usesOut.store(0);
            nw->freed.store(0);
            nw->alive.store(1);
            g_handles.add(nw, +freeFn);
            cur[s].store(nw);
            old->alive.store(0);
            if (g_handles.destroy(old, +freeFn)) freeFn(old);
        }
    stop.store(true);
    for (auto& t : th) t.join();
    for (int s = 0; s < slots; s++) {
        Fake* f = cur[s].load();
        f->alive.store(0);
        // a handle of one kind is neither granted nor destroyed as another (a recycled address must not pass for the old type)
        auto otherKind = [](void*) {};
        {
            HandleUses wrong;
            if (wrong.take(f, +otherKind)) violations.fetch_add(1);
        }
        if (g_handles.destroy(f, +otherKind)) violations.fetch_add(1);
        if (g_handles.destroy(f, +freeFn)) freeFn(f);
        if (f->freed.load() != 1) violations.fetch_add(1);
    }
    return violations.load();
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
    // (the tree of ORBvoc.txt is 35 MB of descriptors in the caller's pageable memory: in pieces through page-locked memory of
    // this thread, orbfe_pageable.h; on the null stream, once per vocabulary)
    using orbfe_pageable::up;
    ok = ok && up(d->desc, v->node_desc, (size_t)v->nnodes * 32, nullptr) == hipSuccess &&
         up(d->childOff, v->child_off, (size_t)(v->nnodes + 1) * 4, nullptr) == hipSuccess &&
         (nchild == 0 || up(d->childIds, v->child_ids, (size_t)nchild * 4, nullptr) == hipSuccess) &&
         up(d->word, v->node_word, (size_t)v->nnodes * 4, nullptr) == hipSuccess &&
         up(d->weight, v->node_weight, (size_t)v->nnodes * 8, nullptr) == hipSuccess;
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

