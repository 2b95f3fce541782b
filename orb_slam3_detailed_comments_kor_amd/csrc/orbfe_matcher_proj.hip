// orbfe_matcher_proj.hip -- K-PROJ (SearchByProjection / Fuse / SearchBySim3), K-INIT (SearchForInitialization): kernels.
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
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

