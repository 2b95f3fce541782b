// orbfe_matcher_tri.hip -- K-TRI (SearchForTriangulation_: pinhole, KB8, 3-D), k_fisheye_stereo: kernels.
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
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

