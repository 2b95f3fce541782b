// orbfe_matcher_api_tri.hip -- entry points: SearchForTriangulation (batch, pinhole, KB8, 3-D), fisheye stereo, SearchForInitialization, KB8 triangulation.
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).
// SearchForTriangulation_ of ONE keyframe against `count` neighbours (src/LocalMapping.cc:556-621), all sides resident:
// one upload of the row lists and pair records, ONE launch, one download.
int orbfe_search_tri_batch(orbfe_keyframe* K1, const uint8_t* hasMP1, int count, orbfe_keyframe* const* kf2,
                           const orbfe_tri_pair* pair, int32_t* const* pairs, int* npairs)
{
    if (!K1 || count < 0 || (count && (!kf2 || !pair || !pairs || !npairs))) return ORBFE_ERR_ARGS;
    HandleUses uses;
    if (!uses.take(K1, keyframe_free)) return ORBFE_ERR_ARGS;
    for (int p = 0; p < count; p++)
        if (!kf2[p] || !uses.take(kf2[p], keyframe_free)) return ORBFE_ERR_ARGS;
    if (!K1->hasTri) return ORBFE_ERR_ARGS;
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
    // those.  Form B: the rows themselves come back and the host does it (single neighbours, and batches whose
    // pair lists would not fit the mirror).
    const bool compactOn = true;
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
        if (true)
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
    if (mirrored)
        fprintf(stderr, "tri_batch count=%d rows=%zu: rows %.1f stage %.1f launch %.1f wait %.1f tail %.1f us\n", count, rows.size(), trT[0], trT[1],
                trT[2], trT[3], trT[4]);
#endif
    return 0;
}

int orbfe_search_tri(int device, const orbfe_tri_args* a0, int32_t* pairs)
{
    orbfe_tri_args aLocal;
    const orbfe_tri_args* a = a0;
    HandleUses fvUses; // (a vector that names an orbfe_bow handle: the handle is held until this call returns)
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1, fvUses); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2, fvUses); rr < 0) return rr;
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
    HandleUses fvUses; // (a vector that names an orbfe_bow handle: the handle is held until this call returns)
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1, fvUses); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2, fvUses); rr < 0) return rr;
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
    HandleUses fvUses; // (a vector that names an orbfe_bow handle: the handle is held until this call returns)
    if (a0 && (a0->fv1.nn == ORBFE_FV_RESIDENT || a0->fv2.nn == ORBFE_FV_RESIDENT)) { // vectors of orbfe_bow handles: their host copies
        aLocal = *a0;
        if (int rr = fv_resolve(&aLocal.fv1, fvUses); rr < 0) return rr;
        if (int rr = fv_resolve(&aLocal.fv2, fvUses); rr < 0) return rr;
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

