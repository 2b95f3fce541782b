"""Randomised parity sweep of the matcher entry points vs the oracle.  tests/test_gpu_sweeps.py runs a bounded
fixed-seed slice of it on the GPU box; alone: python tools/stress_matcher.py [cases] [seed] [only this kind]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
import orb_oracle_py as O  # noqa: E402
import matcher_inputs as MI  # noqa: E402

def _near_leaf_features(vocab, n, rng, flips):
    # descriptors a few bit flips away from the vocabulary's words (with repetition: several features per word)
    leaves = np.nonzero(vocab["word"] >= 0)[0]
    bits = np.unpackbits(vocab["desc"][rng.choice(leaves, size=n)], axis=1)
    flip = rng.random(bits.shape) < (rng.integers(0, flips + 1, size=(n, 1)) / 256.0)
    return np.packbits(bits ^ flip.astype(np.uint8), axis=1)


def run(ncases=60, seed=3, scale=1.0, log=print, kinds=12, only=None):
    """Returns the list of mismatch descriptions.  scale < 1 shrinks the random problem sizes (test-suite slice).
    kinds = 6: the one-shot entry points only; 10: also the resident forms (keyframe / frame handles, the triangulation
    batch) and the stereo pair in one call; 11: also batched calls whose problems share sides (staged once per call);
    12: also Frame::ComputeBoW on the device (random trees, types, stop words) and its resident vector through SearchByBoW."""
    O.build()
    rng = np.random.default_rng(seed)
    bad = []

    def size(lo, hi):
        return int(rng.integers(lo, max(lo + 1, int(lo + (hi - lo) * scale))))

    for case in range(ncases):
        kind = case % kinds if only is None else int(only)  # (only: one kind over and over, for a new kind's first runs)
        seed = int(rng.integers(0, 1 << 30))
        ok = True
        desc = ""
        if kind == 0:    # SearchByProjection local map / last frame
            mode = int(rng.integers(0, 2))
            kw = dict(seed=seed, mode=mode, n=size(50, 3000), nq=size(1, 2500),
                      th=float(rng.choice([1.0, 3.0, 7.0, 15.0])), stereo=bool(rng.integers(0, 2)),
                      check_orientation=bool(mode and rng.integers(0, 2)), taken_frac=float(rng.uniform(0, 0.4)),
                      nnratio=float(rng.choice([0.6, 0.8, 0.9])))
            if rng.random() < 0.3:
                kw.update(stereo=False, Nleft=int(kw["n"] * rng.uniform(0.3, 0.7)), partners=bool(mode == 0 and rng.integers(0, 2)))
            if rng.random() < 0.3:  # (round 5: also together with a rig's stereo-partner writes -- the in-order walk)
                kw["blocks"] = float(rng.uniform(0.0, 0.9))
            pr = MI.projection_problem(**kw)
            a, b = O.search_projection(pr), pkg.search_projection(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            desc = "proj %s -> %d" % (kw, a[0])
        elif kind == 1:  # Fuse / Sim3 family
            loop = str(rng.choice(["sim3_projection", "fuse", "fuse_sim3", "search_by_sim3"]))
            kw = dict(seed=seed, mode=1, n=size(50, 2500), nq=size(1, 2000),
                      th=float(rng.choice([2.5, 3.0, 4.0, 7.5])), loop=loop, stereo=bool(loop == "fuse" and rng.integers(0, 2)))
            if loop == "fuse" and not kw["stereo"] and rng.random() < 0.4:
                kw["Nleft"] = int(kw["n"] * rng.uniform(0.3, 0.7))
            pr = MI.projection_problem(**kw)
            a, b = O.search_projection(pr), pkg.search_projection(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            desc = "%s %s -> %d" % (loop, {k: kw[k] for k in ("n", "nq", "th")}, a[0])
        elif kind == 2:  # SearchForInitialization
            kw = dict(seed=seed, n1=size(20, 3000), n2=size(2, 3000), window=int(rng.choice([30, 100, 200])),
                      nnratio=float(rng.choice([0.7, 0.9, 1.0])), check_orientation=bool(rng.integers(0, 2)), crowd=bool(rng.integers(0, 2)))
            pr = MI.initialization_problem(**kw)
            a, b = O.search_initialization(pr), pkg.search_initialization(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1])
            desc = "init %s -> %d" % (kw, a[0])
        elif kind == 3:  # SearchByBoW both variants
            n1, n2 = size(30, 2000), size(30, 2000)
            d1, d2, a1, a2 = MI.descriptor_sets(n1, n2, seed % 100000)
            fv1, fv2 = MI.feature_vectors(d1, d2, seed % 1000, int(rng.integers(3, 11)), 2)
            m1 = (rng.uniform(size=n1) < 0.6).astype(np.uint8)
            m2 = (rng.uniform(size=n2) < 0.6).astype(np.uint8)
            ori = bool(rng.integers(0, 2))
            ratio = float(rng.choice([0.6, 0.7, 0.9]))
            a = O.search_bow_kf_f(d1, m1, a1, fv1, d2, a2, fv2, -1, ratio, ori)
            b = pkg.search_bow(d1, m1, a1, fv1, d2, None, a2, fv2, 0, ratio, ori)
            c = O.search_bow_kf_kf(d1, m1, a1, fv1, d2, m2, a2, fv2, -1, -1, ratio, ori)
            d = pkg.search_bow(d1, m1, a1, fv1, d2, m2, a2, fv2, 1, ratio, ori)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and c[0] == d[0] and np.array_equal(c[1], d[1])
            desc = "bow n1=%d n2=%d -> %d / %d" % (n1, n2, a[0], c[0])
        elif kind == 4:  # SearchForTriangulation_ pinhole
            n1, n2 = size(50, 2000), size(50, 2000)
            I = MI.tri_inputs(n1, n2, seed % 100000)
            os_, co = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            args = (I["d1"], I["has1"], I["kp1"], I["a1"], I["oct1"], I["u1"], I["fv1"], I["d2"], I["has2"], I["kp2"], I["a2"],
                    I["oct2"], I["u2"], I["fv2"], I["F12"], I["ep"], I["sf"], I["sig"], os_, co, True)
            a, b = O.search_triangulation(*args), pkg.search_triangulation(*args)
            ok = np.array_equal(a, b)
            desc = "tri n1=%d n2=%d stereo=%s coarse=%s -> %d" % (n1, n2, os_, co, len(a))
        elif kind == 6:  # SearchByBoW with one or both sides in keyframe handles
            n1, n2 = size(30, 2000), size(30, 2000)
            d1, d2, a1, a2 = MI.descriptor_sets(n1, n2, seed % 100000)
            fv1, fv2 = MI.feature_vectors(d1, d2, seed % 1000, int(rng.integers(3, 11)), 2)
            m1 = (rng.uniform(size=n1) < 0.6).astype(np.uint8)
            m2 = (rng.uniform(size=n2) < 0.6).astype(np.uint8)
            ori = bool(rng.integers(0, 2))
            ratio = float(rng.choice([0.6, 0.7, 0.9]))
            h1, h2 = pkg.KeyFrameHandle(d1, m1, a1, fv1), pkg.KeyFrameHandle(d2, m2, a2, fv2)
            a = O.search_bow_kf_f(d1, m1, a1, fv1, d2, a2, fv2, -1, ratio, ori)
            c = O.search_bow_kf_kf(d1, m1, a1, fv1, d2, m2, a2, fv2, -1, -1, ratio, ori)
            b, d = pkg.search_bow_keyframes([dict(kf1=h1, desc2=d2, ang2=a2, fv2=fv2, variant=0, nnratio=ratio, check_ori=ori),
                                             dict(kf1=h1, kf2=h2, variant=1, nnratio=ratio, check_ori=ori)])
            h1.close()
            h2.close()
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and c[0] == d[0] and np.array_equal(c[1], d[1])
            desc = "bow handles n1=%d n2=%d -> %d / %d" % (n1, n2, a[0], c[0])
        elif kind == 7:  # SearchForTriangulation_ of one keyframe against several neighbours in one launch
            n1 = size(50, 1500)
            cnt = int(rng.integers(1, 9))
            I0 = MI.tri_inputs(n1, 60, seed % 100000)
            cur = pkg.KeyFrameHandle(I0["d1"], I0["has1"], I0["a1"], I0["fv1"], kp_xy=I0["kp1"], octave=I0["oct1"], uRight=I0["u1"])
            neigh, want = [], []
            for k in range(cnt):
                I = MI.tri_inputs(n1, size(50, 1500), seed % 100000)  # (the same set 1 and vocabulary: same seed)
                flags = dict(only_stereo=bool(rng.integers(0, 2)), coarse=bool(rng.integers(0, 2)), check_ori=bool(rng.integers(0, 2)))
                F12 = (I["F12"] * np.float32(1.0 + 0.01 * k)).astype(np.float32)
                h = pkg.KeyFrameHandle(I["d2"], I["has2"], I["a2"], I["fv2"], kp_xy=I["kp2"], octave=I["oct2"], uRight=I["u2"])
                neigh.append(dict(kf=h, F12=F12, ep=I["ep"], sf=I["sf"], sig=I["sig"], **flags))
                want.append(O.search_triangulation(I0["d1"], I0["has1"], I0["kp1"], I0["a1"], I0["oct1"], I0["u1"], I0["fv1"], I["d2"],
                                                   I["has2"], I["kp2"], I["a2"], I["oct2"], I["u2"], I["fv2"], F12, I["ep"], I["sf"],
                                                   I["sig"], flags["only_stereo"], flags["coarse"], flags["check_ori"]))
            got = pkg.search_tri_batch(cur, neigh)
            cur.close()
            for q in neigh:
                q["kf"].close()
            ok = len(got) == cnt and all(np.array_equal(g, w) for g, w in zip(got, want))
            desc = "tri batch n1=%d x %d -> %s" % (n1, cnt, [len(w) for w in want])
        elif kind == 8:  # projection searches against a resident frame, several calls on one handle
            n = size(50, 2500)
            base = MI.projection_problem(seed, n=n, nq=5, mode=1, stereo=bool(rng.integers(0, 2)), check_orientation=True)
            fr = pkg.ProjectionFrame(base)
            ok, outs = True, []
            for rep in range(3):
                mode = int(rng.integers(0, 2))
                nq = size(1, 2000)
                q = MI.projection_problem(seed + 1 + rep, n=n, nq=nq, mode=mode, stereo="uright" in base and base["uright"] is not None,
                                          th=float(rng.choice([1.0, 3.0, 7.0, 15.0])), check_orientation=bool(mode and rng.integers(0, 2)),
                                          nnratio=float(rng.choice([0.6, 0.8, 0.9])), taken_frac=float(rng.uniform(0, 0.4)))
                pr = dict(base)
                for key, v in q.items():
                    if key.startswith("q") or key in ("mode", "nnratio", "th_high", "check_orientation", "taken", "chi2_gate", "inv_level_sigma2"):
                        pr[key] = v
                tgt = rng.integers(0, n, nq)
                pr["qx"] = (base["kx"][tgt] + rng.normal(0, 2, nq)).astype(np.float32)
                pr["qy"] = (base["ky"][tgt] + rng.normal(0, 2, nq)).astype(np.float32)
                bits = np.unpackbits(base["desc"][tgt], axis=1)
                pr["qdesc"] = np.packbits(bits ^ (rng.random(bits.shape) < 0.1), axis=1)
                if pr.get("qxr") is not None and base.get("uright") is not None:
                    pr["qxr"] = np.where(base["uright"][tgt] > 0, base["uright"][tgt] + 1.0, pr["qx"] - 10).astype(np.float32)
                a, b = O.search_projection(pr), fr.search(pr)
                ok = ok and a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
                outs.append(a[0])
            fr.close()
            desc = "proj frame n=%d -> %s" % (n, outs)
        elif kind == 9:  # a stereo pair in one call: both extractions + Frame::ComputeStereoMatches
            from orb_slam3_detailed_comments_kor_amd import synth
            h, w = int(rng.integers(200, 520)), int(rng.integers(320, 800))
            nf = int(rng.choice([300, 800, 1200, 2000]))
            shift = int(rng.integers(0, 70))
            left, right = synth.make_stereo_pair(h, w, seed % 100000, shift=min(shift, w // 4))
            if rng.random() < 0.3:  # other content on both sides
                kind_ = str(rng.choice(["blurred", "plateaus", "checker2", "sinus", "ramp", "mixed"]))
                left = synth.make_frame_kind(h, w, seed % 100000, kind_)
                right = np.ascontiguousarray(np.roll(left, -min(shift, w // 4), axis=1))
            mbf = float(rng.choice([47.90639384423901, 20.0, 120.0]))
            mb = mbf / float(rng.choice([435.2046959714599, 300.0, 700.0]))
            ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
            oL, oR = O.Extractor(nf, 1.2, 8, 20, 7), O.Extractor(nf, 1.2, 8, 20, 7)
            try:
                rL, rR = oL.extract(left, (0, 0)), oR.extract(right, (0, 0))
            except Exception as e:  # noqa: BLE001 (geometry the oracle refuses: the library must refuse it too)
                rL = None
                why = repr(e)
            if rL is None:
                try:
                    pkg.binding.extract_stereo_pair(ex, left, right, mb, mbf)
                    ok = False
                except pkg.OrbfeError:
                    ok = True
                desc = "stereo pair %dx%d refused (%s)" % (h, w, why[:40])
            else:
                rn, ruR, rdep = O.compute_stereo_matches(oL, oR, rL[1], rL[2], rR[1], rR[2], mb, mbf)
                m, gL, gR, uR, dep = pkg.binding.extract_stereo_pair(ex, left, right, mb, mbf)
                ok = (m == rn and gL[0] == rL[0] and gR[0] == rR[0] and np.array_equal(gL[1], rL[1]) and np.array_equal(gL[2], rL[2])
                      and np.array_equal(gR[1], rR[1]) and np.array_equal(gR[2], rR[2]) and np.array_equal(uR, ruR) and np.array_equal(dep, rdep))
                desc = "stereo pair %dx%d nF=%d shift=%d -> %d kp, %d matches" % (h, w, nf, shift, len(rL[1]), rn)
            ex.close()
        elif kind == 10:  # batched calls: problems that share a side, repeat each other, or share nothing; small and large batches
            nb = int(rng.choice([2, 3, 7, 20, 70, 120]))
            sets = []
            for k in range(3):  # three (set 1, set 2) worlds to draw from
                n1, n2 = size(30, 1200), size(30, 1200)
                d1, d2, a1, a2 = MI.descriptor_sets(n1, n2, (seed + k) % 100000)
                fv1, fv2 = MI.feature_vectors(d1, d2, seed % 1000, int(rng.integers(3, 9)), 2)
                sets.append(dict(d1=d1, d2=d2, a1=a1, a2=a2, fv1=fv1, fv2=fv2, m1=(rng.uniform(size=n1) < 0.6).astype(np.uint8),
                                 m2=(rng.uniform(size=n2) < 0.6).astype(np.uint8)))
            P, want = [], {}
            for j in range(nb):
                S1, S2 = sets[int(rng.integers(0, 3))], sets[int(rng.integers(0, 3))]
                if rng.random() < 0.6:
                    S2 = S1  # (matching sides most of the time: something to find)
                variant = int(rng.integers(0, 2))
                ratio = float(rng.choice([0.6, 0.75, 0.9]))
                ori = bool(rng.integers(0, 2))
                P.append(dict(desc1=S1["d1"], mask1=S1["m1"], ang1=S1["a1"], fv1=S1["fv1"], desc2=S2["d2"], mask2=S2["m2"] if variant else None,
                              ang2=S2["a2"], fv2=S2["fv2"], variant=variant, nnratio=ratio, check_ori=ori))
                key = (id(S1), id(S2), variant, ratio, ori)
                if key not in want:
                    want[key] = (O.search_bow_kf_kf(S1["d1"], S1["m1"], S1["a1"], S1["fv1"], S2["d2"], S2["m2"], S2["a2"], S2["fv2"], -1, -1, ratio, ori)
                                 if variant else O.search_bow_kf_f(S1["d1"], S1["m1"], S1["a1"], S1["fv1"], S2["d2"], S2["a2"], S2["fv2"], -1, ratio, ori))
                P[-1]["_key"] = key
            got = pkg.search_bow_batch([{k: v for k, v in pr.items() if k != "_key"} for pr in P])
            ok = all(g[0] == want[pr["_key"]][0] and np.array_equal(g[1], want[pr["_key"]][1]) for g, pr in zip(got, P))
            # ... the same problems with some sides in keyframe handles (round 5: a call whose results exceed the mirror pairs the
            # FeatureVectors' nodes in the kernel, whatever mix of handles and arrays its problems are)
            hs = {}
            for S in sets:
                hs[id(S), 1] = pkg.KeyFrameHandle(S["d1"], S["m1"], S["a1"], S["fv1"])
                hs[id(S), 2] = pkg.KeyFrameHandle(S["d2"], S["m2"], S["a2"], S["fv2"])
            Q = []
            for pr in P:
                q = {k: v for k, v in pr.items() if k != "_key"}
                k1, k2, var = pr["_key"][0], pr["_key"][1], pr["_key"][2]
                if rng.random() < 0.6:
                    for f in ("desc1", "mask1", "ang1", "fv1"):
                        q.pop(f)
                    q["kf1"] = hs[k1, 1]
                if rng.random() < 0.4:
                    for f in ("desc2", "mask2", "ang2", "fv2"):
                        q.pop(f, None)
                    q["kf2"] = hs[k2, 2]
                elif not var:
                    q.pop("mask2", None)
                Q.append(q)
            gotk = pkg.search_bow_keyframes(Q)
            ok = ok and all(g[0] == want[pr["_key"]][0] and np.array_equal(g[1], want[pr["_key"]][1]) for g, pr in zip(gotk, P))
            for h in hs.values():
                h.close()
            # ... and a projection batch: the same frame with other queries, other frames with the same query descriptors
            n = size(50, 1500)
            nq = size(1, 1200)
            base = MI.projection_problem(seed, n=n, nq=nq, mode=1, stereo=True, th=3.0, loop="fuse")
            other = MI.projection_problem(seed + 1, n=n, nq=nq, mode=1, stereo=True, th=7.0, loop="fuse")
            B = [base]
            q2 = dict(base)
            for key, v in other.items():
                if key.startswith("q") or key in ("th_high", "taken"):
                    q2[key] = v
            B.append(q2)
            f2 = dict(other)
            f2["qdesc"] = base["qdesc"]
            B.append(f2)
            B.append(base)
            B = [B[int(rng.integers(0, len(B)))] for _ in range(int(rng.choice([2, 5, 12])))]
            gotp = pkg.search_projection_batch(B)
            for pr, g in zip(B, gotp):
                a = O.search_projection(pr)
                ok = ok and a[0] == g[0] and np.array_equal(a[1], g[1]) and np.array_equal(a[2], g[2])
            desc = "batches: bow x %d, projection x %d" % (nb, len(B))
        elif kind == 11:  # ComputeBoW (orbfe_bow_*): both maps against the oracle's, then the resident vector in a search
            k, L = int(rng.integers(2, 11)), int(rng.integers(1, 5))
            if k ** L > 20000:
                L -= 1
            vocab = pkg.synth.make_vocabulary(seed % 100000, k, L, bool(rng.integers(0, 2)))
            vocab["weight"] = vocab["weight"].copy()
            leaves = np.nonzero(vocab["word"] >= 0)[0]
            if rng.random() < 0.6:
                vocab["weight"][leaves[::int(rng.integers(2, 9))]] = 0.0  # stop words
            weighting, scoring = int(rng.integers(0, 4)), int(rng.choice([0, 1, 2, 3, 4, 5]))
            levelsup = int(rng.integers(0, L + 3))
            V = pkg.Vocabulary(vocab)
            V.set_types(weighting, scoring)
            n = int(rng.choice([0, 1, 2, 64, 65])) if rng.random() < 0.15 else size(3, 4000)
            feats = _near_leaf_features(vocab, max(n, 1), rng, int(rng.integers(0, 40)))[:n]
            if n > 20:
                feats[n - n // 10:] = feats[:n // 10]  # exact duplicates: the c-fold sum of addWeight
            B = pkg.Bow(V, max(n, 1) if rng.random() < 0.5 else int(min(65535, max(n, 1) + rng.integers(0, 3000))))
            lazy = bool(rng.integers(0, 2))
            B.set_lazy_norm(lazy)
            want = O.compute_bow(vocab, feats, levelsup, weighting, scoring)
            ondev = n > 0 and bool(rng.integers(0, 2))
            if ondev:
                import torch
                d_f = torch.from_numpy(feats).pin_memory().cuda()
                torch.cuda.synchronize()
            ok = True
            for rep in range(2):  # (the handle reused)
                got = B.compute((d_f.data_ptr(), n) if ondev else feats, levelsup).host()
                (ids, vals), (nodes, offs, ind) = got
                ok = ok and np.array_equal(ids, want[0][0]) and np.array_equal(vals.view(np.uint64), want[0][1].view(np.uint64)) and \
                    np.array_equal(nodes, want[1][0]) and np.array_equal(offs, want[1][1]) and np.array_equal(ind, want[1][2])
            # the vector where ComputeBoW left it, against keyframe handles (one candidate: host node list; many: paired in the kernel)
            ncand = int(rng.choice([1, 3, 40]))
            if n >= 30 and len(want[1][0]) > 0:
                aF = rng.uniform(0, 360, n).astype(np.float32)
                kfs, wants = [], []
                for c in range(min(ncand, 3)):
                    nk = size(30, 1500)
                    dK = _near_leaf_features(vocab, nk, rng, 30)
                    dK[:min(nk, n) // 2] = feats[:min(nk, n) // 2]
                    aK = rng.uniform(0, 360, nk).astype(np.float32)
                    mK = (rng.uniform(size=nk) < 0.7).astype(np.uint8)
                    fvK = O.compute_bow(vocab, dK, levelsup, weighting, scoring)[1]
                    if len(fvK[0]) == 0:  # (every feature of the candidate on a stop word: nothing to search)
                        continue
                    kfs.append(pkg.KeyFrameHandle(dK, mK, aK, fvK))
                    wants.append(O.search_bow_kf_f(dK, mK, aK, fvK, feats, aF, want[1], -1, 0.75, True))
                B.compute((d_f.data_ptr(), n) if ondev else feats, levelsup)  # asynchronous; no host() in between
                gotS = [] if not kfs else pkg.search_bow_keyframes([dict(kf1=kfs[c % len(kfs)], desc2=(d_f.data_ptr(), n) if ondev else feats, ang2=aF, fv2=B,
                                                      variant=0, nnratio=0.75, check_ori=True) for c in range(ncand)])
                ok = ok and all(g[0] == wants[c % len(kfs)][0] and np.array_equal(g[1], wants[c % len(kfs)][1]) for c, g in enumerate(gotS))
                for h in kfs:
                    h.close()
            B.close()
            V.close()
            desc = "compute_bow k=%d L=%d n=%d levelsup=%d w/s=%d/%d lazy=%d dev=%d cand=%d" % (k, L, n, levelsup, weighting, scoring, lazy, ondev, ncand)
        else:            # knn-2 and all-pairs distances, ragged sizes
            nq, nt = size(1, 1800), size(1, 1800)
            Q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            T = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            T[rng.integers(0, nt, max(nt // 10, 1))] = Q[rng.integers(0, nq, max(nt // 10, 1))]
            a, b = O.bfknn2(Q, T), pkg.bfknn2(Q, T)
            ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(O.hamming_matrix(Q, T), pkg.hamming_pairs(Q, T))
            desc = "knn %dx%d" % (nq, nt)
        log(case, desc, "OK" if ok else "MISMATCH")
        if not ok:
            bad.append(desc)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 3,
              log=(lambda *a: None) if os.environ.get("STRESS_QUIET") else print, only=sys.argv[3] if len(sys.argv) > 3 else None)
    print("mismatches:", len(bad))
    sys.exit(1 if bad else 0)
