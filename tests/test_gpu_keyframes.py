"""Resident keyframes (orbfe_keyframe_*, round 4): SearchByBoW with one or both sides in handles and SearchForTriangulation_ of one
keyframe against many neighbours in one launch give exactly what the per-call forms give -- which the other matcher tests compare
with the oracle -- and what the oracle gives directly."""
import numpy as np
import pytest

import matcher_inputs as MI

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import orb_slam3_detailed_comments_kor_amd as p
    return p


def _noisy_copy(d1, n2, seed, flip=12, frac=0.6):
    """n2 descriptors of which `frac` are noisy copies of rows of d1 (returns d2 and, per row, its source row or -1)."""
    rng = np.random.default_rng(seed)
    d2 = rng.integers(0, 256, size=(n2, 32), dtype=np.uint8)
    k = int(min(len(d1), n2) * frac)
    src = rng.permutation(len(d1))[:k]
    dst = rng.permutation(n2)[:k]
    bits = np.unpackbits(d1[src], axis=1)
    for r in range(k):
        bits[r, rng.permutation(256)[: rng.integers(0, 2 * flip)]] ^= 1
    d2[dst] = np.packbits(bits, axis=1)
    origin = -np.ones(n2, np.int64)
    origin[dst] = src
    return d2, origin


def test_bow_with_keyframe_handles(pkg, oracle):
    from orb_slam3_detailed_comments_kor_amd import synth
    rng = np.random.default_rng(5)
    kfs, sets = [], []
    dF = rng.integers(0, 256, size=(900, 32), dtype=np.uint8)
    aF = rng.uniform(0, 360, 900).astype(np.float32)
    fvF = synth.make_feature_vectors(dF, 300, 6, 2)  # one vocabulary (seed) for the frame and every keyframe
    for k in range(6):  # six candidate keyframes, one current frame (relocalisation, src/Tracking.cc:3784)
        d, origin = _noisy_copy(dF, 700 + 40 * k, 200 + k)
        a = np.where(origin >= 0, aF[np.maximum(origin, 0)] + rng.normal(0, 4, len(d)), rng.uniform(0, 360, len(d))).astype(np.float32) % 360
        fvK = synth.make_feature_vectors(d, 300, 6, 2)
        mask = (rng.uniform(size=len(d)) < 0.7).astype(np.uint8)
        sets.append((d, mask, a, fvK, fvF))
        kfs.append(pkg.KeyFrameHandle(d, mask, a, fvK))
    # variant 0 (KeyFrame*, Frame&): set 1 in a handle, the frame side passed per call
    probs = [dict(kf1=kfs[k], desc2=dF, ang2=aF, fv2=sets[k][4], variant=0, nnratio=0.75, check_ori=True) for k in range(6)]
    got = pkg.search_bow_keyframes(probs)
    for k in range(6):
        d, mask, a, fvK, fvF = sets[k]
        rn, rm = oracle.search_bow_kf_f(d, mask, a, fvK, dF, aF, fvF, -1, 0.75, True)
        assert got[k][0] == rn and np.array_equal(got[k][1], rm), k
        n1, m1 = pkg.search_bow(d, mask, a, fvK, dF, None, aF, fvF, 0, 0.75, True)
        assert n1 == rn and np.array_equal(m1, rm)
    assert sum(g[0] for g in got) > 300
    # the flags of a keyframe change as the map grows: set_mask, then the same search
    mask2 = (rng.uniform(size=len(sets[2][0])) < 0.3).astype(np.uint8)
    kfs[2].set_mask(mask2)
    d, _, a, fvK, fvF = sets[2]
    rn, rm = oracle.search_bow_kf_f(d, mask2, a, fvK, dF, aF, fvF, -1, 0.75, True)
    n, m = pkg.search_bow_keyframes([probs[2]])[0]
    assert n == rn and np.array_equal(m, rm)
    # ... or travel with the call (what adapters/ORBmatcher.h does: several threads search one keyframe)
    mask3 = (rng.uniform(size=len(sets[2][0])) < 0.5).astype(np.uint8)
    rn, rm = oracle.search_bow_kf_f(d, mask3, a, fvK, dF, aF, fvF, -1, 0.75, True)
    n, m = pkg.search_bow_keyframes([dict(probs[2], mask1=mask3)])[0]
    assert n == rn and np.array_equal(m, rm)
    n, m = pkg.search_bow_keyframes([probs[2]])[0]  # (the handle's own flags are untouched)
    rn, rm = oracle.search_bow_kf_f(d, mask2, a, fvK, dF, aF, fvF, -1, 0.75, True)
    assert n == rn and np.array_equal(m, rm)
    # variant 1 (KeyFrame*, KeyFrame*): both sides in handles (loop closing); per-pair FeatureVectors are the handles' own
    dA, mA, aA, fvA, _ = sets[0]
    dB, mB, aB, _, _ = sets[1]
    fvA2, fvB2 = MI.feature_vectors(dA, dB, 999)
    hA = pkg.KeyFrameHandle(dA, mA, aA, fvA2)
    hB = pkg.KeyFrameHandle(dB, mB, aB, fvB2)
    rn, rm = oracle.search_bow_kf_kf(dA, mA, aA, fvA2, dB, mB, aB, fvB2, -1, -1, 0.8, True)
    (n, m), (nh, mh) = pkg.search_bow_keyframes([dict(kf1=hA, kf2=hB, variant=1, nnratio=0.8, check_ori=True),
                                                  dict(kf1=hA, desc2=dB, mask2=mB, ang2=aB, fv2=fvB2, variant=1, nnratio=0.8)])
    assert n == rn and np.array_equal(m, rm) and nh == rn and np.array_equal(mh, rm)
    for h in kfs + [hA, hB]:
        h.close()


def test_triangulation_search_against_many_neighbours(pkg, oracle):
    from orb_slam3_detailed_comments_kor_amd import synth
    I0 = MI.tri_inputs(1100, 900, 40)
    rng = np.random.default_rng(77)
    fv1 = synth.make_feature_vectors(I0["d1"], 41, 5, 2)  # (= I0["fv1"]: tri_inputs uses seed + 1, branching 5, depth 2)
    assert all(np.array_equal(x, y) for x, y in zip(fv1, I0["fv1"]))
    cur = pkg.KeyFrameHandle(I0["d1"], I0["has1"], I0["a1"], fv1, kp_xy=I0["kp1"], octave=I0["oct1"], uRight=I0["u1"])
    neigh, want, argsets = [], [], []
    for k in range(12):  # the current keyframe against 12 covisible keyframes (src/LocalMapping.cc:556-621)
        n2 = 800 + 30 * k
        d2, origin = _noisy_copy(I0["d1"], n2, 500 + k)
        src = np.maximum(origin, 0)
        kp2 = np.stack([rng.uniform(20, 730, n2), np.where(origin >= 0, I0["kp1"][src, 1] + rng.normal(0, 0.7, n2),
                                                           rng.uniform(20, 460, n2))], 1).astype(np.float32)
        a2 = np.where(origin >= 0, I0["a1"][src] + rng.normal(0, 3, n2), rng.uniform(0, 360, n2)).astype(np.float32) % 360
        oct2 = rng.integers(0, 8, n2).astype(np.int32)
        u2 = np.where(rng.uniform(size=n2) < 0.3, rng.uniform(0, 700, n2), -1).astype(np.float32)
        has2 = (rng.uniform(size=n2) < 0.4).astype(np.uint8)
        fv2 = synth.make_feature_vectors(d2, 41, 5, 2)  # the same vocabulary as the current keyframe's
        flags = dict(only_stereo=(k % 5 == 4), coarse=(k % 4 == 3), check_ori=(k % 3 != 2))
        h = pkg.KeyFrameHandle(d2, has2, a2, fv2, kp_xy=kp2, octave=oct2, uRight=u2)
        F12 = I0["F12"] * np.float32(1.0 + 0.01 * k)
        ep = (900.0 - 10 * k, 250.0)
        neigh.append(dict(kf=h, F12=F12, ep=ep, sf=I0["sf"], sig=I0["sig"], **flags))
        args = (I0["d1"], I0["has1"], I0["kp1"], I0["a1"], I0["oct1"], I0["u1"], fv1, d2, has2, kp2, a2, oct2, u2, fv2, F12, ep,
                I0["sf"], I0["sig"], flags["only_stereo"], flags["coarse"], flags["check_ori"])
        argsets.append(args)
        want.append(pkg.search_triangulation(*args))
        assert np.array_equal(want[-1], oracle.search_triangulation(*args)), k
    got = pkg.search_tri_batch(cur, neigh)
    assert sum(len(g) for g in got) > 200
    for k in range(12):
        assert np.array_equal(got[k], want[k]), k
    # this call's has-MapPoint flags instead of the handles' (current keyframe and two of the neighbours)
    h1 = (rng.uniform(size=len(I0["d1"])) < 0.2).astype(np.uint8)
    ov = {3: (rng.uniform(size=neigh[3]["kf"].n) < 0.6).astype(np.uint8), 7: np.zeros(neigh[7]["kf"].n, np.uint8)}
    got2 = pkg.search_tri_batch(cur, [dict(q, hasMP2=ov.get(k)) for k, q in enumerate(neigh)], hasMP1=h1)
    for k in (0, 3, 7):
        a_ = list(argsets[k])
        a_[1] = h1
        if k in ov:
            a_[8] = ov[k]
        assert np.array_equal(got2[k], oracle.search_triangulation(*a_)), k
    # a handle without keypoints cannot be a side of the triangulation search
    bow_only = pkg.KeyFrameHandle(I0["d1"], I0["has1"], I0["a1"], I0["fv1"])
    with pytest.raises(pkg.OrbfeError):
        pkg.search_tri_batch(bow_only, neigh[:1])
    bow_only.close()
    cur.close()
    for q in neigh:
        q["kf"].close()


def test_matcher_latency_paths_in_their_other_forms():
    # The latency-path calls (SearchByBoW, SearchForTriangulation_, SearchByProjection against resident sets) end on a
    # completion word and write their results straight into the pinned mirror.  ORBFE_SPIN=0 (a deployment switch: stream
    # synchronisation instead of the word) is the one other form left after round 6's pruning: the handle, adapter and matcher
    # tests again under it, in a child process (the switch is read once per process).
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for env in ({"ORBFE_SPIN": "0"},):
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_gpu_keyframes.py"),
                            os.path.join(here, "test_gpu_matcher.py"), "-k", "not other_forms and (bow or tri or projection or handles or neighbours)"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, str(env) + r.stdout[-1500:] + r.stderr[-1500:]
        assert " passed" in r.stdout and "failed" not in r.stdout, str(env) + r.stdout[-500:]


def test_three_threads_search_shared_handles_while_handles_come_and_go(pkg, oracle):
    """ORB-SLAM3 calls ORBmatcher from Tracking, LocalMapping and LoopClosing at once (src/Tracking.cc:2817-2927,
    src/LocalMapping.cc:556-621, src/LoopClosing.cc:470-520).  Three host threads here: projection + BoW searches against a
    resident frame and shared keyframes, the triangulation batch against the same keyframes, and a third that creates and
    destroys keyframe / frame handles (so that pooled device blocks change hands) between KF-KF searches.  Every thread has
    its own staging, completion word and stream; every answer must equal what the same call gave alone (checked against the
    oracle first)."""
    import threading
    from matcher_inputs import projection_problem
    from orb_slam3_detailed_comments_kor_amd import synth
    I0 = MI.tri_inputs(1000, 800, 60)
    rng = np.random.default_rng(91)
    fv1 = I0["fv1"]
    cur = pkg.KeyFrameHandle(I0["d1"], I0["has1"], I0["a1"], fv1, kp_xy=I0["kp1"], octave=I0["oct1"], uRight=I0["u1"])
    neigh, sets = [], []
    for k in range(6):
        n2 = 700 + 50 * k
        d2, origin = _noisy_copy(I0["d1"], n2, 900 + k)
        src = np.maximum(origin, 0)
        kp2 = np.stack([rng.uniform(20, 730, n2), np.where(origin >= 0, I0["kp1"][src, 1] + rng.normal(0, 0.7, n2),
                                                           rng.uniform(20, 460, n2))], 1).astype(np.float32)
        a2 = np.where(origin >= 0, I0["a1"][src] + rng.normal(0, 3, n2), rng.uniform(0, 360, n2)).astype(np.float32) % 360
        oct2 = rng.integers(0, 8, n2).astype(np.int32)
        u2 = np.where(rng.uniform(size=n2) < 0.3, rng.uniform(0, 700, n2), -1).astype(np.float32)
        has2 = (rng.uniform(size=n2) < 0.5).astype(np.uint8)
        fv2 = synth.make_feature_vectors(d2, 61, 5, 2)
        h = pkg.KeyFrameHandle(d2, has2, a2, fv2, kp_xy=kp2, octave=oct2, uRight=u2)
        neigh.append(dict(kf=h, F12=I0["F12"] * np.float32(1.0 + 0.01 * k), ep=(880.0 - 10 * k, 250.0), sf=I0["sf"], sig=I0["sig"],
                          only_stereo=(k == 4), coarse=(k == 3), check_ori=(k != 2)))
        sets.append((d2, has2, a2, fv2))
    base = projection_problem(801, n=1500, nq=10, mode=1, stereo=True, check_orientation=True)
    fr = pkg.ProjectionFrame(base)
    pq = []
    for k, kw in enumerate((dict(mode=1, th=7.0, check_orientation=True), dict(mode=0, th=3.0, nnratio=0.9))):
        q = projection_problem(810 + k, n=1500, nq=700, stereo=True, **kw)
        pr = dict(base)
        for key, v in q.items():
            if key.startswith("q") or key in ("mode", "nnratio", "th_high", "check_orientation", "taken", "chi2_gate", "inv_level_sigma2"):
                pr[key] = v
        tgt = np.random.default_rng(820 + k).integers(0, 1500, 700)
        pr["qx"] = (base["kx"][tgt] + np.random.default_rng(830 + k).normal(0, 2, 700)).astype(np.float32)
        pr["qy"] = (base["ky"][tgt] + np.random.default_rng(840 + k).normal(0, 2, 700)).astype(np.float32)
        bits = np.unpackbits(base["desc"][tgt], axis=1)
        pr["qdesc"] = np.packbits(bits ^ (np.random.default_rng(850 + k).random(bits.shape) < 0.1), axis=1)
        if "qxr" in pr:
            pr["qxr"] = np.where(base["uright"][tgt] > 0, base["uright"][tgt] + 1.0, pr["qx"] - 10).astype(np.float32)
        pq.append(pr)
    # what each call gives alone (and the oracle's word on it)
    wantP = [fr.search(pr) for pr in pq]
    for pr, w in zip(pq, wantP):
        ref = oracle.search_projection(pr)
        assert ref[0] == w[0] and np.array_equal(ref[1], w[1]) and np.array_equal(ref[2], w[2])
    bowProbs = [dict(kf1=neigh[k]["kf"], kf2=cur, variant=1, nnratio=0.8, check_ori=True) for k in range(6)]
    wantB = pkg.search_bow_keyframes(bowProbs)
    for k in range(6):
        d2, has2, a2, fv2 = sets[k]
        rn, rm = oracle.search_bow_kf_kf(d2, has2, a2, fv2, I0["d1"], I0["has1"], I0["a1"], fv1, -1, -1, 0.8, True)
        assert wantB[k][0] == rn and np.array_equal(wantB[k][1], rm), k
    wantT = pkg.search_tri_batch(cur, neigh)
    wantT1 = [pkg.search_tri_batch(cur, [q])[0] for q in neigh]
    for a, b in zip(wantT, wantT1):
        assert np.array_equal(a, b)

    bad, stop = [], threading.Event()

    def guard(fn):
        def run():
            try:
                fn()
            except Exception as e:  # noqa: BLE001 (reported below)
                bad.append(repr(e))
                stop.set()
        return run

    import os
    ROUNDS = int(os.environ.get("ORBFE_TEST_ROUNDS", "500"))  # (a soak run sets more)

    def tracking():
        for it in range(ROUNDS):
            if stop.is_set():
                return
            for pr, w in zip(pq, wantP):
                g = fr.search(pr)
                if g[0] != w[0] or not np.array_equal(g[1], w[1]) or not np.array_equal(g[2], w[2]):
                    bad.append("projection differs in round %d" % it)
            k = it % 6
            g = pkg.search_bow_keyframes([bowProbs[k]])[0]
            if g[0] != wantB[k][0] or not np.array_equal(g[1], wantB[k][1]):
                bad.append("BoW differs in round %d" % it)

    def mapping():
        for it in range(ROUNDS):
            if stop.is_set():
                return
            g = pkg.search_tri_batch(cur, neigh) if it % 2 == 0 else [pkg.search_tri_batch(cur, [neigh[it % 6]])[0]]
            w = wantT if it % 2 == 0 else [wantT1[it % 6]]
            if len(g) != len(w) or any(not np.array_equal(a, b) for a, b in zip(g, w)):
                bad.append("triangulation differs in round %d" % it)

    def closing():
        for it in range(ROUNDS):
            if stop.is_set():
                return
            k = it % 6
            d2, has2, a2, fv2 = sets[k]
            h = pkg.KeyFrameHandle(d2, has2, a2, fv2)                 # a block from the pool ...
            f2 = pkg.ProjectionFrame(base)
            g = pkg.search_bow_keyframes([dict(kf1=h, kf2=cur, variant=1, nnratio=0.8, check_ori=True)])[0]
            gp = f2.search(pq[it % 2])
            h.close()                                                 # ... and back
            f2.close()
            if g[0] != wantB[k][0] or not np.array_equal(g[1], wantB[k][1]):
                bad.append("BoW on a fresh handle differs in round %d" % it)
            w = wantP[it % 2]
            if gp[0] != w[0] or not np.array_equal(gp[1], w[1]) or not np.array_equal(gp[2], w[2]):
                bad.append("projection on a fresh frame differs in round %d" % it)

    ts = [threading.Thread(target=guard(f)) for f in (tracking, mapping, closing)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ts), "a thread hangs"
    assert not bad, bad[:5]
    fr.close()
    cur.close()
    for q in neigh:
        q["kf"].close()


def test_bow_batch_stages_shared_sets_once(pkg, oracle):
    """A call's problems usually share a side (one frame against every relocalisation candidate, src/Tracking.cc:3784): arrays
    that come with the same pointers and sizes are staged and uploaded once.  Mixed here: two keyframes against the same frame,
    a third against another frame, the same arrays again as a KF-KF problem with real flags (a different set: variant 0 reads
    all-ones flags), a set that appears as set 1 of one problem and set 2 of another, and a repeat of the first problem."""
    rng = np.random.default_rng(17)
    dA, dF, aA, aF = MI.descriptor_sets(900, 1000, 31)
    dB, dG, aB, aG = MI.descriptor_sets(700, 800, 32)
    fvA, fvF = MI.feature_vectors(dA, dF, 5)
    fvB, fvG = MI.feature_vectors(dB, dG, 5)
    mA = (rng.uniform(size=900) < 0.6).astype(np.uint8)
    mB = (rng.uniform(size=700) < 0.6).astype(np.uint8)
    mF = (rng.uniform(size=1000) < 0.5).astype(np.uint8)
    P = [dict(desc1=dA, mask1=mA, ang1=aA, fv1=fvA, desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75),
         dict(desc1=dB, mask1=mB, ang1=aB, fv1=fvB, desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75),
         dict(desc1=dA, mask1=mA, ang1=aA, fv1=fvA, desc2=dG, ang2=aG, fv2=fvG, variant=0, nnratio=0.9, check_ori=False),
         dict(desc1=dB, mask1=mB, ang1=aB, fv1=fvB, desc2=dF, mask2=mF, ang2=aF, fv2=fvF, variant=1, nnratio=0.8),
         dict(desc1=dF, mask1=mF, ang1=aF, fv1=fvF, desc2=dA, mask2=mA, ang2=aA, fv2=fvA, variant=1, nnratio=0.8),
         dict(desc1=dA, mask1=mA, ang1=aA, fv1=fvA, desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75)]
    got = pkg.search_bow_batch(P)
    total = 0
    for k, pr in enumerate(P):
        if pr["variant"] == 0:
            rn, rm = oracle.search_bow_kf_f(pr["desc1"], pr["mask1"], pr["ang1"], pr["fv1"], pr["desc2"], pr["ang2"], pr["fv2"], -1,
                                            pr["nnratio"], pr.get("check_ori", True))
        else:
            rn, rm = oracle.search_bow_kf_kf(pr["desc1"], pr["mask1"], pr["ang1"], pr["fv1"], pr["desc2"], pr["mask2"], pr["ang2"],
                                             pr["fv2"], -1, -1, pr["nnratio"], pr.get("check_ori", True))
        assert got[k][0] == rn and np.array_equal(got[k][1], rm), k
        one = pkg.search_bow_batch([pr])[0]
        assert one[0] == rn and np.array_equal(one[1], rm), k
        total += rn
    assert total > 300
    assert got[0][0] == got[5][0] and np.array_equal(got[0][1], got[5][1])
    # sixty problems: the results no longer fit the mirror -- download form, rotation cull and counts by k_bow_cull on the device
    big = pkg.search_bow_batch(P * 10)
    assert sum(len(g[1]) for g in big) * 5 > 256 * 1024
    for k, g in enumerate(big):
        assert g[0] == got[k % 6][0] and np.array_equal(g[1], got[k % 6][1]), k
    # ... also with an empty problem among them (no features on one side: nothing matches, nothing is read)
    empty = dict(P[0], desc2=np.zeros((0, 32), np.uint8), ang2=np.zeros(0, np.float32), fv2=synth_empty_fv())
    mixed = pkg.search_bow_batch(P * 5 + [empty] + P * 5)
    assert mixed[30][0] == 0 and len(mixed[30][1]) == 0
    for k, g in enumerate(mixed[:30] + mixed[31:]):
        assert g[0] == got[k % 6][0] and np.array_equal(g[1], got[k % 6][1]), k


@pytest.mark.parametrize("branching,depth", [(2, 2), (10, 2), (12, 2)])
def test_bow_of_many_candidates_pairs_the_nodes_on_the_device(pkg, oracle, branching, depth):
    """A relocalisation's worth of candidates (src/Tracking.cc:3784) in one call: the results are downloaded, and such a call
    leaves the merge-join of the two FeatureVectors to k_search_bow (round 5) -- keyframes in handles against a frame that comes
    with the call, handles on both sides, and host arrays on both sides; vocabularies of 4 nodes (hundreds of features per node:
    the wide-node path), 100 and 144 nodes (more than one 64-lane chunk of ids per look-up, ids missing on either side).  Every
    problem against the oracle and against the same problem alone (whose small result takes the host's node list)."""
    from orb_slam3_detailed_comments_kor_amd import synth
    rng = np.random.default_rng(1000 + branching)
    NK = 64
    dF = rng.integers(0, 256, size=(1100, 32), dtype=np.uint8)
    aF = rng.uniform(0, 360, 1100).astype(np.float32)
    mF = (rng.uniform(size=1100) < 0.6).astype(np.uint8)
    fvF = synth.make_feature_vectors(dF, 77, branching, depth)
    sets, kfs = [], []
    for k in range(NK):
        n = 1000 + 3 * k  # (64 x ~1100 x 5 bytes of results: above the mirror's 256 KB)
        d, origin = _noisy_copy(dF, n, 300 + k)
        a = np.where(origin >= 0, aF[np.maximum(origin, 0)] + rng.normal(0, 4, n), rng.uniform(0, 360, n)).astype(np.float32) % 360
        if k % 7 == 3:  # a keyframe that misses part of the vocabulary: ids of the frame without a partner, and the reverse
            d[: n // 2] = d[n // 2: 2 * (n // 2)]
        fvK = synth.make_feature_vectors(d, 77, branching, depth)
        mask = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        sets.append((d, mask, a, fvK))
        kfs.append(pkg.KeyFrameHandle(d, mask, a, fvK))
    hF = pkg.KeyFrameHandle(dF, mF, aF, fvF)
    assert sum(len(s[0]) for s in sets) * 5 > 256 * 1024
    want0 = [oracle.search_bow_kf_f(d, m, a, fv, dF, aF, fvF, -1, 0.75, True) for d, m, a, fv in sets]
    want1 = [oracle.search_bow_kf_kf(d, m, a, fv, dF, mF, aF, fvF, -1, -1, 0.8, True) for d, m, a, fv in sets[:NK]]
    assert sum(w[0] for w in want0) > 2000
    # variant 0, set 1 in handles, the frame passed with the call (staged once)
    got = pkg.search_bow_keyframes([dict(kf1=kfs[k], desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75, check_ori=True) for k in range(NK)])
    for k in range(NK):
        assert got[k][0] == want0[k][0] and np.array_equal(got[k][1], want0[k][1]), k
    one = pkg.search_bow_keyframes([dict(kf1=kfs[5], desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75, check_ori=True)])[0]
    assert one[0] == want0[5][0] and np.array_equal(one[1], want0[5][1])
    # variant 1, both sides in handles (the results of variant 1 have set 1's length: 64 x ~1100 again)
    got = pkg.search_bow_keyframes([dict(kf1=kfs[k], kf2=hF, variant=1, nnratio=0.8, check_ori=True) for k in range(NK)])
    for k in range(NK):
        assert got[k][0] == want1[k][0] and np.array_equal(got[k][1], want1[k][1]), k
    # host arrays on both sides, and the three forms mixed in one call
    got = pkg.search_bow_batch([dict(desc1=sets[k][0], mask1=sets[k][1], ang1=sets[k][2], fv1=sets[k][3], desc2=dF, ang2=aF, fv2=fvF,
                                     variant=0, nnratio=0.75) for k in range(NK)])
    for k in range(NK):
        assert got[k][0] == want0[k][0] and np.array_equal(got[k][1], want0[k][1]), k
    mixed = []
    for k in range(NK):
        if k % 3 == 0:
            mixed.append(dict(kf1=kfs[k], kf2=hF, variant=1, nnratio=0.8, check_ori=True))
        elif k % 3 == 1:
            mixed.append(dict(kf1=kfs[k], desc2=dF, mask2=mF, ang2=aF, fv2=fvF, variant=1, nnratio=0.8, check_ori=True))
        else:
            mixed.append(dict(desc1=sets[k][0], mask1=sets[k][1], ang1=sets[k][2], fv1=sets[k][3], kf2=hF, variant=1, nnratio=0.8,
                              check_ori=True))
    got = pkg.search_bow_keyframes(mixed)
    for k in range(NK):
        assert got[k][0] == want1[k][0] and np.array_equal(got[k][1], want1[k][1]), k
    for h in kfs + [hF]:
        h.close()


def synth_empty_fv():
    return (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))


def test_projection_batch_shares_arrays_between_neighbouring_searches(pkg, oracle):
    """The searches of a batch usually share a side (src/LocalMapping.cc:803-870: one keyframe's map points into every
    neighbour = the same query descriptors; every neighbour's points into the one keyframe = the same frame side): a read-only
    array that comes with its predecessor's pointer and size is uploaded once.  Every search must still be what it is alone."""
    from matcher_inputs import projection_problem
    A = projection_problem(901, n=1400, nq=900, mode=1, stereo=True, th=3.0, loop="fuse")
    B = projection_problem(902, n=1100, nq=900, mode=1, stereo=True, th=3.0, loop="fuse")
    C = projection_problem(903, n=1400, nq=700, mode=0, stereo=True, th=3.0)
    same_frame_other_queries = dict(A)
    for k, v in projection_problem(904, n=1400, nq=900, mode=1, stereo=True, th=7.0, loop="fuse").items():
        if k.startswith("q") or k in ("th_high", "taken"):
            same_frame_other_queries[k] = v
    same_queries_other_frame = dict(B)
    same_queries_other_frame["qdesc"] = A["qdesc"]            # (the map points of one keyframe projected into another)
    same_frame_fewer_queries = dict(C)
    for k in ("desc", "kx", "ky", "octave", "angle", "uright"):
        same_frame_fewer_queries[k] = A[k]                    # A's frame side with C's (fewer) queries
    same_frame_fewer_queries["taken"] = A["taken"] if "taken" in A else None
    batch = [A, same_frame_other_queries, same_queries_other_frame, A, same_frame_fewer_queries, B, C]
    got = pkg.search_projection_batch(batch)
    hits = 0
    for k, pr in enumerate(batch):
        ref = oracle.search_projection(pr)
        one = pkg.search_projection(pr)
        for a, b in ((ref, one), (ref, got[k])):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        hits += ref[0]
    assert hits > 500


def test_projection_searches_against_resident_frames_in_one_call(pkg, oracle):
    """orbfe_search_projection_frames (round 5): many searches in one call, every frame side a resident handle.
    (a) relocalisation shape: 24 different query sets (mode 1, one with the orientation cull) against ONE frame handle named 24
    times; (b) fusion shape: one query set against six different handles, two of them rigs with stereo partners, one search in the
    in-order state (non-blocking points + partner writes); (c) a call that mixes both.  Against the oracle, and against the
    one-at-a-time form of the same handles."""
    from matcher_inputs import projection_problem
    base = projection_problem(900, n=1400, nq=2400, mode=1, th=7.0, stereo=True)
    fr = pkg.ProjectionFrame(base)
    rng = np.random.default_rng(17)
    probs = []
    for k in range(24):  # random subsets of the frame's 2400 queries, in random order: the occupancy rule sees another sequence
        idx = rng.permutation(2400)[: 300 + 80 * k]
        pr = dict(base)
        for key in ("qdesc", "qx", "qy", "qr", "qmin_level", "qmax_level", "qxr", "qangle", "qflags", "qblocks"):
            if key in base:
                pr[key] = np.ascontiguousarray(base[key][idx])
        pr["check_orientation"] = int(k == 5)
        pr["taken"] = (rng.random(1400) < 0.1 * (k % 4)).astype(np.uint8)
        probs.append(pr)
    got = pkg.ProjectionFrame.search_many([fr] * len(probs), probs)
    for k, pr in enumerate(probs):
        rn, rq, rf = oracle.search_projection(pr)
        assert got[k][0] == rn and np.array_equal(got[k][1], rq) and np.array_equal(got[k][2], rf), k
        one = fr.search(pr)
        assert one[0] == rn and np.array_equal(one[1], rq)
    # (b) six frames, two of them rigs
    sides = [projection_problem(950 + k, n=900 + 150 * k, nq=700, mode=0, Nleft=(500 if k in (2, 4) else -1), partners=(k in (2, 4)),
                                blocks=(0.5 if k == 4 else None), th=3.0) for k in range(6)]
    frames = [pkg.ProjectionFrame(p_) for p_ in sides]
    got = pkg.ProjectionFrame.search_many(frames, sides)
    for k, pr in enumerate(sides):
        rn, rq, rf = oracle.search_projection(pr)
        assert got[k][0] == rn and np.array_equal(got[k][1], rq) and np.array_equal(got[k][2], rf), ("frames", k)
    # (c) mixed
    got = pkg.ProjectionFrame.search_many([fr, frames[4], fr, frames[0]], [probs[3], sides[4], probs[7], sides[0]])
    for g, pr in zip(got, (probs[3], sides[4], probs[7], sides[0])):
        rn, rq, rf = oracle.search_projection(pr)
        assert g[0] == rn and np.array_equal(g[1], rq) and np.array_equal(g[2], rf)
    assert pkg.ProjectionFrame.search_many([], []) == []
    for f in frames:
        f.close()
    fr.close()


def test_destroyed_handles_are_refused_and_destroy_under_a_search_is_deferred(pkg, oracle):
    """Round 6: orbfe_keyframe_destroy / orbfe_frame_destroy no longer rely on the caller's discipline.
    A handle that has been destroyed is refused by every entry point (ORBFE_ERR_ARGS, not a dereference), and a destroy that
    arrives while another thread's search holds the handle is deferred to that search's return: one thread searches a handle
    over and over while the main thread destroys and re-creates it -- every search either returns the oracle's result or the
    refusal, nothing else."""
    import ctypes as C
    import threading
    from orb_slam3_detailed_comments_kor_amd import synth
    rng = np.random.default_rng(77)
    dF = rng.integers(0, 256, size=(700, 32), dtype=np.uint8)
    aF = rng.uniform(0, 360, 700).astype(np.float32)
    fvF = synth.make_feature_vectors(dF, 11, 6, 2)
    d, origin = _noisy_copy(dF, 800, 5)
    a = rng.uniform(0, 360, 800).astype(np.float32)
    fvK = synth.make_feature_vectors(d, 11, 6, 2)
    mask = np.ones(800, np.uint8)
    want = oracle.search_bow_kf_f(d, mask, a, fvK, dF, aF, fvF, -1, 0.75, True)

    class Stale:  # a handle's address kept after its destruction
        def __init__(self, kf):
            self.h, self.n = C.c_void_p(kf.h.value), kf.n

    kf = pkg.KeyFrameHandle(d, mask, a, fvK)
    stale = Stale(kf)
    kf.close()
    with pytest.raises(pkg.OrbfeError) as e:
        pkg.search_bow_keyframes([dict(kf1=stale, desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75, check_ori=True)])
    assert e.value.code == pkg.binding.ERR_ARGS
    assert pkg.lib().orbfe_keyframe_set_mask(stale.h, mask.ctypes.data) == pkg.binding.ERR_ARGS
    pkg.lib().orbfe_keyframe_destroy(stale.h)  # a second destroy of the same address: ignored

    box = {"kf": pkg.KeyFrameHandle(d, mask, a, fvK)}
    stop, bad, seen = threading.Event(), [], {"ok": 0, "refused": 0}

    def searcher():
        while not stop.is_set():
            h = Stale(box["kf"])
            try:
                n, m = pkg.search_bow_keyframes([dict(kf1=h, desc2=dF, ang2=aF, fv2=fvF, variant=0, nnratio=0.75, check_ori=True)])[0]
                if n != want[0] or not np.array_equal(m, want[1]):
                    bad.append("wrong result")
                seen["ok"] += 1
            except pkg.OrbfeError as ex:
                if ex.code != pkg.binding.ERR_ARGS:
                    bad.append(ex.code)
                seen["refused"] += 1
            except AttributeError:  # (box["kf"].h is None for a moment between close and re-create)
                seen["refused"] += 1
    import time
    t = threading.Thread(target=searcher)
    t.start()
    cycles, t0 = 0, time.time()
    try:
        # at least 150 destroy / re-create cycles AND at least 30 completed searches under them, however the two threads' speeds
        # compare (a fixed number of cycles alone made the number of searches a matter of timing: the searching thread's first call
        # builds its staging, stream and completion word while the cycles run); bounded by a deadline that is itself a failure
        while (cycles < 150 or seen["ok"] < 30) and time.time() - t0 < 60.0:
            old = box["kf"]
            box["kf"] = pkg.KeyFrameHandle(d, mask, a, fvK)
            old.close()
            cycles += 1
    finally:
        stop.set()
        t.join()
    box["kf"].close()
    tally = "%d cycles, %d searches answered, %d refused in %.2f s" % (cycles, seen["ok"], seen["refused"], time.time() - t0)
    print(tally)
    assert not bad, bad[:5]
    assert cycles >= 150 and seen["ok"] >= 30, tally
