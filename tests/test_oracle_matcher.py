"""CPU-only checks of the matcher oracle (KATs from first principles)."""
import numpy as np

import matcher_inputs as MI


def test_descriptor_distance_kat(oracle):
    z = np.zeros(32, np.uint8)
    f = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, f) == 256
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_three_maxima_kat(oracle):
    h = np.zeros(30, np.int32)
    h[[3, 7, 11]] = [50, 40, 4]          # third < 10 % of the first
    assert oracle.three_maxima(h) == (3, 7, -1)
    h[[3, 7, 11]] = [50, 4, 3]           # second < 10 %
    assert oracle.three_maxima(h) == (3, -1, -1)
    h[[3, 7, 11]] = [20, 20, 20]         # strict '>' cascade: first seen wins
    assert oracle.three_maxima(h) == (3, 7, 11)
    assert oracle.three_maxima(np.zeros(30, np.int32)) == (-1, -1, -1)


def test_bfknn2_vs_numpy(oracle):
    d1, d2, _, _ = MI.descriptor_sets(120, 90, 1)
    d2[5] = d2[2]
    idx, dist = oracle.bfknn2(d1, d2)
    D = np.unpackbits(d1[:, None, :] ^ d2[None, :, :], axis=2).sum(axis=2)
    order = np.argsort(D, axis=1, kind="stable")
    assert np.array_equal(idx, order[:, :2])
    assert np.array_equal(dist, np.take_along_axis(D, order[:, :2], 1))


def test_search_bow_properties(oracle):
    d1, d2, a1, a2 = MI.descriptor_sets(600, 700, 2)
    fv1, fv2 = MI.feature_vectors(d1, d2, 2)
    mask = np.ones(600, np.uint8)
    n, m = oracle.search_bow_kf_f(d1, mask, a1, fv1, d2, a2, fv2, -1, 0.7, False)
    assert n == (m >= 0).sum() > 50
    used = m[m >= 0]
    assert len(set(used.tolist())) == len(used)          # a KF feature matches at most one F feature per node
    D = oracle.hamming_matrix(d1, d2)
    for j in np.nonzero(m >= 0)[0]:
        assert D[m[j], j] <= 50                           # TH_LOW
    n2, m2 = oracle.search_bow_kf_f(d1, mask, a1, fv1, d2, a2, fv2, -1, 0.7, True)
    assert n2 <= n and ((m2 == m) | (m2 == -1)).all()     # orientation check only removes matches
    n3, _ = oracle.search_bow_kf_f(d1, np.zeros(600, np.uint8), a1, fv1, d2, a2, fv2, -1, 0.7, True)
    assert n3 == 0                                        # no MapPoints -> no matches


def test_feature_vector_csr(oracle):
    d1, _, _, _ = MI.descriptor_sets(500, 10, 4)
    node_ids, offsets, indices = MI.synth.make_feature_vectors(d1, 9, 6, 2)
    assert (np.diff(node_ids.astype(np.int64)) > 0).all() and offsets[0] == 0 and offsets[-1] == 500
    assert sorted(indices.tolist()) == list(range(500))
    for k in range(len(node_ids)):
        seg = indices[offsets[k]:offsets[k + 1]]
        assert (np.diff(seg) > 0).all()                   # ascending inside a node (push_back order)


def test_vocab_transform_properties(oracle):
    """DBoW2 transform on a synthetic tree: results are leaves, the node id sits `levelsup` levels above
    the leaf level on the root-to-leaf path, and a feature equal to a leaf descriptor finds that leaf."""
    from orb_slam3_detailed_comments_kor_amd import synth
    vocab = synth.make_vocabulary(5, 8, 4, ragged=False)
    nn = len(vocab["word"])
    parent = -np.ones(nn, np.int64)
    for i in range(nn):
        for c in vocab["child_ids"][vocab["child_off"][i]:vocab["child_off"][i + 1]]:
            parent[c] = i
    leaves = np.nonzero(vocab["word"] >= 0)[0]
    feats = vocab["desc"][leaves[:300]]
    w, nid, wt = oracle.vocab_transform(vocab, feats, 2)
    # greedy descent is not guaranteed to reach the generating leaf, but it must reach *a* leaf whose
    # distance to the feature is small, and for most features the generating leaf itself
    assert (w >= 0).all()
    assert (w == vocab["word"][leaves[:300]]).mean() > 0.9
    for f in range(50):
        leaf = int(np.nonzero(vocab["word"] == w[f])[0][0])
        path = [leaf]
        while parent[path[-1]] >= 0:
            path.append(int(parent[path[-1]]))
        path = path[::-1]            # root ... leaf ; level = index
        assert nid[f] == path[vocab["L"] - 2]
        assert wt[f] == vocab["weight"][leaf]
    # levelsup >= L -> node id is the root
    _, nid0, _ = oracle.vocab_transform(vocab, feats[:10], 4)
    assert (nid0 == 0).all()


def test_distinctive_descriptors_against_numpy_spec(oracle):
    """ComputeDistinctiveDescriptors (src/MapPoint.cc:387-419): least median of the sorted distance row
    (self distance included, index int(0.5*(N-1))), first minimum wins."""
    rng = np.random.default_rng(4)
    sizes = [0, 1, 2, 3, 4, 9, 10, 33]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pool = rng.integers(0, 256, (offsets[-1], 32), dtype=np.uint8)
    pool[offsets[5] + 2] = pool[offsets[5] + 6]
    got = oracle.distinctive_descriptors(pool, offsets)
    for p, n in enumerate(sizes):
        if n == 0:
            assert got[p] == -1
            continue
        d = pool[offsets[p]:offsets[p + 1]]
        dist = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(axis=2)
        med = np.sort(dist, axis=1)[:, int(0.5 * (n - 1))]
        assert got[p] == int(np.argmin(med))
