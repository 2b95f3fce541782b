"""The oracle's set-level DBoW2 transform (BowVector + FeatureVector, oracle/orb_oracle.cpp:orb_oracle_compute_bow) against an
independent pure-Python fold on dicts of Python floats (IEEE doubles, the same sequential additions) -- CPU only.
Reference: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1192, BowVector.cpp:34-86, FeatureVector.cpp:31-45."""
import math

import numpy as np
import pytest


def py_fold(word, node, weight, weighting, scoring):
    v, fv = {}, {}
    tf = weighting in (0, 1)
    for i, (w_id, nd, w) in enumerate(zip(word.tolist(), node.tolist(), weight.tolist())):
        if w > 0:
            if tf:
                v[w_id] = v[w_id] + w if w_id in v else w
            elif w_id not in v:
                v[w_id] = w
            fv.setdefault(nd, []).append(i)
    must = scoring != 5
    if tf and v and not must:
        nd = float(len(v))
        v = {k: x / nd for k, x in v.items()}
    if must:
        norm = 0.0
        for k in sorted(v):
            norm += abs(v[k]) if scoring != 1 else v[k] * v[k]
        if scoring == 1:
            norm = math.sqrt(norm)
        if norm > 0.0:
            v = {k: x / norm for k, x in v.items()}
    ids = sorted(v)
    nodes = sorted(fv)
    offsets = np.zeros(len(nodes) + 1, np.int32)
    for s, nd in enumerate(nodes):
        offsets[s + 1] = offsets[s] + len(fv[nd])
    indices = np.array([i for nd in nodes for i in fv[nd]], np.int32)
    return (np.array(ids, np.uint32), np.array([v[k] for k in ids], np.float64)), (np.array(nodes, np.uint32), offsets, indices)


def near_leaf_features(vocab, n, seed, flips=12):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(vocab["word"] >= 0)[0]
    pick = rng.choice(leaves, size=n)          # with repetition: several features per word
    bits = np.unpackbits(vocab["desc"][pick], axis=1)
    for r in range(n):
        bits[r, rng.permutation(256)[:rng.integers(0, flips)]] ^= 1
    return np.packbits(bits, axis=1)


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (0, 5), (1, 5), (3, 5), (0, 3)])
@pytest.mark.parametrize("levelsup", [0, 2, 4])
def test_oracle_compute_bow_equals_python_fold(oracle, weighting, scoring, levelsup):
    import orb_slam3_detailed_comments_kor_amd as pkg
    vocab = pkg.synth.make_vocabulary(77, 8, 4, True)
    vocab["weight"] = vocab["weight"].copy()
    leaves = np.nonzero(vocab["word"] >= 0)[0]
    vocab["weight"][leaves[::7]] = 0.0          # stop words (:1157)
    feats = near_leaf_features(vocab, 700, 5)
    feats[650:] = feats[:50]                     # exact duplicates: the c-fold sum of addWeight
    (ids, vals), (nodes, offs, ind) = oracle.compute_bow(vocab, feats, levelsup, weighting, scoring)
    w, nid, wt = oracle.vocab_transform(vocab, feats, levelsup)
    (rids, rvals), (rnodes, roffs, rind) = py_fold(w, nid, wt, weighting, scoring)
    assert np.array_equal(ids, rids) and np.array_equal(vals, rvals)          # bit for bit
    assert np.array_equal(nodes, rnodes) and np.array_equal(offs, roffs) and np.array_equal(ind, rind)
    assert (wt == 0).any() and len(ids) < (wt > 0).sum()                       # stop words and shared words both occur
    if scoring == 0 and len(vals):
        assert abs(vals.sum() - 1.0) < 1e-12
    if levelsup == 4:                                                          # L - levelsup <= 0: every feature under the root
        assert list(nodes) == [0]


def test_oracle_compute_bow_empty_and_single(oracle):
    import orb_slam3_detailed_comments_kor_amd as pkg
    vocab = pkg.synth.make_vocabulary(3, 5, 3, False)
    (ids, vals), (nodes, offs, ind) = oracle.compute_bow(vocab, np.zeros((0, 32), np.uint8), 2)
    assert len(ids) == 0 and len(nodes) == 0 and list(offs) == [0]
    f = near_leaf_features(vocab, 1, 1)
    (ids, vals), (nodes, offs, ind) = oracle.compute_bow(vocab, f, 2)
    assert len(ids) == 1 and vals[0] == 1.0 and list(offs) == [0, 1] and list(ind) == [0]
