#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.

The reference ships no tests or golden vectors for this path and cannot be built or imported here
(C++ + OpenCV), so these fixtures are produced by the repo's own CPU oracle (oracle/) on seeded
synthetic inputs -- they pin the oracle against regressions and give the GPU path a fixed target
that does not depend on the oracle binary being rebuilt identically.  Data only: inputs are
re-generated from the seed (their SHA-256 is stored), expected outputs are stored verbatim.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orb_oracle_py as O  # noqa: E402
from orb_slam3_detailed_comments_kor_amd import synth  # noqa: E402
import matcher_inputs as MI  # noqa: E402

CASES = [  # (name, rows, cols, seed, nfeatures, lap)
    ("mono_376x240", 240, 376, 2024, 300, (0, 1000)),
    ("stereo_376x240", 240, 376, 2025, 400, (0, 0)),
    ("fisheye_320x320", 320, 320, 2026, 500, (60, 250)),
]

# round 4: content the rectangle frames do not have (synth.make_frame_kind)
KIND_CASES = [  # (name, rows, cols, seed, nfeatures, lap, kind)
    ("blurred_376x240", 240, 376, 2031, 400, (0, 0), "blurred"),
    ("checker_320x320", 320, 320, 2032, 600, (0, 1000), "checker2"),
    ("plateaus_376x240", 240, 376, 2033, 400, (50, 300), "plateaus"),
]

# round 5: camera imagery (tests/natural.py): two photographs that ship inside scikit-learn, decoded once here, luma committed
NATURAL_CASES = [  # (name, photograph, rows, cols, (oy, ox, flip), nfeatures, lap)
    ("natural_china_640x427", "china", 427, 640, (0, 0, False), 1000, (0, 1000)),      # the photograph itself, mono protocol
    ("natural_flower_640x427", "flower", 427, 640, (0, 0, False), 1200, (0, 0)),       # stereo protocol, dark low-texture frame
    ("natural_china_752x480", "china", 480, 752, (200, 310, False), 1200, (120, 600)),  # EuRoC size, mirror-tiled, lapping range
    ("natural_flower_1280x720", "flower", 720, 1280, (100, 500, True), 2000, (0, 1000)),  # C4 size, x > 1000 in the front block
]

WINDOW_CASES = [("proj_local_map", dict(seed=7101, mode=0, n=600, nq=500)),
                ("proj_last_frame", dict(seed=7102, mode=1, n=600, nq=500, th=7.0, check_orientation=True)),
                ("proj_rig", dict(seed=7103, mode=0, n=640, nq=520, Nleft=350, partners=True, th=3.0)),
                ("fuse_stereo", dict(seed=7104, mode=1, n=600, nq=500, th=3.0, loop="fuse", stereo=True)),
                ("sim3_projection", dict(seed=7105, mode=1, n=600, nq=500, th=4.0, loop="sim3_projection", taken_frac=0.3))]


def extractor_cases(cases):
    for name, rows, cols, seed, nf, lap, kind in cases:
        img = synth.make_frame_kind(rows, cols, seed, kind)
        out = {"image_sha256": np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)}
        for mode, tag in ((O.TRIG_LIBM, "libm"), (O.TRIG_CR, "cr")):
            ex = O.Extractor(nf, 1.2, 8, 20, 7, trig=mode)
            mono, kps, desc = ex.extract(img, lap)
            out["mono_" + tag] = np.int32(mono)
            out["kps_" + tag] = kps
            out["desc_" + tag] = desc
            if tag == "libm":
                out["ncand"] = np.array([len(ex.candidates(l)) for l in range(8)], np.int32)
                out["level3"] = ex.level(3)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, int(out["mono_libm"]), len(out["kps_libm"]))


def natural_luma():
    """Decode the two photographs (scikit-learn + Pillow, build container only) and commit their luma planes."""
    from sklearn.datasets import load_sample_image
    import natural
    planes = {n: natural.rgb_to_luma(load_sample_image(n + ".jpg")) for n in natural.NAMES}
    np.savez_compressed(natural.LUMA_FILE, **planes)
    print("natural_luma", {n: (p.shape, hashlib.sha256(p.tobytes()).hexdigest()[:16]) for n, p in planes.items()})


def natural_cases():
    import natural
    if not os.path.exists(natural.LUMA_FILE):
        natural_luma()
    for name, photo, rows, cols, (oy, ox, flip), nf, lap in NATURAL_CASES:
        img = natural.frame(photo, rows, cols, oy, ox, flip)
        out = {"image_sha256": np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)}
        for mode, tag in ((O.TRIG_LIBM, "libm"), (O.TRIG_CR, "cr")):
            ex = O.Extractor(nf, 1.2, 8, 20, 7, trig=mode)
            mono, kps, desc = ex.extract(img, lap)
            out["mono_" + tag] = np.int32(mono)
            out["kps_" + tag] = kps
            out["desc_" + tag] = desc
            if tag == "libm":
                out["ncand"] = np.array([len(ex.candidates(l)) for l in range(8)], np.int32)
                out["nlevel"] = np.array([len(ex.level_keypoints(l)) for l in range(8)], np.int32)
                out["level3"] = ex.level(3)
                out["level7"] = ex.level(7)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, int(out["mono_libm"]), len(out["kps_libm"]), out["ncand"].tolist())


# round 6: Frame::ComputeBoW (TemplatedVocabulary::transform with both maps): (weighting, scoring, levelsup)
BOW_CASES = [(0, 0, 4), (0, 0, 0), (0, 0, 2), (1, 1, 2), (3, 5, 1), (2, 0, 6), (1, 3, 3)]
BOW_VOCAB = dict(seed=4242, k=9, L=4, ragged=True, stop_every=6)
BOW_FEATURES = dict(n=1500, seed=4243, flips=14, duplicates=120)


def bow_inputs():
    """The vocabulary and the descriptors of the ComputeBoW fixture, from their seeds (tests/test_golden.py makes them again)."""
    vocab = synth.make_vocabulary(BOW_VOCAB["seed"], BOW_VOCAB["k"], BOW_VOCAB["L"], BOW_VOCAB["ragged"])
    vocab["weight"] = vocab["weight"].copy()
    leaves = np.nonzero(vocab["word"] >= 0)[0]
    vocab["weight"][leaves[::BOW_VOCAB["stop_every"]]] = 0.0  # stop words (TemplatedVocabulary.h:1157)
    rng = np.random.default_rng(BOW_FEATURES["seed"])
    n = BOW_FEATURES["n"]
    bits = np.unpackbits(vocab["desc"][rng.choice(leaves, size=n)], axis=1)
    for r in range(n):
        bits[r, rng.permutation(256)[:rng.integers(0, BOW_FEATURES["flips"])]] ^= 1
    feats = np.packbits(bits, axis=1)
    feats[n - BOW_FEATURES["duplicates"]:] = feats[:BOW_FEATURES["duplicates"]]  # exact duplicates: the c-fold sum of addWeight
    return vocab, feats


def bow_cases():
    vocab, feats = bow_inputs()
    out = {"vocab_desc_sha256": np.frombuffer(hashlib.sha256(vocab["desc"].tobytes()).digest(), np.uint8),
           "features_sha256": np.frombuffer(hashlib.sha256(feats.tobytes()).digest(), np.uint8)}
    for w, sc, lu in BOW_CASES:
        (ids, vals), (nodes, offs, ind) = O.compute_bow(vocab, feats, lu, w, sc)
        t = "w%d_s%d_l%d_" % (w, sc, lu)
        out[t + "word_ids"], out[t + "word_value_bits"] = ids, vals.view(np.uint64)  # (the doubles bit for bit)
        out[t + "node_ids"], out[t + "offsets"], out[t + "indices"] = nodes, offs, ind
        print("bow", (w, sc, lu), len(ids), "words", len(nodes), "nodes", int(offs[-1]) if len(offs) else 0, "features kept")
    np.savez_compressed(os.path.join(HERE, "bow_k9L4.npz"), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "bow":  # (only this fixture: the others are untouched)
        bow_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "natural":  # only the fixtures added in round 5
        natural_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "kinds":  # only the fixtures added in round 4
        extractor_cases(KIND_CASES)
        return
    extractor_cases([c + ("rects",) for c in CASES] + KIND_CASES)
    natural_cases()
    # matcher fixture
    d1, d2, a1, a2 = MI.descriptor_sets(400, 380, 77)
    fv1, fv2 = MI.feature_vectors(d1, d2, 77)
    mask1 = (np.random.default_rng(77).uniform(size=400) < 0.6).astype(np.uint8)
    n, m = O.search_bow_kf_f(d1, mask1, a1, fv1, d2, a2, fv2, -1, 0.7, True)
    idx, dist = O.bfknn2(d1[:100], d2)
    np.savez_compressed(os.path.join(HERE, "matcher_400x380.npz"), bow_n=np.int32(n), bow_match=m, knn_idx=idx,
                        knn_dist=dist, d1_sha256=np.frombuffer(hashlib.sha256(d1.tobytes()).digest(), np.uint8))
    print("matcher", n)
    # window searches (SearchByProjection modes, Fuse, SearchForInitialization, SearchForTriangulation_)
    out = {}
    for tag, kw in WINDOW_CASES:
        pr = MI.projection_problem(**kw)
        n, qm, fm = O.search_projection(pr)
        out[tag + "_n"], out[tag + "_q"], out[tag + "_f"] = np.int32(n), qm, fm
    pr = MI.initialization_problem(7001, n1=700, n2=650)
    n, m = O.search_initialization(pr)
    out["init_n"], out["init_m"] = np.int32(n), m
    I = MI.tri_inputs(1100, 1000, 30)
    out["tri_pairs"] = O.search_triangulation(I["d1"], I["has1"], I["kp1"], I["a1"], I["oct1"], I["u1"], I["fv1"], I["d2"],
                                              I["has2"], I["kp2"], I["a2"], I["oct2"], I["u2"], I["fv2"], I["F12"], I["ep"],
                                              I["sf"], I["sig"], False, True, True)
    np.savez_compressed(os.path.join(HERE, "matcher_window_searches.npz"), **out)
    print("window searches", {k: int(v) for k, v in out.items() if k.endswith("_n")}, len(out["tri_pairs"]))
    bow_cases()


if __name__ == "__main__":
    main()
