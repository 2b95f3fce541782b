"""adapters/ORBVocabulary.h through a C++ caller written like System / Frame::ComputeBoW / KeyFrameDatabase: the text-file
loader, transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) from two threads, score() -- BowVector and FeatureVector printed
with hex floats and compared with the oracle's maps bit for bit.  Reference: src/Frame.cc:724-731, src/System.cc:82,
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1192."""
import os
import subprocess

import numpy as np
import pytest

from test_oracle_bow import near_leaf_features

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_text(voc, k, L, path, scoring=0, weighting=0):
    nn = len(voc["word"])
    parent = np.zeros(nn, np.int64)
    for i in range(nn):
        for c in voc["child_ids"][voc["child_off"][i]:voc["child_off"][i + 1]]:
            parent[c] = i
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (k, L, scoring, weighting))
        for i in range(1, nn):
            leaf = int(voc["child_off"][i + 1] == voc["child_off"][i])
            f.write("%d %d %s %r\n" % (parent[i], leaf, " ".join(str(int(b)) for b in voc["desc"][i]), float(voc["weight"][i])))
        f.write("\n")


def _parse(lines, tag):
    bow = [ln for ln in lines if ln.startswith(tag + " bow ")][0].split()[3:]
    ids = np.array([int(t.split(":")[0]) for t in bow], np.uint32)
    vals = np.array([float.fromhex(t.split(":")[1]) for t in bow], np.float64)
    fv = [ln for ln in lines if ln.startswith(tag + " fv ")][0].split()[3:]
    nodes = np.array([int(t.split("[")[0]) for t in fv], np.uint32)
    lists = [[int(x) for x in t.split("[")[1].rstrip("]").split(",")] for t in fv]
    offs = np.zeros(len(nodes) + 1, np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    ind = np.array([x for l in lists for x in l], np.int32)
    return (ids, vals), (nodes, offs, ind)


def _l1_score(a, b):
    da, db = dict(zip(a[0].tolist(), a[1].tolist())), dict(zip(b[0].tolist(), b[1].tolist()))
    s = 0.0
    for k in sorted(set(da) & set(db)):
        s += abs(da[k] - db[k]) - abs(da[k]) - abs(db[k])
    return -s / 2.0


@pytest.mark.parametrize("levelsup", [4, 1])
def test_cpp_vocabulary_adapter(tmp_path, oracle, levelsup):
    import orb_slam3_detailed_comments_kor_amd as pkg
    exe = str(tmp_path / "test_vocabulary_adapter")
    libdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "adapters"),
                           os.path.join(ROOT, "adapters", "test_vocabulary_adapter.cpp"), "-o", exe, "-L" + libdir, "-lorbfe",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    k, L = 9, 5
    voc = pkg.synth.make_vocabulary(21, k, L, True)
    _write_text(voc, k, L, tmp_path / "voc.txt")
    d1 = near_leaf_features(voc, 1200, 1)
    d1[1100:] = d1[:100]
    d2 = near_leaf_features(voc, 900, 2)
    d2[:300] = d1[:300]                       # common words: a score between 0 and 1
    (tmp_path / "d1.raw").write_bytes(d1.tobytes())
    (tmp_path / "d2.raw").write_bytes(d2.tobytes())
    out = subprocess.run([exe, str(tmp_path / "voc.txt"), str(tmp_path / "d1.raw"), "1200", str(levelsup), str(tmp_path / "d2.raw"), "900"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr
    lines = out.stdout.split("\n")
    assert lines[0] == "words %d k %d L %d" % (int((voc["word"] >= 0).sum()), k, L) and lines[-2] == "ok"
    for tag, d in (("A", d1), ("B", d2)):
        got = _parse(lines, tag)
        want = oracle.compute_bow(voc, d, levelsup)
        assert np.array_equal(got[0][0], want[0][0]) and np.array_equal(got[0][1], want[0][1]), tag
        assert all(np.array_equal(x, y) for x, y in zip(got[1], want[1])), tag
    a, b = oracle.compute_bow(voc, d1, levelsup)[0], oracle.compute_bow(voc, d2, levelsup)[0]
    sc = [float.fromhex(t) for t in [ln for ln in lines if ln.startswith("score ")][0].split()[1:]]
    assert sc[0] == _l1_score(a, b) and 0.0 < sc[0] < 1.0 and abs(sc[1] - 1.0) < 1e-12
