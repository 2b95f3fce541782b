"""adapters/ORBmatcher.h keeps the reference's ORBmatcher signatures (include/ORBmatcher.h:39-97): a C++ caller
written like Tracking / LocalMapping / LoopClosing (adapters/test_matcher_adapter.cpp, object graphs of stand-in
KeyFrame / Frame / MapPoint) gets, from every wrapped method, the object state the oracle predicts.

The scenarios are generated here as OBJECT-level data (per-feature map-point states, vocabulary node per feature,
poses, world points); the expectation is derived by this file's own flattening of the same data + the oracle, so the
adapter's toCSR, mask / angle flattening, query construction (projections, radii, level ranges) and write-back are
all checked against an independent implementation.  Geometry uses dyadic numbers so that projections are exact in
every evaluation order."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SF = (1.2 ** np.arange(8)).astype(np.float32)


# ------------------------------------------------------------------ scenario / result files
def _put(f, name, arr):
    arr = np.ascontiguousarray(arr)
    code = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2}[arr.dtype]
    nb = name.encode()
    f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", code, arr.size) + arr.tobytes())


def _read(path):
    out = {}
    raw = open(path, "rb").read()
    o = 0
    while o < len(raw):
        (nl,) = struct.unpack_from("<I", raw, o)
        name = raw[o + 4:o + 4 + nl].decode()
        code, cnt = struct.unpack_from("<II", raw, o + 4 + nl)
        o += 12 + nl
        dt = [np.uint8, np.int32, np.float32][code]
        out[name] = np.frombuffer(raw, dt, cnt, o).copy()
        o += cnt * np.dtype(dt).itemsize
    return out


def _csr_from_nodes(node):
    """DBoW2::FeatureVector as the C++ side builds it from the same per-feature node ids: std::map order (ascending
    node id), features of a node in ascending index order."""
    ids = np.unique(node[node >= 0])
    off = [0]
    ind = []
    for v in ids:
        ind += list(np.nonzero(node == v)[0])
        off.append(len(ind))
    return ids.astype(np.uint32), np.array(off, np.int32), np.array(ind, np.int32)


def _nodes(desc, seed, branching=6):
    """vocabulary node per feature (nearest of `branching`^2 random centroids), -1 for a tenth of them (stop words)."""
    rng = np.random.default_rng(seed)
    cent = rng.integers(0, 256, (branching * branching, 32), dtype=np.uint8)
    D = np.unpackbits(desc[:, None, :] ^ cent[None, :, :], axis=2).sum(axis=2)
    node = (100 + 3 * D.argmin(axis=1)).astype(np.int32)
    node[rng.random(len(desc)) < 0.1] = -1
    return node


def _desc_pair(n1, n2, seed):
    import matcher_inputs as MI
    return MI.descriptor_sets(n1, n2, seed)


# ------------------------------------------------------------------ scenario builders (+ expectation closures)
def _bow_kf_f(f, tag, oracle, seed, n1, n2, nleft, ratio, ori):
    rng = np.random.default_rng(seed)
    d1, d2, a1, a2 = _desc_pair(n1, n2, seed)
    node1 = _nodes(np.concatenate([d1, d2]), seed + 1)
    node1, node2 = node1[:n1].copy(), node1[n1:].copy()
    mp1 = rng.choice([0, 1, 1, 2], n1).astype(np.int32)
    for k, v in (("d1", d1), ("a1", a1), ("node1", node1), ("mp1", mp1), ("d2", d2), ("a2", a2), ("node2", node2),
                 ("Nleft", np.array([nleft], np.int32)), ("ratio", np.array([ratio], np.float32)),
                 ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, v)

    def check(res):
        n, m = oracle.search_bow_kf_f(d1, (mp1 == 1).astype(np.uint8), a1, _csr_from_nodes(node1), d2, a2,
                                      _csr_from_nodes(node2), nleft, ratio, ori)
        assert res[tag + "n"][0] == n and n > 20, tag
        assert np.array_equal(res[tag + "match"], m), tag
        # keyframe handle: created by the first search, hit by the two that follow (one with changed MapPoint flags, which travel
        # with the call); the same results with handles switched off
        assert list(res[tag + "kfhandles"]) == [0, 2, 1], (tag, res[tag + "kfhandles"])
        # searched before ComputeBoW (empty FeatureVector) -> no match and NO handle cached; after it -> the same
        # matches as the keyframe above, through a handle created then
        assert list(res[tag + "fvlate"]) == [0, 0, 1, 1], (tag, res[tag + "fvlate"])
    return check


def _bow_kf_kf(f, tag, oracle, seed, n1, n2, ratio, ori):
    rng = np.random.default_rng(seed)
    d1, d2, a1, a2 = _desc_pair(n1, n2, seed)
    nodes = _nodes(np.concatenate([d1, d2]), seed + 1)
    node1, node2 = nodes[:n1].copy(), nodes[n1:].copy()
    mp1 = rng.choice([0, 1, 1, 2], n1).astype(np.int32)
    mp2 = rng.choice([0, 1, 1, 1, 2], n2).astype(np.int32)
    for k, v in (("d1", d1), ("a1", a1), ("node1", node1), ("mp1", mp1), ("d2", d2), ("a2", a2), ("node2", node2),
                 ("mp2", mp2), ("ratio", np.array([ratio], np.float32)), ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, v)

    def check(res):
        n, m = oracle.search_bow_kf_kf(d1, (mp1 == 1).astype(np.uint8), a1, _csr_from_nodes(node1), d2,
                                       (mp2 == 1).astype(np.uint8), a2, _csr_from_nodes(node2), -1, -1, ratio, ori)
        assert res[tag + "n"][0] == n and n > 20, tag
        assert np.array_equal(res[tag + "match"], m), tag
        assert list(res[tag + "kfhandles"]) == [0, 2, 1], (tag, res[tag + "kfhandles"])  # both keyframes' handles hit again
    return check


def _tri(f, tag, oracle, seed, n1, n2, stereo, coarse, ori):
    rng = np.random.default_rng(seed)
    d1, d2, a1, a2 = _desc_pair(n1, n2, seed)
    nodes = _nodes(np.concatenate([d1, d2]), seed + 1, 5)
    node1, node2 = nodes[:n1].copy(), nodes[n1:].copy()
    cam = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    R2 = np.eye(3)
    t2 = np.array([-0.11, 0.004, 0.02])
    kp1 = np.stack([rng.uniform(20, 730, n1), rng.uniform(20, 460, n1)], 1)
    # keypoints of keyframe 2 = projections of points seen by keyframe 1 (world = camera-1 frame) + noise
    src = rng.integers(0, n1, n2)
    z = rng.uniform(2.0, 12.0, n2)
    X = np.stack([(kp1[src, 0] - cam[2]) / cam[0] * z, (kp1[src, 1] - cam[3]) / cam[1] * z, z], 1)
    Xc2 = X @ R2.T + t2
    kp2 = np.stack([cam[0] * Xc2[:, 0] / Xc2[:, 2] + cam[2], cam[1] * Xc2[:, 1] / Xc2[:, 2] + cam[3]], 1)
    kp2 += rng.normal(0, 0.6, kp2.shape)
    far = rng.random(n2) < 0.3
    kp2[far] = np.stack([rng.uniform(20, 730, far.sum()), rng.uniform(20, 460, far.sum())], 1)
    kp1, kp2 = kp1.astype(np.float32), kp2.astype(np.float32)
    d2 = _noisy(d1[src], rng, 0.0, 0.12)   # the descriptor of a point seen again, a few bits away
    nodes = _nodes(np.concatenate([d1, d2]), seed + 1, 5)
    node1, node2 = nodes[:n1].copy(), nodes[n1:].copy()
    oct1, oct2 = rng.integers(0, 8, n1).astype(np.int32), rng.integers(0, 8, n2).astype(np.int32)
    u1 = np.where(rng.random(n1) < 0.3, rng.uniform(0, 700, n1), -1).astype(np.float32)
    u2 = np.where(rng.random(n2) < 0.3, rng.uniform(0, 700, n2), -1).astype(np.float32)
    mp1 = (rng.random(n1) < 0.4).astype(np.int32)
    mp2 = (rng.random(n2) < 0.4).astype(np.int32)
    for s, (kp, a, oc, d, u, node, mp) in (("1", (kp1, a1, oct1, d1, u1, node1, mp1)), ("2", (kp2, a2, oct2, d2, u2, node2, mp2))):
        for k, v in (("x", kp[:, 0]), ("y", kp[:, 1]), ("a", a), ("oct", oc), ("d", d), ("ur", u), ("node", node), ("mp", mp)):
            _put(f, tag + k + s, v)
    for k, v in (("sf", SF), ("cam1", cam), ("cam2", cam), ("R1", np.eye(3, dtype=np.float32)), ("t1", np.zeros(3, np.float32)),
                 ("O1", np.zeros(3, np.float32)), ("R2", R2.astype(np.float32)), ("t2", t2.astype(np.float32)),
                 ("stereo", np.array([int(stereo)], np.int32)), ("coarse", np.array([int(coarse)], np.int32)),
                 ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, v)

    def check(res):
        K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
        t12 = -(R2.T @ t2.astype(np.float32).astype(np.float64))  # R1 = I, t1 = 0: R12 = R2^T, t12 = -R2^T t2
        tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
        F = np.linalg.inv(K).T @ tx @ R2.T @ np.linalg.inv(K)
        F12 = res[tag + "F12"].reshape(3, 3)
        assert np.allclose(F12, F, rtol=2e-3, atol=1e-9), tag  # the adapter's float pose algebra
        t2f = t2.astype(np.float32).astype(np.float64)
        ep = (cam[0] * t2f[0] / t2f[2] + cam[2], cam[1] * t2f[1] / t2f[2] + cam[3])
        assert np.allclose(res[tag + "ep"], ep, rtol=1e-5), tag
        pairs = oracle.search_triangulation(d1, mp1, kp1, a1, oct1, u1, _csr_from_nodes(node1), d2, mp2, kp2, a2, oct2, u2,
                                            _csr_from_nodes(node2), F12, res[tag + "ep"], SF, SF * SF, stereo, coarse, ori)
        assert res[tag + "n"][0] == len(pairs) and (len(pairs) > 15 or stereo), (tag, len(pairs))
        assert np.array_equal(res[tag + "pairs"].reshape(-1, 2), pairs), tag
        assert res[tag + "n3d"][0] == 0, tag     # the triangulating overload with pinhole cameras: never a pair
        assert list(res[tag + "kfhandles"]) == [0, 4, 1], (tag, res[tag + "kfhandles"])  # two searches x two resident keyframes
    return check


def _tri_kb8(f, tag, oracle, seed, n1, n2, rig, coarse):
    """SearchForTriangulation_ between fisheye keyframes (KannalaBrandt8 gate): a monocular pair, or two-camera rigs with
    the four relative poses of src/ORBmatcher.cc:1238-1248 formed by the adapter from the keyframes' poses."""
    import matcher_inputs as MI
    I = MI.tri_kb8_inputs(n1, n2, seed, rig=rig)
    nodes = _nodes(np.concatenate([I["d1"], I["d2"]]), seed + 1, 5)
    node1, node2 = nodes[:n1].copy(), nodes[n1:].copy()
    I["fv1"], I["fv2"] = _csr_from_nodes(node1), _csr_from_nodes(node2)
    I["u1"], I["u2"] = np.full(n1, -1, np.float32), np.full(n2, -1, np.float32)
    # keyframe poses that give the generator's relative poses: keyframe 1 (left) is the world frame
    R64, t64 = I["R12"].astype(np.float64), I["t12"].astype(np.float64)
    Rll, tll = R64[0], t64[0]
    R2, t2 = Rll.T, -Rll.T @ tll
    arrays = [("R1", np.eye(3)), ("t1", np.zeros(3)), ("O1", np.zeros(3)), ("R2", R2), ("t2", t2)]
    Rrl = trl = None
    if rig:
        Rrl = R64[2] @ Rll.T                      # x_right = Rrl x_left + trl inside a rig
        trl = t64[2] - Rrl @ tll
        arrays += [("R1R", Rrl), ("t1R", trl), ("R2R", Rrl @ R2), ("t2R", Rrl @ t2 + trl),
                   ("NLeft1", np.array([I["Nleft1"]], np.int32)), ("NLeft2", np.array([I["Nleft2"]], np.int32)),
                   ("cam1R", I["P1R"]), ("cam2R", I["P2R"])]
    for s_, (kp, a, oc, d, u, node, mp) in (("1", (I["kp1"], I["a1"], I["oct1"], I["d1"], I["u1"], node1, I["has1"])),
                                           ("2", (I["kp2"], I["a2"], I["oct2"], I["d2"], I["u2"], node2, I["has2"]))):
        for k, v in (("x", kp[:, 0]), ("y", kp[:, 1]), ("a", a), ("oct", oc), ("d", d), ("ur", u), ("node", node),
                     ("mp", mp.astype(np.int32))):
            _put(f, tag + k + s_, np.ascontiguousarray(v))
    for k, v in arrays + [("sf", SF), ("cam1L", I["P1L"]), ("cam2L", I["P2L"]), ("rig", np.array([int(rig)], np.int32)),
                          ("stereo", np.array([0], np.int32)), ("coarse", np.array([int(coarse)], np.int32)),
                          ("ori", np.array([1], np.int32))]:
        _put(f, tag + k, v if np.asarray(v).dtype in (np.int32, np.uint8) else np.asarray(v, np.float32))

    def check(res):
        nposes = 4 if rig else 1
        R12, t12 = res[tag + "R12"].reshape(nposes, 3, 3), res[tag + "t12"].reshape(nposes, 3)
        assert np.allclose(R12, I["R12"], atol=2e-6) and np.allclose(t12, I["t12"], atol=2e-6), tag   # the float pose algebra
        ep = MI.kb8_project64(I["P2L"], (R2 @ np.zeros(3) + t2)[None, :])[0]
        assert np.allclose(res[tag + "ep"], ep, rtol=1e-4), (tag, res[tag + "ep"], ep)
        J = dict(I, R12=R12, t12=t12, ep=res[tag + "ep"])
        pairs = oracle.search_triangulation_kb8(J, coarse=coarse)
        assert res[tag + "n"][0] == len(pairs) and len(pairs) > 40, (tag, len(pairs))
        assert np.array_equal(res[tag + "pairs"].reshape(-1, 2), pairs), tag
        if not coarse:
            assert len(pairs) < len(oracle.search_triangulation_kb8(J, coarse=True)), tag   # the gate rejected something
        # the overload that also triangulates (:1452-1641), on the poses the keyframes were given
        T = [np.hstack([np.eye(3), np.zeros((3, 1))]), None, np.hstack([R2, t2[:, None]]), None]
        if rig:
            T[1] = np.hstack([Rrl, trl[:, None]])
            T[3] = np.hstack([Rrl @ R2, (Rrl @ t2 + trl)[:, None]])
        else:
            T[1], T[3] = T[0], T[2]
        J3 = dict(I, Tcw=np.stack(T).astype(np.float32))
        p3, x3 = oracle.search_triangulation_3d(J3)
        assert res[tag + "n3d"][0] == len(p3) and len(p3) > 40, (tag, len(p3))
        assert np.array_equal(res[tag + "pairs3d"].reshape(-1, 2), p3), tag
        assert np.allclose(res[tag + "points3d"].reshape(-1, 3), x3, rtol=2e-5, atol=2e-5), tag
    return check


GRID = dict(minX=np.float32(0), minY=np.float32(0), maxX=np.float32(768), maxY=np.float32(512))


def _frame_arrays(rng, n, stereo):
    kx = rng.uniform(2, 766, n).astype(np.float32)
    ky = rng.uniform(2, 510, n).astype(np.float32)
    k = n // 4  # clusters: several features inside one search window
    c = rng.integers(0, 12, k)
    cx, cy = rng.uniform(60, 700, 12), rng.uniform(60, 450, 12)
    kx[:k] = (cx[c] + rng.normal(0, 6, k)).astype(np.float32)
    ky[:k] = (cy[c] + rng.normal(0, 6, k)).astype(np.float32)
    d = dict(kx=kx, ky=ky, oct=rng.integers(0, 8, n).astype(np.int32), desc=rng.integers(0, 256, (n, 32), dtype=np.uint8),
             ang=rng.uniform(0, 360, n).astype(np.float32))
    if stereo:
        d["uright"] = np.where(rng.random(n) < 0.6, kx - rng.uniform(1, 40, n), -1).astype(np.float32)
    return d


def _put_frame(f, tag, fr, fstate):
    for k in ("kx", "ky", "oct", "desc", "ang"):
        _put(f, tag + k, fr[k])
    if "uright" in fr:
        _put(f, tag + "uright", fr["uright"])
    _put(f, tag + "fstate", fstate)
    _put(f, tag + "grid", np.array([0, 0, 768, 512, np.float32(64) / np.float32(768), np.float32(48) / np.float32(512)], np.float32))
    _put(f, tag + "sf", SF)


def _frame_problem(fr, fstate):
    pr = dict(desc=fr["desc"], kx=fr["kx"], ky=fr["ky"], octave=fr["oct"], angle=fr["ang"], Nleft=-1,
              minX=np.float32(0), minY=np.float32(0), gridWInv=np.float32(64) / np.float32(768),
              gridHInv=np.float32(48) / np.float32(512), taken=((fstate == 1) | (fstate == 2)).astype(np.uint8))
    pr["uright"] = fr["uright"] if "uright" in fr else np.full(len(fr["kx"]), -1, np.float32)
    return pr


def _noisy(desc, rng, lo=0.0, hi=0.25):
    bits = np.unpackbits(desc, axis=1)
    return np.packbits(bits ^ (rng.random(bits.shape) < rng.uniform(lo, hi, (len(desc), 1))), axis=1)


def _expect_points(fstate, feat_match, q_match, ids):
    exp = np.where(fstate > 0, 100000 + np.arange(len(fstate)), -1).astype(np.int32)
    for k, fidx in enumerate(q_match):            # written by a point and holding none now: cleared by the cull
        if fidx >= 0 and feat_match[fidx] < 0:
            exp[fidx] = -1
    w = feat_match >= 0
    exp[w] = np.asarray(ids, np.int32)[feat_match[w]]
    return exp


def _proj_local(f, tag, oracle, seed, n, m, th, stereo, far):
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, stereo)
    fstate = rng.choice([0, 0, 0, 0, 0, 0, 1, 3], n).astype(np.int32)   # 1 point with observations, 3 without
    _put_frame(f, tag, fr, fstate)
    tgt = rng.integers(0, n, m)
    tgt[: m // 3] = rng.integers(0, max(n // 4, 1), m // 3)              # compete inside the clusters
    tgt[m // 3: m // 2] = tgt[: m // 2 - m // 3]
    pdesc = _noisy(fr["desc"][tgt], rng)
    px = (fr["kx"][tgt] + rng.normal(0, 2.0, m)).astype(np.float32)
    py = (fr["ky"][tgt] + rng.normal(0, 2.0, m)).astype(np.float32)
    level = np.clip(fr["oct"][tgt] + rng.integers(0, 2, m), 0, 7).astype(np.int32)
    pcos = np.where(rng.random(m) < 0.5, 0.9995, 0.97).astype(np.float32)
    pxr = np.where(fr.get("uright", np.full(n, -1, np.float32))[tgt] > 0,
                   fr.get("uright", np.zeros(n, np.float32))[tgt] + rng.normal(0, 3.0, m), px - 10).astype(np.float32)
    depth = rng.uniform(1, 60, m).astype(np.float32)
    pstate = np.where(rng.random(m) < 0.85, 1, 0).astype(np.int32)        # bit0: in view
    pstate |= (rng.random(m) < 0.06).astype(np.int32) << 1               # bit1: bad
    pstate |= (rng.random(m) < 0.15).astype(np.int32) << 2               # bit2: no observations
    thfar = np.float32(45.0)
    for k, v in (("px", px), ("py", py), ("pxr", pxr), ("pcos", pcos), ("pdepth", depth), ("plevel", level), ("pdesc", pdesc),
                 ("pstate", pstate), ("ratio", np.array([0.8], np.float32)), ("th", np.array([th], np.float32)),
                 ("far", np.array([int(far)], np.int32)), ("thfar", np.array([thfar], np.float32))):
        _put(f, tag + k, v)

    def check(res):
        keep = [k for k in range(m) if (pstate[k] & 1) and not (far and depth[k] > thfar) and not (pstate[k] & 2)]
        keep = np.array(keep)
        r = np.where(pcos[keep] > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)
        if th != 1.0:
            r = (r * np.float32(th)).astype(np.float32)
        pr = _frame_problem(fr, fstate)
        pr.update(mode=0, nnratio=0.8, th_high=100, check_orientation=0, qdesc=pdesc[keep], qx=px[keep], qy=py[keep],
                  qr=(r * SF[level[keep]]).astype(np.float32), qmin_level=(level[keep] - 1).astype(np.int32),
                  qmax_level=level[keep], qxr=pxr[keep], qangle=np.zeros(len(keep), np.float32),
                  qblocks=((pstate[keep] & 4) == 0).astype(np.uint8))
        nm, qm, fm = oracle.search_projection(pr)
        assert res[tag + "n"][0] == nm and nm > 30, (tag, nm)
        assert np.array_equal(res[tag + "points"], _expect_points(fstate, fm, [-1] * 0, keep)), tag
    return check


def _proj_last(f, tag, oracle, seed, n, nl, th, stereo, tlz, mono, ori):
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, stereo)
    fstate = rng.choice([0, 0, 0, 0, 0, 0, 1, 3], n).astype(np.int32)
    _put_frame(f, tag, fr, fstate)
    cam = np.array([512, 512, 384, 256], np.float32)
    bf, mb = np.float32(40.0), np.float32(0.125)
    tc = np.array([0.25, -0.5, 0.0], np.float32)
    tl = np.array([0.25, -0.5, tlz], np.float32)  # tlc = Rlw (-Rcw^T tcw) + tlw = (0, 0, tlz)
    tgt = rng.integers(0, n, nl)
    tgt[: nl // 3] = rng.integers(0, max(n // 4, 1), nl // 3)
    u = np.round((fr["kx"][tgt] + rng.normal(0, 2.0, nl)) * 8) / 8          # dyadic pixels: exact projections
    v = np.round((fr["ky"][tgt] + rng.normal(0, 2.0, nl)) * 8) / 8
    out = rng.random(nl) < 0.04
    u[out] = rng.choice([-40.0, 800.0], out.sum())                         # outside the image bounds: skipped
    z = rng.choice([2.0, 4.0, 8.0], nl)
    z[rng.random(nl) < 0.04] = -2.0                                        # behind the camera: skipped
    xc = np.stack([(u - 384) * z / 512, (v - 256) * z / 512, z], 1)
    lpos = (xc - tc.astype(np.float64)).astype(np.float32)                  # Rcw = I: x_c = x_w + tcw
    assert np.array_equal(lpos.astype(np.float64) + tc, xc)
    lstate = rng.choice([0, 1, 1, 1, 1, 2, 3], nl).astype(np.int32)
    loct = np.clip(fr["oct"][tgt] + rng.integers(-1, 2, nl), 0, 7).astype(np.int32)
    lang = rng.uniform(0, 360, nl).astype(np.float32)
    ldesc = _noisy(fr["desc"][tgt], rng)
    for k, val in (("cam", cam), ("bf", np.array([bf, mb], np.float32)), ("Rc", np.eye(3, dtype=np.float32)), ("tc", tc),
                   ("Rl", np.eye(3, dtype=np.float32)), ("tl", tl), ("loct", loct), ("lang", lang), ("lstate", lstate),
                   ("lpos", lpos), ("ldesc", ldesc), ("th", np.array([th], np.float32)),
                   ("mono", np.array([int(mono)], np.int32)), ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, val)

    def check(res):
        fwd, bwd = (tlz > mb and not mono), (-tlz > mb and not mono)
        keep = [i for i in range(nl) if lstate[i] in (1, 3) and z[i] > 0 and 0 <= u[i] <= 768 and 0 <= v[i] <= 512]
        keep = np.array(keep)
        o = loct[keep]
        lo = o if fwd else (np.zeros_like(o) if bwd else o - 1)
        hi = np.full_like(o, -1) if fwd else (o if bwd else o + 1)
        pr = _frame_problem(fr, fstate)
        pr.update(mode=1, nnratio=0.9, th_high=100, check_orientation=int(ori), qdesc=ldesc[keep],
                  qx=u[keep].astype(np.float32), qy=v[keep].astype(np.float32), qr=(np.float32(th) * SF[o]).astype(np.float32),
                  qmin_level=lo.astype(np.int32), qmax_level=hi.astype(np.int32),
                  qxr=(u[keep] - 40.0 / z[keep]).astype(np.float32), qangle=lang[keep],
                  qblocks=(lstate[keep] == 1).astype(np.uint8))
        nm, qm, fm = oracle.search_projection(pr)
        assert res[tag + "n"][0] == nm and nm > 30, (tag, nm)
        assert np.array_equal(res[tag + "points"], _expect_points(fstate, fm, qm, keep)), tag
        assert list(res[tag + "handles"]) == [0, 3, 1], tag   # Tracking's three searches of one frame: one upload
    return check


def _proj_local_rig(f, tag, oracle, seed, n, m, th):
    """Tracking::SearchLocalPoints on a two-camera frame: per point a left-camera search and / or a right-camera one
    (src/ORBmatcher.cc:58-139, :141-195), stereo partners through mvLeftToRightMatch / mvRightToLeftMatch."""
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, False)
    nleft = n * 3 // 5
    fstate = rng.choice([0, 0, 0, 0, 0, 0, 1], n).astype(np.int32)
    tl = rng.integers(0, nleft, m)
    tl[: m // 3] = rng.integers(0, max(nleft // 4, 1), m // 3)
    tr = rng.integers(nleft, n, m)
    tr[m // 3: m // 2] = tr[: m // 2 - m // 3]
    pdesc = _noisy(fr["desc"][tl], rng)
    for k in range(m):                                                  # the right targets look like their points too
        fr["desc"][tr[k]] = _noisy(pdesc[k:k + 1], rng, 0.0, 0.15)[0]
    _put_frame(f, tag, fr, fstate)
    nr = n - nleft
    l2r, r2l = np.full(nleft, -1, np.int32), np.full(nr, -1, np.int32)
    li, ri = rng.permutation(nleft)[: nr // 3], rng.permutation(nr)[: nr // 3]
    l2r[li], r2l[ri] = ri, li
    px = (fr["kx"][tl] + rng.normal(0, 2.0, m)).astype(np.float32)
    py = (fr["ky"][tl] + rng.normal(0, 2.0, m)).astype(np.float32)
    pxr = (fr["kx"][tr] + rng.normal(0, 2.0, m)).astype(np.float32)
    pyr = (fr["ky"][tr] + rng.normal(0, 2.0, m)).astype(np.float32)
    level = np.clip(fr["oct"][tl] + rng.integers(0, 2, m), 0, 7).astype(np.int32)
    levelr = np.clip(fr["oct"][tr] + rng.integers(0, 2, m), 0, 7).astype(np.int32)
    levelr[rng.random(m) < 0.1] = -1                                     # no prediction for the right camera: no search
    pcos = np.where(rng.random(m) < 0.5, 0.9995, 0.97).astype(np.float32)
    pcosr = np.where(rng.random(m) < 0.5, 0.9995, 0.97).astype(np.float32)
    pstate = np.where(rng.random(m) < 0.8, 1, 0).astype(np.int32)        # bit0: in view (left)
    pstate |= (rng.random(m) < 0.05).astype(np.int32) << 1              # bit1: bad
    pstate |= (rng.random(m) < 0.7).astype(np.int32) << 3               # bit3: in view of the right camera
    for k, v in (("Nleft", np.array([nleft], np.int32)), ("l2r", l2r), ("r2l", r2l), ("px", px), ("py", py), ("pxr", pxr),
                 ("pyR", pyr), ("pcos", pcos), ("pcosR", pcosr), ("pdepth", np.ones(m, np.float32)), ("plevel", level),
                 ("plevelR", levelr), ("pdesc", pdesc), ("pstate", pstate), ("ratio", np.array([0.8], np.float32)),
                 ("th", np.array([th], np.float32)), ("far", np.array([0], np.int32)), ("thfar", np.array([50.0], np.float32))):
        _put(f, tag + k, v)

    def check(res):
        rad = lambda c: np.float32(2.5) if c > 0.998 else np.float32(4.0)
        Q = dict(ids=[], x=[], y=[], r=[], lo=[], hi=[], fl=[])
        for k in range(m):
            if not (pstate[k] & 9) or (pstate[k] & 2):
                continue
            left = False
            if pstate[k] & 1:
                r = rad(pcos[k])
                if th != 1.0:
                    r = np.float32(r * np.float32(th))
                for key, v in (("ids", k), ("x", px[k]), ("y", py[k]), ("r", np.float32(r * SF[level[k]])), ("lo", level[k] - 1),
                               ("hi", level[k]), ("fl", 0)):
                    Q[key].append(v)
                left = True
            if (pstate[k] & 8) and levelr[k] != -1:
                for key, v in (("ids", k), ("x", pxr[k]), ("y", pyr[k]), ("r", np.float32(rad(pcosr[k]) * SF[levelr[k]])),
                               ("lo", levelr[k] - 1), ("hi", levelr[k]), ("fl", 1 | (2 if left else 0))):
                    Q[key].append(v)
        ids = np.array(Q["ids"])
        pr = _frame_problem(fr, fstate)
        pr.update(Nleft=nleft, left_to_right=l2r, right_to_left=r2l, mode=0, nnratio=0.8, th_high=100, check_orientation=0,
                  qdesc=pdesc[ids], qx=np.array(Q["x"], np.float32), qy=np.array(Q["y"], np.float32),
                  qr=np.array(Q["r"], np.float32), qmin_level=np.array(Q["lo"], np.int32), qmax_level=np.array(Q["hi"], np.int32),
                  qxr=np.zeros(len(ids), np.float32), qflags=np.array(Q["fl"], np.uint8),
                  qangle=np.zeros(len(ids), np.float32), qblocks=np.ones(len(ids), np.uint8))
        pr.pop("uright")
        nm, qm, fm = oracle.search_projection(pr)
        assert res[tag + "n"][0] == nm and nm > 60, (tag, nm)
        assert (fm[nleft:] >= 0).sum() > 20 and (np.array(Q["fl"]) == 3).sum() > 20, tag
        assert np.array_equal(res[tag + "points"], _expect_points(fstate, fm, [], ids)), tag
    return check


def _proj_last_rig(f, tag, oracle, seed, n, nl, th, tlz, ori):
    """Tracking::TrackWithMotionModel on a two-camera frame: every point of the last frame is also searched in the right
    camera (src/ORBmatcher.cc:2326-2395), unless its left window held no feature (:2255)."""
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, False)
    nleft = n * 3 // 5
    fstate = rng.choice([0, 0, 0, 0, 0, 0, 1, 3], n).astype(np.int32)
    cam = np.array([512, 512, 384, 256], np.float32)
    tc = np.array([0.25, -0.5, 0.0], np.float32)
    tl = np.array([0.25, -0.5, tlz], np.float32)
    trl = np.array([[1, 0, 0, -0.125], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)   # right camera 0.125 to the right
    tgt = rng.integers(0, nleft, nl)
    tgt[: nl // 3] = rng.integers(0, max(nleft // 4, 1), nl // 3)
    u = np.round((fr["kx"][tgt] + rng.normal(0, 2.0, nl)) * 8) / 8
    v = np.round((fr["ky"][tgt] + rng.normal(0, 2.0, nl)) * 8) / 8
    lonely = rng.random(nl) < 0.15                                        # anywhere: mostly an empty left window
    u[lonely] = np.round(rng.uniform(20, 740, lonely.sum()) * 8) / 8
    v[lonely] = np.round(rng.uniform(20, 490, lonely.sum()) * 8) / 8
    z = rng.choice([2.0, 4.0, 8.0], nl)
    z[rng.random(nl) < 0.03] = -2.0
    ur = u - 64.0 / z                                                     # fx * (-0.125) / z: exact
    xc = np.stack([(u - 384) * z / 512, (v - 256) * z / 512, z], 1)
    lpos = (xc - tc.astype(np.float64)).astype(np.float32)
    assert np.array_equal(lpos.astype(np.float64) + tc, xc)
    lstate = rng.choice([0, 1, 1, 1, 1, 2, 3], nl).astype(np.int32)
    loct = np.clip(fr["oct"][tgt] + rng.integers(-1, 2, nl), 0, 7).astype(np.int32)
    lang = rng.uniform(0, 360, nl).astype(np.float32)
    ldesc = _noisy(fr["desc"][tgt], rng)
    # the right camera's features: most of them are the last frame's points seen from there
    nr = n - nleft
    src = np.where(rng.random(nr) < 0.8, rng.integers(0, nl, nr), -1)
    for j in range(nr):
        if src[j] >= 0 and z[src[j]] > 0:
            p = src[j]
            fr["kx"][nleft + j] = np.float32(ur[p] + rng.normal(0, 1.5))
            fr["ky"][nleft + j] = np.float32(v[p] + rng.normal(0, 1.5))
            fr["oct"][nleft + j] = np.clip(loct[p] + rng.integers(-1, 2), 0, 7)
            fr["desc"][nleft + j] = _noisy(ldesc[p:p + 1], rng, 0.0, 0.2)[0]
    _put_frame(f, tag, fr, fstate)
    for k, val in (("Nleft", np.array([nleft], np.int32)), ("Trl", trl), ("cam", cam), ("bf", np.array([40.0, 0.125], np.float32)),
                   ("Rc", np.eye(3, dtype=np.float32)), ("tc", tc), ("Rl", np.eye(3, dtype=np.float32)), ("tl", tl),
                   ("loct", loct), ("lang", lang), ("lstate", lstate), ("lpos", lpos), ("ldesc", ldesc),
                   ("th", np.array([th], np.float32)), ("mono", np.array([0], np.int32)), ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, val)

    def check(res):
        fwd, bwd = tlz > 0.125, -tlz > 0.125
        keep = [i for i in range(nl) if lstate[i] in (1, 3) and z[i] > 0 and 0 <= u[i] <= 768 and 0 <= v[i] <= 512]
        ids = np.repeat(keep, 2)
        o = loct[ids]
        lo = o if fwd else (np.zeros_like(o) if bwd else o - 1)
        hi = np.full_like(o, -1) if fwd else (o if bwd else o + 1)
        qx = np.stack([u[keep], ur[keep]], 1).reshape(-1).astype(np.float32)
        pr = _frame_problem(fr, fstate)
        pr.update(Nleft=nleft, mode=1, nnratio=0.9, th_high=100, check_orientation=int(ori), qdesc=ldesc[ids], qx=qx,
                  qy=v[ids].astype(np.float32), qr=(np.float32(th) * SF[o]).astype(np.float32), qmin_level=lo.astype(np.int32),
                  qmax_level=hi.astype(np.int32), qxr=np.zeros(len(ids), np.float32), qangle=lang[ids],
                  qflags=np.tile(np.array([0, 5], np.uint8), len(keep)), qblocks=(lstate[ids] == 1).astype(np.uint8))
        pr.pop("uright")
        nm, qm, fm = oracle.search_projection(pr)
        pr["qflags"] = pr["qflags"] & 1
        nm_all, _, _ = oracle.search_projection(pr)                         # (the rule of :2255 matters in this scenario)
        assert nm_all > nm, (tag, nm, nm_all)
        assert res[tag + "n"][0] == nm and nm > 60 and (fm[nleft:] >= 0).sum() > 20, (tag, nm)
        assert np.array_equal(res[tag + "points"], _expect_points(fstate, fm, qm, ids)), tag
        assert list(res[tag + "handles"]) == [0, 3, 1], tag   # three more searches of the frame: no upload, three handle hits, same results
    return check


def _fuse(f, tag, oracle, seed, n, m, th, rig=None, dup=False):
    """rig = None, "left" or "right": a keyframe of a two-camera rig, fused into through its left / right camera
    (bRight, src/ORBmatcher.cc:1647-1658: right pose, mpCamera2, right grid, features NLeft..).
    dup: a tenth of the candidates are the SAME MapPoint object as an earlier entry of the list (what
    GetMapPointMatches() of a rig keyframe holds at the left and the right index): the reference tests isBad() /
    IsInKeyFrame() per iteration (:1690-1692), so the second occurrence is skipped once the first one fused."""
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, True)
    fr["uright"] = np.where(rng.random(n) < 0.5, fr["kx"] - rng.uniform(0, 30, n), -1).astype(np.float32)
    nleft = n * 3 // 5 if rig else -1
    lo_f, hi_f = (nleft, n) if rig == "right" else (0, nleft if rig else n)   # the features this camera owns
    cx = 400.0 if rig == "right" else 384.0
    tx = -0.125 if rig == "right" else 0.0                                 # the right camera sits 0.125 to the right
    if rig == "right":                                                      # the clusters of _frame_arrays, seen by this camera too
        fr["kx"][nleft:nleft + n // 8] = fr["kx"][:n // 8]
        fr["ky"][nleft:nleft + n // 8] = fr["ky"][:n // 8]
    fstate = rng.choice([0, 0, 1, 1, 2], n).astype(np.int32)               # none / good / bad point at the feature
    fobs = rng.integers(1, 6, n).astype(np.int32)
    _put_frame(f, tag, fr, fstate)
    _put(f, tag + "fobs", fobs)
    cam = np.array([512, 512, 384, 256], np.float32)
    bf = np.float32(40.0)
    tgt = rng.integers(lo_f, hi_f, m)
    tgt[: m // 3] = rng.integers(lo_f, lo_f + max((hi_f - lo_f) // 4, 1), m // 3)
    tgt[m // 3: m // 2] = tgt[: m // 2 - m // 3]                            # several candidates fuse into one feature
    u = np.round((fr["kx"][tgt] + rng.normal(0, 1.0, m)) * 8) / 8
    v = np.round((fr["ky"][tgt] + rng.normal(0, 1.0, m)) * 8) / 8
    out = rng.random(m) < 0.04
    u[out] = 900.0                                                         # not IsInImage
    z = rng.choice([2.0, 4.0, 8.0], m)
    z[rng.random(m) < 0.04] = -4.0                                         # negative depth
    xc = np.stack([(u - cx) * z / 512, (v - 256) * z / 512, z], 1)        # in the camera that is fused into
    pos = (xc - np.array([tx, 0, 0])).astype(np.float32)                   # R = I, t = (tx, 0, 0), O = (-tx, 0, 0)
    assert np.array_equal(pos.astype(np.float64) + np.array([tx, 0, 0]), xc)
    dist = np.linalg.norm(xc, axis=1)
    level = np.clip(fr["oct"][tgt] + rng.integers(0, 2, m), 0, 7)
    maxd = (dist * 1.2 ** (level - 0.5)).astype(np.float32)               # PredictScale = ceil(level - 0.5) = level
    mind = (maxd / np.float32(1.2 ** 7)).astype(np.float32)
    wrong = rng.random(m) < 0.05
    maxd[wrong] = (dist[wrong] * 0.5).astype(np.float32)                   # outside the scale-invariance range
    normal = (xc / np.maximum(dist, 1e-9)[:, None])
    flip = rng.random(m) < 0.05
    normal[flip] *= -1                                                     # viewing angle > 60 degrees
    pstate = rng.choice([0, 1, 1, 1, 1, 1, 1, 2, 3], m).astype(np.int32)
    pobs = rng.integers(1, 6, m).astype(np.int32)
    pdesc = _noisy(fr["desc"][tgt], rng, 0.0, 0.2)
    pdup = np.full(m, -1, np.int32)
    if dup:
        for q in np.sort(rng.choice(np.arange(1, m), m // 10, replace=False)):  # ascending: a root's state is final
            r = int(rng.integers(0, q))
            r = int(pdup[r]) if pdup[r] >= 0 else r
            if pstate[r] == 0 or pstate[q] == 0:
                continue
            pdup[q] = r                                                     # the same object: every property is the root's
            for arr in (u, v, z, pos, xc, dist, level, maxd, mind, wrong, normal, flip, pstate, pobs, pdesc, out):
                arr[q] = arr[r]
        _put(f, tag + "pdup", pdup)
    for k, val in (("cam", cam), ("bf", np.array([bf], np.float32)), ("R", np.eye(3, dtype=np.float32)),
                   ("t", np.zeros(3, np.float32)), ("O", np.zeros(3, np.float32)), ("logsf", np.array([np.log(1.2)], np.float32)),
                   ("pstate", pstate), ("pobs", pobs), ("ppos", pos), ("pnormal", normal.astype(np.float32)),
                   ("pdist", np.stack([mind, maxd], 1)), ("pdesc", pdesc), ("th", np.array([th], np.float32))):
        _put(f, tag + k, val)
    if rig:
        for k, val in (("NLeft", np.array([nleft], np.int32)), ("bRight", np.array([int(rig == "right")], np.int32)),
                       ("cam2", np.array([512, 512, 400, 256], np.float32)), ("RR", np.eye(3, dtype=np.float32)),
                       ("tR", np.array([-0.125, 0, 0], np.float32)), ("OR", np.array([0.125, 0, 0], np.float32))):
            _put(f, tag + k, val)

    def check(res):
        ok = (pstate == 1) & (z > 0) & (u >= 0) & (u < 768) & (v >= 0) & (v < 512) & ~wrong & ~flip
        keep = np.nonzero(ok)[0]
        lv = level[keep].astype(np.int32)
        pr = _frame_problem(fr, np.zeros(n, np.int32))
        if rig:
            pr.update(Nleft=nleft, qflags=np.full(len(keep), int(rig == "right"), np.uint8))
        pr.update(mode=1, nnratio=0.6, th_high=50, check_orientation=0, chi2_gate=1,
                  inv_level_sigma2=(np.float32(1.0) / (SF * SF)).astype(np.float32), qdesc=pdesc[keep],
                  qx=u[keep].astype(np.float32), qy=v[keep].astype(np.float32), qr=(np.float32(th) * SF[lv]).astype(np.float32),
                  qmin_level=lv - 1, qmax_level=lv, qxr=(u[keep] - 40.0 / z[keep]).astype(np.float32),
                  qangle=np.zeros(len(keep), np.float32), qblocks=np.zeros(len(keep), np.uint8), taken=np.zeros(n, np.uint8))
        _, qm, _ = oracle.search_projection(pr)
        # the sequential object logic of :1813-1838 on this side
        point = np.where(fstate > 0, 100000 + np.arange(n), -1)              # id of the point each feature holds
        obs_of = {100000 + i: int(fobs[i]) for i in range(n)}
        obs_of.update({int(q): int(pobs[q]) for q in range(m)})
        bad_of = {100000 + i: bool(fstate[i] == 2) for i in range(n)}
        obs_idx, repl, kf_repl = np.full(m, -1), np.full(m, -1), np.full(n, -1)
        root = np.where(pdup >= 0, pdup, np.arange(m))                      # the object behind a list entry
        in_kf = set()
        nfused = 0
        for k, q in enumerate(keep):
            idx = qm[k]
            if idx < 0:
                continue
            o = int(root[q])
            if bad_of.get(o, False) or o in in_kf:                           # :1690-1692, evaluated per iteration
                continue
            pid = point[idx]
            if pid >= 0:
                if not bad_of.get(pid, False):
                    if obs_of[pid] > obs_of[o]:
                        repl[o] = pid
                        bad_of[o] = True                                     # MapPoint::Replace kills the replaced point
                    elif pid >= 100000:
                        kf_repl[pid - 100000] = o
                        bad_of[pid] = True
                    else:
                        repl[pid] = o
                        bad_of[pid] = True
            else:
                obs_idx[o] = idx
                point[idx] = o
                in_kf.add(o)
            nfused += 1
        obs_idx, repl = obs_idx[root], repl[root]                            # list entries of one object report the same
        assert res[tag + "n"][0] == nfused and nfused > 50, (tag, nfused)
        assert np.array_equal(res[tag + "obsIdx"], obs_idx), tag
        assert np.array_equal(res[tag + "replacedBy"], repl), tag
        assert np.array_equal(res[tag + "kfPoint"], point), tag
        assert np.array_equal(res[tag + "kfReplacedBy"], kf_repl), tag
    return check


CAM = np.array([512, 512, 384, 256], np.float32)


def _projected_points(rng, fr, m, tcw, level_of, far_frac=0.04, zchoices=(2.0, 4.0, 8.0), noise=1.0):
    """m world points that project (camera pose R = I, t = tcw, CAM) onto dyadic pixels near random features of `fr`.
    Returns (tgt, u, v, z, world pos float32, camera-frame distance, predicted level)."""
    n = len(fr["kx"])
    tgt = rng.integers(0, n, m)
    tgt[: m // 3] = rng.integers(0, max(n // 4, 1), m // 3)
    tgt[m // 3: m // 2] = tgt[: m // 2 - m // 3]
    u = np.round((fr["kx"][tgt] + rng.normal(0, noise, m)) * 8) / 8
    v = np.round((fr["ky"][tgt] + rng.normal(0, noise, m)) * 8) / 8
    u[rng.random(m) < far_frac] = 900.0                                   # outside the image
    z = rng.choice(list(zchoices), m)
    z[rng.random(m) < far_frac] = -4.0                                    # behind the camera
    xc = np.stack([(u - 384) * z / 512, (v - 256) * z / 512, z], 1)
    pos = (xc - np.asarray(tcw, np.float64)).astype(np.float32)
    assert np.array_equal(pos.astype(np.float64) + np.asarray(tcw, np.float64), xc)
    dist = np.linalg.norm(xc, axis=1)
    level = np.clip(fr["oct"][tgt] + level_of(rng, m), 0, 7)
    return tgt, u, v, z, pos, xc, dist, level


def _dist_range(dist, level, wrong):
    maxd = (dist * 1.2 ** (level - 0.5)).astype(np.float32)               # PredictScale = ceil(level - 0.5) = level
    maxd[wrong] = (dist[wrong] * 0.5).astype(np.float32)                   # outside the scale-invariance range
    return np.stack([(maxd / np.float32(1.2 ** 7)).astype(np.float32), maxd], 1)


def _kf_common(f, tag, fr, sfx=""):
    for k in ("kx", "ky", "oct", "desc", "ang"):
        _put(f, tag + k + sfx, fr[k])
    if not sfx or sfx == "1":
        _put(f, tag + "grid", np.array([0, 0, 768, 512, np.float32(64) / np.float32(768), np.float32(48) / np.float32(512)], np.float32))
        _put(f, tag + "sf", SF)
        _put(f, tag + "logsf", np.array([np.log(1.2)], np.float32))
        _put(f, tag + "cam", CAM)


def _reloc(f, tag, oracle, seed, n, nk, th, orbdist, ori):
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, False)
    fstate = rng.choice([0, 0, 0, 0, 0, 1, 3], n).astype(np.int32)         # any point at a feature hides it (:2484)
    _put_frame(f, tag, fr, fstate)
    tc = np.array([0.25, -0.5, 0.125], np.float32)
    tgt, u, v, z, pos, xc, dist, level = _projected_points(rng, fr, nk, tc, lambda r, m: r.integers(-1, 2, m), noise=2.0)
    wrong = rng.random(nk) < 0.05
    kstate = rng.choice([0, 1, 1, 1, 1, 2, 3], nk).astype(np.int32)
    kang = rng.uniform(0, 360, nk).astype(np.float32)
    pdesc = _noisy(fr["desc"][tgt], rng)
    for k, val in (("cam", CAM), ("Rc", np.eye(3, dtype=np.float32)), ("tc", tc), ("logsf", np.array([np.log(1.2)], np.float32)),
                   ("kstate", kstate), ("kang", kang), ("ppos", pos), ("pdist", _dist_range(dist, level, wrong)), ("pdesc", pdesc),
                   ("th", np.array([th], np.float32)), ("orbdist", np.array([orbdist], np.int32)), ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, val)

    def check(res):
        # (this overload has no depth test: a point behind the camera projects to the same pixel, :2444-2449)
        ok = (kstate == 1) & (u >= 0) & (u <= 768) & (v >= 0) & (v <= 512) & ~wrong
        keep = np.nonzero(ok)[0]
        lv = level[keep].astype(np.int32)
        pr = _frame_problem(fr, fstate)
        pr["taken"] = (fstate > 0).astype(np.uint8)
        pr.update(mode=1, nnratio=0.9, th_high=orbdist, check_orientation=int(ori), qdesc=pdesc[keep], qx=u[keep].astype(np.float32),
                  qy=v[keep].astype(np.float32), qr=(np.float32(th) * SF[lv]).astype(np.float32), qmin_level=lv - 1,
                  qmax_level=lv + 1, qxr=np.zeros(len(keep), np.float32), qangle=kang[keep], qblocks=np.ones(len(keep), np.uint8))
        nm, qm, fm = oracle.search_projection(pr)
        assert res[tag + "n"][0] == nm and nm > 30, (tag, nm)
        assert np.array_equal(res[tag + "points"], _expect_points(fstate, fm, qm, keep)), tag
    return check


def _sim3_objects(rng, fr, m, tcw):
    tgt, u, v, z, pos, xc, dist, level = _projected_points(rng, fr, m, tcw, lambda r, mm: r.integers(0, 2, mm))
    wrong = rng.random(m) < 0.05
    flip = rng.random(m) < 0.05
    normal = xc / np.maximum(dist, 1e-9)[:, None]
    normal[flip] *= -1
    visible = (z > 0) & (u >= 0) & (u < 768) & (v >= 0) & (v < 512) & ~wrong & ~flip
    return tgt, u, v, z, pos, dist, level, wrong, normal.astype(np.float32), visible


def _scw(s, tcw):
    S = np.zeros((4, 4), np.float32)
    S[:3, :3] = np.float32(s) * np.eye(3, dtype=np.float32)
    S[:3, 3] = np.float32(s) * np.asarray(tcw, np.float32)
    S[3, 3] = 1
    return S


def _sim3_projection(f, tag, oracle, seed, n, m, th, ratio, twin):
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, False)
    _kf_common(f, tag, fr)
    tcw = np.array([0.5, 0.25, -0.125])
    tgt, u, v, z, pos, dist, level, wrong, normal, visible = _sim3_objects(rng, fr, m, tcw)
    pstate = rng.choice([1, 1, 1, 1, 1, 2], m).astype(np.int32)
    pdesc = _noisy(fr["desc"][tgt], rng, 0.0, 0.2)
    # vpMatched on entry: a tenth of the features hold one of the candidate points already (those points are then
    # skipped, :488-495), another tenth a foreign point
    mpre = np.full(n, -1, np.int32)
    pick = rng.permutation(n)[: n // 5]
    cand = rng.permutation(m)[: n // 10]
    mpre[pick[: n // 10]] = cand
    mpre[pick[n // 10:]] = -2
    for k, val in (("Scw", _scw(2.0, tcw)), ("pstate", pstate), ("ppos", pos), ("pnormal", normal), ("pdist", _dist_range(dist, level, wrong)),
                   ("pdesc", pdesc), ("mpre", mpre), ("th", np.array([th], np.int32)), ("ratio", np.array([ratio], np.float32)),
                   ("twin", np.array([int(twin)], np.int32))):
        _put(f, tag + k, val)

    def check(res):
        found = set(int(c) for c in cand)
        keep = np.array([q for q in range(m) if pstate[q] == 1 and q not in found and visible[q]])
        lv = level[keep].astype(np.int32)
        pr = _frame_problem(fr, np.zeros(n, np.int32))
        pr.pop("uright")
        pr.update(taken=(mpre != -1).astype(np.uint8), mode=1, nnratio=0.6, th_high=int(np.floor(50 * ratio)), check_orientation=0,
                  qdesc=pdesc[keep], qx=u[keep].astype(np.float32), qy=v[keep].astype(np.float32),
                  qr=(np.float32(th) * SF[lv]).astype(np.float32), qmin_level=lv - 1, qmax_level=lv,
                  qangle=np.zeros(len(keep), np.float32), qblocks=np.ones(len(keep), np.uint8))
        nm, qm, fm = oracle.search_projection(pr)
        assert res[tag + "n"][0] == nm and nm > 30, (tag, nm)
        exp = np.where(mpre >= 0, mpre, np.where(mpre == -2, 500000 + np.arange(n), -1)).astype(np.int32)
        w = fm >= 0
        exp[w] = keep[fm[w]]
        assert np.array_equal(res[tag + "matched"], exp), tag
        expkf = np.full(n, -1, np.int32)
        if twin:
            expkf[w] = keep[fm[w]] % 5
        assert np.array_equal(res[tag + "matchedKF"], expkf), tag
    return check


def _fuse_sim3(f, tag, oracle, seed, n, m, th):
    rng = np.random.default_rng(seed)
    fr = _frame_arrays(rng, n, False)
    _kf_common(f, tag, fr)
    fstate = rng.choice([0, 0, 1, 1, 2], n).astype(np.int32)
    _put(f, tag + "fstate", fstate)
    tcw = np.array([-0.25, 0.5, 0.0625])
    tgt, u, v, z, pos, dist, level, wrong, normal, visible = _sim3_objects(rng, fr, m, tcw)
    pstate = rng.choice([1, 1, 1, 1, 1, 1, 2, 3], m).astype(np.int32)
    good = np.nonzero(fstate == 1)[0]
    own = good[rng.integers(0, len(good), m)].astype(np.int32)             # for pstate 3: one of the keyframe's points
    pdesc = _noisy(fr["desc"][tgt], rng, 0.0, 0.2)
    for k, val in (("Scw", _scw(2.0, tcw)), ("pstate", pstate), ("own", own), ("ppos", pos), ("pnormal", normal),
                   ("pdist", _dist_range(dist, level, wrong)), ("pdesc", pdesc), ("th", np.array([th], np.float32))):
        _put(f, tag + k, val)

    def check(res):
        keep = np.nonzero((pstate == 1) & visible)[0]
        lv = level[keep].astype(np.int32)
        pr = _frame_problem(fr, np.zeros(n, np.int32))
        pr.pop("uright")
        pr.update(mode=1, nnratio=0.6, th_high=50, check_orientation=0, qdesc=pdesc[keep], qx=u[keep].astype(np.float32),
                  qy=v[keep].astype(np.float32), qr=(np.float32(th) * SF[lv]).astype(np.float32), qmin_level=lv - 1, qmax_level=lv,
                  qangle=np.zeros(len(keep), np.float32), qblocks=np.zeros(len(keep), np.uint8), taken=np.zeros(n, np.uint8))
        _, qm, _ = oracle.search_projection(pr)
        point = np.where(fstate > 0, 100000 + np.arange(n), -1)
        bad = {100000 + i: bool(fstate[i] == 2) for i in range(n)}
        repl, obs_idx = np.full(m, -1), np.full(m, -1)
        nfused = 0
        for k, q in enumerate(keep):                                        # :1941-1958
            idx = qm[k]
            if idx < 0:
                continue
            if point[idx] >= 0:
                if not bad.get(int(point[idx]), False):
                    repl[q] = point[idx]
            else:
                obs_idx[q] = idx
                point[idx] = q
            nfused += 1
        assert res[tag + "n"][0] == nfused and nfused > 50, (tag, nfused)
        assert np.array_equal(res[tag + "replace"], repl) and np.array_equal(res[tag + "obsIdx"], obs_idx), tag
        assert np.array_equal(res[tag + "kfPoint"], point), tag
    return check


def _search_by_sim3(f, tag, oracle, seed, n, th):
    """Two keyframes that see the same n world points; keyframe 2 is keyframe 1 shifted by 1/16 along x (so that the
    projections into both are exact), the Sim3 handed over is the true relative pose (s12 = 1, R12 = I)."""
    rng = np.random.default_rng(seed)
    fr1 = _frame_arrays(rng, n, False)
    t1 = np.array([0.25, -0.5, 0.0])
    t2 = t1 + np.array([0.0625, 0.0, 0.0])
    u1 = np.round(fr1["kx"] * 8) / 8
    v1 = np.round(fr1["ky"] * 8) / 8
    z = rng.choice([2.0, 4.0, 8.0], n)
    xc1 = np.stack([(u1 - 384) * z / 512, (v1 - 256) * z / 512, z], 1)
    world = xc1 - t1
    xc2 = world + t2
    u2, v2 = 512 * xc2[:, 0] / z + 384, 512 * xc2[:, 1] / z + 256
    assert np.array_equal(np.round(u2 * 8) / 8, u2)
    fr1["kx"], fr1["ky"] = (u1 + rng.normal(0, 1.0, n)).astype(np.float32), (v1 + rng.normal(0, 1.0, n)).astype(np.float32)
    perm = rng.permutation(n)                                               # keyframe 2 lists the points in another order
    fr2 = dict(kx=(u2 + rng.normal(0, 1.0, n)).astype(np.float32)[perm], ky=(v2 + rng.normal(0, 1.0, n)).astype(np.float32)[perm],
               oct=np.clip(fr1["oct"] + rng.integers(-1, 1, n), 0, 7).astype(np.int32)[perm],
               desc=_noisy(fr1["desc"], rng, 0.0, 0.15)[perm], ang=fr1["ang"][perm])
    _kf_common(f, tag, fr1, "1")
    _kf_common(f, tag, fr2, "2")
    pos = world.astype(np.float32)
    assert np.array_equal(pos.astype(np.float64), world)
    level = np.clip(fr1["oct"] + rng.integers(0, 2, n), 0, 7)
    dmean = 0.5 * (np.linalg.norm(xc1, axis=1) + np.linalg.norm(xc2, axis=1))
    pdist = _dist_range(dmean, level, rng.random(n) < 0.04)
    k1 = rng.choice([0, 1, 1, 1, 1, 2], n).astype(np.int32)
    k2 = rng.choice([0, 1, 1, 1, 1, 2], n).astype(np.int32)
    d1p, d2p = _noisy(fr1["desc"], rng, 0.0, 0.1), _noisy(fr1["desc"], rng, 0.0, 0.1)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)                                               # world point j is feature inv[j] of keyframe 2
    pre12 = np.full(n, -1, np.int32)
    cand = np.nonzero((k1 == 1) & (k2 == 1))[0][:: 9]
    pre12[cand] = inv[cand]                                                # some pairs are matched already
    for k, val in (("kstate1", k1), ("ppos1", pos), ("pdist1", pdist), ("pdesc1", d1p),
                   ("kstate2", k2[perm]), ("ppos2", pos[perm]), ("pdist2", pdist[perm]), ("pdesc2", d2p[perm]),
                   ("R1", np.eye(3, dtype=np.float32)), ("t1", t1.astype(np.float32)), ("R2", np.eye(3, dtype=np.float32)),
                   ("t2", t2.astype(np.float32)), ("s12", np.array([1.0], np.float32)), ("R12", np.eye(3, dtype=np.float32)),
                   ("t12", (t1 - t2).astype(np.float32)), ("pre12", pre12), ("th", np.array([th], np.float32))):
        _put(f, tag + k, val)

    def check(res):
        def direction(fr_into, pts_state, done, uu, vv, xc, lv_src, desc, order):
            dist = np.linalg.norm(xc, axis=1)
            lvl = np.array([int(np.clip(np.ceil(np.log(pdist[j, 1] / dist[j]) / np.log(1.2)), 0, 7)) for j in range(n)])
            ok = [j for j in order if pts_state[j] == 1 and not done[j] and 0 <= uu[j] < 768 and 0 <= vv[j] < 512
                  and pdist[j, 0] * np.float32(0.8) <= np.float32(dist[j]) <= pdist[j, 1] * np.float32(1.2)]
            ok = np.array(ok)
            lv = lvl[ok].astype(np.int32)
            pr = _frame_problem(fr_into, np.zeros(n, np.int32))
            pr.pop("uright")
            pr.update(mode=1, nnratio=0.6, th_high=100, check_orientation=0, qdesc=desc[ok], qx=uu[ok].astype(np.float32),
                      qy=vv[ok].astype(np.float32), qr=(np.float32(th) * SF[lv]).astype(np.float32), qmin_level=lv - 1,
                      qmax_level=lv, qangle=np.zeros(len(ok), np.float32), qblocks=np.zeros(len(ok), np.uint8),
                      taken=np.zeros(n, np.uint8))
            _, qm, _ = oracle.search_projection(pr)
            return ok, qm
        done1 = pre12 >= 0                                                   # by keyframe-1 feature (= world point) index
        done2w = np.zeros(n, bool)
        done2w[np.nonzero(done1)[0]] = True                                 # the same world points, seen from keyframe 2
        ok1, qm1 = direction(fr2, k1, done1, u2, v2, xc2, level, d1p, range(n))          # points of 1 into keyframe 2
        ok2, qm2 = direction(fr1, k2, done2w, u1, v1, xc1, level, d2p, list(perm))       # points of 2 (its order) into 1
        m1 = np.full(n, -1)
        m1[ok1] = qm1                                                       # keyframe-1 feature -> keyframe-2 feature
        m2 = np.full(n, -1)
        m2[inv[ok2]] = qm2                                                  # keyframe-2 feature -> keyframe-1 feature
        exp = pre12.copy()
        nfound = 0
        for i1 in range(n):
            if m1[i1] >= 0 and m2[m1[i1]] == i1:
                exp[i1] = m1[i1]
                nfound += 1
        assert res[tag + "n"][0] == nfound and nfound > 100, (tag, nfound)
        assert np.array_equal(res[tag + "matches12"], exp), tag
    return check


def _initialization(f, tag, oracle, seed, n1, n2, window, ratio, ori):
    import matcher_inputs as MI
    pr = MI.initialization_problem(seed, n1=n1, n2=n2, window=window, nnratio=ratio, check_orientation=ori)
    for k, val in (("oct1", pr["octave1"]), ("ang1", pr["angle1"]), ("desc1", pr["desc1"]), ("prev", pr["prev_xy"]),
                   ("oct2", pr["octave2"]), ("ang2", pr["angle2"]), ("desc2", pr["desc2"]), ("kx2", pr["kx2"]), ("ky2", pr["ky2"]),
                   ("grid", np.array([pr["minX"], pr["minY"], pr["gridWInv"], pr["gridHInv"]], np.float32)),
                   ("window", np.array([pr["window_size"]], np.int32)), ("ratio", np.array([ratio], np.float32)),
                   ("ori", np.array([int(ori)], np.int32))):
        _put(f, tag + k, np.ascontiguousarray(val))

    def check(res):
        nm, m12 = oracle.search_initialization(pr)
        assert res[tag + "n"][0] == nm and nm > 50, (tag, nm)
        assert np.array_equal(res[tag + "matches12"], m12), tag
        prev = np.ascontiguousarray(pr["prev_xy"], np.float32).reshape(-1, 2).copy()
        w = m12 >= 0
        prev[w, 0] = pr["kx2"][m12[w]]
        prev[w, 1] = pr["ky2"][m12[w]]
        assert np.array_equal(res[tag + "prev"].reshape(-1, 2), prev), tag
    return check


def test_cpp_matcher_adapter_matches_oracle(tmp_path, oracle):
    exe = str(tmp_path / "test_matcher_adapter")
    libdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "adapters"),
                           os.path.join(ROOT, "adapters", "test_matcher_adapter.cpp"), "-o", exe, "-L" + libdir, "-lorbfe",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    scen, resf = str(tmp_path / "scenario.bin"), str(tmp_path / "result.bin")
    checks = []
    with open(scen, "wb") as f:
        checks.append(_bow_kf_f(f, "bow0.", oracle, 11, 1100, 1200, -1, 0.7, True))
        checks.append(_bow_kf_f(f, "bow1.", oracle, 12, 900, 1000, 600, 0.8, True))   # two-camera frame: left / right tracks
        checks.append(_bow_kf_f(f, "bow2.", oracle, 13, 300, 500, -1, 0.9, False))
        checks.append(_bow_kf_kf(f, "kk0.", oracle, 21, 1000, 1100, 0.8, True))
        checks.append(_bow_kf_kf(f, "kk1.", oracle, 22, 400, 350, 0.75, False))
        checks.append(_tri(f, "tri0.", oracle, 31, 1000, 1100, False, False, True))
        checks.append(_tri(f, "tri1.", oracle, 32, 800, 700, True, False, True))
        checks.append(_tri(f, "tri2.", oracle, 33, 600, 600, False, True, False))
        checks.append(_tri_kb8(f, "tk0.", oracle, 35, 1000, 900, False, False))               # monocular fisheye pair
        checks.append(_tri_kb8(f, "tk1.", oracle, 36, 1100, 1000, True, False))               # two-camera rigs
        checks.append(_tri_kb8(f, "tk2.", oracle, 37, 800, 900, True, True))
        checks.append(_proj_local(f, "p0_0.", oracle, 41, 1500, 1200, 1.0, True, False))
        checks.append(_proj_local(f, "p0_1.", oracle, 42, 1200, 900, 3.0, False, True))
        checks.append(_proj_last(f, "p1_0.", oracle, 51, 1500, 1000, 7.0, True, 0.0, False, True))     # neither direction
        checks.append(_proj_last(f, "p1_1.", oracle, 52, 1200, 900, 15.0, False, 1.0, False, True))    # forward
        checks.append(_proj_last(f, "p1_2.", oracle, 53, 1200, 900, 7.0, True, -1.0, False, False))    # backward
        checks.append(_proj_last(f, "p1_3.", oracle, 54, 1000, 800, 15.0, False, 1.0, True, True))     # monocular
        checks.append(_fuse(f, "fu0.", oracle, 61, 1500, 1200, 3.0))
        checks.append(_fuse(f, "fu1.", oracle, 62, 900, 700, 4.0))
        checks.append(_fuse(f, "fu2.", oracle, 63, 1500, 1000, 3.0, rig="left"))                       # two-camera rigs
        checks.append(_fuse(f, "fu3.", oracle, 64, 1500, 1000, 3.0, rig="right"))
        checks.append(_fuse(f, "fu4.", oracle, 65, 1500, 1000, 3.0, rig="left", dup=True))             # one point, two list entries
        checks.append(_fuse(f, "fu5.", oracle, 66, 1200, 900, 3.0, dup=True))
        checks.append(_proj_local_rig(f, "p0_2.", oracle, 43, 1500, 1200, 1.0))
        checks.append(_proj_local_rig(f, "p0_3.", oracle, 44, 1200, 1000, 3.0))
        checks.append(_proj_last_rig(f, "p1_4.", oracle, 55, 1500, 1000, 7.0, 0.0, True))
        checks.append(_proj_last_rig(f, "p1_5.", oracle, 56, 1400, 900, 15.0, 1.0, False))
        checks.append(_reloc(f, "p2_0.", oracle, 71, 1500, 1100, 10.0, 100, True))
        checks.append(_reloc(f, "p2_1.", oracle, 72, 1000, 900, 3.0, 64, False))
        checks.append(_sim3_projection(f, "s3_0.", oracle, 81, 1500, 1200, 8, 0.9, True))
        checks.append(_sim3_projection(f, "s3_1.", oracle, 82, 1000, 800, 10, 1.0, False))
        checks.append(_fuse_sim3(f, "fs0.", oracle, 91, 1500, 1200, 4.0))
        checks.append(_search_by_sim3(f, "ss0.", oracle, 101, 1200, 7.5))
        checks.append(_initialization(f, "in0.", oracle, 111, 1500, 1400, 100, 0.9, True))
        checks.append(_initialization(f, "in1.", oracle, 112, 900, 1000, 30, 0.7, False))
    subprocess.check_call([exe, scen, resf])
    res = _read(resf)
    for c in checks:
        c(res)
