"""Seeded matcher inputs shared by the CPU (oracle-only) and GPU matcher tests."""
import numpy as np

from orb_slam3_detailed_comments_kor_amd import synth


def descriptor_sets(n1, n2, seed, flip=18):
    """Two descriptor sets with correspondences: set 2 = permuted noisy copy of part of set 1 + randoms."""
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, size=(n1, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, size=(n2, 32), dtype=np.uint8)
    k = min(n1, n2) * 2 // 3
    src = rng.permutation(n1)[:k]
    dst = rng.permutation(n2)[:k]
    bits = np.unpackbits(d1[src], axis=1)
    for r in range(k):
        nf = rng.integers(0, flip * 2)
        bits[r, rng.permutation(256)[:nf]] ^= 1
    d2[dst] = np.packbits(bits, axis=1)
    # exact duplicates to force distance ties
    if n2 > 8:
        d2[dst[0] if k else 0] = d2[(dst[1] if k > 1 else 1)]
    ang1 = rng.uniform(0, 360, n1).astype(np.float32)
    ang2 = ang1[rng.integers(0, n1, n2)] + rng.normal(0, 8, n2).astype(np.float32)
    ang2[dst] = ang1[src] + rng.normal(0, 5, k).astype(np.float32) + np.float32(20.0)
    ang2 = np.mod(ang2, 360).astype(np.float32)
    return d1, d2, ang1, ang2


def feature_vectors(d1, d2, seed, branching=6, depth=2):
    # same centroids for both sets (same vocabulary)
    return synth.make_feature_vectors(d1, seed, branching, depth), synth.make_feature_vectors(d2, seed, branching, depth)


def tri_inputs(n1, n2, seed):
    rng = np.random.default_rng(seed)
    d1, d2, a1, a2 = descriptor_sets(n1, n2, seed, flip=12)
    fv1, fv2 = feature_vectors(d1, d2, seed + 1, 5, 2)
    kp1 = np.stack([rng.uniform(20, 730, n1), rng.uniform(20, 460, n1)], 1).astype(np.float32)
    kp2 = np.stack([rng.uniform(20, 730, n2), rng.uniform(20, 460, n2)], 1).astype(np.float32)
    # make many true correspondences satisfy a pure horizontal-translation epipolar geometry
    kp2[:, 1] = np.where(rng.uniform(size=n2) < 0.7, kp1[rng.integers(0, n1, n2), 1] + rng.normal(0, 0.7, n2), kp2[:, 1])
    oct1 = rng.integers(0, 8, n1).astype(np.int32)
    oct2 = rng.integers(0, 8, n2).astype(np.int32)
    u1 = np.where(rng.uniform(size=n1) < 0.3, rng.uniform(0, 700, n1), -1).astype(np.float32)
    u2 = np.where(rng.uniform(size=n2) < 0.3, rng.uniform(0, 700, n2), -1).astype(np.float32)
    has1 = (rng.uniform(size=n1) < 0.4).astype(np.uint8)
    has2 = (rng.uniform(size=n2) < 0.4).astype(np.uint8)
    # F for x-translation with K = [[458,0,367],[0,457,248],[0,0,1]]: l = F^T... rows chosen so a*x2+b*y2+c ~ y2-y1
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    t = np.array([0.11, 0.0, 0.0])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ np.eye(3) @ np.linalg.inv(K)
    F12 = F.astype(np.float32)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    sig = (sf * sf).astype(np.float32)
    ep = (900.0, 250.0)
    return dict(d1=d1, d2=d2, a1=a1, a2=a2, fv1=fv1, fv2=fv2, kp1=kp1, kp2=kp2, oct1=oct1, oct2=oct2, u1=u1, u2=u2,
                has1=has1, has2=has2, F12=F12, sf=sf, sig=sig, ep=ep)


def initialization_problem(seed, n1=1500, n2=1400, window=100, nnratio=0.9, check_orientation=True, crowd=True, w=752, h=480):
    """SearchForInitialization inputs (fields of orbfe_init_args): F2 features are noisy copies of F1 features that
    moved a little; with `crowd`, several F1 keypoints resemble the same F2 feature to different degrees, so the
    stealing rule (vMatchedDistance / vnMatches21, src/ORBmatcher.cc:744, :765-772) decides who keeps it."""
    rng = np.random.default_rng(seed)
    kx1, ky1 = rng.uniform(5, w - 5, n1).astype(np.float32), rng.uniform(5, h - 5, n1).astype(np.float32)
    octave1 = np.where(rng.random(n1) < 0.6, 0, rng.integers(1, 8, n1)).astype(np.int32)
    desc1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    angle1 = rng.uniform(0, 360, n1).astype(np.float32)
    src = rng.integers(0, n1, n2)
    if crowd:  # groups of F1 keypoints share one descriptor family and sit close together
        g = n1 // 6
        base = rng.integers(0, n1, g)
        members = rng.integers(0, n1, g)
        bits = np.unpackbits(desc1[base], axis=1)
        flips = rng.random(bits.shape) < rng.uniform(0.0, 0.08, (g, 1))
        desc1[members] = np.packbits(bits ^ flips, axis=1)
        kx1[members] = (kx1[base] + rng.normal(0, 8, g)).astype(np.float32)
        ky1[members] = (ky1[base] + rng.normal(0, 8, g)).astype(np.float32)
        octave1[members] = 0
        octave1[base] = 0
    bits = np.unpackbits(desc1[src], axis=1)
    flips = rng.random(bits.shape) < rng.uniform(0.0, 0.12, (n2, 1))
    desc2 = np.packbits(bits ^ flips, axis=1)
    kx2 = (kx1[src] + rng.normal(0, 12, n2)).astype(np.float32)
    ky2 = (ky1[src] + rng.normal(0, 12, n2)).astype(np.float32)
    octave2 = np.where(rng.random(n2) < 0.7, 0, rng.integers(1, 8, n2)).astype(np.int32)
    angle2 = np.mod(angle1[src] + rng.normal(0, 6, n2) + 15.0, 360).astype(np.float32)
    flip = rng.random(n2) < 0.15
    angle2[flip] = rng.uniform(0, 360, flip.sum()).astype(np.float32)
    prev = np.stack([kx1, ky1], 1).astype(np.float32)  # first call: vbPrevMatched = F1 keypoint positions
    return dict(desc1=desc1, octave1=octave1, angle1=angle1, prev_xy=prev, desc2=desc2, kx2=kx2, ky2=ky2, octave2=octave2,
                angle2=angle2, minX=np.float32(0.0), minY=np.float32(0.0), gridWInv=np.float32(64) / np.float32(w),
                gridHInv=np.float32(48) / np.float32(h), window_size=window, nnratio=nnratio,
                check_orientation=int(check_orientation))


KB8_TUMVI = np.array([190.978477, 190.973307, 254.931706, 256.897442, 0.003482389, 0.000715034, -0.002053236,
                      0.000202937], np.float32)  # Examples/Stereo-Inertial/TUM_512.yaml:9-30


def kb8_project64(P, X):
    """KannalaBrandt8::project in float64 (data generation only)."""
    P = np.asarray(P, np.float64)
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    th = np.arctan2(np.sqrt(x * x + y * y), z)
    psi = np.arctan2(y, x)
    r = th + P[4] * th ** 3 + P[5] * th ** 5 + P[6] * th ** 7 + P[7] * th ** 9
    return np.stack([P[0] * r * np.cos(psi) + P[2], P[1] * r * np.sin(psi) + P[3]], 1)


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def kb8_pairs(seed, n=4000):
    """Keypoint pairs for TriangulateMatches_: true correspondences of 3-D points seen by two fisheye cameras
    (a few pixels of noise, so the reprojection tests bite), random pairs, and low-parallax pairs."""
    rng = np.random.default_rng(seed)
    P1 = KB8_TUMVI.copy()
    P2 = (KB8_TUMVI * np.float32(1.0) + np.array([0.7, -0.4, 1.3, -2.1, 0, 0, 0, 0], np.float32)).astype(np.float32)
    R12 = _rot(0.03, -0.05, 0.02)
    t12 = np.array([0.25, -0.03, 0.06])
    X1 = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(0.6, 12, n)], 1)
    far = rng.random(n) < 0.15
    X1[far] *= 60.0                                   # distant points: parallax below the 0.9998 gate
    X2 = (R12.T @ (X1 - t12).T).T                     # x1 = R12 x2 + t12
    kp1 = kb8_project64(P1, X1) + rng.normal(0, 0.4, (n, 2))
    noise = np.where(rng.random(n)[:, None] < 0.3, rng.normal(0, 3.0, (n, 2)), rng.normal(0, 0.5, (n, 2)))
    kp2 = kb8_project64(P2, X2) + noise
    rnd = rng.random(n) < 0.2
    kp2[rnd] = np.stack([rng.uniform(40, 470, rnd.sum()), rng.uniform(40, 470, rnd.sum())], 1)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    sig = (sf * sf).astype(np.float32)
    o1, o2 = rng.integers(0, 8, n), rng.integers(0, 8, n)
    return dict(P1=P1, P2=P2, R12=R12.astype(np.float32), t12=t12.astype(np.float32), kp1=kp1.astype(np.float32),
                kp2=kp2.astype(np.float32), sigma1=sig[o1], sigma2=sig[o2], X1=X1)


def stereo_fisheye_inputs(seed, nL=900, nR=850):
    """Lapping-area slices of a fisheye stereo frame: right descriptors are noisy copies of left ones, keypoints are
    projections of common 3-D points through two KB8 cameras related by (Rlr, tlr); some pairs are mismatched."""
    rng = np.random.default_rng(seed)
    P1 = KB8_TUMVI.copy()
    P2 = (KB8_TUMVI + np.array([0.6, -0.2, -1.4, 0.9, 0, 0, 0, 0], np.float32)).astype(np.float32)
    Rlr, tlr = _rot(0.01, -0.02, 0.005), np.array([0.101, 0.0006, -0.0012])   # x_left = Rlr x_right + tlr
    descL = rng.integers(0, 256, (nL, 32), dtype=np.uint8)
    X = np.stack([rng.uniform(-1.5, 1.5, nL), rng.uniform(-1.2, 1.2, nL), rng.uniform(0.4, 6, nL)], 1)
    far = rng.random(nL) < 0.1
    X[far] *= 200.0
    kpL = kb8_project64(P1, X) + rng.normal(0, 0.3, (nL, 2))
    src = rng.permutation(nL)[:nR] if nR <= nL else rng.integers(0, nL, nR)
    bits = np.unpackbits(descL[src], axis=1)
    flips = rng.random(bits.shape) < rng.uniform(0.0, 0.1, (nR, 1))
    descR = np.packbits(bits ^ flips, axis=1)
    rnd = rng.random(nR) < 0.25
    descR[rnd] = rng.integers(0, 256, (rnd.sum(), 32), dtype=np.uint8)
    XR = (Rlr.T @ (X[src] - tlr).T).T
    kpR = kb8_project64(P2, XR) + np.where(rng.random(nR)[:, None] < 0.2, rng.normal(0, 4.0, (nR, 2)), rng.normal(0, 0.4, (nR, 2)))
    octL = rng.integers(0, 8, nL).astype(np.int32)
    octR = rng.integers(0, 8, nR).astype(np.int32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    return dict(descL=descL, kpL=kpL.astype(np.float32), octL=octL, descR=descR, kpR=kpR.astype(np.float32), octR=octR,
                P1=P1, P2=P2, Rlr=Rlr.astype(np.float32), tlr=tlr.astype(np.float32), sig=(sf * sf).astype(np.float32))


def tri_kb8_inputs(n1, n2, seed, rig=False):
    """SearchForTriangulation_ inputs for fisheye keyframes: descriptors / FeatureVectors as tri_inputs, keypoints
    from a common 3-D scene so that true correspondences pass the triangulation gate."""
    rng = np.random.default_rng(seed)
    d1, d2, a1, a2 = descriptor_sets(n1, n2, seed, flip=12)
    # recover which rows of set 2 are noisy copies of rows of set 1: nearest by Hamming distance
    lut = np.array([bin(v).count("1") for v in range(256)], np.uint8)
    D = lut[d1[:, None, :] ^ d2[None, :, :]].sum(2, dtype=np.int32)
    fv1, fv2 = feature_vectors(d1, d2, seed + 1, 5, 2)
    P1L = KB8_TUMVI.copy()
    P1R = (KB8_TUMVI + np.array([0.5, 0.3, -1.1, 0.8, 0, 0, 0, 0], np.float32)).astype(np.float32)
    P2L, P2R = P1L.copy(), P1R.copy()
    Nleft1 = n1 * 3 // 5 if rig else -1
    Nleft2 = n2 * 3 // 5 if rig else -1
    # poses: x_{1,cam} = R x_{2,cam} + t for the four camera combinations
    Rll, tll = _rot(0.02, -0.04, 0.01), np.array([0.22, -0.02, 0.05])
    Rlr_rig, tlr_rig = _rot(0.0, 0.06, 0.0), np.array([0.10, 0.0, 0.0])   # right camera of a rig w.r.t. its left one
    def compose(Ra, ta, Rb, tb):  # x_a = Ra x_b + ta, x_b = Rb x_c + tb -> x_a = Ra Rb x_c + Ra tb + ta
        return Ra @ Rb, Ra @ tb + ta
    # left1 <- left2: (Rll, tll); left1 <- right2: left2 <- right2 composed; right1 <- *: invert rig transform first
    Rrl_inv, trl_inv = Rlr_rig.T, -Rlr_rig.T @ tlr_rig                        # x_right = R^T x_left - R^T t
    R_lr, t_lr = compose(Rll, tll, Rlr_rig, tlr_rig)
    R_rl, t_rl = compose(Rrl_inv, trl_inv, Rll, tll)
    R_rr, t_rr = compose(R_rl, t_rl, Rlr_rig, tlr_rig)
    Rs = np.stack([Rll, R_lr, R_rl, R_rr]).astype(np.float32)
    ts = np.stack([tll, t_lr, t_rl, t_rr]).astype(np.float32)
    # scene points in the frame of camera 1-left; feature i of set 1 sees point i, feature j of set 2 sees the point
    # of its nearest descriptor in set 1 (so descriptor matches are geometric matches) or a random point
    X = np.stack([rng.uniform(-4, 4, n1), rng.uniform(-3, 3, n1), rng.uniform(0.8, 10, n1)], 1)
    src = D.argmin(0)
    good = D.min(0) < 60
    Xs2 = np.where(good[:, None], X[src], np.stack([rng.uniform(-4, 4, n2), rng.uniform(-3, 3, n2), rng.uniform(0.8, 10, n2)], 1))
    def to_cam1(Xl, right):   # camera-1-left frame -> camera-1-left / right frame
        return np.where(right[:, None], (Rrl_inv @ Xl.T).T + trl_inv, Xl)
    def to_cam2(Xl, right):   # camera-1-left frame -> camera-2-left / right frame
        X2l = (Rll.T @ (Xl - tll).T).T
        return np.where(right[:, None], (Rrl_inv @ X2l.T).T + trl_inv, X2l)
    right1 = (np.arange(n1) >= Nleft1) if rig else np.zeros(n1, bool)
    right2 = (np.arange(n2) >= Nleft2) if rig else np.zeros(n2, bool)
    Xc1, Xc2 = to_cam1(X, right1), to_cam2(Xs2, right2)
    kp1 = np.where(right1[:, None], kb8_project64(P1R, Xc1), kb8_project64(P1L, Xc1)) + rng.normal(0, 0.3, (n1, 2))
    kp2 = np.where(right2[:, None], kb8_project64(P2R, Xc2), kb8_project64(P2L, Xc2)) + rng.normal(0, 0.5, (n2, 2))
    oct1 = rng.integers(0, 8, n1).astype(np.int32)
    oct2 = rng.integers(0, 8, n2).astype(np.int32)
    has1 = (rng.uniform(size=n1) < 0.3).astype(np.uint8)
    has2 = (rng.uniform(size=n2) < 0.3).astype(np.uint8)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    sig = (sf * sf).astype(np.float32)
    return dict(d1=d1, d2=d2, a1=a1, a2=a2, fv1=fv1, fv2=fv2, kp1=kp1.astype(np.float32), kp2=kp2.astype(np.float32),
                oct1=oct1, oct2=oct2, has1=has1, has2=has2, Nleft1=Nleft1, Nleft2=Nleft2, P1L=P1L, P1R=P1R, P2L=P2L,
                P2R=P2R, R12=Rs if rig else Rs[:1], t12=ts if rig else ts[:1], ep=(300.0, 250.0), sf=sf, sig1=sig, sig2=sig)


def tri3d_inputs(n1, n2, seed, rig=False):
    """tri_kb8_inputs plus the world poses (rows 0..2, 3x4) of the cameras 1L, 1R, 2L, 2R that give its relative poses:
    keyframe 1's left camera is the world frame."""
    I = tri_kb8_inputs(n1, n2, seed, rig=rig)
    R64, t64 = I["R12"].astype(np.float64), I["t12"].astype(np.float64)
    Rll, tll = R64[0], t64[0]                                   # x_1 = Rll x_2 + tll
    R2, t2 = Rll.T, -Rll.T @ tll
    T = [np.hstack([np.eye(3), np.zeros((3, 1))])]
    if rig:
        Rrl = R64[2] @ Rll.T                                     # x_right = Rrl x_left + trl inside a rig
        trl = t64[2] - Rrl @ tll
        T += [np.hstack([Rrl, trl[:, None]]), np.hstack([R2, t2[:, None]]), np.hstack([Rrl @ R2, (Rrl @ t2 + trl)[:, None]])]
    else:
        T += [T[0], np.hstack([R2, t2[:, None]]), np.hstack([R2, t2[:, None]])]
    I["Tcw"] = np.stack(T).astype(np.float32)
    return I


def projection_problem(seed, n=1200, nq=900, mode=0, stereo=False, Nleft=-1, th=1.0, nnratio=0.8, taken_frac=0.1,
                       crowd=True, check_orientation=False, partners=False, blocks=None, w=752, h=480, loop=None):
    """Flattened SearchByProjection problem (fields of orbfe_proj_args).  Queries are map points that project
    near existing features, with descriptors a few bits away; `crowd` makes several queries compete for the
    same feature so that the sequential occupancy rule matters."""
    rng = np.random.default_rng(seed)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    kx = rng.uniform(-3, w + 3, n).astype(np.float32)
    ky = rng.uniform(-3, h + 3, n).astype(np.float32)
    if crowd:  # clusters: many features inside one window
        k = n // 4
        cx, cy = rng.uniform(50, w - 50, 12), rng.uniform(50, h - 50, 12)
        c = rng.integers(0, 12, k)
        kx[:k] = (cx[c] + rng.normal(0, 6, k)).astype(np.float32)
        ky[:k] = (cy[c] + rng.normal(0, 6, k)).astype(np.float32)
    octave = rng.integers(0, 8, n).astype(np.int32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    angle = rng.uniform(0, 360, n).astype(np.float32)
    pr = dict(desc=desc, kx=kx, ky=ky, octave=octave, angle=angle, Nleft=Nleft,
              minX=np.float32(-2.5), minY=np.float32(-1.5),
              gridWInv=np.float32(64) / np.float32(w + 4.0), gridHInv=np.float32(48) / np.float32(h + 3.5),
              mode=mode, nnratio=nnratio, th_high=100, check_orientation=int(check_orientation))
    pr["taken"] = (rng.random(n) < taken_frac).astype(np.uint8)
    if stereo:
        ur = np.where(rng.random(n) < 0.6, kx - rng.uniform(1, 40, n), -1).astype(np.float32)
        pr["uright"] = ur
    tgt = rng.integers(0, n, nq)
    if crowd:
        tgt[: nq // 3] = rng.integers(0, max(n // 4, 1), nq // 3)  # compete inside the clusters
        tgt[nq // 3: nq // 2] = tgt[: nq // 2 - nq // 3]           # exact duplicates of earlier targets
    qdesc = desc[tgt].copy()
    bits = np.unpackbits(qdesc, axis=1)
    flips = rng.random(bits.shape) < rng.uniform(0.0, 0.25, (nq, 1))
    qdesc = np.packbits(bits ^ flips, axis=1)
    qx = (kx[tgt] + rng.normal(0, 2.0, nq)).astype(np.float32)
    qy = (ky[tgt] + rng.normal(0, 2.0, nq)).astype(np.float32)
    lvl = np.clip(octave[tgt] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    if mode == 0:  # RadiusByViewingCos (:199-205) times th
        rad = np.where(rng.random(nq) < 0.5, np.float32(2.5), np.float32(4.0)).astype(np.float32)
        if th != 1.0:
            rad = (rad * np.float32(th)).astype(np.float32)
    else:          # radius = th * mvScaleFactors[nLastOctave] (:2244)
        rad = np.full(nq, th, np.float32)
    pr["qr"] = (rad * sf[lvl]).astype(np.float32)
    if mode == 0:
        pr["qmin_level"] = (lvl - 1).astype(np.int32)
        pr["qmax_level"] = lvl
    else:
        kind = rng.integers(0, 3, nq)  # forward / backward / neither (:2248-2253)
        pr["qmin_level"] = np.where(kind == 0, lvl, np.where(kind == 1, 0, lvl - 1)).astype(np.int32)
        pr["qmax_level"] = np.where(kind == 0, -1, np.where(kind == 1, lvl, lvl + 1)).astype(np.int32)
    pr.update(qdesc=qdesc, qx=qx, qy=qy, qangle=rng.uniform(0, 360, nq).astype(np.float32))
    if stereo:
        t_ur = pr["uright"][tgt]
        pr["qxr"] = np.where(t_ur > 0, t_ur + rng.normal(0, 3.0, nq), qx - 10).astype(np.float32)
    if Nleft != -1:
        right = (tgt >= Nleft)
        flags = right.astype(np.uint8)
        prev_left = np.concatenate([[False], ~right[:-1]])
        if mode == 0:  # right-camera searches directly after a left one are linked to it
            flags |= ((right & prev_left & (rng.random(nq) < 0.7)).astype(np.uint8) << 1)
        elif loop is None:  # frame-to-frame: the right search of a point is skipped when its left window was empty (:2255)
            behind = right & prev_left & (rng.random(nq) < 0.7)
            flags |= behind.astype(np.uint8) << 2
            # empty a good share of those left windows: far outside the grid, or no level passes
            prev = np.nonzero(behind)[0] - 1
            far = prev[rng.random(len(prev)) < 0.25]
            qx[far] = np.float32(w + 400)
            lvlless = prev[rng.random(len(prev)) < 0.25]
            pr["qmin_level"][lvlless] = 9
        pr["qflags"] = flags
        if partners:
            nr = n - Nleft
            l2r = np.full(Nleft, -1, np.int32)
            r2l = np.full(nr, -1, np.int32)
            m = min(Nleft, nr) // 3
            li = rng.permutation(Nleft)[:m]
            ri = rng.permutation(nr)[:m]
            l2r[li] = ri
            r2l[ri] = li
            pr["left_to_right"], pr["right_to_left"] = l2r, r2l
    if blocks is not None:
        pr["qblocks"] = (rng.random(nq) < blocks).astype(np.uint8)
    # the other ORBmatcher loops that run on mode 1 (include/orbfe.h):
    if loop in ("sim3_projection", "fuse", "fuse_sim3", "search_by_sim3"):
        assert mode == 1
        pr["qmin_level"] = (lvl - 1).astype(np.int32)  # kpLevel<nPredictedLevel-1 || kpLevel>nPredictedLevel
        pr["qmax_level"] = lvl
        pr["check_orientation"] = 0
        pr["th_high"] = {"sim3_projection": int(np.floor(50 * 0.9)), "fuse": 50, "fuse_sim3": 50, "search_by_sim3": 100}[loop]
        if loop != "sim3_projection":                  # independent queries: nothing a query takes is hidden
            pr["qblocks"] = np.zeros(nq, np.uint8)
            pr["taken"] = np.zeros(n, np.uint8)
        if loop == "fuse":                             # per-candidate chi2 test on the reprojection error
            pr["chi2_gate"] = 1
            sigma2 = (sf * sf).astype(np.float32)
            pr["inv_level_sigma2"] = (np.float32(1.0) / sigma2).astype(np.float32)
            if stereo:                                 # mvuRight >= 0 selects the 3-dof test (7.8)
                pr["uright"] = np.where(rng.random(n) < 0.5, kx - rng.uniform(0, 30, n), -1).astype(np.float32)
                t_ur = pr["uright"][tgt]
                pr["qxr"] = np.where(t_ur >= 0, t_ur + rng.normal(0, 1.5, nq), qx - 10).astype(np.float32)
        else:
            pr.pop("uright", None)
            pr.pop("qxr", None)
    return pr
