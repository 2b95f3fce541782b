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
