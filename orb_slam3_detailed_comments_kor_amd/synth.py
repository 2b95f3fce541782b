"""Seeded synthetic grayscale frames with EuRoC-like corner density.

There are no datasets in the build or measurement environment, so the parity tests
and bench.py use these frames: a low-frequency value-noise
background, a few hundred random axis-aligned / rotated rectangles of uniform grey
(FAST corners at many scales), additive Gaussian noise, plus a flat region so that
some cells exercise the minThFAST fallback (reference src/ORBextractor.cc:825-828).
"""
import numpy as np


def _value_noise(rng, h, w, lattice=64, sigma=40.0):
    gh, gw = h // lattice + 2, w // lattice + 2
    g = rng.normal(0.0, sigma, size=(gh, gw))
    ys = np.arange(h) / lattice
    xs = np.arange(w) / lattice
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_frame(h, w, seed, nrect=None, noise_sigma=2.0, flat_frac=0.08):
    """Return an (h, w) uint8 frame; deterministic in (h, w, seed)."""
    rng = np.random.default_rng(int(seed))
    img = 128.0 + _value_noise(rng, h, w)
    if nrect is None:
        nrect = int(300 + 300 * (h * w) / (752.0 * 480.0) ** 1.0 * 0.5)
        nrect = min(nrect, 900)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nrect):
        cx = rng.uniform(0, w)
        cy = rng.uniform(0, h)
        hw = rng.uniform(3, 0.08 * w)
        hh = rng.uniform(3, 0.08 * h)
        grey = rng.uniform(0, 255)
        if rng.uniform() < 0.5:
            ang = 0.0
        else:
            ang = rng.uniform(0, np.pi)
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        ca, sa = np.cos(ang), np.sin(ang)
        u = dx * ca + dy * sa
        v = -dx * sa + dy * ca
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        sub = img[y0:y1, x0:x1]
        sub[m] = grey
    # a flat patch (exercises the minThFAST fallback and empty cells)
    fw = int(w * np.sqrt(flat_frac))
    fh = int(h * np.sqrt(flat_frac))
    fx0 = int(rng.uniform(0, w - fw))
    fy0 = int(rng.uniform(0, h - fh))
    img[fy0:fy0 + fh, fx0:fx0 + fw] = 0.5 * img[fy0:fy0 + fh, fx0:fx0 + fw].mean() + 60.0
    img[fy0:fy0 + fh // 2, fx0:fx0 + fw // 2] += 9.0  # weak corner: only visible at minThFAST
    img += rng.normal(0.0, noise_sigma, size=(h, w))
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def _blur(img, sigma):
    """Separable Gaussian blur (float64, reflect-101 borders), self-contained so that the frames do not depend on scipy."""
    r = max(1, int(np.ceil(3.0 * sigma)))
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    a = np.pad(np.asarray(img, np.float64), r, mode="reflect")
    a = sum(k[i] * a[:, i:i + a.shape[1] - 2 * r] for i in range(2 * r + 1))
    a = sum(k[i] * a[i:i + a.shape[0] - 2 * r, :] for i in range(2 * r + 1))
    return a


FRAME_KINDS = ("rects", "blurred", "plateaus", "checker2", "sinus", "ramp", "mixed")


def make_frame_kind(h, w, seed, kind):
    """Frames that stress what `make_frame` does not; deterministic in (h, w, seed, kind).

    rects     make_frame itself.
    blurred   make_frame blurred with sigma 3..6: nearly every FAST cell falls through to the minThFAST call
              (reference src/ORBextractor.cc:825-828) and most of those stay empty.
    plateaus  saturated 0 / 255 regions and large exactly-flat regions between textured ones: runs of equal pixels at the
              clamp values, score ties, empty cells next to busy ones.
    checker2  a 2-px checkerboard (contrast from the seed) with a few flat holes: thousands of equal scores -> the strict
              `>` of the 8-neighbour NMS and the quadtree's (size, sequence) ties at scale.
    sinus     fine sinusoidal texture (periods 3..7 px, two orientations) on a slow gradient.
    ramp      a pure linear ramp (no corner anywhere at iniThFAST; a few at low thresholds from quantisation steps).
    mixed     quadrants of the above, seams included."""
    rng = np.random.default_rng(int(seed) * 7 + 13)
    if kind == "rects":
        return make_frame(h, w, seed)
    if kind == "blurred":
        sigma = float(rng.uniform(3.0, 6.0))
        return np.ascontiguousarray(np.clip(np.rint(_blur(make_frame(h, w, seed), sigma)), 0, 255).astype(np.uint8))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "plateaus":
        img = make_frame(h, w, seed, noise_sigma=1.0).astype(np.float64)
        img = (img - 128.0) * 2.6 + 128.0  # stretch: a good part of the frame clips to 0 / 255
        for _ in range(int(rng.integers(6, 14))):  # exactly flat boxes, some at the clamp values
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = min(w, x0 + int(rng.integers(20, max(21, w // 3)))), min(h, y0 + int(rng.integers(20, max(21, h // 3))))
            img[y0:y1, x0:x1] = float(rng.choice([0.0, 255.0, 255.0, 0.0, rng.uniform(0, 255)]))
        return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    if kind == "checker2":
        lo = float(rng.uniform(0, 110))
        hi = lo + float(rng.uniform(30, 145))
        p = int(rng.choice([2, 2, 3]))
        img = np.where(((xx // p) + (yy // p)) % 2 == 0, lo, hi)
        for _ in range(int(rng.integers(3, 9))):  # flat holes: corners along their rims, empty cells inside
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y0:y0 + int(rng.integers(10, 90)), x0:x0 + int(rng.integers(10, 90))] = float(rng.uniform(0, 255))
        return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    if kind == "sinus":
        p1, p2 = float(rng.uniform(3, 7)), float(rng.uniform(3, 7))
        a1, a2 = float(rng.uniform(0, np.pi)), float(rng.uniform(0, np.pi))
        amp = float(rng.uniform(25, 60))
        img = 128.0 + 50.0 * (xx / max(w - 1, 1) - 0.5) + amp * np.sin(2 * np.pi * (xx * np.cos(a1) + yy * np.sin(a1)) / p1) \
            * np.sin(2 * np.pi * (xx * np.cos(a2) + yy * np.sin(a2)) / p2)
        return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    if kind == "ramp":
        gx, gy = float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4))
        img = 128.0 + gx * (xx - w / 2.0) + gy * (yy - h / 2.0)
        return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    if kind == "mixed":
        out = make_frame(h, w, seed).copy()
        hy, hx = h // 2, w // 2
        out[:hy, hx:] = make_frame_kind(h, w, seed + 1, "checker2")[:hy, hx:]
        out[hy:, :hx] = make_frame_kind(h, w, seed + 2, "blurred")[hy:, :hx]
        out[hy:, hx:] = make_frame_kind(h, w, seed + 3, "plateaus")[hy:, hx:]
        return np.ascontiguousarray(out)
    raise ValueError("unknown frame kind %r" % (kind,))


def narrow_last_column_widths(scale=1.2, nlevels=8, lo=300, hi=2000):
    """Image widths at which the LAST FAST cell column of some pyramid level is at most 12 px wide, i.e. its detection zone
    is 1..6 px wide, or the column is skipped altogether by the `iniX >= maxBorderX - 6` test of
    src/ORBextractor.cc:792-802 (zone <= 0).  Returns (width, level, zone_width) triples, the arithmetic restated from
    :771-802 (minBorderX = 16, maxBorderX = cols - 16, W = 35).  Needs >= 25 cell columns, i.e. levels >= ~900 px wide."""
    out = []
    for wd in range(lo, hi):
        for lvl in range(nlevels):
            wl = int(np.rint(np.float32(wd) * (np.float32(1.0) / np.float32(scale) ** np.float32(lvl))))
            width = wl - 32
            if width < 35:
                break
            ncols = int(np.float32(width) / np.float32(35))
            wcell = int(np.ceil(np.float32(width) / np.float32(ncols)))
            zone = width - (ncols - 1) * wcell - 6
            if zone <= 6:
                out.append((wd, lvl, zone))
                break
    return out


def make_stereo_pair(h, w, seed, shift=24):
    """Left frame and a horizontally shifted, re-noised right frame (config C3)."""
    left = make_frame(h, w, seed)
    rng = np.random.default_rng(int(seed) + 7919)
    right = np.empty_like(left)
    right[:, : w - shift] = left[:, shift:]
    right[:, w - shift:] = left[:, w - shift - 1: w - 2 * shift - 1: -1] if shift > 0 else left[:, :0]
    right = np.clip(right.astype(np.float64) + rng.normal(0, 1.5, size=right.shape), 0, 255)
    return left, np.ascontiguousarray(np.rint(right).astype(np.uint8))


def make_batch(n, h, w, seed0=1234):
    """n frames, frame i uses seed seed0+i."""
    return np.stack([make_frame(h, w, seed0 + i) for i in range(n)], axis=0)


def make_feature_vectors(desc, seed, branching=10, depth=2):
    """Synthetic DBoW2-like FeatureVector in CSR form.

    Assigns every 32-byte descriptor to the nearest (Hamming) of branching**depth
    random 256-bit centroids; node ids ascending, feature indices ascending in a node.
    Returns (node_ids uint32[nn], offsets int32[nn+1], indices int32[N]).
    """
    rng = np.random.default_rng(int(seed))
    ncent = branching ** depth
    cent = rng.integers(0, 256, size=(ncent, 32), dtype=np.uint8)
    d = np.asarray(desc, dtype=np.uint8).reshape(-1, 32)
    if d.shape[0] == 0:
        return (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    x = np.bitwise_xor(d[:, None, :], cent[None, :, :])
    dist = np.unpackbits(x, axis=2).sum(axis=2)
    node = dist.argmin(axis=1)
    order = np.argsort(node, kind="stable")
    nodes_sorted = node[order]
    node_ids, counts = np.unique(nodes_sorted, return_counts=True)
    offsets = np.zeros(len(node_ids) + 1, np.int32)
    offsets[1:] = np.cumsum(counts)
    return node_ids.astype(np.uint32), offsets, order.astype(np.int32)


def make_vocabulary(seed, k=10, L=4, ragged=True):
    """Synthetic DBoW2-style vocabulary tree (the real ORBvoc.txt, k=10 L=6, is not available here).

    Breadth-first node ids like DBoW2's loader produces; node 0 is the root.  With ragged=True some
    inner nodes have fewer than k children and some branches end early (leaves at different depths),
    which DBoW2 trees built by k-means can show.  Children descriptors are noisy copies of their parent
    so that descending the tree is meaningful.  Returns the CSR dict used by the oracle and the product.
    """
    rng = np.random.default_rng(int(seed))
    desc = [np.zeros(32, np.uint8)]
    children = [[]]
    depth = [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for p in frontier:
            if ragged and lvl > 1 and rng.uniform() < 0.08:
                continue  # early leaf
            nk = k if not ragged else int(rng.integers(max(2, k - 3), k + 1))
            for _ in range(nk):
                bits = np.unpackbits(desc[p]) if lvl > 1 else rng.integers(0, 2, 256, dtype=np.uint8)
                flip = rng.permutation(256)[: int(rng.integers(20, 70))]
                bits = bits.copy()
                bits[flip] ^= 1
                desc.append(np.packbits(bits))
                children.append([])
                depth.append(lvl)
                children[p].append(len(desc) - 1)
                nxt.append(len(desc) - 1)
        frontier = nxt
    nn = len(desc)
    child_off = np.zeros(nn + 1, np.int32)
    for i in range(nn):
        child_off[i + 1] = child_off[i] + len(children[i])
    child_ids = np.array([c for ch in children for c in ch], np.int32)
    word = -np.ones(nn, np.int32)
    weight = np.zeros(nn, np.float64)
    wid = 0
    for i in range(nn):
        if not children[i] and i != 0:
            word[i] = wid
            wid += 1
            weight[i] = float(rng.uniform(0.5, 9.0)) if rng.uniform() > 0.02 else 0.0  # a few stop words
    return dict(desc=np.ascontiguousarray(np.stack(desc)), child_off=child_off, child_ids=child_ids, word=word,
                weight=weight, L=L)


def make_vocabulary_full(seed, k=10, L=6, flip_p=0.18):
    """A COMPLETE k-ary tree of depth L in the layout of make_vocabulary -- the size of the vocabulary ORB-SLAM3 ships
    (Vocabulary/ORBvoc.txt: k = 10, L = 6, 1 111 111 nodes incl. the root, 10^6 words, ~35 MB of descriptors; the file
    itself is a blob this repo does not have).  Vectorised: level by level, the children of a node are copies of it with
    each bit flipped with probability flip_p (the first level is random).  Breadth-first ids, node 0 = root."""
    rng = np.random.default_rng(int(seed))
    levels = [np.zeros((1, 32), np.uint8)]
    for lvl in range(1, L + 1):
        parent = levels[-1]
        n = parent.shape[0] * k
        out = np.empty((n, 32), np.uint8)
        for i0 in range(0, n, 1 << 17):
            i1 = min(n, i0 + (1 << 17))
            if lvl == 1:
                out[i0:i1] = rng.integers(0, 256, size=(i1 - i0, 32), dtype=np.uint8)
            else:
                flips = np.packbits(rng.random((i1 - i0, 256)) < flip_p, axis=1)
                out[i0:i1] = parent[np.arange(i0, i1) // k] ^ flips
        levels.append(out)
    desc = np.ascontiguousarray(np.concatenate(levels))
    nn = desc.shape[0]
    first = np.cumsum([0] + [lv.shape[0] for lv in levels])       # first node id of every level
    inner = int(first[L])                                          # nodes with children
    child_off = np.zeros(nn + 1, np.int32)
    child_off[1:inner + 1] = k * np.arange(1, inner + 1)
    child_off[inner + 1:] = k * inner
    child_ids = np.arange(1, nn, dtype=np.int32)                   # breadth-first: node i's children are 1 + k i .. k + k i
    word = -np.ones(nn, np.int32)
    word[inner:] = np.arange(nn - inner, dtype=np.int32)
    weight = np.zeros(nn, np.float64)
    weight[inner:] = rng.uniform(0.5, 9.0, nn - inner)
    weight[inner:][rng.random(nn - inner) < 0.02] = 0.0            # a few stop words
    return dict(desc=desc, child_off=child_off, child_ids=child_ids, word=word, weight=weight, L=L)
