"""Level-synchronous restatement of DistributeOctTree (reference src/ORBextractor.cc:537-761).

This is the algorithm the K-QT device kernel implements (keys never move; every key
carries the index of the list node that holds it; passes are histogram + prefix-sum
steps).  tests/test_oracle_octree.py checks it against the oracle's literal std::list
transcription, which pins the kernel's design before any HIP is involved.
"""
import numpy as np


def _child_bounds(b, q):
    ulx, uly, brx, bry = b
    hx = (brx - ulx + 1) >> 1  # ceil((float)(UR.x-UL.x)/2) for non-negative ints
    hy = (bry - uly + 1) >> 1
    if q == 0:
        return (ulx, uly, ulx + hx, uly + hy)
    if q == 1:
        return (ulx + hx, uly, brx, uly + hy)
    if q == 2:
        return (ulx, uly + hy, ulx + hx, bry)
    return (ulx + hx, uly + hy, brx, bry)


def _quadrant(b, x, y):
    ulx, uly, brx, bry = b
    hx = (brx - ulx + 1) >> 1
    hy = (bry - uly + 1) >> 1
    left = x < ulx + hx
    top = y < uly + hy
    if left:
        return 0 if top else 2
    return 1 if top else 3


def distribute(xs, ys, resp, minX, maxX, minY, maxY, N):
    """xs, ys: integer-valued candidate coords (relative to minX/minY); returns indices of kept keys in output order."""
    n = len(xs)
    W, H = maxX - minX, maxY - minY
    nIni = int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5)))  # C round() of a positive float
    if nIni < 1:
        return []
    hX = np.float32(W) / np.float32(nIni)
    bounds = []
    for i in range(nIni):
        bounds.append((int(np.float32(hX) * np.float32(i)), 0, int(np.float32(hX) * np.float32(i + 1)), H))
    key_node = np.array([min(int(np.float32(x) / hX), nIni - 1) for x in xs], dtype=np.int64)
    counts = np.bincount(key_node, minlength=nIni) if n else np.zeros(nIni, np.int64)
    # initial list = non-empty roots
    remap = -np.ones(nIni, np.int64)
    lst = []
    for i in range(nIni):
        if counts[i] > 0:
            remap[i] = len(lst)
            lst.append([bounds[i], int(counts[i])])
    key_node = remap[key_node] if n else key_node

    def expand(nodes_to_expand_in_push_order, lst, key_node):
        """Expand the given list indices (in the order the reference expands them).
        Returns new list, new key_node and the multi-key children in creation order."""
        exp_set = {p: k for k, p in enumerate(nodes_to_expand_in_push_order)}
        cc = np.zeros((len(nodes_to_expand_in_push_order), 4), np.int64)
        kq = np.zeros(n, np.int64)
        for i in range(n):
            p = key_node[i]
            if p in exp_set:
                q = _quadrant(lst[p][0], xs[i], ys[i])
                kq[i] = q
                cc[exp_set[p], q] += 1
        new_lst = []
        child_pos = {}
        for k in range(len(nodes_to_expand_in_push_order) - 1, -1, -1):
            p = nodes_to_expand_in_push_order[k]
            for q in (3, 2, 1, 0):
                if cc[k, q] > 0:
                    child_pos[(k, q)] = len(new_lst)
                    new_lst.append([_child_bounds(lst[p][0], q), int(cc[k, q])])
        old_pos = {}
        for p in range(len(lst)):
            if p not in exp_set:
                old_pos[p] = len(new_lst)
                new_lst.append(lst[p])
        new_key_node = key_node.copy()
        for i in range(n):
            p = key_node[i]
            if p in exp_set:
                new_key_node[i] = child_pos[(exp_set[p], kq[i])]
            else:
                new_key_node[i] = old_pos[p]
        multi = []
        for k in range(len(nodes_to_expand_in_push_order)):
            for q in range(4):
                if cc[k, q] > 1:
                    multi.append(child_pos[(k, q)])
        return new_lst, new_key_node, multi, cc

    finish = False
    while not finish:
        prev = len(lst)
        to_exp = [p for p in range(len(lst)) if lst[p][1] > 1]
        if not to_exp:
            break
        lst, key_node, multi, _ = expand(to_exp, lst, key_node)
        size = len(lst)
        if size >= N or size == prev:
            finish = True
        elif size + 3 * len(multi) > N:
            while not finish:
                prev = len(lst)
                # descending (count, creation seq)
                order = sorted(range(len(multi)), key=lambda k: (lst[multi[k]][1], k), reverse=True)
                # growth of each candidate = nonempty children - 1 -> find the prefix that reaches N
                chosen = []
                size = len(lst)
                for k in order:
                    p = multi[k]
                    qs = set()
                    for i in range(n):
                        if key_node[i] == p:
                            qs.add(_quadrant(lst[p][0], xs[i], ys[i]))
                    chosen.append(p)
                    size += len(qs) - 1
                    if size >= N:
                        break
                lst, key_node, multi, _ = expand(chosen, lst, key_node)
                assert len(lst) == size
                if size >= N or size == prev:
                    finish = True
    out = []
    for p in range(len(lst)):
        best = -1
        for i in range(n):
            if key_node[i] == p and (best < 0 or resp[i] > resp[best]):
                best = i
        out.append(best)
    return out
