"""Sequential Python model of the breadth-first, closed-form introsort emulation that
csrc/hamming_map.hip::map_query_kernel runs on the GPU (same decomposition: median-to-first,
L/R stopper lists from ORIGINAL values, swap the prefix of pairs with L_i < R_i,
cut = min(L_s, R_{s-1}), heapsort when the depth budget hits 0, stable sort of each leaf).
Test-only; lets the algorithm be checked against libstdc++ on the CPU without a GPU."""
import numpy as np

LEAF = 16


def _median_to_first(e, f, l):
    a, b, c = f + 1, f + (l - f) // 2, l - 1
    ka, kb, kc = e[a][0], e[b][0], e[c][0]
    if ka < kb:
        pick = b if kb < kc else (c if ka < kc else a)
    elif ka < kc:
        pick = a
    elif kb < kc:
        pick = c
    else:
        pick = b
    e[f], e[pick] = e[pick], e[f]


def _partition(e, f, l):
    _median_to_first(e, f, l)
    p, n = e[f][0], l - f
    cap = n // 2 + 1
    Lall = [x for x in range(f + 1, l) if e[x][0] >= p]
    Rall = [x for x in range(l - 1, f, -1) if e[x][0] <= p]
    L, R = Lall[:cap], Rall[:cap]
    npairs = min(len(Lall), len(Rall), cap)
    s = 0
    for i in range(npairs):
        if L[i] < R[i]:
            e[L[i]], e[R[i]] = e[R[i]], e[L[i]]
            s += 1
    c1 = L[s] if s < min(len(Lall), cap) else 1 << 30
    c2 = R[s - 1] if s >= 1 else 1 << 30
    return min(c1, c2)


def _adjust_heap(a, base, hole, length, value):
    top, child = hole, hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if a[base + child][0] < a[base + child - 1][0]:
            child -= 1
        a[base + hole] = a[base + child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        a[base + hole] = a[base + child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and a[base + parent][0] < value[0]:
        a[base + hole] = a[base + parent]
        hole = parent
        parent = (hole - 1) // 2
    a[base + hole] = value


def _heap_sort(e, f, l):
    length = l - f
    if length < 2:
        return
    parent = (length - 2) // 2
    while True:
        _adjust_heap(e, f, parent, length, e[f + parent])
        if parent == 0:
            break
        parent -= 1
    last = length
    while last > 1:
        last -= 1
        v = e[f + last]
        e[f + last] = e[f]
        _adjust_heap(e, f, 0, last, v)


def emulate(keys, depth_override=None):
    """-> permutation (int64) equal to libstdc++ std::sort on (key, idx) with a key-only '<'."""
    N = len(keys)
    e = [(int(k), i) for i, k in enumerate(keys)]
    starts = {0}
    queue = [(0, N)] if N > LEAF else []
    depth = 2 * (N.bit_length() - 1) if depth_override is None else depth_override
    while queue:
        if depth == 0:
            for f, l in queue:
                _heap_sort(e, f, l)
            break
        depth -= 1
        nxt = []
        for f, l in queue:
            cut = _partition(e, f, l)
            starts.add(cut)
            if cut - f > LEAF:
                nxt.append((f, cut))
            if l - cut > LEAF:
                nxt.append((cut, l))
        queue = nxt
    st = sorted(starts) + [N]
    for s0, e1 in zip(st[:-1], st[1:]):
        if e1 - s0 <= LEAF:
            seg = e[s0:e1]
            seg.sort(key=lambda t: t[0])      # stable
            e[s0:e1] = seg
    return np.array([i for _, i in e], np.int64)
