#!/usr/bin/env python
"""CPU study: LDS bank conflicts of the pair reads of forest_q_kernel for different
numberings of a level's node pairs.  A wave reads, per level, the pair of each lane's
current node with one ds_read_b64 (two halves of 32 lanes; a half is conflict-free when
its distinct pairs fall into distinct 8-byte bank pairs = index mod 32).
Cost of a half = max over the 32 bank pairs of the number of DISTINCT pair indices."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import oracle_np
from peakachu_amd.forest import FlatForest

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
Mf, e, x, y, upper = bench.build_workload(0, n, 200, 5, 6, 200)
fo = FlatForest.load("peakachu_amd/data/forest_w5_t100.npz")
sel = slice(0, 64 * 300)
fea, keep = oracle_np.extract(Mf, e, 5, x[sel], y[sel])
fea = fea.astype(np.float32)
nw = fea.shape[0] // 64
fea = fea[:nw * 64]
print("candidates", fea.shape[0], "waves", nw)

def numbering(t, mode):
    o, o1 = fo.tree_off[t], fo.tree_off[t + 1]
    left, right = fo.left[o:o1], fo.right[o:o1]
    p1 = fo.p1[o:o1]
    nn = o1 - o
    dep = np.zeros(nn, int); size = np.ones(nn, int)
    order = []
    st = [0]
    while st:
        v = st.pop()
        if left[v] == -1: continue
        order.append(v)
        dep[left[v]] = dep[right[v]] = dep[v] + 1
        st.append(right[v]); st.append(left[v])
    for v in reversed(order):
        size[v] = 1 + size[left[v]] + size[right[v]]
    pure = lambda v: left[v] == -1 and (p1[v] == 0.0 or p1[v] == 1.0)
    idx = np.full(nn, -1)
    nxt = 8
    if mode == "level":
        key = lambda v: (dep[v],)
    elif mode == "level_size":
        key = lambda v: (dep[v], -size[v])
    elif mode == "pre":
        key = lambda v: (0,)
    lst = sorted(order, key=key)
    for v in lst:
        if pure(left[v]) and pure(right[v]):
            vl, vr = int(p1[left[v]]), int(p1[right[v]])
            idx[v] = 4 + (0 if (vl, vr) == (0, 1) else 1 if (vl, vr) == (1, 0) else 2 if vl == 0 else 3)
        else:
            idx[v] = nxt; nxt += 1
    vb = {}
    for v in range(nn):
        if left[v] == -1:
            if p1[v] == 0.0: idx[v] = 0
            elif p1[v] == 1.0: idx[v] = 2
            else:
                k = p1[v].tobytes() if hasattr(p1[v], "tobytes") else p1[v]
                if k not in vb: vb[k] = nxt; nxt += 2
                idx[v] = vb[k]
    return idx, int(dep.max())

def cost_half(ix):
    u = np.unique(ix)
    return np.bincount(u % 32, minlength=32).max()

tot = {}
per_level = {}
for mode in ("pre", "level", "level_size"):
    c = 0; ideal = 0
    pl = np.zeros(24)
    for t in range(0, fo.T, 5):
        idx, D = numbering(t, mode)
        o = fo.tree_off[t]
        left, right, feat, thr = fo.left[o:], fo.right[o:], fo.feat[o:], fo.thr[o:]
        node = np.zeros(fea.shape[0], int)
        for d in range(D):
            pi = idx[node].reshape(nw, 64)
            for wv in range(nw):
                k = cost_half(pi[wv, :32]) + cost_half(pi[wv, 32:])
                c += k; pl[d] += k
            ideal += 2 * nw
            isleaf = left[node] == -1
            f = np.where(isleaf, 0, feat[node])
            go_left = fea[np.arange(fea.shape[0]), f] <= thr[node].astype(np.float32)
            nxt = np.where(go_left, left[node], right[node])
            node = np.where(isleaf, node, nxt)
    tot[mode] = c / ideal
    per_level[mode] = pl / (2 * nw * len(range(0, fo.T, 5)))
    print(mode, "mean passes per half-wave read: %.3f" % tot[mode])
for mode in per_level:
    print(mode, np.round(per_level[mode][:20], 2).tolist())
