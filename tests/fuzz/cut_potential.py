#!/usr/bin/env python
"""CPU helper (round 5; uses the CPU oracle's extractor, hence under tests/): how early is a candidate of
config 2 DECIDED -- its partial sum plus 1.0 per remaining tree can no longer exceed thre * T -- per
candidate, per 64-candidate wave and per 256-candidate tile, and what a cut of the forest at k trees
with a compaction of the open candidates behind it would cost.  The figures behind q_pick_cut
(peakachu_amd/csrc/pk_forest_q.hip).  Also: the same question for a bound that walks every tree a few
levels only (maximum leaf value below the node reached): useless, every subtree holds a 1.0 leaf.

usage: python tests/fuzz/cut_potential.py [tiles=600] [w=5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
from peakachu_amd.forest import FlatForest  # noqa: E402


def per_tree_values(fo, fea, depth_bounds=()):
    """[T, N] leaf values of every tree for float32 feature rows (numpy walk, sklearn's comparison), and
    for every depth in depth_bounds the sum over the trees of the largest leaf value below the node a
    candidate has reached after that many levels."""
    N = fea.shape[0]
    T = fo.tree_off.size - 1
    V = np.zeros((T, N))
    UB = {d: np.zeros(N) for d in depth_bounds}
    levels = np.zeros(N)
    for t in range(T):
        o, e = fo.tree_off[t], fo.tree_off[t + 1]
        L, R, Fe, Th, P = fo.left[o:e], fo.right[o:e], fo.feat[o:e], fo.thr[o:e], fo.p1[o:e]
        submax = np.where(L == -1, P, -1.0)
        for i in range(e - o - 1, -1, -1):   # preorder: children behind their parent
            if L[i] != -1:
                submax[i] = max(submax[L[i]], submax[R[i]])
        node = np.zeros(N, np.int64)
        lev = 0
        while True:
            a = np.nonzero(L[node] != -1)[0]
            if a.size == 0 and lev >= max(depth_bounds, default=0):
                break
            if a.size:
                nd = node[a]
                go = fea[a, Fe[nd]].astype(np.float64) <= Th[nd]
                node[a] = np.where(go, L[nd], R[nd])
                levels[a] += 1
            lev += 1
            if lev in UB:
                UB[lev] += submax[node]
        V[t] = P[node]
    return V, UB, levels / T


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    Mf, e, x, y, upper = bench.build_workload(0, 30000, 200 if w == 5 else 300, w, 6, 200 if w == 5 else 300)
    fo = bench.load_forest(None, w, (2 * w + 1) ** 2)
    rng = np.random.default_rng(1)
    starts = rng.integers(0, x.size // 256 - 1, tiles) * 256
    idx = (starts[:, None] + np.arange(256)[None, :]).reshape(-1)
    fea = onp.extract(Mf, e, w, x[idx], y[idx])[0].astype(np.float32)
    V, UB, lev = per_tree_values(fo, fea, (4, 8, 12, 16))
    T, N = V.shape
    p = V.sum(0) / T
    print("w=%d: %d candidates in %d tiles; mean levels per walk %.1f" % (w, N, tiles, lev.mean()))
    print("p quantiles 50/90/99/99.9 %s  max %.3f  share > 0.5: %.5f" % (np.quantile(p, [.5, .9, .99, .999]).round(3), p.max(), (p > 0.5).mean()))
    for d, ub in UB.items():
        print("bound after %2d levels of every tree: mean %.3f, candidates it rejects at 0.5: %.5f" % (d, ub.mean() / T, (ub <= 0.5 * T).mean()))
    S = np.cumsum(V, 0)
    for thre in (0.5, 0.6, 0.7, 0.9):
        print("threshold %.1f" % thre)
        for name, gran in (("candidate", 1), ("wave of 64", 64), ("tile of 256", 256)):
            n = N // gran * gran
            ks = np.arange(8, T + 1, 8)
            Sm = S[ks - 1][:, :n].reshape(len(ks), -1, gran).max(2)
            ok = Sm <= (thre * T - (T - ks))[:, None]
            first = np.where(ok.any(0), ks[ok.argmax(0)], T)
            print("   decided per %-12s (checked every 8 trees): after %.1f trees on average" % (name, first.mean()))
        for k in range(8, T, 8):
            lim = thre * T - (T - k)
            if lim < 0:
                continue
            und = (S[k - 1] > lim).mean()
            print("   cut at %3d trees: %.4f still open -> forest work x %.3f" % (k, und, k / T + und * (T - k) / T))


if __name__ == "__main__":
    main()
