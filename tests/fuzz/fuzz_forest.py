#!/usr/bin/env python
"""GPU-box helper: random forests of many shapes through HipForest.predict against the CPU
oracle, bit for bit.  Shapes: 1 .. 1100 features (narrow word, wide word, too many for
either), stumps to depth-40 combs, 1 .. 300 trees, trees too large for the pair field, more
than 2047 thresholds on one feature, missing_go_to_left nodes, NaN / inf / exact-threshold
inputs, 1 .. 70 000 rows (partial workgroups, persistent launches).
usage: tests/fuzz/fuzz_forest.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_np as onp
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest


def random_forest(rng, F, T, max_nodes, depth, miss_frac, thr_pool=None, comb=False):
    offs, cols = [0], {k: [] for k in ("left", "right", "feat", "thr", "miss_left", "p1")}
    for _ in range(T):
        left, right, feat, thr, p1, dep = [-1], [-1], [-2], [-2.0], [float(rng.random())], [0]
        frontier = [0]
        while frontier and len(left) + 2 <= max_nodes:
            i = frontier.pop(-1 if comb else int(rng.integers(0, len(frontier))))
            if dep[i] >= depth:
                continue
            feat[i] = int(rng.integers(0, F))
            thr[i] = float(rng.choice(thr_pool)) if thr_pool is not None else float(rng.random() * 2 - 0.5)
            for side in (left, right):
                side[i] = len(left)
                left.append(-1); right.append(-1); feat.append(-2); thr.append(-2.0)
                p1.append(float(rng.integers(0, 2)) if rng.random() < 0.8 else float(rng.random()))
                dep.append(dep[i] + 1)
                frontier.append(len(left) - 1)
        n = len(left)
        cols["left"].append(np.array(left, np.int32)); cols["right"].append(np.array(right, np.int32))
        cols["feat"].append(np.array(feat, np.int32)); cols["thr"].append(np.array(thr, np.float64))
        m = (rng.random(n) < miss_frac).astype(np.uint8)
        cols["miss_left"].append(m); cols["p1"].append(np.array(p1, np.float64))
        offs.append(offs[-1] + n)
    fo = {k: np.concatenate(v) for k, v in cols.items()}
    fo["tree_off"] = np.array(offs, np.int32)
    return fo


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    L = _lib.require_device()
    t0 = time.time()
    paths = {"rank": 0, "other": 0, "rank12": 0}
    for case in range(n_cases):
        F = int(rng.choice([1, 2, 9, 25, 49, 81, 121, 169, 192, 193, 225, 255, 256, 300, 529, 768, 900, 1023, 1024, 1100]))
        T = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 50, 100, 300]))
        depth = int(rng.choice([0, 1, 2, 5, 12, 20, 25, 40]))
        max_nodes = int(rng.choice([3, 31, 301, 2501, 9001]))
        comb = bool(rng.random() < 0.2)
        miss = float(rng.choice([0.0, 0.0, 0.3]))
        pool = None
        if rng.random() < 0.15:  # few distinct thresholds / many equal ones
            pool = rng.random(int(rng.choice([1, 3, 50])))
        if rng.random() < 0.07:  # > 2047 thresholds on a feature: the rank format must give way
            F, T, max_nodes, depth = int(rng.choice([1, 2])), 3, 9001, 40
        elif rng.random() < 0.12:  # 2 048 .. 4 095 thresholds on the features: one row each with 12-bit ranks, two with 11
            F, T, max_nodes, depth = int(rng.choice([2, 3, 5, 9])), int(rng.choice([4, 8, 16])), 2501, 25
            miss = float(rng.choice([0.0, 0.3, 0.5]))
        if T * max_nodes > 400000:
            T = max(1, 400000 // max_nodes)
        fo = random_forest(rng, F, T, max_nodes, depth, miss, pool, comb)
        N = int(rng.choice([1, 63, 64, 65, 127, 129, 255, 257, 1000, 5000, 70000]))
        if N * F > 3e7:
            N = max(1, int(3e7 // F))
        X = (rng.random((N, F)) * 2 - 0.5).astype(np.float32)
        inner = np.flatnonzero(fo["left"] != -1)
        if inner.size:
            for k in range(min(N, 200)):
                j = inner[int(rng.integers(0, inner.size))]
                t = np.float32(fo["thr"][j])
                X[k, fo["feat"][j]] = (t, np.nextafter(t, np.float32(-np.inf)), np.nextafter(t, np.float32(np.inf)))[k % 3]
        if N > 3:
            X[N // 2, :] = np.nan
            X[N // 3, int(rng.integers(0, F))] = np.nan
            X[N // 4, int(rng.integers(0, F))] = np.inf
            X[N // 5, int(rng.integers(0, F))] = -np.inf
        ref = onp.predict(fo, X)
        ff = FlatForest(F, *(fo[k] for k in FlatForest.FIELDS))
        opts = {}
        if rng.random() < 0.3:
            opts["forest_q_persist"] = int(rng.choice([0, -1, -3, 2]))
        if rng.random() < 0.2:
            opts["forest_slots"] = int(rng.choice([2, 3, 5, 8, 11, 16]))
        if rng.random() < 0.1:
            opts["forest_q"] = 0
        if rng.random() < 0.5:  # the 12-bit rank word: never / when it keeps a larger shape (default) / whenever it saves rows
            opts["forest_q_rank12"] = int(rng.choice([0, 2, 2]))
        if F > 255 and rng.random() < 0.4:  # the wide word: one tile or two per trip, walkers loading early or late
            opts[str(rng.choice(["forest_q_two", "forest_q_help"]))] = 0
        L.pk_prof_enable(1); L.pk_prof_reset()
        try:
            hf = _lib.HipForest(ff, options=opts)   # (options are this handle's own)
            p = hf.predict(X)
            q_mode = hf.get_option("stat_q_mode")
        except _lib.PeakachuHipError as e:
            L.pk_prof_enable(0)
            if F > 1023 or "unsupported" in str(e).lower() or "does not fit" in str(e).lower():
                print("case %3d F=%4d T=%3d nodes<=%4d depth=%2d N=%5d %s: refused (%s)" % (
                    case, F, T, max_nodes, depth, N, opts, str(e)[:70]))
                continue
            raise
        used_rank = _lib.prof_get("quant")[1] > 0
        L.pk_prof_enable(0)
        paths["rank" if used_rank else "other"] += 1
        if used_rank and q_mode == 2:
            paths["rank12"] += 1
        ok = np.array_equal(p.view(np.uint64), ref.view(np.uint64))
        print("case %3d F=%4d T=%3d nodes<=%4d depth=%2d comb=%d miss=%.1f N=%5d %s: %s %s" % (
            case, F, T, max_nodes, depth, comb, miss, N, opts, ("rank12" if q_mode == 2 else "rank") if used_rank else "float", "ok" if ok else "MISMATCH"))
        if not ok:
            bad = np.flatnonzero(p.view(np.uint64) != ref.view(np.uint64))
            print("   first mismatches:", bad[:5], p[bad[:5]], ref[bad[:5]])
            sys.exit(1)
        sys.stdout.flush()
    print("all %d cases bit-exact in %.0f s; kernel paths: %s" % (n_cases, time.time() - t0, paths))


if __name__ == "__main__":
    main()
