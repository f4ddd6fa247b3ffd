#!/usr/bin/env python
"""Model-file ingestion fuzz, two stages (build container only):
  1. /opt/conda/bin/python3.9 tests/fuzz/fuzz_old_sklearn.py make <dir> [n]   -- scikit-learn 0.24.2 +
     joblib 1.1.0 fit n random forests (criteria, class weights, depths, leaf sizes, bootstrap
     on/off, few or many features) and dump each with a random joblib compression / pickle
     protocol, next to that scikit-learn's own predict_proba[:, 1] on random inputs;
  2. python tests/fuzz/fuzz_old_sklearn.py check <dir>   -- the system interpreter (no access to
     that scikit-learn) reads every file with peakachu_amd.sk_pickle and walks it with the
     oracle: bit-identical probabilities required."""
import os, sys
import numpy as np


def make(out, n):
    import joblib, pickle, sklearn
    from sklearn.ensemble import RandomForestClassifier
    os.makedirs(out, exist_ok=True)
    rng = np.random.RandomState(12345)
    for k in range(n):
        F = int(rng.choice([9, 25, 49, 121, 169, 529]))
        ns = int(rng.choice([60, 300, 1200]))
        X = rng.rand(ns, F)
        y = ((X[:, F // 2] + 0.3 * X[:, 0] > 0.7) | (rng.rand(ns) < 0.1)).astype(int)
        if y.min() == y.max():
            y[0] = 1 - y[0]
        kw = dict(n_estimators=int(rng.choice([1, 3, 10, 40])), criterion=str(rng.choice(["gini", "entropy"])),
                  max_depth=[None, 3, 8, 20][rng.randint(4)], min_samples_leaf=int(rng.choice([1, 1, 3, 10])),
                  class_weight=[None, "balanced", "balanced_subsample", {0: 1.0, 1: 3.5}][rng.randint(4)],
                  bootstrap=bool(rng.rand() < 0.8), max_features=["sqrt", "log2", None][rng.randint(3)],
                  n_jobs=1, random_state=int(rng.randint(1 << 30)))
        rf = RandomForestClassifier(**kw).fit(X, y)
        Xt = rng.rand(257, F).astype(np.float32)
        Xt[::6] = (Xt[::6] > 0.5)
        t = rf.estimators_[0].tree_
        inner = np.flatnonzero(t.children_left != -1)[:30]
        for j, node in enumerate(inner):
            Xt[j, t.feature[node]] = np.float32(t.threshold[node])
        p = rf.predict_proba(Xt)[:, 1]
        how = rng.randint(6)
        path = os.path.join(out, "m%03d" % k)
        if how == 0:
            joblib.dump(rf, path + ".joblib", compress=("xz", 3))
        elif how == 1:
            joblib.dump(rf, path + ".joblib", compress=int(rng.choice([1, 3, 9])))
        elif how == 2:
            joblib.dump(rf, path + ".joblib", compress=(str(rng.choice(["gzip", "bz2", "lzma", "zlib"])), 3))
        elif how == 3:
            joblib.dump(rf, path + ".joblib")
        elif how == 4:
            joblib.dump(rf, path + ".joblib", protocol=int(rng.choice([2, 3, 4])))
        else:
            with open(path + ".joblib", "wb") as fh:
                pickle.dump(rf, fh, protocol=int(rng.choice([2, 3, 4, 5])))
        np.savez_compressed(path + ".npz", X=Xt, p=p, F=F, how=how)
    print("wrote %d models with scikit-learn %s / joblib %s" % (n, sklearn.__version__, joblib.__version__))


def check(out):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import oracle_np as onp
    from peakachu_amd.forest import FlatForest, load_model
    names = sorted(f[:-4] for f in os.listdir(out) if f.endswith(".npz"))
    nodes = 0
    for name in names:
        z = np.load(os.path.join(out, name + ".npz"))
        ff = load_model(os.path.join(out, name + ".joblib"))
        assert ff.F == int(z["F"])
        p = onp.predict({k: getattr(ff, k) for k in FlatForest.FIELDS}, z["X"])
        ok = np.array_equal(p.view(np.uint64), z["p"].view(np.uint64))
        nodes += int(ff.n_nodes)
        print("%s how=%d F=%3d trees=%2d nodes=%6d: %s" % (name, int(z["how"]), ff.F, ff.T, int(ff.n_nodes), "ok" if ok else "MISMATCH"))
        if not ok:
            sys.exit(1)
    print("all %d model files read without scikit-learn / joblib and bit-identical to their scikit-learn's "
          "predict_proba (%d nodes)" % (len(names), nodes))


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    else:
        check(sys.argv[2])
