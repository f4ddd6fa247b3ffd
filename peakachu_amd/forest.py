"""Flat forest: the model object of the scoring path without sklearn.

The reference loads a pickled sklearn RandomForestClassifier
(peakachu/score_genome.py:14) and only ever uses `predict_proba(X)[:, 1]`
(peakachu/scoreUtils.py:109) and `feature_importances_.size`
(peakachu/score_genome.py:23).  FlatForest carries exactly the arrays the
C ABI takes (include/peakachu_hip.h: pk_forest_create) -- sklearn's per-tree
node arrays laid end to end -- and round-trips through a numpy-only .npz so
that scoring needs neither sklearn nor joblib on the GPU box.
"""
import numpy as np


class FlatForest:
    FIELDS = ("tree_off", "left", "right", "feat", "thr", "miss_left", "p1")

    def __init__(self, F, tree_off, left, right, feat, thr, miss_left, p1):
        self.F = int(F)
        self.tree_off = np.ascontiguousarray(tree_off, np.int32)
        self.left = np.ascontiguousarray(left, np.int32)
        self.right = np.ascontiguousarray(right, np.int32)
        self.feat = np.ascontiguousarray(feat, np.int32)
        self.thr = np.ascontiguousarray(thr, np.float64)
        self.miss_left = np.ascontiguousarray(miss_left, np.uint8)
        self.p1 = np.ascontiguousarray(p1, np.float64)
        self.T = int(self.tree_off.size - 1)
        n = int(self.tree_off[-1])
        for name in self.FIELDS[1:]:
            if getattr(self, name).size != n:
                raise ValueError("forest array %s has %d entries, expected %d"
                                 % (name, getattr(self, name).size, n))

    # the reference infers w from feature_importances_.size
    # (peakachu/score_genome.py:23); only the size is ever used
    @property
    def feature_importances_(self):
        return np.zeros(self.F)

    @property
    def width(self):
        return int((np.sqrt(self.F) - 1) / 2)

    @property
    def n_nodes(self):
        return int(self.tree_off[-1])

    @classmethod
    def from_sklearn(cls, rf):
        """RandomForestClassifier (2 classes) -> FlatForest."""
        if len(getattr(rf, "classes_", [0, 1])) != 2:
            raise ValueError("the scoring path needs a 2-class forest")
        offs, parts = [0], {k: [] for k in cls.FIELDS[1:]}
        for est in rf.estimators_:
            t = est.tree_
            k = int(t.node_count)
            parts["left"].append(np.asarray(t.children_left, np.int32))
            parts["right"].append(np.asarray(t.children_right, np.int32))
            parts["feat"].append(np.asarray(t.feature, np.int32))
            parts["thr"].append(np.asarray(t.threshold, np.float64))
            m = getattr(t, "missing_go_to_left", None)
            parts["miss_left"].append(np.zeros(k, np.uint8) if m is None
                                      else np.asarray(m, np.uint8))
            parts["p1"].append(class1_fraction(np.asarray(t.value)[:, 0, :]))
            offs.append(offs[-1] + k)
        F = int(getattr(rf, "n_features_in_", rf.feature_importances_.size))
        return cls(F, np.asarray(offs), *[np.concatenate(parts[k]) for k in cls.FIELDS[1:]])

    @classmethod
    def from_tree_states(cls, F, trees):
        """Per-tree node arrays as peakachu_amd.sk_pickle.forest_arrays reads them from a
        pickled forest of any scikit-learn version (no scikit-learn needed)."""
        offs, parts = [0], {k: [] for k in cls.FIELDS[1:]}
        for t in trees:
            parts["left"].append(np.asarray(t["left"], np.int32))
            parts["right"].append(np.asarray(t["right"], np.int32))
            parts["feat"].append(np.asarray(t["feature"], np.int32))
            parts["thr"].append(np.asarray(t["threshold"], np.float64))
            parts["miss_left"].append(np.asarray(t["missing_go_to_left"], np.uint8))
            parts["p1"].append(class1_fraction(t["value"]))
            offs.append(offs[-1] + int(parts["left"][-1].size))
        return cls(int(F), np.asarray(offs), *[np.concatenate(parts[k]) for k in cls.FIELDS[1:]])

    def save(self, path):
        np.savez_compressed(path, F=np.int32(self.F),
                            **{k: getattr(self, k) for k in self.FIELDS})

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        return cls(int(z["F"]), *[z[k] for k in cls.FIELDS])

    def stats(self):
        sizes = np.diff(self.tree_off)
        return dict(T=self.T, F=self.F, nodes=int(self.n_nodes),
                    nodes_per_tree_mean=float(sizes.mean()), nodes_per_tree_max=int(sizes.max()))


def class1_fraction(value):
    """value: [nodes, 2] of `tree_.value`.  scikit-learn >= 1.4 stores class fractions
    (rows sum to 1); older versions (the reference's README pins 1.1.2) store weighted
    class COUNTS and `DecisionTreeClassifier.predict_proba` divides by the row sum at
    predict time (`normalizer[normalizer == 0.0] = 1.0`).  The same division here, so an
    old pickle yields the fractions its own scikit-learn would have predicted with."""
    v = np.asarray(value, np.float64)
    s = v.sum(axis=1)
    if not np.allclose(s, 1.0):
        s = np.where(s == 0.0, 1.0, s)
        v = v / s[:, None]
    return np.ascontiguousarray(v[:, 1], np.float64)


def as_flat_forest(model):
    """Accept a FlatForest, a path to one, or an sklearn forest."""
    if isinstance(model, FlatForest):
        return model
    if isinstance(model, str):
        return load_model(model)
    if hasattr(model, "estimators_"):
        return FlatForest.from_sklearn(model)
    raise TypeError("unsupported model object %r" % type(model))


def load_model(path):
    """`-m/--model`: a flat-forest .npz, or the reference's joblib pickle of
    an sklearn RandomForestClassifier (peakachu/score_genome.py:14)."""
    if str(path).endswith(".npz"):
        return FlatForest.load(path)
    # read the pickle's arrays directly: works whatever scikit-learn / joblib wrote it
    # (and whether or not they are installed here); see sk_pickle.py
    from . import sk_pickle
    fa = sk_pickle.forest_arrays(path)
    return FlatForest.from_tree_states(fa["F"], fa["trees"])
