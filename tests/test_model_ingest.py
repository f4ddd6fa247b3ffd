"""`-m/--model`: reading the reference's model files (peakachu/score_genome.py:14,
`joblib.load`) without scikit-learn, whatever version pickled them.

peakachu_amd.sk_pickle reads the pickle's arrays directly.  Checked here:
  * files written by the installed scikit-learn / joblib exactly as the reference
    writes them (`joblib.dump(model, path, compress=('xz', 3))`,
    peakachu/train_models.py:116), other compressors and plain pickle: the
    FlatForest equals FlatForest.from_sklearn of the live object, bit for bit;
  * a file in the layout of the README's pin (scikit-learn 1.1.2: node array without
    `missing_go_to_left`, `tree_.value` holding class COUNTS, `n_features_`):
    the same fractions come out as the version that wrote it would predict with;
  * the count -> fraction normalisation itself, on a count-valued value array.
CPU only.
"""
import io
import os
import pickle

import numpy as np
import pytest

from peakachu_amd import sk_pickle
from peakachu_amd.forest import FlatForest, class1_fraction, load_model

sklearn = pytest.importorskip("sklearn")
joblib = pytest.importorskip("joblib")


@pytest.fixture(scope="module")
def rf():
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(0)
    X = rng.random((600, 25)).astype(np.float32)
    y = ((X[:, 3] + X[:, 7] * X[:, 11]) > 0.8).astype(int)
    m = RandomForestClassifier(n_estimators=12, max_depth=9, max_features="sqrt", n_jobs=1,
                               random_state=0, class_weight="balanced")
    m.fit(X, y)
    return m


def _same(a, b):
    assert a.F == b.F and a.T == b.T
    for k in FlatForest.FIELDS:
        u, v = getattr(a, k), getattr(b, k)
        assert u.dtype == v.dtype and np.array_equal(u.view(np.uint8), v.view(np.uint8)), k


@pytest.mark.parametrize("compress", [("xz", 3), 0, 3, ("gzip", 3), ("bz2", 3), ("lzma", 3)])
def test_joblib_files_of_the_installed_sklearn(rf, tmp_path, compress):
    path = str(tmp_path / "model.pkl")
    joblib.dump(rf, path, compress=compress)
    _same(load_model(path), FlatForest.from_sklearn(rf))


def test_plain_pickle(rf, tmp_path):
    path = str(tmp_path / "model.pickle")
    with open(path, "wb") as fh:
        pickle.dump(rf, fh, protocol=4)
    _same(load_model(path), FlatForest.from_sklearn(rf))


def test_no_sklearn_code_is_touched(rf, tmp_path, monkeypatch):
    """The reader must not import sklearn classes: poison their constructors."""
    path = str(tmp_path / "model.pkl")
    joblib.dump(rf, path, compress=("xz", 3))
    import sklearn.ensemble
    import sklearn.tree

    def boom(*a, **k):
        raise AssertionError("sklearn was instantiated while reading the model file")
    monkeypatch.setattr(sklearn.ensemble.RandomForestClassifier, "__setstate__", boom, raising=False)
    monkeypatch.setattr(sklearn.tree.DecisionTreeClassifier, "__setstate__", boom, raising=False)
    ref = FlatForest.from_sklearn(rf)
    _same(load_model(path), ref)


class _Fake:
    """Pickles as `module.name` with a given state (the layout an old scikit-learn wrote)."""

    def __init__(self, module, name, args=(), state=None):
        self.module, self.name, self.args, self.state = module, name, args, state


class _OldPickler(pickle._Pickler):
    def save(self, obj, save_persistent_id=True):
        if isinstance(obj, _Fake):
            # GLOBAL module name, args tuple, REDUCE, state, BUILD
            self.write(pickle.GLOBAL + obj.module.encode() + b"\n" + obj.name.encode() + b"\n")
            pickle._Pickler.save(self, tuple(obj.args))
            self.write(pickle.REDUCE)
            self.memoize(obj)
            if obj.state is not None:
                pickle._Pickler.save(self, obj.state)
                self.write(pickle.BUILD)
            return
        pickle._Pickler.save(self, obj, save_persistent_id)


def _as_sklearn_1_1_2(rf):
    """The object graph scikit-learn 1.1.2 pickles for this forest: node records without
    `missing_go_to_left`, weighted class counts in `values`, `n_features_`."""
    old_dtype = np.dtype([("left_child", "<i8"), ("right_child", "<i8"), ("feature", "<i8"),
                          ("threshold", "<f8"), ("impurity", "<f8"), ("n_node_samples", "<i8"),
                          ("weighted_n_node_samples", "<f8")])
    ests = []
    for est in rf.estimators_:
        st = est.tree_.__getstate__()
        nodes = np.zeros(st["nodes"].shape[0], old_dtype)
        for f in old_dtype.names:
            nodes[f] = st["nodes"][f]
        counts = st["values"] * st["nodes"]["weighted_n_node_samples"][:, None, None]
        tree = _Fake("sklearn.tree._tree", "Tree",
                     (int(rf.n_features_in_), np.array([2], np.intp), 1),
                     dict(max_depth=int(st["max_depth"]), node_count=int(st["node_count"]),
                          nodes=nodes, values=counts))
        ests.append(_Fake("sklearn.tree._classes", "DecisionTreeClassifier", (),
                          dict(tree_=tree, n_features_=int(rf.n_features_in_), n_outputs_=1,
                               classes_=np.array([0, 1]), _sklearn_version="1.1.2")))
    top = _Fake("sklearn.ensemble._forest", "RandomForestClassifier", (),
                dict(estimators_=ests, n_features_=int(rf.n_features_in_), n_features_in_=int(rf.n_features_in_),
                     classes_=np.array([0, 1]), n_classes_=2, n_outputs_=1, _sklearn_version="1.1.2"))
    return top


def test_old_layout_counts_and_no_missing_field(rf, tmp_path):
    buf = io.BytesIO()
    _OldPickler(buf, protocol=2).dump(_as_sklearn_1_1_2(rf))
    path = str(tmp_path / "old.pkl")
    open(path, "wb").write(buf.getvalue())
    fa = sk_pickle.forest_arrays(path)
    assert fa["version"] == "1.1.2" and fa["F"] == 25 and len(fa["trees"]) == 12
    got = load_model(path)
    ref = FlatForest.from_sklearn(rf)
    assert (got.miss_left == 0).all()
    for k in ("tree_off", "left", "right", "feat", "thr"):
        assert np.array_equal(getattr(got, k), getattr(ref, k)), k
    # the fractions 1.1.2 would predict with: counts / row sum (its predict_proba)
    for t, est in zip(fa["trees"], rf.estimators_):
        st = est.tree_.__getstate__()
        counts = (st["values"] * st["nodes"]["weighted_n_node_samples"][:, None, None])[:, 0, :]
        assert np.array_equal(t["value"], counts)
    p_old = np.concatenate([class1_fraction(t["value"]) for t in fa["trees"]])
    assert np.array_equal(got.p1.view(np.uint64), p_old.view(np.uint64))
    assert np.abs(got.p1 - ref.p1).max() < 1e-15      # same fractions up to the last bit


def test_count_valued_values_normalise_bit_identically():
    """Unweighted forest: fractions k/n; counts k and n-k are exact, so normalising the
    counts must give the very same doubles scikit-learn stores as fractions."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(3)
    X = rng.random((500, 16)).astype(np.float32)
    y = (X[:, 0] + 0.3 * rng.standard_normal(500) > 0.5).astype(int)
    m = RandomForestClassifier(n_estimators=8, max_depth=7, n_jobs=1, random_state=1).fit(X, y)
    for est in m.estimators_:
        t = est.tree_
        frac = np.asarray(t.value)[:, 0, :]
        counts = np.rint(frac * t.weighted_n_node_samples[:, None])
        assert np.abs(counts - frac * t.weighted_n_node_samples[:, None]).max() < 1e-9
        assert np.array_equal(class1_fraction(counts).view(np.uint64),
                              np.ascontiguousarray(frac[:, 1]).view(np.uint64))
    # a node nobody reached (all-zero counts) keeps probability 0, as predict_proba's
    # `normalizer[normalizer == 0.0] = 1.0` does
    assert class1_fraction(np.array([[0.0, 0.0], [3.0, 1.0]])).tolist() == [0.0, 0.25]


def test_rejects_non_forests(tmp_path):
    path = str(tmp_path / "x.pkl")
    joblib.dump({"a": 1}, path)
    with pytest.raises(ValueError):
        load_model(path)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OLD_FILES = [("balanced", "old_sklearn_rf_balanced.xz.joblib"), ("plain", "old_sklearn_rf_plain.xz.joblib"),
             ("plain", "old_sklearn_rf_plain.raw.joblib")]


@pytest.mark.parametrize("tag,name", OLD_FILES)
def test_genuine_old_sklearn_pickles(tag, name):
    """Pickles written by scikit-learn 0.24.2 + joblib 1.1.0 themselves (tools/
    make_old_sklearn_fixture.py, run with this image's Anaconda interpreter): the node records
    have no `missing_go_to_left`, `tree_.value` holds (weighted) class COUNTS, the file is
    `joblib.dump(model, path, compress=('xz', 3))` as peakachu/train_models.py:116 writes it.
    The forest read here, walked by the oracle, must give THAT scikit-learn's own
    predict_proba[:, 1] bit for bit."""
    from oracle import oracle_np as onp
    z = np.load(os.path.join(GOLDEN, "old_sklearn_rf.npz"))
    assert list(z["versions"][:2]) == ["0.24.2", "1.1.0"]
    arr = sk_pickle.forest_arrays(os.path.join(GOLDEN, name))
    assert arr["version"] == "0.24.2" and arr["F"] == 121 and len(arr["trees"]) == 12
    assert arr["trees"][0]["value"].max() > 1.0  # counts
    ff = load_model(os.path.join(GOLDEN, name))
    p = onp.predict({k: getattr(ff, k) for k in FlatForest.FIELDS}, z["X"])
    assert np.array_equal(p.view(np.uint64), z["p_" + tag].view(np.uint64))


def _reduce_pickle(module, name, *args):
    """A protocol-2 pickle that calls module.name(*args) when loaded with pickle.load."""
    out = io.BytesIO()
    out.write(b"\x80\x02c" + module.encode() + b"\n" + name.encode() + b"\n")
    out.write(pickle.dumps(tuple(args), protocol=2)[2:-1])  # the argument tuple, without PROTO / STOP
    out.write(b"R.")
    return out.getvalue()


def test_malicious_pickles_execute_nothing(tmp_path):
    """The two ways a model file could run code: a global under `numpy.*` that
    executes its argument, and an object-dtype joblib array (a nested, formerly unrestricted
    pickle).  Neither may run; what they name becomes an inert placeholder."""
    marker = tmp_path / "executed"
    code = "open(%r, 'w').write('x')" % str(marker)
    # (1) numpy.testing._private.utils.runstring(code, {}) -- execs a string under pickle.load
    evil1 = _reduce_pickle("numpy.testing._private.utils", "runstring", code, {})
    # (2) builtins.exec through a nested pickle behind a NumpyArrayWrapper that says dtype 'O'
    evil_inner = _reduce_pickle("builtins", "exec", code)
    import joblib.numpy_pickle as jnp
    buf = io.BytesIO()
    wrapper = jnp.NumpyArrayWrapper(np.ndarray, (1,), "C", np.dtype("O"), allow_mmap=False)
    pickle.dump(wrapper, buf, protocol=2)
    evil2 = buf.getvalue() + evil_inner
    # sanity: the stock unpickler does run payload (1)
    pickle.loads(evil1)
    assert marker.exists()
    marker.unlink()
    for k, payload in enumerate((evil1, evil2, evil_inner)):
        path = tmp_path / ("evil%d.pkl" % k)
        path.write_bytes(payload)
        try:
            obj = sk_pickle.load(str(path))
        except Exception:
            obj = None  # refusing the file is as good as neutralising it
        assert not marker.exists(), "payload %d was executed" % k
        assert obj is None or isinstance(obj, sk_pickle._Placeholder)
        with pytest.raises(Exception):
            load_model(str(path))  # and it certainly is not a forest
        assert not marker.exists()


def test_only_reconstruction_helpers_of_numpy_are_allowed():
    up = sk_pickle._Unpickler(io.BytesIO(b""))
    assert up.find_class("numpy", "ndarray") is np.ndarray
    assert up.find_class("numpy", "dtype") is np.dtype
    for module, name in (("numpy.testing._private.utils", "runstring"), ("numpy", "load"),
                         ("numpy.lib.npyio", "load"), ("numpy.random._pickle", "__randomstate_ctor"),
                         ("os", "system"), ("builtins", "eval"), ("builtins", "exec")):
        cls = up.find_class(module, name)
        assert isinstance(cls, type) and issubclass(cls, sk_pickle._Placeholder), (module, name)
