"""Read a pickled scikit-learn RandomForestClassifier without scikit-learn.

The reference loads its model with `joblib.load(args.model)`
(peakachu/score_genome.py:14, peakachu/score_chromosome.py:14); the released
models were written by `joblib.dump(model, ..., compress=('xz', 3))`
(peakachu/train_models.py:116) under the README's pin scikit-learn 1.1.2 /
joblib 1.1.0 (README.md:19).  Unpickling such a file with a newer scikit-learn
is fragile: the Cython `Tree` of 1.3+ expects a node array with a
`missing_go_to_left` field, and 1.4+ stores class FRACTIONS in `tree_.value`
where older versions stored weighted class COUNTS (and normalised them in
`predict_proba`), so an old pickle that does load predicts from counts.

This module needs neither scikit-learn nor joblib: a pure-Python Unpickler
  * replaces every `sklearn.*` class by a placeholder that just keeps the
    constructor arguments and the state (`Tree.__reduce__` gives
    `(n_features, n_classes, n_outputs)` + `{'nodes', 'values', ...}`),
  * reads joblib's in-stream arrays (`NumpyArrayWrapper` followed by the raw
    array bytes, with or without joblib >= 1.2's alignment padding),
  * opens xz / gzip / bz2 / zlib / lzma / uncompressed files by their magic.
`forest_arrays()` then returns the per-tree node arrays `FlatForest` is made
of; `FlatForest.from_tree_states` applies the same count -> fraction
normalisation scikit-learn <= 1.3 applied at predict time.

Only the handful of numpy constructors listed in `_ALLOWED_EXACT` (ndarray, dtype and their
reconstruction helpers), a few builtin containers and the placeholders are ever
instantiated or called: every other global a file names -- anything else under `numpy.*`
included -- becomes an inert placeholder class, and object-dtype arrays (a nested pickle
in joblib's format) go through the same restricted unpickler.  Loading a model file
therefore executes no code from it beyond numpy's array / dtype reconstruction
(tests/test_model_ingest.py::test_malicious_pickles_execute_nothing).
"""
import bz2
import gzip
import io
import lzma
import pickle
import zlib

import numpy as np


class _Placeholder:
    """Stands in for any non-numpy class of the pickle."""
    _pk_module = _pk_name = "?"

    def __init__(self, *args, **kwargs):
        self._pk_args = args
        self._pk_state = None

    def __setstate__(self, state):
        self._pk_state = state

    # containers that are subclassed by pickled objects (e.g. a Bunch) keep working
    def __setitem__(self, k, v):
        self.__dict__.setdefault("_pk_items", {})[k] = v

    def append(self, v):
        self.__dict__.setdefault("_pk_list", []).append(v)

    def extend(self, vs):
        self.__dict__.setdefault("_pk_list", []).extend(vs)

    @property
    def state(self):
        st = self.__dict__.get("_pk_state")
        if st is None:  # plain objects: BUILD updated __dict__ directly
            return {k: v for k, v in self.__dict__.items() if not k.startswith("_pk_")}
        return st


class _ArrayWrapper(_Placeholder):
    """joblib.numpy_pickle.NumpyArrayWrapper: metadata in the pickle, bytes behind it."""

    def read(self, fh):
        st = self.state
        shape, order, dtype = st["shape"], st["order"], st["dtype"]
        count = int(np.prod([int(x) for x in shape], dtype=np.int64)) if len(shape) else 1
        if dtype.hasobject:
            # joblib pickles an object array as a nested pickle: same restrictions as outside
            # (forests do not hold such arrays; a parameter grid around one may)
            arr = _Unpickler(fh).load()
        else:
            if st.get("numpy_array_alignment_bytes") is not None:  # joblib >= 1.2
                pad = int.from_bytes(fh.read(1), "little")
                if pad:
                    fh.read(pad)
            need = count * dtype.itemsize
            buf = fh.read(need)
            if len(buf) != need:
                raise ValueError("model file truncated inside an array (%d of %d bytes)" % (len(buf), need))
            arr = np.frombuffer(buf, dtype=dtype, count=count).copy()
            if order == "F":
                arr = arr.reshape(shape[::-1]).transpose()
            else:
                arr = arr.reshape(shape)
        return arr


_ALLOWED_EXACT = {# numpy: array / dtype / scalar reconstruction only (old and new module paths)
                  ("numpy", "ndarray"), ("numpy", "dtype"),
                  ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                  ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                  ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
                  ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"),
                  ("builtins", "complex"), ("builtins", "range"), ("builtins", "bytearray"),
                  ("collections", "OrderedDict"), ("collections", "defaultdict"),
                  ("copyreg", "_reconstructor"), ("builtins", "object"), ("builtins", "int"),
                  ("builtins", "float"), ("builtins", "str"), ("builtins", "list"),
                  ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "bool"),
                  # protocol <= 2 spells bytes as _codecs.encode(text, 'latin1')
                  ("_codecs", "encode"), ("__builtin__", "object"), ("copy_reg", "_reconstructor")}


class _Unpickler(pickle._Unpickler):  # the pure-Python one: its dispatch table can be extended
    dispatch = pickle._Unpickler.dispatch.copy()

    def __init__(self, fh):
        super().__init__(fh)
        self._fh = fh
        self._classes = {}

    def find_class(self, module, name):
        if (module, name) in (("__builtin__", "object"), ("copy_reg", "_reconstructor")):  # Python 2 names
            module = {"__builtin__": "builtins", "copy_reg": "copyreg"}[module]
        if (module, name) in _ALLOWED_EXACT:
            return super().find_class(module, name)
        key = (module, name)
        if key not in self._classes:
            base = _ArrayWrapper if name in ("NumpyArrayWrapper", "NDArrayWrapper") else _Placeholder
            self._classes[key] = type(name, (base,), {"_pk_module": module, "_pk_name": name})
        return self._classes[key]

    def load_build(self):
        pickle._Unpickler.load_build(self)
        top = self.stack[-1]
        if isinstance(top, _ArrayWrapper):  # the array's bytes follow the wrapper in the stream
            self.stack[-1] = top.read(self._fh)

    dispatch[pickle.BUILD[0]] = load_build


def _open(path):
    raw = open(path, "rb").read()
    if raw[:6] == b"\xfd7zXZ\x00":
        return io.BytesIO(lzma.decompress(raw))
    if raw[:2] == b"\x1f\x8b":
        return io.BytesIO(gzip.decompress(raw))
    if raw[:3] == b"BZh":
        return io.BytesIO(bz2.decompress(raw))
    if raw[:1] == b"\x78" and len(raw) > 2 and ((raw[0] << 8) | raw[1]) % 31 == 0:
        return io.BytesIO(zlib.decompress(raw))
    if raw[:1] == b"\x5d":  # legacy .lzma
        return io.BytesIO(lzma.decompress(raw, format=lzma.FORMAT_ALONE))
    if raw[:10].startswith(b"ZF"):
        raise ValueError("joblib's pre-0.10 'ZF' container is not supported; re-save the model")
    return io.BytesIO(raw)


def load(path):
    """The unpickled object graph (placeholders + numpy arrays)."""
    return _Unpickler(_open(path)).load()


def _find_forest(obj, depth=0):
    if isinstance(obj, _Placeholder) and "estimators_" in (obj.state if isinstance(obj.state, dict) else {}):
        return obj
    if depth < 3 and isinstance(obj, _Placeholder) and isinstance(obj.state, dict):
        for v in obj.state.values():  # e.g. a GridSearchCV / Pipeline around the forest
            f = _find_forest(v, depth + 1)
            if f is not None:
                return f
    if depth < 3 and isinstance(obj, (list, tuple)):
        for v in obj:
            f = _find_forest(v, depth + 1)
            if f is not None:
                return f
    return None


def forest_arrays(path):
    """-> dict(F, version, trees=[dict(left, right, feature, threshold, missing_go_to_left,
    value [nodes, classes])]) from a pickled RandomForestClassifier."""
    rf = _find_forest(load(path))
    if rf is None:
        raise ValueError("%s does not hold a fitted scikit-learn forest (no estimators_)" % path)
    st = rf.state
    classes = np.asarray(st.get("classes_", [0, 1]))
    if classes.size != 2:
        raise ValueError("the scoring path needs a 2-class forest, this one has %d classes" % classes.size)
    trees = []
    for est in st["estimators_"]:
        t = est.state["tree_"]
        ts = t.state
        nodes = ts["nodes"]
        names = nodes.dtype.names
        k = int(ts.get("node_count", nodes.shape[0]))
        values = np.asarray(ts["values"], np.float64)[:k]
        if values.ndim != 3 or values.shape[1] != 1 or values.shape[2] < 2:
            raise ValueError("unexpected tree value array of shape %r" % (values.shape,))
        trees.append(dict(
            left=np.asarray(nodes["left_child"][:k], np.int32),
            right=np.asarray(nodes["right_child"][:k], np.int32),
            feature=np.asarray(nodes["feature"][:k], np.int32),
            threshold=np.asarray(nodes["threshold"][:k], np.float64),
            # scikit-learn < 1.3 has no such field: NaN features were an error there
            missing_go_to_left=(np.asarray(nodes["missing_go_to_left"][:k], np.uint8)
                                if "missing_go_to_left" in names else np.zeros(k, np.uint8)),
            value=values[:, 0, :2]))
    F = st.get("n_features_in_", st.get("n_features_"))
    if F is None and trees:
        F = st["estimators_"][0].state["tree_"]._pk_args[0]
    return dict(F=int(F), version=st.get("_sklearn_version"), trees=trees)
