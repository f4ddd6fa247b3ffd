"""INTEGRATION.md section 3 shows the binding a maintainer of the reference would add to
peakachu/scoreUtils.py (raw ctypes, nothing of this package).  This test EXECUTES that block as it
stands in the document -- on a forest fitted by scikit-learn and the matrix of golden fixture G3 --
and compares with the package's own wrapper and with scikit-learn's predict_proba through the oracle
chain, so that the documented call sequence, argument order and forest conversion stay true."""
import os
import re

import numpy as np
import pytest

import golden_io as gio
from peakachu_amd import _lib, utils
from peakachu_amd.forest import FlatForest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def doc_block():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = txt[txt.index("## 3."):txt.index("## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert blocks and "pk_forest_from_sklearn" in blocks[0]
    return blocks[0]


def test_document_block_is_python():
    compile(doc_block(), "INTEGRATION.md#3", "exec")


@pytest.mark.gpu
def test_reference_side_binding_as_documented(hip_lib, monkeypatch):
    sk = pytest.importorskip("sklearn.ensemble")
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    x, y = z["ridx"].astype(np.int32), z["cidx"].astype(np.int32)
    # features of the fixture's own candidates (through the package) to fit a forest on
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], z["exp_arr"], -2 * w + 1, upper + 2 * w - 1)
    fea, _, keep = hm.extract(w, x, y)
    rng = np.random.default_rng(7)
    labels = (fea[:, fea.shape[1] // 2] + 0.2 * rng.standard_normal(fea.shape[0]) > np.median(fea[:, fea.shape[1] // 2])).astype(int)
    model = sk.RandomForestClassifier(n_estimators=30, max_depth=12, max_features="sqrt", n_jobs=1, random_state=0)
    model.fit(fea, labels)
    monkeypatch.setenv("PEAKACHU_HIP_LIB", _lib.LIB_PATH)
    ns = {}
    exec(compile(doc_block(), "INTEGRATION.md#3", "exec"), ns)
    forest = ns["pk_forest_from_sklearn"](model)
    matrix = ns["pk_matrix_from_scipy"](Mf, z["exp_arr"], w, upper)
    ri, ci, p, sig = ns["pk_score"](matrix, forest, w, 0.5, x, y)
    # (1) the package's wrapper on the package's conversion of the same model
    want = hm.score(_lib.HipForest(FlatForest.from_sklearn(model)), w, 0.5, x, y)
    assert ri.size == want[0].size and ri.size > 10
    assert np.array_equal(ri, want[0]) and np.array_equal(ci, want[1])
    assert np.array_equal(gio.bits(p), gio.bits(want[2])) and np.array_equal(gio.bits(sig), gio.bits(want[3]))
    # (2) scikit-learn itself on the same windows: the reference's line scoreUtils.py:109
    proba = model.predict_proba(fea)[:, 1]
    sel = proba > 0.5
    assert np.array_equal(np.asarray(x)[keep][sel], ri) and np.array_equal(np.asarray(y)[keep][sel], ci)
    assert np.array_equal(gio.bits(proba[sel]), gio.bits(p))
