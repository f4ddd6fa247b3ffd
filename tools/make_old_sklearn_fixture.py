#!/opt/conda/bin/python3.9
"""Writes tests/golden/old_sklearn_rf.* with an OLD scikit-learn (0.24.2) and joblib 1.1.0 --
the README's pin era (scikit-learn 1.1.2 / joblib 1.1.0, README.md:19): node records without
`missing_go_to_left`, class COUNTS in `tree_.value`, the model written exactly as
peakachu/train_models.py:116 writes it (`joblib.dump(model, path, compress=('xz', 3))`).

Run with the Anaconda interpreter of this image (the system python has scikit-learn 1.7):
    /opt/conda/bin/python3.9 tools/make_old_sklearn_fixture.py
The fixture = the pickles + inputs + that scikit-learn's own predict_proba[:, 1]."""
import os, sys
import numpy as np
import joblib, sklearn
from sklearn.ensemble import RandomForestClassifier

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rng = np.random.RandomState(20261004)
F = 121
n = 1500
X = rng.rand(n, F)
# a planted "loop" signal: bright centre relative to the lower-left corner
y = ((X[:, 60] > 0.55) & (X[:, 0:5].mean(1) < 0.6) | (rng.rand(n) < 0.05)).astype(int)
Xt = rng.rand(400, F).astype(np.float32)
Xt[::7] = (Xt[::7] > 0.5)
res, models = {}, {}
for tag, kw in (("balanced", dict(class_weight="balanced", criterion="gini", max_depth=12)),
                ("plain", dict(class_weight=None, criterion="entropy", max_depth=None, min_samples_leaf=3))):
    rf = RandomForestClassifier(n_estimators=12, max_features="sqrt", n_jobs=1, random_state=7, **kw)
    rf.fit(X, y)
    models[tag] = rf
    # thresholds that test features hit exactly (rows 0-39 for the first model, 40-79 for the second)
    t = rf.estimators_[0].tree_
    inner = np.flatnonzero(t.children_left != -1)[:40]
    for k, node in enumerate(inner):
        Xt[k + 40 * (len(models) - 1), t.feature[node]] = np.float32(t.threshold[node])
for tag, rf in models.items():
    res["p_" + tag] = rf.predict_proba(Xt)[:, 1]
    joblib.dump(rf, os.path.join(out, "old_sklearn_rf_%s.xz.joblib" % tag), compress=("xz", 3))
    v = rf.estimators_[0].tree_.value
    assert v.max() > 1.0, "expected class counts, not fractions"
    assert "missing_go_to_left" not in rf.estimators_[0].tree_.__getstate__()["nodes"].dtype.names
joblib.dump(rf, os.path.join(out, "old_sklearn_rf_plain.raw.joblib"))  # uncompressed: in-stream arrays
np.savez_compressed(os.path.join(out, "old_sklearn_rf.npz"), X=Xt, versions=np.array(
    [sklearn.__version__, joblib.__version__, np.__version__]), **res)
print("sklearn", sklearn.__version__, "joblib", joblib.__version__, "numpy", np.__version__)
for f in sorted(os.listdir(out)):
    if f.startswith("old_sklearn"):
        print(f, os.path.getsize(os.path.join(out, f)))
