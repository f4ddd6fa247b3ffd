#!/opt/conda/bin/python3.9
"""tests/golden/isotonic_sk0242.npz: expected-value curves and what scikit-learn 0.24.2's
IsotonicRegression(increasing=False, out_of_bounds='clip') makes of them, exactly as
peakachu/utils.py:173-178 calls it.   /opt/conda/bin/python3.9 tools/make_isotonic_fixture.py"""
import os
import numpy as np
import sklearn
from sklearn.isotonic import IsotonicRegression

rng = np.random.RandomState(5)
ins, outs = [], []
for k in range(120):
    n = int(rng.randint(1, 340))
    kind = k % 6
    if kind == 0: e = rng.rand(n)
    elif kind == 1: e = np.sort(rng.rand(n))[::-1] * rng.choice([1, 1e-3, 1e4]) + rng.normal(0, 0.01, n)
    elif kind == 2: e = 200.0 / (1 + np.arange(n)) ** 0.9 * (1 + rng.normal(0, 0.05, n))
    elif kind == 3: e = np.round(rng.rand(n) * 5) / 5
    elif kind == 4: e = 100.0 / (1 + np.arange(n)) + rng.normal(0, 1.0, n)
    else: e = np.full(n, 3.0)
    e = np.abs(e)
    e[rng.rand(n) < rng.choice([0.0, 0.1, 0.6])] = 0.0
    if not (e > 0).any():
        e[rng.randint(0, n)] = 1.0
    IR = IsotonicRegression(increasing=False, out_of_bounds="clip")
    d = np.where(e > 0)[0]
    IR.fit(d, e[d])
    ins.append(e); outs.append(IR.predict(list(range(n))))
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "isotonic_sk0242.npz")
np.savez_compressed(p, n=np.array([a.size for a in ins]), x=np.concatenate(ins), y=np.concatenate(outs),
                    version=np.array(sklearn.__version__))
print(sklearn.__version__, os.path.getsize(p))
