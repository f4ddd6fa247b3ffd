#!/opt/conda/bin/python3.9
"""tests/golden/gauss_scipy171.npz: scipy.ndimage.gaussian_filter(sigma=1) of random windows
as computed by scipy 1.7.1 (this image's Anaconda interpreter) -- the blur of
peakachu/utils.py:211-237 under a scipy of the reference's own era, next to the 1.15.3
the other fixtures were made with.   /opt/conda/bin/python3.9 tools/make_scipy171_fixture.py"""
import os
import numpy as np
import scipy
from scipy.ndimage import gaussian_filter

rng = np.random.RandomState(99)
out = {}
for w in (5, 6, 11, 2):
    S = 2 * w + 1
    wins = rng.rand(12, S, S) * rng.choice([1.0, 1e-3, 1e4], (12, 1, 1))
    wins[::5] *= rng.rand(*wins[::5].shape) < 0.3  # sparse windows
    out["in_w%d" % w] = wins
    out["out_w%d" % w] = np.stack([gaussian_filter(a, sigma=1, order=0) for a in wins])
import scipy.ndimage.filters as _F
out["taps"] = _F._gaussian_kernel1d(1.0, 0, 4)  # numpy's exp decides their last bits
out["version"] = np.array([scipy.__version__, np.__version__])
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gauss_scipy171.npz")
np.savez_compressed(p, **out)
print(scipy.__version__, os.path.getsize(p))
