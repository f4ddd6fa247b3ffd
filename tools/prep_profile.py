#!/usr/bin/env python
"""GPU-box helper: wall time of every library call of one chromosome's preparation
(Chromosome.__init__ of a chr1-sized raw map), call by call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peakachu_amd import _lib, synth, utils

n, band, w, upper = int(sys.argv[1]) if len(sys.argv) > 1 else 24926, 320, 6, 300
M = synth.band_counts_to_csr(synth.band_counts(n, band, seed=1))
L = _lib.require_device()
T = {}


def tick(name, fn):
    t0 = time.perf_counter()
    r = fn()
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return r


for rep in range(4):
    Mc = tick("canonical_csr", lambda: utils.canonical_csr(M))
    csr = tick("HipCsr (upload + facts)", lambda: _lib.HipCsr(Mc))
    dlo, dhi = -2 * w + 1, upper + 2 * w - 1
    bandm = tick("csr.band (scoring band)", lambda: csr.band(dlo, dhi))
    bandm.dlo, bandm.dhi = dlo, dhi
    means = tick("csr.expected_means 0..dhi", lambda: csr.expected_means(bandm, 0, dhi, False))
    extra = tick("csr.band (one more diagonal)", lambda: csr.band(dhi + 1, dhi + 1))
    m2 = tick("csr.expected_means last", lambda: csr.expected_means(extra, dhi + 1, dhi + 1, False))
    tick("extra.close", extra.close)
    e = np.concatenate([means, m2])
    exp_arr = tick("isotonic_expected", lambda: utils.isotonic_expected(e))
    tick("band.set_expected", lambda: bandm.set_expected(exp_arr))
    bg = np.ascontiguousarray(exp_arr[:upper + 1])
    kstar = tick("_poisson_count_thresholds", lambda: utils._poisson_count_thresholds(bg))
    cands, amb = tick("HipCands.from_band", lambda: _lib.HipCands.from_band(bandm, w + 1, upper, bg, kstar=kstar))
    xy = tick("cands.coords", cands.coords)
    tick("csr.close", csr.close)
    tick("cands.close", cands.close)
    tick("band.close", bandm.close)
print("%d bins, %d stored entries, %d candidates" % (n, M.nnz, xy[0].size))
for k, v in T.items():
    print("%-32s %s ms" % (k, "  ".join("%7.2f" % x for x in v)))
print("%-32s %s ms" % ("total", "  ".join("%7.2f" % sum(v[i] for v in T.values()) for i in range(4))))
