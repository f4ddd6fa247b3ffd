#!/usr/bin/env python
"""How much of the pinned parity rests on the ORDER of one sum?

`distance_normalize` tests `window[:w,:w].mean() > 0` and `centre / that mean > 0.1`
(peakachu/utils.py:228-232).  Under numba (production) `mean()` adds the w*w cells sequentially in C
order; the golden fixtures were made by the reference's own Python with `numba.njit` as the
identity, i.e. with numpy's PAIRWISE mean; the build follows numba's order.  On integer counts the
two sums are equal; on BALANCED (non-integer) maps they can differ in the last bit, and only the two
`>` decisions consume them.  This tool recomputes, for every candidate window of every committed
fixture with a non-integer matrix, both means and both decisions, and reports how many windows
differ in the mean's bits and how many would DECIDE differently (expected: 0) -- so that the
"unpinned on balanced maps" caveat of DESIGN.md 2 / INTEGRATION.md is a number, not a sentence.
Needs only tests/golden (any machine).  usage: tools/mean_order_check.py > profiles/r04_mean_order.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io as gio  # noqa: E402
from peakachu_amd import utils  # noqa: E402


def check(name, M, w, xs, ys):
    n = M.shape[0]
    D = np.asarray(M.todense(), np.float64)
    D[np.isnan(D)] = 0.0
    tot = bits = dec = 0
    near = np.inf
    for x, y in zip(xs, ys):
        if not (x - w >= 0 and y + w + 1 <= n):
            continue
        blk = D[x - w:x, y - w:y]
        seq = 0.0
        for v in blk.ravel():          # numba: sequential, C order
            seq += v
        seq /= w * w
        pw = blk.mean()                # numpy: pairwise
        c = D[x, y]
        tot += 1
        bits += seq != pw
        with np.errstate(divide="ignore", invalid="ignore"):
            d_seq = (seq > 0) and (c / seq > 0.1)
            d_pw = (pw > 0) and (c / pw > 0.1)
            if seq > 0:
                near = min(near, abs(c / seq - 0.1) / 0.1)
        dec += d_seq != d_pw
    print("%-28s w=%2d windows %7d  mean differs in its bits: %6d (%.1f %%)  DECISION differs: %d  "
          "closest p2LL to the 0.1 boundary (relative): %.3g" % (name, w, tot, bits, 100.0 * bits / max(tot, 1), dec, near))
    return dec


def main():
    total = 0
    z = gio.load("g1_extract_w5_balanced.npz")
    M = gio.balance(gio.sym_matrix(z, "R"), z["weights"])
    total += check("g1_extract_w5_balanced", M, int(z["w"]), z["x"], z["y"])
    for name in ("g3_score_weights.npz", "g3_score_hicstyle.npz"):
        z = gio.load(name)
        raw = gio.sym_matrix(z, "R")
        M = gio.balance(raw, z["weights"]) if str(z["mode"]) == "weights" else gio.hicstyle(raw, z["weights"])
        total += check(name[:-4], M, int(z["w"]), z["ridx"], z["cidx"])
    # a larger balanced map than any fixture: every band pixel of a synthetic chromosome
    from peakachu_amd import synth
    Ms, _ = synth.synth_band(3000, 120, seed=5)
    rng = np.random.default_rng(5)
    wts = 1.0 / np.sqrt(200.0 * rng.uniform(0.7, 1.3, Ms.shape[0]))
    B = gio.balance(Ms, wts)
    Bf = utils.band_filter(B, 5, 100)
    x, y = synth.all_band_pixels(Bf, 6, 100)
    total += check("synthetic balanced 3000 bins", B, 5, x[::3], y[::3])
    print("windows whose keep / drop decision depends on the order of the sum:", total)
    return 0 if total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
