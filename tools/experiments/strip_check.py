#!/usr/bin/env python3
"""GPU-box check + timing of the LDS-staged extractor (option extract_strip = batches per wave) against the
register-gather kernel: status and probability of EVERY candidate, bit for bit, on dense, gappy, shuffled
and edge-hugging lists; then the extract kernel's HIP-event time per step for a sweep of batches per wave."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from peakachu_amd import _lib, synth, utils  # noqa: E402
from peakachu_amd.forest import FlatForest  # noqa: E402


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def run(M, e, w, upper, fo, x, y, strip, reps=0):
    L = _lib.load()
    hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], e, -2 * w + 1, upper + 2 * w - 1,
                        options={"extract_strip": strip})
    hf = _lib.HipForest(fo)
    cd = _lib.HipCands(x, y)
    cd.run(hm, hf, w, 0.5)
    st, pr = cd.fetch_all()
    d = digest(st, pr.view(np.uint64))
    t = None
    if reps:
        cd.run(hm, hf, w, 0.5)
        L.pk_prof_enable(1); L.pk_prof_reset()
        for _ in range(reps):
            cd.run(hm, hf, w, 0.5)
        L.pk_prof_enable(0)
        t = {k: _lib.prof_get(k)[0] / reps for k in ("extract", "quant", "forest")}
    cd.close(); hf.close(); hm.close()
    return d, st, t


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    rng = np.random.default_rng(7)
    ok = True
    for w in (5, 6):
        fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % w))
        if mode in ("all", "check"):
            for n, band, upper in ((700, 80, 60), (3000, 120, 100), (2050, 40, 40)):
                M, _ = synth.synth_band(n, band, seed=n)
                upper = min(upper, n - 2 * w)
                e = utils.calculate_expected(M, upper + 2 * w, raw=True)
                Mf = utils.band_filter(M, w, upper)
                x, y = synth.all_band_pixels(Mf, 0, upper)      # from the main diagonal on: edge windows, d < 2w
                lists = {"all": (x, y), "every3rd": (x[::3].copy(), y[::3].copy()),
                         "every47th": (x[::47].copy(), y[::47].copy())}
                p = rng.permutation(x.size)
                lists["shuffled"] = (x[p].copy(), y[p].copy())
                # blocks of 100 consecutive candidates in random order: diagonal changes inside batches
                nblk = x.size // 100
                idx = (rng.permutation(nblk)[:, None] * 100 + np.arange(100)[None, :]).ravel()
                lists["blocks"] = (x[idx].copy(), y[idx].copy())
                lists["tail"] = (x[-77:].copy(), y[-77:].copy())
                for name, (lx, ly) in lists.items():
                    ref, st0, _ = run(Mf, e, w, upper, fo, lx, ly, 0)
                    for strip in (1,):
                        got, st1, _ = run(Mf, e, w, upper, fo, lx, ly, strip)
                        same = got == ref
                        ok &= same
                        print("w=%d n=%d %-9s N=%7d strip=%4d %s  (status: %s)" % (
                            w, n, name, lx.size, strip, "same" if same else "DIFFERENT",
                            np.bincount(st0, minlength=3).tolist()), flush=True)
        if mode in ("all", "time"):
            band = 200 if w == 5 else 300
            M, _ = synth.synth_band(30000, band, seed=0)
            upper = band
            e = utils.calculate_expected(M, upper + 2 * w, raw=True)
            Mf = utils.band_filter(M, w, upper)
            x, y = synth.all_band_pixels(Mf, 6, upper)
            ref, st0, t0 = run(Mf, e, w, upper, fo, x, y, 0, reps=10)
            print("w=%d config list N=%d  strip=0: extract %.3f ms quant %.3f forest %.3f" % (
                w, x.size, t0["extract"], t0["quant"], t0["forest"]), flush=True)
            for strip in (1,):
                got, st1, t1 = run(Mf, e, w, upper, fo, x, y, strip, reps=10)
                same = got == ref
                ok &= same
                print("w=%d strip=%3d: extract %.3f ms  %s" % (w, strip, t1["extract"], "same" if same else "DIFFERENT"),
                      flush=True)
    print("ALL SAME" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
