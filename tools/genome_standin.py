#!/usr/bin/env python
"""hg19-shaped synthetic stand-in for BASELINE.json configs[0] / configs[2]
(`score_chromosome chr21` / `score_genome` on the GM12878 10 kb .cool, README.md:57,127 -- a
map and models that cannot be had offline).

What is real: the 25 chromosomes of hg19 at their real 10 kb bin counts (303 641 bins in the 23
that `-C '#' X` selects), the container (a .cool written by the genuine HDF5 library in
cooler's layout, tools/write_cool_h5py.py), the command lines, the released models' window
(w = 6).  What is synthetic: the counts (synth_band's law per chromosome, planted loops), the
balancing weights (with NaNs), the forest (fitted on such maps, peakachu_amd/data).

  synthesize(work, ...)     per-chromosome pixel tables + weights + manifest
  write_cool(work, out)     the .cool, through /opt/conda/bin/python3.9 (h5py)
  oracle_bedpe(work, ...)   the reference's chain on the CPU, independent of the reader:
                            matrices straight from the synthetic counts -> utils.calculate_expected
                            -> band_filter -> candidates (scipy) -> oracle.score -> write_bedpe
  `python tools/genome_standin.py e2e [--band B] [--work DIR] [--keep]`: the timed end-to-end
  run (GPU box): file size, CLI wall time and stage split for score_genome raw / weight and
  score_chromosome chr21, the CPU chain beside it, bedpe bytes compared.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peakachu_amd import synth  # noqa: E402

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def host_cores():
    import bench
    return bench.host_cores()


def have_h5py_writer():
    return os.path.exists(H5PY_PYTHON)


def synthesize(work, binsize=10000, band=320, seed=0, chroms=synth.HG19_CHROMS, n_nan=5, trans=0):
    """Pixel tables (bin1, bin2, count sorted as a .cool holds them), weights and a manifest.
    trans: pixels per bin whose second bin lies on a LATER chromosome (a real map's rows carry
    them behind their cis pixels; the scoring path reads past them)."""
    os.makedirs(work, exist_ok=True)
    table, off = [], 0
    total = sum((int(length) + binsize - 1) // binsize for _, length in chroms)
    for i, (name, length) in enumerate(chroms):
        n = (int(length) + binsize - 1) // binsize
        cnt = synth.band_counts(n, band, seed=seed + i)
        b1, b2, c = synth.band_counts_to_pixels(cnt, off)
        if trans and off + n < total:
            rng = np.random.default_rng(seed + 1000 + i)
            t1 = np.repeat(np.arange(off, off + n, dtype=np.int64), trans)
            t2 = rng.integers(off + n, total, size=t1.size).astype(np.int64)
            key = np.unique(t1 * total + t2)
            t1, t2 = key // total, key % total
            b1, b2, c = np.concatenate([b1, t1]), np.concatenate([b2, t2]), np.concatenate([c, np.ones(t1.size, c.dtype)])
            o = np.lexsort((b2, b1))
            b1, b2, c = b1[o], b2[o], c[o]
        np.savez(os.path.join(work, "pixels_%d.npz" % i), bin1=b1, bin2=b2, count=c)
        np.save(os.path.join(work, "weights_%d.npy" % i), synth.synth_weights(n, seed + i, n_nan=min(n_nan, n // 4)))
        table.append(dict(name=name, length=int(length), bins=n, offset=off, pixels=int(b1.size)))
        off += n
    man = dict(binsize=binsize, band=band, seed=seed, chroms=table)
    json.dump(man, open(os.path.join(work, "manifest.json"), "w"))
    return man


def write_cool(work, out, level=6, chunk=1 << 20):
    subprocess.check_call([H5PY_PYTHON, os.path.join(ROOT, "tools", "write_cool_h5py.py"), work, out,
                           str(level), str(chunk)])


def write_pkmap(man, work, out):
    """The same genome in the package's own container (boxes without an h5py interpreter)."""
    from peakachu_amd import io as pkio
    chroms = {}
    for i, c in enumerate(man["chroms"]):
        raw = synth.band_counts_to_csr(synth.band_counts(c["bins"], man["band"], seed=man["seed"] + i))
        chroms[c["name"]] = (raw, np.load(os.path.join(work, "weights_%d.npy" % i)))
    pkio.write_pkmap(out, chroms, resolution=man["binsize"])


def chrom_inputs(man, work, i, wname):
    """(M, raw_M, weights) of chromosome i as cooler would hand them to the driver
    (peakachu/score_genome.py:55-57), made from the synthetic counts WITHOUT the .cool reader:
    raw = the symmetric count matrix, M = bias[row] * bias[col] * count (cooler's api.matrix
    multiplies the two biases first), NaN where a weight is NaN."""
    c = man["chroms"][i]
    raw = synth.band_counts_to_csr(synth.band_counts(c["bins"], man["band"], seed=man["seed"] + i))
    if wname == "raw":
        return raw, raw, None
    w = np.load(os.path.join(work, "weights_%d.npy" % i))
    f = np.repeat(w, np.diff(raw.indptr))
    f = f * w[raw.indices]
    f = f * raw.data
    M = raw.copy()
    M.data = f
    return M, raw, w


def selected(man, chroms=("#", "X")):
    """peakachu/score_genome.py:39-44 on the manifest's names."""
    out = []
    for i, c in enumerate(man["chroms"]):
        label = c["name"].lstrip("chr")
        if (not chroms) or (label.isdigit() and "#" in chroms) or (label in chroms):
            out.append(i)
    return out


def oracle_bedpe(man, work, model_path, wname, lower, upper, thre, out, only=None, threads=None, log=None):
    """The reference's per-chromosome chain on the CPU (peakachu/score_genome.py:46-84,
    scoreUtils.py:10-68,95-135) with the oracle as the scorer.  Returns seconds per stage."""
    from scipy import sparse
    from oracle import oracle_np as onp
    from peakachu_amd import scoreUtils, utils
    from peakachu_amd.forest import FlatForest, load_model
    fo = load_model(model_path)
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    w = fo.width
    if os.path.exists(out):
        os.remove(out)
    T = dict(matrices=0.0, expected=0.0, band_filter=0.0, candidates=0.0, score=0.0, bedpe=0.0)
    n_cand = 0
    for i in (selected(man) if only is None else only):
        c = man["chroms"][i]
        t = time.perf_counter()
        M, raw, weights = chrom_inputs(man, work, i, wname)
        T["matrices"] += time.perf_counter() - t
        n = c["bins"]
        lo, up = max(lower, w + 1), min(upper, n - 2 * w)
        t = time.perf_counter()
        e = utils.calculate_expected(M, up + 2 * w, raw=weights is None)
        T["expected"] += time.perf_counter() - t
        t = time.perf_counter()
        Mf = utils.band_filter(M, w, up)
        T["band_filter"] += time.perf_counter() - t
        t = time.perf_counter()
        rx, ry = utils.candidates(raw, e, weights, lo, up)
        T["candidates"] += time.perf_counter() - t
        n_cand += len(rx)
        t = time.perf_counter()
        px, py, pp, ps = onp.score(Mf, e, w, fod, thre, np.asarray(rx, np.int32), np.asarray(ry, np.int32),
                                   batch=100000, threads=threads or host_cores())
        T["score"] += time.perf_counter() - t
        t = time.perf_counter()
        prob = sparse.csr_matrix((pp, (px, py)), shape=(n, n))
        sig = sparse.csr_matrix((ps, (px, py)), shape=(n, n))
        scoreUtils.write_bedpe(out, c["name"], man["binsize"], prob, sig)
        T["bedpe"] += time.perf_counter() - t
        if log:
            log("  oracle chain %-6s %6d bins %8d candidates %6d pixels" % (c["name"], n, len(rx), px.size))
    T["candidates_total"] = n_cand
    return T


def run_cli(argv, report=None, **extra_env):
    """The product's command line in a FRESH process (as a user starts it); returns wall seconds."""
    env = dict(os.environ, PK_NO_SPAWN="1", **extra_env)
    if report:
        env.update(PK_STAGE_TIMES="1", PK_STAGE_REPORT=report)
    t0 = time.perf_counter()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "peakachu-amd")] + argv, env=env,
                          stdout=subprocess.DEVNULL)
    return time.perf_counter() - t0


def e2e(a):
    work = a.work
    cool = os.path.join(work, "standin.cool")
    say = lambda s: (print(s), sys.stdout.flush())
    t0 = time.perf_counter()
    man = synthesize(work, band=a.band, seed=a.seed, trans=a.trans)
    sel = selected(man)
    say("synthesised hg19-shaped genome: %d chromosomes (%d selected by -C '#' X: %d bins), band %d bins, "
        "%d pixels, %.0f s" % (len(man["chroms"]), len(sel), sum(man["chroms"][i]["bins"] for i in sel), a.band,
                                sum(c["pixels"] for c in man["chroms"]), time.perf_counter() - t0))
    t0 = time.perf_counter()
    write_cool(work, cool, level=a.gzip, chunk=a.chunk)
    say(".cool written by h5py in %.0f s: %.0f MB" % (time.perf_counter() - t0, os.path.getsize(cool) / 1e6))
    model = os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % a.width)
    legs = [("score_genome raw", ["score_genome", "-p", cool, "-m", model, "--clr-weight-name", "raw"], "raw", None),
            ("score_genome weight", ["score_genome", "-p", cool, "-m", model, "--clr-weight-name", "weight"],
             "weight", None),
            ("score_chromosome chr21 weight", ["score_chromosome", "-p", cool, "-m", model, "-C", "chr21"], "weight",
             [i for i, c in enumerate(man["chroms"]) if c["name"] == "chr21"])]
    ok_all = True
    for name, argv, wname, only in legs:
        out = os.path.join(work, name.replace(" ", "_") + ".bedpe")
        rep = out + ".stages.json"
        walls = []
        for k in range(a.repeats):
            walls.append(run_cli(argv + ["-O", out, "-u", str(a.upper)], report=rep))
        say("== %s: wall %s s (fresh process each; the first reads the file cold)"
            % (name, " / ".join("%.2f" % v for v in walls)))
        st = json.load(open(rep))
        tot = sum(v[0] for k, v in st.items() if "[reader thread]" not in k)
        for k, (s, c) in sorted(st.items(), key=lambda kv: -kv[1][0]):
            say("   %8.3f s  %5d x  %s" % (s, c, k))
        say("   (main-thread stages sum to %.2f s of the last run's %.2f s; the rest is interpreter start, imports, "
            "argument parsing)" % (tot, walls[-1]))
        ref = out + ".oracle"
        t0 = time.perf_counter()
        T = oracle_bedpe(man, work, model, wname, 6, a.upper, 0.5, ref, only=only)
        cpu = time.perf_counter() - t0
        same = open(out, "rb").read() == open(ref, "rb").read()
        ok_all &= same
        say("   CPU chain (%d threads for the oracle's scoring, the rest one thread): %.1f s -- %s"
            % (host_cores(), cpu, ", ".join("%s %.1f" % (k, v) for k, v in T.items() if k != "candidates_total")))
        say("   %d candidates, %d bedpe lines, bedpe bytes %s the oracle chain's"
            % (T["candidates_total"], sum(1 for _ in open(out)), "EQUAL" if same else "DIFFER FROM"))
    if a.prefetch_sweep:
        out = os.path.join(work, "sweep.bedpe")
        for depth in (1, 2, 3, 4, 6):
            walls = [run_cli(legs[0][1] + ["-O", out, "-u", str(a.upper)], PK_PREFETCH=str(depth)) for _ in range(2)]
            say("PK_PREFETCH=%d: score_genome raw wall %s s" % (depth, " / ".join("%.2f" % v for v in walls)))
    if not a.keep:
        for f in os.listdir(work):
            os.remove(os.path.join(work, f))
    return 0 if ok_all else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["e2e"])
    ap.add_argument("--work", default="/tmp/pk_genome_standin")
    ap.add_argument("--band", type=int, default=320)
    ap.add_argument("--upper", type=int, default=300)
    ap.add_argument("--width", type=int, default=6)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gzip", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=1 << 20)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--trans", type=int, default=0, help="trans pixels per bin")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--prefetch-sweep", action="store_true", help="also time score_genome raw for reader-thread counts 1..6")
    sys.exit(e2e(ap.parse_args()))
