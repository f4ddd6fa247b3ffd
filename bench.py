#!/usr/bin/env python
"""Headline benchmark: candidate pixels scored per second on MI355X.

Workload (BASELINE.json configs[1]): synthetic 30 000 x 30 000 band-diagonal
contact matrix (200-bin = 2 Mb band at 10 kb), w = 5 (11x11 windows, 121
features), 100-tree Random Forest, every non-zero band pixel with
6 <= col-row <= 200 a candidate (about 5.6 M), threshold 0.5, reference batch
size 100 000.  One step = one pass of the hot path (extract -> forest ->
threshold/compact [-> RCCL gather of the scored pixels when N > 1]) over the
candidate list, with matrix, forest and candidates already resident in HBM.

N > 1 (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the launcher --
torch.distributed.run is the LAUNCHER only, a parent that never touches HIP): every
rank scores its own synthetic chromosome (weak scaling, the default; chromosomes
shard embarrassingly, peakachu/score_genome.py:46-84) or, with --scaling strong,
its batch-aligned block of ONE chromosome's candidate list (matrix and forest
replicated); rank 0 collects the scored pixels with one RCCL gather.  The ranks
themselves import no deep-learning framework: the unique-id broadcast, the
barriers and the max-over-ranks of the timings go through the product's own host
rendezvous (peakachu_amd.rendezvous, standard library), so that every point of a
scaling curve runs on the ROCm that `ldd libpeakachu_hip.so` names (a framework
imported first would bind its bundled librccl / libamdhip64 under the same
sonames instead).  The line says which runtime ran (`hip_runtime_version`,
`rccl_version`, `rocm_libs`); a run on any other copy ends with exit code 3, and
so does one whose RCCL communicator cannot be built.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def b_alg(F):
    """Algorithmic bytes per candidate: 8F window cells +
    4F feature write + 4F feature read + 8 coords + 8 prob."""
    return 16 * F + 16


def own_alg_bytes(kernel, F):
    """The share of B_alg that passes through one kernel class: the extractor reads
    8F of window cells and the coordinates and writes 4F of features; the forest reads those 4F
    and writes the probability; the rank quantizer re-encodes an intermediate and owns none."""
    return {"extract": 12 * F + 8, "forest": 4 * F + 8}.get(kernel)


KCLASSES = ("extract", "quant", "forest", "forest_tail", "compact")


def kernel_times(_lib):
    """The library's HIP-event times per kernel class: (ms, launches).  `forest` brackets the forest
    stage of a launch; when the forest is cut in two (pk_forest_q.hip) that is the head kernel, the gap
    and the tail kernel, and `forest_tail` is the tail kernel alone -- `forest_head` = the difference is
    what the dominant KERNEL took (rocprofv3's forest_qr_kernel<.., 1>)."""
    k = {c: _lib.prof_get(c) for c in KCLASSES}
    k["forest_head"] = (k["forest"][0] - k["forest_tail"][0], k["forest"][1])
    return k


def cut_info(hf, steps_since_reset=1):
    """How the last call's forest launches were cut (read-only options of the forest handle)."""
    g = int(hf.get_option("stat_split_group"))
    if g <= 0:
        return None
    return {"group": g, "of_groups": int(hf.get_option("stat_q_groups")), "trees_in_front": int(hf.get_option("stat_split_trees")),
            "parked_slots_last_call": int(hf.get_option("stat_split_parked")),
            "note": "the forest kernel runs in two launches: the head walks the trees in front of the cut over every "
                    "candidate and parks those whose sum plus 1.0 per remaining tree could still exceed thre x T "
                    "(partial sum, rank codes), the tail walks the rest over the parked ones only and continues "
                    "their sums in tree order; decided candidates are reported at probability 0 (pk_cands_set_prune: "
                    "the permission Chromosome.score gives); same scored pixels, bit for bit"}


def build_workload(seed, n, band, w, lower, upper, with_matrix=False):
    """(band-filtered matrix, expected curve, candidate coordinates, clamped upper[, the
    unfiltered matrix]) of one synthetic chromosome."""
    from peakachu_amd import synth, utils
    M, _ = synth.synth_band(n, band, seed=seed)
    upper = min(upper, n - 2 * w)
    exp_arr = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, max(lower, w + 1), upper)
    return (Mf, exp_arr, x, y, upper, M) if with_matrix else (Mf, exp_arr, x, y, upper)


def load_forest(spec, w, F):
    """The committed trained forest for this window size, a flat-forest file,
    or `random:T[:depth]`: seeded untrained random trees (stress configs for
    which no trained model is shipped; same node format and walk cost profile,
    but NOT a trained model -- named as such in the output)."""
    from peakachu_amd.forest import FlatForest
    if spec is None:
        return FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % w))
    if not spec.startswith("random:"):
        return FlatForest.load(spec)
    parts = spec.split(":")
    T = int(parts[1])
    depth = int(parts[2]) if len(parts) > 2 else 16
    rng = np.random.default_rng(12345)
    offs, cols = [0], {k: [] for k in ("left", "right", "feat", "thr", "miss_left", "p1")}
    for _ in range(T):
        # random binary tree grown from a random frontier to about 2 500 nodes
        left, right, feat, thr, p1, dep = [-1], [-1], [-2], [-2.0], [0.0], [0]
        frontier = [0]
        while frontier and len(left) < 2500:
            i = frontier.pop(int(rng.integers(0, len(frontier))))
            if dep[i] >= depth:
                continue
            feat[i] = int(rng.integers(0, F))
            thr[i] = float(rng.random())
            for side in (left, right):
                side[i] = len(left)
                left.append(-1); right.append(-1); feat.append(-2); thr.append(-2.0)
                p1.append(float(rng.integers(0, 2)) if rng.random() < 0.9 else float(rng.random()))
                dep.append(dep[i] + 1)
                frontier.append(len(left) - 1)
        k = len(left)
        cols["left"].append(np.array(left, np.int32)); cols["right"].append(np.array(right, np.int32))
        cols["feat"].append(np.array(feat, np.int32)); cols["thr"].append(np.array(thr, np.float64))
        cols["miss_left"].append(np.zeros(k, np.uint8)); cols["p1"].append(np.array(p1, np.float64))
        offs.append(offs[-1] + k)
    return FlatForest(F, np.array(offs, np.int32), *[np.concatenate(cols[k]) for k in
                                                     ("left", "right", "feat", "thr", "miss_left", "p1")])


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by
    the cgroup CPU quota (the GPU box gives one GPU's job a share of the host)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    cores = min(cores, max(1, q // per))
            break
        except Exception:
            continue
    env = os.environ.get("PK_BENCH_CPU_THREADS")
    if env:
        cores = int(env)
    return cores


def cpu_model():
    """The host CPU as /proc/cpuinfo names it (BASELINE.md 4 wants the model next to the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(Mf, exp_arr, w, fo, thre, x, y, batch, gpu_pixels=None, target_s=15.0, target_1t_s=5.0):
    """The CPU oracle (a port of the reference's algorithm, bit-exact against
    its golden vectors) timed on this box's host cores on a strided sample of
    the same candidate list.  When the sample is the whole list, its scored
    pixels are compared bit for bit with the GPU's (`pixels_equal`): the oracle
    acts as the checker here, the number it produces stays a reported baseline."""
    from oracle import oracle_np as onp
    from peakachu_amd.forest import FlatForest
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    cores = host_cores()
    N = x.size
    m, dt, xs = 50000, 0.0, x
    for _ in range(4):  # grow the sample until it costs about target_s of wall time
        stride = max(1, N // max(1, m))
        xs, ys = x[::stride], y[::stride]
        t0 = time.perf_counter()
        res = onp.score(Mf, exp_arr, w, fod, thre, xs, ys, batch=batch, threads=cores)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * target_s or xs.size >= N:
            break
        m = int(min(N, xs.size * min(10.0, 1.1 * target_s / max(dt, 1e-3))))
    out = dict(value=xs.size / dt, unit="candidates/s", cores=cores, kind="port", cpu_model=cpu_model(),
               sample="every %d-th candidate of the workload (%d of %d), %.1f s wall, "
                      "oracle/pk_oracle.c pko_score_mt with OpenMP over candidates"
                      % (stride, xs.size, N, dt))
    # BASELINE.md 4's `cpu-1t`: the same restatement on ONE thread (the reference is single-threaded:
    # peakachu/score_genome.py:46), on a smaller strided sample sized from the rate just measured
    m1 = int(max(2000, min(N, out["value"] / max(cores, 1) * target_1t_s)))
    s1 = max(1, N // m1)
    x1, y1 = x[::s1], y[::s1]
    t0 = time.perf_counter()
    onp.score(Mf, exp_arr, w, fod, thre, x1, y1, batch=batch, threads=1)
    d1 = time.perf_counter() - t0
    out["cpu_1t"] = dict(value=x1.size / d1, unit="candidates/s", cores=1,
                         sample="every %d-th candidate (%d of %d), %.1f s wall, one thread" % (s1, x1.size, N, d1))
    if gpu_pixels is not None and xs.size == N:
        # rows / columns as integers, probability and signal bit for bit
        out["pixels_equal"] = bool(
            res[0].size == gpu_pixels[0].size
            and np.array_equal(res[0], gpu_pixels[0].astype(np.int64))
            and np.array_equal(res[1], gpu_pixels[1].astype(np.int64))
            and np.array_equal(res[2].view(np.uint64), np.ascontiguousarray(gpu_pixels[2]).view(np.uint64))
            and np.array_equal(res[3].view(np.uint64), np.ascontiguousarray(gpu_pixels[3]).view(np.uint64)))
        out["pixels_compared"] = int(res[0].size)
    else:
        out["pixels_equal"] = None  # the sample did not grow to the whole list in the time budget
    return out


def source_sha():
    """Hash of the kernel sources' token stream -- comments and layout do not change it
    (tools/srchash.py; tools/make_traffic.py stamps profiles/pmc*.json with it)."""
    from tools.srchash import source_sha as sha
    return sha()


def pmc_rooflines(dom, launches_per_step_live, pmc_file="pmc.json"):
    """HBM traffic and the two binding rooflines of the dominant kernel from the committed
    PMC passes (profiles/pmc.json, made by tools/pmc.sh + tools/make_traffic.py).  Every
    figure can be recomputed from that file; `stale` tells whether the kernel sources
    have changed since it was measured."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    if not os.path.exists(path):
        return None, None, None
    try:
        P = json.load(open(path))
        d = P[dom]
        c = d["counters_per_launch"]
        cyc = d["cycles_per_launch"]
    except Exception:
        return None, None, None
    stale = P.get("source_sha") != source_sha()
    note = "profiles/%s (sources %s)" % (pmc_file, "CHANGED since: stale" if stale else "unchanged")
    traffic = None if stale else d.get("hbm_bytes_per_launch")
    issue = {"bound": "valu issue", "valu_wave_insts_per_launch": c.get("SQ_INSTS_VALU"),
             "cycles_per_inst": 4, "simds": 1024, "cycles_per_launch": cyc,
             "frac": (c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc)) if "SQ_INSTS_VALU" in c else None,
             "formula": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8)", "stale": stale,
             "source": note}
    # tools/micro/lds_indep.hip on MI355X (profiles/r03_lds_indep.log): INDEPENDENT ds_read_u16 /
    # b32 / b64 wave-instructions cost the CU 1.6-2.0 cycles each with >= 64 in flight (16 waves x
    # 4), as the guide's LDS table says (2 LDS-array cycles) -- not the 4 cycles EXPERIMENTS.md (round 2)
    # took from dependent chains.  The walk is a DEPENDENT chain (2 048 of them fit the LDS), so
    # the array-busy fraction is the roofline and the instruction count x 2 cycles its floor.
    lds = {"bound": "lds", "lds_wave_insts_per_launch": c.get("SQ_INSTS_LDS"),
           "lds_array_cycles_per_launch": c.get("SQ_LDS_IDX_ACTIVE"),
           "bank_conflict_cycles_per_launch": c.get("SQ_LDS_BANK_CONFLICT"), "cus": 256,
           "cycles_per_launch": cyc, "cycles_per_lds_inst": 2,
           "frac_issue_slots": (c["SQ_INSTS_LDS"] * 2.0 / (256.0 * cyc)) if "SQ_INSTS_LDS" in c else None,
           "frac_array_busy": (c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)) if "SQ_LDS_IDX_ACTIVE" in c else None,
           "formula": "SQ_INSTS_LDS x 2 cycles / (256 CUs x GRBM_GUI_ACTIVE/8) [conflict-free floor]; "
                      "SQ_LDS_IDX_ACTIVE / (256 x cycles) [array busy, bank conflicts included]",
           "cost_model_source": "tools/micro/lds_indep.hip, profiles/r03_lds_indep.log",
           "stale": stale, "source": note}
    return traffic, issue, lds


def pmc_path_traffic(kern, steps, pmc_file="pmc.json"):
    """HBM bytes of a whole STEP from the committed PMC passes: per kernel class (extract, quantizer, forest head,
    forest tail) the measured bytes per launch x the launches this run made per step.  None when the kernel
    sources changed since the passes (or a class was not measured)."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    try:
        P = json.load(open(path))
        if P.get("source_sha") != source_sha():
            return None
        total = 0.0
        for cls, pmc_cls in (("extract", "extract"), ("quant", "quant"), ("forest", "forest"), ("forest_tail", "forest_tail")):
            launches = kern[cls][1] / float(steps)
            if launches > 0:
                total += P[pmc_cls]["hbm_bytes_per_launch"] * launches
        return total
    except Exception:
        return None


def extra_config(L, dev, name, n, band, w, upper, forest_spec, thre, batch, steps, pmc_file=None):
    """One of the non-headline BASELINE.json configurations, measured the same way in the same
    process (device-resident candidates, pk_score_run, HIP-event kernel times), a few steps
    only: so that the driver's record carries them too, not only the builder's logs.  Its
    `roofline` object is the headline's, for this leg's dominant kernel: algorithmic bytes over
    the kernel's HIP-event time, `traffic` and the LDS / VALU fractions from the leg's own tracked
    PMC summary (profiles/<pmc_file>, tools/pmc_legs.sh), null when the kernels changed since."""
    from peakachu_amd import _lib
    F = (2 * w + 1) ** 2
    fo = load_forest(forest_spec, w, F)
    Mf, exp_arr, x, y, upper = build_workload(0, n, band, w, 6, upper)
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], exp_arr, -2 * w + 1, upper + 2 * w - 1,
                        device=dev)
    hf = _lib.HipForest(fo, device=dev)
    cd = _lib.HipCands(x, y, device=dev)
    cd.set_prune(True)   # as Chromosome.score runs its lists
    try:
        # (untimed calls first: the cut forest places its cut by what the calls themselves show -- a forest
        # that parks everybody, like the untrained random trees, is left uncut after a few of them)
        for _ in range(8):
            cd.run(hm, hf, w, thre, batch)
        L.pk_prof_enable(1)
        L.pk_prof_reset()
        _lib.check(L.pk_device_synchronize(dev), "sync")
        t0 = time.perf_counter()
        for _ in range(steps):
            n_out = cd.run(hm, hf, w, thre, batch)
        _lib.check(L.pk_device_synchronize(dev), "sync")
        el = time.perf_counter() - t0
        L.pk_prof_enable(0)
        kern = kernel_times(_lib)
        cut = cut_info(hf)
    finally:
        cd.close(); hf.close(); hm.close()
    value = x.size * steps / el
    dom = max(("extract", "quant", "forest"), key=lambda k: kern[k][0])
    dom_ms, dom_n = kern["forest_head" if dom == "forest" else dom]
    achieved = float(x.size) * steps * b_alg(F) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = issue = lds = None
    if pmc_file and dom_n:
        traffic, issue, lds = pmc_rooflines(dom, dom_n / steps, pmc_file)
    fst = fo.stats()
    return {"workload": "%s: synthetic %dx%d (%d-bin band), w=%d, %d-tree RF%s" % (
                name, n, n, band, w, fo.T, " (untrained random trees)" if (forest_spec or "").startswith("random:")
                else " (fitted: %.0f nodes per tree)" % fst["nodes_per_tree_mean"]),
            "value": value, "unit": "candidates/s", "steps": steps, "ms_per_step": el / steps * 1e3,
            "candidates": int(x.size), "scored_pixels": int(n_out), "early_exit_allowed": True, "forest_cut": cut,
            "kernel_ms_per_step": {k: v[0] / steps for k, v in kern.items()},
            # (as in the headline: `achieved` / `frac` = the whole path, the per-kernel recipe beside it)
            "roofline": {"bound": "hbm", "kernel": dom + (" (head of the cut forest)" if (cut and dom == "forest") else ""),
                         "achieved": value * b_alg(F) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": value * b_alg(F) / 1e9 / HBM_PEAK_GBS,
                         "frac_is": "whole path: candidates/s x B_alg / peak (SURVEY 8d)",
                         "dominant_kernel_achieved": achieved, "dominant_kernel_frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_path_traffic(kern, steps, pmc_file) if pmc_file else None,
                         "dominant_kernel_traffic": traffic,
                         "alg_bytes_per_candidate": b_alg(F),
                         "avg_launch_ms": dom_ms / dom_n if dom_n else None, "launches": dom_n,
                         "candidates_per_launch": float(x.size) * steps / dom_n if dom_n else None,
                         "lds_array_busy_frac": lds and lds.get("frac_array_busy"),
                         "lds_issue_slots_frac": lds and lds.get("frac_issue_slots"),
                         "valu_issue_frac": issue and issue.get("frac"),
                         "counters": (issue or {}).get("source") or ("no tracked PMC summary (profiles/%s)" % pmc_file)},
            "roofline_frac_dominant_kernel": achieved / HBM_PEAK_GBS,
            "whole_path_frac": value * b_alg(F) / 1e9 / HBM_PEAK_GBS}


def _timed_runs(L, dev, cd, hm, hf, w, thre, batch, reps):
    """`reps` warm pk_score_run calls on one resident list: wall microseconds per call
    (host clock around the loop) and the library's own HIP-event kernel times per call."""
    from peakachu_amd import _lib
    cd.run(hm, hf, w, thre, batch)
    cd.run(hm, hf, w, thre, batch)
    L.pk_prof_enable(1)
    L.pk_prof_reset()
    _lib.check(L.pk_device_synchronize(dev), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        n_out = cd.run(hm, hf, w, thre, batch)
    _lib.check(L.pk_device_synchronize(dev), "sync")
    el = time.perf_counter() - t0
    L.pk_prof_enable(0)
    kern = {k: _lib.prof_get(k)[0] / reps * 1e3 for k in KCLASSES}
    return el / reps * 1e6, kern, int(n_out)


def real_regime(L, dev, M, fo, w, lower, upper, thre, batch, x_all, y_all, hm, hf, strided=True):
    """The regime the CLI runs in (peakachu/scoreUtils.py:40-68, 95-106): the Poisson-filtered
    candidate list get_candidate makes -- a few percent of the band's non-zero pixels, scattered
    along the diagonals -- and short lists, scored by pk_score_run with the early exit
    Chromosome.score switches on.  Warm microseconds per call, candidates/s and the kernels' own
    HIP-event times; plus the cold cost of one Chromosome(...) + .score() of this matrix."""
    import io as _io
    from peakachu_amd import _lib, scoreUtils
    out = {"note": "pk_score_run with early exit (what Chromosome.score runs), device-resident lists; "
                   "us = warm wall time per call"}
    legs = []

    def leg(name, x, y, reps):
        cd = _lib.HipCands(x, y, device=dev)
        try:
            cd.set_prune(True)
            us, kern, n_out = _timed_runs(L, dev, cd, hm, hf, w, thre, batch, reps)
        finally:
            cd.close()
        legs.append({"list": name, "candidates": int(x.size), "us_per_call": us,
                     "value": x.size / (us * 1e-6), "unit": "candidates/s", "scored_pixels": n_out,
                     "kernel_us_per_call": kern, "whole_path_frac": x.size / (us * 1e-6) * b_alg((2 * w + 1) ** 2)
                     / 1e9 / HBM_PEAK_GBS})

    # (i) the list Chromosome.get_candidate makes for this matrix (raw mode), cold and warm
    sink = _io.StringIO()
    cold = []
    for rep in range(3):
        old = sys.stdout
        sys.stdout = sink
        try:
            t0 = time.perf_counter()
            X = scoreUtils.Chromosome(M, model=fo, raw_M=M, weights=None, lower=lower, upper=upper,
                                      cname="chr1", res=10000, width=w, device=dev)
            t1 = time.perf_counter()
            res, R = X.score(thre=thre)
            t2 = time.perf_counter()
        finally:
            sys.stdout = old
        cold.append({"construct_ms": (t1 - t0) * 1e3, "score_ms": (t2 - t1) * 1e3,
                     "candidates": int(X.ridx.size), "scored_pixels": int(res.nnz)})
        px, py = X.ridx.astype(np.int32), X.cidx.astype(np.int32)
        del X
    out["chromosome_cold"] = {"bins": int(M.shape[0]), "passes": cold,
                              "note": "Chromosome(...) [upload, band, expected curve, Poisson candidates] and "
                                      ".score() [pk_score_run + fetch + CSR build], three times in one process: "
                                      "the first carries this process's one-off costs"}
    leg("get_candidate (Poisson p < 0.01)", px, py, 50)
    # (ii) strided sub-lists of the all-non-zero-band-pixels list
    for m in ((1000, 10000, 100000, 1000000) if strided else ()):
        s = max(1, x_all.size // m)
        leg("every %d-th non-zero band pixel" % s, x_all[::s].copy(), y_all[::s].copy(), 200 if m <= 10000 else 50)
    out["legs"] = legs
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run the same command under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) as a child process and
    hand back its exit code (3 = the RCCL communicator could not be built)."""
    import subprocess
    # (--standalone: the launcher picks its own rendezvous port -- no port is chosen here and
    # handed over, which another process could have taken in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
           "--nnodes=1", "--nproc-per-node", str(n), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    # torch.distributed.run answers every failed rank with exit code 1; a rank that ends the
    # run for a NAMED reason (3 = no RCCL communicator) leaves its code in this file
    import tempfile
    fd, code_file = tempfile.mkstemp(prefix="pk_bench_exit_")
    os.close(fd)
    env["PK_BENCH_EXIT_FILE"] = code_file
    try:
        # the ranks' stdout carries library chatter ("[Gloo] Rank 0 is connected ...") beside rank
        # 0's JSON line: only the JSON goes to this process's stdout, the rest to stderr
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for line in proc.stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
        rc = proc.wait()
        txt = open(code_file).read().strip()
        return int(txt) if (rc != 0 and txt) else rc
    finally:
        os.unlink(code_file)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a timed region of about 0.7 s, long enough for an outside observer of GPU
    # activity to see it; the whole default run still takes well under a minute)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bins", dest="n", type=int, default=30000, help="matrix side in bins")
    ap.add_argument("--band", type=int, default=200)
    ap.add_argument("-w", "--width", dest="w", type=int, default=5)
    ap.add_argument("--thre", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=100000)
    ap.add_argument("--upper", type=int, default=None, help="largest candidate distance in bins (default: band)")
    ap.add_argument("--stride", type=int, default=1, help="score every stride-th band pixel")
    ap.add_argument("--forest", default=None,
                    help="flat-forest .npz, or random:T[:depth] for untrained random trees "
                         "(default: the committed forest for -w)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="time budget of the CPU baseline; large enough for the whole workload it also "
                         "compares every scored pixel with the GPU's (cpu_baseline.pixels_equal)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank its own chromosome (default); strong = one chromosome, "
                         "the candidate list cut into batch-aligned blocks")
    ap.add_argument("--allow-host-gather", "--allow-gloo-gather", dest="allow_host_gather", action="store_true",
                    help="N > 1: fall back to a gather over the host rendezvous (TCP) when the RCCL communicator "
                         "cannot be built (otherwise the run fails)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive extra leg")
    ap.add_argument("--full-evaluation", action="store_true",
                    help="headline on the kernels without the exact early exit (every candidate's complete "
                         "probability; rounds 1-4's headline); default: as Chromosome.score runs them")
    ap.add_argument("--no-real-regime", action="store_true", help="skip the short / scattered candidate-list legs")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="rehearsal only: all ranks use device 0 and the RCCL gather is skipped "
                         "(RCCL refuses two ranks on one GPU); the result is not a valid measurement")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short legs on the other single-GPU BASELINE.json shapes (w=6 / 300-bin band; "
                         "configs[3]: 60 000 bins, 800-bin band; configs[4]: w=11 x 500 trees, fitted and random) "
                         "that the default single-GPU run appends")
    ap.add_argument("--busy-seconds", type=float, default=0.0,
                    help="opt-in: an UNTIMED scoring loop of this many seconds after the timed region, so that "
                         "an outside GPU-activity sampler sees the device working; it never touches `value` "
                         "(single GPU only)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (dmabuf IPC: what RCCL needs between processes on this pool; read when the runtime initialises, so before
    # the library is loaded -- the launchers set it too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.gpus != world:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # started bare (`python bench.py --gpus N`): become the launcher.  The ranks are
            # CHILD processes of torch.distributed.run (one per GPU); this parent has touched
            # neither HIP nor the library, relays the child's output and exits with its code.
            sys.exit(self_launch(a.gpus, sys.argv[1:]))
        a.gpus = world

    from peakachu_amd import _lib
    from peakachu_amd.forest import FlatForest
    # which ROCm this process bound -- stamped into the line; anything but the copy `ldd` names is refused
    # (checked before the first device call: the answer does not need a GPU)
    rt = _lib.runtime_info()
    rdzv = None
    if world > 1:
        from peakachu_amd.rendezvous import Rendezvous
        rdzv = Rendezvous(rank, world)
    bad_rt = rt["product_runtime"] is False and not os.environ.get("PK_BENCH_ANY_RUNTIME")
    if rdzv:
        bad_rt = any(rdzv.all_gather_obj(bool(bad_rt)))
    if bad_rt:
        sys.stderr.write("bench.py: rank %d runs on %s, not on the ROCm `ldd` resolves for the library (%s): "
                         "refusing to measure another runtime\n" % (rank, rt["rocm_libs"], rt["rocm_dir_ldd"]))
        if rank == 0 and os.environ.get("PK_BENCH_EXIT_FILE"):
            open(os.environ["PK_BENCH_EXIT_FILE"], "w").write("3")
        if rdzv:
            rdzv.barrier()
            rdzv.close()
        sys.exit(3)
    L = _lib.require_device()
    dev = 0 if a.rehearse_shared_gpu else local_rank
    for kv in a.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))

    w = a.w
    F = (2 * w + 1) ** 2
    fo = load_forest(a.forest, w, F)
    # weak scaling: rank r scores its own synthetic chromosome (seed r); strong scaling: all
    # ranks hold chromosome 0 and rank r scores block r of its candidate list, cut at
    # multiples of the reference batch so that the batch rule sees the same batches
    strong = a.scaling == "strong" and world > 1
    Mf, exp_arr, x, y, upper, M_full = build_workload(0 if strong else rank, a.n, a.band, w, 6, a.upper or a.band,
                                                       with_matrix=True)
    if a.stride > 1:
        x, y = x[::a.stride].copy(), y[::a.stride].copy()
    x_all, y_all = x, y
    if strong:
        from peakachu_amd import dist as pkdist
        lo, hi = pkdist.block_ranges(x.size, world, a.batch)[rank]
        x, y = x[lo:hi].copy(), y[lo:hi].copy()
    t0 = time.perf_counter()
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], exp_arr,
                        -2 * w + 1, upper + 2 * w - 1, device=dev)
    hf = _lib.HipForest(fo, device=dev)
    cd = _lib.HipCands(x, y, device=dev)
    # what ships: Chromosome.score ALLOWS the exact early exit on its candidate list
    # (peakachu_amd/scoreUtils.py: cd.set_prune(True)) and the library applies it where it pays
    # (long launches: the forest cut in two at a tree-group boundary, pk_forest_q.hip q_pick_cut);
    # --full-evaluation withdraws the permission
    cd.set_prune(not a.full_evaluation)
    _lib.check(L.pk_device_synchronize(dev), "sync")
    upload_s = time.perf_counter() - t0

    # the gather of the scored pixels: RCCL (pk_comm_*); if the communicator cannot be
    # built on every rank, all ranks agree to send the (small) result through the host
    # rendezvous instead -- reported as "gather" in the JSON line, and only on request
    comm = None
    gather_mode = "none" if world == 1 else "rccl"
    if world > 1:
        if not a.rehearse_shared_gpu:
            buf = np.zeros(128, np.uint8)
            if rank == 0:
                _lib.check(L.pk_comm_unique_id(buf), "pk_comm_unique_id")
            uid = rdzv.broadcast(buf.tobytes())
            comm = L.pk_comm_create(dev, world, rank, np.frombuffer(uid, np.uint8).copy())
            if not comm:
                sys.stderr.write("rank %d: pk_comm_create failed: %s\n" % (rank, _lib.last_error()))
        if not all(rdzv.all_gather_obj(bool(comm))):
            if comm:
                L.pk_comm_destroy(comm)
            comm = None
            gather_mode = "host-tcp"
            if not (a.allow_host_gather or a.rehearse_shared_gpu):
                if rank == 0:
                    sys.stderr.write("bench.py: the RCCL communicator could not be built on every rank; "
                                     "refusing to measure a host gather (pass --allow-host-gather to do so)\n")
                    if os.environ.get("PK_BENCH_EXIT_FILE"):
                        open(os.environ["PK_BENCH_EXIT_FILE"], "w").write("3")
                rdzv.barrier()
                rdzv.close()
                sys.exit(3)
    rccl_ranks = L.pk_comm_ranks(comm) if comm else 0   # what RCCL itself counts (ncclCommCount)
    cap = int(x_all.size) * (1 if strong else world)
    cap = max(cap, int(x.size) * world)
    counts = np.zeros(world, np.int64)
    if rank == 0 and world > 1:
        gx = np.empty(cap, np.int32); gy = np.empty(cap, np.int32)
        gp = np.empty(cap, np.float64); gs = np.empty(cap, np.float64)

    t_run = [0.0]
    t_gather = [0.0]

    def step():
        ts = time.perf_counter()
        n_out = cd.run(hm, hf, w, a.thre, a.batch)   # returns after the stream has drained
        tg = time.perf_counter()
        t_run[0] += tg - ts
        _gather()
        t_gather[0] += time.perf_counter() - tg
        return n_out

    def _gather():
        if comm:
            if rank == 0:
                _lib.check(L.pk_comm_gather_scored(comm, cd.h, counts, cap, gx.ctypes.data,
                                                   gy.ctypes.data, gp.ctypes.data,
                                                   gs.ctypes.data), "gather")
            else:
                _lib.check(L.pk_comm_gather_scored(comm, cd.h, counts, 0, None, None, None,
                                                   None), "gather")
        elif gather_mode == "host-tcp":
            mine = cd.fetch()
            blobs = rdzv.gather(b"".join(np.ascontiguousarray(v).tobytes() for v in mine))
            if rank == 0:
                o = 0
                for r, blob in enumerate(blobs):
                    k = len(blob) // 24   # int32 x, int32 y, float64 prob, float64 signal per pixel
                    px = np.frombuffer(blob, np.int32, k, 0); py = np.frombuffer(blob, np.int32, k, 4 * k)
                    pp = np.frombuffer(blob, np.float64, k, 8 * k); ps = np.frombuffer(blob, np.float64, k, 16 * k)
                    counts[r] = px.size
                    gx[o:o + px.size] = px; gy[o:o + px.size] = py
                    gp[o:o + px.size] = pp; gs[o:o + px.size] = ps
                    o += px.size

    def sync():
        _lib.check(L.pk_device_synchronize(dev), "sync")
        if rdzv:
            rdzv.barrier()

    for _ in range(a.warmup):
        step()
    L.pk_prof_enable(1)
    L.pk_prof_reset()
    sync()
    t_run[0] = t_gather[0] = 0.0
    t0 = time.perf_counter()
    n_out = 0
    for _ in range(a.steps):
        n_out = step()
    sync()
    elapsed = time.perf_counter() - t0
    L.pk_prof_enable(0)
    kern = kernel_times(_lib)
    cut = cut_info(hf)
    run_ms, gather_ms = t_run[0] / a.steps * 1e3, t_gather[0] / a.steps * 1e3
    gpu_pixels = cd.fetch() if world == 1 else None
    # untimed: the timed region can be a fraction of a second (the driver fixes --steps), too
    # short for a 1 Hz GPU-activity sampler; keep the same scoring loop running for >= 2 s
    busy_steps = 0
    if a.busy_seconds > 0 and world == 1:
        t_b = time.perf_counter()
        while time.perf_counter() - t_b < a.busy_seconds:
            cd.run(hm, hf, w, a.thre, a.batch)
            busy_steps += 1

    # extra, not the headline: the same pass WITHOUT the permission to leave decided candidates at
    # probability 0 -- every candidate's complete probability, one forest launch per chunk (the headline
    # of rounds 1-4, and what pk_score_fetch_all's callers get)
    full = None
    if world == 1 and not a.full_evaluation:
        cd.set_prune(False)
        step()
        sync()
        e_steps = min(a.steps, 20)
        L.pk_prof_enable(1)
        L.pk_prof_reset()
        t0 = time.perf_counter()
        for _ in range(e_steps):
            n_full = step()
        sync()
        e_el = time.perf_counter() - t0
        L.pk_prof_enable(0)
        e_forest = _lib.prof_get("forest")
        cd.set_prune(True)
        f_avg = e_forest[0] / e_forest[1] if e_forest[1] else None
        full = {"value": int(x.size) * e_steps / e_el, "ms_per_step": e_el / e_steps * 1e3, "steps": e_steps,
                "forest_avg_launch_ms": f_avg, "forest_ms_per_step": e_forest[0] / e_steps,
                "roofline_frac_dominant_kernel": (float(x.size) * e_steps * b_alg(F) / (e_forest[0] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                  if e_forest[0] > 0 else None),
                "whole_path_frac": int(x.size) * e_steps / e_el * b_alg(F) / 1e9 / HBM_PEAK_GBS,
                "scored_pixels": int(n_full), "same_pixels_as_headline": bool(n_full == n_out),
                "note": "the headline's pass without pk_cands_set_prune: every candidate's complete probability "
                        "(forest_qr_kernel<.., 0>, one launch per chunk); threshold %g" % a.thre}

    # extra, not the headline: SURVEY.md 8d's literal metric -- the same steps through
    # pk_score with HOST coordinate / result buffers (H2D of the candidates and D2H of the
    # scored pixels inside the timed region; matrix and forest stay resident)
    pcie = None
    if world == 1 and not a.no_pcie:
        hm.score(hf, w, a.thre, x, y, batch=a.batch)
        sync()
        p_steps = min(a.steps, 20)
        t0 = time.perf_counter()
        for _ in range(p_steps):
            r_pcie = hm.score(hf, w, a.thre, x, y, batch=a.batch)
        sync()
        p_el = time.perf_counter() - t0
        n_local_rate = int(x.size) * a.steps / elapsed
        pcie = {"value": int(x.size) * p_steps / p_el, "unit": "candidates/s", "steps": p_steps,
                "ms_per_step": p_el / p_steps * 1e3, "scored_pixels": int(r_pcie[0].size),
                "frac_of_device_resident": (int(x.size) * p_steps / p_el) / (n_local_rate or 1.0),
                "note": "SURVEY 8d's literal metric: pk_score with HOST coordinate and result buffers -- upload "
                        "of the candidates (8 B each, chunk by chunk behind the kernels, checked on the device) "
                        "and download of the scored pixels inside the timed region; `value` is the same pass over a device-resident list (the bench contract: "
                        "inputs resident in HBM when the timed region starts)"}

    # extra, not the headline: the other single-GPU shapes of BASELINE.json, a few steps each
    extras = None
    if (world == 1 and not a.no_extra_configs and w == 5 and a.n == 30000 and a.band == 200 and a.stride == 1
            and not a.forest and not a.opt):
        extras = []
        w11_fitted = os.path.join(ROOT, "peakachu_amd", "data", "forest_w11_t500.npz")
        legs = [("w6 (the released 5/10 kb models' window)",
                 dict(n=30000, band=300, w=6, upper=300, forest_spec=None, steps=5, pmc_file="pmc_w6.json")),
                ("configs[3] (5 kb map: 2x bins, upper = 800 bins)",
                 dict(n=60000, band=800, w=5, upper=800, forest_spec=None, steps=3, pmc_file="pmc_5kb.json")),
                # SURVEY 8d's stress forest: RandomForestClassifier(500, max_depth=20) FITTED on buildmatrix
                # features (tools/make_forest.py -w 11 -T 500), and the untrained random trees of rounds 1-3
                ("configs[4] (fitted forest)",
                 dict(n=8000, band=200, w=11, upper=200, forest_spec=w11_fitted, steps=5, pmc_file="pmc_w11.json")),
                ("configs[4] (random trees: second stress leg)",
                 dict(n=8000, band=200, w=11, upper=200, forest_spec="random:500:20", steps=5,
                      pmc_file="pmc_w11_random.json"))]
        for name, kw in legs:
            if kw["forest_spec"] == w11_fitted and not os.path.exists(w11_fitted):
                extras.append({"workload": name, "error": "peakachu_amd/data/forest_w11_t500.npz is missing "
                                                          "(tools/make_forest.py -w 11 -T 500)"})
                continue
            try:
                extras.append(extra_config(L, dev, name, thre=a.thre, batch=a.batch, **kw))
            except Exception as e:  # reported, never fatal for the headline
                extras.append({"workload": name, "error": "%s: %s" % (type(e).__name__, e)})

    # extra, not the headline: the regime the CLI runs in -- get_candidate's Poisson-filtered list and
    # short strided lists, plus the cold cost of a Chromosome(...) + .score() of this matrix
    regime = None
    if world == 1 and not a.no_real_regime and a.stride == 1:
        try:
            regime = real_regime(L, dev, M_full, fo, w, 6, upper, a.thre, a.batch, x, y, hm, hf)
        except Exception as e:  # reported, never fatal for the headline
            regime = {"error": "%s: %s" % (type(e).__name__, e)}

    # strong scaling: the merged result must equal the single-GPU result of the whole list
    strong_check = None
    if strong:
        total_pix = int(counts.sum())
        if rank == 0:
            import hashlib
            cd_all = _lib.HipCands(x_all, y_all, device=dev)
            cd_all.run(hm, hf, w, a.thre, a.batch)
            ref = cd_all.fetch()
            cd_all.close()
            got = (gx[:total_pix], gy[:total_pix], gp[:total_pix], gs[:total_pix])
            same = all(np.array_equal(np.ascontiguousarray(r).view(np.uint8),
                                      np.ascontiguousarray(g_).view(np.uint8)) for r, g_ in zip(ref, got))
            h = hashlib.sha256()
            for arr in got:
                h.update(np.ascontiguousarray(arr).tobytes())
            strong_check = {"merged_equals_single_gpu": bool(same), "pixels": total_pix,
                            "sha256": h.hexdigest()[:16]}

    n_local = int(x.size)
    per_rank_ms = [run_ms]
    rt_all = [rt]
    if rdzv:
        # (one exchange: per-rank step time, the timed region's length -- MAX over ranks is the job's --,
        # candidates per rank, and what each rank's runtime says about itself)
        got = rdzv.all_gather_obj([float(run_ms), float(elapsed), n_local, rt])
        per_rank_ms = [float(g[0]) for g in got]
        elapsed = max(float(g[1]) for g in got)
        n_total = sum(int(g[2]) for g in got)
        rt_all = [g[3] for g in got]
    else:
        n_total = n_local

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = n_total * a.steps / elapsed
        # dominant kernel = the class with the most device time on rank 0
        dom = max(("extract", "quant", "forest"), key=lambda k: kern[k][0])
        dom_ms, dom_n = kern["forest_head" if dom == "forest" else dom]
        alg_bytes_total = float(n_local) * a.steps * b_alg(F)
        achieved = alg_bytes_total / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        own = own_alg_bytes(dom, F)
        own_achieved = (float(n_local) * a.steps * own / (dom_ms * 1e-3) / 1e9) if (own and dom_ms > 0) else None
        # HBM bytes per launch of the dominant kernel and its binding rooflines, from the
        # committed PMC passes (profiles/pmc.json); only valid for the default workload and
        # reported as stale (traffic = null) when the kernel sources changed since
        traffic = issue_roof = lds_roof = path_traffic = None
        default_workload = (w == 5 and a.n == 30000 and a.band == 200 and a.stride == 1 and not a.forest
                            and not a.opt and world == 1)
        if default_workload and dom_n:
            traffic, issue_roof, lds_roof = pmc_rooflines(dom, dom_n / a.steps)
            path_traffic = pmc_path_traffic(kern, a.steps)
        fst = fo.stats()
        out = {
            "metric": "candidate pixels scored/sec",
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": a.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic %dx%d band-diagonal (%d-bin band) CSR, w=%d, %d-tree RF%s, "
                            "%s non-zero band pixels %d<=d<=%d; value = device-resident candidate list through "
                            "pk_score_run %s; SURVEY 8d's literal metric "
                            "(host coordinate / result buffers) is the pcie_inclusive leg%s"
                            % (a.n, a.n, a.band, w, fo.T,
                               " (untrained random trees)" if (a.forest or "").startswith("random:") else "",
                               "all" if a.stride == 1 else "every %d-th of the" % a.stride,
                               max(6, w + 1), upper,
                               ("without permission to stop early (every candidate's complete probability)" if a.full_evaluation
                                else "AS Chromosome.score runs it: decided candidates may end at probability 0 "
                                     "(pk_cands_set_prune), which lets the library cut the forest in two -- see "
                                     "forest_cut; full_evaluation = the same pass without the permission"),
                               "" if world == 1 else
                               ("; strong scaling: every rank holds chromosome seed 0" if strong else
                                "; weak scaling: rank r scores its own synthetic chromosome, seed = r "
                                "(seeds 0..%d), so N = 1 equals the single-GPU bench" % (world - 1))),
                "candidates_per_gpu": n_local,
                "features": F,
                "trees": fo.T,
                "nodes_per_tree_mean": round(fst["nodes_per_tree_mean"], 1),
                "threshold": a.thre,
                "reference_batch": a.batch,
                "scored_pixels_rank0": int(n_out),
                "parallelism": "%s x%d, one %s gather of the scored pixels%s"
                               % ("batch-aligned candidate blocks of one chromosome" if strong
                                  else "chromosome-sharded", world,
                                  {"rccl": "RCCL", "host-tcp": "host-rendezvous (RCCL unavailable)",
                                          "none": "(single rank: no)"}[gather_mode],
                                  " (REHEARSAL: shared GPU)" if a.rehearse_shared_gpu else ""),
                "gather": gather_mode,
                "gathered_pixels": int(counts.sum()) if world > 1 else int(n_out),
            },
            # `achieved` / `frac` lead with the WHOLE PATH (SURVEY 8d's own definition: candidates/s x B_alg /
            # peak, per GPU): since the forest is cut in two, the dominant KERNEL (the head) walks only the
            # trees in front of the cut, and pricing it with the path's bytes flatters it.  The brief's
            # per-kernel recipe (B_alg x candidates per launch / the kernel's average launch time) is kept as
            # `dominant_kernel_frac`, next to the same kernel charged with the whole forest stage
            # (`dominant_kernel_frac_stage_time`: head + gap + tail) and credited with its own bytes only
            # (`kernel_own_frac`).
            "roofline": {
                "bound": "hbm",
                "kernel": dom + (" (forest_qr_kernel<..,1>: the head of the cut forest, %d of %d tree groups over every "
                                 "candidate; the tail -- forest_tail in kernel_ms_per_step -- is a launch of its own)"
                                 % (cut["group"], cut["of_groups"]) if (cut and dom == "forest") else ""),
                "achieved": value / world * b_alg(F) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": value / world * b_alg(F) / 1e9 / HBM_PEAK_GBS,
                "frac_is": "whole path: candidates/s x B_alg / peak (SURVEY 8d)",
                "dominant_kernel_achieved": achieved,
                "dominant_kernel_frac": achieved / HBM_PEAK_GBS,
                "dominant_kernel_frac_stage_time": (alg_bytes_total / (kern[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                    if kern[dom][0] > 0 else None),
                # HBM bytes the PMC counters saw (the guide's recipe, profiles/pmc.json): `traffic` = the whole
                # path's per STEP, beside the algorithmic bytes of a step it is to be read against;
                # `dominant_kernel_traffic` = that kernel's per launch
                "traffic": path_traffic,
                "alg_bytes_per_step": float(n_local) * b_alg(F),
                "traffic_per_candidate": (path_traffic / n_local) if path_traffic else None,
                "dominant_kernel_traffic": traffic,
                # measured HBM rate of that kernel: PMC bytes per launch / its average duration
                "traffic_GBs": (traffic / (dom_ms / dom_n * 1e-3) / 1e9) if (traffic and dom_n) else None,
                "alg_bytes_per_candidate": b_alg(F),
                "avg_launch_ms": dom_ms / dom_n if dom_n else None,
                "launches": dom_n,
                "candidates_per_launch": n_local * a.steps / dom_n if dom_n else None,
                # that kernel's OWN share of B_alg (extract: 12F+8, forest: 4F+8) over its time
                "kernel_own_alg_bytes_per_candidate": own,
                "kernel_own_frac": own_achieved / HBM_PEAK_GBS if own_achieved is not None else None,
                # the same algorithmic bytes over (i) the forest STAGE = rank quantizer + forest
                # kernel, (ii) the WHOLE path = SURVEY 8d's definition, value x B_alg / peak
                "stage_frac": (alg_bytes_total / ((kern["forest"][0] + kern["quant"][0]) * 1e-3) / 1e9
                               / HBM_PEAK_GBS) if (kern["forest"][0] + kern["quant"][0]) > 0 else None,
                "whole_path_frac": value / world * b_alg(F) / 1e9 / HBM_PEAK_GBS,
            },
            # SURVEY 8d's definition: candidates/s x B_alg / peak (per GPU) -- the WHOLE path, not one kernel
            "roofline_whole_path_frac": value / world * b_alg(F) / 1e9 / HBM_PEAK_GBS,
            # the same pass without the permission to stop early (rounds 1-4's headline), beside `value`
            "full_evaluation_value": full["value"] if full else (value if a.full_evaluation else None),
            "early_exit_allowed": not a.full_evaluation,
            "forest_cut": cut,
            # which runtime ran (rank 0; `runtimes_agree`: every rank reports the same versions and paths)
            "hip_runtime_version": rt["hip_runtime_version"],
            "hip_driver_version": rt["hip_driver_version"],
            "rccl_version": rt["rccl_version"],
            "rocm_libs": rt["rocm_libs"],
            "rocm_dir_ldd": rt["rocm_dir_ldd"],
            "product_runtime": rt["product_runtime"],
            "runtimes_agree": all(r == rt for r in rt_all),
            "kernel_ms_per_step": {k: v[0] / a.steps for k, v in kern.items()},
            "whole_path_alg_GBs": value * b_alg(F) / 1e9,
            "upload_s": upload_s,
            "per_rank_ms": per_rank_ms,       # device work of a step (pk_score_run), per rank
            "gather_ms": gather_ms,           # the exchange of a step, as seen by rank 0
            "untimed_busy_steps": busy_steps,  # same scoring loop, after the timed region (--busy-seconds)
        }
        if issue_roof is not None:
            out["roofline_issue"] = issue_roof
            out["roofline_lds"] = lds_roof
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if strong_check is not None:
            out["strong_check"] = strong_check
        if full is not None:
            out["full_evaluation"] = full
        if regime is not None:
            out["real_regime"] = regime
        if world > 1:
            out["rccl_ranks"] = int(rccl_ranks)
        if extras is not None:
            out["other_configs"] = extras
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Mf, exp_arr, w, fo, a.thre, x, y, a.batch, gpu_pixels,
                                               target_s=a.cpu_seconds)
        print(json.dumps(out))
        sys.stdout.flush()

    if comm:
        L.pk_comm_destroy(comm)
    if rdzv:
        rdzv.barrier()
        rdzv.close()


def guarded_main():
    """A library error inside a run (a gather that gave up on its peers after PK_COMM_TIMEOUT, a HIP
    failure) ends the process with a diagnostic and a non-zero code instead of leaving the other ranks
    -- and the driver -- waiting: every failing rank writes one JSON diagnostic line to STDERR (stdout
    carries a JSON line only for a run that measured something), exit code 4.  (No re-exec, no retry: the process has touched the GPU; a retry is a fresh one.)"""
    try:
        main()
    except Exception as e:
        from peakachu_amd import _lib
        if not isinstance(e, _lib.PeakachuHipError):
            raise
        rank = int(os.environ.get("RANK", "0"))
        sys.stderr.write("bench.py: rank %d of %s: %s\n" % (rank, os.environ.get("WORLD_SIZE", "1"), e))
        sys.stderr.write(json.dumps({"bench_error": str(e), "rank": rank,
                                     "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "exit_code": 4}) + "\n")
        sys.stdout.flush()
        sys.stderr.flush()
        if os.environ.get("PK_BENCH_EXIT_FILE"):
            open(os.environ["PK_BENCH_EXIT_FILE"], "w").write("4")
        os._exit(4)


if __name__ == "__main__":
    guarded_main()
