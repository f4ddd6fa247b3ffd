"""`peakachu pool`: the host-side loop caller against the
reference's own output (tests/golden/g7_pool.npz, made by tools/make_golden.py g7 by
running peakachu/call_loops.py:main on synthetic scored-pixel files)."""
import os
import warnings

import numpy as np
import pytest

import golden_io as gio
from peakachu_amd import cli, peakacluster


@pytest.fixture(scope="module")
def g7():
    return gio.load("g7_pool.npz")


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("thre", [0.9, 0.5, 0.97])
def test_pool_cli_byte_identical(g7, tmp_path, seed, thre):
    res = int(g7["res%d" % seed])
    fin = tmp_path / "in.bedpe"
    fin.write_bytes(bytes(g7["in%d" % seed]))
    fout = tmp_path / "out.bedpe"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")        # scipy's PeakPropertyWarning, as in the reference
        cli.run(["pool", "-r", str(res), "-i", str(fin), "-o", str(fout), "-t", str(thre)])
    want = bytes(g7["out%d_t%g" % (seed, thre)])
    assert fout.read_bytes() == want and want.count(b"\n") > 3


@pytest.mark.parametrize("seed", [1, 2])
def test_local_clustering_and_anchors(g7, seed):
    res = int(g7["res%d" % seed])
    D = {}
    for line in bytes(g7["in%d" % seed]).decode().splitlines():
        q = line.split()
        if q[0] == "chr1" and float(q[6]) >= 0.5:
            D[(int(q[1]) // res, int(q[4]) // res)] = float(q[7])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        reps = sorted(set(t[0] for t in peakacluster.local_clustering(D, min_count=3, r=2)))
        xa = sorted(peakacluster.find_anchors(np.r_[[k[0] for k in D]], min_count=3, min_dis=2))
    assert np.array_equal(np.array(reps, np.int64), g7["reps%d" % seed])
    assert np.array_equal(np.array(xa, np.int64), g7["xanchors%d" % seed])


def test_pool_default_threshold_and_help(g7, tmp_path, capsys):
    fin = tmp_path / "in.bedpe"
    fin.write_bytes(bytes(g7["in1"]))
    fout = tmp_path / "out.bedpe"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cli.run(["pool", "-i", str(fin), "-o", str(fout)])      # -r 10000, -t 0.9
    assert fout.read_bytes() == bytes(g7["out1_t0.9"])
    with pytest.raises(SystemExit):
        cli.run(["pool"])                                        # no arguments -> help
    assert "--threshold" in capsys.readouterr().out


def test_edge_cases(tmp_path):
    # empty input, one pixel, one cluster representative: no loops (peakacluster.py:27-30)
    for text in ("", "chr1\t100000\t110000\tchr1\t300000\t310000\t0.95\t7.0\n"):
        fin = tmp_path / "e.bedpe"
        fin.write_text(text)
        loops, pool = peakacluster.parse_peakachu(str(fin), 0.9, 10000)
        assert all(v == [] for v in loops.values())
    assert peakacluster.local_clustering({}) == []
    # two far-apart singletons on no anchor: nothing is called
    D = {(10, 40): 3.0, (500, 800): 4.0}
    assert peakacluster.local_clustering(D) == []


def test_dbscan_min_samples_2_matches_sklearn():
    """The reference clusters with sklearn.cluster.dbscan(min_samples=2); the build uses
    connected components -- same labels, including duplicates and label numbering."""
    skc = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(0)
    for trial in range(40):
        n = int(rng.integers(2, 300))
        pts = rng.integers(0, int(rng.integers(5, 80)), size=(n, 2))
        if trial % 5 == 0:
            pts[n // 2:] = pts[: n - n // 2]        # duplicates
        for eps in (2, 3):
            want = skc.dbscan(pts, eps=eps, min_samples=2)[1]
            got = peakacluster._dbscan2(pts, eps)
            assert np.array_equal(got, want), (trial, eps)
