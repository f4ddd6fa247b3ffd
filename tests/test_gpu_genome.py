"""BASELINE.json configs[0] / configs[2] at their real SHAPE (`-m gpu`): `score_chromosome chr21`
and `score_genome` on an hg19-shaped 10 kb map -- 25 chromosomes at their real bin counts, 23
selected by the default `-C '#' X` (303 641 bins), a .cool written by the genuine HDF5 library
in cooler's layout, the released models' window (w = 6) -- against the reference's chain on the
CPU with the oracle as the scorer (tools/genome_standin.oracle_bedpe: matrices made from the
synthetic counts without the .cool reader, utils.calculate_expected -> band_filter ->
candidates (scipy per pixel) -> oracle.score -> write_bedpe; peakachu/score_genome.py:26-84,
score_chromosome.py:3-71).  The GM12878 map and the released models cannot be had offline
(SURVEY.md 8c): counts, weights and forest are synthetic stand-ins, the shapes, the container
format and the command lines are the real ones.  The band is 120 bins (`-u 100`) so that the CPU
chain of the whole genome stays within half a minute; tools/genome_standin.py e2e runs the
default `-u 300` on a full-width map and keeps its log under profiles/."""
import os

import pytest

from tools import genome_standin as gs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join(ROOT, "peakachu_amd", "data", "forest_w6_t100.npz")
UPPER = 100

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome(hip_lib, tmp_path_factory):
    work = str(tmp_path_factory.mktemp("hg19_standin"))
    man = gs.synthesize(work, band=120, seed=5)
    if gs.have_h5py_writer():
        path = os.path.join(work, "standin.cool")
        gs.write_cool(work, path, level=1)
    else:  # no h5py interpreter on this box: the same genome through the package's own container
        path = os.path.join(work, "standin.pkmap.npz")
        gs.write_pkmap(man, work, path)
    return man, work, path


def _cli(argv):
    from peakachu_amd import cli
    cli.run(argv)


@pytest.mark.parametrize("wname", ["raw", "weight"])
def test_score_genome_on_an_hg19_shaped_map(genome, tmp_path, wname):
    man, work, path = genome
    out, ref = str(tmp_path / "gpu.bedpe"), str(tmp_path / "oracle.bedpe")
    _cli(["score_genome", "-p", path, "-m", MODEL, "-O", out, "--clr-weight-name", wname, "-u", str(UPPER)])
    T = gs.oracle_bedpe(man, work, MODEL, wname, 6, UPPER, 0.5, ref)
    got, want = open(out, "rb").read(), open(ref, "rb").read()
    assert got == want
    names = {line.split(b"\t")[0] for line in got.splitlines()}
    assert names == {("chr%s" % c).encode() for c in list(range(1, 23)) + ["X"]}  # chrY / chrM filtered out
    assert T["candidates_total"] > 400000 and got.count(b"\n") > 20000  # not vacuous


def test_score_chromosome_chr21_on_an_hg19_shaped_map(genome, tmp_path):
    man, work, path = genome
    out, ref = str(tmp_path / "gpu.bedpe"), str(tmp_path / "oracle.bedpe")
    _cli(["score_chromosome", "-p", path, "-m", MODEL, "-O", out, "-C", "chr21", "-u", str(UPPER)])
    i21 = [i for i, c in enumerate(man["chroms"]) if c["name"] == "chr21"]
    assert man["chroms"][i21[0]]["bins"] == 4813  # 48 129 895 bp at 10 kb
    gs.oracle_bedpe(man, work, MODEL, "weight", 6, UPPER, 0.5, ref, only=i21)
    got = open(out, "rb").read()
    assert got == open(ref, "rb").read()
    assert got.count(b"\n") > 300


def test_score_chromosome_chr1_at_the_default_upper(hip_lib, tmp_path):
    """The CLI's DEFAULT -u 300 on the genome's largest chromosome (chr1 of hg19: 24 926 bins) over a map whose
    band is wider than upper + 2w (320 bins), weights on: bedpe bytes against the oracle chain.  (The genome-wide
    tests above run -u 100 on a 120-bin band to keep their CPU chain short.)"""
    from peakachu_amd import synth
    work = str(tmp_path / "chr1")
    man = gs.synthesize(work, band=320, seed=11, chroms=synth.HG19_CHROMS[:1])
    assert man["chroms"][0]["name"] == "chr1" and man["chroms"][0]["bins"] == 24926
    if gs.have_h5py_writer():
        path = os.path.join(work, "chr1.cool")
        gs.write_cool(work, path, level=1)
    else:
        path = os.path.join(work, "chr1.pkmap.npz")
        gs.write_pkmap(man, work, path)
    out, ref = str(tmp_path / "gpu.bedpe"), str(tmp_path / "oracle.bedpe")
    _cli(["score_chromosome", "-p", path, "-m", MODEL, "-O", out, "-C", "chr1"])   # -l 6 -u 300, weight: the defaults
    T = gs.oracle_bedpe(man, work, MODEL, "weight", 6, 300, 0.5, ref, only=[0])
    got = open(out, "rb").read()
    assert got == open(ref, "rb").read()
    assert T["candidates_total"] > 50000 and got.count(b"\n") > 1000   # not vacuous


def _same_chromosome(A, B, thre=0.3):
    import numpy as np
    bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
    assert np.array_equal(bits(A.exp_arr), bits(B.exp_arr)) and np.array_equal(bits(A.background), bits(B.background))
    assert np.array_equal(A.ridx, B.ridx) and np.array_equal(A.cidx, B.cidx)
    (pa, sa), (pb, sb) = A.score(thre), B.score(thre)
    for X, Y in ((pa, pb), (sa, sb)):
        X, Y = X.tocsr(), Y.tocsr()
        X.sort_indices(), Y.sort_indices()
        assert np.array_equal(X.indptr, Y.indptr) and np.array_equal(X.indices, Y.indices)
        assert np.array_equal(bits(X.data), bits(Y.data))
    return pa.nnz


@pytest.mark.parametrize("mode", ["raw", "weight", "float_counts"])
def test_chromosome_from_the_pixel_table_equals_the_mirrored_flow(hip_lib, mode):
    """Chromosome.from_upper (one upload of the upper triangle as a .cool stores it, mirrored and
    balanced on the device: pk_csr_upload_upper / pk_csr_view) against the reference's flow, the
    mirrored and balanced host matrices handed to Chromosome(...) (peakachu/score_genome.py:55-58):
    expected curve, candidates, scored pixels bit for bit -- with trans pixels riding in the rows,
    explicit zero counts and NaN weights."""
    import numpy as np
    from peakachu_amd import scoreUtils, synth, utils
    from peakachu_amd.forest import load_model
    n, band, w = 3000, 150, 6
    cnt = synth.band_counts(n, band, seed=21)
    cnt[5, 3] = 0
    i, d = np.nonzero(cnt)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(i, minlength=n), out=indptr[1:])
    cols, counts = (i + d).astype(np.int32), cnt[i, d].astype(np.int32)
    # trans pixels behind every 7th row's cis pixels, and a stored zero count
    rows = np.arange(0, n, 7)
    ins = indptr[rows + 1]
    cols = np.insert(cols, ins, n + 5 + (rows % 11)).astype(np.int32)
    counts = np.insert(counts, ins, 3).astype(np.int32)
    indptr = indptr + np.searchsorted(rows, np.arange(n + 1), side="left")
    counts[int(indptr[40])] = 0
    if mode == "float_counts":
        counts = counts.astype(np.float64)
    px = utils.UpperPixels(n, indptr.astype(np.int32), cols, counts)
    model = load_model(MODEL)
    kw = dict(lower=6, upper=120, cname="chr1", res=10000, width=w)
    if mode == "weight":
        wts = synth.synth_weights(n, 3, n_nan=7)
        A = scoreUtils.Chromosome.from_upper(px, model, bias=wts, weights=wts, **kw)
        B = scoreUtils.Chromosome(px.symmetric(wts), model, raw_M=px.symmetric(), weights=wts, **kw)
    else:
        A = scoreUtils.Chromosome.from_upper(px, model, **kw)
        raw = px.symmetric()
        B = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, **kw)
    assert A._cands is not None and B._cands is not None   # both on the device path
    assert A._raw_val is None                               # no host matrix was made
    assert _same_chromosome(A, B) > 100
    # the host copies exist for whoever reads them, as the reference's attributes do
    assert (A.raw_M != px.symmetric()).nnz == 0


def test_a_pixel_table_out_of_order_is_mirrored_on_the_host(hip_lib):
    import numpy as np
    from peakachu_amd import scoreUtils, synth, utils
    from peakachu_amd.forest import load_model
    n, band, w = 900, 80, 6
    cnt = synth.band_counts(n, band, seed=4)
    i, d = np.nonzero(cnt)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(i, minlength=n), out=indptr[1:])
    cols, counts = (i + d).astype(np.int32), cnt[i, d].astype(np.int32)
    a = int(indptr[10])
    cols[a], cols[a + 1] = cols[a + 1], cols[a]          # two pixels of a row swapped
    px = utils.UpperPixels(n, indptr.astype(np.int32), cols, counts)
    model = load_model(MODEL)
    kw = dict(lower=6, upper=60, cname="chr1", res=10000, width=w)
    A = scoreUtils.Chromosome.from_upper(px, model, **kw)
    assert A._pixels is None                               # the device refused the table
    raw = px.symmetric()
    B = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, **kw)
    assert _same_chromosome(A, B) > 10


@pytest.mark.parametrize("n", [4, 13, 14, 40])
def test_tiny_chromosomes_from_the_pixel_table(hip_lib, n):
    """Chromosomes too short for a window (n <= 2w: the reference clamps `upper` below `lower` and finds
    no candidate) and barely long enough: the pixel-table constructor takes the same way out as the
    matrix constructor (host path where the device path does not apply)."""
    import numpy as np
    from peakachu_amd import scoreUtils, synth, utils
    from peakachu_amd.forest import load_model
    cnt = synth.band_counts(n, min(n - 1, 20), seed=n)
    i, d = np.nonzero(cnt)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(i, minlength=n), out=indptr[1:])
    px = utils.UpperPixels(n, indptr.astype(np.int32), (i + d).astype(np.int32), cnt[i, d].astype(np.int32))
    model = load_model(MODEL)
    kw = dict(lower=6, upper=300, cname="chrT", res=10000, width=6)
    raw = px.symmetric()
    try:
        B = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, **kw)
    except Exception as e:   # what the reference's arithmetic does with such a map, the other constructor does too
        with pytest.raises(type(e)):
            scoreUtils.Chromosome.from_upper(px, model, **kw)
        return
    A = scoreUtils.Chromosome.from_upper(px, model, **kw)
    _same_chromosome(A, B)
