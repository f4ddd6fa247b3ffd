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
