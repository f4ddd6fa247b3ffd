"""Hostile contact-map files: the product's HDF5 reader (peakachu_amd/h5lite.py, cool.py -- what stands
in for cooler -> h5py -> libhdf5 behind peakachu/score_genome.py:26-35,55-57) parses a binary format in
Python and hands chunks to C.  Seeded corruptions -- bit flips, truncation, random / zero / 0xff blocks --
of the tracked fixtures must each end in a Python exception (or in a clean read when the damage hit
nothing that is read), within a time bound, never in a crash, a hang or an unbounded allocation.  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name,tail,count", [("cool_small.cool", "", 120), ("cool_small_latest.cool", "", 120),
                                             ("cool_small.mcool", "::/resolutions/10000", 60)])
def test_corrupt_files_end_in_exceptions(tmp_path, name, tail, count):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_h5_corrupt_worker.py"), os.path.join(G, name), tail,
                        str(tmp_path), "0", str(count)], capture_output=True, text=True, timeout=900)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1] == "DONE", \
        "the reader took the interpreter down after case %s (rc %d): %s" % (lines[-1] if lines else "-", r.returncode, r.stderr[-2000:])
    res = [ln.split(None, 1)[1] for ln in lines[:-1]]
    assert len(res) == count
    assert "HANG" not in res, [ln for ln in lines if "HANG" in ln]
    assert all(x == "ok" or x.startswith("exc ") for x in res)
    # the corruptions do bite, and a refused file is refused with an exception a caller can catch
    assert sum(x.startswith("exc ") for x in res) >= count // 4, res
    assert not any(x in ("exc SystemExit", "exc KeyboardInterrupt") for x in res)
