"""A fixed-seed slice of the fuzz campaigns inside `-m gpu` (tests/fuzz/*.py: seeded random
forests, scoring cases, chromosomes and coordinate sets through the C ABI against the CPU
oracle, bit for bit), so that the driver's GPU test record carries them and no lease minutes
go to re-running whole campaigns by hand.  The scripts stay usable on their own with other
seeds and sizes (`python tests/fuzz/fuzz_forest.py 3000 <seed>`)."""
import importlib
import io
import os
import sys
from contextlib import redirect_stdout

import pytest

FUZZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,cases,seed", [
    ("fuzz_forest", 150, 50501),       # model.predict_proba: 1..1100 features, stumps to combs, cut trees, 12-bit ranks
    ("fuzz_score", 300, 50502),        # Chromosome.score's body: w = 1..15, dirty matrices, batch sizes, options
    ("fuzz_chromosome", 150, 50503),   # Chromosome.__init__ + score: raw / balanced / hic-style, device-side preparation
    ("fuzz_getwindow", 500, 50504),    # getwindow at arbitrary coordinates
    ("fuzz_cut", 40, 50505),           # the forest cut in two: every rank kernel, forced and learnt cuts, NaN features
])
def test_fuzz_slice(hip_lib, monkeypatch, script, cases, seed):
    monkeypatch.syspath_prepend(FUZZ)
    mod = importlib.import_module(script)
    monkeypatch.setattr(sys, "argv", [script + ".py", str(cases), str(seed)])
    out = io.StringIO()
    try:
        with redirect_stdout(out):
            mod.main()
    except SystemExit as e:  # the scripts leave with 1 at the first mismatch, after printing the case
        assert not e.code, out.getvalue()[-3000:]
    tail = out.getvalue().strip().splitlines()[-1]
    assert tail.startswith("all %d cases" % cases), tail
