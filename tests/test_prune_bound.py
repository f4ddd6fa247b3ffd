"""The early-exit / cut bound of the forest kernels is a PROVEN bound (csrc/pk_common.h: pk_prune_bound).

A kernel decides a candidate -- stops walking it, reports probability 0 -- when
`fl(acc + remaining) < bound`.  That may only happen when the reference's own arithmetic
(peakachu/scoreUtils.py:109-110: sequential float64 sum of the trees' leaf values, `/ T`, `> thre`)
would not report the pixel.  Checked here without a device: the inequality of the proof in exact
rational arithmetic, and sums built to round UPWARDS at every addition driven through the very test
the kernels make (numpy float64 = the device's IEEE arithmetic)."""
from fractions import Fraction

import numpy as np
import pytest

from peakachu_amd import _lib

SIZES = (100, 9000, 20000, 60000)


def bound(thre, T, additions=None):
    return float(_lib.load().pk_debug_prune_bound(float(thre), int(T), int(T if additions is None else additions)))


@pytest.mark.parametrize("T", SIZES + (1, 2, (1 << 26) - 5))
def test_the_inequality_of_the_proof_holds_exactly(T):
    u = Fraction(1, 1 << 53)
    M = 1 + (T + 4) * Fraction(1, 1 << 52)
    assert float(M) == 1.0 + (T + 4) * 2.0 ** -52 and Fraction(float(M)) == M   # M is a binary64 number
    # (1 + u)^(T + 1) <= 1 + (T + 2) u  <=  M (1 - u); the power itself for the sizes that finish quickly
    assert 1 + (T + 2) * u <= M * (1 - u)
    if T <= 60000:
        assert (1 + u) ** (T + 1) <= 1 + (T + 2) * u
    else:   # Bernoulli-type bound: (1+u)^n <= 1/(1 - n u) for n u < 1
        n = T + 1
        assert n * u < 1 and 1 / (1 - n * u) <= 1 + (T + 2) * u


@pytest.mark.parametrize("T", SIZES)
def test_the_bound_lies_below_thre_T_by_the_margin(T):
    for thre in (0.5, 0.1, 0.9, 0.55, 1.0 / 3.0, np.nextafter(1.0, 0.0), 1e-3):
        b = bound(thre, T)
        M = 1 + (T + 4) * Fraction(1, 1 << 52)
        assert Fraction(b) < Fraction(float(thre) * T) / M     # rounded DOWN, strictly
        assert b > thre * T * (1 - (T + 8) * 2.0 ** -52)       # ... and no looser than the margin needs
    # more additions than trees (trees cut into pieces): the larger margin; absurd sizes decide nobody
    assert bound(0.5, 100, 5000) < bound(0.5, 100, 100)
    assert bound(0.5, 100, 1 << 26) == -np.inf


def _never_decided(vals, thre):
    """True when no prefix of the sequential sum passes the kernels' test although the final p > thre."""
    T = vals.size
    acc = np.concatenate(([0.0], np.cumsum(vals)))          # cumsum is sequential: sklearn's order
    assert acc[-1] / T > thre
    rem = (T - np.arange(T + 1)).astype(np.float64)
    return not np.any(acc + rem < bound(thre, T))


@pytest.mark.parametrize("T", SIZES)
def test_sums_that_round_upwards_are_never_decided_too_early(T):
    rng = np.random.default_rng(T)
    k = int(np.floor(np.log2(T)))
    seqs = []
    # (a) every addition rounds UP: values just under 1 whose shortfall is below half an ulp of the
    #     running sum (the sum gains up to half an ulp per tree over the exact one)
    for e in (k - 52 - 1, k - 52 - 2, -40, -45):
        seqs.append(np.full(T, 1.0 - 2.0 ** e * (1 - 2.0 ** -10)))
    # (b) a fractional start, then ones: acc + 1.0 re-rounds at every binade crossing
    for frac in (2.0 ** -30 + 2.0 ** -52, 1.0 - 2.0 ** -53, 0.3):
        v = np.ones(T); v[0] = frac
        seqs.append(v)
    # (c) leaf-like values: mostly 0 / 1 with some fractions, and thirds (inexact in binary)
    v = rng.choice([0.0, 1.0, 1.0 / 3.0, 2.0 / 3.0, 0.1], size=T, p=[0.3, 0.4, 0.1, 0.1, 0.1])
    seqs.append(v)
    seqs.append(np.sort(v))            # the large values LAST: the decision hangs on the remaining trees
    seqs.append(np.sort(v)[::-1].copy())
    for vals in seqs:
        p = np.cumsum(vals)[-1] / T
        # the threshold just below the final probability: the pixel IS reported by the reference
        for thre in (np.nextafter(p, 0.0), np.nextafter(np.nextafter(p, 0.0), 0.0), p * (1 - 1e-13), 0.5 * p):
            if thre > 0:
                assert _never_decided(vals, float(thre)), (T, thre)


@pytest.mark.parametrize("T", SIZES)
def test_the_bound_still_decides_what_it_should(T):
    """Not vacuous: a candidate whose trees all say 0 is decided as soon as the remaining trees cannot
    lift it over thre * T -- within a few trees of the exact point."""
    thre = 0.5
    acc = np.zeros(T + 1)
    rem = (T - np.arange(T + 1)).astype(np.float64)
    first = int(np.argmax(acc + rem < bound(thre, T)))
    assert T - first in (int(np.ceil(thre * T)) - 1, int(np.ceil(thre * T)) - 2, int(np.ceil(thre * T)))
