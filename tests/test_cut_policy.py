"""Where the library cuts a forest in two and what it learns from its calls (pk_forest_q.hip: q_pick_cut,
pk_forest_cut_feedback) -- the host logic of the cut, through the device-free hook pk_debug_cut_policy.
A run that only wants the pixels with p > thre (Chromosome.score, peakachu/scoreUtils.py:110-113) lets
the head of the forest decide candidates: one whose partial sum plus 1.0 per remaining tree cannot exceed
thre * T.  The cut goes to the first tree-group boundary where a partial sum of `forest_split_frac` per
mille of the trees walked is already decided; calls that leave too many candidates open move it later,
and it is given up where nothing worth-while would be left behind it."""
import numpy as np

from peakachu_amd import _lib


def policy(trees_in_front, split_sum, open_frac, frac=200):
    L = _lib.load()
    t = np.ascontiguousarray(trees_in_front, np.int32)
    o = np.ascontiguousarray(open_frac, np.float64)
    cuts = np.zeros(max(len(open_frac), 1), np.int32)
    _lib.check(L.pk_debug_cut_policy(t, len(t) - 1, float(split_sum), int(frac), o if len(o) else np.zeros(1), len(open_frac), cuts),
               "pk_debug_cut_policy")
    return cuts[:len(open_frac)].tolist()


# the benchmark forest's groups: twelve of eight trees and one of four
CONFIG2 = [8 * g for g in range(13)] + [100]


def test_first_boundary_where_a_fifth_of_the_trees_walked_is_decided():
    # thre * T - (T - k) >= 0.2 k  <=>  k >= (1 - thre) T / 0.8
    assert policy(CONFIG2, 50.0, [0.04]) == [8]      # 64 trees: 14 of 64 decided
    assert policy(CONFIG2, 60.0, [0.04]) == [7]      # 56 trees (0.6: 50 needed)
    assert policy(CONFIG2, 70.0, [0.04]) == [5]      # 40 trees
    assert policy(CONFIG2, 90.0, [0.04]) == [2]      # 16 trees
    assert policy(CONFIG2, 20.0, [0.04]) == [0]      # 0.2: 100 trees would be needed -- no cut
    assert policy(CONFIG2, 0.0, [0.04]) == [0]       # --minimum-prob 0: nobody is ever decided
    # a stricter rule (a tenth of the trees walked) cuts earlier, a laxer one later
    assert policy(CONFIG2, 50.0, [0.04], frac=100) == [7]
    assert policy(CONFIG2, 50.0, [0.04], frac=400) == [11]


def test_cut_stays_where_few_candidates_stay_open():
    assert policy(CONFIG2, 50.0, [0.04] * 6) == [8] * 6
    assert policy(CONFIG2, 50.0, [0.149] * 4) == [8] * 4


def test_cut_moves_later_when_many_stay_open_and_is_given_up_at_the_end():
    # more than 15 % open: one group later; more than half: two
    assert policy(CONFIG2, 50.0, [0.2, 0.2, 0.05, 0.05]) == [8, 9, 10, 10]
    assert policy(CONFIG2, 50.0, [0.6, 0.1, 0.1]) == [8, 10, 10]
    # untrained random trees: everybody stays open wherever the cut is -- later and later, then not at all
    # (a tail of less than a seventh of the groups does not pay for two launches)
    cuts = policy(CONFIG2, 50.0, [1.0] * 6)
    assert cuts[:2] == [8, 10] and cuts[-1] == 0 and all(b > a or b == 0 for a, b in zip(cuts, cuts[1:]))
    assert 0 not in cuts[:cuts.index(0)] and set(cuts[cuts.index(0):]) == {0}


def test_forest_of_many_small_groups():
    # 500 trees in groups of 14 and 15 (configs[4]): 0.2-rule -> 312.5 trees -> the boundary at 322
    t = [0]
    while t[-1] < 500:
        t.append(min(500, t[-1] + (14 if len(t) % 2 else 15)))
    cut = policy(t, 250.0, [0.01])[0]
    assert t[cut] >= 313 and t[cut - 1] < 313


def test_bad_arguments_are_refused():
    L = _lib.load()
    cuts = np.zeros(1, np.int32)
    assert L.pk_debug_cut_policy(np.zeros(2, np.int32), 0, 50.0, 200, np.zeros(1), 1, cuts) != 0
