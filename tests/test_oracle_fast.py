"""Pins oracle/fast.c (flat, table-driven, threaded) against the faithful restatement oracle/gadgets.c."""
import time

import numpy as np
import pytest

from oracle import pyoracle as po
from plonk_gadgets_amd import synth

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values", "result_vars")


@pytest.mark.parametrize("mn,mx,count,threads", [(0, 2**64, 40, 1), (50_000, 250_000, 33, 3), (0, 2**254, 12, 4),
                                                (2**126, 2**127 + 1, 9, 8), (0, 2, 5, 2)])
def test_fast_equals_faithful(mn, mx, count, threads):
    inside = synth.scalars_from_ints([mn + int(v) % max(mx - mn, 1) for v in synth.splitmix64(count, 5)])
    wit = np.ascontiguousarray(np.concatenate([inside, synth.random_scalars(count, 6)]))
    slow = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    fast = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit, threads=threads, var_base=5)
    for k in COLS:
        assert np.array_equal(slow[k], fast[k]), k


def test_fast_is_fast():
    wit = synth.random_scalars(256, 1)
    t0 = time.perf_counter()
    po.range_check_fast(synth.mont(0), synth.mont(2**254), wit, threads=4)
    assert time.perf_counter() - t0 < 5.0
