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


def _c4_like(count, seed):
    """bounds of every size class (0, 1, 2, small, 253-bit, q - 1) and witnesses below / above them"""
    import random
    rng = random.Random(seed)
    Q = synth.Q
    bounds, wits = [], []
    for i in range(count):
        b = [0, 1, 2, 3, 200, 2**64, Q - 1, 2**253 - 1, 2**254, 2**254 + 5][i] if i < 10 else \
            rng.randrange(1 << rng.randrange(1, 255))
        bounds.append(b)
        wits.append(rng.randrange(b) if (i % 2 == 0 and b > 0) else rng.randrange(Q))
    return synth.scalars_from_ints(bounds), synth.scalars_from_ints(wits)


@pytest.mark.parametrize("count,threads,cuts", [(37, 1, (0, 37)), (64, 3, (0, 5, 6, 40, 64)), (23, 8, (0, 1, 22, 23))])
def test_max_bound_fast_equals_faithful(count, threads, cuts):
    """oracle_max_bound_fast, chunk by chunk at the numbering of the whole batch == oracle_max_bound_batch (gadgets.c:
    allocate + max_bound per item, src/range.rs:82-113), every array"""
    mr, wt = _c4_like(count, seed=count)
    slow = po.max_bound_batch(mr, wt)
    plan = po.max_bound_plan(mr, threads=threads)
    nb, roff, voff = plan
    assert np.array_equal(nb, slow["num_bits"]) and int(roff[-1]) == slow["n_gates"] and int(voff[-1]) == slow["n_vars"]
    assert {2, 252} <= set(nb.tolist()) and int(nb.max()) == 255
    for lo, hi in zip(cuts, cuts[1:]):
        fast = po.max_bound_fast(mr, wt, plan, lo, hi, var_base=5, threads=threads)
        r0, r1, v0, v1 = int(roff[lo]), int(roff[hi]), int(voff[lo]), int(voff[hi])
        for k in COLS[:8]:
            assert np.array_equal(fast[k], slow[k][r0:r1]), (k, lo, hi)
        assert np.array_equal(fast["var_values"], slow["var_values"][v0:v1]), (lo, hi)
        assert np.array_equal(fast["result_vars"], slow["result_vars"][lo:hi])


@pytest.mark.parametrize("count,threads,cuts", [(50, 1, (0, 50)), (77, 4, (0, 3, 4, 50, 77))])
def test_scalar_mix_fast_equals_faithful(count, threads, cuts):
    """oracle_scalar_mix_fast == oracle_scalar_mix_batch (gadgets.c: src/scalar.rs:36-140), with items whose v is zero
    (is_non_zero stops after its first row), a == b and a != b, s = 0 / 1 / anything"""
    import bench
    v, y, s, a, b = bench.mix_inputs(count, seed=count)
    v[[0, 7, 8, count - 1]] = 0
    s[3] = synth.random_scalars(1, 9)[0]
    slow = po.scalar_mix_batch(v, y, s, a, b)
    plan = po.scalar_mix_plan(v)
    roff, voff, err = plan
    assert np.array_equal(err, slow["err_mask"]) and int(roff[-1]) == slow["n_gates"] and int(voff[-1]) == slow["n_vars"]
    for lo, hi in zip(cuts, cuts[1:]):
        fast = po.scalar_mix_fast(v, y, s, a, b, plan, lo, hi, var_base=5, zero_var=0, threads=threads)
        r0, r1, v0, v1 = int(roff[lo]), int(roff[hi]), int(voff[lo]), int(voff[hi])
        for k in COLS[:8]:
            assert np.array_equal(fast[k], slow[k][r0:r1]), (k, lo, hi)
        assert np.array_equal(fast["var_values"], slow["var_values"][v0:v1]), (lo, hi)
        assert np.array_equal(fast["result_vars"], slow["result_vars"][lo:hi])


def test_range_check_fast_into_callers_buffers():
    """the chunked form the exhaustive GPU tests use: a chunk of the batch written into caller-owned buffers at the
    global numbering == the same rows of the whole batch"""
    mn, mx = 50_000, 250_000
    wit = np.ascontiguousarray(np.concatenate([synth.scalars_from_ints([60_000, 49_999, 250_000]), synth.random_scalars(9, 2)]))
    whole = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit, threads=2, var_base=5)
    G, V = 4 * whole["num_bits"] + 11, 2 * whole["num_bits"] + 524
    lo, hi = 5, 11
    bufs = {k: np.full(4 * (hi - lo) * max(G, V) + 8, 0xEE, dtype=np.uint64) for k in COLS[:9]}
    part = po.range_check_fast(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit[lo:hi]), threads=3, var_base=5 + lo * V,
                               out=bufs)
    for k in COLS[:8]:
        assert np.array_equal(part[k], whole[k][lo * G:hi * G]), k
    assert np.array_equal(part["var_values"], whole["var_values"][lo * V:hi * V])
    assert part["q_m"].ctypes.data == bufs["q_m"].ctypes.data and int(bufs["q_m"][-1]) == 0xEE
