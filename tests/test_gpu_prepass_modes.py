"""GPU: the two homes of a call's inverses agree.  A call of 2^11 ... 2^18 - 1 inverted elements runs the inversion pre-pass BESIDE
the emitter (inverses written in place, the emitter skips their slots); smaller and bigger calls run it FIRST and the emitter takes
the inverses from the pre-pass's dense output (csrc/invert.hpp, EmitOut::inv_in_place).  Items are independent, so a call of N items
and a call of N - 1 of the same items must agree on every byte of the common prefix -- with N on one side of a threshold and N - 1
on the other that compares the two paths with each other at sizes the CPU oracle would take minutes for; around the lower threshold
the oracle itself is the judge."""
import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu
Q = synth.Q
NAMES = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def same_prefix(small, big, rows, nvars):
    for n in NAMES[:8]:
        assert torch.equal(getattr(small, n)[:rows], getattr(big, n)[:rows]), n
    assert torch.equal(small.var_values[:nvars], big.var_values[:nvars]), "var_values"


def half_in_range(n, bound, seed):
    """every other witness below the bound, the others uniform field elements (their block ends in a non-zero u: an inverse)"""
    w = synth.random_scalars(n, seed)
    inside = synth.scalars_from_ints([int(x) % max(bound, 1) for x in synth.splitmix64((n + 1) // 2, seed + 1)])
    w[::2] = inside
    return w


@pytest.mark.parametrize("n_big", [2048, 1 << 18])
def test_maybe_equal_and_is_non_zero_across_the_threshold(engine, n_big):
    a = synth.random_scalars(n_big, 3)
    b = synth.random_scalars(n_big, 4)
    b[::3] = a[::3]
    av = torch.arange(n_big, dtype=torch.int64, device="cuda:0") + 7
    bv = av + n_big
    outs = []
    for n in (n_big - 1, n_big):
        cols, res = engine.maybe_equal_batch(av[:n], dev(a[:n]), bv[:n], dev(b[:n]), 3, 5)
        torch.cuda.synchronize()
        outs.append((cols, res))
    same_prefix(outs[0][0], outs[1][0], 3 * (n_big - 1), 3 * (n_big - 1))
    assert torch.equal(outs[0][1], outs[1][1][:n_big - 1])
    vals = synth.random_scalars(n_big, 9)
    vals[::5] = 0
    outs = []
    for n in (n_big - 1, n_big):
        cols, err, nerr = engine.is_non_zero_batch(av[:n], dev(vals[:n]), 3, 5, zero_var=0)
        torch.cuda.synchronize()
        outs.append((cols, err, nerr))
    rows = int(outs[0][0].q_m.shape[0])
    same_prefix(outs[0][0], outs[1][0], rows, int(outs[0][0].var_values.shape[0]))
    assert torch.equal(outs[0][1], outs[1][1][:n_big - 1]) and outs[1][2] - outs[0][2] in (0, 1)


@pytest.mark.parametrize("batch", [2047, 2048, 2049])
def test_small_gadgets_around_the_lower_threshold_vs_oracle(engine, batch):
    from oracle import pyoracle as po
    import test_gpu_gadgets as tg
    a, b = tg.pair_inputs(batch, 21, equal_every=2)
    exp, res, g0, v0, av, bv = tg.oracle_two_input("maybe_equal", a, b, True)
    cols, got = engine.maybe_equal_batch(dev(np.array(av, np.uint64)), dev(a), dev(np.array(bv, np.uint64)), dev(b), g0, v0)
    torch.cuda.synchronize()
    tg.assert_cols(cols.to_numpy(), exp)
    vals = synth.random_scalars(batch, 31)
    vals[::7] = 0
    c = po.Composer()
    vars_ = [c.add_input(v) for v in vals]
    g0, v0 = c.n, c.num_vars
    errs = [int(c.L.is_non_zero(c.c, vars_[i], po.fr(vals[i]))) for i in range(batch)]
    exp = c.export(g0, v0)
    cols, err, nerr = engine.is_non_zero_batch(dev(np.array(vars_, np.uint64)), dev(vals), g0, v0, zero_var=0)
    torch.cuda.synchronize()
    tg.assert_cols(cols.to_numpy(), exp)
    assert err.cpu().numpy().tolist() == errs


def test_range_check_across_the_upper_threshold(engine):
    """2 inverses per item: 2^17 items = 2^18 elements (dense) against 2^17 - 1 (in place); n = 3"""
    import plonk_gadgets_amd as pg
    n_big = 1 << 17
    mn, mx = pg.BlsScalar.from_int(1), pg.BlsScalar.from_int(7)
    wit = half_in_range(n_big, 7, 41)
    lay = engine.range_check_layout(mn, mx, 1)
    G, V = lay.gates_per_item, lay.vars_per_item
    outs = []
    for n in (n_big - 1, n_big):
        cols, res = engine.range_check_batch(mn, mx, dev(wit[:n]), 3, 5)
        torch.cuda.synchronize()
        assert engine.check_rows(cols, var_base=5) == -1
        outs.append((cols, res))
    same_prefix(outs[0][0], outs[1][0], G * (n_big - 1), V * (n_big - 1))
    assert torch.equal(outs[0][1], outs[1][1][:n_big - 1])
    # the lower threshold, against the oracle: 1023 / 1024 / 1025 items = 2046 / 2048 / 2050 elements
    from oracle import pyoracle as po
    for n in (1023, 1024, 1025):
        ora = po.range_check_batch(synth.mont(1), synth.mont(7), wit[:n])
        cols, res = engine.range_check_batch(mn, mx, dev(wit[:n]), 3, 5)
        torch.cuda.synchronize()
        got = cols.to_numpy()
        for k in NAMES:
            assert np.array_equal(got[k], ora[k]), (n, k)


def test_max_bound_uniform_and_ragged_across_the_upper_threshold(engine):
    import plonk_gadgets_amd as pg
    n_big = 1 << 18
    wit = half_in_range(n_big, 200, 43)
    outs = []
    for n in (n_big - 1, n_big):
        cols, res, nb = engine.max_bound_batch(pg.BlsScalar.from_int(200), dev(wit[:n]), 3, 5)
        torch.cuda.synchronize()
        assert engine.check_rows(cols, var_base=5) == -1
        outs.append((cols, res, nb))
    G, V = 2 * outs[0][2] + 5, outs[0][2] + 262
    same_prefix(outs[0][0], outs[1][0], G * (n_big - 1), V * (n_big - 1))
    del outs
    torch.cuda.empty_cache()
    # ragged: bounds of 1 ... 12 bits, one per item
    bounds = [1 + int(x) % 4000 for x in synth.splitmix64(n_big, 47)]
    mr = synth.scalars_from_ints(bounds)
    wit = synth.random_scalars(n_big, 49)
    wit[::2] = synth.scalars_from_ints([int(x) % b for x, b in zip(synth.splitmix64(n_big, 51)[::2], bounds[::2])])
    outs = []
    for n in (n_big - 1, n_big):
        cols, res, lay = engine.max_bound_ragged_batch(dev(mr[:n]), dev(wit[:n]), 3, 5)[:3]
        torch.cuda.synchronize()
        assert engine.check_rows(cols, var_base=5) == -1
        outs.append((cols, res, lay))
    rows, nv = int(outs[0][0].q_m.shape[0]), int(outs[0][0].var_values.shape[0])
    same_prefix(outs[0][0], outs[1][0], rows, nv)
