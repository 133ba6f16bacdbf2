"""GPU parity of the witness refresh (pg_*_values_batch): the variable assignments of a batched gadget call and nothing else.

The reference's prover builds a circuit, preprocesses it and then rebuilds the SAME circuit with other witnesses after
clear_witness() (/root/reference/tests/scalar_gadgets_tests.rs:108-119, 168-177); its verifier builds the rows from other
witnesses altogether (:36 vs :43).  So: values == the CPU oracle's var_values for the new witnesses, limb for limb; the row
columns of a circuit built from OTHER witnesses stay untouched (they are compared with the oracle's rows afterwards) and
together with the refreshed assignments satisfy every gate (pg_check_rows); canaries around the table are intact."""
import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

ROWS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o")
Q = synth.Q
CANARY = 0x5A5A5A5A5A5A5A5A


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def u64(t):
    return t.cpu().numpy().view(np.uint64)


def guarded(n_vars):
    """a variable table with 64 canary scalars on either side"""
    buf = torch.full((n_vars + 128, 4), CANARY, dtype=torch.int64, device="cuda:0")
    return buf, buf[64:64 + n_vars]


def assert_guards(buf, n_vars):
    assert bool((buf[:64] == CANARY).all()) and bool((buf[64 + n_vars:] == CANARY).all()), "a store outside the variable table"


def mixed(mn, mx, n, seed):
    span = max(mx - mn, 1)
    inside = [mn + int(v) % span for v in synth.splitmix64(n, seed)]
    arr = np.concatenate([synth.scalars_from_ints(inside), synth.random_scalars(n, seed + 1),
                          synth.scalars_from_ints([mn, mx - 1, mx % Q, (mn - 1) % Q, 0, Q - 1])])
    return np.ascontiguousarray(arr[np.argsort(synth.splitmix64(len(arr), seed + 2), kind="stable")])


@pytest.mark.parametrize("mn,mx,n", [(0, 2**64, 40), (50_000, 250_000, 33), (2**126, 2**127 + 1, 9), (0, 2**254, 45), (0, 2, 3),
                                     (1, Q - 1, 5)])
def test_range_check_values(engine, mn, mx, n):
    """build with witnesses A (full call), refresh with witnesses B (values only): rows unchanged == the oracle's rows for B (the
    structure is witness-independent), assignments == the oracle's for B, every gate satisfied"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    wa, wb = mixed(mn, mx, n, 3), mixed(mn, mx, n, 17)
    bmn, bmx = pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx)
    cols, _ = engine.range_check_batch(bmn, bmx, dev(wa), 3, 5)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wb)
    assert ora["satisfied"]
    buf, table = guarded(ora["n_vars"])
    engine.range_check_values_batch(bmn, bmx, dev(wb), table)
    torch.cuda.synchronize()
    assert_guards(buf, ora["n_vars"])
    assert np.array_equal(u64(table), ora["var_values"]), "refreshed assignments differ from the oracle's"
    got = cols.to_numpy()
    for k in ROWS:
        assert np.array_equal(got[k], ora[k]), k  # rows written for witnesses A are the rows of witnesses B
    cols.var_values.copy_(table)
    assert engine.check_rows(cols, 5, 0) == -1
    # ... and equal to what the full call writes for B
    full, _ = engine.range_check_batch(bmn, bmx, dev(wb), 3, 5)
    torch.cuda.synchronize()
    assert torch.equal(full.var_values, table)


@pytest.mark.parametrize("shift", [1, 2, 3])
@pytest.mark.parametrize("mn,mx,n", [(50_000, 250_000, 70), (0, 2**254, 41), (0, 2, 67)])
def test_range_check_values_at_every_line_alignment(engine, mn, mx, n, shift):
    """the same refresh into a table that starts 32 / 64 / 96 bytes into a 128-byte line (a composer appending behind other
    variables): several tiles, canaries on both sides"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    w = mixed(mn, mx, n, 29 + shift)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), w)
    nv = ora["n_vars"]
    buf = torch.full((nv + 128 + shift, 4), CANARY, dtype=torch.int64, device="cuda:0")
    table = buf[64 + shift:64 + shift + nv]
    engine.range_check_values_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), dev(w), table)
    torch.cuda.synchronize()
    assert bool((buf[:64 + shift] == CANARY).all()) and bool((buf[64 + shift + nv:] == CANARY).all())
    assert np.array_equal(u64(table), ora["var_values"])


def test_values_calls_touch_no_row_pointer(engine):
    """the C entry point takes no row columns at all; through it, 5000 items at tile edges (32-item tiles: 4999, 5000, 5001)"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    mn, mx = 50_000, 250_000
    bmn, bmx = pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx)
    for n in (4999, 5001):
        w = np.ascontiguousarray(np.concatenate([synth.scalars_from_ints([mn + int(v) % (mx - mn) for v in synth.splitmix64(n // 2, n)]),
                                                 synth.random_scalars(n - n // 2, n + 1)]))
        ora = po.range_check_fast(synth.mont(mn), synth.mont(mx), w, threads=8)
        buf, table = guarded(ora["n_vars"])
        engine.range_check_values_batch(bmn, bmx, dev(w), table)
        torch.cuda.synchronize()
        assert_guards(buf, ora["n_vars"])
        assert np.array_equal(u64(table), ora["var_values"])


@pytest.mark.parametrize("mx,count", [(200, 20), (2**128 - 1, 9), (2**253 + 5, 7), (2, 5), (1, 3), (0, 3)])
def test_max_bound_values(engine, mx, count):
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    below = [int(v) % max(mx, 1) for v in synth.splitmix64(count, 5)]
    w = np.concatenate([synth.scalars_from_ints(below), synth.random_scalars(count, 6)])
    bounds = np.tile(synth.scalars_from_ints([mx]), (len(w), 1))
    ora = po.max_bound_batch(bounds, w)
    buf, table = guarded(ora["n_vars"])
    engine.max_bound_values_batch(pg.BlsScalar.from_int(mx), dev(w), table)
    torch.cuda.synchronize()
    assert_guards(buf, ora["n_vars"])
    assert np.array_equal(u64(table), ora["var_values"])


def test_max_bound_ragged_values(engine):
    """per-item public bounds: the plan (structure) is made once and reused, the witnesses change"""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    batch = 700
    mr_np, wa = bench.c4_inputs(batch, seed=0xC4)
    wb = np.ascontiguousarray(np.roll(wa, 7, axis=0))  # other witnesses under the same bounds
    mr = dev(mr_np)
    nb, roff, voff = engine.ragged_buffers(batch)
    lay = engine.max_bound_ragged_plan(mr, nb, roff, voff)
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0", 3, 5)
    engine.max_bound_ragged_emit(mr, dev(wa), nb, roff, voff, cols, None, 3, 5)
    ora = po.max_bound_batch(mr_np, wb)
    assert (ora["n_gates"], ora["n_vars"]) == (lay.n_gates, lay.n_vars)
    buf, table = guarded(lay.n_vars)
    engine.max_bound_ragged_values(mr, dev(wb), nb, roff, voff, table)
    torch.cuda.synchronize()
    assert_guards(buf, lay.n_vars)
    assert np.array_equal(u64(table), ora["var_values"])
    got = cols.to_numpy()
    for k in ROWS:
        assert np.array_equal(got[k], ora[k]), k
    cols.var_values.copy_(table)
    assert engine.check_rows(cols, 5, 0) == -1


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_max_bound_ragged_values_every_ladder_length_every_alignment(engine, shift):
    """the region sweep writes every 128-byte line in one piece and hands the lines between runs, items and tiles to one pass
    each: ladders of 1 ... 254 bits side by side (runs shorter than a half, than a line), the table starting 0 / 32 / 64 / 96
    bytes into a line, canaries on both sides"""
    from oracle import pyoracle as po
    batch = 333
    bits = [1 + int(x) % 254 for x in synth.splitmix64(batch, 77 + shift)]
    bounds = [max(1, int(x) % (1 << b)) for x, b in zip(synth.splitmix64(batch, 78), bits)]
    bounds[:6] = [1, 2, 3, 4, 5, (1 << 254) - 1]
    mr_np = synth.scalars_from_ints(bounds)
    wit = synth.random_scalars(batch, 79)
    wit[::2] = synth.scalars_from_ints([int(x) % b for x, b in zip(synth.splitmix64(batch, 80)[::2], bounds[::2])])
    mr = dev(mr_np)
    nb, roff, voff = engine.ragged_buffers(batch)
    lay = engine.max_bound_ragged_plan(mr, nb, roff, voff)
    ora = po.max_bound_batch(mr_np, wit)
    assert ora["n_vars"] == lay.n_vars
    buf = torch.full((lay.n_vars + 128 + shift, 4), CANARY, dtype=torch.int64, device="cuda:0")
    table = buf[64 + shift:64 + shift + lay.n_vars]
    engine.max_bound_ragged_values(mr, dev(wit), nb, roff, voff, table)
    torch.cuda.synchronize()
    assert bool((buf[:64 + shift] == CANARY).all()) and bool((buf[64 + shift + lay.n_vars:] == CANARY).all())
    assert np.array_equal(u64(table), ora["var_values"])


@pytest.mark.parametrize("batch,zeros", [(1, ()), (129, (0, 127, 128)), (5003, (0, 63, 64, 255, 256, 257, 1023, 1024, 4095, 4096, 4992, 5002))])
def test_scalar_mix_values(engine, batch, zeros):
    """the fused mix's refresh plans again (an item's shape depends on its witness): prefix sums, error mask and totals == the
    oracle's for the new witnesses; assignments limb for limb; same shape as the built circuit => its rows still hold"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    import test_gpu_gadgets as tg
    va = tg.mix_inputs(batch, 41, zeros)
    vb = tg.mix_inputs(batch, 77, zeros)  # other witnesses, the same items failing: the same circuit
    cols, _, _, _, lay = engine.scalar_mix_batch(*[dev(x) for x in va], 3, 5, zero_var=0)
    ora = po.scalar_mix_batch(*vb)
    assert (lay.n_gates, lay.n_vars) == (ora["n_gates"], ora["n_vars"])
    roff = torch.full((batch + 1,), -1, dtype=torch.int64, device="cuda:0")
    voff = torch.full((batch + 1,), -1, dtype=torch.int64, device="cuda:0")
    err = torch.full((batch,), 7, dtype=torch.uint8, device="cuda:0")
    buf, table = guarded(15 * batch)  # the worst case: the totals are an output
    engine.scalar_mix_values(*[dev(x) for x in vb], roff, voff, table, err)
    torch.cuda.synchronize()
    got_lay, nerr = engine.plan_result()
    assert (got_lay.n_gates, got_lay.n_vars, nerr) == (ora["n_gates"], ora["n_vars"], len(zeros))
    assert_guards(buf, 15 * batch)
    assert np.array_equal(u64(table)[:ora["n_vars"]], ora["var_values"])
    assert bool((table[ora["n_vars"]:] == CANARY).all())
    assert err.cpu().numpy().tolist() == ora["err_mask"].tolist()
    e = np.cumsum(np.concatenate([[0], ora["err_mask"].astype(np.int64)]))
    assert np.array_equal(roff.cpu().numpy(), 10 * np.arange(batch + 1) - 2 * e)
    assert np.array_equal(voff.cpu().numpy(), 15 * np.arange(batch + 1) - 2 * e)
    got = cols.to_numpy()
    for k in ROWS:
        assert np.array_equal(got[k], ora[k]), k
    # another witness set with ANOTHER item failing is another circuit: the plan says so
    vc = tg.mix_inputs(batch, 78, tuple(z for z in zeros[1:]) + ((batch - 1,) if batch - 1 not in zeros else ()))
    engine.scalar_mix_values(*[dev(x) for x in vc], roff, voff, table, err)
    torch.cuda.synchronize()
    assert err.cpu().numpy().tolist() != ora["err_mask"].tolist() or batch == 1 and not zeros


def test_values_argument_checks(engine):
    import ctypes as C
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**64)
    w = dev(synth.random_scalars(4, 1))
    t = torch.empty((4 * 654, 4), dtype=torch.int64, device="cuda:0")
    st = engine._stream()
    assert lib.pg_range_check_values_batch(None, C.byref(mn.c), C.byref(mx.c), w.data_ptr(), 4, t.data_ptr(), st) == 2
    assert lib.pg_range_check_values_batch(engine._h, C.byref(mn.c), C.byref(mx.c), None, 4, t.data_ptr(), st) == 2
    assert lib.pg_range_check_values_batch(engine._h, C.byref(mn.c), C.byref(mx.c), w.data_ptr(), 4, None, st) == 2
    assert lib.pg_range_check_values_batch(engine._h, C.byref(mn.c), C.byref(mx.c), w.data_ptr(), 4, t.data_ptr() + 8, st) == 2
    assert lib.pg_range_check_values_batch(engine._h, C.byref(mn.c), C.byref(mx.c), None, 0, None, st) == 0  # an empty batch
    assert lib.pg_max_bound_values_batch(engine._h, C.byref(mx.c), w.data_ptr(), 4, None, st) == 2
    assert lib.pg_max_bound_ragged_values_batch(engine._h, w.data_ptr(), w.data_ptr(), 4, None, None, None, t.data_ptr(), st) == 2
    assert lib.pg_scalar_mix_values_batch(engine._h, w.data_ptr(), w.data_ptr(), w.data_ptr(), w.data_ptr(), None, 4, None, None, None,
                                          t.data_ptr(), st) == 2
