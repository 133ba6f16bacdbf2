"""GPU: BASELINE configs C3 and C4 at their FULL size (2^20 items, one launch), through size-independent properties --
the layout totals, every row's gate equation over the emitted variable table (pg_check_rows, on the device), the
prefix sums -- plus items sampled across the whole batch (the irregular ones included) against the CPU oracle, limb for
limb.  The C2 counterpart is tests/test_gpu_range_check.py::test_config_c2_full_size_properties."""
import os
import sys

import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEL = ("q_m", "q_l", "q_r", "q_o", "q_c")
WIRES = ("w_l", "w_r", "w_o")
BATCH = 1 << 20


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


def compare_items(cols, roff, voff, idx, ora, o_rows, o_vars, var_base=5):
    """item idx[s] of the device batch == item s of the oracle's batch: selectors and values limb for limb, wires up to
    the two batches' different numbering (Variable 0 = zero_var stays 0 in both)"""
    orow = np.concatenate([[0], np.cumsum(o_rows)]).astype(np.int64)
    ovar = np.concatenate([[0], np.cumsum(o_vars)]).astype(np.int64)
    for s, i in enumerate(idx):
        r0, r1, v0, v1 = int(roff[i]), int(roff[i + 1]), int(voff[i]), int(voff[i + 1])
        assert (r1 - r0, v1 - v0) == (int(o_rows[s]), int(o_vars[s])), (i, r1 - r0, v1 - v0, o_rows[s], o_vars[s])
        for name in SEL:
            assert np.array_equal(u64(getattr(cols, name)[r0:r1]), ora[name][orow[s]:orow[s + 1]]), (name, i)
        assert np.array_equal(u64(cols.var_values[v0:v1]), ora["var_values"][ovar[s]:ovar[s + 1]]), ("var_values", i)
        for name in WIRES:
            got = u64(getattr(cols, name)[r0:r1]).astype(np.int64)
            exp = ora[name][orow[s]:orow[s + 1]].astype(np.int64)
            zero = exp == 0
            assert np.array_equal(got[zero], exp[zero]), (name, i, "zero_var")
            assert np.array_equal(got[~zero] - (var_base + v0), exp[~zero] - (5 + ovar[s])), (name, i)


@pytest.mark.parametrize("form", ["planned", "two_step"])
def test_config_c3_full_size_properties(engine, form):
    """BASELINE config 3 at full size: 2^20 fused items in one launch, with items whose v is 0 sprinkled in (alone, and
    a run of 70 in a row) so that tiles of every kind occur: all items complete (the uniform fast path), some stopped
    early, all stopped early.  `planned` is the call bench.py times (pg_scalar_mix_planned_batch: worst-case buffers, the
    totals read afterwards with pg_plan_result); `two_step` is pg_scalar_mix_plan followed by pg_scalar_mix_batch."""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    v, y, s, a, b = bench.mix_inputs(BATCH, seed=0xC3)
    zeros = sorted(set(range(7, BATCH, 4099)) | set(range(500_000, 500_070)) | {0, BATCH - 1})
    v[zeros] = 0
    ins = [dev(x) for x in (v, y, s, a, b)]
    _, roff, voff = engine.ragged_buffers(BATCH)
    roff.fill_(-1)
    voff.fill_(-1)
    err = torch.full((BATCH,), 7, dtype=torch.uint8, device="cuda:0")
    res = torch.empty((BATCH, 2), dtype=torch.int64, device="cuda:0")
    if form == "planned":
        cols = pg.Columns.allocate(10 * BATCH, 15 * BATCH, "cuda:0", 3, 5)  # the worst case: the layout is not known yet
        for name in SEL + WIRES + ("var_values",):
            getattr(cols, name).fill_(-1)  # a slot nobody writes would fail the row check / the comparison
        engine.scalar_mix_planned(*ins, roff, voff, cols, res, err, 3, 5, 0)
        torch.cuda.synchronize()
        lay, nerr = engine.plan_result()
        # what lies beyond the totals was not touched
        assert bool((cols.q_m[lay.n_gates:] == -1).all()) and bool((cols.var_values[lay.n_vars:] == -1).all())
        cols = pg.Columns(**{n: getattr(cols, n)[:lay.n_gates] for n in SEL + WIRES}, var_values=cols.var_values[:lay.n_vars],
                          gate_base=3, var_base=5)
    else:
        lay, nerr = engine.scalar_mix_plan(ins[0], roff, voff, err)
    assert nerr == len(zeros)
    assert (lay.n_gates, lay.n_vars) == (10 * BATCH - 2 * nerr, 15 * BATCH - 2 * nerr)
    is_zero = torch.zeros((BATCH,), dtype=torch.bool, device="cuda:0")
    is_zero[torch.tensor(zeros, device="cuda:0")] = True
    assert bool((err.bool() == is_zero).all()) and int(err.max()) == 1
    # the prefix sums, item by item
    assert bool((roff[1:] - roff[:-1] == torch.where(is_zero, 8, 10)).all()) and int(roff[0]) == 0 and int(roff[-1]) == lay.n_gates
    assert bool((voff[1:] - voff[:-1] == torch.where(is_zero, 13, 15)).all()) and int(voff[0]) == 0 and int(voff[-1]) == lay.n_vars
    if form == "two_step":
        cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0", 3, 5)
        for name in SEL + WIRES + ("var_values",):
            getattr(cols, name).fill_(-1)
        engine.scalar_mix_emit(*ins, roff, voff, cols, res, 3, 5, 0)
        torch.cuda.synchronize()
    # every one of the 10.5 M rows satisfies its gate equation over the emitted variable table
    assert engine.check_rows(cols, var_base=5, zero_var=0) == -1
    # the five inputs are each item's first five variables
    first = voff[:-1]
    for k, t in enumerate(ins):
        assert bool((cols.var_values[first + k] == t).all()), k
    # result Variables: select_one's output and maybe_equal's
    nz = torch.where(is_zero, 1, 3)
    assert bool((res[:, 0] == 5 + first + 5 + nz + 3).all()) and bool((res[:, 1] == 5 + first + 5 + nz + 6).all())
    # maybe_equal's outcome: a == b for about half of the items
    same = torch.from_numpy((a == b).all(axis=1)).to("cuda:0")
    one = torch.tensor(np.array(synth.mont(1), dtype=np.uint64).view(np.int64), device="cuda:0")
    yeq = cols.var_values[res[:, 1] - 5]
    assert bool(((yeq == one).all(dim=1) == same).all()) and 0.4 < float(same.float().mean()) < 0.6
    # sampled items, the irregular ones and their neighbours included, limb for limb against the oracle
    idx = sorted(set([0, 1, 6, 7, 8, 63, 64, 65, 4105, 4106, 4107, 499_999, 500_000, 500_001, 500_063, 500_064, 500_069, 500_070,
                      BATCH - 2, BATCH - 1] + [int(x) % BATCH for x in synth.splitmix64(40, 17)]))
    ora = po.scalar_mix_batch(*[np.ascontiguousarray(x[idx]) for x in (v, y, s, a, b)])
    assert ora["satisfied"]
    o_err = ora["err_mask"].astype(bool)
    assert o_err.tolist() == [i in set(zeros) for i in idx]
    compare_items(cols, roff.cpu().numpy(), voff.cpu().numpy(), idx, ora, np.where(o_err, 8, 10), np.where(o_err, 13, 15))
    del cols
    torch.cuda.empty_cache()


@pytest.mark.parametrize("form", ["async_plan", "sync_plan"])
def test_config_c4_full_size_properties(engine, form):
    """BASELINE config 4 at full size: 2^20 x max_bound with random 253-bit bounds (ladder length from the bound, ragged
    rows), 115.8 GB of columns in one launch.  `async_plan` is what bench.py times: pg_max_bound_ragged_plan_async and
    pg_max_bound_ragged_batch back to back without a host round trip (buffers sized from an earlier plan, as the bench
    does); `sync_plan` the plan that returns its totals."""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    free, _ = torch.cuda.mem_get_info()
    if free < BATCH * (515 * 184 + 517 * 32) + (16 << 30):
        pytest.skip("not enough free HBM for the full-size batch")
    mr_np, wt_np = bench.c4_inputs(BATCH, seed=0xC4)
    mr, wt = dev(mr_np), dev(wt_np)
    nb, roff, voff = engine.ragged_buffers(BATCH)
    lay = engine.max_bound_ragged_plan(mr, nb, roff, voff)
    n64 = nb.to(torch.int64)
    assert int(n64.min()) == 2 and 252 <= int(n64.max()) <= 254  # bounds 0, 1, 2 give the shortest ladder
    assert (lay.n_gates, lay.n_vars) == (int((2 * n64 + 5).sum()), int((n64 + 262).sum()))
    assert bool((roff[1:] - roff[:-1] == 2 * n64 + 5).all()) and int(roff[0]) == 0 and int(roff[-1]) == lay.n_gates
    assert bool((voff[1:] - voff[:-1] == n64 + 262).all()) and int(voff[0]) == 0 and int(voff[-1]) == lay.n_vars
    # every output array 64 entries longer than the layout and filled with a sentinel: a slot nobody writes fails the row check
    # below, a store beyond the layout shows in the tails
    GUARD = 64
    big = pg.Columns.allocate(lay.n_gates + GUARD, lay.n_vars + GUARD, "cuda:0", 3, 5)
    for name in SEL + WIRES + ("var_values",):
        getattr(big, name).fill_(-1)
    cols = pg.Columns(**{n: getattr(big, n)[:lay.n_gates] for n in SEL + WIRES}, var_values=big.var_values[:lay.n_vars],
                      gate_base=3, var_base=5)
    res = torch.full((BATCH + GUARD,), -1, dtype=torch.int64, device="cuda:0")[:BATCH]
    if form == "async_plan":  # plan again, this time without the round trip, straight into the emit call
        nb.fill_(0)
        roff.fill_(-1)
        voff.fill_(-1)
        engine.max_bound_ragged_plan_async(mr, nb, roff, voff)
    engine.max_bound_ragged_emit(mr, wt, nb, roff, voff, cols, res, 3, 5)
    torch.cuda.synchronize()
    if form == "async_plan":
        lay2, nerr2 = engine.plan_result()
        assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, 0)
        assert bool((nb.to(torch.int64) == n64).all())
        assert bool((roff[1:] - roff[:-1] == 2 * n64 + 5).all()) and int(roff[0]) == 0 and int(roff[-1]) == lay.n_gates
        assert bool((voff[1:] - voff[:-1] == n64 + 262).all()) and int(voff[0]) == 0 and int(voff[-1]) == lay.n_vars
    for name in SEL + WIRES:  # nothing beyond the layout was touched
        assert bool((getattr(big, name)[lay.n_gates:] == -1).all()), name
    assert bool((big.var_values[lay.n_vars:] == -1).all())
    # every one of the 5.3e8 rows satisfies its gate equation over the emitted variable table
    assert engine.check_rows(cols, var_base=5) == -1
    # witness = the item's first variable, result = its last
    assert bool((cols.var_values[voff[:-1]] == wt).all())
    assert bool((res == 5 + voff[1:] - 1).all())
    # q_c of an item's first row is mont(bound - 1): the one data-dependent selector
    qc = u64(cols.q_c[roff[:-1][:4096]])
    exp = synth.scalars_from_ints([(synth.to_int(m) - 1) % synth.Q for m in mr_np[:4096]])
    assert np.array_equal(qc, exp)
    # outcomes: the ladder accepts iff (bound - 1 - w) mod q fits its n bits; the pool alternates inside / anywhere
    one = torch.tensor(np.array(synth.mont(1), dtype=np.uint64).view(np.int64), device="cuda:0")
    acc = (cols.var_values[res - 5] == one).all(dim=1)
    assert 0.55 < float(acc.float().mean()) < 0.85  # half inside their bound, the rest accepted when the wrapped difference fits n bits
    idx = sorted(set([0, 1, 2, 3, 15, 16, 17, 4095, 4096, 4097, BATCH // 2, BATCH - 2, BATCH - 1] +
                     [int(x) % BATCH for x in synth.splitmix64(12, 23)]))
    ora = po.max_bound_batch(np.ascontiguousarray(mr_np[idx]), np.ascontiguousarray(wt_np[idx]))
    assert ora["satisfied"]
    ns = ora["num_bits"].astype(np.int64)
    assert ns.tolist() == nb.cpu().numpy()[idx].tolist()
    compare_items(cols, roff.cpu().numpy(), voff.cpu().numpy(), idx, ora, 2 * ns + 5, ns + 262)
    assert u64(acc[torch.tensor(idx, device="cuda:0")].to(torch.int64)).tolist() == [
        int(synth.to_int(ora["var_values"][int(r) - 5]) == 1) for r in ora["result_vars"]]
    del cols, big
    torch.cuda.empty_cache()


def test_prefix_sums_above_the_single_pass_limit(engine):
    """more than 4096 plan blocks (4 M items): the block sums go through scan_top_kernel first; offsets == a cumulative
    sum computed by torch"""
    batch = 4096 * 1024 + 1537
    vals = torch.ones((batch, 4), dtype=torch.int64, device="cuda:0")
    zeros = torch.arange(0, batch, 1013, device="cuda:0")
    vals[zeros] = 0
    import ctypes as C
    from plonk_gadgets_amd import _lib
    roff = torch.empty((batch + 1,), dtype=torch.int64, device="cuda:0")
    voff = torch.empty((batch + 1,), dtype=torch.int64, device="cuda:0")
    lay, nerr = _lib.LayoutC(), C.c_uint64()
    st = engine._lib.pg_is_non_zero_plan(engine._h, vals.data_ptr(), batch, roff.data_ptr(), voff.data_ptr(), None, C.byref(lay),
                                         C.byref(nerr), engine._stream())
    assert st == 1 and nerr.value == zeros.numel()  # PG_ERR_NON_EXISTING_INVERSE: some items have no inverse
    counts = torch.full((batch,), 3, dtype=torch.int64, device="cuda:0")
    counts[zeros] = 1
    exp = torch.cumsum(counts, 0)
    assert int(roff[0]) == 0 and bool((roff[1:] == exp).all()) and bool((voff[1:] == exp).all())
    assert lay.n_gates == int(exp[-1]) == lay.n_vars


def test_calls_that_move_to_another_stream(engine):
    """the engine's scratch is shared by consecutive calls: a caller that changes stream between two calls (no
    synchronisation of its own) is ordered behind the work still in flight on the previous stream"""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    n = 1 << 16
    v, y, s, a, b = bench.mix_inputs(n, seed=5)
    ins = [dev(x) for x in (v, y, s, a, b)]
    small = [dev(x) for x in bench.mix_inputs(300, seed=6)]
    ora_small = po.scalar_mix_batch(*bench.mix_inputs(300, seed=6))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            big = engine.scalar_mix_batch(*ins, 3, 5, zero_var=0)      # 131 072 inversions on the side stream
        with torch.cuda.stream(s2):
            out = engine.scalar_mix_batch(*small, 3, 5, zero_var=0)    # same scratch, another stream, no sync
        torch.cuda.synchronize()
        got = out[0].to_numpy()
        for k in SEL + WIRES + ("var_values",):
            assert np.array_equal(got[k], ora_small[k]), k
        assert engine.check_rows(big[0], var_base=5, zero_var=0) == -1


def test_a_call_after_its_predecessors_stream_was_destroyed(engine):
    """the header lets a caller destroy a stream it has used: the engine only ever COMPARES the previous stream's handle (the
    event that orders a later call behind it was recorded on that stream when the earlier call ended) -- a big call on a
    raw HIP stream that is destroyed at once (its work still pending), then a call on another stream, three times over so
    that a recycled handle occurs too"""
    import ctypes as C
    import bench
    from oracle import pyoracle as po
    from plonk_gadgets_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes, hip.hipStreamDestroy.argtypes = [C.POINTER(C.c_void_p)], [C.c_void_p]
    lib = _lib.load()
    n = 1 << 16
    ins = [dev(x) for x in bench.mix_inputs(n, seed=5)]
    small_np = bench.mix_inputs(300, seed=6)
    small = [dev(x) for x in small_np]
    ora_small = po.scalar_mix_batch(*small_np)
    _, roff, voff = engine.ragged_buffers(n)
    import plonk_gadgets_amd as pg
    big = pg.Columns.allocate(10 * n, 15 * n, "cuda:0", 3, 5)
    bc = big.as_c()
    torch.cuda.synchronize()
    for _ in range(3):
        raw = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(raw)) == 0
        st = lib.pg_scalar_mix_planned_batch(engine._h, *[t.data_ptr() for t in ins], n, roff.data_ptr(), voff.data_ptr(), None, 3, 5, 0,
                                             C.byref(bc), None, raw)
        assert st == 0, lib.pg_last_error()
        assert hip.hipStreamDestroy(raw) == 0  # no synchronisation: the launches are still in flight
        out = engine.scalar_mix_batch(*small, 3, 5, zero_var=0)  # same scratch, torch's current stream
        torch.cuda.synchronize()
        got = out[0].to_numpy()
        for k in SEL + WIRES + ("var_values",):
            assert np.array_equal(got[k], ora_small[k]), k
        lay, nerr = engine.plan_result()
    assert engine.check_rows(pg.Columns(**{k: getattr(big, k) for k in SEL + WIRES}, var_values=big.var_values, gate_base=3, var_base=5),
                             var_base=5, zero_var=0) == -1


@pytest.mark.parametrize("batch", [1, 1023, 1025, 65 * 1024 + 7, 2_000_000, 2_500_000])
def test_plans_against_a_host_prefix_sum(engine, batch):
    """the plans' prefix sums -- one launch with a decoupled look-back up to 2048 blocks of 1024 items, the two-level scan
    beyond (2.5 M items) -- against numpy's cumsum, for the mix (zeros sprinkled everywhere, at block edges too) and for the
    ragged max_bound; totals and error count through the result record; twice, so that the second launch finds what the
    first one left behind"""
    rng = np.random.default_rng(batch)
    v = np.zeros((batch, 4), dtype=np.uint64)
    v[:, 0] = 1
    zeros = np.unique(np.concatenate([rng.integers(0, batch, size=min(batch, 1000)), np.array([0, batch - 1]),
                                      np.arange(1023, batch, 1024)[:64], np.arange(1024, batch, 1024)[:64]]))
    v[zeros] = 0
    dv = dev(v)
    _, roff, voff = engine.ragged_buffers(batch)
    err = torch.zeros((batch,), dtype=torch.uint8, device="cuda:0")
    rows = np.where((v == 0).all(axis=1), 8, 10).astype(np.uint64)
    vars_ = np.where((v == 0).all(axis=1), 13, 15).astype(np.uint64)
    for _ in range(2):
        roff.fill_(-1); voff.fill_(-1)
        lay, nerr = engine.scalar_mix_plan(dv, roff, voff, err)
        assert nerr == len(zeros) and (lay.n_gates, lay.n_vars) == (int(rows.sum()), int(vars_.sum()))
        assert np.array_equal(roff.cpu().numpy().view(np.uint64), np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64))
        assert np.array_equal(voff.cpu().numpy().view(np.uint64), np.concatenate([[0], np.cumsum(vars_)]).astype(np.uint64))
        assert np.array_equal(err.cpu().numpy().nonzero()[0], zeros)
    # asynchronous form, then the result record
    roff.fill_(-1)
    engine.scalar_mix_plan_async(dv, roff, voff)
    torch.cuda.synchronize()
    lay2, nerr2 = engine.plan_result()
    assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, nerr)
    assert int(roff[-1].item()) == lay.n_gates
    if batch <= 2_000_000:
        import bench
        mr, _ = bench.c4_inputs(batch, seed=3)
        nb, roff, voff = engine.ragged_buffers(batch)
        lay = engine.max_bound_ragged_plan(dev(mr), nb, roff, voff)
        n = nb.cpu().numpy().astype(np.uint64)
        assert np.array_equal(roff.cpu().numpy().view(np.uint64), np.concatenate([[0], np.cumsum(2 * n + 5)]).astype(np.uint64))
        assert np.array_equal(voff.cpu().numpy().view(np.uint64), np.concatenate([[0], np.cumsum(n + 262)]).astype(np.uint64))
        assert (lay.n_gates, lay.n_vars) == (int((2 * n + 5).sum()), int((n + 262).sum()))


def test_fuzz_mix_sizes_and_error_densities(engine):
    """the fused mix over random batch sizes (1 .. 70 000: below and above the size where the launches overlap, any
    remainder modulo the tile widths 64 / 256 and the plan's 1024-item blocks) and random densities of failing items,
    every column and the plan's outputs against the faithful oracle (small batches) or against the call's own two-step
    form (large ones) -- the planned call, whose plan runs beside the pre-pass"""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    rng = np.random.default_rng(20261004)
    for trial in range(24):
        batch = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 70000)]))
        density = float(rng.choice([0.0, 0.001, 0.05, 0.5, 1.0]))
        v, y, s, a, b = bench.mix_inputs(batch, seed=1000 + trial)
        zeros = np.nonzero(rng.random(batch) < density)[0]
        v[zeros] = 0
        ins = [dev(x) for x in (v, y, s, a, b)]
        _, roff, voff = engine.ragged_buffers(batch)
        big = pg.Columns.allocate(10 * batch, 15 * batch, "cuda:0")
        res = torch.zeros((batch, 2), dtype=torch.int64, device="cuda:0")
        err = torch.zeros((batch,), dtype=torch.uint8, device="cuda:0")
        engine.scalar_mix_planned(*ins, roff, voff, big, res, err, 3, 5, 0)
        torch.cuda.synchronize()
        lay, nerr = engine.plan_result()
        assert nerr == len(zeros) and np.array_equal(err.cpu().numpy().nonzero()[0], zeros), (trial, batch, density)
        assert (lay.n_gates, lay.n_vars) == (10 * batch - 2 * len(zeros), 15 * batch - 2 * len(zeros))
        got = big.to_numpy()
        if batch <= 5000:
            ora = po.scalar_mix_batch(v, y, s, a, b)
            for k in SEL + WIRES:
                assert np.array_equal(got[k][:lay.n_gates], ora[k]), (trial, batch, density, k)
            assert np.array_equal(got["var_values"][:lay.n_vars], ora["var_values"]), (trial, batch, density)
            assert np.array_equal(res.cpu().numpy().view(np.uint64).reshape(-1, 2), ora["result_vars"].reshape(-1, 2))
        else:
            _, roff2, voff2 = engine.ragged_buffers(batch)
            lay2, nerr2 = engine.scalar_mix_plan(ins[0], roff2, voff2)
            ref = pg.Columns.allocate(lay2.n_gates, lay2.n_vars, "cuda:0")
            res2 = torch.zeros((batch, 2), dtype=torch.int64, device="cuda:0")
            engine.scalar_mix_emit(*ins, roff2, voff2, ref, res2, 3, 5, 0)
            torch.cuda.synchronize()
            assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, nerr)
            assert torch.equal(roff, roff2) and torch.equal(voff, voff2) and torch.equal(res, res2)
            exp = ref.to_numpy()
            for k in SEL + WIRES + ("var_values",):
                n = lay.n_vars if k == "var_values" else lay.n_gates
                assert np.array_equal(got[k][:n], exp[k]), (trial, batch, density, k)
            assert engine.check_rows(ref, var_base=5, zero_var=0) == -1
