"""GPU: pg_composer's command queue.  The reference drives a composer ONE call at a time
(/root/reference/tests/range_gadgets_tests.rs:29-44); here those calls are recorded on the host (Variables are numbered
there) and reach the device as few launches.  What must hold: the columns are those of the same calls on the CPU
oracle's composer -- and of the same calls with recording off -- whatever the interleaving, and the launch counts are
what the header promises."""
import ctypes as C

import numpy as np
import pytest
import torch

import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
Q = synth.Q
S = pg.BlsScalar.from_int


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


def same(dev, ora):
    got, exp = dev.export(), ora.export()
    assert dev.circuit_size() == ora.n and dev.num_variables() == ora.num_vars
    for k in COLS:
        assert got[k].shape == exp[k].shape, k
        assert np.array_equal(got[k], exp[k]), k


def f(x):
    from oracle import pyoracle as po
    return po.fr(synth.mont(x))


def test_reference_loop_is_one_launch(engine):
    """allocate, range_check, one call at a time, 300 times: recorded, then ONE batched launch (queue statistics); same
    columns as the oracle's loop and as the same loop with recording off (one launch per call)"""
    from oracle import pyoracle as po
    vals = [50_000 + 997 * i for i in range(150)] + [int(x) % Q for x in synth.splitmix64(150, 77)]
    mn, mx = 50_000, 250_000
    ora = po.Composer()
    for v in vals:
        ora.L.range_check(ora.c, f(mn), f(mx), ora.allocate(synth.mont(v)))
    exports = []
    for queued in (True, False):
        dev = pg.StandardComposer(engine, 1 << 16, 1 << 18)
        dev.queue(queued)  # turning it off flushes what creation recorded
        _, f0, l0 = dev.queue_stats()
        res = [pg.range_check(dev, S(mn), S(mx), pg.AllocatedScalar.allocate(dev, S(v))) for v in vals]
        if queued:
            assert dev.queue_stats()[0] >= 2 * len(vals)  # nothing has been launched yet
            dev.flush()
            pending, f1, l1 = dev.queue_stats()
            # the composer's initial state (gate calls) + the 300 pairs: two launches in one flush
            assert pending == 0 and f1 - f0 == 1 and l1 - l0 <= 2, (pending, f1 - f0, l1 - l0)
        same(dev, ora)
        assert dev.check() == -1 and ora.check() == -1
        assert [dev.value(r).to_int() for r in res[:150]] == [int(mn <= v < mx) for v in vals[:150]]
        exports.append(dev.export())
    for k in COLS:
        assert np.array_equal(exports[0][k], exports[1][k]), k


@pytest.mark.parametrize("prealloc", [False, True])
@pytest.mark.parametrize("gadget,gates", [("range_check", 1), ("range_check", 2), ("max_bound", 1), ("max_bound", 3)])
def test_reference_loop_with_gates_behind_every_gadget_call(engine, gadget, gates, prealloc):
    """The loop the reference's tests actually run (tests/range_gadgets_tests.rs:29-44): allocate, range_check, then
    constrain_to_constant on the result -- a gate between any two gadget calls.  A flush sends the gadget calls of such a loop out as ONE
    launch whose items lie `gates` rows apart (csrc/emit.hpp, EmitOut::stride_rows) and the gates as one run behind it: a handful of
    launches for 200 iterations instead of one per call.  Columns == the oracle's loop == the same loop with recording off; check()
    passes; a body of another shape in the middle (one more gate) and a different bound at the end only cut the loop in pieces; and a
    prove-twice pass (clear_witness, the same loop on other witnesses) finds the rows in place.  prealloc: every witness allocated
    before the loop (the gadget calls then run on Variables from elsewhere: the `_allocated` footprints)."""
    from oracle import pyoracle as po
    import ctypes
    n_iter = 200
    vals = [50_000 + 997 * i for i in range(n_iter)]
    vals2 = [60_000 + 991 * i for i in range(n_iter)]
    mn, mx, mx2 = 50_000, 250_000, 2**40
    nb = ctypes.c_uint64()

    def loop(comp, is_oracle, values):
        res = []
        pre = None
        if prealloc:
            pre = [comp.allocate(synth.mont(v)) if is_oracle else pg.AllocatedScalar.allocate(comp, S(v)) for v in values]
        for k, v in enumerate(values):
            bound = mx2 if k >= n_iter - 20 else mx  # (the last twenty iterations: another ladder length)
            if is_oracle:
                a = pre[k] if prealloc else comp.allocate(synth.mont(v))
                r = int(comp.L.range_check(comp.c, f(mn), f(bound), a)) if gadget == "range_check" else \
                    int(comp.L.max_bound(comp.c, f(bound), a, ctypes.byref(nb)))
                exp = int(mn <= v < bound) if gadget == "range_check" else int(v < bound)
                for _ in range(gates + (1 if k == 77 else 0)):
                    comp.L.composer_constrain_to_constant(comp.c, r, f(exp), None)
            else:
                a = pre[k] if prealloc else pg.AllocatedScalar.allocate(comp, S(v))
                r = pg.range_check(comp, S(mn), S(bound), a) if gadget == "range_check" else pg.max_bound(comp, S(bound), a)[0]
                exp = int(mn <= v < bound) if gadget == "range_check" else int(v < bound)
                for _ in range(gates + (1 if k == 77 else 0)):
                    comp.constrain_to_constant(r, S(exp), None)
            res.append(r)
        return res

    ora = po.Composer()
    ores = loop(ora, True, vals)
    exports = []
    for queued in (True, False):
        dev = pg.StandardComposer(engine, 1 << 17, 1 << 18)
        dev.queue(queued)
        _, f0, l0 = dev.queue_stats()
        res = loop(dev, False, vals)
        assert res == ores
        if queued:
            dev.flush()
            pending, f1, l1 = dev.queue_stats()
            assert pending == 0 and l1 - l0 <= 16, (pending, f1 - f0, l1 - l0)  # (one launch per call would be > 400)
        same(dev, ora)
        assert dev.check() == -1 and ora.check() == -1
        exports.append(dev.export())
        # the next rows: the loop's bodies are items of (gadget rows + gates rows) with closed-form wires and cycles (PermSeg::tail) where
        # the runs are long enough, rows of single calls elsewhere -- sigma == the oracle's bookkeeping, wire values == variables[w]
        n = dev.circuit_size()
        padded = 1 << (n - 1).bit_length()
        got, exp = dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded)
        if not np.array_equal(got, exp):
            w, g = np.argwhere(got != exp)[0]
            raise AssertionError(f"sigma differs first at wire {w}, gate {g}: {got[w, g]} != {exp[w, g]}")
        cols, m = dev.device_columns(), dev.materialize()
        for wname in ("w_l", "w_r", "w_o"):
            wv = getattr(cols, wname)[:n]
            assert torch.equal(m[wname + "_value"], cols.var_values[wv]), wname
        if queued:  # prove twice: the same loop on other witnesses finds every row in place
            dev.clear_witness()
            ora2 = po.Composer()
            assert loop(dev, False, vals2) == loop(ora2, True, vals2)
            same(dev, ora2)
            assert dev.check() == -1
            in_place, rewritten, _ = dev.refresh_stats()
            assert rewritten == 0 and in_place >= dev.circuit_size() - 8, (in_place, rewritten, dev.circuit_size())
    for k in COLS:
        assert np.array_equal(exports[0][k], exports[1][k]), k


def test_reference_loop_across_a_full_queue(engine):
    """3000 iterations of allocate, range_check, constrain_to_constant = 9000 recorded calls: the queue (8192 entries) flushes itself in
    the middle of a loop body, so one flush ends on a pair without its gate and the next begins with that gate.  Same columns, sigma and
    wire values as the oracle's loop."""
    from oracle import pyoracle as po
    n_iter = 3000
    vals = [50_000 + 67 * i for i in range(n_iter)]
    mn, mx = 50_000, 250_000
    ora = po.Composer()
    dev = pg.StandardComposer(engine, 1 << 19, 1 << 21)
    dev.queue(True)
    _, f0, l0 = dev.queue_stats()
    for v in vals:
        r = pg.range_check(dev, S(mn), S(mx), pg.AllocatedScalar.allocate(dev, S(v)))
        dev.constrain_to_constant(r, S(int(mn <= v < mx)), None)
        ro = int(ora.L.range_check(ora.c, f(mn), f(mx), ora.allocate(synth.mont(v))))
        ora.L.composer_constrain_to_constant(ora.c, ro, f(int(mn <= v < mx)), None)
        assert r == ro
    dev.flush()
    pending, f1, l1 = dev.queue_stats()
    assert pending == 0 and f1 - f0 >= 2 and l1 - l0 <= 24, (pending, f1 - f0, l1 - l0)
    same(dev, ora)
    assert dev.check() == -1
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    got, exp = dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded)
    if not np.array_equal(got, exp):
        w, g = np.argwhere(got != exp)[0]
        raise AssertionError(f"sigma differs first at wire {w}, gate {g}: {got[w, g]} != {exp[w, g]}")
    cols, m = dev.device_columns(), dev.materialize()
    for wname in ("w_l", "w_r", "w_o"):
        wv = getattr(cols, wname)[:n]
        assert torch.equal(m[wname + "_value"], cols.var_values[wv]), wname


def test_dependent_gate_calls(engine):
    """a run of gate calls whose outputs feed later calls of the same run (a chain of add / mul over earlier results,
    1500 calls: more than one launch of the queue kernel): values are computed in command order, level by level"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 14, 1 << 14), po.Composer()
    L = ora.L
    vs = [dev.add_input(S(3 + i)) for i in range(4)]
    assert vs == [ora.add_input(synth.mont(3 + i)) for i in range(4)]
    for i, r in enumerate(int(x) for x in synth.splitmix64(1500, 5)):
        a, b = vs[r % len(vs)], vs[(r >> 20) % len(vs)]
        if i % 3 == 2:
            a = vs[-1]  # a dependency on the call just before: the chain grows deep
        k1, k2, k3 = (r >> 8) % 97, (r >> 16) % 89, (r >> 24) % 83
        if r & 1:
            v = dev.add((S(k1), a), (S(k2), b), S(k3), None)
            assert v == L.composer_add(ora.c, f(k1), a, f(k2), b, f(k3), None)
        else:
            pi = f(k2) if i % 5 == 0 else None
            v = dev.mul(S(k1 + 1), a, b, S(k3), S(k2) if pi is not None else None)
            assert v == L.composer_mul(ora.c, f(k1 + 1), a, b, f(k3), C.byref(pi) if pi is not None else None)
        vs.append(v)
        if i % 50 == 7:
            dev.boolean_gate(dev.zero_var)
            L.composer_boolean_gate(ora.c, 0)
    assert dev.queue_stats()[0] > 1500
    same(dev, ora)
    assert dev.check() == ora.check()  # the public inputs and random constants leave rows unsatisfied: the SAME first one


def test_interleaving_and_self_flush(engine):
    """gadget calls on witnesses allocated long before (their own kind of run, two sets of bounds), a direct gadget call
    and a batched append in between, and more entries than the queue holds (it flushes itself): everything lands in
    call order"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 17, 1 << 19), po.Composer()
    L = ora.L
    ws = [pg.AllocatedScalar.allocate(dev, S(1000 + 7 * i)) for i in range(40)]
    ows = [ora.allocate(synth.mont(1000 + 7 * i)) for i in range(40)]
    for w, ow in zip(ws[:20], ows[:20]):   # witnesses allocated earlier: an "allocated" run
        assert pg.range_check(dev, S(0), S(2**16), w) == int(L.range_check(ora.c, f(0), f(2**16), ow))
    for w, ow in zip(ws[20:30], ows[20:30]):   # another gadget: a new run
        nb = C.c_uint64()
        r, n = pg.max_bound(dev, S(5000), w)
        assert r == int(L.max_bound(ora.c, f(5000), ow, C.byref(nb))) and n == nb.value == 14
    # a direct (not queued) gadget call: flushes what is pending first
    e = pg.maybe_equal(dev, ws[0], ws[1])
    assert e == int(L.maybe_equal(ora.c, ows[0], ows[1]))
    for w, ow in zip(ws[30:], ows[30:]):   # allocate-free calls with other bounds, then fused pairs again
        assert pg.range_check(dev, S(1000), S(1200), w) == int(L.range_check(ora.c, f(1000), f(1200), ow))
    for v in (5, 1100, 77777):
        a, oa = pg.AllocatedScalar.allocate(dev, S(v)), ora.allocate(synth.mont(v))
        assert pg.range_check(dev, S(1000), S(1200), a) == int(L.range_check(ora.c, f(1000), f(1200), oa))
        dev.constrain_to_constant(a.var, S(v), None)   # a gate call between the pairs: they no longer fuse into one run
        L.composer_constrain_to_constant(ora.c, oa.var, f(v), None)
    # a batched append in the middle
    wit = synth.uniform_below(16, 2**16, seed=4)
    dev.range_check_batch(S(0), S(2**16), torch.from_numpy(wit.view(np.int64)).to("cuda:0"))
    for w in wit:
        L.range_check(ora.c, f(0), f(2**16), ora.allocate(w))
    same(dev, ora)
    assert dev.check() == -1 and ora.check() == -1
    for i in range(9000):                  # > 8192 queue entries between two flushes
        dev.boolean_gate(dev.zero_var)
        L.composer_boolean_gate(ora.c, 0)
    assert 0 < dev.queue_stats()[0] < 8192
    same(dev, ora)
    assert dev.check() == -1


def test_errors_are_reported_at_the_call(engine):
    """recording does not defer errors: an unknown Variable, a non-reduced scalar or a full composer fail at the call
    that causes them, and nothing of that call is recorded"""
    dev = pg.StandardComposer(engine, 64, 64)
    a = dev.add_input(S(5))
    n0, v0, p0 = dev.circuit_size(), dev.num_variables(), dev.queue_stats()[0]
    with pytest.raises(pg.PgError, match="unknown Variable"):
        dev.boolean_gate(a + 100)
    with pytest.raises(pg.PgError, match="unknown Variable"):
        dev.add((S(1), a), (S(1), a + 1), S(0), None)
    with pytest.raises(pg.PgError, match="capacity"):
        pg.range_check(dev, S(0), S(2**64), pg.AllocatedScalar(a, S(5)))
    assert (dev.circuit_size(), dev.num_variables(), dev.queue_stats()[0]) == (n0, v0, p0)
    dev.boolean_gate(dev.zero_var)
    assert dev.check() == -1
