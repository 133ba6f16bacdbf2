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


# ---- the f-rows of a whole circuit from its wire columns (oracle_sigma_fast_*, oracle_materialize_fast) ----------------------

def _composer_full(ora):
    """everything oracle/composer.c holds that the f-rows are made of: four wire columns, the dense table, all eleven selectors"""
    import ctypes as C
    n = ora.n
    exp, full = ora.export(), ora.full_columns()
    wires = [exp["w_l"], exp["w_r"], exp["w_o"], full["w_4"]]
    sel = {}
    for name, col in (("q_4", 5), ("q_arith", 6), ("q_range", 7), ("q_logic", 8), ("q_fixed_group_add", 9), ("q_variable_group_add", 10)):
        p = ora.L.composer_selector(ora.c, col)
        sel[name] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, 4)).copy()
    return wires, exp["var_values"], sel


def _check_f_rows_fast(ora, chunkings=((1 << 20, 1), (97, 3), (1000, 8), (5, 2))):
    """oracle_sigma_fast == composer_sigma and oracle_materialize_fast == the composer's own columns / table, for several
    chunkings and thread counts, padded and unpadded"""
    wires, values, sel = _composer_full(ora)
    n = ora.n
    q4 = {r: [int(x) for x in sel["q_4"][r]] for r in range(n) if sel["q_4"][r].any()}
    for padded in (n, 1 << (n - 1).bit_length(), n + 3):
        exp = ora.sigma(padded)
        for chunk, threads in chunkings:
            sf = po.SigmaFast(*wires, padded_n=padded, n_vars=ora.num_vars, threads=threads)
            got = sf.whole(chunk)
            sf.close()
            if not np.array_equal(got, exp):
                w, g = np.argwhere(got != exp)[0]
                raise AssertionError(f"sigma (padded {padded}, chunk {chunk}, {threads} threads) differs first at wire {w}, gate {g}: "
                                     f"{got[w, g]} != {exp[w, g]}")
    for chunk, threads in chunkings:
        for r0 in range(0, n, chunk):
            r1 = min(n, r0 + chunk)
            m = po.materialize_fast(*wires, values, r0, r1, q4=q4, threads=threads)
            for k in po.FULL_SCALAR_COLS:
                assert np.array_equal(m[k], sel[k][r0:r1]), (k, r0)
            assert np.array_equal(m["w_4"], wires[3][r0:r1])
            for k, w in zip(po.FULL_VALUE_COLS, wires):
                assert np.array_equal(m[k], values[w[r0:r1].astype(np.int64)]), (k, r0)


def _random_gate_program(ora, rng, steps, hot=()):
    """composer calls on random EXISTING Variables (old ones, recent ones, a few hot ones again and again): every cycle shape
    sigma has to get right -- Variables on several wires of one row, Variables never on a wire, zero_var on the other wires"""
    L = ora.L
    F = lambda x: po.fr(synth.mont(x))
    for _ in range(steps):
        nv = ora.num_vars
        pick = lambda: (rng.choice(hot) if hot and rng.random() < 0.2 else
                        rng.randrange(max(0, nv - 6), nv) if rng.random() < 0.5 else rng.randrange(nv))
        op = rng.randrange(8)
        if op == 0:
            ora.add_input(synth.mont(rng.randrange(1000)))
        elif op == 1:
            L.composer_add(ora.c, F(rng.randrange(5)), pick(), F(rng.randrange(5)), pick(), F(rng.randrange(9)), None)
        elif op == 2:
            L.composer_mul(ora.c, F(rng.randrange(5)), pick(), pick(), F(rng.randrange(9)), None)
        elif op == 3:
            L.composer_boolean_gate(ora.c, pick())
        elif op == 4:
            L.composer_assert_equal(ora.c, pick(), pick())
        elif op == 5:
            L.composer_constrain_to_constant(ora.c, pick(), F(rng.randrange(9)), None)
        elif op == 6:
            L.composer_poly_gate(ora.c, pick(), pick(), pick(), F(1), F(2), F(3), F(4), F(5), None)
        else:
            L.composer_add_witness_to_circuit_description(ora.c, F(rng.randrange(3)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_f_rows_fast_on_random_programs(seed):
    import random
    rng = random.Random(seed)
    ora = po.Composer(dummy=bool(seed & 1))
    _random_gate_program(ora, rng, 400, hot=(0,) if seed == 1 else (0, 1, 2))
    _check_f_rows_fast(ora)


def test_f_rows_fast_on_gadget_circuits():
    """gadget loops as the reference's tests write them, witnesses allocated inside and before the loop, results used by later
    rows (a Variable's cycle crosses chunk and thread ranges), a fused-mix loop with failing items: the fast forms == composer.c"""
    import ctypes as C
    import random
    rng = random.Random(9)
    ora = po.Composer()
    L, F = ora.L, lambda x: po.fr(synth.mont(x))
    x = ora.add_input(synth.mont(5))
    res = [int(L.range_check(ora.c, F(50_000), F(250_000), ora.allocate(synth.mont(v)))) for v in (60_000, 7, 250_000, 123_456)]
    allocs = [ora.allocate(synth.mont(v)) for v in (3, 1 << 20, 77)]
    nb = C.c_uint64()
    for a in allocs:
        res.append(int(L.max_bound(ora.c, F(1 << 16), a, C.byref(nb))))
        res.append(int(L.range_check(ora.c, F(0), F(1 << 7), a)))
    L.is_non_zero(ora.c, x, F(5))
    assert L.is_non_zero(ora.c, ora.add_input(synth.mont(0)), F(0)) == 1     # Err after the partial emission
    for r in res[:3] + res[-2:]:
        L.conditionally_select_one(ora.c, x, r)
        L.composer_constrain_to_constant(ora.c, r, F(1), None)
    L.maybe_equal(ora.c, allocs[0], allocs[1])
    _random_gate_program(ora, rng, 120, hot=(0, x, res[0]))
    _check_f_rows_fast(ora, chunkings=((1 << 20, 1), (211, 4), (64, 16)))


def _same_circuit(host, ora):
    """tests/frows_oracle.HostCircuit == the composer.c composer the same calls made"""
    wires, values, sel = _composer_full(ora)
    assert (host.n, host.nv) == (ora.n, ora.num_vars)
    for k in range(4):
        assert np.array_equal(host.wires()[k], wires[k]), ("w_l", "w_r", "w_o", "w_4")[k]
    assert np.array_equal(host.table(), values)
    assert host.q4 == {r: [int(x) for x in sel["q_4"][r]] for r in range(ora.n) if sel["q_4"][r].any()}


@pytest.mark.parametrize("shape", ["range_check", "allocated", "ragged", "mix", "loop"])
def test_host_circuit_equals_faithful_composer(shape):
    """the five circuits of tests/test_gpu_frows_exhaustive.py at a size composer.c runs in seconds: the host assembly from the
    threaded gadget forms == the same loop of reference calls on oracle/composer.c + oracle/gadgets.c, and the f-rows computed
    from it == composer_sigma / the composer's own columns"""
    import ctypes as C
    import bench
    from tests.frows_oracle import HostCircuit
    ora = po.Composer()
    L, F = ora.L, lambda x: po.fr(synth.mont(x))
    host = HostCircuit(60_000, 60_000, threads=3)
    if shape == "range_check":
        wit = synth.uniform_below(21, 2**64 + 2**60, seed=3)
        res = host.range_check_batch(0, 2**64, wit, chunk=8)
        assert list(res) == [int(L.range_check(ora.c, F(0), F(2**64), ora.allocate(w))) for w in wit]
    elif shape == "allocated":
        wit = synth.uniform_below(19, 2**16 + 2**14, seed=4)
        first = host.add_input_batch(wit)
        allocs = [ora.allocate(w) for w in wit]
        assert first == int(allocs[0].var)
        wv = np.arange(first, first + 19, dtype=np.uint64)
        for mn, mx in ((0, 2**16), (1000, 70_000)):   # (twice over the same Variables: their cycles run through both calls)
            res = host.range_check_allocated_batch(mn, mx, wv, wit, chunk=7)
            assert list(res) == [int(L.range_check(ora.c, F(mn), F(mx), a)) for a in allocs]
    elif shape == "ragged":
        mr, wt = _c4_like(24, seed=5)
        res, nb = host.max_bound_ragged_batch(mr, wt, chunk=5)
        n_out = C.c_uint64()
        for i in range(24):
            assert int(res[i]) == int(L.max_bound(ora.c, po.fr(mr[i]), ora.allocate(wt[i]), C.byref(n_out)))
            assert int(nb[i]) == n_out.value
    elif shape == "mix":
        v, y, s, a, b = bench.mix_inputs(40, seed=6)
        v[[0, 7, 8, 39]] = 0
        res, err = host.scalar_mix_batch(v, y, s, a, b, chunk=9)
        for i in range(40):
            vs = [ora.add_input(x[i]) for x in (v, y, s, a, b)]
            st = L.is_non_zero(ora.c, vs[0], po.fr(v[i]))
            assert st == int(err[i])
            assert int(res[i, 0]) == int(L.conditionally_select_one(ora.c, vs[1], vs[2]))
            A, B = po.AllocatedScalar(vs[3], po.fr(a[i])), po.AllocatedScalar(vs[4], po.fr(b[i]))
            assert int(res[i, 1]) == int(L.maybe_equal(ora.c, A, B))
    else:
        wit = synth.uniform_below(17, 300_000, seed=7)
        res = host.range_check_loop_with_constrain(50_000, 250_000, wit)
        for i, w in enumerate(wit):
            r = int(L.range_check(ora.c, F(50_000), F(250_000), ora.allocate(w)))
            assert r == int(res[i])
            L.composer_constrain_to_constant(ora.c, r, F(1), None)
    _same_circuit(host, ora)
    n = host.n
    padded = 1 << (n - 1).bit_length()
    sf = host.sigma_plan(padded)
    assert np.array_equal(sf.whole(1 << 10), ora.sigma(padded))
    sf.close()
    _check_f_rows_fast(ora, chunkings=((500, 3),))


def test_small_batch_fast_equals_faithful():
    """oracle_small_batch_fast (the scalar gadgets and the gate calls over arrays of existing Variables, threaded) == the same
    calls one by one on oracle/composer.c + oracle/gadgets.c (src/scalar.rs:21-140): every kind, all nine arrays; inputs that are
    zero, equal, zero_var itself, results of earlier batches as inputs of later ones, is_non_zero items that stop at their error"""
    import random
    from tests.frows_oracle import HostCircuit
    rng = random.Random(12)
    Q = synth.Q
    ora = po.Composer()
    L, F = ora.L, lambda x: po.fr(synth.mont(x))
    host = HostCircuit(20_000, 20_000, threads=3)
    vals = [0, 0, 1, 1, 5, 5, Q - 1] + [rng.randrange(Q) for _ in range(40)]
    scal = synth.scalars_from_ints(vals)
    first = host.add_input_batch(scal)
    assert first == int(ora.allocate(scal[0]).var)
    for x in scal[1:]:
        ora.allocate(x)
    pick = lambda k: np.array([rng.randrange(ora.num_vars) for _ in range(k)], dtype=np.uint64)
    value = lambda v: L.composer_value(ora.c, int(v))
    full_checks = []

    def both(kind, faithful, a, b=None, c=None, selectors=None, chunk=7):
        """the batch on the host circuit (wires and assignments, chunked) and call by call on composer.c; the same batch once more with
        all nine arrays, compared with composer.c's rows at the end"""
        g0, v0, table = host.n, host.nv, host.table().copy()
        sel = None if selectors is None else synth.scalars_from_ints(list(selectors))
        plan = po.is_non_zero_plan(a, table) if kind == "is_non_zero" else None
        full = po.small_batch_fast(kind, a, b, c, table, 0, len(a), var_base=v0, zero_var=0, selectors=sel, plan=plan, threads=2)
        res = host.small_batch(kind, a, b, c, selectors=selectors, chunk=chunk)
        exp = [faithful(*[int(x[i]) for x in (a, b, c) if x is not None]) for i in range(len(a))]
        full_checks.append((g0, v0, full))
        return res, exp

    for rnd in range(2):   # (the second round's inputs include the first round's results)
        a, b = pick(37), pick(37)
        a[:3], b[:3] = [0, first, first + 2], [first + 1, 0, first + 3]
        res, exp = both("select_zero", lambda x, y: int(L.conditionally_select_zero(ora.c, x, y)), a, b)
        assert list(res) == exp
        res, exp = both("select_one", lambda x, y: int(L.conditionally_select_one(ora.c, x, y)), pick(29), pick(29), chunk=11)
        assert list(res) == exp
        a, b = pick(33), pick(33)
        a[:4], b[:4] = [first + 4, first, first + 2, 0], [first + 5, first + 1, first + 6, 0]   # equal pairs, a zero difference of zeros
        res, exp = both("maybe_equal", lambda x, y: int(L.maybe_equal(ora.c, po.AllocatedScalar(x, value(x)), po.AllocatedScalar(y, value(y)))), a, b)
        assert list(res) == exp
        a = pick(41)
        a[:3] = [first, 0, first + 1]                                  # values 0 (once: zero_var itself) -> Err after one row
        (res, err), exp = both("is_non_zero", lambda x: int(L.is_non_zero(ora.c, x, value(x))), a, chunk=9)
        assert list(err) == exp and err[:3].all() and not err.all()
        q = [rng.randrange(Q) for _ in range(5)]
        res, exp = both("add", lambda x, y: int(L.composer_add(ora.c, F(q[1]), x, F(q[2]), y, F(q[4]), None)), pick(25), pick(25),
                        selectors=(0, q[1], q[2], Q - 1, q[4]), chunk=8)
        assert list(res) == exp
        res, exp = both("mul", lambda x, y: int(L.composer_mul(ora.c, F(q[0]), x, y, F(q[4]), None)), pick(25), pick(25),
                        selectors=(q[0], 0, 0, Q - 1, q[4]), chunk=8)
        assert list(res) == exp
        both("rows", lambda x, y, z: L.composer_poly_gate(ora.c, x, y, z, *[F(v) for v in q], None), pick(19), pick(19), pick(19), selectors=q, chunk=6)
        a = pick(13)
        both("rows", lambda x, y, z: L.composer_boolean_gate(ora.c, x), a, a, a, selectors=(1, 0, 0, Q - 1, 0))   # boolean_gate over an array
    _same_circuit(host, ora)
    exp = ora.export()
    for g0, v0, full in full_checks:
        G, V = full["n_gates"], full["n_vars"]
        for k in COLS[:8]:
            assert np.array_equal(full[k], exp[k][g0:g0 + G]), (k, g0)
        assert np.array_equal(full["var_values"], exp["var_values"][v0:v0 + V]), g0
    n = host.n
    padded = 1 << (n - 1).bit_length()
    sf = host.sigma_plan(padded)
    assert np.array_equal(sf.whole(1 << 9), ora.sigma(padded))
    sf.close()
