"""GPU: the reference's prove-twice flow on the device composer -- pg_composer_clear_witness.

/root/reference/tests/scalar_gadgets_tests.rs:108-119, 168-177, 226-235: a circuit is built and preprocessed, then
`prover.clear_witness()` and the same gadget calls on OTHER witnesses, then the second proof.  Here: after clear_witness the
composer counts from StandardComposer::new()'s state again; appends that repeat the previous build find their rows in
place and write only assignments.  Every test compares the whole composer (all nine columns, every limb) with the CPU oracle's
composer built FRESH from the second witnesses: rows kept from the first build must equal rows the reference would emit for
the second (structure is witness-independent), assignments must be the second build's.  A deliberately corrupted limb in a
kept row must survive the refresh (nothing rewrites rows in place) and vanish when the circuit changes (everything is
emitted again)."""
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
        if not np.array_equal(got[k], exp[k]):
            bad = np.argwhere(got[k] != exp[k])[0]
            raise AssertionError(f"{k} differs first at {bad.tolist()}")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def F(x):
    from oracle import pyoracle as po
    return po.fr(synth.mont(x))


def program(dev, ora, ws, bounds=(50_000, 250_000), extra_rows=0):
    """the reference's loop (allocate, range_check, constrain the outcome) over `ws`, then single composer calls with and
    without public inputs, a max_bound, a select (a call on existing Variables); on the device
    composer and on the oracle's.  Returns the outcomes' Variables."""
    from oracle import pyoracle as po
    mn, mx = bounds
    out = []
    for w in ws:
        a = pg.AllocatedScalar.allocate(dev, S(w))
        r = pg.range_check(dev, S(mn), S(mx), a)
        oa = ora.allocate(synth.mont(w))
        orr = int(ora.L.range_check(ora.c, F(mn), F(mx), oa))
        assert r == orr
        out.append(r)
    x, y = dev.add_input(S(ws[0] + 1)), dev.add_input(S(ws[-1] + 2))
    ox, oy = ora.add_input(synth.mont(ws[0] + 1)), ora.add_input(synth.mont(ws[-1] + 2))
    s1 = dev.add((S(3), x), (S(-5), y), S(7), None)
    assert s1 == ora.L.composer_add(ora.c, F(3), ox, F(-5), oy, F(7), None)
    pi = F(77)
    s2 = dev.mul(S(2), s1, x, S(1), S(77))
    assert s2 == ora.L.composer_mul(ora.c, F(2), s1, ox, F(1), C.byref(pi))
    one = dev.add_witness_to_circuit_description(S(1))
    assert one == ora.L.composer_add_witness_to_circuit_description(ora.c, F(1))
    dev.boolean_gate(one)
    ora.L.composer_boolean_gate(ora.c, one)
    mb, nb = pg.max_bound(dev, S(2**40), pg.AllocatedScalar(x, S(ws[0] + 1)))
    onb = C.c_uint64()
    assert mb == int(ora.L.max_bound(ora.c, F(2**40), po.AllocatedScalar(ox, F(ws[0] + 1)), C.byref(onb)))
    z = pg.conditionally_select_zero(dev, s2, out[0])
    assert z == int(ora.L.conditionally_select_zero(ora.c, s2, out[0]))
    for _ in range(extra_rows):
        dev.assert_equal(x, x)
        ora.L.composer_assert_equal(ora.c, ox, ox)
    return out


def test_prove_twice_flow(engine):
    """build, clear_witness, the same calls on other witnesses: == a fresh oracle composer of the second witnesses; the rows
    were found in place (stats), a corrupted limb in one of them survives, the public inputs are the second build's"""
    from oracle import pyoracle as po
    wa = [50_001, 250_000, 49_999, 123_456, 7]
    wb = [249_999, 50_000, 260_000, 3, 100_000]
    dev = pg.StandardComposer(engine, 1 << 12, 1 << 13)
    ora_a = po.Composer()
    program(dev, ora_a, wa)
    same(dev, ora_a)
    assert dev.check() == -1
    n_rows = dev.circuit_size()
    # a limb of a range_check row (a queued gadget call) and of a single gate row, bent
    cols = dev.device_columns()
    keep = (int(cols.q_l[40, 1]), int(cols.q_c[n_rows - 3, 0]))
    cols.q_l[40, 1] ^= 0x55
    torch.cuda.synchronize()
    dev.clear_witness()
    assert (dev.circuit_size(), dev.num_variables()) == (3, 5)
    ora_b = po.Composer()
    res = program(dev, ora_b, wb)
    kept, rewritten, refreshing = dev.refresh_stats()
    # every row after StandardComposer::new()'s three was found in place
    assert refreshing and rewritten == 0 and kept == n_rows - 3, (kept, rewritten, n_rows)
    got = dev.export()
    assert got["q_l"][40, 1] == np.uint64(keep[0] ^ 0x55), "a row found in place was written again"
    cols = dev.device_columns()
    cols.q_l[40, 1] ^= 0x55
    torch.cuda.synchronize()
    same(dev, ora_b)
    assert dev.check() == -1
    assert [dev.value(r).to_int() for r in res] == [1, 1, 0, 0, 1]
    pi = dev.construct_dense_pi_vec().cpu().numpy().view(np.uint64)
    assert int(np.count_nonzero(pi.any(axis=1))) == 1  # one public input, not two


def test_another_circuit_after_clear_witness_is_emitted_in_full(engine):
    """the second build changes a public bound half way: what matched before it stays in place, everything from there on is
    written again -- a bent limb in a LATER row disappears -- and the composer equals the oracle's for the new circuit; a
    third build then refreshes against the second"""
    from oracle import pyoracle as po
    wa = [50_001, 250_000, 49_999, 123_456]
    dev = pg.StandardComposer(engine, 1 << 12, 1 << 13)
    program(dev, po.Composer(), wa)
    n1 = dev.circuit_size()
    cols = dev.device_columns()
    cols.q_c[n1 - 2, 0] ^= 0x1234  # the max_bound's last row region: will be rewritten
    torch.cuda.synchronize()
    dev.clear_witness()
    ora = po.Composer()
    # two loop iterations as before, then ANOTHER bound: a longer ladder, more rows
    first = program_prefix(dev, ora, wa[:2], (50_000, 250_000))
    rest = program(dev, ora, [5, 6], bounds=(0, 2**64), extra_rows=2)
    kept, rewritten, refreshing = dev.refresh_stats()
    assert not refreshing and kept == 2 * 87 and rewritten > 0
    same(dev, ora)
    assert dev.check() == -1
    # third build == second: everything signed is in place again
    dev.clear_witness()
    ora3 = po.Composer()
    program_prefix(dev, ora3, [70_000, 1], (50_000, 250_000))
    program(dev, ora3, [2**63, 2**65], bounds=(0, 2**64), extra_rows=2)
    kept3, rewritten3, refreshing3 = dev.refresh_stats()
    assert refreshing3 and rewritten3 == 0 and kept3 == dev.circuit_size() - 3
    same(dev, ora3)
    assert dev.check() == -1 and first and rest


def program_prefix(dev, ora, ws, bounds):
    mn, mx = bounds
    out = []
    for w in ws:
        a = pg.AllocatedScalar.allocate(dev, S(w))
        out.append(pg.range_check(dev, S(mn), S(mx), a))
        assert out[-1] == int(ora.L.range_check(ora.c, F(mn), F(mx), ora.allocate(synth.mont(w))))
    return out


@pytest.mark.parametrize("queue", [True, False])
def test_shorter_and_longer_rebuilds(engine, queue):
    """a rebuild that stops early, then one that runs past the first build's end; with the command queue and with one launch
    per call"""
    from oracle import pyoracle as po
    dev = pg.StandardComposer(engine, 1 << 12, 1 << 13)
    dev.queue(queue)
    program(dev, po.Composer(), [60_000, 70_000, 80_000], extra_rows=3)
    dev.clear_witness()
    ora = po.Composer()
    program_prefix(dev, ora, [1, 200_000], (50_000, 250_000))  # stops early
    same(dev, ora)
    assert dev.refresh_stats() == (2 * 87, 0, True)
    dev.clear_witness()
    ora = po.Composer()
    program(dev, ora, [90_000, 2, 250_001], extra_rows=6)  # the whole first build again, and three rows more
    kept, rewritten, _ = dev.refresh_stats()
    assert rewritten == 3 and kept > 3 * 87
    same(dev, ora)
    assert dev.check() == -1


def test_batched_appends_refresh(engine):
    """pg_composer_range_check_batch / _max_bound_batch / _scalar_mix_batch on witness scalars: in place on the second build
    (a mix batch with a failing item is not: its shape depends on its witnesses), equal to the oracle's composer throughout"""
    from oracle import pyoracle as po
    import test_gpu_gadgets as tg
    mn, mx = 7, 2**40 + 3

    def build(dev, ora, seed, zeros):
        w = np.concatenate([synth.scalars_from_ints([mn + int(v) % (mx - mn) for v in synth.splitmix64(30, seed)]),
                            synth.random_scalars(31, seed + 1)])
        r = dev.range_check_batch(S(mn), S(mx), t(w))
        o = [int(ora.L.range_check(ora.c, F(mn), F(mx), ora.allocate(x))) for x in w]
        assert r.cpu().numpy().view(np.uint64).tolist() == o
        w2 = synth.scalars_from_ints([int(v) % (2 * 10**9) for v in synth.splitmix64(50, seed + 2)])
        r2, nb = dev.max_bound_batch(S(10**9), t(w2))
        onb = C.c_uint64()
        o2 = [int(ora.L.max_bound(ora.c, F(10**9), ora.allocate(x), C.byref(onb))) for x in w2]
        assert r2.cpu().numpy().view(np.uint64).tolist() == o2 and nb == onb.value
        v, y, s, a, b = tg.mix_inputs(300, seed + 3, zeros)
        _, err, nerr = dev.scalar_mix_batch(t(v), t(y), t(s), t(a), t(b))
        assert nerr == len(zeros)
        for i in range(300):  # the fused item on the oracle's composer: five add_input, is_non_zero, select_one, maybe_equal
            vv, yy, ss = ora.add_input(v[i]), ora.add_input(y[i]), ora.add_input(s[i])
            aa, bb = ora.allocate(a[i]), ora.allocate(b[i])
            ora.L.is_non_zero(ora.c, vv, po.fr(v[i]))
            ora.L.conditionally_select_one(ora.c, yy, ss)
            ora.L.maybe_equal(ora.c, aa, bb)
        x = dev.add_input(S(5))  # (a constant is public: the same in every build)
        assert x == ora.add_input(synth.mont(5))
        dev.constrain_to_constant(x, S(5), None)
        ora.L.composer_constrain_to_constant(ora.c, x, F(5), None)

    dev = pg.StandardComposer(engine, 1 << 16, 1 << 17)
    ora = po.Composer()
    build(dev, ora, 100, ())
    same(dev, ora)
    n1 = dev.circuit_size()
    dev.clear_witness()
    ora = po.Composer()
    build(dev, ora, 200, ())
    kept, rewritten, refreshing = dev.refresh_stats()
    assert refreshing and kept == n1 - 3 and rewritten == 0, (kept, rewritten, n1)
    same(dev, ora)
    assert dev.check() == -1
    # a failing item in the mix: that append (and nothing before it) is emitted in full, and what follows it too
    dev.clear_witness()
    ora = po.Composer()
    build(dev, ora, 300, (17,))
    kept, rewritten, refreshing = dev.refresh_stats()
    assert not refreshing and kept == 61 * (4 * 42 + 11) + 50 * (2 * 31 + 5) and rewritten == 10 * 300 - 2 + 1
    same(dev, ora)


def test_growing_during_a_refresh_keeps_the_rows_in_place(engine):
    from oracle import pyoracle as po
    dev = pg.StandardComposer(engine, 400, 2600)
    program(dev, po.Composer(), [60_000, 70_000, 80_000])
    n1 = dev.circuit_size()
    dev.clear_witness()
    dev.reserve(5000, 9000)  # new buffers while only the initial rows are live: the first build's rows must come along
    ora = po.Composer()
    program(dev, ora, [1, 2, 100_000])
    assert dev.refresh_stats() == (n1 - 3, 0, True)
    same(dev, ora)
    assert dev.check() == -1


def test_clear_witness_argument_checks(engine):
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    assert lib.pg_composer_clear_witness(None) == 2
    assert lib.pg_composer_refresh_stats(None, None, None, None) == 2
    dev = pg.StandardComposer(engine)
    dev.clear_witness()  # nothing built yet
    assert dev.refresh_stats() == (0, 0, False) and (dev.circuit_size(), dev.num_variables()) == (3, 5)
    assert dev.check() == -1


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 101, 102])
def test_fuzz_programs_rebuilt_on_other_witnesses(engine, seed):
    """tests/test_gpu_composer.py's random programs of 30 operations -- every single call and every batched append -- built,
    then clear_witness and the SAME program (same public structure: operations, sizes, bounds, selectors, Variables referred to)
    on other witnesses, three builds in all, the last one cut short and followed by ANOTHER program: after every build the
    composer == a fresh oracle composer of that build, its first unsatisfied row and its sigma too.  (Every append is signed --
    those on device arrays of Variables or bounds by a digest of the arrays -- and found in place; the fused mix and
    is_non_zero only when no item fails: a witness-dependent shape ends the refresh and everything after it is emitted in
    full.)"""
    from oracle import pyoracle as po
    import test_gpu_composer as tc
    dev = pg.StandardComposer(engine, 1 << 14, 1 << 14)
    dev.auto_grow()
    kept_total = 0
    for build, (wit_seed, steps, prog) in enumerate([(1000 + seed, 30, seed), (2000 + seed, 30, seed), (3000 + seed, 17, seed),
                                                     (4000 + seed, 30, seed + 50)]):
        if build:
            dev.clear_witness()
        ora = po.Composer()
        log = tc.run_fuzz_program(dev, ora, prog, wit_seed, steps)
        kept, rewritten, refreshing = dev.refresh_stats()
        kept_total += kept
        try:
            same(dev, ora)
            assert dev.check() == ora.check()
            n = dev.circuit_size()
            padded = 1 << (n - 1).bit_length()
            assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
        except AssertionError as e:
            raise AssertionError(f"seed {seed}, build {build} (kept {kept}, rewritten {rewritten}, refreshing {refreshing}), program {log}: {e}")
        if build == 0:
            assert kept == 0
            first_log = log
    # the refresh ends at the first append whose rows differ, and in these programs only a fused mix or an is_non_zero batch can
    # (random values hold zeros: a shape that depends on the witnesses): whatever appends rows before the first of them must have
    # been found in place
    ops = [op for op, _ in first_log]
    cut = min([ops.index(o) for o in ("mix", "inz") if o in ops] or [len(ops)])
    if any(op not in ("add_input", "alloc_batch") for op in ops[:cut]):
        assert kept_total > 0, f"seed {seed}: appends {ops[:cut]} before the first witness-dependent shape, yet no row was found in place"


def test_appends_on_device_arrays_are_signed_by_their_contents(engine):
    """batched appends whose rows depend on device arrays -- Variables (maybe_equal, select, gate batches, gadgets on allocated
    witnesses) and per-item bounds (ragged max_bound): rebuilt with the SAME arrays they are found in place (a bent limb in
    their rows survives); with ONE entry of one array changed, that append and everything after it is written again and the
    composer equals the oracle's for the new circuit"""
    from oracle import pyoracle as po
    tv = lambda xs: torch.tensor(xs, dtype=torch.int64, device="cuda:0")
    bounds = [200, 2**40 + 1, 3, 2**100, 77, 2**64]

    def build(dev, ora, seed, b_idx, bnds):
        w = synth.random_scalars(40, seed)
        first = dev.add_input_batch(t(w))
        assert first == int(ora.allocate(w[0]).var)
        for x in w[1:]:
            ora.allocate(x)
        a_idx = [first + i for i in range(0, 20)]
        r = dev.maybe_equal_batch(tv(a_idx), tv(b_idx))
        o = [ora.L.maybe_equal(ora.c, po.AllocatedScalar(a, ora.L.composer_value(ora.c, a)),
                               po.AllocatedScalar(b, ora.L.composer_value(ora.c, b))) for a, b in zip(a_idx, b_idx)]
        assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        r = dev.add_batch(S(3), tv(a_idx), S(Q - 1), tv(b_idx), S(9))
        o = [ora.L.composer_add(ora.c, F(3), a, F(Q - 1), b, F(9), None) for a, b in zip(a_idx, b_idx)]
        assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        vals = [ora.L.composer_value(ora.c, v) for v in a_idx]
        wv = np.array([[x.l[i] for i in range(4)] for x in vals], dtype=np.uint64)
        r = dev.range_check_allocated_batch(S(5), S(2**30), tv(a_idx), t(wv))
        o = [ora.L.range_check(ora.c, F(5), F(2**30), po.AllocatedScalar(v, x)) for v, x in zip(a_idx, vals)]
        assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        wr = synth.scalars_from_ints([int(v) % (2**70) for v in synth.splitmix64(len(bnds), seed + 5)])
        nb = C.c_uint64()
        r, _ = dev.max_bound_ragged_batch(t(synth.scalars_from_ints(bnds)), t(wr))
        o = [ora.L.max_bound(ora.c, F(b), ora.allocate(x), C.byref(nb)) for b, x in zip(bnds, wr)]
        assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        dev.boolean_gate_batch(tv(a_idx[:7]))
        for a in a_idx[:7]:
            ora.L.composer_boolean_gate(ora.c, a)
        return first

    dev = pg.StandardComposer(engine, 1 << 14, 1 << 15)
    ora = po.Composer()
    first = build(dev, ora, 1, [5 + 20 + i for i in range(20)], bounds)
    same(dev, ora)
    n1 = dev.circuit_size()
    cols = dev.device_columns()
    cols.q_m[3 + 1, 2] ^= 0x77  # a row of the maybe_equal batch (rows 3 .. 62)
    torch.cuda.synchronize()
    dev.clear_witness()
    ora = po.Composer()
    build(dev, ora, 2, [first + 20 + i for i in range(20)], bounds)
    assert dev.refresh_stats() == (n1 - 3, 0, True)
    got = dev.export()
    assert not np.array_equal(got["q_m"][4], ora.export()["q_m"][4]), "a row found in place was written again"
    cols = dev.device_columns()
    cols.q_m[3 + 1, 2] ^= 0x77  # still bent: nothing rewrote the row; straighten it
    torch.cuda.synchronize()
    same(dev, ora)
    assert dev.check() == ora.check()
    # one entry of the second Variable array differs: the maybe_equal batch and all that follows are emitted again
    dev.clear_witness()
    ora = po.Composer()
    other = [first + 20 + i for i in range(20)]
    other[13] = first + 3
    build(dev, ora, 3, other, bounds)
    kept, rewritten, refreshing = dev.refresh_stats()
    assert not refreshing and kept == 0 and rewritten == n1 - 3
    same(dev, ora)
    # ... and one bound of the ragged batch: everything before it stays
    dev.clear_witness()
    ora = po.Composer()
    b2 = list(bounds)
    b2[4] = 78
    build(dev, ora, 4, other, b2)
    kept, rewritten, refreshing = dev.refresh_stats()
    assert not refreshing and kept == 60 + 20 + 20 * (4 * 31 + 11) and rewritten > 0, (kept, rewritten)
    same(dev, ora)
    assert dev.check() == ora.check()


def test_a_failed_append_leaves_no_signature(engine):
    """an append that fails AFTER its in-place decision (a misaligned witness array is found by the engine call, behind the
    composer's own checks) must leave nothing in the log: the failed call, a DIFFERENT append at the same row, clear_witness,
    then the first call again -- successfully, at the same place -- must be emitted in full, not taken for a refresh of rows
    that are the other call's.  First in a fresh build, then with the failure in the middle of a refresh (which ends the
    refresh: what follows is written in full)."""
    from oracle import pyoracle as po
    lib = engine._lib
    mn, mx = S(50_000), S(250_000)
    wits_a = synth.scalars_from_ints([60_000, 40_000, 250_000, 77_777])
    wits_b = synth.scalars_from_ints([50_000, 249_999, 1, 100_000])
    d_a, d_b = t(wits_a), t(wits_b)
    spare = torch.zeros((4 * 4 + 1,), dtype=torch.int64, device="cuda:0")
    misaligned = spare.data_ptr() + 8  # 8 mod 16: pg_scalar arrays must be 16-byte aligned (plonk_gadgets_hip.h)

    def oracle_batch(ora, wits, lo, hi):
        for w in wits:
            a = ora.allocate(w)
            ora.L.range_check(ora.c, F(lo), F(hi), a)

    dev = pg.StandardComposer(engine, 1 << 13, 1 << 14)
    st = lib.pg_composer_range_check_batch(dev._h, C.byref(mn.c), C.byref(mx.c), C.c_void_p(misaligned), 4, None)
    assert st == 2 and dev.circuit_size() == 3, "PG_ERR_INVALID_ARGUMENT, nothing appended"
    dev.max_bound_batch(S(2**40), d_a)  # another append at the row where the failed one would have landed
    dev.clear_witness()
    dev.range_check_batch(mn, mx, d_b)  # the first call again, valid this time: must NOT be taken for a refresh
    kept, rewritten, refreshing = dev.refresh_stats()
    assert kept == 0 and not refreshing, (kept, rewritten, refreshing)
    ora = po.Composer()
    oracle_batch(ora, wits_b, 50_000, 250_000)
    same(dev, ora)
    assert dev.check() == -1

    # the failure in the middle of a refresh: [range_check_batch, max_bound_batch] built, cleared; the first append is found in
    # place, the second fails, a different one takes its place: the rest is emitted in full and equals the oracle's
    dev = pg.StandardComposer(engine, 1 << 13, 1 << 14)
    dev.range_check_batch(mn, mx, d_a)
    dev.max_bound_batch(S(2**40), d_a)
    rows_first = 4 * (4 * 19 + 11)
    dev.clear_witness()
    dev.range_check_batch(mn, mx, d_b)
    m40 = S(2**40)
    st = lib.pg_composer_max_bound_batch(dev._h, C.byref(m40.c), C.c_void_p(misaligned), 4, None, None)
    assert st == 2 and dev.circuit_size() == 3 + rows_first
    kept, rewritten, refreshing = dev.refresh_stats()
    assert kept == rows_first and not refreshing, "the failed append ended the refresh"
    dev.max_bound_batch(S(2**41), d_b)  # (would have been a mismatch anyway)
    dev.max_bound_batch(S(2**40), d_b)
    ora = po.Composer()
    oracle_batch(ora, wits_b, 50_000, 250_000)
    for bound in (2**41, 2**40):
        for w in wits_b:
            ora.L.max_bound(ora.c, F(bound), ora.allocate(w), None)
    same(dev, ora)
    assert dev.check() == -1
    # and a third build of that circuit is a refresh again from its first row to its last
    dev.clear_witness()
    dev.range_check_batch(mn, mx, d_a)
    dev.max_bound_batch(S(2**41), d_a)
    dev.max_bound_batch(S(2**40), d_a)
    kept, rewritten, refreshing = dev.refresh_stats()
    assert refreshing and rewritten == 0 and kept == dev.circuit_size() - 3
    ora = po.Composer()
    oracle_batch(ora, wits_a, 50_000, 250_000)
    for bound in (2**41, 2**40):
        for w in wits_a:
            ora.L.max_bound(ora.c, F(bound), ora.allocate(w), None)
    same(dev, ora)
