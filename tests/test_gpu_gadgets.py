"""GPU parity: max_bound (uniform + ragged), the scalar gadgets and the fused mix vs the CPU oracle, limb for limb."""
import os

import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
SEL, WIRES = COLS[:5], COLS[5:8]
Q = synth.Q
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def assert_cols(got, exp):
    for k in COLS:
        a, b = got[k], exp[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)[0]
            raise AssertionError(f"{k} differs first at {bad.tolist()}: gpu={a[tuple(bad)]:#x} oracle={b[tuple(bad)]:#x}")


def u64(t):
    return t.cpu().numpy().view(np.uint64)


# ---- max_bound ------------------------------------------------------------------

@pytest.mark.parametrize("mx,count", [(200, 20), (2**128 - 1, 9), (2**253 + 5, 7), (2, 5), (1, 3), (0, 3)])
def test_max_bound_uniform(engine, mx, count):
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    inside = [int(v) % max(mx, 1) for v in synth.splitmix64(count, mx % 9973 + 1)]
    wit = np.concatenate([synth.scalars_from_ints(inside), synth.random_scalars(count, 5),
                          synth.scalars_from_ints([0, (mx - 1) % Q, mx % Q, Q - 1])])
    ora = po.max_bound_batch(np.repeat(synth.scalars_from_ints([mx]), len(wit), axis=0), wit)
    assert ora["satisfied"]
    cols, res, n = engine.max_bound_batch(pg.BlsScalar.from_int(mx), dev(wit), 3, 5)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), ora)
    assert np.array_equal(u64(res), ora["result_vars"]) and (ora["num_bits"] == n).all()


def test_max_bound_ragged_and_golden(engine):
    from oracle import pyoracle as po
    g = dict(np.load(os.path.join(GOLD, "max_bound_ref.npz")))
    cols, res, nb, lay = engine.max_bound_ragged_batch(dev(g["max_range"]), dev(g["witness"]), 3, 5)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), g)
    assert np.array_equal(u64(res), g["result_vars"]) and nb.cpu().numpy().tolist() == g["num_bits"].tolist()
    # random 253-bit bounds (BASELINE config C4 shape) + the edge bounds 0, 1, 2, q-1 and tile-boundary batch sizes
    for batch in (1, 15, 16, 17, 70):
        bounds = [int.from_bytes(synth.splitmix64(4, 100 + i).tobytes(), "little") % 2**253 for i in range(batch)]
        bounds[:4] = [0, 1, 2, Q - 1][:min(4, batch)]
        wits = []
        for i, b in enumerate(bounds):
            r = int.from_bytes(synth.splitmix64(4, 900 + i).tobytes(), "little")
            wits.append(r % b if (i % 2 == 0 and b > 0) else r % Q)
        mr, wt = synth.scalars_from_ints(bounds), synth.scalars_from_ints(wits)
        ora = po.max_bound_batch(mr, wt)
        assert ora["satisfied"]
        cols, res, nb, lay = engine.max_bound_ragged_batch(dev(mr), dev(wt), 3, 5)
        torch.cuda.synchronize()
        assert (lay.n_gates, lay.n_vars) == (ora["n_gates"], ora["n_vars"])
        assert_cols(cols.to_numpy(), ora)
        assert np.array_equal(u64(res), ora["result_vars"])
        assert nb.cpu().numpy().astype(np.uint64).tolist() == ora["num_bits"].tolist()


# ---- stand-alone scalar gadgets -------------------------------------------------------

def oracle_two_input(fn_name, a_vals, b_vals, allocated: bool):
    """oracle: add_input all a's, then all b's (existing Variables), then the gadget per item"""
    from oracle import pyoracle as po
    c = po.Composer()
    batch = len(a_vals)
    a_vars = [c.add_input(x) for x in a_vals]
    b_vars = [c.add_input(x) for x in b_vals]
    g0, v0 = c.n, c.num_vars
    res = []
    for i in range(batch):
        if allocated:
            a = po.AllocatedScalar(a_vars[i], po.fr(a_vals[i]))
            b = po.AllocatedScalar(b_vars[i], po.fr(b_vals[i]))
            res.append(int(getattr(c.L, fn_name)(c.c, a, b)))
        else:
            res.append(int(getattr(c.L, fn_name)(c.c, a_vars[i], b_vars[i])))
    assert c.check() == -1
    return c.export(g0, v0), np.array(res, dtype=np.uint64), g0, v0, a_vars, b_vars


def pair_inputs(batch, seed, equal_every=0, bits=False):
    a = synth.random_scalars(batch, seed)
    b = synth.random_scalars(batch, seed + 1)
    if bits:
        b = synth.scalars_from_ints([int(x) & 1 for x in synth.splitmix64(batch, seed + 2)])
    if equal_every:
        b[::equal_every] = a[::equal_every]
    return a, b


@pytest.mark.parametrize("batch", [1, 5, 256, 300])
def test_maybe_equal_batch(engine, batch):
    a, b = pair_inputs(batch, 11, equal_every=2)
    exp, res, g0, v0, av, bv = oracle_two_input("maybe_equal", a, b, True)
    cols, got = engine.maybe_equal_batch(dev(np.array(av, np.uint64)), dev(a), dev(np.array(bv, np.uint64)), dev(b), g0, v0)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), exp)
    assert np.array_equal(u64(got), res)


def test_maybe_equal_reference_cases_golden(engine):
    g = dict(np.load(os.path.join(GOLD, "maybe_equal_ref.npz")))
    n = len(g["a"])
    # golden numbering: allocate(a_i), allocate(b_i), maybe_equal, per item -> inputs interleaved with outputs;
    # compare per item against a single-item call placed at the same indices
    for i in range(n):
        vb = 5 + 5 * i
        cols, got = engine.maybe_equal_batch(dev(np.array([vb], np.uint64)), dev(g["a"][i:i + 1]),
                                             dev(np.array([vb + 1], np.uint64)), dev(g["b"][i:i + 1]), 3 + 3 * i, vb + 2)
        torch.cuda.synchronize()
        out = cols.to_numpy()
        for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
            assert np.array_equal(out[k], g[k][3 * i:3 * i + 3]), k
        assert np.array_equal(out["var_values"], g["var_values"][5 * i + 2:5 * i + 5])
        assert int(u64(got)[0]) == int(g["result_vars"][i])
        assert synth.to_int(out["var_values"][2]) == int(g["expected"][i])


@pytest.mark.parametrize("batch", [1, 7, 256, 513])
def test_select_one_and_zero_batch(engine, batch):
    y, s = pair_inputs(batch, 21, bits=True)
    exp, res, g0, v0, yv, sv = oracle_two_input("conditionally_select_one", y, s, False)
    cols, got = engine.conditionally_select_one_batch(dev(np.array(yv, np.uint64)), dev(y), dev(np.array(sv, np.uint64)),
                                                      dev(s), g0, v0)
    torch.cuda.synchronize()
    out = cols.to_numpy()
    assert_cols(out, exp)
    assert np.array_equal(u64(got), res)
    # semantics: selector 1 -> y, selector 0 -> 1 (tests/scalar_gadgets_tests.rs:142-177)
    for i in (0, batch - 1):
        sel = synth.to_int(s[i])
        assert out["var_values"][4 * i + 3].tolist() == (y[i].tolist() if sel else synth.mont(1))
    exp, res, g0, v0, xv, sv = oracle_two_input("conditionally_select_zero", y, s, False)
    cols, got = engine.conditionally_select_zero_batch(dev(np.array(xv, np.uint64)), dev(y), dev(np.array(sv, np.uint64)),
                                                       dev(s), g0, v0)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), exp)
    assert np.array_equal(u64(got), res)


@pytest.mark.parametrize("batch,zeros", [(1, []), (1, [0]), (9, [3]), (300, [0, 1, 255, 256, 299]), (64, list(range(64)))])
def test_is_non_zero_batch(engine, batch, zeros):
    from oracle import pyoracle as po
    vals = synth.random_scalars(batch, 31)
    vals[zeros] = 0
    c = po.Composer()
    vars_ = [c.add_input(v) for v in vals]
    g0, v0 = c.n, c.num_vars
    errs = [int(c.L.is_non_zero(c.c, vars_[i], po.fr(vals[i]))) for i in range(batch)]
    assert c.check() == -1
    exp = c.export(g0, v0)
    cols, err, nerr = engine.is_non_zero_batch(dev(np.array(vars_, np.uint64)), dev(vals), g0, v0, zero_var=0)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), exp)
    assert err.cpu().numpy().tolist() == errs and nerr == len(zeros)


# ---- the fused mix (BASELINE config C3) -------------------------------------------------

def mix_inputs(batch, seed, zeros=()):
    v = synth.random_scalars(batch, seed)
    v[list(zeros)] = 0
    y = synth.random_scalars(batch, seed + 1)
    s = synth.scalars_from_ints([int(x) & 1 for x in synth.splitmix64(batch, seed + 2)])
    a = synth.random_scalars(batch, seed + 3)
    b = synth.random_scalars(batch, seed + 4)
    b[::2] = a[::2]
    return v, y, s, a, b


# 5003 items: past the size where the rows launch runs beside the pre-pass on the engine's second stream, not a multiple of
# any tile width, and ragged (items that stop at their error, at tile edges of the rows / variable-image launches too)
@pytest.mark.parametrize("batch,zeros", [(1, ()), (4, (2,)), (128, ()), (129, (0, 127, 128)), (1000, (5, 500, 999)),
                                         (5003, (0, 63, 64, 255, 256, 257, 1023, 1024, 4095, 4096, 4992, 5002))])
def test_scalar_mix_batch(engine, batch, zeros):
    from oracle import pyoracle as po
    v, y, s, a, b = mix_inputs(batch, 41, zeros)
    ora = po.scalar_mix_batch(v, y, s, a, b)
    assert ora["satisfied"]
    cols, res, err, nerr, lay = engine.scalar_mix_batch(dev(v), dev(y), dev(s), dev(a), dev(b), 3, 5, zero_var=0)
    torch.cuda.synchronize()
    assert (lay.n_gates, lay.n_vars) == (ora["n_gates"], ora["n_vars"])
    assert_cols(cols.to_numpy(), ora)
    assert np.array_equal(u64(res), ora["result_vars"])
    assert err.cpu().numpy().tolist() == ora["err_mask"].tolist() and nerr == len(zeros)


def test_scalar_mix_golden(engine):
    g = dict(np.load(os.path.join(GOLD, "scalar_mix.npz")))
    cols, res, err, nerr, lay = engine.scalar_mix_batch(*(dev(g[k]) for k in "vysab"), 3, 5, zero_var=0)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), g)
    assert np.array_equal(u64(res), g["result_vars"]) and err.cpu().numpy().tolist() == g["err_mask"].tolist()


# ---- canaries: no emitter writes outside the ranges its layout announces ------------------------

PAD = 24  # rows / variables of guard on each side of every column


def guarded_columns(n_gates, n_vars):
    import plonk_gadgets_amd as pg
    big = pg.Columns.allocate(n_gates + 2 * PAD, n_vars + 2 * PAD, "cuda:0")
    for name in COLS:
        getattr(big, name).fill_(0x5A5A5A5A5A5A5A5A)
    view = pg.Columns(*[getattr(big, n)[PAD:PAD + n_gates] for n in COLS[:8]], big.var_values[PAD:PAD + n_vars])
    return big, view


def assert_guards_intact(big, n_gates, n_vars):
    for name in COLS:
        t = getattr(big, name)
        n = n_vars if name == "var_values" else n_gates
        assert bool((t[:PAD] == 0x5A5A5A5A5A5A5A5A).all()) and bool((t[PAD + n:] == 0x5A5A5A5A5A5A5A5A).all()), name
        assert bool((t[PAD:PAD + n] != 0x5A5A5A5A5A5A5A5A).any()) or n == 0, name


@pytest.mark.parametrize("batch", [1, 15, 33, 257])
def test_no_writes_outside_layout(engine, batch):
    import ctypes as C
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib, h, st = engine._lib, engine._h, engine._stream()
    wit = dev(synth.random_scalars(batch, 3))
    # range_check, n = 19
    mn, mx = pg.BlsScalar.from_int(50_000), pg.BlsScalar.from_int(250_000)
    lay = engine.range_check_layout(mn, mx, batch)
    big, view = guarded_columns(lay.n_gates, lay.n_vars)
    engine.range_check_batch(mn, mx, wit, 3, 5, out=view)
    torch.cuda.synchronize()
    assert_guards_intact(big, lay.n_gates, lay.n_vars)
    # max_bound uniform, n = 129
    mxb = pg.BlsScalar.from_int(2**128 - 1)
    lay = engine.max_bound_layout(mxb, batch)
    big, view = guarded_columns(lay.n_gates, lay.n_vars)
    engine.max_bound_batch(mxb, wit, 3, 5, out=view)
    torch.cuda.synchronize()
    assert_guards_intact(big, lay.n_gates, lay.n_vars)
    # max_bound ragged
    bounds = dev(synth.scalars_from_ints([(int(x) % 2**60) << (int(x) % 190) for x in synth.splitmix64(batch, 8)]))
    nb = torch.empty((batch,), dtype=torch.int32, device="cuda:0")
    roff = torch.empty((batch + 1,), dtype=torch.int64, device="cuda:0")
    voff = torch.empty((batch + 1,), dtype=torch.int64, device="cuda:0")
    lc = _lib.LayoutC()
    assert lib.pg_max_bound_ragged_plan(h, bounds.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(),
                                        C.byref(lc), st) == 0
    big, view = guarded_columns(int(lc.n_gates), int(lc.n_vars))
    cc = view.as_c()
    assert lib.pg_max_bound_ragged_batch(h, bounds.data_ptr(), wit.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(),
                                         voff.data_ptr(), 3, 5, C.byref(cc), None, st) == 0
    torch.cuda.synchronize()
    assert_guards_intact(big, int(lc.n_gates), int(lc.n_vars))
    # the fused mix with some failing items
    v, y, s, a, b = mix_inputs(batch, 5, zeros=range(0, batch, 7))
    ins = [dev(x) for x in (v, y, s, a, b)]
    nerr = C.c_uint64()
    assert lib.pg_scalar_mix_plan(h, ins[0].data_ptr(), batch, roff.data_ptr(), voff.data_ptr(), None, C.byref(lc),
                                  C.byref(nerr), st) in (0, 1)
    big, view = guarded_columns(int(lc.n_gates), int(lc.n_vars))
    cc = view.as_c()
    assert lib.pg_scalar_mix_batch(h, *[t.data_ptr() for t in ins], batch, roff.data_ptr(), voff.data_ptr(), 3, 5, 0,
                                   C.byref(cc), None, st) == 0
    torch.cuda.synchronize()
    assert_guards_intact(big, int(lc.n_gates), int(lc.n_vars))
    # stand-alone scalar gadgets
    vars_a = dev(np.arange(5, 5 + batch, dtype=np.uint64))
    vars_b = dev(np.arange(5 + batch, 5 + 2 * batch, dtype=np.uint64))
    for fn, per in (("pg_conditionally_select_zero_batch", 1), ("pg_conditionally_select_one_batch", 4),
                    ("pg_maybe_equal_batch", 3)):
        big, view = guarded_columns(per * batch, per * batch)
        cc = view.as_c()
        assert getattr(lib, fn)(h, vars_a.data_ptr(), ins[1].data_ptr(), vars_b.data_ptr(), ins[3].data_ptr(), batch, 3,
                                5 + 2 * batch, C.byref(cc), None, st) == 0
        torch.cuda.synchronize()
        assert_guards_intact(big, per * batch, per * batch)


def test_invalid_arguments_are_rejected(engine):
    """NULL / misaligned pointers and non-reduced scalars come back as PG_ERR_INVALID_ARGUMENT, never as a launch"""
    import ctypes as C
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib, h, st = engine._lib, engine._h, engine._stream()
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**64)
    wit = dev(synth.random_scalars(4, 1))
    lay = engine.range_check_layout(mn, mx, 4)
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0")
    cc = cols.as_c()
    assert lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(mx.c), None, 4, 0, 0, C.byref(cc), None, st) == 2
    assert lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr() + 8, 4, 0, 0, C.byref(cc), None, st) == 2
    bad = _lib.ColumnsC(*[getattr(cols, n).data_ptr() for n in COLS])
    bad.q_l = None
    assert lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), 4, 0, 0, C.byref(bad), None, st) == 2
    bad = _lib.ColumnsC(*[getattr(cols, n).data_ptr() for n in COLS])
    bad.var_values = cols.var_values.data_ptr() + 8
    assert lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), 4, 0, 0, C.byref(bad), None, st) == 2
    notred = _lib.Scalar.of([2**64 - 1] * 4)
    assert lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(notred), wit.data_ptr(), 4, 0, 0, C.byref(cc), None, st) == 2
    assert lib.pg_range_check_batch(None, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), 4, 0, 0, C.byref(cc), None, st) == 2
    assert b"NULL" in lib.pg_last_error()


def test_config_c4_medium_every_limb(engine):
    """BASELINE config 4 shape: 1 500 x max_bound with random 253-bit bounds (data-dependent ladder length),
    half of the witnesses below their bound, every limb vs the faithful oracle"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import pyoracle as po
    mr, wt = bench.c4_inputs(1500, seed=0xC4)
    ora = po.max_bound_batch(mr, wt)
    assert ora["satisfied"]
    cols, res, nb, lay = engine.max_bound_ragged_batch(dev(mr), dev(wt), 3, 5)
    torch.cuda.synchronize()
    assert (lay.n_gates, lay.n_vars) == (ora["n_gates"], ora["n_vars"])
    assert_cols(cols.to_numpy(), ora)
    assert np.array_equal(u64(res), ora["result_vars"])
    assert nb.cpu().numpy().astype(np.uint64).tolist() == ora["num_bits"].tolist()
    # what the gadget decides: (max - 1 - w) mod q fits the ladder's n bits.  (With 253-bit bounds that is NOT
    # w < max for every field element: a witness close to q wraps to a small difference -- the reference's
    # max_bound alone has the same blind spot, range_check closes it with min_bound.)
    outcomes = [synth.to_int(cols.to_numpy()["var_values"][int(r) - 5]) for r in ora["result_vars"][:64]]
    ns = ora["num_bits"][:64]
    expect = [int((synth.to_int(m) - 1 - synth.to_int(w)) % Q < 2**int(n)) for w, m, n in zip(wt[:64], mr[:64], ns)]
    assert outcomes == expect and 0 < sum(expect) < 64


def test_config_c3_medium_every_limb(engine):
    """BASELINE config 3 shape: 20 000 fused items (is_non_zero + conditionally_select_one + maybe_equal), b = a for
    about half of them, every limb vs the faithful oracle"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import pyoracle as po
    v, y, s, a, b = bench.mix_inputs(20000, seed=0xC3)
    ora = po.scalar_mix_batch(v, y, s, a, b)
    assert ora["satisfied"]
    cols, res, err, nerr, lay = engine.scalar_mix_batch(dev(v), dev(y), dev(s), dev(a), dev(b), 3, 5, zero_var=0)
    torch.cuda.synchronize()
    assert nerr == 0 and (lay.n_gates, lay.n_vars) == (200000, 300000)
    assert_cols(cols.to_numpy(), ora)
    assert np.array_equal(u64(res), ora["result_vars"])


@pytest.mark.parametrize("batch", [1, 31, 32, 33, 100])
def test_gadgets_on_allocated_witnesses(engine, batch):
    """range_check / max_bound exactly as the reference takes them: on AllocatedScalars that already exist"""
    import ctypes as C
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    mn, mx = 50_000, 250_000
    vals = synth.scalars_from_ints([mn + int(v) % (2 * (mx - mn)) for v in synth.splitmix64(batch, 77)])
    for gadget in ("range_check", "max_bound"):
        c = po.Composer()
        ws = [c.allocate(v) for v in vals]
        g0, v0 = c.n, c.num_vars
        res = []
        for w in ws:
            if gadget == "range_check":
                res.append(int(c.L.range_check(c.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), w)))
            else:
                nb = C.c_uint64()
                res.append(int(c.L.max_bound(c.c, po.fr(synth.mont(mx)), w, C.byref(nb))))
        assert c.check() == -1
        exp = c.export(g0, v0)
        wv = dev(np.array([w.var for w in ws], dtype=np.uint64))
        if gadget == "range_check":
            cols, got = engine.range_check_allocated_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), wv,
                                                           dev(vals), g0, v0)
        else:
            cols, got, _ = engine.max_bound_allocated_batch(pg.BlsScalar.from_int(mx), wv, dev(vals), g0, v0)
        torch.cuda.synchronize()
        assert_cols(cols.to_numpy(), exp)
        assert u64(got).tolist() == res


def test_check_rows_detects_corruption(engine):
    """the device-side satisfiability check used at full size: green on good output, points at a corrupted row"""
    import plonk_gadgets_amd as pg
    mn, mx = pg.BlsScalar.from_int(50_000), pg.BlsScalar.from_int(250_000)
    wit = dev(synth.random_scalars(200, 3))
    cols, _ = engine.range_check_batch(mn, mx, wit, 3, 5)
    assert engine.check_rows(cols) == -1
    # n = 19: 87 rows, 562 variables per item.  Variable 2*562+3 is bit 1 of item 2's max block (used by its boolean
    # and ladder rows); variable 1234 is bit 108 of the same block: allocated (range.rs:128-131) but on no row.
    cols.var_values[2 * 562 + 3, 0] ^= 1   # flip one bit of a used variable
    assert engine.check_rows(cols) == 2 * 87 + 2 + 2 * 1  # the boolean row of that bit
    cols.var_values[2 * 562 + 3, 0] ^= 1
    cols.var_values[1234, 0] ^= 1          # an unused bit variable: no row can notice
    assert engine.check_rows(cols) == -1
    cols.var_values[1234, 0] ^= 1
    cols.q_c[4321, 2] ^= 4                 # ... of one selector
    assert engine.check_rows(cols) == 4321
    cols.q_c[4321, 2] ^= 4
    cols.w_o[777] = 10**12                 # a wire pointing outside the batch's variables
    assert engine.check_rows(cols) == 777
    # ragged max_bound and the fused mix are self-contained as well
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    mr, wt = bench.c4_inputs(300, seed=9)
    c4, _, _, _ = engine.max_bound_ragged_batch(dev(mr), dev(wt), 3, 5)
    assert engine.check_rows(c4) == -1
    c3, _, _, _, _ = engine.scalar_mix_batch(*(dev(x) for x in bench.mix_inputs(5000, seed=3)), 3, 5, zero_var=0)
    assert engine.check_rows(c3) == -1


@pytest.mark.parametrize("num_bits,batch", [(8, 5), (64, 33), (255, 70), (256, 3), (0, 4)])
def test_scalar_decomposition_batch(engine, num_bits, batch):
    """scalar_decomposition_gadget (src/range.rs:119-158) batched over existing witnesses vs the oracle"""
    import ctypes as C
    from plonk_gadgets_amd import _lib
    from oracle import pyoracle as po
    small = synth.scalars_from_ints([int(x) % (1 << max(num_bits, 1)) for x in synth.splitmix64(batch // 2 + 1, 4)])
    vals = np.ascontiguousarray(np.concatenate([small, synth.random_scalars(batch, 5)])[:batch])
    c = po.Composer()
    ws = [c.allocate(v) for v in vals]
    g0, v0 = c.n, c.num_vars
    res = [int(c.L.scalar_decomposition_gadget(c.c, num_bits, w, None)) for w in ws]
    assert c.check() == -1
    exp = c.export(g0, v0)
    import plonk_gadgets_amd as pg
    lay = _lib.LayoutC()
    assert engine._lib.pg_scalar_decomposition_layout(num_bits, batch, C.byref(lay)) == 0
    cols = pg.Columns.allocate(int(lay.n_gates), int(lay.n_vars), "cuda:0", g0, v0)
    cc = cols.as_c()
    out = torch.empty((batch,), dtype=torch.int64, device="cuda:0")
    wv = dev(np.array([w.var for w in ws], dtype=np.uint64))
    assert engine._lib.pg_scalar_decomposition_batch(engine._h, num_bits, wv.data_ptr(), dev(vals).data_ptr(), batch, g0, v0,
                                                     C.byref(cc), out.data_ptr(), engine._stream()) == 0
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), exp)
    assert u64(out).tolist() == res


def test_fuzz_very_ragged_max_bound(engine):
    """per-item bounds of every bit length 1..255 mixed inside the same tiles (ladder lengths 2..255 side by side),
    batch sizes around the tile width: every limb vs the faithful oracle"""
    import random
    from oracle import pyoracle as po
    rng = random.Random(77)
    for batch in (1, 15, 16, 17, 47, 120):
        bounds, wits = [], []
        for i in range(batch):
            bits = rng.randrange(1, 256)
            b = rng.randrange(1 << (bits - 1), min(1 << bits, Q))
            bounds.append(b)
            c = rng.random()
            wits.append(rng.randrange(0, b) if c < 0.5 else rng.randrange(Q) if c < 0.85 else rng.choice([b - 1, b, 0, Q - 1]))
        mr, wt = synth.scalars_from_ints(bounds), synth.scalars_from_ints(wits)
        ora = po.max_bound_batch(mr, wt)
        assert ora["satisfied"]
        cols, res, nb, lay = engine.max_bound_ragged_batch(dev(mr), dev(wt), 3, 5)
        torch.cuda.synchronize()
        assert (lay.n_gates, lay.n_vars) == (ora["n_gates"], ora["n_vars"])
        assert_cols(cols.to_numpy(), ora)
        assert np.array_equal(u64(res), ora["result_vars"])
        assert nb.cpu().numpy().astype(np.uint64).tolist() == ora["num_bits"].tolist()
        assert engine.check_rows(cols) == -1


@pytest.mark.parametrize("batch,zeros", [
    (1, (0,)), (31, ()), (33, (32,)), (700, (3, 77, 699)), (6000, (0, 64, 255, 256, 4096, 5999)),
    # every step count of the launch that plans and inverts (a wave owns 32 x steps items): 65 536 items is the last batch with
    # one step per wave, 70 000 the first with two and a ragged last wave
    (65536, (0, 31, 32, 65535)), (70000, tuple(range(1000, 1040)) + (69999,)),
    # beyond 2 M items the step count is capped and the launch has more workgroups than the chip holds at once: the look-back
    # then crosses workgroups that start later
    (2 * 1024 * 1024 + 12345, (0, 1, 1048576, 2097152, 2 * 1024 * 1024 + 12344) + tuple(range(5, 2 * 1024 * 1024, 65537))),
])
def test_planned_mix_call_matches_plan_then_emit(engine, batch, zeros):
    """pg_scalar_mix_planned_batch (the call plans itself: the launch that inverts makes the prefix sums by a look-back over
    its waves) == the synchronous plan (the plan kernel) followed by the emit call: offsets, totals, error mask, result
    Variables, every column"""
    import plonk_gadgets_amd as pg
    v, y, s, a, b = mix_inputs(batch, 13, zeros)
    ins = [dev(x) for x in (v, y, s, a, b)]
    _, roff, voff = engine.ragged_buffers(batch)
    err = torch.zeros((batch,), dtype=torch.uint8, device="cuda:0")
    lay, nerr = engine.scalar_mix_plan(ins[0], roff, voff, err)
    ref = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0")
    ref_res = torch.zeros((batch, 2), dtype=torch.int64, device="cuda:0")
    engine.scalar_mix_emit(*ins, roff, voff, ref, ref_res, 3, 5, 0)
    torch.cuda.synchronize()
    _, roff2, voff2 = engine.ragged_buffers(batch)
    roff2.fill_(-1); voff2.fill_(-1)
    err2 = torch.full((batch,), 7, dtype=torch.uint8, device="cuda:0")
    big = pg.Columns.allocate(10 * batch, 15 * batch, "cuda:0")  # worst case
    res = torch.zeros((batch, 2), dtype=torch.int64, device="cuda:0")
    for _ in range(2):  # twice: the second call finds the first one's state on the engine's streams
        engine.scalar_mix_planned(*ins, roff2, voff2, big, res, err2, 3, 5, 0)
    torch.cuda.synchronize()
    lay2, nerr2 = engine.plan_result()
    assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, nerr) == (10 * batch - 2 * len(zeros), 15 * batch - 2 * len(zeros), len(zeros))
    assert torch.equal(roff2, roff) and torch.equal(voff2, voff) and torch.equal(err2, err) and torch.equal(res, ref_res)
    for k in COLS:
        n = lay.n_vars if k == "var_values" else lay.n_gates
        assert torch.equal(getattr(big, k)[:n], getattr(ref, k)), k


def test_async_plans_match_sync_plans(engine):
    """plan_async + emit + (later) plan_result == the synchronous plan: same offsets, same totals, same columns"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import plonk_gadgets_amd as pg
    # scalar mix with some failing items
    v, y, s, a, b = mix_inputs(700, 9, zeros=(3, 77, 699))
    ins = [dev(x) for x in (v, y, s, a, b)]
    _, roff, voff = engine.ragged_buffers(700)
    lay, nerr = engine.scalar_mix_plan(ins[0], roff, voff)
    ref_roff, ref_voff = roff.clone(), voff.clone()
    ref = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0")
    engine.scalar_mix_emit(*ins, roff, voff, ref, None, 3, 5, 0)
    roff.zero_(); voff.zero_()
    big = pg.Columns.allocate(10 * 700, 15 * 700, "cuda:0")  # worst case
    engine.scalar_mix_plan_async(ins[0], roff, voff)
    engine.scalar_mix_emit(*ins, roff, voff, big, None, 3, 5, 0)
    torch.cuda.synchronize()
    lay2, nerr2 = engine.plan_result()
    assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, nerr) == (7000 - 6, 10500 - 6, 3)
    assert torch.equal(roff, ref_roff) and torch.equal(voff, ref_voff)
    for k in COLS:
        n = lay.n_vars if k == "var_values" else lay.n_gates
        assert torch.equal(getattr(big, k)[:n], getattr(ref, k)), k
    # ragged max_bound
    mr, wt = bench.c4_inputs(300, seed=5)
    nb, roff, voff = engine.ragged_buffers(300)
    lay = engine.max_bound_ragged_plan(dev(mr), nb, roff, voff)
    ref_roff = roff.clone()
    roff.zero_()
    engine.max_bound_ragged_plan_async(dev(mr), nb, roff, voff)
    torch.cuda.synchronize()
    lay2, nerr2 = engine.plan_result()
    assert (lay2.n_gates, lay2.n_vars, nerr2) == (lay.n_gates, lay.n_vars, 0) and torch.equal(roff, ref_roff)


def test_back_to_back_calls_without_syncs(engine):
    """40 calls of different gadgets and sizes enqueued back to back on one stream with no synchronisation in between:
    the engine's side stream, its fork/join events and its grow-only scratch must keep every call's output intact"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(40):
        kind = i % 4
        batch = int(rng.integers(1, 400))
        if kind == 0:
            mn, mx = 50_000, 250_000
            wit = mixed_witnesses_local(mn, mx, batch, seed=i)
            cols, res = engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), dev(wit), 3, 5)
            jobs.append(("rc", (mn, mx, wit), cols, res))
        elif kind == 1:
            v, y, s, a, b = mix_inputs(batch, 100 + i, zeros=(0,) if i % 8 == 1 else ())
            cols, res, err, nerr, lay = engine.scalar_mix_batch(dev(v), dev(y), dev(s), dev(a), dev(b), 3, 5, zero_var=0)
            jobs.append(("mix", (v, y, s, a, b), cols, res))
        elif kind == 2:
            a, b = pair_inputs(batch, 200 + i, equal_every=3)
            av, bv = np.arange(5, 5 + batch, dtype=np.uint64), np.arange(5 + batch, 5 + 2 * batch, dtype=np.uint64)
            cols, res = engine.maybe_equal_batch(dev(av), dev(a), dev(bv), dev(b), 3, 5 + 2 * batch)
            jobs.append(("me", (a, b), cols, res))
        else:
            mx = 2**64
            wit = synth.uniform_below(batch, 2**64 + 2**62, seed=300 + i)
            cols, res, _ = engine.max_bound_batch(pg.BlsScalar.from_int(mx), dev(wit), 3, 5)
            jobs.append(("mb", (mx, wit), cols, res))
    torch.cuda.synchronize()
    for kind, inp, cols, res in jobs:
        got = cols.to_numpy()
        if kind == "rc":
            ora = po.range_check_fast(synth.mont(inp[0]), synth.mont(inp[1]), inp[2], threads=4, var_base=5)
        elif kind == "mix":
            ora = po.scalar_mix_batch(*inp)
        elif kind == "me":
            ora, r, g0, v0, _, _ = oracle_two_input("maybe_equal", inp[0], inp[1], True)
            ora = dict(ora, result_vars=r)
        else:
            ora = po.max_bound_batch(np.repeat(synth.scalars_from_ints([inp[0]]), len(inp[1]), axis=0), inp[1])
        assert_cols(got, ora)
        assert np.array_equal(u64(res).reshape(-1), np.asarray(ora["result_vars"]).reshape(-1)), kind


def mixed_witnesses_local(mn, mx, n, seed):
    inside = synth.scalars_from_ints([mn + int(v) % (mx - mn) for v in synth.splitmix64(n, seed)])
    outside = synth.random_scalars(n, seed + 1)
    pick = (synth.splitmix64(n, seed + 2) & np.uint64(1)).astype(bool)
    return np.ascontiguousarray(np.where(pick[:, None], inside, outside))


def test_device_inversion_edge_values(engine):
    """the device's division-step inversion on values chosen around its 30-bit limb boundaries and the ends of the
    field (and a zero in between): is_non_zero's inverse variable must equal the oracle's a^(q-2), bit for bit"""
    from oracle import pyoracle as po
    Q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    ints = [1, 2, 3, Q - 1, Q - 2, (Q + 1) // 2, (Q - 1) // 2, 2**30 - 1, 2**30, 2**30 + 1, 2**60 - 1, 2**60, 2**90, 2**120 - 1,
            2**150, 2**180 + 2**30, 2**210 - 1, 2**240, 2**254, 2**255 % Q, 0, 2**32, 2**64 - 1, 2**128, 2**192 + 5, 7**80 % Q]
    ints += [(3**k) % Q for k in range(1, 300, 7)] + [Q - (5**k) % Q for k in range(1, 300, 11)]
    vals = synth.scalars_from_ints(ints)
    c = po.Composer()
    vars_ = [c.add_input(v) for v in vals]
    g0, v0 = c.n, c.num_vars
    errs = [int(c.L.is_non_zero(c.c, vars_[i], po.fr(vals[i]))) for i in range(len(ints))]
    exp = c.export(g0, v0)
    cols, err, nerr = engine.is_non_zero_batch(dev(np.array(vars_, np.uint64)), dev(vals), g0, v0, zero_var=0)
    torch.cuda.synchronize()
    assert_cols(cols.to_numpy(), exp)
    assert err.cpu().numpy().tolist() == errs and nerr == 1


@pytest.mark.parametrize("planned", [False, True])
def test_calls_can_be_captured_in_a_hip_graph(engine, planned):
    """after one warm-up call (scratch allocated) the asynchronous plan + emit of the ragged mix -- five kernels on the
    caller's stream, the inversion pre-pass on the engine's side stream, forked and joined with events -- records into a
    HIP graph; replays write the same bytes as the eager calls, also for new inputs in the same buffers"""
    import plonk_gadgets_amd as pg
    batch = 3000
    v, y, s, a, b = mix_inputs(batch, 77, zeros=(5, 2999))  # (3 blocks of the plan kernel: its look-back is replayed too)
    ins = [dev(x) for x in (v, y, s, a, b)]
    _, roff, voff = engine.ragged_buffers(batch)
    res = torch.empty((batch, 2), dtype=torch.int64, device="cuda:0")
    cols = pg.Columns.allocate(10 * batch, 15 * batch, "cuda:0")

    def step():
        if planned:  # the same in one call (the plan on the engine's rows stream for big batches, here on the caller's)
            engine.scalar_mix_planned(*ins, roff, voff, cols, res, None, 3, 5, 0)
        else:
            engine.scalar_mix_plan_async(ins[0], roff, voff)
            engine.scalar_mix_emit(*ins, roff, voff, cols, res, 3, 5, 0)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = {k: t.copy() for k, t in cols.to_numpy().items()}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    for k in ("q_m", "w_l", "var_values"):
        getattr(cols, k).zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = cols.to_numpy()
    lay, nerr = engine.plan_result()
    assert nerr == 2
    ng, nv = lay.n_gates, lay.n_vars
    for k in eager:
        n = nv if k == "var_values" else ng
        assert np.array_equal(got[k][:n], eager[k][:n]), k
    # new inputs, same buffers: the replay follows them
    v2, y2, s2, a2, b2 = mix_inputs(batch, 78)
    for t, x in zip(ins, (v2, y2, s2, a2, b2)):
        t.copy_(dev(x))
    graph.replay()
    torch.cuda.synchronize()
    from oracle import pyoracle as po
    ora = po.scalar_mix_batch(v2, y2, s2, a2, b2)
    got = cols.to_numpy()
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
        assert np.array_equal(got[k][:ora["n_gates"]], ora[k]), k
    assert np.array_equal(got["var_values"][:ora["n_vars"]], ora["var_values"])


def test_two_engines_on_two_streams():
    """one engine per stream (the documented rule): two engines driven from two torch streams at the same time, twelve
    calls each, interleaved from the host with no synchronisation until the end -- every output equals the oracle's"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    engines = [pg.Engine(0), pg.Engine(0)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    jobs = []
    for i in range(12):
        for e, (eng, st) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(st):
                batch = 150 + 37 * i + e
                if (i + e) % 2 == 0:
                    mn, mx = 50_000, 250_000
                    wit = mixed_witnesses_local(mn, mx, batch, seed=1000 + 10 * i + e)
                    cols, res = eng.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), dev(wit), 3, 5)
                    jobs.append(("rc", (mn, mx, wit), cols, res))
                else:
                    v, y, s, a, b = mix_inputs(batch, 2000 + 10 * i + e, zeros=(1,))
                    cols, res, err, nerr, lay = eng.scalar_mix_batch(dev(v), dev(y), dev(s), dev(a), dev(b), 3, 5, zero_var=0)
                    jobs.append(("mix", (v, y, s, a, b), cols, res))
    torch.cuda.synchronize()
    for kind, inp, cols, res in jobs:
        got = cols.to_numpy()
        if kind == "rc":
            ora = po.range_check_fast(synth.mont(inp[0]), synth.mont(inp[1]), inp[2], threads=4, var_base=5)
        else:
            ora = po.scalar_mix_batch(*inp)
        assert_cols(got, ora)
        assert np.array_equal(u64(res).reshape(-1), np.asarray(ora["result_vars"]).reshape(-1)), kind


def test_bulk_encodings(engine):
    """BlsScalar::from_bytes / to_bytes over a batch, device to device: Montgomery limbs of the model, the round trip, and
    the three encodings that are not below q flagged and zeroed"""
    from oracle.model import Q, mont_limbs
    ints = [0, 1, 2, Q - 1, 2**64, 2**128 + 5, 2**254, (Q - 1) // 2] + [int(x) for x in synth.splitmix64(300, 9)]
    ints += [(7**k) % Q for k in range(1, 200, 3)]
    raw = np.array([[(x >> (64 * i)) & (2**64 - 1) for i in range(4)] for x in ints], dtype=np.uint64)
    out, bad, nbad = engine.scalars_from_canonical(dev(raw))
    assert nbad == 0 and not bool(bad.any())
    assert np.array_equal(u64(out), np.array([mont_limbs(x) for x in ints], dtype=np.uint64))
    assert np.array_equal(u64(engine.scalars_to_canonical(out)), raw)
    over = [Q, Q + 1, 2**256 - 1, Q - 1]
    raw2 = np.array([[(x >> (64 * i)) & (2**64 - 1) for i in range(4)] for x in over], dtype=np.uint64)
    out2, bad2, nbad2 = engine.scalars_from_canonical(dev(raw2))
    assert nbad2 == 3 and bad2.cpu().numpy().tolist() == [1, 1, 1, 0]
    assert np.array_equal(u64(out2)[:3], np.zeros((3, 4), dtype=np.uint64)) and u64(out2)[3].tolist() == mont_limbs(Q - 1)
    # flagged encodings have a status of their own (PG_ERR_BAD_ENCODING = 6); an argument error stays an argument error
    import ctypes as C
    r2, o2, n = dev(raw2), torch.empty((4, 4), dtype=torch.int64, device="cuda:0"), C.c_uint64()
    lib, st = engine._lib, engine._stream()
    assert lib.pg_scalars_from_canonical_batch(engine._h, r2.data_ptr(), 4, o2.data_ptr(), None, C.byref(n), st) == 6 and n.value == 3
    assert lib.pg_scalars_from_canonical_batch(engine._h, r2.data_ptr() + 8, 3, o2.data_ptr(), None, C.byref(n), st) == 2
    assert lib.pg_scalars_from_canonical_batch(engine._h, None, 3, o2.data_ptr(), None, C.byref(n), st) == 2


@pytest.mark.parametrize("batch,zeros", [(52, ()), (300, ()), (1000, (0, 51, 255, 256, 700)), (2600, (2599,))])
@pytest.mark.parametrize("shifted", [("w_l",), ("w_l", "w_r", "w_o"), ("w_r",)])
def test_fused_mix_into_misaligned_wire_columns(engine, batch, zeros, shifted):
    """the fused mix into wire columns that start on an ODD 8-byte boundary -- what a composer appending at an odd row hands the
    call (include/plonk_gadgets_hip.h asks for 8-byte alignment of the wire columns, no more).  More than 51 items: the periodic
    rows launch pairs rows so that its 16-byte stores stay aligned, and from its second pass on the pair that straddles two
    passes belongs to the item BEFORE the pass's first (round 5 found row 512 of every such call with item 0's wires; the
    engine-level tests only ever passed aligned columns).  Every column == the oracle's, the elements around the columns intact."""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    v, y, s, a, b = bench.mix_inputs(batch, seed=batch)
    v[list(zeros)] = 0
    ora = po.scalar_mix_batch(v, y, s, a, b)
    ins = [dev(x) for x in (v, y, s, a, b)]
    _, roff, voff = engine.ragged_buffers(batch)
    big = pg.Columns.allocate(10 * batch + 2, 15 * batch + 2, "cuda:0")
    for t in (big.w_l, big.w_r, big.w_o):
        t.fill_(-1)
    G, V = ora["n_gates"], ora["n_vars"]
    wires = {k: (getattr(big, k)[1:1 + 10 * batch] if k in shifted else getattr(big, k)[2:2 + 10 * batch]) for k in WIRES}
    view = pg.Columns(big.q_m[:10 * batch], big.q_l[:10 * batch], big.q_r[:10 * batch], big.q_o[:10 * batch], big.q_c[:10 * batch],
                      wires["w_l"], wires["w_r"], wires["w_o"], big.var_values[:15 * batch])
    assert all((wires[k].data_ptr() % 16 == 8) == (k in shifted) for k in WIRES)
    res = torch.zeros((batch, 2), dtype=torch.int64, device="cuda:0")
    engine.scalar_mix_planned(*ins, roff, voff, view, res, None, 3, 5, 0)
    torch.cuda.synchronize()
    lay, nerr = engine.plan_result()
    assert (lay.n_gates, lay.n_vars, nerr) == (G, V, len(zeros))
    got = view.to_numpy()
    for k in SEL + WIRES:
        if not np.array_equal(got[k][:G], ora[k]):
            raise AssertionError(f"{k} differs first at row {int(np.argwhere(got[k][:G] != ora[k])[0][0])}")
    assert np.array_equal(got["var_values"][:V], ora["var_values"])
    for k in WIRES:
        lo = 1 if k in shifted else 2
        t = getattr(big, k)
        assert bool((t[:lo] == -1).all()) and bool((t[lo + G:] == -1).all()), k


@pytest.mark.parametrize("streams", [0, 1, 5, 16])
@pytest.mark.parametrize("units", [1, 511, 512, 513, 100_003])
def test_fill_bytes_writes_its_buffer_and_nothing_else(engine, streams, units):
    """pg_fill_bytes (the bare store streams bench.py times beside every workload): every 16-byte unit of the buffer gets the pattern
    (low half, then its complement), the units behind it keep theirs -- for sizes around the one-shot form's 8 KiB block (512 units) and
    one that is no multiple of anything"""
    buf = torch.full((2 * (units + 8),), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda:0")
    pat = 0x0123456789ABCDEF
    engine.fill_bytes(buf[: 2 * units], streams, pat)
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint64)
    assert (got[2 * units:] == 0x5A5A5A5A5A5A5A5A).all()
    body = got[: 2 * units].reshape(units, 2)
    assert (body[:, 0] == pat).all() and (body[:, 1] == (~pat & 0xFFFFFFFFFFFFFFFF)).all()
