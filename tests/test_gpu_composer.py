"""GPU: the reference's own test circuits expressed on the device-resident composer through the reference-named
host API (plonk_gadgets_amd.composer), compared limb for limb with the same calls on the CPU oracle's composer."""
import ctypes as C

import numpy as np
import pytest
import torch

import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
from tests.refcases import MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES, fits_trim_degree

pytestmark = pytest.mark.gpu

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
Q = synth.Q
S = pg.BlsScalar.from_int


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


def same(dev: pg.StandardComposer, ora):
    got, exp = dev.export(), ora.export()
    assert dev.circuit_size() == ora.n and dev.num_variables() == ora.num_vars
    for k in COLS:
        assert got[k].shape == exp[k].shape, k
        assert np.array_equal(got[k], exp[k]), k


def test_initial_state(engine):
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine), po.Composer()
    assert (dev.circuit_size(), dev.num_variables(), dev.zero_var) == (3, 5, 0)
    same(dev, ora)
    assert dev.check() == -1
    bare = pg.StandardComposer(engine, with_dummy=False)
    assert (bare.circuit_size(), bare.num_variables()) == (1, 1)
    same(bare, po.Composer(dummy=False))
    # the fourth wire of the dummy rows lives in the materialised columns (q_4 = 1 on w_4 = Variable(2) in row 1)
    m = dev.materialize()
    assert m["w_4"].cpu().tolist() == [0, 2, 0]
    assert m["q_4"].cpu().numpy().view(np.uint64).tolist() == [[0] * 4, synth.mont(1), [0] * 4]
    assert (m["q_arith"].cpu().numpy().view(np.uint64) == np.array(synth.mont(1), dtype=np.uint64)).all()
    for k in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        assert int(m[k].abs().sum()) == 0


@pytest.mark.parametrize("min_range,max_range,witness,expected", RANGE_CHECK_CASES)
def test_range_check_reference_circuit(engine, min_range, max_range, witness, expected):
    """tests/range_gadgets_tests.rs:29-44: allocate, range_check, constrain_to_constant(res, outcome)"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine), po.Composer()
    w = pg.AllocatedScalar.allocate(dev, S(witness))
    res = pg.range_check(dev, S(min_range), S(max_range), w)
    dev.constrain_to_constant(res, S(int(expected)), None)
    ow = ora.allocate(synth.mont(witness))
    ores = int(ora.L.range_check(ora.c, po.fr(synth.mont(min_range)), po.fr(synth.mont(max_range)), ow))
    ora.L.composer_constrain_to_constant(ora.c, ores, po.fr(synth.mont(int(expected))), None)
    assert res == ores and w.var == ow.var
    same(dev, ora)
    assert dev.value(res).to_int() == int(expected)
    assert dev.check() == -1
    # the reference proves this very circuit under trim(1 << 10) (tests/range_gadgets_tests.rs:111-112)
    assert fits_trim_degree(dev.circuit_size(), "range_check")
    dev.constrain_to_constant(res, S(1 - int(expected)), None)  # the wrong outcome must not verify
    assert dev.check() == dev.circuit_size() - 1


@pytest.mark.parametrize("max_range,witness,expected", MAX_BOUND_CASES)
def test_max_bound_reference_circuit(engine, max_range, witness, expected):
    """tests/range_gadgets_tests.rs:13-27"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine), po.Composer()
    res, nbits = pg.max_bound(dev, S(max_range), pg.AllocatedScalar.allocate(dev, S(witness)))
    dev.constrain_to_constant(res, S(int(expected)), None)
    onb = C.c_uint64()
    ores = int(ora.L.max_bound(ora.c, po.fr(synth.mont(max_range)), ora.allocate(synth.mont(witness)), C.byref(onb)))
    ora.L.composer_constrain_to_constant(ora.c, ores, po.fr(synth.mont(int(expected))), None)
    assert (res, nbits) == (ores, onb.value)
    same(dev, ora)
    assert dev.check() == -1 and dev.value(res).to_int() == int(expected)
    assert fits_trim_degree(dev.circuit_size(), "max_bound")  # tests/range_gadgets_tests.rs:49-50


@pytest.mark.parametrize("a,b,expected", MAYBE_EQUAL_CASES)
def test_maybe_equal_reference_circuit(engine, a, b, expected):
    """tests/scalar_gadgets_tests.rs:19-31"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine), po.Composer()
    bit = pg.maybe_equal(dev, pg.AllocatedScalar.allocate(dev, S(a)), pg.AllocatedScalar.allocate(dev, S(b)))
    dev.constrain_to_constant(bit, S(int(expected)), None)
    obit = int(ora.L.maybe_equal(ora.c, ora.allocate(synth.mont(a)), ora.allocate(synth.mont(b))))
    ora.L.composer_constrain_to_constant(ora.c, obit, po.fr(synth.mont(int(expected))), None)
    assert bit == obit
    same(dev, ora)
    assert dev.check() == -1
    assert fits_trim_degree(dev.circuit_size(), "maybe_equal")  # tests/scalar_gadgets_tests.rs:16-17


def test_select_zero_reference_circuit(engine):
    """tests/scalar_gadgets_tests.rs:70-122: selector 0 -> result 0 verifies; selector 1 -> constraining to 0 fails"""
    from oracle import pyoracle as po
    for sel in (0, 1):
        value = 123456789 + sel
        dev, ora = pg.StandardComposer(engine), po.Composer()
        v, s = dev.add_input(S(value)), dev.add_input(S(sel))
        res = pg.conditionally_select_zero(dev, v, s)
        dev.constrain_to_constant(res, S(0), None)
        ov, os_ = ora.add_input(synth.mont(value)), ora.add_input(synth.mont(sel))
        ores = int(ora.L.conditionally_select_zero(ora.c, ov, os_))
        ora.L.composer_constrain_to_constant(ora.c, ores, po.fr(synth.mont(0)), None)
        assert res == ores
        same(dev, ora)
        assert (dev.check() == -1) == (sel == 0)
        assert fits_trim_degree(dev.circuit_size(), "select_zero")  # tests/scalar_gadgets_tests.rs:82-83


def test_select_one_reference_circuit_with_public_input(engine):
    """tests/scalar_gadgets_tests.rs:124-178: expected value injected as public input, dense PI vector"""
    from oracle import pyoracle as po
    for sel in (0, 1):
        value = Q - 12345
        expected = value if sel else 1
        dev, ora = pg.StandardComposer(engine), po.Composer()
        v, s = dev.add_input(S(value)), dev.add_input(S(sel))
        res = pg.conditionally_select_one(dev, v, s)
        dev.constrain_to_constant(res, S(0), S(-expected))
        ov, os_ = ora.add_input(synth.mont(value)), ora.add_input(synth.mont(sel))
        ores = int(ora.L.conditionally_select_one(ora.c, ov, os_))
        pi = po.fr(synth.mont(-expected))
        ora.L.composer_constrain_to_constant(ora.c, ores, po.fr(synth.mont(0)), C.byref(pi))
        assert res == ores
        same(dev, ora)
        assert dev.check() == -1 and dev.value(res).to_int() == expected
        assert fits_trim_degree(dev.circuit_size(), "select_one")  # tests/scalar_gadgets_tests.rs:139-140
        dense = dev.construct_dense_pi_vec().cpu().numpy().view(np.uint64)
        odense = np.zeros((ora.n, 4), dtype=np.uint64)
        ora.L.composer_dense_pi(ora.c, odense.ctypes.data)
        assert np.array_equal(dense, odense) and dense[-1].tolist() == synth.mont(-expected)


def test_is_non_zero_reference_circuit(engine):
    """tests/scalar_gadgets_tests.rs:180-236"""
    from oracle import pyoracle as po
    # (0, 0): Err after the partial emission
    dev, ora = pg.StandardComposer(engine), po.Composer()
    v = dev.add_input(S(0))
    with pytest.raises(pg.NonExistingInverse):
        pg.is_non_zero(dev, v, S(0))
    assert ora.L.is_non_zero(ora.c, ora.add_input(synth.mont(0)), po.fr(synth.mont(0))) == 1
    same(dev, ora)
    # mismatching var / value_assigned: Ok but unsatisfied (:224)
    dev, ora = pg.StandardComposer(engine), po.Composer()
    pg.is_non_zero(dev, dev.add_input(S(777)), S(778))
    assert ora.L.is_non_zero(ora.c, ora.add_input(synth.mont(777)), po.fr(synth.mont(778))) == 0
    same(dev, ora)
    assert dev.check() >= 0 and ora.check() == dev.check()
    # equal and non-zero: satisfied (:235)
    dev, ora = pg.StandardComposer(engine), po.Composer()
    r = 2**200 + 17
    pg.is_non_zero(dev, dev.add_input(S(r)), S(r))
    assert ora.L.is_non_zero(ora.c, ora.add_input(synth.mont(r)), po.fr(synth.mont(r))) == 0
    same(dev, ora)
    assert dev.check() == -1
    assert fits_trim_degree(dev.circuit_size(), "is_non_zero")  # tests/scalar_gadgets_tests.rs:193-194


def test_composer_gate_calls(engine):
    """every composer call of SURVEY 3.4, with and without public inputs"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine), po.Composer()
    L = ora.L
    f = lambda x: po.fr(synth.mont(x))
    a, b = dev.add_input(S(11)), dev.add_input(S(Q - 3))
    oa, ob = ora.add_input(synth.mont(11)), ora.add_input(synth.mont(Q - 3))
    one = dev.add_witness_to_circuit_description(S(1))
    assert one == L.composer_add_witness_to_circuit_description(ora.c, f(1))
    c1 = dev.add((S(5), a), (S(-7), b), S(9), None)
    assert c1 == L.composer_add(ora.c, f(5), oa, f(-7), ob, f(9), None)
    pi = f(1000)
    c2 = dev.add((S(2), a), (S(3), c1), S(0), S(1000))
    assert c2 == L.composer_add(ora.c, f(2), oa, f(3), c1, f(0), C.byref(pi))
    c3 = dev.mul(S(-1), c2, b, S(4), None)
    assert c3 == L.composer_mul(ora.c, f(-1), c2, ob, f(4), None)
    c4 = dev.mul(S(6), c3, c3, S(1), S(1000))
    assert c4 == L.composer_mul(ora.c, f(6), c3, c3, f(1), C.byref(pi))
    dev.boolean_gate(one)
    L.composer_boolean_gate(ora.c, one)
    dev.assert_equal(a, a)
    L.composer_assert_equal(ora.c, oa, oa)
    dev.mul_gate(a, one, a, S(1), S(-1), S(0), None)
    L.composer_mul_gate(ora.c, oa, one, oa, f(1), f(-1), f(0), None)
    dev.poly_gate(a, b, one, S(0), S(1), S(1), S(0), S(-8), None)  # 11 + (q-3) - 8 = 0
    L.composer_poly_gate(ora.c, oa, ob, one, f(0), f(1), f(1), f(0), f(-8), None)
    same(dev, ora)
    assert dev.check() == -1 and ora.check() == -1
    assert dev.value(c1).to_int() == (5 * 11 - 7 * (Q - 3) + 9) % Q
    # wire-value columns == variables gathered by wire index
    m = dev.materialize()
    exp = dev.export()
    for wname, vname in (("w_l", "w_l_value"), ("w_r", "w_r_value"), ("w_o", "w_o_value")):
        assert np.array_equal(m[vname].cpu().numpy().view(np.uint64), exp["var_values"][exp[wname].astype(np.int64)])
    w4 = m["w_4"].cpu().numpy()
    assert np.array_equal(m["w_4_value"].cpu().numpy().view(np.uint64), exp["var_values"][w4])


def test_batched_append_interleaved_with_single_calls(engine):
    """a circuit that mixes composer calls, a batched range_check append and single gadgets"""
    from oracle import pyoracle as po
    mn, mx = 50_000, 250_000
    wit = synth.scalars_from_ints([50_001, 250_000, 49_999, 123_456, 0])
    dev, ora = pg.StandardComposer(engine, 1 << 12, 1 << 13), po.Composer()
    x = dev.add_input(S(5))
    ox = ora.add_input(synth.mont(5))
    res = dev.range_check_batch(S(mn), S(mx), torch.from_numpy(wit.view(np.int64)).to("cuda:0"))
    ores = []
    for w in wit:
        ores.append(int(ora.L.range_check(ora.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), ora.allocate(w))))
    assert res.cpu().numpy().view(np.uint64).tolist() == ores
    y = pg.conditionally_select_zero(dev, x, ores[0])
    assert y == int(ora.L.conditionally_select_zero(ora.c, ox, ores[0]))
    r2 = pg.range_check(dev, S(0), S(2**64), pg.AllocatedScalar(x, S(5)))
    assert r2 == int(ora.L.range_check(ora.c, po.fr(synth.mont(0)), po.fr(synth.mont(2**64)),
                                       po.AllocatedScalar(ox, po.fr(synth.mont(5)))))
    same(dev, ora)
    assert dev.check() == -1
    assert [dev.value(r).to_int() for r in ores] == [1, 0, 0, 1, 0]


def test_errors_do_not_crash(engine):
    dev = pg.StandardComposer(engine, gate_capacity=8, var_capacity=8)
    with pytest.raises(pg.PgError, match="capacity"):
        pg.range_check(dev, S(0), S(2**64), pg.AllocatedScalar.allocate(dev, S(1)))
    with pytest.raises(pg.PgError, match="unknown Variable"):
        dev.boolean_gate(999)
    assert dev.circuit_size() == 3  # nothing was appended by the failed calls


def test_permutation_matches_oracle_bookkeeping(engine):
    """sigma from the device columns == the cycles of the oracle's per-gate Variable -> [WireData] map"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 12, 1 << 13), po.Composer()
    for padded in (3, 4, 8):
        assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    mn, mx = 50_000, 250_000
    wit = synth.scalars_from_ints([50_001, 250_000, 7])
    x = dev.add_input(S(5))
    ox = ora.add_input(synth.mont(5))
    res = dev.range_check_batch(S(mn), S(mx), torch.from_numpy(wit.view(np.int64)).to("cuda:0"))
    ores = [int(ora.L.range_check(ora.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), ora.allocate(w))) for w in wit]
    pg.is_non_zero(dev, x, S(5))            # an assert_equal row: zero_var on an output wire
    ora.L.is_non_zero(ora.c, ox, po.fr(synth.mont(5)))
    y = pg.conditionally_select_one(dev, x, ores[0])
    ora.L.conditionally_select_one(ora.c, ox, ores[0])
    dev.constrain_to_constant(y, S(5), None)
    ora.L.composer_constrain_to_constant(ora.c, y, po.fr(synth.mont(5)), None)
    same(dev, ora)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    sig = dev.permutation(padded).cpu().numpy().view(np.uint64)
    assert np.array_equal(sig, ora.sigma(padded))
    # sigma is a permutation of the 4 * padded positions, and it only ever links positions of the same Variable
    assert np.array_equal(np.sort(sig.reshape(-1)), np.arange(4 * padded, dtype=np.uint64))
    exp = dev.export()
    wires = np.stack([exp["w_l"], exp["w_r"], exp["w_o"], dev.materialize()["w_4"].cpu().numpy().view(np.uint64)])
    src_var = wires.reshape(-1)
    flat = sig[:, :n]
    dst_var = np.concatenate([np.pad(wires[w], (0, padded - n)) for w in range(4)])[flat.reshape(-1).astype(np.int64)]
    assert np.array_equal(dst_var, src_var)


def test_permutation_on_a_batched_circuit(engine):
    """sigma of a 300-item batched range_check circuit (n = 65: 81 303 rows, 325 k wire positions) vs the oracle"""
    from oracle import pyoracle as po
    mn, mx, batch = 0, 2**64, 300
    wit = synth.uniform_below(batch, 2**64 + 2**60, seed=5)
    dev = pg.StandardComposer(engine, 1 << 17, 1 << 18)
    ora = po.Composer()
    dev.range_check_batch(S(mn), S(mx), torch.from_numpy(wit.view(np.int64)).to("cuda:0"))
    for w in wit:
        ora.L.range_check(ora.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), ora.allocate(w))
    n = dev.circuit_size()
    assert n == ora.n == 3 + batch * 271
    padded = 1 << (n - 1).bit_length()
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    assert dev.check() == -1


def test_scalar_decomposition_reference_unit_test(engine):
    """src/range.rs:205-233 (scalar_decomposition_test): -100 does not fit 8 bits -> is_eq constrained to 0 holds; the
    verifier-side circuit, built from witness 1, has the same structure"""
    from oracle import pyoracle as po
    built = {}
    for name, wv in (("prover", Q - 100), ("verifier", 1)):
        dev, ora = pg.StandardComposer(engine), po.Composer()
        witness = pg.AllocatedScalar.allocate(dev, S(wv))
        is_eq, bits = pg.scalar_decomposition_gadget(dev, 8, witness)
        dev.constrain_to_constant(is_eq, S(0), None)
        obits = (C.c_uint64 * 8)()
        ois = int(ora.L.scalar_decomposition_gadget(ora.c, 8, ora.allocate(synth.mont(wv)), obits))
        ora.L.composer_constrain_to_constant(ora.c, ois, po.fr(synth.mont(0)), None)
        assert is_eq == ois and bits == list(obits)
        same(dev, ora)
        built[name] = dev.export()
        # prover: -100 needs more than 8 bits -> is_eq = 0 and the circuit is satisfied; verifier's witness 1 fits
        assert (dev.check() == -1) == (name == "prover")
        assert dev.value(is_eq).to_int() == (0 if name == "prover" else 1)
        assert fits_trim_degree(dev.circuit_size(), "scalar_decomposition")  # src/range.rs:208-209
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
        assert np.array_equal(built["prover"][k], built["verifier"][k])
    # num_bits edge cases: 0, 255, 256 bits, and > 256 (the reference panics on the slice, src/range.rs:134)
    for nbits in (0, 1, 255, 256):
        dev, ora = pg.StandardComposer(engine), po.Composer()
        w = pg.AllocatedScalar.allocate(dev, S(Q - 12345))
        is_eq, bits = pg.scalar_decomposition_gadget(dev, nbits, w)
        ois = int(ora.L.scalar_decomposition_gadget(ora.c, nbits, ora.allocate(synth.mont(Q - 12345)), None))
        assert is_eq == ois and len(bits) == nbits
        same(dev, ora)
        assert dev.check() == -1
    dev = pg.StandardComposer(engine)
    with pytest.raises(pg.PgError, match="num_bits"):
        pg.scalar_decomposition_gadget(dev, 257, pg.AllocatedScalar.allocate(dev, S(5)))


def _sigma_properties(dev, padded):
    """size-independent check of sigma, on the device: it is a permutation of the 4 * padded positions, it only links
    positions of one Variable, and walking a Variable's cycle the position 4 * gate + wire (= recording order) goes down
    exactly once (or the position maps to itself) -- i.e. every Variable's positions form ONE cycle in ascending order"""
    n = dev.circuit_size()
    sig = dev.permutation(padded).view(-1)
    P = 4 * padded
    assert torch.equal(torch.sort(sig).values, torch.arange(P, device=sig.device, dtype=sig.dtype))
    exp = dev.export()
    w4 = dev.materialize()["w_4"]
    wires = torch.full((4, padded), -1, dtype=torch.int64, device=sig.device)
    for k, name in enumerate(("w_l", "w_r", "w_o")):
        wires[k, :n] = torch.from_numpy(exp[name].view(np.int64)).to(sig.device)
    wires[3, :n] = w4.view(torch.int64)
    # padding rows: distinct pseudo-variables, each mapping to itself
    pad = torch.arange(P, device=sig.device, dtype=torch.int64).view(4, padded)[:, n:]
    wires[:, n:] = -2 - pad
    flat = wires.view(-1)
    assert torch.equal(flat[sig], flat)
    gate, wire = torch.arange(P, device=sig.device) % padded, torch.arange(P, device=sig.device) // padded
    order = 4 * gate + wire
    succ_order = 4 * (sig % padded) + sig // padded
    wraps = int((succ_order <= order).sum())
    assert wraps == int(torch.unique(flat).numel())


def test_permutation_two_segments_and_later_references(engine):
    """two batched calls with different ladders, single calls before, between and after them that reference result
    Variables of both (the cycle of such a Variable is spliced: local positions, then the sorted ones), a Variable
    referenced many times, and a constrain_to_constant row -- sigma vs the oracle's per-gate bookkeeping"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 16, 1 << 16), po.Composer()
    x = dev.add_input(S(9))
    ox = ora.add_input(synth.mont(9))
    assert x == ox
    segs = [(0, 2**16, synth.uniform_below(37, 2**16 + 2**14, seed=3)), (50_000, 250_000, synth.scalars_from_ints([7, 60_000, 250_001] * 5))]
    res_all = []
    for k, (mn, mx, wit) in enumerate(segs):
        res = dev.range_check_batch(S(mn), S(mx), torch.from_numpy(wit.view(np.int64)).to("cuda:0")).cpu().numpy().view(np.uint64)
        ores = [int(ora.L.range_check(ora.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), ora.allocate(w))) for w in wit]
        assert list(res) == ores
        res_all.append(ores)
        # between the segments: results of this batch, and x again and again
        for r in ores[:5] + ores[-2:]:
            y = pg.conditionally_select_one(dev, x, int(r))
            assert y == int(ora.L.conditionally_select_one(ora.c, ox, int(r)))
            dev.boolean_gate(int(r))
            ora.L.composer_boolean_gate(ora.c, int(r))
    # after both: one result of the FIRST batch again, twice, and its neighbour never
    r0 = res_all[0][3]
    for _ in range(2):
        pg.conditionally_select_zero(dev, r0, res_all[1][0])
        ora.L.conditionally_select_zero(ora.c, r0, res_all[1][0])
    dev.constrain_to_constant(res_all[1][1], S(0), None)
    ora.L.composer_constrain_to_constant(ora.c, res_all[1][1], po.fr(synth.mont(0)), None)
    same(dev, ora)
    n = dev.circuit_size()
    for padded in (n, 1 << (n - 1).bit_length()):
        assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    _sigma_properties(dev, 1 << (n - 1).bit_length())
    # asking twice reuses the composer's scratch and gives the same answer
    assert torch.equal(dev.permutation(n), dev.permutation(n))


def test_permutation_properties_at_scale(engine):
    """2^13 x range_check(0, 2^254) = 8.4 M rows, 33.8 M wire positions: the size-independent properties of sigma"""
    batch = 1 << 13
    dev = pg.StandardComposer(engine, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = torch.from_numpy(synth.uniform_below(batch, 2**254 + 2**250, seed=4).view(np.int64)).to("cuda:0")
    res = dev.range_check_batch(S(0), S(2**254), wit)
    pg.conditionally_select_one(dev, int(res[5]), int(res[batch - 1]))
    n = dev.circuit_size()
    _sigma_properties(dev, 1 << (n - 1).bit_length())


def _allocated_batch_case(engine, batch=40, seed=8):
    """allocate a batch, then range_check on the allocated witnesses (the reference's order when the caller allocates),
    twice over the same Variables with different ranges, on the device composer and on the oracle"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 19, 1 << 19), po.Composer()
    wit = synth.uniform_below(batch, 2**16 + 2**14, seed=seed)
    d_wit = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    first = dev.add_input_batch(d_wit)
    allocs = [ora.allocate(w) for w in wit]
    assert first == int(allocs[0].var) and dev.num_variables() == ora.num_vars
    wv = torch.arange(first, first + batch, dtype=torch.int64, device="cuda:0")
    for mn, mx in ((0, 2**16), (1000, 70_000)):
        res = dev.range_check_allocated_batch(S(mn), S(mx), wv, d_wit).cpu().numpy().view(np.uint64)
        ores = [int(ora.L.range_check(ora.c, po.fr(synth.mont(mn)), po.fr(synth.mont(mx)), a)) for a in allocs]
        assert list(res) == ores
    pg.conditionally_select_one(dev, first + 1, int(res[0]))
    ora.L.conditionally_select_one(ora.c, first + 1, int(res[0]))
    return dev, ora


def test_allocated_batch_on_the_composer(engine):
    """pg_composer_add_input_batch + pg_composer_range_check_allocated_batch: columns, satisfiability and sigma -- every
    item references a Variable created before the batch, i.e. positions inside the items go through the sorted list"""
    dev, ora = _allocated_batch_case(engine)
    same(dev, ora)
    assert dev.check() == -1 and ora.check() == -1
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    _sigma_properties(dev, padded)


def test_permutation_sparse_list_regrows(engine):
    """pg_composer_permutation_reserve(16): the sorted list starts far too short, the witness references inside the items
    overflow it, and the pass runs a second time with the size the first one reported -- same sigma"""
    dev, ora = _allocated_batch_case(engine, batch=300, seed=9)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    dev.permutation_reserve(16)
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    dev.permutation_reserve(16)
    assert np.array_equal(dev.permutation(n).cpu().numpy().view(np.uint64), ora.sigma(n))
    dev.permutation_reserve(0)
    assert np.array_equal(dev.permutation(n).cpu().numpy().view(np.uint64), ora.sigma(n))


def test_every_uniform_gadget_as_a_batched_append(engine):
    """one circuit built twice -- on the device composer through the batched appends (max_bound, max_bound on allocated
    witnesses, scalar_decomposition, conditionally_select_zero/one, maybe_equal) with single calls in between, and on
    the oracle through the reference's loops: same columns, satisfied, same sigma (the small items are linked
    several per workgroup; every item of the scalar gadgets references Variables created before its batch)"""
    from oracle import pyoracle as po
    import ctypes as C
    batch = 77
    dev, ora = pg.StandardComposer(engine, 1 << 17, 1 << 17), po.Composer()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    wit = synth.uniform_below(batch, 2**16 + 2**14, seed=21)
    wit2 = wit.copy()
    wit2[1::3] = synth.uniform_below(batch, 2**16, seed=22)[1::3]      # a third differ from wit
    bits = synth.scalars_from_ints([int(x) & 1 for x in synth.splitmix64(batch, 23)])
    firsts = [dev.add_input_batch(t(a)) for a in (wit, wit2, bits)]
    allocs = [[ora.allocate(w) for w in a] for a in (wit, wit2, bits)]
    assert firsts == [int(a[0].var) for a in allocs]
    wv, wv2, sv = (torch.arange(f, f + batch, dtype=torch.int64, device="cuda:0") for f in firsts)

    def both(dev_res, ora_res):
        assert list(dev_res.cpu().numpy().view(np.uint64)) == [int(r) for r in ora_res]
        return [int(r) for r in ora_res]

    nb = C.c_uint64()
    r, n1 = dev.max_bound_batch(S(2**16), t(wit))
    mb = both(r, [ora.L.max_bound(ora.c, po.fr(synth.mont(2**16)), ora.allocate(w), C.byref(nb)) for w in wit])
    assert n1 == nb.value
    y = pg.conditionally_select_one(dev, mb[0], mb[1])
    assert y == int(ora.L.conditionally_select_one(ora.c, mb[0], mb[1]))
    r, n2 = dev.max_bound_allocated_batch(S(1000), wv, t(wit))
    mba = both(r, [ora.L.max_bound(ora.c, po.fr(synth.mont(1000)), a, C.byref(nb)) for a in allocs[0]])
    assert n2 == nb.value
    dec = both(dev.scalar_decomposition_batch(10, wv2, t(wit2)),
               [ora.L.scalar_decomposition_gadget(ora.c, 10, a, None) for a in allocs[1]])
    sz = both(dev.conditionally_select_zero_batch(wv, sv),
              [ora.L.conditionally_select_zero(ora.c, int(a.var), int(s.var)) for a, s in zip(allocs[0], allocs[2])])
    so = both(dev.conditionally_select_one_batch(wv2, sv),
              [ora.L.conditionally_select_one(ora.c, int(a.var), int(s.var)) for a, s in zip(allocs[1], allocs[2])])
    dev.boolean_gate(firsts[2] + 3)
    ora.L.composer_boolean_gate(ora.c, firsts[2] + 3)
    me = both(dev.maybe_equal_batch(wv, wv2), [ora.L.maybe_equal(ora.c, a, b) for a, b in zip(allocs[0], allocs[1])])
    assert [dev.value(v).to_int() for v in me[:6]] == [int((wit[i] == wit2[i]).all()) for i in range(6)]
    pg.conditionally_select_zero(dev, me[-1], dec[-1])
    ora.L.conditionally_select_zero(ora.c, me[-1], dec[-1])
    same(dev, ora)
    assert dev.check() == -1 and ora.check() == -1
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    _sigma_properties(dev, padded)


def test_batched_appends_fail_cleanly(engine):
    """capacity errors and NULL arrays on the batched appends leave the composer as it was"""
    dev = pg.StandardComposer(engine, gate_capacity=64, var_capacity=64)
    wit = torch.from_numpy(synth.uniform_below(8, 2**16, seed=1).view(np.int64)).to("cuda:0")
    before = (dev.circuit_size(), dev.num_variables())
    with pytest.raises(pg.PgError, match="capacity"):
        dev.max_bound_batch(S(2**16), wit)
    with pytest.raises(pg.PgError, match="capacity"):
        dev.add_input_batch(torch.zeros((100, 4), dtype=torch.int64, device="cuda:0"))
    first = dev.add_input_batch(wit)
    v = torch.arange(first, first + 8, dtype=torch.int64, device="cuda:0")
    with pytest.raises(pg.PgError, match="capacity"):
        dev.conditionally_select_one_batch(v.repeat(4), v.repeat(4))      # 128 rows
    lib = dev._lib
    assert lib.pg_composer_maybe_equal_batch(dev._h, None, v.data_ptr(), 8, None) == 2      # PG_ERR_INVALID_ARGUMENT
    assert lib.pg_composer_scalar_decomposition_batch(dev._h, 300, v.data_ptr(), wit.data_ptr(), 8, None) == 2  # > 256 bits
    assert (dev.circuit_size(), dev.num_variables()) == (before[0], before[1] + 8)
    res = dev.maybe_equal_batch(v, v)
    assert [dev.value(int(r)).to_int() for r in res] == [1] * 8 and dev.check() == -1
    n = dev.circuit_size()
    _sigma_properties(dev, 1 << (n - 1).bit_length())


def test_batched_appends_reject_unknown_variables(engine):
    """every batched append that takes device arrays of Variables checks them against the composer's variable count
    first (the reference panics on an unknown Variable): an index past the end -- or an int64 -1 -- is
    PG_ERR_INVALID_ARGUMENT, nothing is appended, no out-of-bounds read happens, and the composer carries on"""
    dev = pg.StandardComposer(engine, gate_capacity=1 << 14, var_capacity=1 << 14)
    wit = torch.from_numpy(synth.uniform_below(8, 2**16, seed=2).view(np.int64)).to("cuda:0")
    first = dev.add_input_batch(wit)
    good = torch.arange(first, first + 8, dtype=torch.int64, device="cuda:0")
    nv = dev.num_variables()
    for poison in (nv, nv + 12345, -1, 1 << 40):
        bad = good.clone()
        bad[5] = poison
        before = (dev.circuit_size(), dev.num_variables())
        calls = [
            lambda: dev.range_check_allocated_batch(S(0), S(2**16), bad, wit),
            lambda: dev.max_bound_allocated_batch(S(2**16), bad, wit),
            lambda: dev.scalar_decomposition_batch(16, bad, wit),
            lambda: dev.conditionally_select_zero_batch(bad, good),
            lambda: dev.conditionally_select_one_batch(good, bad),
            lambda: dev.maybe_equal_batch(bad, bad),
            lambda: dev.is_non_zero_batch(bad),
            lambda: dev.poly_gate_batch(good, bad, good, S(0), S(1), S(0), S(0), S(0)),
            lambda: dev.add_batch(S(1), bad, S(1), good, S(0)),
            lambda: dev.mul_batch(S(1), good, bad, S(0)),
            lambda: dev.constrain_to_constant_batch(bad, S(7)),
            lambda: dev.boolean_gate_batch(bad),
        ]
        for k, call in enumerate(calls):
            with pytest.raises(pg.PgError, match="unknown Variable"):
                call()
            assert (dev.circuit_size(), dev.num_variables()) == before, (poison, k)
    # the composer is intact: the same calls with valid Variables append and check
    res = dev.maybe_equal_batch(good, good)
    dev.boolean_gate_batch(res)
    assert [dev.value(int(r)).to_int() for r in res] == [1] * 8 and dev.check() == -1


def test_ragged_batched_appends(engine):
    """max_bound with one public bound per item, is_non_zero (two items stop at their error) and the fused mix as batched
    appends, with single calls around them, against the reference's loops on the oracle: same columns, same first
    unsatisfied row, same sigma (ragged items are linked through the call's prefix sums)"""
    from oracle import pyoracle as po
    import ctypes as C
    batch = 61
    dev, ora = pg.StandardComposer(engine, 1 << 17, 1 << 17), po.Composer()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    rnd = synth.splitmix64(batch, 31)
    bounds = synth.scalars_from_ints([(1 << (3 + int(r) % 250)) + int(r) % 7 for r in rnd])
    wit = synth.scalars_from_ints([int(r) % (1 << (2 + int(r >> np.uint64(8)) % 253)) for r in synth.splitmix64(batch, 32)])
    nb = C.c_uint64()
    res, nbits = dev.max_bound_ragged_batch(t(bounds), t(wit))
    ores, onb = [], []
    for bnd, w in zip(bounds, wit):
        ores.append(int(ora.L.max_bound(ora.c, po.fr(bnd), ora.allocate(w), C.byref(nb))))
        onb.append(nb.value)
    assert list(res.cpu().numpy().view(np.uint64)) == ores and nbits.cpu().numpy().tolist() == onb
    assert len(set(onb)) > 20      # really ragged
    y = pg.conditionally_select_one(dev, ores[0], ores[-1])
    assert y == int(ora.L.conditionally_select_one(ora.c, ores[0], ores[-1]))

    vals = synth.random_scalars(batch, 33)
    vals[[4, 40]] = 0
    first = dev.add_input_batch(t(vals))
    ovars = [ora.add_input(v) for v in vals]
    assert first == ovars[0]
    err, nerr = dev.is_non_zero_batch(torch.arange(first, first + batch, dtype=torch.int64, device="cuda:0"))
    oerr = [int(ora.L.is_non_zero(ora.c, ovars[i], po.fr(vals[i]))) for i in range(batch)]
    assert err.cpu().numpy().tolist() == oerr and nerr == 2

    v, yv, s, a, b = (synth.random_scalars(batch, 34 + k) for k in range(5))
    v[[0, 17]] = 0
    s = synth.scalars_from_ints([int(x) & 1 for x in synth.splitmix64(batch, 39)])
    b[::2] = a[::2]
    res2, err2, nerr2 = dev.scalar_mix_batch(t(v), t(yv), t(s), t(a), t(b))
    ores2, oerr2 = [], []
    for i in range(batch):
        vv, yy, ss = ora.add_input(v[i]), ora.add_input(yv[i]), ora.add_input(s[i])
        aa, bb = ora.allocate(a[i]), ora.allocate(b[i])
        oerr2.append(int(ora.L.is_non_zero(ora.c, vv, po.fr(v[i]))))
        ores2.append([int(ora.L.conditionally_select_one(ora.c, yy, ss)), int(ora.L.maybe_equal(ora.c, aa, bb))])
    assert res2.cpu().numpy().view(np.uint64).tolist() == ores2 and err2.cpu().numpy().tolist() == oerr2 and nerr2 == 2
    pg.conditionally_select_zero(dev, ores2[3][1], ores[5])
    ora.L.conditionally_select_zero(ora.c, ores2[3][1], ores[5])
    same(dev, ora)
    assert dev.check() == ora.check()
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
    _sigma_properties(dev, padded)


class DeviceOps:
    """tests/refcases.py:full_circuit on the device composer (the range_check loop as ONE batched append when `batched`)"""

    def __init__(self, dev, batched):
        self.dev, self.batched = dev, batched

    def add_input(self, v): return self.dev.add_input(S(v))
    def allocate(self, v): return pg.AllocatedScalar.allocate(self.dev, S(v))
    def max_bound(self, mx, a): return pg.max_bound(self.dev, S(mx), a)[0]
    def maybe_equal(self, a, b): return pg.maybe_equal(self.dev, a, b)
    def is_non_zero(self, var, value): pg.is_non_zero(self.dev, var, S(value))
    def conditionally_select_one(self, y, s): return pg.conditionally_select_one(self.dev, y, s)
    def conditionally_select_zero(self, x, s): return pg.conditionally_select_zero(self.dev, x, s)
    def constrain_to_constant(self, a, c, pi): self.dev.constrain_to_constant(a, S(c), S(pi) if pi is not None else None)
    def boolean_gate(self, a): self.dev.boolean_gate(a)

    def range_check_loop(self, mn, mx, ws):
        if not self.batched:
            return [pg.range_check(self.dev, S(mn), S(mx), self.allocate(w)) for w in ws]
        wit = torch.from_numpy(synth.scalars_from_ints(ws).view(np.int64)).to("cuda:0")
        return [int(r) for r in self.dev.range_check_batch(S(mn), S(mx), wit)]


@pytest.mark.parametrize("batched", [False, True])
def test_full_composer_golden_fixture(engine, batched):
    """tests/golden/composer_full.npz (frozen from the big-int model; the C oracle reproduces it in the CPU suite): the
    device composer's live columns, q_4 / q_arith / w_4, wire values, dense public inputs and sigma, from row 0"""
    import os
    from tests.refcases import full_circuit
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composer_full.npz")))
    dev = pg.StandardComposer(engine)
    full_circuit(DeviceOps(dev, batched))
    assert dev.check() == -1
    exp = dev.export()
    for k in COLS:
        assert np.array_equal(exp[k], g[k]), k
    mat = {k: v.cpu().numpy().view(np.uint64) for k, v in dev.materialize().items()}
    for k in ("q_4", "q_arith", "w_4"):
        assert np.array_equal(mat[k], g[k]), k
    for k in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        assert not mat[k].any()
    for wire, col in (("w_l", "w_l_value"), ("w_r", "w_r_value"), ("w_o", "w_o_value"), ("w_4", "w_4_value")):
        assert np.array_equal(mat[col], g["var_values"][g[wire].astype(np.int64)]), col
    assert np.array_equal(dev.construct_dense_pi_vec().cpu().numpy().view(np.uint64), g["dense_pi"])
    padded = int(g["padded_n"][0])
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64).reshape(-1), g["sigma"])


def test_composer_grows(engine):
    """pg_composer_auto_grow / pg_composer_reserve: a composer created with room for 8 rows and 8 Variables builds the
    golden circuit (single calls and a batched append) by doubling; nothing already appended is lost"""
    import os
    from tests.refcases import full_circuit
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composer_full.npz")))
    for batched in (False, True):
        dev = pg.StandardComposer(engine, gate_capacity=8, var_capacity=8)
        dev.auto_grow()
        full_circuit(DeviceOps(dev, batched))
        gc, vc = dev.capacity()
        assert gc >= dev.circuit_size() > 8 and vc >= dev.num_variables() > 8
        assert dev.check() == -1
        exp = dev.export()
        for k in COLS:
            assert np.array_equal(exp[k], g[k]), k
        padded = int(g["padded_n"][0])
        assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64).reshape(-1), g["sigma"])
    dev = pg.StandardComposer(engine, gate_capacity=8, var_capacity=8)
    with pytest.raises(pg.PgError, match="capacity"):
        pg.range_check(dev, S(0), S(2**64), pg.AllocatedScalar.allocate(dev, S(1)))
    before = dev.export()
    dev.reserve(1000, 2000)
    assert dev.capacity() == (1000, 2000)
    dev.reserve(10, 10)                 # never shrinks
    assert dev.capacity() == (1000, 2000)
    after = dev.export()
    for k in COLS:
        assert np.array_equal(before[k], after[k]), k
    r = pg.range_check(dev, S(0), S(2**64), pg.AllocatedScalar.allocate(dev, S(1)))
    assert dev.value(r).to_int() == 1 and dev.check() == -1


def test_empty_batches_and_minimal_composer(engine):
    """zero-item batched appends are no-ops; sigma of the smallest composers (only zero_var's row; the initial three
    rows) equals the oracle's, for padded sizes from exactly n upwards"""
    from oracle import pyoracle as po
    for dummy in (False, True):
        dev, ora = pg.StandardComposer(engine, with_dummy=dummy), po.Composer(dummy)
        same(dev, ora)
        n = dev.circuit_size()
        for padded in (n, n + 1, 8):
            assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded)), (dummy, padded)
        empty = torch.empty((0, 4), dtype=torch.int64, device="cuda:0")
        novars = torch.empty((0,), dtype=torch.int64, device="cuda:0")
        assert dev.range_check_batch(S(0), S(2**64), empty).numel() == 0
        assert dev.max_bound_batch(S(2**64), empty)[0].numel() == 0
        assert dev.max_bound_ragged_batch(empty, empty)[0].numel() == 0
        assert dev.maybe_equal_batch(novars, novars).numel() == 0
        assert dev.is_non_zero_batch(novars)[1] == 0
        assert dev.scalar_mix_batch(empty, empty, empty, empty, empty)[2] == 0
        assert dev.add_input_batch(empty) == dev.num_variables()
        same(dev, ora)
        assert dev.check() == -1
        assert np.array_equal(dev.permutation(8).cpu().numpy().view(np.uint64), ora.sigma(8))


def run_fuzz_program(dev, ora, seed, wit_seed=None, steps=30):
    """a random program of `steps` composer operations -- single calls and every batched append, on random earlier Variables,
    random bounds and batch sizes -- on the device composer and, call for call, on the oracle's.  Everything PUBLIC (operations,
    batch sizes, bounds, selectors, constants, which earlier Variables a call refers to) comes from `seed`; the WITNESS scalars
    from `wit_seed` (default: the same stream, as before the two were told apart).  Returns the log of (operation, size)."""
    from oracle import pyoracle as po
    import ctypes as C
    import random
    rng = random.Random(seed)
    wrng = rng if wit_seed is None else random.Random(wit_seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    tv = lambda xs: torch.tensor(xs, dtype=torch.int64, device="cuda:0")
    F = lambda x: po.fr(synth.mont(x))
    nb = C.c_uint64()

    def rand_scalars(k):
        return synth.scalars_from_ints([wrng.choice([0, 1, wrng.randrange(1 << 16), wrng.randrange(po_Q)]) for _ in range(k)])

    po_Q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    log = []
    for step in range(steps):
        nv = dev.num_variables()
        assert nv == ora.num_vars
        op = rng.choice(["add_input", "rc", "rc_alloc", "mb", "mb_alloc", "mb_ragged", "dec", "sel0", "sel1", "meq", "inz", "mix",
                         "bool", "single_sel0", "ctc", "alloc_batch", "poly_b", "add_b", "mul_b", "ctc_b", "bool_b"])
        k = rng.randrange(1, 40)
        if seed > 100 and op in ("sel0", "sel1", "meq", "inz", "mix", "alloc_batch", "poly_b", "add_b", "mul_b", "ctc_b", "bool_b"):
            k = rng.randrange(400, 2500)  # small items by the thousand: many groups per segment, a shorter last group
        old = [rng.randrange(nv) for _ in range(k)]
        old2 = [rng.randrange(nv) for _ in range(k)]
        log.append((op, k))
        if op == "add_input":
            x = wrng.randrange(po_Q)
            assert dev.add_input(S(x)) == ora.add_input(synth.mont(x))
        elif op == "alloc_batch":
            w = rand_scalars(k)
            assert dev.add_input_batch(t(w)) == int(ora.allocate(w[0]).var)
            for x in w[1:]:
                ora.allocate(x)
        elif op in ("rc", "rc_alloc"):
            mx = rng.randrange(2, 1 << rng.randrange(2, 40))
            mn = rng.randrange(0, mx)
            w = rand_scalars(k)
            if op == "rc":
                r = dev.range_check_batch(S(mn), S(mx), t(w))
                o = [ora.L.range_check(ora.c, F(mn), F(mx), ora.allocate(x)) for x in w]
            else:
                vals = [ora.L.composer_value(ora.c, v) for v in old]
                wv = np.array([[x.l[i] for i in range(4)] for x in vals], dtype=np.uint64)
                r = dev.range_check_allocated_batch(S(mn), S(mx), tv(old), t(wv))
                o = [ora.L.range_check(ora.c, F(mn), F(mx), po.AllocatedScalar(v, x)) for v, x in zip(old, vals)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op in ("mb", "mb_alloc"):
            mx = rng.randrange(1, 1 << rng.randrange(1, 60))
            w = rand_scalars(k)
            if op == "mb":
                r, _ = dev.max_bound_batch(S(mx), t(w))
                o = [ora.L.max_bound(ora.c, F(mx), ora.allocate(x), C.byref(nb)) for x in w]
            else:
                vals = [ora.L.composer_value(ora.c, v) for v in old]
                wv = np.array([[x.l[i] for i in range(4)] for x in vals], dtype=np.uint64)
                r, _ = dev.max_bound_allocated_batch(S(mx), tv(old), t(wv))
                o = [ora.L.max_bound(ora.c, F(mx), po.AllocatedScalar(v, x), C.byref(nb)) for v, x in zip(old, vals)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op == "mb_ragged":
            bounds = [rng.randrange(1, 1 << rng.randrange(1, 100)) for _ in range(k)]
            w = rand_scalars(k)
            r, _ = dev.max_bound_ragged_batch(t(synth.scalars_from_ints(bounds)), t(w))
            o = [ora.L.max_bound(ora.c, F(b), ora.allocate(x), C.byref(nb)) for b, x in zip(bounds, w)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op == "dec":
            bits = rng.randrange(0, 30)
            vals = [ora.L.composer_value(ora.c, v) for v in old]
            wv = np.array([[x.l[i] for i in range(4)] for x in vals], dtype=np.uint64)
            r = dev.scalar_decomposition_batch(bits, tv(old), t(wv))
            o = [ora.L.scalar_decomposition_gadget(ora.c, bits, po.AllocatedScalar(v, x), None) for v, x in zip(old, vals)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op in ("sel0", "sel1"):
            fn = dev.conditionally_select_zero_batch if op == "sel0" else dev.conditionally_select_one_batch
            ofn = ora.L.conditionally_select_zero if op == "sel0" else ora.L.conditionally_select_one
            r = fn(tv(old), tv(old2))
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(ofn(ora.c, a, b)) for a, b in zip(old, old2)]
        elif op == "meq":
            r = dev.maybe_equal_batch(tv(old), tv(old2))
            o = [ora.L.maybe_equal(ora.c, po.AllocatedScalar(a, ora.L.composer_value(ora.c, a)),
                                   po.AllocatedScalar(b, ora.L.composer_value(ora.c, b))) for a, b in zip(old, old2)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op == "inz":
            err, nerr = dev.is_non_zero_batch(tv(old))
            o = [int(ora.L.is_non_zero(ora.c, v, ora.L.composer_value(ora.c, v))) for v in old]
            assert err.cpu().numpy().tolist() == o and nerr == sum(o)
        elif op == "mix":
            v, y, s, a, b = (rand_scalars(k) for _ in range(5))
            r, err, nerr = dev.scalar_mix_batch(t(v), t(y), t(s), t(a), t(b))
            o, oe = [], []
            for i in range(k):
                vv, yy, ss = ora.add_input(v[i]), ora.add_input(y[i]), ora.add_input(s[i])
                aa, bb = ora.allocate(a[i]), ora.allocate(b[i])
                oe.append(int(ora.L.is_non_zero(ora.c, vv, po.fr(v[i]))))
                o.append([int(ora.L.conditionally_select_one(ora.c, yy, ss)), int(ora.L.maybe_equal(ora.c, aa, bb))])
            assert r.cpu().numpy().view(np.uint64).tolist() == o and err.cpu().numpy().tolist() == oe
        elif op == "poly_b":
            old3 = [rng.randrange(nv) for _ in range(k)]
            q = [rng.choice([0, 1, po_Q - 1, rng.randrange(po_Q)]) for _ in range(5)]
            dev.poly_gate_batch(tv(old), tv(old2), tv(old3), *[S(x) for x in q])
            for a, b, c3 in zip(old, old2, old3):
                ora.L.composer_poly_gate(ora.c, a, b, c3, *[F(x) for x in q], None)
        elif op in ("add_b", "mul_b"):
            q1, q2, q3 = (rng.choice([1, po_Q - 1, rng.randrange(po_Q)]) for _ in range(3))
            if op == "add_b":
                r = dev.add_batch(S(q1), tv(old), S(q2), tv(old2), S(q3))
                o = [ora.L.composer_add(ora.c, F(q1), a, F(q2), b, F(q3), None) for a, b in zip(old, old2)]
            else:
                r = dev.mul_batch(S(q1), tv(old), tv(old2), S(q3))
                o = [ora.L.composer_mul(ora.c, F(q1), a, b, F(q3), None) for a, b in zip(old, old2)]
            assert r.cpu().numpy().view(np.uint64).tolist() == [int(x) for x in o]
        elif op == "ctc_b":
            x = rng.randrange(po_Q)
            dev.constrain_to_constant_batch(tv(old), S(x))
            for a in old:
                ora.L.composer_constrain_to_constant(ora.c, a, F(x), None)
        elif op == "bool_b":
            dev.boolean_gate_batch(tv(old))
            for a in old:
                ora.L.composer_boolean_gate(ora.c, a)
        elif op == "bool":
            dev.boolean_gate(old[0])
            ora.L.composer_boolean_gate(ora.c, old[0])
        elif op == "single_sel0":
            assert pg.conditionally_select_zero(dev, old[0], old2[0]) == int(ora.L.conditionally_select_zero(ora.c, old[0], old2[0]))
        elif op == "ctc":
            x = rng.randrange(po_Q)
            dev.constrain_to_constant(old[0], S(x), None)
            ora.L.composer_constrain_to_constant(ora.c, old[0], F(x), None)
    return log


@pytest.mark.parametrize("seed", list(range(1, 11)) + [101, 102, 103])
def test_fuzz_composer_programs(engine, seed):
    """random programs of 30 composer operations (run_fuzz_program) replayed on the oracle call for call: same columns, same
    first unsatisfied row, same sigma.  (Segments of every shape, gaps between them, references across segments, hot Variables.)"""
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 14, 1 << 14), po.Composer()
    dev.auto_grow()
    log = run_fuzz_program(dev, ora, seed)
    try:
        same(dev, ora)
        assert dev.check() == ora.check()
        n = dev.circuit_size()
        padded = 1 << (n - 1).bit_length()
        assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
        # unpadded: sigma's four columns n entries apart -- for an odd n no column but the first is 16-byte aligned (the closed-form
        # kernels' two-gates-per-store path falls back to single entries)
        assert np.array_equal(dev.permutation(n).cpu().numpy().view(np.uint64), ora.sigma(n))
        # the sorted list of foreign positions reserved far too short (fewer entries than the closed-form segments' own slots): the pass
        # reports its size and runs a second time -- holes, slots and all
        dev.permutation_reserve(16)
        assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))
        dev.permutation_reserve(0)
        m, cols = dev.materialize(), dev.device_columns()
        for wname in ("w_l", "w_r", "w_o"):   # the wire-value columns == the assignments of the rows' Variables
            assert torch.equal(m[wname + "_value"], cols.var_values[getattr(cols, wname)[:n]]), wname
    except AssertionError as e:
        raise AssertionError(f"seed {seed}, program {log}: {e}")


def test_spread_columns_keeps_the_circuit(engine):
    """pg_composer_spread_columns: the columns move into one block (selector columns a stride apart) in the middle of a program,
    the composer grows in that layout, and moves back: columns, first unsatisfied row and sigma == the oracle's throughout"""
    import ctypes as C
    from oracle import pyoracle as po
    from plonk_gadgets_amd import _lib
    dev, ora = pg.StandardComposer(engine, 1 << 12, 1 << 12), po.Composer()
    dev.auto_grow()
    run_fuzz_program(dev, ora, 301, steps=8)
    same(dev, ora)
    dev.spread_columns(6 / 1024)  # 6 MiB between the selector columns
    cc = _lib.ColumnsC()
    assert dev._lib.pg_composer_columns(dev._h, C.byref(cc)) == 0
    sel = [cc.q_m, cc.q_l, cc.q_r, cc.q_o, cc.q_c]
    gate_cap = dev.capacity()[0]
    want = max(6 << 20, (gate_cap * 32 + (2 << 20) - 1) // (2 << 20) * (2 << 20))
    assert [b - a for a, b in zip(sel, sel[1:])] == [want] * 4 and cc.w_l > cc.q_c and cc.var_values > cc.w_o
    same(dev, ora)
    cap0 = dev.capacity()
    run_fuzz_program(dev, ora, 302, steps=25)  # (grows: seeds above 100 append items by the thousand)
    assert dev.capacity() != cap0, "the program was meant to outgrow the composer"
    same(dev, ora)
    assert dev.check() == ora.check()
    dev.spread_columns(0)
    same(dev, ora)
    run_fuzz_program(dev, ora, 303, steps=6)
    same(dev, ora)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded))


def test_two_host_threads_two_composers():
    """one engine + composer per host thread (ctypes releases the GIL during the calls): both threads build the golden
    circuit twenty times over, each result equals the fixture"""
    import os
    import threading
    from tests.refcases import full_circuit
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composer_full.npz")))
    padded = int(g["padded_n"][0])
    failures = []

    def work(tid):
        try:
            eng = pg.Engine(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for it in range(20):
                    dev = pg.StandardComposer(eng, gate_capacity=64, var_capacity=64)
                    dev.auto_grow()
                    full_circuit(DeviceOps(dev, batched=(it + tid) % 2 == 0))
                    exp = dev.export()
                    for k in COLS:
                        assert np.array_equal(exp[k], g[k]), (tid, it, k)
                    assert np.array_equal(dev.permutation(padded).cpu().numpy().view(np.uint64).reshape(-1), g["sigma"]), (tid, it)
                    assert dev.check() == -1
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            import traceback
            failures.append(traceback.format_exc())

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures[0]


def test_materialize_rows_of_batched_calls(engine):
    """pg_composer_materialize on a circuit whose rows come from every kind of batched append -- big enough that each call is
    served by the windowed kernel (csrc/materialize.hpp: the items' own Variables from LDS; the ladder gadgets' wires in closed
    form, never read back) -- with single calls in between (the generic gather).  Expected: the sentinel-filled outputs equal
    variables[w] gathered by torch from the composer's own columns, row for row; q_arith = 1, the other constant columns 0,
    w_4 = zero_var; nothing beyond the circuit's last row is written."""
    import bench
    n_items = 70
    dev = pg.StandardComposer(engine, 1 << 18, 1 << 18)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    wit = d(synth.random_scalars(n_items, 11))  # (the mix's inputs below are random field elements: none is 0)
    small = d(synth.uniform_below(n_items, 300_000, seed=12))
    x = dev.add_input(S(9))
    dev.range_check_batch(S(0), S(2**254), wit[:9])                    # n = 255: one item per window, lines shared with the next
    dev.boolean_gate(dev.add_witness_to_circuit_description(S(1)))
    dev.range_check_batch(S(50_000), S(250_000), small)                # n = 19: eight items per window
    first = dev.add_input_batch(wit)                                  # Variables allocated elsewhere, then gadgets on them
    vars_ = torch.arange(first, first + n_items, dtype=torch.int64, device="cuda:0")
    dev.range_check_allocated_batch(S(0), S(2**200), vars_, wit)
    dev.assert_equal(x, x)
    dev.max_bound_batch(S(2**100), wit)
    dev.max_bound_allocated_batch(S(2**64), vars_, wit)
    dev.scalar_decomposition_batch(64, vars_, wit)
    dev.scalar_decomposition_batch(256, vars_[:20], wit[:20])           # (range.rs:134 allows 256)
    mr, wt = bench.c4_inputs(40, seed=5)
    dev.max_bound_ragged_batch(d(mr), d(wt))                           # ragged: wires read from the columns
    v, y, s, a, b = bench.mix_inputs(900, seed=6)
    v[[3, 500]] = 0
    assert dev.scalar_mix_batch(d(v), d(y), d(s), d(a), d(b))[2] == 2  # two items stop at is_non_zero's error: a ragged call
    dev.boolean_gate(dev.add_witness_to_circuit_description(S(0)))       # (an odd row in between)
    assert dev.scalar_mix_batch(*[d(t) for t in bench.mix_inputs(700, seed=16)])[2] == 0  # every item complete: its wires from the table
    big = torch.arange(first, first + n_items, dtype=torch.int64, device="cuda:0").repeat(70)
    dev.maybe_equal_batch(big, big.flip(0))
    dev.add_batch(S(3), big, S(5), big.flip(0), S(1))
    dev.mul(S(2), x, x, S(0), None)
    assert dev.check() == -1
    n = dev.circuit_size()
    cols = dev.device_columns()
    m = dev.materialize()
    one = torch.tensor(np.array(synth.mont(1), dtype=np.uint64).view(np.int64), device="cuda:0")
    for wname in ("w_l", "w_r", "w_o"):
        w = getattr(cols, wname)[:n]
        exp, got = cols.var_values[w], m[wname + "_value"]
        if not torch.equal(got, exp):
            bad = int((got != exp).any(dim=1).nonzero()[0])
            raise AssertionError(f"{wname}_value differs first at row {bad} (Variable {int(w[bad])})")
    assert bool((m["q_arith"] == one).all())
    for k in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        assert not bool(m[k].any()), k
    w4 = m["w_4"]
    assert int((w4 != 0).sum()) == 1 and int(w4[1]) == 2  # the dummy rows' live fourth wire
    assert torch.equal(m["w_4_value"], cols.var_values[w4])


@pytest.mark.parametrize("lead", [0, 1, 5])
@pytest.mark.parametrize("kind,extra", [("max_bound", 1), ("max_bound", 2), ("range_check", 1), ("range_check", 0),
                                        ("max_bound_allocated", 1), ("range_check_allocated", 1), ("decomposition", 1), ("decomposition", 2)])
def test_materialize_short_items_and_a_short_last_group(engine, kind, extra, lead):
    """the windowed kernel writes whole lines: a group's range of rows is cut at multiples of 16 rows, so the first rows of the next
    group come along with it (csrc/materialize.hpp: their values through the loader wave's side table).  The corner: items SHORTER
    than 16 rows (several come along, over more than one item) and a LAST group of fewer rows than that, at the very end of the
    circuit -- nothing past the circuit's last row may be touched (outputs 64 rows longer, sentinel-filled), whatever the first
    row's alignment (`lead` single rows before the call)."""
    import ctypes as C
    from plonk_gadgets_amd import _lib
    dev = pg.StandardComposer(engine, 1 << 15, 1 << 21)
    for _ in range(lead):
        dev.add_input(S(7))
    bound = S(3) if kind.startswith("max_bound") else S(2)

    def append(comp, w):
        """the call under test; the `_allocated` kinds and scalar_decomposition on Variables allocated by a call before it (their
        values reach the rows through the loader wave's witness table, the rows that come along through its side table)"""
        if kind == "max_bound":
            return comp.max_bound_batch(bound, w)
        if kind == "range_check":
            return comp.range_check_batch(S(0), bound, w)
        first = comp.add_input_batch(w)
        vars_ = torch.arange(first, first + w.shape[0], dtype=torch.int64, device="cuda:0")
        if kind == "max_bound_allocated":
            return comp.max_bound_allocated_batch(bound, vars_, w)
        if kind == "range_check_allocated":
            return comp.range_check_allocated_batch(S(0), bound, vars_, w)
        return comp.scalar_decomposition_batch(2, vars_, w)

    probe = pg.StandardComposer(engine, 1024, 1024)
    one = torch.from_numpy(np.ascontiguousarray(synth.uniform_below(1, 2, seed=1)).view(np.int64)).to("cuda:0")
    v0, r0 = probe.num_variables(), probe.circuit_size()
    append(probe, one)
    V, L = probe.num_variables() - v0 - (0 if kind in ("max_bound", "range_check") else 1), probe.circuit_size() - r0
    assert L < 16 or kind.startswith("range_check")
    group = 1040 // V
    items = group * (4096 // (group * L) + 1) + extra
    wit = torch.from_numpy(np.ascontiguousarray(synth.uniform_below(items, 2, seed=3)).view(np.int64)).to("cuda:0")
    append(dev, wit)
    assert dev.check() == -1
    n, cols = dev.circuit_size(), dev.device_columns()
    assert n >= 4096 + lead
    names = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add")
    vals = ("w_l_value", "w_r_value", "w_o_value", "w_4_value")
    t = {k: torch.full((n + 64, 4), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda:0") for k in names + vals}
    t["w_4"] = torch.full((n + 64,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda:0")
    fc = _lib.FullColumnsC(**{k: v.data_ptr() for k, v in t.items()})
    assert dev._lib.pg_composer_materialize(dev._h, C.byref(fc)) == 0
    torch.cuda.synchronize()
    for k, v in t.items():
        assert bool((v[n:] == 0x5A5A5A5A5A5A5A5A).all()), f"{k}: rows past the circuit's last were written"
    for wname in ("w_l", "w_r", "w_o"):
        w = getattr(cols, wname)[:n]
        exp, got = cols.var_values[w], t[wname + "_value"][:n]
        if not torch.equal(got, exp):
            bad = int((got != exp).any(dim=1).nonzero()[0])
            raise AssertionError(f"{wname}_value differs first at row {bad} of {n} (Variable {int(w[bad])}; items of {L} rows, {V} Variables)")
    assert torch.equal(t["w_4_value"][:n], cols.var_values[t["w_4"][:n]])


def test_sigma_when_the_sparse_list_is_reserved_too_small(engine):
    """pg_composer_permutation_reserve with a figure far below what the circuit puts on the sparse list: the ladder segments' closed-form
    slots alone (four per item of range_check on witnesses from elsewhere, csrc/permutation.hpp) exceed it, nothing may be written past
    the list's end, the pass reports what it needs and runs once more -- sigma is the same as with room from the start, and as the
    oracle's."""
    import ctypes as C
    from oracle import pyoracle as po
    F = lambda x: po.fr(synth.mont(x))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    n_items = 60
    vals = [int(v) for v in synth.splitmix64(n_items, 5) % np.uint64(300_000)]
    wit = synth.scalars_from_ints(vals)
    sigmas = []
    for reserve in (None, 8):
        dev, ora = pg.StandardComposer(engine, 1 << 14, 1 << 17), po.Composer()
        first = dev.add_input_batch(t(wit))
        allocs = [ora.allocate(w) for w in wit]
        wv = torch.arange(first, first + n_items, dtype=torch.int64, device="cuda:0")
        r = dev.range_check_allocated_batch(S(50_000), S(250_000), wv, t(wit))
        res = [int(ora.L.range_check(ora.c, F(50_000), F(250_000), a)) for a in allocs]
        assert list(r.cpu().numpy().view(np.uint64)) == res
        dev.assert_equal(res[0], res[1])  # (rows of a single call behind the batch: the counter continues behind the reserved slots)
        ora.L.composer_assert_equal(ora.c, res[0], res[1])
        n = dev.circuit_size()
        padded = 1 << (n - 1).bit_length()
        if reserve is not None:
            dev.permutation_reserve(reserve)
        got = dev.permutation(padded).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ora.sigma(padded))
        sigmas.append(got)
    assert np.array_equal(sigmas[0], sigmas[1])


def test_f_rows_of_runs_of_single_calls(engine):
    """A circuit built the reference's way -- ONE allocate + range_check at a time (tests/range_gadgets_tests.rs:29-44), the command
    queue on -- leaves runs of identical calls, which a flush sends out as one batched launch and (csrc/capi_composer.inc,
    add_run_segment) registers with the batched append's footprint: the f-rows then take the closed forms.  Runs long enough to be
    registered (>= 4096 rows) of all four shapes: allocate + range_check pairs, allocate + max_bound pairs, and both gadgets on
    witnesses allocated beforehand, with gates between the runs, a run cut in two by a flush, and results used by later rows.
    sigma == the oracle's bookkeeping; the wire-value columns == variables[w] gathered by torch; check() passes."""
    import ctypes as C
    from oracle import pyoracle as po
    dev, ora = pg.StandardComposer(engine, 1 << 17, 1 << 17), po.Composer()
    dev.queue(True)
    F = lambda x: po.fr(synth.mont(x))
    mn, mx, mb = 50_000, 250_000, 2**30
    nb = C.c_uint64()
    vals = [int(v) for v in synth.splitmix64(400, 77) % np.uint64(300_000)]
    res = []
    for k, v in enumerate(vals[:70]):       # 70 x (allocate, range_check): n = 18, 83 rows each
        a = pg.AllocatedScalar.allocate(dev, S(v))
        res.append(pg.range_check(dev, S(mn), S(mx), a))
        assert res[-1] == int(ora.L.range_check(ora.c, F(mn), F(mx), ora.allocate(synth.scalars_from_ints([v])[0])))
        if k == 40:
            dev.flush()                      # (the run goes out in two launches: one footprint all the same)
    dev.boolean_gate(dev.add_witness_to_circuit_description(S(1)))
    ora.L.composer_boolean_gate(ora.c, ora.L.composer_add_witness_to_circuit_description(ora.c, F(1)))
    for v in vals[70:140]:                  # 70 x (allocate, max_bound): n = 30, 65 rows each
        a = pg.AllocatedScalar.allocate(dev, S(v))
        r, _ = pg.max_bound(dev, S(mb), a)
        res.append(r)
        assert r == int(ora.L.max_bound(ora.c, F(mb), ora.allocate(synth.scalars_from_ints([v])[0]), C.byref(nb)))
    allocs = [pg.AllocatedScalar.allocate(dev, S(v)) for v in vals[140:280]]
    oallocs = [ora.allocate(synth.scalars_from_ints([v])[0]) for v in vals[140:280]]
    for a, oa in zip(allocs[:70], oallocs[:70]):    # gadgets on witnesses allocated beforehand
        res.append(pg.range_check(dev, S(mn), S(mx), a))
        assert res[-1] == int(ora.L.range_check(ora.c, F(mn), F(mx), oa))
    for a, oa in zip(allocs[70:], oallocs[70:]):
        r, _ = pg.max_bound(dev, S(mb), a)
        res.append(r)
        assert r == int(ora.L.max_bound(ora.c, F(mb), oa, C.byref(nb)))
    z = pg.conditionally_select_zero(dev, res[3], res[-1])
    assert z == int(ora.L.conditionally_select_zero(ora.c, res[3], res[-1]))
    dev.assert_equal(res[100], res[100])
    ora.L.composer_assert_equal(ora.c, res[100], res[100])
    same(dev, ora)
    n = dev.circuit_size()
    assert n > 4 * 4096
    padded = 1 << (n - 1).bit_length()
    got, exp = dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded)
    if not np.array_equal(got, exp):
        w, g = np.argwhere(got != exp)[0]
        raise AssertionError(f"sigma differs first at wire {w}, gate {g}: {got[w, g]} != {exp[w, g]}")
    _sigma_properties(dev, padded)
    cols, m = dev.device_columns(), dev.materialize()
    for wname in ("w_l", "w_r", "w_o"):
        w = getattr(cols, wname)[:n]
        expv, gotv = cols.var_values[w], m[wname + "_value"]
        if not torch.equal(gotv, expv):
            bad = int((gotv != expv).any(dim=1).nonzero()[0])
            raise AssertionError(f"{wname}_value differs first at row {bad} (Variable {int(w[bad])})")


@pytest.mark.parametrize("bits", [1, 2, 3, 19, 128, 251, 252, 253, 254, 255])
def test_ladder_sigma_closed_form_over_the_ladder_lengths(engine, bits):
    """sigma of the ladder gadgets' rows is written in closed form (csrc/permutation.hpp, perm_ladder_kernel: which positions hold
    one Variable is a function of a row's place in its item): every kind -- range_check and max_bound, allocating and on witnesses
    allocated elsewhere, scalar_decomposition -- over ladder lengths from the shortest (n = 2) to the longest (255; 252 for a bound of
    255 bits), with a witness Variable that IS zero_var (its positions belong to the zero chain, not to the sparse list), results used by
    later rows (the sparse list splices them into the items' own cycles) and an odd first row.  == the oracle's bookkeeping."""
    import ctypes as C
    from oracle import pyoracle as po
    mx = 2**bits if bits < 255 else Q - 1  # (bound - 1 of `bits` bits: n = bits + 1, and 252 for 255 bits)
    mn = mx // 3
    batch = 5
    dev, ora = pg.StandardComposer(engine, 1 << 15, 1 << 15), po.Composer()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
    F = lambda x: po.fr(synth.mont(x))
    ws = [0, mn, mx - 1, mx, Q - 1][:batch]
    wit = synth.scalars_from_ints(ws)
    dev.boolean_gate(dev.add_witness_to_circuit_description(S(1)))      # (an odd number of rows before the first batch)
    ora.L.composer_boolean_gate(ora.c, ora.L.composer_add_witness_to_circuit_description(ora.c, F(1)))
    first = dev.add_input_batch(t(wit))
    allocs = [ora.allocate(w) for w in wit]
    assert first == int(allocs[0].var)
    # witnesses elsewhere: the allocated ones, and zero_var itself (value 0) in the middle
    wv_list = [first, first + 1, 0, first + 3, first + 4]
    wv = torch.tensor(wv_list, dtype=torch.int64, device="cuda:0")
    wvals = synth.scalars_from_ints([ws[0], ws[1], 0, ws[3], ws[4]])
    oallocs = [po.AllocatedScalar(v, po.fr(x)) for v, x in zip(wv_list, wvals)]
    nb = C.c_uint64()
    res = []
    r = dev.range_check_batch(S(mn), S(mx), t(wit))
    res += [int(ora.L.range_check(ora.c, F(mn), F(mx), ora.allocate(w))) for w in wit]
    assert list(r.cpu().numpy().view(np.uint64)) == res[-batch:]
    r = dev.range_check_allocated_batch(S(mn), S(mx), wv, t(wvals))
    res += [int(ora.L.range_check(ora.c, F(mn), F(mx), a)) for a in oallocs]
    assert list(r.cpu().numpy().view(np.uint64)) == res[-batch:]
    r, n1 = dev.max_bound_batch(S(mx), t(wit))
    res += [int(ora.L.max_bound(ora.c, F(mx), ora.allocate(w), C.byref(nb))) for w in wit]
    assert list(r.cpu().numpy().view(np.uint64)) == res[-batch:] and n1 == nb.value
    r, _ = dev.max_bound_allocated_batch(S(mx), wv, t(wvals))
    res += [int(ora.L.max_bound(ora.c, F(mx), a, C.byref(nb))) for a in oallocs]
    assert list(r.cpu().numpy().view(np.uint64)) == res[-batch:]
    nd = min(bits + 1, 256)
    r = dev.scalar_decomposition_batch(nd, wv, t(wvals))
    res += [int(ora.L.scalar_decomposition_gadget(ora.c, nd, a, None)) for a in oallocs]
    assert list(r.cpu().numpy().view(np.uint64)) == res[-batch:]
    # later rows that use the items' results and an item's inner Variable
    z = pg.conditionally_select_zero(dev, res[0], res[-1])
    assert z == int(ora.L.conditionally_select_zero(ora.c, res[0], res[-1]))
    dev.assert_equal(res[7], res[7])
    ora.L.composer_assert_equal(ora.c, res[7], res[7])
    same(dev, ora)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    got, exp = dev.permutation(padded).cpu().numpy().view(np.uint64), ora.sigma(padded)
    if not np.array_equal(got, exp):
        w, g = np.argwhere(got != exp)[0]
        raise AssertionError(f"sigma differs first at wire {w}, gate {g}: {got[w, g]} != {exp[w, g]} (n = {n1})")
    _sigma_properties(dev, padded)
