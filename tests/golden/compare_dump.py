#!/usr/bin/env python3
"""Compares a dump of the REAL reference's composers with the fixtures committed here -- the hand-off that turns
"parity unpinned" into pinned.

The reference is Rust and cannot run in the environment this repository is built in, so tests/golden/*.npz hold the
output of this repository's own restatement (oracle/model.py).  A maintainer with a Rust toolchain produces the same
arrays from dusk-plonk's `StandardComposer` itself:

    bindings/rust/fork/dump_columns.rs       -> add to a fork of dusk-plonk 0.8 (the columns are pub(crate) there)
    bindings/rust/examples/dump_golden.rs    -> examples/ of a plonk_gadgets v0.6.0 checkout patched to that fork
    cargo run --release --features std --example dump_golden -- /tmp/dump
    python tests/golden/compare_dump.py /tmp/dump

and this script diffs them, array by array, against the fixtures and names the FIRST difference together with the
recollection of dusk-plonk ([DEP-RECALL] in SURVEY.md section 3.4) it falsifies.

    python tests/golden/compare_dump.py --from-model DIR    writes the same layout from oracle/model.py (what the
                                                            fixtures were made from): the script's own self-test

Dump layout (one directory per circuit): <name>.u64 = raw little-endian u64 words;
q_m q_l q_r q_o q_c q_4 q_arith (n x 4), w_l w_r w_o w_4 (n), var_values (V x 4), dense_pi (n x 4),
sigma (4 x padded_n), meta = [n, V, zero_var, padded_n].  Rows and Variables are numbered from 0 (initial state
included); the partial fixtures (everything but composer_full) start at row 3 / Variable 5.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = ("range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65", "range_check_c2_n255",
         "max_bound_ref", "maybe_equal_ref", "scalar_mix", "composer_full")
SCALAR = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "dense_pi")
WIRES = ("w_l", "w_r", "w_o", "w_4")

# what a first difference in an array says about the recollection of dusk-plonk 0.8 the restatement rests on
DIAGNOSIS = {
    "meta": "composer sizes: the number of rows / Variables a gadget appends (SURVEY 3.1 table) or the initial state "
            "(StandardComposer::new(): zero_var + two dummy constraints = 3 rows, 5 Variables; SURVEY 3.4 'least certain item')",
    "initial": "the initial composer state: zero_var = Variable(0) via add_witness_to_circuit_description(0), then "
               "add_dummy_constraints() with variables (6, 1, 7, -20) [SURVEY 3.4, DEP-RECALL 'least certain item']",
    "w": "Variable numbering / wire placement: add_input hands out sequential indices; add/mul put the new Variable on "
         "w_o; max_bound/min_bound use w_r = x with q_r = 0 (src/range.rs:62,95); w_4 = zero_var [SURVEY 3.4 table]",
    "q": "the selector convention of a composer call: add -> (0, q_l, q_r, -1, q_c), mul -> (q_m, 0, 0, -1, q_c), "
         "boolean_gate -> (1, 0, 0, -1, 0), constrain_to_constant -> (0, 1, 0, 0, -c), assert_equal -> (0, 1, -1, 0, 0) "
         "[SURVEY 3.4 table], or q_arith = 1 / q_4 = 0 on every row of this path",
    "var_values": "BlsScalar's representation (four Montgomery limbs, fully reduced: R = 2^256 mod q) or the value the "
                  "composer computes for an add/mul output (c_eval) [SURVEY 8a row a15]",
    "dense_pi": "public-input bookkeeping: constrain_to_constant(a, c, Some(pi)) stores pi at the row's index and "
                "construct_dense_pi_vec scatters it [SURVEY 8f3]",
    "sigma": "Permutation::compute_sigma_permutations: every Variable's positions, in the order the rows recorded them, "
             "form one cycle (each position maps to the next, the last to the first) [SURVEY 8f2]",
}


def read_dump(d):
    out = {}
    meta = np.fromfile(os.path.join(d, "meta.u64"), dtype="<u8")
    n, nv, zero_var, padded = (int(x) for x in meta[:4])
    out["meta"] = (n, nv, zero_var, padded)
    for k in SCALAR:
        out[k] = np.fromfile(os.path.join(d, k + ".u64"), dtype="<u8").reshape(-1, 4)
    for k in WIRES:
        out[k] = np.fromfile(os.path.join(d, k + ".u64"), dtype="<u8")
    out["var_values"] = np.fromfile(os.path.join(d, "var_values.u64"), dtype="<u8").reshape(-1, 4)
    out["sigma"] = np.fromfile(os.path.join(d, "sigma.u64"), dtype="<u8")
    return out


def first_diff(a, b):
    if a.shape != b.shape:
        return "shapes differ: dump %s, fixture %s" % (a.shape, b.shape)
    if np.array_equal(a, b):
        return None
    idx = np.argwhere(a != b)[0]
    return "first difference at %s: dump %#x, fixture %#x" % (idx.tolist(), int(a[tuple(idx)]), int(b[tuple(idx)]))


def compare_case(name, dump, fix, full0):
    """-> list of (array, message, diagnosis key); empty when the circuit matches"""
    bad = []
    n, nv, zero_var, padded = dump["meta"]
    partial = name != "composer_full"
    g0, v0 = (3, 5) if partial else (0, 0)
    exp_n, exp_v = fix["q_m"].shape[0] + g0, fix["var_values"].shape[0] + v0
    if (n, nv) != (exp_n, exp_v) or zero_var != 0:
        bad.append(("meta", "dump has %d rows / %d Variables / zero_var %d, fixture %d / %d / 0" % (n, nv, zero_var, exp_n, exp_v),
                    "meta"))
        return bad
    # the initial state (rows 0-2, Variables 0-4) is the same in every circuit: composer_full's fixture holds it
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith"):
        m = first_diff(dump[k][:3], full0[k][:3])
        if m:
            bad.append((k + "[:3]", m, "initial"))
    for k in WIRES:
        m = first_diff(dump[k][:3], full0[k][:3])
        if m:
            bad.append((k + "[:3]", m, "initial"))
    m = first_diff(dump["var_values"][:5], full0["var_values"][:5])
    if m:
        bad.append(("var_values[:5]", m, "initial"))
    if bad:
        return bad
    for k in WIRES:
        if k in fix.files:
            m = first_diff(dump[k][g0:], fix[k])
            if m:
                bad.append((k, m, "w"))
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith"):
        if k in fix.files:
            m = first_diff(dump[k][g0:], fix[k])
            if m:
                bad.append((k, m, "q"))
    m = first_diff(dump["var_values"][v0:], fix["var_values"])
    if m:
        bad.append(("var_values", m, "var_values"))
    if "dense_pi" in fix.files:
        m = first_diff(dump["dense_pi"], fix["dense_pi"])
        if m:
            bad.append(("dense_pi", m, "dense_pi"))
    if "sigma" in fix.files:
        if padded != int(fix["padded_n"][0]):
            bad.append(("sigma", "padded_n %d, fixture %d" % (padded, int(fix["padded_n"][0])), "sigma"))
        else:
            m = first_diff(dump["sigma"].reshape(4, padded), fix["sigma"].reshape(4, padded))
            if m:
                bad.append(("sigma", m, "sigma"))
    return bad


def compare(dump_root, out=sys.stdout):
    full0 = np.load(os.path.join(HERE, "composer_full.npz"))
    failures = 0
    for name in CASES:
        d = os.path.join(dump_root, name)
        if not os.path.isdir(d):
            print("%-32s MISSING in the dump" % name, file=out)
            failures += 1
            continue
        bad = compare_case(name, read_dump(d), np.load(os.path.join(HERE, name + ".npz")), full0)
        if not bad:
            print("%-32s identical (rows, wires, limbs%s)" % (name, ", fourth wire, public inputs, sigma" if name == "composer_full" else ""),
                  file=out)
            continue
        failures += 1
        arr, msg, key = bad[0]
        print("%-32s DIFFERS: %s: %s" % (name, arr, msg), file=out)
        print("    -> falsifies: %s" % DIAGNOSIS[key], file=out)
        for arr, msg, _ in bad[1:4]:
            print("    also %s: %s" % (arr, msg), file=out)
    print("parity %s" % ("PINNED: the restatement reproduces the reference's composers bit for bit" if failures == 0
                         else "NOT pinned: %d of %d circuits differ" % (failures, len(CASES))), file=out)
    return failures


def write_from_model(root):
    """the same directory layout from oracle/model.py -- what make_golden.py froze; compare() of it must be clean"""
    sys.path.insert(0, ROOT)
    from oracle import model
    from tests.golden.make_golden import ModelOps
    from tests.refcases import MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES, full_circuit
    Q = model.Q

    def dump(m, name):
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        e = model.export(m, 0, 0)
        padded = 1 << (m.n - 1).bit_length()

        def put(k, a):
            np.asarray(a, dtype="<u8").reshape(-1).tofile(os.path.join(d, k + ".u64"))
        for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "var_values"):
            put(k, np.array(e[k], dtype=np.uint64))
        for k in ("w_l", "w_r", "w_o"):
            put(k, np.array(e[k], dtype=np.uint64))
        put("w_4", np.array(m.w_4, dtype=np.uint64))
        put("q_4", np.array([model.mont_limbs(x) for x in m.q_4], dtype=np.uint64))
        put("q_arith", np.array([model.mont_limbs(x) for x in m.q_arith], dtype=np.uint64))
        put("dense_pi", np.array([model.mont_limbs(x) for x in m.dense_pi()], dtype=np.uint64))
        put("sigma", np.array(m.sigma(padded), dtype=np.uint64))
        put("meta", np.array([m.n, len(m.variables), 0, padded], dtype=np.uint64))

    def rc(name, mn, mx, ws):
        m = model.Composer()
        for w in ws:
            model.range_check(m, mn, mx, model.AllocatedScalar.allocate(m, w))
        dump(m, name)
    rc("range_check_ref_50k_250k", 50_000, 250_000, [c[2] for c in RANGE_CHECK_CASES if c[0] == 50_000])
    rc("range_check_ref_2p126_2p127", 2**126, 2**127 + 1, [c[2] for c in RANGE_CHECK_CASES if c[0] != 50_000])
    rc("range_check_c1_n65", 0, 2**64, [0, 2**64 - 1, 2**64 + 2**59])
    rc("range_check_c2_n255", 0, 2**254, [5, 2**254 - 1, Q - 1])
    m = model.Composer()
    for mx, w, _ in MAX_BOUND_CASES:
        model.max_bound(m, mx, model.AllocatedScalar.allocate(m, w))
    dump(m, "max_bound_ref")
    m = model.Composer()
    for a, b, _ in MAYBE_EQUAL_CASES:
        model.maybe_equal(m, model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b))
    dump(m, "maybe_equal_ref")
    m = model.Composer()
    for v, y, s, a, b in [(7, 1234567, 1, 100, 100), (Q - 1, 42, 0, 20, 3330), (0, 5, 1, 0, 0), (2**200 + 1, Q - 5, 1, 9, Q - 9)]:
        vv, yv, sv = m.add_input(v), m.add_input(y), m.add_input(s)
        aa, bb = model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b)
        try:
            model.is_non_zero(m, vv, v)
        except model.NonExistingInverse:
            pass
        model.conditionally_select_one(m, yv, sv)
        model.maybe_equal(m, aa, bb)
    dump(m, "scalar_mix")
    m = model.Composer()
    full_circuit(ModelOps(m))
    dump(m, "composer_full")


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--from-model":
        write_from_model(sys.argv[2])
        print("wrote the model's dump under", sys.argv[2])
    elif len(sys.argv) == 2:
        sys.exit(1 if compare(sys.argv[1]) else 0)
    else:
        print(__doc__)
        sys.exit(2)
