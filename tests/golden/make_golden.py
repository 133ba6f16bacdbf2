#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the big-int model (oracle/model.py).

The reference is Rust and cannot run in this environment (no cargo/rustc, crates not vendored), and its own tests
hold no row/limb vectors, so these fixtures are NOT outputs of the reference: they are the model's restatement of
it, frozen so that the C oracle, the HIP path and future rounds are all compared with the same bytes.  Inputs are
the reference's own test cases (tests/refcases.py cites them) plus the BASELINE config shapes.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import model  # noqa: E402
from tests.refcases import MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES  # noqa: E402

Q = model.Q


def pack(m: model.Composer, extra: dict) -> dict:
    e = model.export(m, 3, 5)
    out = {k: np.array(v, dtype=np.uint64).reshape(-1, 4) for k, v in e.items() if k.startswith("q_") or k == "var_values"}
    out.update({k: np.array(e[k], dtype=np.uint64) for k in ("w_l", "w_r", "w_o")})
    out.update({k: np.array(v, dtype=np.uint64) for k, v in extra.items()})
    return out


def limbs(xs):
    return np.array([model.mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


class ModelOps:
    """tests/refcases.py:full_circuit on the big-int model"""

    def __init__(self, m):
        self.m = m

    def add_input(self, v): return self.m.add_input(v)
    def allocate(self, v): return model.AllocatedScalar.allocate(self.m, v)
    def range_check_loop(self, mn, mx, ws): return [model.range_check(self.m, mn, mx, self.allocate(w)) for w in ws]
    def max_bound(self, mx, a): return model.max_bound(self.m, mx, a)[0]
    def maybe_equal(self, a, b): return model.maybe_equal(self.m, a, b)
    def is_non_zero(self, var, value): model.is_non_zero(self.m, var, value)
    def conditionally_select_one(self, y, s): return model.conditionally_select_one(self.m, y, s)
    def conditionally_select_zero(self, x, s): return model.conditionally_select_zero(self.m, x, s)
    def constrain_to_constant(self, a, c, pi): self.m.constrain_to_constant(a, c, pi)
    def boolean_gate(self, a): self.m.boolean_gate(a)


def main():
    # 1. the reference's 8 range_check cases, one batch (all share min/max except case 7 -> its own file)
    for name, cases in (("range_check_ref_50k_250k", [c for c in RANGE_CHECK_CASES if c[0] == 50_000]),
                        ("range_check_ref_2p126_2p127", [c for c in RANGE_CHECK_CASES if c[0] != 50_000])):
        m = model.Composer()
        mn, mx = cases[0][0], cases[0][1]
        res = [model.range_check(m, mn, mx, model.AllocatedScalar.allocate(m, c[2])) for c in cases]
        assert m.check() == -1 and [m.variables[r] for r in res] == [int(c[3]) for c in cases]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(m, {
            "min_range": limbs([mn]), "max_range": limbs([mx]), "witness": limbs([c[2] for c in cases]),
            "result_vars": res, "expected": [int(c[3]) for c in cases]}))
    # 2. the reference's 4 max_bound cases (per-item bounds: a ragged batch)
    m = model.Composer()
    res, nbits = [], []
    for mx, w, exp in MAX_BOUND_CASES:
        r, n = model.max_bound(m, mx, model.AllocatedScalar.allocate(m, w))
        assert m.variables[r] == int(exp)
        res.append(r)
        nbits.append(n)
    assert m.check() == -1
    np.savez_compressed(os.path.join(HERE, "max_bound_ref.npz"), **pack(m, {
        "max_range": limbs([c[0] for c in MAX_BOUND_CASES]), "witness": limbs([c[1] for c in MAX_BOUND_CASES]),
        "result_vars": res, "num_bits": nbits, "expected": [int(c[2]) for c in MAX_BOUND_CASES]}))
    # 3. BASELINE shapes, 3 witnesses each: C1 (n = 65) and C2 (n = 255)
    for name, mn, mx, ws in (("range_check_c1_n65", 0, 2**64, [0, 2**64 - 1, 2**64 + 2**59]),
                             ("range_check_c2_n255", 0, 2**254, [5, 2**254 - 1, Q - 1])):
        m = model.Composer()
        res = [model.range_check(m, mn, mx, model.AllocatedScalar.allocate(m, w)) for w in ws]
        assert m.check() == -1
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(m, {
            "min_range": limbs([mn]), "max_range": limbs([mx]), "witness": limbs(ws), "result_vars": res,
            "expected": [m.variables[r] for r in res]}))
    # 4. scalar mix (config C3 shape): v, y, s, a, b per item; item 2 has v = 0 (NonExistingInverse path)
    items = [(7, 1234567, 1, 100, 100), (Q - 1, 42, 0, 20, 3330), (0, 5, 1, 0, 0), (2**200 + 1, Q - 5, 1, 9, Q - 9)]
    m = model.Composer()
    res, err = [], []
    for v, y, s, a, b in items:
        vv, yv, sv = m.add_input(v), m.add_input(y), m.add_input(s)
        aa, bb = model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b)
        try:
            model.is_non_zero(m, vv, v)
            err.append(0)
        except model.NonExistingInverse:
            err.append(1)
        res.append([model.conditionally_select_one(m, yv, sv), model.maybe_equal(m, aa, bb)])
    assert m.check() == -1
    np.savez_compressed(os.path.join(HERE, "scalar_mix.npz"), **pack(m, {
        **{k: limbs([it[i] for it in items]) for i, k in enumerate(("v", "y", "s", "a", "b"))},
        "result_vars": res, "err_mask": err}))
    # 5. the three maybe_equal reference cases as one batch
    m = model.Composer()
    res = [model.maybe_equal(m, model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b))
           for a, b, _ in MAYBE_EQUAL_CASES]
    assert [m.variables[r] for r in res] == [int(c[2]) for c in MAYBE_EQUAL_CASES]
    np.savez_compressed(os.path.join(HERE, "maybe_equal_ref.npz"), **pack(m, {
        "a": limbs([c[0] for c in MAYBE_EQUAL_CASES]), "b": limbs([c[1] for c in MAYBE_EQUAL_CASES]),
        "result_vars": res, "expected": [int(c[2]) for c in MAYBE_EQUAL_CASES]}))
    # 6. one whole composer: every gadget once, a public input, from row 0 (the initial rows with their live fourth wire
    #    included), with what the prover consumes next: constant columns, fourth wire, dense PI, sigma.
    #    The circuit is spelled out in tests/refcases.py:full_circuit so that the C oracle and the device composer can
    #    replay it call for call.
    from tests.refcases import full_circuit  # noqa: E402
    m = model.Composer()
    full_circuit(ModelOps(m))
    assert m.check() == -1
    padded = 1 << (m.n - 1).bit_length()
    e = model.export(m, 0, 0)
    out = {k: np.array(v, dtype=np.uint64).reshape(-1, 4) for k, v in e.items() if k.startswith("q_") or k == "var_values"}
    out.update({k: np.array(e[k], dtype=np.uint64) for k in ("w_l", "w_r", "w_o")})
    out["w_4"] = np.array(m.w_4, dtype=np.uint64)
    out["q_4"] = limbs(m.q_4)
    out["q_arith"] = limbs(m.q_arith)
    out["dense_pi"] = limbs(m.dense_pi())
    out["sigma"] = np.array(m.sigma(padded), dtype=np.uint64)
    out["padded_n"] = np.array([padded], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "composer_full.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
