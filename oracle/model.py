"""oracle/model.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, big-int form).

A second, independent restatement of the reference's gadget path in plain
Python integers mod q (no limbs, no Montgomery arithmetic inside: a scalar is
its canonical integer; Montgomery limbs are produced only when exporting).
It exists so that three statements of the path can be compared limb for limb
-- this model, oracle/*.c, and the HIP kernels -- and to generate the small
golden fixtures under tests/golden/ (tests/golden/make_golden.py).

Follows, function for function:
  /root/reference/src/allocated_scalar.rs:17-30
  /root/reference/src/range.rs:21-189
  /root/reference/src/scalar.rs:21-140
and the dusk-plonk 0.8 `StandardComposer` slice described in SURVEY.md section
3.4 (source of that crate is not under /root/reference: PARITY UNPINNED for
gate rows -- the reference's tests pin only outcomes and satisfiability).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % Q
MASK64 = (1 << 64) - 1


def mont_limbs(x: int) -> list[int]:
    """canonical integer -> the 4 little-endian u64 limbs of x*R mod q (BlsScalar's inner [u64;4])."""
    m = (x % Q) * R % Q
    return [(m >> (64 * i)) & MASK64 for i in range(4)]


def from_mont_limbs(limbs) -> int:
    m = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return m * pow(R, -1, Q) % Q


class Error(Exception):
    """errors.rs:13-18"""


class NonExistingInverse(Error):
    pass


class Composer:
    """StandardComposer slice."""

    def __init__(self, dummy: bool = True):
        self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.q_4, self.q_arith = [], [], [], [], [], [], []
        self.w_l, self.w_r, self.w_o, self.w_4 = [], [], [], []
        self.pi = {}
        self.variables = []  # Variable(k) -> value
        self.perm = []  # Variable(k) -> [(wire, gate)]
        self.zero_var = 0
        self.zero_var = self.add_witness_to_circuit_description(0)
        if dummy:
            six, one, seven, m20 = (self.add_input(v) for v in (6, 1, 7, Q - 20))
            self._row(six, seven, m20, one, 1, 2, 3, 4, 4, 1, None)
            self._row(m20, six, seven, self.zero_var, 1, 1, 1, 1, 127, 0, None)

    @property
    def n(self):
        return len(self.q_m)

    def add_input(self, s: int) -> int:
        self.variables.append(s % Q)
        self.perm.append([])
        return len(self.variables) - 1

    def _row(self, a, b, c, d, q_m, q_l, q_r, q_o, q_c, q_4, pi):
        g = self.n
        for col, v in ((self.q_m, q_m), (self.q_l, q_l), (self.q_r, q_r), (self.q_o, q_o), (self.q_c, q_c),
                       (self.q_4, q_4), (self.q_arith, 1)):
            col.append(v % Q)
        for w, (col, v) in enumerate(((self.w_l, a), (self.w_r, b), (self.w_o, c), (self.w_4, d))):
            col.append(v)
            self.perm[v].append((w, g))
        if pi is not None:
            self.pi[g] = pi % Q

    def poly_gate(self, a, b, c, q_m, q_l, q_r, q_o, q_c, pi=None):
        self._row(a, b, c, self.zero_var, q_m, q_l, q_r, q_o, q_c, 0, pi)

    def constrain_to_constant(self, a, constant, pi=None):
        self.poly_gate(a, a, a, 0, 1, 0, 0, -constant, pi)

    def add_witness_to_circuit_description(self, v):
        var = self.add_input(v)
        self.constrain_to_constant(var, v)
        return var

    def assert_equal(self, a, b):
        self.poly_gate(a, b, self.zero_var, 0, 1, -1, 0, 0)

    def add(self, q_l_a, q_r_b, q_c, pi=None):
        (q_l, a), (q_r, b) = q_l_a, q_r_b
        c_eval = (q_l * self.variables[a] + q_r * self.variables[b] + q_c + (pi or 0)) % Q
        c = self.add_input(c_eval)
        self._row(a, b, c, self.zero_var, 0, q_l, q_r, -1, q_c, 0, pi)
        return c

    def mul(self, q_m, a, b, q_c, pi=None):
        c_eval = (q_m * self.variables[a] * self.variables[b] + q_c + (pi or 0)) % Q
        c = self.add_input(c_eval)
        self._row(a, b, c, self.zero_var, q_m, 0, 0, -1, q_c, 0, pi)
        return c

    def mul_gate(self, a, b, c, q_m, q_o, q_c, pi=None):
        self._row(a, b, c, self.zero_var, q_m, 0, 0, q_o, q_c, 0, pi)

    def boolean_gate(self, a):
        self._row(a, a, a, self.zero_var, 1, 0, 0, -1, 0, 0, None)
        return a

    def sigma(self, padded_n: int) -> list:
        """Permutation::compute_sigma_permutations [dusk-plonk 0.8, restated]: the positions recorded under one Variable,
        in recording order, form one cycle; position (wire, gate) is encoded wire * padded_n + gate; rows past the
        circuit keep the identity.  4 * padded_n entries."""
        out = list(range(4 * padded_n))
        for lst in self.perm:
            for i, (w, g) in enumerate(lst):
                w2, g2 = lst[(i + 1) % len(lst)]
                out[w * padded_n + g] = w2 * padded_n + g2
        return out

    def dense_pi(self) -> list:
        """construct_dense_pi_vec: one scalar per row, zero where no public input was given"""
        return [self.pi.get(i, 0) for i in range(self.n)]

    def check(self) -> int:
        """index of the first unsatisfied row, or -1"""
        v = self.variables
        for i in range(self.n):
            a, b, c, d = v[self.w_l[i]], v[self.w_r[i]], v[self.w_o[i]], v[self.w_4[i]]
            t = (self.q_m[i] * a * b + self.q_l[i] * a + self.q_r[i] * b + self.q_o[i] * c + self.q_4[i] * d
                 + self.q_c[i] + self.pi.get(i, 0))
            if t % Q:
                return i
        return -1


class AllocatedScalar:
    """allocated_scalar.rs:17-30"""

    def __init__(self, var: int, scalar: int):
        self.var, self.scalar = var, scalar % Q

    @staticmethod
    def allocate(composer: Composer, scalar: int) -> "AllocatedScalar":
        return AllocatedScalar(composer.add_input(scalar), scalar)


def scalar_to_bits(s: int):
    """range.rs:161-170"""
    return [(s >> i) & 1 for i in range(256)]


def bits_count(s: int) -> int:
    """range.rs:173-181"""
    s %= Q
    counter = 1
    while s > 1:
        s >>= 1
        counter += 1
    return counter


def num_bits_closest_power_of_two(s: int) -> int:
    """range.rs:185-189"""
    return bits_count(pow(2, bits_count(s), Q))


def maybe_equal(composer: Composer, a: AllocatedScalar, b: AllocatedScalar) -> int:
    """scalar.rs:105-140"""
    u = composer.add((1, a.var), (-1, b.var), 0)
    u_scalar = (a.scalar - b.scalar) % Q
    z = composer.add_input(pow(u_scalar, -1, Q) if u_scalar else 0)
    y = composer.mul(-1, z, u, 1)
    composer.mul_gate(y, u, u, 1, 0, 0)
    return y


def scalar_decomposition_gadget(composer: Composer, num_bits: int, witness: AllocatedScalar):
    """range.rs:119-158"""
    bits = scalar_to_bits(witness.scalar)
    bit_vars = [composer.add_input(b) for b in bits]
    if num_bits > 256:
        raise IndexError("range.rs:134 slice [..num_bits] out of range")
    bit_vars = bit_vars[:num_bits]
    acc = AllocatedScalar(composer.add_witness_to_circuit_description(0), 0)
    for power, bit in enumerate(bit_vars):
        composer.boolean_gate(bit)
        two_pow = pow(2, power, Q)
        acc.var = composer.add((two_pow, bit), (1, acc.var), 0)
        acc.scalar = (acc.scalar + two_pow * bits[power]) % Q
    return maybe_equal(composer, acc, witness), bit_vars


def range_proof(composer, value, num_bits):
    """range.rs:21-24"""
    return scalar_decomposition_gadget(composer, num_bits, value)[0]


def max_bound(composer: Composer, max_range: int, witness: AllocatedScalar):
    """range.rs:82-113"""
    max_range = (max_range - 1) % Q
    n = num_bits_closest_power_of_two(max_range)
    var = composer.add((-1, witness.var), (0, witness.var), max_range)
    b = AllocatedScalar(var, max_range - witness.scalar)
    return range_proof(composer, b, n), n


def min_bound(composer: Composer, min_range: int, witness: AllocatedScalar, num_bits: int):
    """range.rs:53-76"""
    var = composer.add((1, witness.var), (0, witness.var), -min_range)
    return range_proof(composer, AllocatedScalar(var, witness.scalar - min_range), num_bits)


def range_check(composer: Composer, min_range: int, max_range: int, witness: AllocatedScalar) -> int:
    """range.rs:27-43"""
    y1, n = max_bound(composer, max_range, witness)
    y2 = min_bound(composer, min_range, witness, n)
    return composer.mul(1, y1, y2, 0)


def conditionally_select_zero(composer, x, select):
    """scalar.rs:21-27"""
    return composer.mul(1, x, select, 0)


def conditionally_select_one(composer, y, selector):
    """scalar.rs:36-59"""
    one = composer.add_witness_to_circuit_description(1)
    sy = composer.mul(1, y, selector, 0)
    oms = composer.add((1, one), (-1, selector), 0)
    return composer.add((1, sy), (1, oms), 0)


def is_non_zero(composer, var, value_assigned):
    """scalar.rs:63-97 -- raises NonExistingInverse after the first var+row were pushed"""
    var_assigned = composer.add_input(value_assigned)
    composer.assert_equal(var, var_assigned)
    if value_assigned % Q == 0:
        raise NonExistingInverse()
    inv = composer.add_input(pow(value_assigned, -1, Q))
    one = composer.add_witness_to_circuit_description(1)
    composer.poly_gate(var, inv, one, 1, 0, 0, -1, 0)


def export(composer: Composer, gate_base: int = 0, var_base: int = 0) -> dict:
    """8 live columns + variable table, scalars as Montgomery limbs (lists of 4 ints)."""
    sl = slice(gate_base, None)
    return {
        "q_m": [mont_limbs(x) for x in composer.q_m[sl]],
        "q_l": [mont_limbs(x) for x in composer.q_l[sl]],
        "q_r": [mont_limbs(x) for x in composer.q_r[sl]],
        "q_o": [mont_limbs(x) for x in composer.q_o[sl]],
        "q_c": [mont_limbs(x) for x in composer.q_c[sl]],
        "w_l": list(composer.w_l[sl]),
        "w_r": list(composer.w_r[sl]),
        "w_o": list(composer.w_o[sl]),
        "var_values": [mont_limbs(x) for x in composer.variables[var_base:]],
    }
