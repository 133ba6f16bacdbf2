"""The reference's own test cases (inputs -> outcome), as data.

/root/reference/tests/range_gadgets_tests.rs:57-78   (max_bound, 4 cases)
/root/reference/tests/range_gadgets_tests.rs:120-169 (range_check, 8 cases)
/root/reference/tests/scalar_gadgets_tests.rs        (maybe_equal :36,:53; select_zero :85-119;
                                                      select_one :142-177; is_non_zero :199,:224,:235)
/root/reference/src/range.rs:205-233                 (decomposition of -100 into 8 bits -> 0)
"""
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001

# (max_range, witness, expected)
MAX_BOUND_CASES = [
    (2**128 - 1, 2**127, True),
    (200, 100, True),
    (100, 200, False),
    (2**128 - 1, 2**130, False),
]

# (min_range, max_range, witness, expected)
RANGE_CHECK_CASES = [
    (50_000, 250_000, 50_001, True),
    (50_000, 250_000, 250_001, False),
    (50_000, 250_000, 250_000, False),
    (50_000, 250_000, 249_000, True),
    (50_000, 250_000, 50_000, True),
    (50_000, 250_000, 49_999, False),
    (2**126, 2**127 + 1, 2**127 - 1, True),
    (50_000, 250_000, 18_598, False),
]

# The commit-key degree every one of the reference's test circuits is proven and verified under:
# `PublicParameters::setup(2 << k)` / `pub_params.trim(1 << k)` at tests/range_gadgets_tests.rs:49-50,111-112 and src/range.rs:208-209
# (k = 10), tests/scalar_gadgets_tests.rs:16-17 (maybe_equal: k = 9), :82-83, :139-140, :193-194 (select_zero / select_one /
# is_non_zero: k = 7).  Those tests pass in the reference's CI, so each circuit -- StandardComposer::new()'s rows, the gadget's,
# the closing constrain_to_constant -- padded to a power of two fits a key of that degree (dusk-plonk 0.8 preprocesses under
# trim(padded_circuit_size): its Circuit::compile does exactly that).  An UPPER bound on the row count, held by the reference
# itself: it is the one thing about rows the reference pins.  E.g. range_check over [2^126, 2^127 + 1) must stay within 1024 rows:
# 3 + (4n + 11) + 1 <= 1024 <=> n <= 252 -- a ladder length of 253 or more (a wrong num_bits_closest_power_of_two) would not.
TRIM_LOG2 = {"max_bound": 10, "range_check": 10, "scalar_decomposition": 10, "maybe_equal": 9, "select_zero": 7, "select_one": 7,
             "is_non_zero": 7}


def padded_circuit_size(circuit_size: int) -> int:
    """next power of two (dusk-plonk pads the circuit to one before it commits to the selector polynomials)"""
    return 1 << max(0, (circuit_size - 1).bit_length())


def fits_trim_degree(circuit_size: int, which: str) -> bool:
    return padded_circuit_size(circuit_size) <= (1 << TRIM_LOG2[which])


# (a, b, expected)
MAYBE_EQUAL_CASES = [(100, 100, True), (20, 3330, False), (0, 0, True)]


def full_circuit(ops):
    """One small circuit that touches every call of the path, spelled once and replayed on the big-int model (which
    freezes it as tests/golden/composer_full.npz), on the C oracle and on the device composer.  `ops` supplies
    add_input(int), allocate(int) -> AllocatedScalar, range_check_loop(min, max, [witness ints]) -> [Variable],
    max_bound(max, allocated) -> Variable, maybe_equal(a, b), is_non_zero(var, int), conditionally_select_one/zero,
    constrain_to_constant(var, int, pi int or None), boolean_gate(var).  Every row is satisfied."""
    x = ops.add_input(9)
    r = ops.range_check_loop(50_000, 250_000, [50_001, 250_000, 7])          # outcomes 1, 0, 0
    ops.is_non_zero(x, 9)
    y = ops.conditionally_select_one(x, r[0])                                  # selector 1 -> y = x = 9
    ops.constrain_to_constant(y, 20, 11)                                       # 9 - 20 + PI(11) = 0
    m = ops.maybe_equal(ops.allocate(5), ops.allocate(5))                      # 1
    z = ops.conditionally_select_zero(m, r[1])                                 # 1 * 0 = 0
    ops.constrain_to_constant(z, 0, None)
    b = ops.max_bound(2**64, ops.allocate(2**64 + 5))                          # out of range: 0
    ops.constrain_to_constant(b, 0, None)
    ops.boolean_gate(m)
    ops.constrain_to_constant(r[0], 1, None)
