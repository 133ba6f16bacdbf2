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

# (a, b, expected)
MAYBE_EQUAL_CASES = [(100, 100, True), (20, 3330, False), (0, 0, True)]
