"""The reference's test-suite re-expressed in C++ against the host mirror include/plonk_gadgets.hpp
(tests/cpp/gadgets_tests.cpp): compiled here on the CPU, run on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "cpp"))


def test_cpp_host_mirror_compiles():
    """CPU: the header-only C++ host layer and its test-suite build against the C ABI (no GPU needed to link)"""
    from oracle import pyoracle
    from plonk_gadgets_amd import build as pg_build
    pg_build.build()
    pyoracle.build()
    import build as cpp_build
    assert os.path.exists(cpp_build.build(force=True))


@pytest.mark.gpu
def test_cpp_reference_suite_on_gpu():
    import build as cpp_build
    binary = cpp_build.build()
    p = subprocess.run([binary], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for name in ("counting_scalar_bits", "scalar_decomposition_test", "max_bound_test", "range_check_test", "test_maybe_equal",
                 "test_conditionally_select_0", "test_conditionally_select_1", "test_is_not_zero", "single_calls_are_queued",
                 "batched_appends_equal_loops"):
        assert f"test {name} ... ok" in p.stdout
