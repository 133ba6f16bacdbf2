"""The C ABI from plain C (examples/range_check_batch.c): compiled with gcc on the CPU, run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "range_check_batch")


def build():
    from plonk_gadgets_amd import build as pg_build
    pg_build.build()
    lib = os.path.join(ROOT, "plonk_gadgets_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "range_check_batch.c"), "-L", lib,
                           "-lplonk_gadgets_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}",
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../plonk_gadgets_amd", "-o", BIN])
    return BIN


def test_c_example_compiles():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_c_example_runs_on_gpu():
    binary = BIN if os.path.exists(BIN) else build()
    p = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK")
    assert "first unsatisfied row: -1" in p.stdout


@pytest.mark.gpu
def test_python_example_runs_on_gpu():
    """examples/circuit_on_device.py: bulk decoding, a batched append, single calls, check, materialize, permutation"""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "circuit_on_device.py"), "10"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and "prover-ready" in p.stdout
