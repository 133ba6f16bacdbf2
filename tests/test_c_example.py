"""The C ABI from plain C (examples/range_check_batch.c): compiled with gcc on the CPU, run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "range_check_batch")


C5_BIN = os.path.join(ROOT, "examples", "c5_rank")


def build():
    from plonk_gadgets_amd import build as pg_build
    pg_build.build()
    lib = os.path.join(ROOT, "plonk_gadgets_amd")
    for src, out in (("range_check_batch.c", BIN), ("c5_rank.c", C5_BIN)):
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-D_DEFAULT_SOURCE", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                               "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", src), "-L", lib,
                               "-lplonk_gadgets_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}",
                               "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../plonk_gadgets_amd", "-o", out])
    return BIN


def test_c_example_compiles():
    assert os.path.exists(build()) and os.path.exists(C5_BIN)


def c5_witness_ints(rank, total, max_bits):
    """examples/c5_rank.c's witnesses: limb k of witness g = splitmix64(4 (g + 1) + k), cut at max_bits + 1 bits"""
    m64 = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9e3779b97f4a7c15) & m64
        x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & m64
        x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & m64
        return x ^ (x >> 31)
    out = []
    for i in range(total):
        g = rank * total + i
        v = sum(sm(4 * (g + 1) + k) << (64 * k) for k in range(4))
        out.append(v & ((1 << (max_bits + 1)) - 1))
    return out


def c5_oracle_digest(world, total, chunk, max_bits):
    """what every rank of examples/c5_rank.c must print: FNV-1a over the nine arrays of every rank's chunk, chunk by chunk,
    as the CPU oracle emits them at the global numbering"""
    import numpy as np
    from oracle import pyoracle as po
    from plonk_gadgets_amd import synth
    mn, mx = synth.mont(0), synth.mont(1 << max_bits)
    h, words = 0xcbf29ce484222325, 0
    # (each rank's items on a fresh composer: rows from gate 3, Variables from 5)
    per_rank = [po.range_check_batch(mn, mx, synth.scalars_from_ints(c5_witness_ints(r, total, max_bits))) for r in range(world)]
    G, V = per_rank[0]["n_gates"] // total, per_rank[0]["n_vars"] // total
    for k in range(total // chunk):
        for r in range(world):
            o = per_rank[r]
            g0, v0 = k * chunk * G, k * chunk * V
            for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
                arrs = o[name][g0:g0 + chunk * G].reshape(-1)
                for w in arrs.tolist():
                    h = ((h ^ w) * 0x100000001b3) & ((1 << 64) - 1)
                words += arrs.size
            for name in ("w_l", "w_r", "w_o"):
                # the oracle numbered rank r's items from its own item 0: shift its Variables to the global numbering
                # (Variable 0 = zero_var is not a Variable of the batch and stays)
                a = o[name][g0:g0 + chunk * G].astype(np.uint64)
                a = np.where(a == 0, a, a + np.uint64(r * total * V))
                for w in a.tolist():
                    h = ((h ^ w) * 0x100000001b3) & ((1 << 64) - 1)
                words += a.size
            arrs = o["var_values"][v0:v0 + chunk * V].reshape(-1)
            for w in arrs.tolist():
                h = ((h ^ w) * 0x100000001b3) & ((1 << 64) - 1)
            words += arrs.size
    return h, words


@pytest.mark.gpu
@pytest.mark.parametrize("variables_only", [0, 1])
def test_c5_rank_from_plain_c(tmp_path, variables_only):
    """BASELINE config 5's pipeline driven by a compiled host with no Python in it (examples/c5_rank.c), as one rank of a
    world of one: the communicator, the double-buffered emit-while-gather loop and the consumer callback all run inside /
    from the library; the digest of everything the rank received == the CPU oracle's"""
    binary = C5_BIN if os.path.exists(C5_BIN) else (build() and C5_BIN)
    total, chunk, bits = 192, 64, 18
    p = subprocess.run([binary, "0", "1", str(tmp_path / "comm.id"), str(total), str(chunk), str(variables_only), str(bits)],
                       capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stderr
    want, words = c5_oracle_digest(1, total, chunk, bits)
    line = p.stdout.strip().splitlines()[-1]
    assert f"{total // chunk} chunks, {words} words" in line
    assert line.endswith("digest %016x" % want), (line, "%016x" % want)


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
