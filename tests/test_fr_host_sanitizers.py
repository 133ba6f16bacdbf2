"""CPU: the field arithmetic of csrc/fr.hpp -- the very functions the kernels run, in their host build -- under
UBSan + ASan (GPU sanitizers are not available on the pool; the host build shares every line but fr_mul's inline asm)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_field_arithmetic_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fr_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=undefined,address", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "plonk_gadgets_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "fr_sanitize.cpp"),
                           "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "bad = 0" in p.stdout, p.stdout + p.stderr


def test_oracle_under_sanitizers(tmp_path):
    """the checker itself: oracle/*.c with a driver that runs every batch driver, the threaded fast form and a full
    composer (sigma, dense PI) under ASan + UBSan"""
    exe = str(tmp_path / "oracle_sanitize")
    ora = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", ora,
                           os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c")] +
                          [os.path.join(ora, f) for f in ("fr.c", "composer.c", "gadgets.c", "fast.c")] + ["-lpthread", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "oracle under sanitizers: ok" in p.stdout, p.stdout + p.stderr
