"""Builds tests/cpp/gadgets_tests (host-only C++: g++, links the C-ABI library, the oracle and the HIP runtime)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(HERE, "gadgets_tests")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gadgets_tests.cpp")
    deps = [src, os.path.join(ROOT, "include", "plonk_gadgets.hpp"), os.path.join(ROOT, "include", "plonk_gadgets_hip.h")]
    if not force and os.path.exists(BIN) and all(os.path.getmtime(d) < os.path.getmtime(BIN) for d in deps):
        return BIN
    lib_dir, ora_dir = os.path.join(ROOT, "plonk_gadgets_amd"), os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", ora_dir, "-I", "/opt/rocm/include",
           src, "-o", BIN, "-L", lib_dir, "-lplonk_gadgets_hip", "-L", ora_dir, "-loracle", "-L", "/opt/rocm/lib",
           "-lamdhip64", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{ora_dir}", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath,$ORIGIN/../../plonk_gadgets_amd", "-Wl,-rpath,$ORIGIN/../../oracle"]
    subprocess.check_call(cmd)
    return BIN


if __name__ == "__main__":
    print(build(force=True))
