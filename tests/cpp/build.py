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
    build_fake_rccl(force)
    return BIN


FAKE_RCCL = os.path.join(HERE, "libfake_rccl.so")


def build_fake_rccl(force: bool = False) -> str:
    """tests/cpp/fake_rccl.c: the test-only collective the world-2-on-one-GPU tests load through PG_RCCL_LIB"""
    src = os.path.join(HERE, "fake_rccl.c")
    if not force and os.path.exists(FAKE_RCCL) and os.path.getmtime(src) < os.path.getmtime(FAKE_RCCL):
        return FAKE_RCCL
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-shared", "-fPIC", "-D_DEFAULT_SOURCE", "-D__HIP_PLATFORM_AMD__",
                           "-I", "/opt/rocm/include", src, "-L", "/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", FAKE_RCCL])
    return FAKE_RCCL


if __name__ == "__main__":
    print(build_fake_rccl(force=True))
    print(build(force=True))
