"""Builds libplonk_gadgets_hip.so in-tree with hipcc for gfx950 (no torch dependency in the library)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplonk_gadgets_hip.so")
SOURCES = ["capi.hip"]
HEADERS = ["experiment.hpp", "fr.hpp", "emit.hpp", "invert.hpp", "range_gadgets.hpp", "scalar_gadgets.hpp", "composer.hpp", "permutation.hpp", "materialize.hpp",
           "capi_composer.inc", "capi_dist.inc"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "plonk_gadgets_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def kernel_sources_sha256() -> str:
    """one hash over everything the library is compiled from (csrc/ and the C header), in a fixed order: what
    profiles/pmc_summary.json is stamped with and bench.py compares against"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "plonk_gadgets_hip.h"), "rb").read())
    return h.hexdigest()


def command(out: str, csrc: str = CSRC, extra_flags: list[str] | None = None) -> list[str]:
    """the compiler command line of the library: NO -DPG_... option (csrc/experiment.hpp: those are tools/ab_emit.py's A/B
    builds, which pass them -- and -DPG_EXPERIMENT -- as extra_flags)"""
    cc = hipcc()
    # ROCm's include directory (only for <rccl/rccl.h>, which capi_dist.inc can do without): beside the compiler, or $ROCM_PATH
    rocm = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(cc)))
    inc = ["-I" + os.path.join(rocm, "include")] if os.path.isdir(os.path.join(rocm, "include")) else []
    return [cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + inc + [
            "-o", out] + [os.path.join(csrc, s) for s in SOURCES] + ["-ldl"] + (extra_flags or [])


def build(force: bool = False, extra_flags: list[str] | None = None, out: str | None = None, csrc: str = CSRC) -> str:
    out = out or LIB
    if not force and out == LIB and not is_stale():
        return out
    subprocess.check_call(command(out, csrc, extra_flags))
    return out


if __name__ == "__main__":
    print(build(force=True))
