"""CPU: what the compiler made of the kernels, read off the gfx950 code object inside the built library (no GPU needed):
no kernel uses scratch memory (a spill, or an aggregate the compiler could not keep in registers), and the kernels whose
residency the design rests on keep the register counts it assumes."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(tmp_path):
    """path of the gfx950 code object unbundled from the built library"""
    code_object_notes(tmp_path)
    return str(tmp_path / "dev.co")


def code_object_notes(tmp_path):
    from plonk_gadgets_amd import build as pg_build
    lib = pg_build.build()
    tools = [shutil.which("objcopy"), os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-readelf")]
    if not all(t and os.path.exists(t) for t in tools):
        pytest.skip("objcopy / clang-offload-bundler / llvm-readelf not found")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--output={co}"])
    return subprocess.run([tools[2], "--notes", co], capture_output=True, text=True, check=True).stdout


def kernels(notes):
    out = {}
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", block).group(1))
        out[name.group(1)] = {"vgpr": get("vgpr_count"), "scratch": get("private_segment_fixed_size"), "lds": get("group_segment_fixed_size")}
    return out


def test_no_kernel_uses_scratch_and_residency_assumptions_hold(tmp_path):
    ks = kernels(code_object_notes(tmp_path))
    assert len(ks) > 100, len(ks)
    spilled = {n: k["scratch"] for n, k in ks.items() if k["scratch"]}
    assert not spilled, spilled

    def one(substr):
        hits = [k for n, k in ks.items() if substr in n]
        assert hits, substr
        return hits
    for k in one("scalar_mix_vars_kernel"):      # two waves per SIMD, one workgroup per CU
        assert k["vgpr"] <= 256 and 100_000 < k["lds"] <= 160 * 1024
    for k in one("rows_periodic_kernel"):        # a store stream: eight waves per SIMD
        assert k["vgpr"] <= 64
    for k in one("perm_item_kernel"):            # seven workgroups per CU by LDS: 72 registers at most
        assert k["vgpr"] <= 72
    for k in one("gate_queue_kernel"):
        assert k["vgpr"] <= 128
    emit = [k for n, k in ks.items() if "emit_kernel" in n and "RangeCheckGDELi0E" in n]  # the full emission (EMIT_ALL): four waves per SIMD
    assert emit and all(k["vgpr"] <= 128 for k in emit)


def test_the_writing_waves_of_materialize_wait_for_no_memory_operation(tmp_path):
    """csrc/materialize.hpp: a wave's loads and stores share ONE counter (vmcnt), so a wait for a load -- or for a scratch reload, which is
    a vector memory operation too -- between two stores drains the wave's store stream; the compiler places such waits where control flow
    meets, taken or not.  In the closed-form instantiations (MAT_SELF, one per wire kind) the eight writing waves must execute none: in the
    disassembly, no `s_waitcnt vmcnt` between the kernel's first and last global store (the loader wave's code comes before them).
    Twice this round a build that computed the right values was 20 % slower for exactly this (NOTES_r05 section 8)."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    asm = subprocess.run([objdump, "-d", code_object(tmp_path)], capture_output=True, text=True, check=True).stdout
    bodies = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", asm)  # (every symbol: a body ends where the next one begins)
    seen = 0
    for name, body in zip(bodies[1::2], bodies[2::2]):
        if "materialize_items_kernelILi2E" not in name:
            continue
        seen += 1
        lines = body.split("\n")
        stores = [i for i, l in enumerate(lines) if "global_store" in l]
        assert len(stores) >= 11, (name, len(stores))
        waits = [lines[i].strip() for i in range(stores[0], stores[-1]) if re.search(r"s_waitcnt.*vmcnt", lines[i])]
        assert not waits, (name, waits[:3])
        assert not any("scratch_" in l for l in lines), name
    assert seen >= 7, seen  # six kinds + the per-item-bounds form
