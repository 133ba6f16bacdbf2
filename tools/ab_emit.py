#!/usr/bin/env python3
"""A/B of emit-kernel build variants on one GPU, interleaved rounds in one process (guide rule 24).

    python tools/ab_emit.py build     # here (no GPU): compile the variants into tools/variants/
    python tools/ab_emit.py run       # on the GPU box: time them on the C2 shape
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "variants")

# the build options that exist in the tree today (every one of them was measured; DESIGN.md section 3 has the numbers)
VARIANTS = {
    "r2": None,  # round 2's library, built by hand from `git archive 74bdd0d` (not rebuilt by `build`)
    "base": [],
    "prev": None,  # the commit before, built by hand (git stash; build(out=...); git stash pop)
    # round 5: row tiles per workgroup that the fused mix's arithmetic launch writes while it inverts (the library: 2)
    "fb": ["-DPG_EXP_FWD_BURST"], "fb_stamps": ["-DPG_EXP_FWD_BURST", "-DPG_MIX_STAMPS"],
    "et2": ["-DPG_MIX_EARLY_TILES=2"],
    "sm": ["-DPG_MIX_EARLY_TILES=0", "-DPG_EXP_STEP_MAJOR"],  # (two-step form only) step m of all waves = one contiguous window
    "et0": ["-DPG_MIX_EARLY_TILES=0"], "et1": ["-DPG_MIX_EARLY_TILES=1"], "et3": ["-DPG_MIX_EARLY_TILES=3"], "et4": ["-DPG_MIX_EARLY_TILES=4"],
    # the fused mix (C3)
    "mix_stamps": ["-DPG_MIX_STAMPS"],  # timing build for tools/mix_phases.py
    "mix_stamps_nomem": ["-DPG_MIX_STAMPS", "-DPG_MIX_ABLATE_MEM"],
    "mix_stamps_nofwdld": ["-DPG_MIX_STAMPS", "-DPG_MIX_ABLATE_FWD_LOADS"],
    "mix_stamps_nofwdst": ["-DPG_MIX_STAMPS", "-DPG_MIX_ABLATE_FWD_STORES"],
    "mix_nomem": ["-DPG_MIX_ABLATE_MEM"],  # timing only (wrong output): the arithmetic launch without its global loads and stores
    "ablate_amul": ["-DPG_ABLATE_AMUL"],   # timing only (wrong output): the big-item emitters without the per-accumulator multiplication
    "rows_wps4": ["-DPG_ROWS_WAVES_PER_SIMD=4"],  # the periodic rows launch allowed 128 registers
    "rows_prio3": ["-DPG_ROWS_SETPRIO=3"],      # the rows launch's waves at the highest issue priority
    "rows_prio1": ["-DPG_ROWS_SETPRIO=1"],
    # round 4 experiments on the mix
    "sbit": ["-DPG_EXP_SBIT"],
    "diag_coal": ["-DPG_MIX_STAMPS", "-DPG_DIAG_FWD_COALESCED"],  # timing only: the forward pass's loads as three contiguous KiB
    "diag_3ld": ["-DPG_MIX_STAMPS", "-DPG_DIAG_FWD_3LOADS"],      # timing only: without the v side's second fetch of v
    "diag_split": ["-DPG_MIX_STAMPS", "-DPG_DIAG_FWD_SPLIT"],     # stamps inside the forward loop (they disturb it: vmcnt(0) waits)
    "spread64": ["-DPG_TILE_SPREAD=64"], "spread1024": ["-DPG_TILE_SPREAD=1024"], "spread4096": ["-DPG_TILE_SPREAD=4096"],  # tile order: consecutive workgroups far apart
    # one tile per workgroup, started in address order by the dispatcher (no grid-stride loop): tools/probes/single_table_fill.hip
    "ga": ["-DPG_GRID_BLOCKS_PER_CU=1000000"], "ga_w16": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_RC_W=16"],
    "ga_w8": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_RC_W=8"], "ga_w4": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_RC_W=4"],
    # pg_composer_materialize (tools/mat_variants.py): timing builds without its reads (wrong output) and residency / shape knobs
    # perm_ladder_kernel: rows per workgroup piece, dynamic LDS that bounds the residency (tools/mat_variants.py times the whole call)
    "lr512": ["-DPG_PERM_LADDER_ROWS=512"], "lr512_l2": ["-DPG_PERM_LADDER_ROWS=512", "-DPG_PERM_LADDER_LDS=80896"],
    "lr512_l4": ["-DPG_PERM_LADDER_ROWS=512", "-DPG_PERM_LADDER_LDS=39936"], "lr1024_l2": ["-DPG_PERM_LADDER_ROWS=1024", "-DPG_PERM_LADDER_LDS=80896"],
    "lr2048_l2": ["-DPG_PERM_LADDER_ROWS=2048", "-DPG_PERM_LADDER_LDS=80896"], "lr1024_l4": ["-DPG_PERM_LADDER_ROWS=1024", "-DPG_PERM_LADDER_LDS=39936"],
    "lr8192_l2": ["-DPG_PERM_LADDER_LDS=80896"], "lr512_l3": ["-DPG_PERM_LADDER_ROWS=512", "-DPG_PERM_LADDER_LDS=53248"],
    "lr512_l5": ["-DPG_PERM_LADDER_ROWS=512", "-DPG_PERM_LADDER_LDS=31744"], "lr512_l6": ["-DPG_PERM_LADDER_ROWS=512", "-DPG_PERM_LADDER_LDS=26624"],
    "lr1024_l3": ["-DPG_PERM_LADDER_ROWS=1024", "-DPG_PERM_LADDER_LDS=53248"], "lr8192_l4": ["-DPG_PERM_LADDER_LDS=39936"],
    "ga_mbw8": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_MB_W=8"], "ga_mbw32": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_MB_W=32"],
    "ga_mbw4": ["-DPG_GRID_BLOCKS_PER_CU=1000000", "-DPG_MB_W=4"], "g256": ["-DPG_GRID_BLOCKS_PER_CU=256"], "g128": ["-DPG_GRID_BLOCKS_PER_CU=128"],
    "mat_g1m": ["-DPG_MAT_GRID_PER_CU=1000000"], "mat_g8": ["-DPG_MAT_GRID_PER_CU=8"], "mat_g2": ["-DPG_MAT_GRID_PER_CU=2"], "mat_g256": ["-DPG_MAT_GRID_PER_CU=256"],
    "mat_l16": ["-DPG_MAT_LOADS=16"], "mat_l4": ["-DPG_MAT_LOADS=4"], "mat_l16_u1": ["-DPG_MAT_LOADS=16", "-DPG_MAT_UNROLL=1"],
    "mat_u1": ["-DPG_MAT_UNROLL=1"], "mat_u4": ["-DPG_MAT_UNROLL=4"], "mat_t256b": ["-DPG_MAT_THREADS=256"], "mat_t1024": ["-DPG_MAT_THREADS=960"],
    "mat_noreads": ["-DPG_MAT_ABLATE_READS"], "mat_nowin": ["-DPG_MAT_ABLATE_WINDOW"], "mat_elsewhere": ["-DPG_MAT_READ_ELSEWHERE"],
    "rc_w4": ["-DPG_RC_W=4"], "rc_w8": ["-DPG_RC_W=8"], "rc_w64": ["-DPG_RC_W=64"],  # items per tile of range_check (rc_w16 below)
    "var_single": ["-DPG_VAR_SWEEP_SINGLE"],        # the variable sweep one scalar per lane everywhere (before round 4)
    "var_pairs_always": ["-DPG_VAR_SWEEP_PAIRS_ALWAYS"],  # the paired sweep in the full emission too
    "side_normal": ["-DPG_SIDE_STREAM_NORMAL_PRIORITY"],
    "inv_grp2": ["-DPG_INV_GRP=2"],
    "p1": ["-DPG_EXP_PREFETCH_INV"],
    "g8": ["-DPG_MIX_ROWS_GRID_PER_CU=8"],
    "p1g8": ["-DPG_EXP_PREFETCH_INV", "-DPG_MIX_ROWS_GRID_PER_CU=8"],
    "f1": ["-DPG_EXP_PARK_BLOCKS"],
    "f1_stamps": ["-DPG_EXP_PARK_BLOCKS", "-DPG_MIX_STAMPS"],
    "f2a_stamps": ["-DPG_EXP_WAVE_STAGGER=2", "-DPG_MIX_STAMPS"],
    "f2b_stamps": ["-DPG_EXP_WAVE_STAGGER=5", "-DPG_MIX_STAMPS"],
    "f3": ["-DPG_EXP_PARK_BLOCKS", "-DPG_EXP_WAVE_STAGGER=2"],
    "f3_stamps": ["-DPG_EXP_PARK_BLOCKS", "-DPG_EXP_WAVE_STAGGER=2", "-DPG_MIX_STAMPS"],
    "iplenv": ["-DPG_EXP_MIX_IPL_ENV"],
    "iplenv_stamps": ["-DPG_EXP_MIX_IPL_ENV", "-DPG_MIX_STAMPS"],
    "rows128": ["-DPG_MIX_ROWS_W=128"],
    "rows512": ["-DPG_MIX_ROWS_W=512"],
    "rowsgrid8": ["-DPG_MIX_ROWS_GRID_PER_CU=8"],
    "rowsgrid12": ["-DPG_MIX_ROWS_GRID_PER_CU=12"],
    "gen_side": ["-DPG_EXP_GENERIC_SIDE"],
    "stag50": ["-DPG_MIX_STAMPS", "-DPG_EXP_STAGGER_TICKS=5000"],
    "stag100": ["-DPG_MIX_STAMPS", "-DPG_EXP_STAGGER_TICKS=10000"],
    # every emitter
    "invert_fermat": ["-DPG_INVERT_FERMAT"],
    "full_barriers": ["-DPG_FULL_BARRIERS"],
    "unaligned_sweeps": ["-DPG_UNALIGNED_SWEEPS"],  # generic sweeps start at the tile's first unit wherever it falls in a 128-byte line
    "xcd_remap": ["-DPG_XCD_REMAP"],
    "nt_stores": ["-DPG_NT_STORES"],
    "grid8": ["-DPG_GRID_BLOCKS_PER_CU=8"],
    "inv_grp8": ["-DPG_INV_GRP=8"],
    "inv_lanes512": ["-DPG_INV_LANES_PER_CU=512"],
    "rc_w16": ["-DPG_RC_W=16"],
    "mb_w32": ["-DPG_MB_W=32"],
    "mb_w8": ["-DPG_MB_W=8"],
}


PATCHES = {  # builds that are NOT in the sources: a patch (tools/patches/) applied to a copy of csrc/ first
    "mix_stamps_nomem": "ablations_wrong_output.patch", "mix_stamps_nofwdld": "ablations_wrong_output.patch",
    "mix_stamps_nofwdst": "ablations_wrong_output.patch", "mix_nomem": "ablations_wrong_output.patch",
    "ablate_amul": "ablations_wrong_output.patch",
    # round 4's rejected schedules of the fused mix (profiles/NOTES_r04.md): apply to the sources of commit "round 4: values" --
    # kept as a record of what was measured, it may no longer apply cleanly
    "diag_coal": "r04_fwd_load_shape_diagnostic.patch", "diag_3ld": "r04_fwd_load_shape_diagnostic.patch",
    "diag_split": "r04_fwd_load_shape_diagnostic.patch",
    "f1": "r04_mix_experiments.patch", "f3": "r04_mix_experiments.patch", "p1": "r04_mix_experiments.patch", "g8": "r04_mix_experiments.patch",
    "p1g8": "r04_mix_experiments.patch", "iplenv": "r04_mix_experiments.patch", "iplenv_stamps": "r04_mix_experiments.patch",
    "f1_stamps": "r04_mix_experiments.patch", "f2a_stamps": "r04_mix_experiments.patch", "f2b_stamps": "r04_mix_experiments.patch",
    "f3_stamps": "r04_mix_experiments.patch", "stag50": "r04_mix_experiments.patch", "stag100": "r04_mix_experiments.patch",
    "rows128": "r04_mix_experiments.patch", "rows512": "r04_mix_experiments.patch", "rowsgrid8": "r04_mix_experiments.patch",
    "rowsgrid12": "r04_mix_experiments.patch", "gen_side": "r04_mix_experiments.patch",
    "mat_noreads": "r05_materialize_ablations.patch", "mat_nowin": "r05_materialize_ablations.patch", "mat_elsewhere": "r05_materialize_ablations.patch",
    "fb": "r05_fwd_burst.patch", "fb_stamps": "r05_fwd_burst.patch", "sm": "r05_step_major.patch",  # round 5: four steps' elements per fetch in the forward pass
}


def build(only=None):
    import shutil
    import subprocess
    import tempfile
    from plonk_gadgets_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, flags in VARIANTS.items():
        if flags is None or (only and name not in only):
            continue
        out = os.path.join(VDIR, f"lib_{name}.so")
        flags = (["-DPG_EXPERIMENT"] + flags) if flags else []  # csrc/experiment.hpp: no option without it
        if name in PATCHES:
            with tempfile.TemporaryDirectory() as tmp:
                shutil.copytree(os.path.join(ROOT, "plonk_gadgets_amd"), os.path.join(tmp, "plonk_gadgets_amd"),
                                ignore=shutil.ignore_patterns("*.so", "__pycache__"))
                shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
                subprocess.check_call(["patch", "-p0", "-s", "-d", tmp, "-i", os.path.join(ROOT, "tools", "patches", PATCHES[name])])
                b.build(force=True, extra_flags=flags, out=out, csrc=os.path.join(tmp, "plonk_gadgets_amd", "csrc"))
        else:
            b.build(force=True, extra_flags=flags, out=out)
        print("built", out)


def run_c3(log2_chunk=20, rounds=4):
    """fused scalar mix (BASELINE config C3): the planned call (plan + rows + variables), what bench.py times"""
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib
    import plonk_gadgets_amd as pg
    import bench
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(chunk)]
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
    cols = pg.Columns.allocate(10 * chunk, 15 * chunk, dev)
    cc = cols.as_c()
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, a) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):  # an older library lacks the newer entry points
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        lay, nerr = _lib.LayoutC(), C.c_uint64()
        assert lib.pg_scalar_mix_plan(h, ins[0].data_ptr(), chunk, roff.data_ptr(), voff.data_ptr(), None, C.byref(lay),
                                      C.byref(nerr), sp) == 0
        libs[name] = (lib, h)
    nbytes = chunk * 2320
    times = {n: [] for n in libs}
    for r in range(rounds + 1):
        order = list(libs.items())
        order = order[r % len(order):] + order[:r % len(order)]  # rotate: whoever runs first in a round pays for it (~3 %)
        for name, (lib, h) in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            st = lib.pg_scalar_mix_planned_batch(h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                                 3, 5, 0, C.byref(cc), res.data_ptr(), sp)
            assert st == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "median_ms": ts[len(ts) // 2], "min_ms": ts[0],
                          "gbps_median": nbytes / ts[len(ts) // 2] / 1e6}))


def run_c3_b2b(log2_chunk=20, steps=40, rounds=5):
    """the C3 step issued back to back, no synchronisation and no events between the calls (what a caller's loop does): ms per
    step over `steps` calls -- the difference to run_c3's per-call figure is what the step's launches cost each other"""
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib
    import plonk_gadgets_amd as pg
    import bench
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(chunk)]
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
    cols = pg.Columns.allocate(10 * chunk, 15 * chunk, dev)
    cc = cols.as_c()
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, a) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        libs[name] = (lib, h)
    times = {n: [] for n in libs}
    for r in range(rounds + 1):
        for name, (lib, h) in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(steps):
                if os.environ.get("C3_FORM") == "twostep":  # the plan as a call of its own (a plan kernel, no look-back in the arithmetic launch)
                    assert lib.pg_scalar_mix_plan_async(h, ins[0].data_ptr(), chunk, roff.data_ptr(), voff.data_ptr(), None, sp) == 0
                    assert lib.pg_scalar_mix_batch(h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), 3, 5, 0,
                                                   C.byref(cc), res.data_ptr(), sp) == 0
                    continue
                assert lib.pg_scalar_mix_planned_batch(h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                                       3, 5, 0, C.byref(cc), res.data_ptr(), sp) == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1) / steps)
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"variant": name, "ms_per_step_back_to_back": ts[len(ts) // 2], "min": ts[0]}))


def run_c4(log2_chunk=19, rounds=4):
    """ragged max_bound (BASELINE config C4 shape): plan once, time invert pre-pass + emit"""
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib
    import plonk_gadgets_amd as pg
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    mr_np, wt_np = bench.c4_inputs(chunk)
    mr = torch.from_numpy(mr_np.view(np.int64)).to(dev)
    wt = torch.from_numpy(wt_np.view(np.int64)).to(dev)
    nb = torch.empty((chunk,), dtype=torch.int32, device=dev)
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    libs = {}
    cols = None
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, a) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):  # an older library lacks the newer entry points
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        lay = _lib.LayoutC()
        assert lib.pg_max_bound_ragged_plan(h, mr.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(),
                                            C.byref(lay), C.c_void_p(stream.cuda_stream)) == 0
        if cols is None:
            cols = pg.Columns.allocate(int(lay.n_gates), int(lay.n_vars), dev)
            nbytes = int(lay.n_gates) * 184 + int(lay.n_vars) * 32
        libs[name] = (lib, h)
    cc = cols.as_c()
    times = {n: [] for n in libs}
    for r in range(rounds + 1):
        order = list(libs.items())
        order = order[r % len(order):] + order[:r % len(order)]  # rotate: whoever runs first in a round pays for it (~3 %)
        for name, (lib, h) in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            st = lib.pg_max_bound_ragged_batch(h, mr.data_ptr(), wt.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(),
                                               voff.data_ptr(), 3, 5, C.byref(cc), res.data_ptr(),
                                               C.c_void_p(stream.cuda_stream))
            assert st == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "median_ms": ts[len(ts) // 2], "min_ms": ts[0],
                          "gbps_median": nbytes / ts[len(ts) // 2] / 1e6}))


def run(log2_chunk=18, rounds=4, mn_int=0, mx_int=2**254):
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib, synth
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    wit = torch.from_numpy(synth.random_scalars(chunk).view(np.int64)).to(dev)
    mn, mx = pg.BlsScalar.from_int(mn_int), pg.BlsScalar.from_int(mx_int)
    lay0 = pg.Engine(0).range_check_layout(mn, mx, 1)
    G, V = lay0.gates_per_item, lay0.vars_per_item
    cols = pg.Columns.allocate(chunk * G, chunk * V, dev)
    res = torch.empty((chunk,), dtype=torch.int64, device=dev)
    cc = cols.as_c()
    stream = torch.cuda.current_stream(dev)
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, a) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):  # an older library lacks the newer entry points
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        libs[name] = (lib, h)
    times = {n: [] for n in libs}
    nbytes = chunk * (G * 184 + V * 32)
    for r in range(rounds + 1):
        order = list(libs.items())
        order = order[r % len(order):] + order[:r % len(order)]  # rotate: whoever runs first in a round pays for it (~3 %)
        for name, (lib, h) in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            st = lib.pg_range_check_batch(h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), chunk, 3, 5, C.byref(cc),
                                          res.data_ptr(), C.c_void_p(stream.cuda_stream))
            assert st == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "median_ms": ts[len(ts) // 2], "min_ms": ts[0],
                          "gbps_median": nbytes / ts[len(ts) // 2] / 1e6, "gbps_best": nbytes / ts[0] / 1e6}))


def run_values(log2_chunk=20, rounds=6, tables=1):
    """the witness refresh of C2's circuit (pg_range_check_values_batch: 1034 variables per witness, no rows); tables > 1: that many
    tables alive together, every variant timed on each (where a table lies decides 5.6 ... 6.8 ms: NOTES_r04 section 7)"""
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib, synth
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    wit = torch.from_numpy(synth.random_scalars(chunk).view(np.int64)).to(dev)
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    T = [torch.empty((chunk * 1034, 4), dtype=torch.int64, device=dev) for _ in range(tables)]
    stream = torch.cuda.current_stream(dev)
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, a) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        libs[name] = (lib, h)
    times = {n: [[] for _ in T] for n in libs}
    nbytes = chunk * 1034 * 32
    for r in range(rounds + 1):
        order = list(libs.items())
        order = order[r % len(order):] + order[:r % len(order)]
        for name, (lib, h) in order:
            for k, table in enumerate(T):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                st = lib.pg_range_check_values_batch(h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), chunk, table.data_ptr(),
                                                     C.c_void_p(stream.cuda_stream))
                assert st == 0
                e1.record(stream)
                torch.cuda.synchronize()
                if r:
                    times[name][k].append(e0.elapsed_time(e1))
    for name, per in times.items():
        med = [sorted(ts)[len(ts) // 2] for ts in per]
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "median_ms": med[0] if tables == 1 else [round(m, 3) for m in med],
                          "min_ms": min(per[0]), "gbps_median": nbytes / med[0] / 1e6}))


def run_maybe_equal(log2_chunk=22, rounds=6):
    """pg_maybe_equal_batch: 3 rows + 3 variables per item (648 B), one inverse per item"""
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib, synth
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    a = torch.from_numpy(synth.random_scalars(chunk, seed=1).view(np.int64)).to(dev)
    b = torch.from_numpy(synth.random_scalars(chunk, seed=2).view(np.int64)).to(dev)
    av = torch.arange(chunk, dtype=torch.int64, device=dev)
    bv = av + chunk
    cols = pg.Columns.allocate(chunk * 3, chunk * 3, dev)
    res = torch.empty((chunk,), dtype=torch.int64, device=dev)
    cc = cols.as_c()
    stream = torch.cuda.current_stream(dev)
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, f"lib_{name}.so")
        if not os.path.exists(path):
            continue
        lib = C.CDLL(path)
        for fn, (r, ar) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):
                f = getattr(lib, fn)
                f.restype, f.argtypes = r, ar
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        libs[name] = (lib, h)
    times = {n: [] for n in libs}
    nbytes = chunk * 3 * (184 + 32)
    for r in range(rounds + 1):
        order = list(libs.items())
        order = order[r % len(order):] + order[:r % len(order)]
        for name, (lib, h) in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            st = lib.pg_maybe_equal_batch(h, av.data_ptr(), a.data_ptr(), bv.data_ptr(), b.data_ptr(), chunk, 3, 5, C.byref(cc),
                                          res.data_ptr(), C.c_void_p(stream.cuda_stream))
            assert st == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"variant": name, "flags": VARIANTS[name], "median_ms": ts[len(ts) // 2], "min_ms": ts[0],
                          "gbps_median": nbytes / ts[len(ts) // 2] / 1e6}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "run_c3":
        run_c3(*(int(x) for x in sys.argv[2:]))
    elif sys.argv[1] == "run_c4":
        run_c4(*(int(x) for x in sys.argv[2:]))
    elif sys.argv[1] == "run_c3_b2b":
        run_c3_b2b(*(int(x) for x in sys.argv[2:]))
    elif sys.argv[1] == "run_maybe_equal":
        run_maybe_equal(*(int(x) for x in sys.argv[2:]))
    elif sys.argv[1] == "run_values":
        run_values(*(int(x) for x in sys.argv[2:]))
    else:
        run(*(int(x) for x in sys.argv[2:]))
