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

VARIANTS = {
    "base": [],
    "w32_grid32": ["-DPG_RC_W=32", "-DPG_GRID_BLOCKS_PER_CU=32"],
    "w32_grid64": ["-DPG_RC_W=32", "-DPG_GRID_BLOCKS_PER_CU=64"],
    "w32_grid1024": ["-DPG_RC_W=32", "-DPG_GRID_BLOCKS_PER_CU=1024"],
    "w16_grid1024": ["-DPG_RC_W=16", "-DPG_GRID_BLOCKS_PER_CU=1024"],
    "w24_grid1024": ["-DPG_RC_W=24", "-DPG_GRID_BLOCKS_PER_CU=1024"],
    "w48_grid1024": ["-DPG_RC_W=48", "-DPG_GRID_BLOCKS_PER_CU=1024"],
}


def build():
    from plonk_gadgets_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, flags in VARIANTS.items():
        out = os.path.join(VDIR, f"lib_{name}.so")
        b.build(force=True, extra_flags=flags, out=out)
        print("built", out)


def run(log2_chunk=18, rounds=4):
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib, synth
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    wit = torch.from_numpy(synth.random_scalars(chunk).view(np.int64)).to(dev)
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    G, V = 1031, 1034
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
            f = getattr(lib, fn)
            f.restype, f.argtypes = r, a
        h = C.c_void_p()
        assert lib.pg_engine_create(0, C.byref(h)) == 0
        libs[name] = (lib, h)
    times = {n: [] for n in libs}
    nbytes = chunk * (G * 184 + V * 32)
    for r in range(rounds + 1):
        for name, (lib, h) in libs.items():
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


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(*(int(x) for x in sys.argv[2:]))
