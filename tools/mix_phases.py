#!/usr/bin/env python3
"""tools/mix_phases.py -- on the GPU box, with tools/variants/lib_mix_stamps.so (python tools/ab_emit.py build mix_stamps):
how long the waves of the fused mix's arithmetic launch spend in each phase (forward, look-back, inversion incl. waiting
for it, backward), averaged over the waves of a 2^20-item launch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from plonk_gadgets_amd import _lib
import plonk_gadgets_amd as pg
import bench

lib = C.CDLL(os.path.join(ROOT, "tools", "variants", "lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "mix_stamps")))
for fn, (r, a) in _lib.SIGNATURES.items():
    if hasattr(lib, fn):
        f = getattr(lib, fn)
        f.restype, f.argtypes = r, a
dev = torch.device("cuda", 0)
n = 1 << 20
ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(n)]
roff = torch.empty((n + 1,), dtype=torch.int64, device=dev)
voff = torch.empty((n + 1,), dtype=torch.int64, device=dev)
res = torch.empty((n, 2), dtype=torch.int64, device=dev)
cols = pg.Columns.allocate(10 * n, 15 * n, dev)
cc = cols.as_c()
h = C.c_void_p()
assert lib.pg_engine_create(0, C.byref(h)) == 0
out = (C.c_ulonglong * 12)()
ipl = int(os.environ.get("PG_EXP_MIX_IPL", "16"))
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.pg_scalar_mix_planned_batch(h, *[t.data_ptr() for t in ins], n, roff.data_ptr(), voff.data_ptr(), None, 3, 5, 0,
                                           C.byref(cc), res.data_ptr(), None) == 0
    e1.record()
    torch.cuda.synchronize()
    assert lib.pg_debug_mix_phases(out) == 0
    if rep:
        if os.environ.get("FWD_SPLIT"):  # tools/patches/r04_fwd_load_shape_diagnostic.patch with -DPG_DIAG_FWD_SPLIT
            w = 2048
            print("forward us per wave: first fetches + wait %.1f | steps 0-3 %.1f | 4-7 %.1f | 8-11 %.1f | 12-15 %.1f ; inversion %.1f backward %.1f" % (
                out[4] / w / 100, out[5] / w / 100, out[6] / w / 100, out[7] / w / 100, out[0] / w / 100, out[2] / w / 100, out[3] / w / 100))
        elif os.environ.get("MIX_DETAIL"):  # round 5's stamps: the wait at the first barrier, the inversion itself, the waiting waves' work
            waves = (n + 32 * ipl - 1) // (32 * ipl)
            print("us per wave: forward %.1f | inversion phase %.1f = wait at the first barrier %.1f + [lower waves: inversion %.1f | upper waves: look-back / early rows %.1f] + second barrier | backward %.1f   (the call %.1f us)"
                  % (out[0] / waves / 100, out[2] / waves / 100, out[4] / waves / 100, out[5] / (waves / 2) / 100, out[6] / (waves / 2) / 100,
                     out[3] / waves / 100, 1e3 * e0.elapsed_time(e1)))
            print("   upper waves: the prefix known after %.1f us; issuing the stores of tile 0 / 1 / 2 / 3+: %.1f / %.1f / %.1f / %.1f us"
                  % tuple(out[k] / (waves / 2) / 100 for k in (7, 8, 9, 10, 11)))
        elif any(out[k] for k in range(4, 8)):  # a staggered build (PG_EXP_STAGGER_TICKS): even / odd workgroups apart
            for g, name in ((0, "even"), (4, "odd ")):
                print("us per wave (%s workgroups): forward %.1f  look-back %.1f  inversion %.1f  backward %.1f" % ((name,) + tuple(out[g + k] / 1024 / 100.0 for k in range(4))))
        else:
            waves = (n + 32 * ipl - 1) // (32 * ipl)
            print("us per wave: forward %.1f  look-back %.1f  inversion %.1f  backward %.1f   (%d waves; the whole call %.1f us)"
                  % (tuple(out[k] / waves / 100.0 for k in range(4)) + (waves, 1e3 * e0.elapsed_time(e1))))
