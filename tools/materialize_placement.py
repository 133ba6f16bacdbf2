#!/usr/bin/env python3
"""tools/materialize_placement.py -- pg_composer_materialize (270 M rows: ten columns of 8.65 GB and w_4 written in lock step)
with its output columns as eleven allocations in a row / in one block at strides of 10, 12, 14 and 16 GiB: ms per call."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GiB = 1 << 30


def main(log2_batch=18):
    import numpy as np
    import torch
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib, synth
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    batch = 1 << log2_batch
    comp = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = torch.from_numpy(synth.random_scalars(batch, seed=synth.SEED + 2).view(np.int64)).to(dev)
    comp.range_check_batch(pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254), wit)
    n = comp.circuit_size()
    names = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add", "w_4_value", "w_l_value",
             "w_r_value", "w_o_value")

    def timed(fc, keep):
        assert lib.pg_composer_materialize(comp._h, C.byref(fc)) == 0
        ms = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            assert lib.pg_composer_materialize(comp._h, C.byref(fc)) == 0
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
        ms.sort()
        return round(ms[2], 3)

    out = {}
    for trial in range(2):
        t = {k: torch.empty((n, 4), dtype=torch.int64, device=dev) for k in names}
        t["w_4"] = torch.empty((n,), dtype=torch.int64, device=dev)
        out.setdefault("eleven allocations", []).append(timed(_lib.FullColumnsC(**{k: v.data_ptr() for k, v in t.items()}), t))
        del t
        torch.cuda.empty_cache()
        for stride_gib in (0, 10, 12, 14, 16):
            col = (n * 32 + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            stride = max(stride_gib * GiB, col)
            slab = torch.empty((10 * stride + n * 8 + (4 << 20),), dtype=torch.uint8, device=dev)
            base = slab.data_ptr() + (-slab.data_ptr()) % (2 << 20)
            ptrs = {k: base + i * stride for i, k in enumerate(names)}
            ptrs["w_4"] = base + 9 * stride + col
            out.setdefault("one block, stride %s GiB" % (stride_gib or "8.06 (packed)"), []).append(timed(_lib.FullColumnsC(**ptrs), slab))
            del slab
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
