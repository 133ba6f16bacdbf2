"""tools/column_skew.py -- does the RELATIVE placement of the nine column arrays matter to the emitters?  The five
selector columns have the same size, and a workgroup writes the same row range of all of them at the same time: if their
base addresses differ by a multiple of the memory system's interleave period, the five streams walk the channels in step.
One allocation, the columns back to back with `skew` extra bytes before column c (c * skew in total); C4 and C3 steps."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import plonk_gadgets_amd as pg
from plonk_gadgets_amd.engine import Columns


ALIGN = 1 << int(os.environ.get("PG_SKEW_ALIGN_LOG2", "22"))  # "delta" placement: every column starts at a multiple of this plus c * delta


def skewed(n_gates, n_vars, skew_bytes, dev, controlled=False):
    sizes = [n_gates * 4] * 5 + [n_gates] * 3 + [n_vars * 4]  # int64 words
    sw = skew_bytes // 8
    total = sum(sizes) + 9 * sw + 16 + (10 * ALIGN // 8 if controlled else 0)
    flat = torch.empty((total,), dtype=torch.int64, device=dev)
    base = flat.data_ptr()
    views, at = [], 0
    for c, n in enumerate(sizes):
        if controlled:  # absolute address = multiple of ALIGN + c * skew
            addr = base + at * 8
            addr = (addr + ALIGN - 1) // ALIGN * ALIGN + c * skew_bytes
            at = (addr - base) // 8
        else:
            at += sw
        at = (at + 1) & ~1  # 16-byte aligned
        views.append(flat[at:at + n])
        at += n
    return flat, Columns(*[v.view(n_gates, 4) for v in views[:5]], *views[5:8], views[8].view(n_vars, 4))


def timed(fn, rounds):
    out = {k: [] for k in fn}
    for r in range(rounds + 1):
        order = list(fn.items())
        order = order[r % len(order):] + order[:r % len(order)]
        for k, f in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            if r:
                out[k].append(e0.elapsed_time(e1))
    return {k: round(sorted(v)[len(v) // 2], 4) for k, v in out.items()}


def main():
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    skews = [0, 4096, 4096 + 256, 65536 + 4096, (1 << 20) + 8192 + 512, 3 * (1 << 20) + 20480]
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    if len(sys.argv) > 2:
        skews = [int(x) for x in sys.argv[2].split(",")]
    controlled = len(sys.argv) > 3 and sys.argv[3] == "delta"
    batch = 1 << (20 if which not in ("c4", "c2") else 18)
    if len(sys.argv) > 4:
        batch = 1 << int(sys.argv[4])
    if which == "c2":
        from plonk_gadgets_amd import synth
        wit = torch.from_numpy(synth.random_scalars(batch).view(np.int64)).to(dev)
        mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
        lay = eng.range_check_layout(mn, mx, batch)
        res = torch.empty((batch,), dtype=torch.int64, device=dev)
        keep, fns = [], {}
        sep = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
        fns["separate allocations"] = lambda: eng.range_check_batch(mn, mx, wit, 3, 5, sep, res)
        for s in skews:
            flat, cols = skewed(lay.n_gates, lay.n_vars, s, dev, controlled)
            keep.append(flat)
            fns[f"skew {s}"] = (lambda c: (lambda: eng.range_check_batch(mn, mx, wit, 3, 5, c, res)))(cols)
        print(json.dumps({"workload": f"c2, {batch} items", **timed(fns, 6)}))
        return
    if which == "c4":
        mr_np, wt_np = bench.c4_inputs(batch)
        mr = torch.from_numpy(mr_np.view(np.int64)).to(dev)
        wt = torch.from_numpy(wt_np.view(np.int64)).to(dev)
        nb, roff, voff = eng.ragged_buffers(batch)
        lay = eng.max_bound_ragged_plan(mr, nb, roff, voff)
        res = torch.empty((batch,), dtype=torch.int64, device=dev)
        keep, fns = [], {}
        sep = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
        fns["separate allocations"] = lambda: eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, sep, res, 3, 5)
        probes = {}
        m = 1 << 14  # does a short emit over the first items of the same buffers rank the placements like the full one?
        probes["separate allocations"] = lambda: eng.max_bound_ragged_emit(mr[:m], wt[:m], nb[:m], roff[:m + 1], voff[:m + 1], sep, res[:m], 3, 5)
        for s in skews:
            flat, cols = skewed(lay.n_gates, lay.n_vars, s, dev, controlled)
            keep.append(flat)
            fns[f"skew {s}"] = (lambda c: (lambda: eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, c, res, 3, 5)))(cols)
            probes[f"skew {s}"] = (lambda c: (lambda: eng.max_bound_ragged_emit(mr[:m], wt[:m], nb[:m], roff[:m + 1], voff[:m + 1], c, res[:m], 3, 5)))(cols)
        print(json.dumps({"workload": f"c4, {batch} items", **timed(fns, 6)}))
        print(json.dumps({"workload": f"c4 probe, first {m} items of the same buffers", **timed(probes, 20)}))
    else:
        ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(batch)]
        _, roff, voff = eng.ragged_buffers(batch)
        res = torch.empty((batch, 2), dtype=torch.int64, device=dev)
        keep, fns = [], {}
        sep = pg.Columns.allocate(10 * batch, 15 * batch, dev)
        fns["separate allocations"] = lambda: eng.scalar_mix_planned(*ins, roff, voff, sep, res, None, 3, 5, 0)
        for s in skews:
            flat, cols = skewed(10 * batch, 15 * batch, s, dev, controlled)
            keep.append(flat)
            fns[f"skew {s}"] = (lambda c: (lambda: eng.scalar_mix_planned(*ins, roff, voff, c, res, None, 3, 5, 0)))(cols)
        print(json.dumps({"workload": "c3, 2^20 items", **timed(fns, 30)}))


if __name__ == "__main__":
    main()
