"""tools/packed_vs_separate.py -- does it matter to the emitters whether the nine column arrays are nine allocations or
sections of one (pg_packed_layout)?  C3 (planned mix call) and C4 (ragged max_bound) steps, interleaved rounds."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import distributed as pd


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
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}


def main():
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    batch = 1 << 20
    # C3
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(batch)]
    _, roff, voff = eng.ragged_buffers(batch)
    res = torch.empty((batch, 2), dtype=torch.int64, device=dev)
    sep = pg.Columns.allocate(10 * batch, 15 * batch, dev)
    _, _, words = pd.packed_layout(10 * batch, 15 * batch)
    flat = torch.empty((words,), dtype=torch.int64, device=dev)
    pk = pd.columns_in(flat, 10 * batch, 15 * batch)
    t = timed({"separate": lambda: eng.scalar_mix_planned(*ins, roff, voff, sep, res, None, 3, 5, 0),
               "packed": lambda: eng.scalar_mix_planned(*ins, roff, voff, pk, res, None, 3, 5, 0)}, 40)
    print(json.dumps({"workload": "c3", **t}))
    del sep, flat, pk
    torch.cuda.empty_cache()
    # C4
    mr_np, wt_np = bench.c4_inputs(batch)
    mr = torch.from_numpy(mr_np.view(np.int64)).to(dev)
    wt = torch.from_numpy(wt_np.view(np.int64)).to(dev)
    nb, roff, voff = eng.ragged_buffers(batch)
    lay = eng.max_bound_ragged_plan(mr, nb, roff, voff)
    res = torch.empty((batch,), dtype=torch.int64, device=dev)
    sep = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
    _, _, words = pd.packed_layout(lay.n_gates, lay.n_vars)
    flat = torch.empty((words,), dtype=torch.int64, device=dev)
    pk = pd.columns_in(flat, lay.n_gates, lay.n_vars)
    t = timed({"separate": lambda: eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, sep, res, 3, 5),
               "packed": lambda: eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, pk, res, 3, 5)}, 6)
    print(json.dumps({"workload": "c4", **t}))


if __name__ == "__main__":
    main()
