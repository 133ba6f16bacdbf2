"""C3 (plan + scans + emit || inversion pre-pass) captured once into a HIP graph and replayed: what the launch path costs.
The engine's calls are capture-safe after one warm-up call (no allocation, no host synchronisation in the async forms;
the side stream joins the capture through its fork/join events)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import plonk_gadgets_amd as pg


def main():
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    batch = 1 << 20
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(batch)]
    roff = torch.empty((batch + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((batch + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((batch, 2), dtype=torch.int64, device=dev)
    cols = pg.Columns.allocate(10 * batch, 15 * batch, dev)

    def step():  # (the planned call; PG_C3_SEPARATE_PLAN=1: the plan as its own call)
        if os.environ.get("PG_C3_SEPARATE_PLAN") == "1":
            eng.scalar_mix_plan_async(ins[0], roff, voff)
            eng.scalar_mix_emit(*ins, roff, voff, cols, res, 3, 5, 0)
        else:
            eng.scalar_mix_planned(*ins, roff, voff, cols, res, None, 3, 5, 0)

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {"items": batch, "eager_ms": timed(step)}
    ref = {k: getattr(cols, k).clone() for k in ("q_c", "w_o", "var_values")}
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for k in ref:
        getattr(cols, k).zero_()
    out["graph_ms"] = timed(g.replay)
    torch.cuda.synchronize()
    out["graph_output_equals_eager"] = all(torch.equal(getattr(cols, k), ref[k]) for k in ref)
    out["constraints_per_s_graph"] = 10 * batch / out["graph_ms"] * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
