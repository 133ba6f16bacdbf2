#!/usr/bin/env python3
"""tools/event_cost.py [log2_chunk] [steps] -- what an event record between the calls of a loop costs on the stream (GPU box).
The C3 step (pg_scalar_mix_planned_batch) issued `steps` times: back to back, with a default timing event after every call
(torch.cuda.Event(enable_timing=True)), and with a timing event created with hipEventDisableSystemFence (no cache write-back
and invalidation when it completes -- hip_runtime_api.h recommends it for events that only measure time).  ms per step."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(log2_chunk=20, steps=40, rounds=5):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    hip = bench.hip_runtime()
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
    eng = pg.Engine(0)

    def call():
        assert lib.pg_scalar_mix_planned_batch(eng._h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                               3, 5, 0, C.byref(cc), res.data_ptr(), sp) == 0

    def loop(mode):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = []
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(steps):
            call()
            if mode == "default_events":
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                keep.append(e)
            elif mode == "no_system_fence_events":
                keep.append(bench.TimingEvent(hip).record(sp))
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    out = {}
    for mode in ("back_to_back", "default_events", "no_system_fence_events"):
        loop(mode)
        ts = sorted(loop(mode) for _ in range(rounds))
        out[mode] = {"ms_per_step_median": ts[len(ts) // 2], "min": ts[0]}
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
