#!/usr/bin/env python3
"""tools/fill_lottery.py [instances] [GiB] -- bare fills on a buffer allocated anew `instances` times: torch's fill and
pg_fill_bytes as one window (0), one stream per workgroup (1), and 5 / 8 parts advanced together (the emitters' shape).  GB/s of
each per instance: which shapes depend on where the allocation landed?"""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(instances=10, gib=2):
    import torch
    import plonk_gadgets_amd as pg
    eng = pg.Engine(0)
    n = (gib << 30) // 8

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(n * 8 / (e0.elapsed_time(e1) / reps) / 1e6)

    for inst in range(instances):
        buf = torch.empty((n,), dtype=torch.int64, device="cuda:0")
        out = {"instance": inst, "torch_fill": timed(lambda: buf.fill_(7))}
        for streams in (0, 1, 5, 8):
            out["pg_fill_%d" % streams] = timed(lambda: eng.fill_bytes(buf, streams))
        out["pg_fill_5_again"] = timed(lambda: eng.fill_bytes(buf, 5))
        print(json.dumps(out), flush=True)
        del buf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
