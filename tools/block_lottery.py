#!/usr/bin/env python3
"""tools/block_lottery.py [blocks] [GiB] -- write bandwidth against WHERE in the card's memory a block lies: `blocks` allocations of
`GiB` each, alive together (26 x 10 GiB cover most of the card), every block filled alone: torch's fill_ (one moving window) and
pg_fill_bytes with 5 and 8 parts advanced in lock step (the emitters' shape), GB/s each, two passes."""
import json
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(blocks=26, gib=10):
    import torch
    import plonk_gadgets_amd as pg
    eng = pg.Engine(0)
    dev = torch.device("cuda", 0)
    n = (gib << 30) // 8
    bl = [torch.empty((n,), dtype=torch.int64, device=dev) for _ in range(blocks)]
    st = torch.cuda.current_stream(dev)

    def bw(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn()
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return round(n * 8 / statistics.median(ts) / 1e6)
    for p in range(2):
        print(json.dumps({"pass": p, "GiB": gib,
                          "torch_fill": [bw(lambda: t.fill_(3)) for t in bl],
                          "pg_fill_5": [bw(lambda: eng.fill_bytes(t, 5)) for t in bl],
                          "pg_fill_8": [bw(lambda: eng.fill_bytes(t, 8)) for t in bl]}), flush=True)
    print(json.dumps({"ptrs": [hex(t.data_ptr()) for t in bl]}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
