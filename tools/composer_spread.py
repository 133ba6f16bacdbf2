#!/usr/bin/env python3
"""tools/composer_spread.py [trials] -- pg_composer_scalar_mix_batch of 2^20 items (10.5 M rows, 2.4 GB) appended to a device composer
whose columns are nine allocations / one block with the selector columns 24 GiB apart (pg_composer_spread_columns): ms per append
(the second and third append of every composer, HIP events around the call), `trials` composers of each kind in turn."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(trials=4):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    chunk = 1 << 20
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(chunk)]
    stream = torch.cuda.current_stream(dev)
    out = {"nine allocations": [], "one block, 24 GiB apart": []}
    for _ in range(trials):
        for name in out:
            comp = pg.StandardComposer(eng, 3 * 10 * chunk + 64, 3 * 15 * chunk + 64)
            if name.startswith("one"):
                comp.spread_columns(24)
            comp.scalar_mix_batch(*ins)
            ts = []
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(stream)
                comp.scalar_mix_batch(*ins)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(round(e0.elapsed_time(e1), 4))
            out[name].append(ts)
            comp.close() if hasattr(comp, "close") else None
            del comp
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
