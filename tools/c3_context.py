#!/usr/bin/env python3
"""tools/c3_context.py -- why the C3 step is slower inside the driver's sequence (C2 -> C3 -> C4 in one process) than alone.
One process on the GPU box; ms per C3 step (20 timed steps after 5 warm-up steps, arrays allocated anew for every line):
alone; beside a 233-GB allocation nobody touches; right after that allocation is freed; after a pause; right after C2 has run
and released its arrays; after a pause."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import plonk_gadgets_amd as pg
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = pg.Engine(0)
    pause = float(os.environ.get("PAUSE_S", "10"))

    def run(name, steps=20, warmup=5):
        wl = bench.Workload(name, eng, dev, 0, 1, 20, -1)
        el, ms = bench.measure(wl, steps, warmup, torch.cuda.synchronize)
        wl.release()
        del wl
        torch.cuda.empty_cache()
        return round(el / steps * 1e3, 4)

    out = {"c3_alone": [run("c3"), run("c3")]}
    big = torch.empty((233 << 30,), dtype=torch.uint8, device=dev)
    out["c3_beside_233GB_untouched"] = run("c3")
    big.fill_(1)
    torch.cuda.synchronize()
    out["c3_beside_233GB_written_once"] = run("c3")
    del big
    torch.cuda.empty_cache()
    out["c3_right_after_freeing_it"] = run("c3")
    time.sleep(pause)
    out["c3_after_pause"] = run("c3")
    out["c2"] = run("c2")
    out["c3_right_after_c2"] = run("c3")
    out["c3_again"] = run("c3")
    time.sleep(pause)
    out["c3_after_pause_2"] = run("c3")
    out["c2_values"] = run("c2_values")
    out["c3_right_after_c2_values"] = run("c3")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
