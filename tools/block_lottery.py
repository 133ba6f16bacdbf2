#!/usr/bin/env python3
"""tools/block_lottery.py [blocks] [MiB] -- is write bandwidth a property of WHERE an allocation lies?  `blocks` separate
allocations of `MiB` each, alive together; every block filled alone (torch fill_, 20 times: median GB/s), twice over -- does the
ranking repeat? -- then pairs of blocks filled by one kernel launch each on two streams at once."""
import json
import statistics
import sys


def main(blocks=24, mib=320):
    import torch
    dev = torch.device("cuda", 0)
    n = mib << 20
    bl = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(blocks)]
    st = torch.cuda.current_stream(dev)

    def bw(t, reps=20):
        t.fill_(1)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            t.fill_(2)
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return n / statistics.median(ts) / 1e6
    first = [round(bw(t)) for t in bl]
    second = [round(bw(t)) for t in bl]
    print(json.dumps({"MiB": mib, "first_pass_GBps": first, "second_pass_GBps": second,
                      "ptr_mod_1GiB_MiB": [(t.data_ptr() % (1 << 30)) >> 20 for t in bl]}))
    # a big one: the whole set as one allocation
    del bl
    torch.cuda.empty_cache()
    big = torch.empty((n * blocks,), dtype=torch.uint8, device=dev)
    parts = [round(bw(big[i * n:(i + 1) * n])) for i in range(blocks)]
    print(json.dumps({"one_allocation_parts_GBps": parts, "whole_GBps": round(bw(big, 5) * blocks)}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
