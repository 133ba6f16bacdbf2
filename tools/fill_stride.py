"""tools/fill_stride.py -- the placement effect in its simplest form: the bare fill kernel writes `streams` equal parts of
one buffer in step (4 KiB of every part per workgroup pass); the distance between the parts is varied by a few bytes to a
few MiB.  If the memory system's channel / bank selection makes some distances collide, the rate shows it."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg


def main():
    eng = pg.Engine(0)
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    part0 = (int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 2 << 30) // 4096 * 4096
    deltas = [0, 32, 96, 256, 1024, 4096, 16384, 32768, 65536, 262144, 1 << 20, 4096 + 32, 65536 + 96]
    buf = torch.empty((streams * (part0 + (4 << 20)) // 8,), dtype=torch.int64, device="cuda:0")
    out = {}
    for rnd in range(3):
        for d in deltas:
            n = streams * (part0 + d) // 8
            mis = int(os.environ.get("PG_FILL_MISALIGN", "0")) // 8  # start the whole thing this many bytes into the buffer
            view = buf[mis:mis + n]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.fill_bytes(view, streams)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                out.setdefault(d, []).append(n * 8 / e0.elapsed_time(e1) / 1e6)
    print(json.dumps({"streams": streams, "part_GiB": part0 / (1 << 30), "GBps_by_extra_distance": {str(d): round(max(v)) for d, v in out.items()}}))


if __name__ == "__main__":
    main()
