"""What other writers reach on this box: torch's fill kernel and hipMemsetAsync over 32 GiB, next to pg_fill_bytes.
(The emit kernel is compared with these in DESIGN.md section 3.)"""
import ctypes as C
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 1e3


def main():
    eng = pg.Engine(0)
    n = 1 << 32  # int64 elements = 32 GiB
    buf = torch.empty((n,), dtype=torch.int64, device="cuda:0")
    out = {"bytes": n * 8}
    out["torch_fill_gbps"] = n * 8 / timed(lambda: buf.fill_(7)) / 1e9
    out["torch_zero_gbps"] = n * 8 / timed(lambda: buf.zero_()) / 1e9
    for streams in (0, 1, 5, 8):
        out["pg_fill_%d_gbps" % streams] = n * 8 / timed(lambda: eng.fill_bytes(buf, streams)) / 1e9
    src = torch.empty((n // 2,), dtype=torch.int64, device="cuda:0")
    out["torch_copy_gbps_read_plus_write"] = 2 * (n // 2) * 8 / timed(lambda: buf[: n // 2].copy_(src)) / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
