#!/usr/bin/env python3
"""tools/placement_c3.py [log2_chunk] -- does the C3 step depend on how the call's arrays lie relative to each other?
Inputs (v, y, s, a, b) and outputs (five selector columns, three wire columns, the variable table, result Variables, prefix sums)
are carved out of two slabs: array k starts on a 2-MiB boundary plus k * skew bytes.  skew = 0 is what separate allocations give
(every big hipMalloc is 2-MiB aligned: all arrays congruent, the same row of every column on the same channel).  ms per step,
40 steps back to back, median of 5."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MiB2 = 2 << 20


def carve(slab, sizes, skew):
    """views of `slab` (uint8), one per size: start k on a 2-MiB boundary (relative to the slab's own alignment) + k * skew"""
    import torch
    base = slab.data_ptr()
    off = (-base) % MiB2
    out = []
    for k, nbytes in enumerate(sizes):
        start = off + k * skew
        out.append(slab[start:start + nbytes])
        off += ((nbytes + k * skew + MiB2 - 1) // MiB2 + 1) * MiB2
    return out


def main(log2_chunk=20, steps=40, rounds=5):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    chunk = 1 << log2_chunk
    eng = pg.Engine(0)
    host_in = [np.ascontiguousarray(x).view(np.uint8).reshape(-1) for x in bench.mix_inputs(chunk)]
    in_sizes = [x.size for x in host_in]
    G, V = 10 * chunk, 15 * chunk
    out_sizes = [G * 32] * 5 + [G * 8] * 3 + [V * 32, chunk * 16, (chunk + 1) * 8, (chunk + 1) * 8]
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    results = []
    for in_skew, out_skew in ((0, 0), (256, 0), (0, 256), (256, 256), (4096 + 256, 4096 + 256), (65536 + 4096 + 256, 65536 + 4096 + 256),
                              (512, 512), (1024, 1024), (2048, 2048), (8192, 8192), (128, 128), (0, 0)):
        slab_in = torch.empty((sum(in_sizes) + (len(in_sizes) + 2) * (MiB2 + 16 * in_skew + MiB2),), dtype=torch.uint8, device=dev)
        slab_out = torch.empty((sum(out_sizes) + (len(out_sizes) + 2) * (MiB2 + 16 * out_skew + MiB2),), dtype=torch.uint8, device=dev)
        ins = carve(slab_in, in_sizes, in_skew)
        for t, h in zip(ins, host_in):
            t.copy_(torch.from_numpy(h))
        outs = carve(slab_out, out_sizes, out_skew)
        cc = _lib.ColumnsC(*[t.data_ptr() for t in outs[:9]])
        res, roff, voff = outs[9], outs[10], outs[11]

        def loop():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(steps):
                assert lib.pg_scalar_mix_planned_batch(eng._h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                                       3, 5, 0, C.byref(cc), res.data_ptr(), sp) == 0
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps
        loop()
        ts = sorted(loop() for _ in range(rounds))
        results.append({"in_skew": in_skew, "out_skew": out_skew, "ms_per_step": round(ts[len(ts) // 2], 4), "min": round(ts[0], 4),
                        "in_mod_2MiB": [t.data_ptr() % MiB2 for t in ins][:3], "out_mod_2MiB": [t.data_ptr() % MiB2 for t in outs][:3]})
        print(json.dumps(results[-1]), flush=True)
        del slab_in, slab_out, ins, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
