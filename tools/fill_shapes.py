"""tools/fill_shapes.py -- on the GPU box: what a bare store stream reaches on the arrays the workloads write, beside the
emitters themselves.  For C2's, C4's and C3's columns (as bench.py allocates them): pg_fill_columns with several tile sizes, the
workload's own launch, torch's fill_ over every array in turn, pg_fill_bytes (one window / one stream / 5 and 9 parts advanced
together) over one column.  GB/s of algorithmic bytes, medians of `reps` launches timed with HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_gadgets_amd as pg  # noqa: E402

NINE = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2]


def main():
    which = sys.argv[1:] or ["c2", "c4", "c3"]
    eng = pg.Engine(0)
    for name in which:
        for spread in ([None] if name == "c2" else ["0", None]):
            if spread is not None:
                os.environ["PG_BENCH_SPREAD_GIB" if name == "c3" else "PG_BENCH_C4_SPREAD_GIB"] = spread
            wl = bench.Workload(name, eng, torch.device("cuda", 0), 0, 1, 20, -1)
            for k in ("PG_BENCH_SPREAD_GIB", "PG_BENCH_C4_SPREAD_GIB"):
                os.environ.pop(k, None)
            cols = wl.cols
            nbytes = wl.algo_bytes_per_launch
            label = "%s (%s)" % (name, "nine allocations" if not getattr(wl, "spread_gib", 0) else "slab, %g GiB strides" % wl.spread_gib)
            print("== %s: %.2f GB per launch" % (label, nbytes / 1e9), flush=True)
            ms = timed(lambda: wl.launch(0))
            print("  workload launch          %8.3f ms  %7.1f GB/s" % (ms, nbytes / ms / 1e6), flush=True)
            for rpt in (4096, 16384, 32768, 65536, 262144):
                ms = timed(lambda: eng.fill_columns(cols, rows_per_tile=rpt))
                print("  fill_columns tile %6d %8.3f ms  %7.1f GB/s" % (rpt, ms, nbytes / ms / 1e6), flush=True)

            def torch_fill():
                for k in NINE:
                    getattr(cols, k).fill_(7)
            ms = timed(torch_fill)
            print("  torch fill_ x 9 in turn  %8.3f ms  %7.1f GB/s" % (ms, nbytes / ms / 1e6), flush=True)
            one = cols.q_m
            ob = one.numel() * 8
            for streams in (0, 1, 5, 9):
                ms = timed(lambda: eng.fill_bytes(one, streams))
                print("  fill_bytes(q_m) parts=%d  %8.3f ms  %7.1f GB/s" % (streams, ms, ob / ms / 1e6), flush=True)
            ms = timed(lambda: one.fill_(3))
            print("  torch fill_(q_m)         %8.3f ms  %7.1f GB/s" % (ms, ob / ms / 1e6), flush=True)
            wl.release()
            del wl, cols, one
            torch.cuda.empty_cache()
    eng.close()


if __name__ == "__main__":
    main()
