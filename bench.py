#!/usr/bin/env python3
"""bench.py -- headline benchmark: gadget constraints/sec, range_check "256-bit" (BASELINE.json config C2).

A step = one pass of the hot path over one batch: 2^20 witnesses x (allocate + range_check(min=0, max=2^254)),
ladder length n = 255, 1031 gate rows + 1034 variables per witness, emitted into the 8 live composer columns +
the variable table, inputs already resident in HBM.  With N GPUs every rank processes its own 2^20 witnesses
(weak scaling, contiguous witness shards, no data-path collective: SURVEY.md section 8e / DESIGN.md).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the dominant
kernel on the stream it runs on; `cpu_baseline` times the CPU oracle ("port" of the reference algorithm, one
thread) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GATES_PER_WITNESS = 1031          # 4n+11, n = 255
VARS_PER_WITNESS = 1034           # 2n+523 + 1 (allocate)
BYTES_PER_GATE = 5 * 32 + 3 * 8   # 5 selector scalars + 3 Variable indices
BYTES_PER_VAR = 32
ALGO_BYTES_PER_WITNESS = GATES_PER_WITNESS * BYTES_PER_GATE + VARS_PER_WITNESS * BYTES_PER_VAR  # 222 792
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=20, help="witnesses per GPU per step = 2^this")
    ap.add_argument("--log2-chunk", type=int, default=-1, help="witnesses per launch = 2^this (-1: largest that fits)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="witnesses of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def cpu_baseline(sample: int):
    """oracle/ (C restatement of the reference algorithm incl. its per-bit pow), single thread, same workload shape"""
    from oracle import pyoracle as po
    from plonk_gadgets_amd import synth
    wit = synth.random_scalars(sample, seed=synth.SEED)
    po.lib()
    t0 = time.perf_counter()
    out = po.range_check_batch(synth.mont(0), synth.mont(2**254), wit, check=False, want_columns=False)
    dt = time.perf_counter() - t0
    return {"value": out["n_gates"] / dt, "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": f"{sample} witnesses x range_check(min=0,max=2^254) (n=255, {out['n_gates']} rows), "
                      f"oracle/gadgets.c single thread, {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group("nccl", device_id=dev)

    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import synth

    eng = pg.Engine(local_rank)
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    batch = 1 << args.log2_batch
    # rank r owns witnesses [r*batch, (r+1)*batch) of the global stream
    wit_np = synth.random_scalars(batch * world, seed=synth.SEED)[rank * batch:(rank + 1) * batch]
    wit = torch.from_numpy(np.ascontiguousarray(wit_np).view(np.int64)).to(dev)

    free, total = torch.cuda.mem_get_info(dev)
    if args.log2_chunk >= 0:
        chunk = min(batch, 1 << args.log2_chunk)
    else:
        chunk = batch
        while chunk > 1 and chunk * (ALGO_BYTES_PER_WITNESS + 8) > 0.85 * free:
            chunk >>= 1
    n_chunks = batch // chunk
    lay = eng.range_check_layout(mn, mx, chunk)
    assert (lay.gates_per_item, lay.vars_per_item) == (GATES_PER_WITNESS, VARS_PER_WITNESS)
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
    res = torch.empty((chunk,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(events=None):
        for c in range(n_chunks):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            # global numbering: this rank's shard starts at item rank*batch
            first = rank * batch + c * chunk
            eng.range_check_batch(mn, mx, wit[c * chunk:(c + 1) * chunk], 3 + first * GATES_PER_WITNESS,
                                  5 + first * VARS_PER_WITNESS, out=cols, result_vars=res)
            if events is not None:
                e1.record(stream)
                events.append((e0, e1))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = [a.elapsed_time(b) for a, b in events]
    avg_launch_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    constraints = world * batch * GATES_PER_WITNESS * args.steps
    value = constraints / elapsed
    achieved = chunk * ALGO_BYTES_PER_WITNESS / avg_launch_s / 1e9

    if rank == 0:
        line = {
            "metric": "gadget constraints/sec (range_check 256-bit)",
            "value": value,
            "unit": "constraints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64x4 (BLS12-381 scalar, Montgomery limbs)",
            "data": "synthetic (splitmix64 witnesses, uniform field elements)",
            "config": {"workload": "C2: 2^%d witnesses/GPU x (allocate + range_check(min=0,max=2^254)), n=255, "
                                   "1031 rows + 1034 vars per witness" % args.log2_batch,
                       "witnesses_per_gpu": batch, "witnesses_per_launch": chunk, "launches_per_step": n_chunks,
                       "sharding": "contiguous witness ranges per rank, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "pg::range_check_kernel<16>",
                         "algorithmic_bytes_per_launch": chunk * ALGO_BYTES_PER_WITNESS,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": len(kernel_ms)},
            "hbm_free_gb_at_start": free / 1e9, "hbm_total_gb": total / 1e9,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
