#!/usr/bin/env python3
"""bench.py -- headline benchmark: gadget constraints/sec, range_check "256-bit" (BASELINE.json config C2).

A step = one pass of the hot path over one batch: 2^20 witnesses x (allocate + range_check(min=0, max=2^254)),
ladder length n = 255, 1031 gate rows + 1034 variables per witness, emitted into the 8 live composer columns +
the variable table, inputs already resident in HBM.  With N GPUs every rank processes its own 2^20 witnesses
(weak scaling, contiguous witness shards emitted at their global numbering, no data-path collective in the timed
region: SURVEY.md section 8e / DESIGN.md "Multi-GPU"); the gather-inclusive rate of the chunked all-gather
pipeline is measured afterwards on a bounded sample and reported beside it ("allgather").

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the dominant
kernel on the stream it runs on; `cpu_baseline` times the CPU oracle ("port" of the reference algorithm, one
thread) on a bounded sample of the same workload on this box's host cores.

--workload c3 / c4 time the other BASELINE configs (fused scalar mix; max_bound with random 253-bit bounds) with
the same contract; they are secondary lines, the default (c2) is the judged one.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_GATE = 5 * 32 + 3 * 8   # 5 selector scalars + 3 Variable indices
BYTES_PER_VAR = 32
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2")
    ap.add_argument("--log2-batch", type=int, default=20, help="items per GPU per step = 2^this")
    ap.add_argument("--log2-chunk", type=int, default=-1, help="items per launch = 2^this (-1: largest that fits)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items timed on the CPU oracle (0: workload default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fill", action="store_true", help="skip the bare fill-kernel comparison measurement")
    ap.add_argument("--allgather-log2-chunk", type=int, default=12, help="N>1: witnesses per rank per gathered chunk")
    ap.add_argument("--allgather-chunks", type=int, default=8, help="N>1: chunks in the gather-inclusive sample (0: skip)")
    return ap.parse_args()


def cpu_baseline(workload: str, sample: int):
    """oracle/ (C restatement of the reference algorithm incl. its per-bit pow), single thread, same workload shape"""
    import numpy as np
    from oracle import pyoracle as po
    from plonk_gadgets_amd import synth
    po.lib()
    if workload == "c2":
        sample = sample or 512
        wit = synth.random_scalars(sample, seed=synth.SEED)
        t0 = time.perf_counter()
        out = po.range_check_batch(synth.mont(0), synth.mont(2**254), wit, check=False, want_columns=False)
        what = f"{sample} witnesses x range_check(min=0,max=2^254) (n=255)"
    elif workload == "c3":
        sample = sample or 8192
        v, y, s, a, b = mix_inputs(sample)
        t0 = time.perf_counter()
        out = po.scalar_mix_batch(v, y, s, a, b, check=False)
        what = f"{sample} items x (5 add_input + is_non_zero + conditionally_select_one + maybe_equal)"
    else:
        sample = sample or 1024
        mr, wt = c4_inputs(sample)
        t0 = time.perf_counter()
        out = po.max_bound_batch(mr, wt, check=False)
        what = f"{sample} items x max_bound(random 253-bit bound)"
    dt = time.perf_counter() - t0
    base = {"value": out["n_gates"] / dt, "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": f"{what}, {out['n_gates']} rows, oracle/gadgets.c single thread, {dt:.1f} s",
            "host_cores_available": os.cpu_count()}
    if workload == "c2":
        # BASELINE config 1, exactly as specified: 1 000 x range_check(v, "64-bit") through the faithful port, one thread
        c1 = synth.uniform_below(1000, 2**64 + 2**60, seed=synth.SEED)
        t1 = time.perf_counter()
        o1 = po.range_check_batch(synth.mont(0), synth.mont(2**64), c1, check=False, want_columns=False)
        d1 = time.perf_counter() - t1
        base["config_c1"] = {"value": o1["n_gates"] / d1, "unit": "constraints/s", "cores": 1, "kind": "port",
                             "sample": f"1000 witnesses x range_check(min=0,max=2^64) (n=65), {o1['n_gates']} rows, {d1:.1f} s"}
        # best-case CPU beside the faithful port: oracle/fast.c (mont(2^i) table, flat arrays, closed-form offsets)
        threads = min(os.cpu_count() or 1, 16)
        fast = po.range_check_fast(synth.mont(0), synth.mont(2**254), synth.random_scalars(16384, seed=synth.SEED),
                                   threads=threads)
        base["fast_variant"] = {"value": fast["n_gates"] / fast["seconds"], "unit": "constraints/s", "cores": threads,
                                "kind": "port (table-driven, threaded: oracle/fast.c)",
                                "sample": f"16384 witnesses, {fast['n_gates']} rows, {fast['seconds']:.3f} s "
                                          "(3.65 GB written to host memory)"}
    return base


def mix_inputs(n, seed=0xC3):
    """C3: v != 0 uniform; y uniform; s a bit; a uniform; b = a with probability 1/2 else uniform"""
    import numpy as np
    from plonk_gadgets_amd import synth
    v = synth.random_scalars(n, seed)
    v[(v == 0).all(axis=1)] = synth.mont(1)
    y = synth.random_scalars(n, seed + 1)
    bits = synth.splitmix64(n, seed + 2)
    one = np.array(synth.mont(1), dtype=np.uint64)
    s = np.where(((bits & np.uint64(1)) == 1)[:, None], one[None, :], np.zeros(4, np.uint64)[None, :])
    a = synth.random_scalars(n, seed + 3)
    b = synth.random_scalars(n, seed + 4)
    same = ((bits >> np.uint64(1)) & np.uint64(1)) == 1
    b[same] = a[same]
    return v, y, np.ascontiguousarray(s), a, b


def c4_inputs(n, seed=0xC4):
    """C4: bound = mont(uniform 253-bit integer) built limb-wise on the host for a small pool, tiled to n items;
    witness uniform field elements (about half below a 253-bit bound's scale is not needed for throughput)."""
    import numpy as np
    from plonk_gadgets_amd import synth
    pool = 4096
    raw = synth.splitmix64(4 * pool, seed).reshape(pool, 4)
    bounds = [(sum(int(raw[i, k]) << (64 * k) for k in range(4)) % (1 << 253)) for i in range(pool)]
    bounds[0:3] = [0, 1, 2]
    mr_pool = synth.scalars_from_ints(bounds)
    wit_pool = []
    wr = synth.splitmix64(4 * pool, seed + 1).reshape(pool, 4)
    for i, b in enumerate(bounds):
        r = sum(int(wr[i, k]) << (64 * k) for k in range(4))
        wit_pool.append(r % b if (i % 2 == 0 and b > 0) else r % synth.Q)
    wt_pool = synth.scalars_from_ints(wit_pool)
    reps = (n + pool - 1) // pool
    return np.ascontiguousarray(np.tile(mr_pool, (reps, 1))[:n]), np.ascontiguousarray(np.tile(wt_pool, (reps, 1))[:n])


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PG_FORCE_DIST=1: take the multi-rank code path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)
    distributed = world > 1 or os.environ.get("PG_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (the engine has no CPU path)")
    # PG_DIST_BACKEND=gloo is a rehearsal aid (several ranks sharing one GPU, where RCCL refuses duplicate devices)
    backend = os.environ.get("PG_DIST_BACKEND", "nccl")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import synth

    eng = pg.Engine(local_rank)
    batch = 1 << args.log2_batch
    free, total = torch.cuda.mem_get_info(dev)
    stream = torch.cuda.current_stream(dev)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

    # ---- workload set-up: returns launch(c) for chunk c, chunk size, rows/vars per launch --------------
    if args.workload == "c2":
        mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
        G, V = 1031, 1034
        per_item = G * BYTES_PER_GATE + V * BYTES_PER_VAR  # 222 792 B
        # rank r owns witnesses [r*batch, (r+1)*batch) of the global stream
        wit = to_dev(synth.random_scalars(batch * world, seed=synth.SEED)[rank * batch:(rank + 1) * batch])
        chunk = batch if args.log2_chunk < 0 else min(batch, 1 << args.log2_chunk)
        while args.log2_chunk < 0 and chunk > 1 and chunk * (per_item + 8) > 0.85 * free:
            chunk >>= 1
        lay = eng.range_check_layout(mn, mx, chunk)
        assert (lay.gates_per_item, lay.vars_per_item) == (G, V)
        cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
        res = torch.empty((chunk,), dtype=torch.int64, device=dev)
        rows_per_launch, vars_per_launch = lay.n_gates, lay.n_vars
        read_bytes = chunk * 32

        def launch(c):
            first = rank * batch + c * chunk  # global numbering of this rank's shard
            eng.range_check_batch(mn, mx, wit[c * chunk:(c + 1) * chunk], 3 + first * G, 5 + first * V, out=cols,
                                  result_vars=res)
        kernel = "pg::emit_kernel<pg::RangeCheckGD>"
        desc = ("C2: 2^%d witnesses/GPU x (allocate + range_check(min=0,max=2^254)), n=255, 1031 rows + 1034 vars "
                "per witness" % args.log2_batch)
    elif args.workload == "c3":
        chunk = batch if args.log2_chunk < 0 else min(batch, 1 << args.log2_chunk)
        ins = [to_dev(x) for x in mix_inputs(batch, seed=0xC3 + rank)]
        _, roff, voff = eng.ragged_buffers(chunk)
        lay, nerr = eng.scalar_mix_plan(ins[0][:chunk], roff, voff)
        assert nerr == 0 and (lay.n_gates, lay.n_vars) == (10 * chunk, 15 * chunk)
        rows_per_launch, vars_per_launch = lay.n_gates, lay.n_vars
        cols = pg.Columns.allocate(rows_per_launch, vars_per_launch, dev)
        res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
        read_bytes = chunk * 160

        def launch(c):
            # the plan (zero test + prefix sums) is part of the pass: the inputs decide the ragged layout
            part = [t[c * chunk:(c + 1) * chunk] for t in ins]
            eng.scalar_mix_plan_async(part[0], roff, voff)  # no host round trip: the buffers hold the worst case
            eng.scalar_mix_emit(*part, roff, voff, cols, res, 3, 5, 0)
        kernel = "pg::emit_kernel<pg::ScalarMixGD> (+ plan/scan kernels and the inversion pre-pass)"
        desc = ("C3: 2^%d items/GPU x (5 add_input + is_non_zero + conditionally_select_one + maybe_equal), one "
                "emit launch, 10 rows + 15 vars per item" % args.log2_batch)
    else:
        chunk = batch if args.log2_chunk < 0 else min(batch, 1 << args.log2_chunk)
        assert batch == chunk, "c4 is timed as one launch over the whole batch"
        mr_np, wt_np = c4_inputs(batch, seed=0xC4 + rank)
        mr, wt = to_dev(mr_np), to_dev(wt_np)
        nb, roff, voff = eng.ragged_buffers(chunk)
        lay = eng.max_bound_ragged_plan(mr, nb, roff, voff)
        rows_per_launch, vars_per_launch = lay.n_gates, lay.n_vars
        cols = pg.Columns.allocate(rows_per_launch, vars_per_launch, dev)
        res = torch.empty((chunk,), dtype=torch.int64, device=dev)
        read_bytes = chunk * 64

        def launch(c):
            eng.max_bound_ragged_plan_async(mr, nb, roff, voff)
            eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, cols, res, 3, 5)
        kernel = "pg::emit_kernel<pg::MaxBoundGD<true>> (+ plan/scan kernels and the inversion pre-pass)"
        desc = ("C4: 2^%d items/GPU x (allocate + max_bound(random 253-bit bound)), data-dependent ladder length, "
                "ragged rows" % args.log2_batch)

    n_chunks = batch // chunk
    algo_bytes_per_launch = rows_per_launch * BYTES_PER_GATE + vars_per_launch * BYTES_PER_VAR

    def step(events=None):
        for c in range(n_chunks):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            launch(c)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = [a.elapsed_time(b) for a, b in events]
    avg_launch_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    if os.environ.get("PG_BENCH_VERBOSE") and rank == 0:
        print("launch ms:", " ".join("%.2f" % t for t in kernel_ms), file=sys.stderr)
    constraints = world * rows_per_launch * n_chunks * args.steps
    value = constraints / elapsed
    achieved = algo_bytes_per_launch / avg_launch_s / 1e9

    # ---- a bare fill kernel on the same box for comparison (not an upper bound: the emitters' tiled eight-column
    # pattern sustains more than one linear stream does) ---------------------------------------------------------
    fill = None
    if not args.no_fill and rank == 0:
        nbytes = cols.q_m.numel() * 8
        fill = {"bytes_per_launch": nbytes,
                "what": "16 B/lane streaming stores over one selector column's buffer: short-lived workgroups of 16 KiB "
                        "(oneshot), long-lived ones writing 1 linear stream / 5 concurrent parts (the emitters' shape), "
                        "and torch's own fill_ kernel"}
        for streams in (0, 1, 5):
            eng.fill_bytes(cols.q_m, streams)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            reps = 5
            for _ in range(reps):
                eng.fill_bytes(cols.q_m, streams)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            key = "gbps_oneshot" if streams == 0 else "gbps_%d_stream%s" % (streams, "" if streams == 1 else "s")
            fill[key] = nbytes * reps / (e0.elapsed_time(e1) / 1e3) / 1e9
        cols.q_m.fill_(1)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            cols.q_m.fill_(1)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        fill["gbps_torch_fill"] = nbytes * 5 / (e0.elapsed_time(e1) / 1e3) / 1e9
        fill["gbps"] = max(v for k, v in fill.items() if k.startswith("gbps_"))

    # ---- N > 1: gather-inclusive rate of the chunked all-gather pipeline (bounded sample) -----------------
    allgather = None
    if distributed and backend == "nccl" and args.workload == "c2" and args.allgather_chunks > 0:
        from plonk_gadgets_amd import distributed as pd
        del cols
        torch.cuda.empty_cache()
        gchunk = 1 << args.allgather_log2_chunk
        pipe = pd.GatherPipeline(eng, mn, mx, gchunk)
        per_rank = gchunk * args.allgather_chunks
        pipe.run(wit[:2 * gchunk], 2 * gchunk)  # warm-up (communicator set-up)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        pipe.run(wit[:per_rank], per_rank)
        torch.cuda.synchronize(dev)
        dist.barrier()
        dt = time.perf_counter() - t1
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        allgather = {"value": world * per_rank * G / dt, "unit": "constraints/s (every rank ends with every shard)",
                     "witnesses_per_rank": per_rank, "witnesses_per_chunk": gchunk,
                     "bytes_per_rank_per_chunk": pipe.bytes_per_chunk(),
                     "ingest_gbps_per_gpu": (world - 1) * pipe.bytes_per_chunk() * args.allgather_chunks / dt / 1e9,
                     "collective": "one all_gather_into_tensor (RCCL) per packed chunk, double-buffered"}
        # the same stream with only the variable tables on the links; the other ranks' rows are regenerated locally
        try:
            del pipe
            torch.cuda.empty_cache()
            vpipe = pd.VariablesOnlyPipeline(eng, mn, mx, gchunk)
            vpipe.run(wit[:2 * gchunk], 2 * gchunk)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t1 = time.perf_counter()
            vpipe.run(wit[:per_rank], per_rank)
            torch.cuda.synchronize(dev)
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtv = float(tt.item())
            allgather["variables_only"] = {
                "value": world * per_rank * G / dtv, "unit": "constraints/s (every rank ends with every shard)",
                "bytes_per_rank_per_chunk": vpipe.bytes_on_the_links_per_chunk(),
                "ingest_gbps_per_gpu": (world - 1) * vpipe.bytes_on_the_links_per_chunk() * args.allgather_chunks / dtv / 1e9,
                "collective": "one all_gather_into_tensor of the variable tables per chunk; selectors and wire indices of "
                              "the other ranks' chunks regenerated locally (pg_range_check_structure_batch)"}
        except Exception as ex:  # the secondary figure must never cost the headline line
            allgather["variables_only"] = {"error": repr(ex)}

    if rank == 0:
        traffic = None
        if os.path.exists(PMC_SUMMARY):
            try:
                pmc = json.load(open(PMC_SUMMARY))
                ent = pmc.get(args.workload, {}).get(str(chunk))
                if ent:
                    traffic = ent["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "gadget constraints/sec (range_check 256-bit)" if args.workload == "c2"
                      else f"gadget constraints/sec ({args.workload})",
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
            "data": "synthetic (splitmix64 streams, uniform field elements)",
            "config": {"workload": desc, "items_per_gpu": batch, "items_per_launch": chunk,
                       "launches_per_step": n_chunks,
                       "sharding": "contiguous witness ranges per rank at global numbering, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": kernel,
                         "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                         "input_bytes_per_launch": read_bytes,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": len(kernel_ms),
                         "bare_fill": fill},
            "hbm_free_gb_at_start": free / 1e9, "hbm_total_gb": total / 1e9,
        }
        if allgather:
            line["allgather"] = allgather
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
