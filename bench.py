#!/usr/bin/env python3
"""bench.py -- headline benchmark: gadget constraints/sec, range_check "256-bit" (BASELINE.json config C2).

A step = one pass of the hot path over one batch: 2^20 witnesses x (allocate + range_check(min=0, max=2^254)),
ladder length n = 255, 1031 gate rows + 1034 variables per witness, emitted into the 8 live composer columns +
the variable table, inputs already resident in HBM.  With N GPUs every rank processes its own 2^20 witnesses
(weak scaling, contiguous witness shards emitted at their global numbering, no data-path collective in the timed
region: SURVEY.md section 8e / DESIGN.md "Multi-GPU"); the gather-inclusive rate of the chunked all-gather
pipeline is measured afterwards on a bounded sample and reported beside it ("allgather").

`--gpus N` is honoured however the script is started:
  * under torch.distributed.run (WORLD_SIZE set): this process is one of the N ranks; WORLD_SIZE must equal N;
  * started plainly with N > 1: a parent that never touches the GPU starts N rank processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), waits for them and exits with their worst code.  A box with
    fewer than N GPUs is refused loudly -- the script never silently runs fewer ranks than it was asked for.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the dominant
kernel on the stream it runs on; `cpu_baseline` times the CPU oracle ("port" of the reference algorithm, one
thread) on a bounded sample of the same workload on this box's host cores.  At N = 1 the line also carries
`secondary`: BASELINE configs C3 (fused scalar mix) and C4 (max_bound, random 253-bit bounds) timed in the same
process with the same steps/warmup, each with its own roofline object.

--workload c3 / c4 make one of those the line's headline instead (profiling aid); the default (c2) is the judged one.

Exit codes: 0 = the line was printed; 2 = refused (fewer GPUs than asked for); 3 = N > 1 only: the headline was measured and
printed, but the gather-inclusive sample hung past --allgather-timeout and was abandoned (its "allgather" object says so).
`roofline.launch_ms` = {min, median, max} of the timed launches (frac is computed from the median); `roofline.traffic` is
null, with the reason in `traffic_source`, when profiles/pmc_summary.json was collected for other kernel sources.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_GATE = 5 * 32 + 3 * 8   # 5 selector scalars + 3 Variable indices
BYTES_PER_VAR = 32
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c2_values"], default="c2")
    ap.add_argument("--log2-batch", type=int, default=20, help="items per GPU per step = 2^this")
    ap.add_argument("--log2-chunk", type=int, default=-1, help="items per launch = 2^this (-1: largest that fits)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items timed on the CPU oracle (0: workload default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 measurements after the headline")
    ap.add_argument("--no-fill", action="store_true", help="skip the bare-fill comparison launches (profiling runs: only the workload's kernels)")
    ap.add_argument("--allgather-log2-chunk", type=int, default=12, help="N>1: witnesses per rank per gathered chunk")
    ap.add_argument("--allgather-timeout", type=float, default=240.0, help="N>1: seconds before the gather-inclusive sample is abandoned")
    ap.add_argument("--allgather-chunks", type=int, default=8, help="N>1: chunks in the gather-inclusive sample (0: skip)")
    ap.add_argument("--n1-value", type=float, default=0.0, help="N>1: the N = 1 run's `value` (constraints/s), for the two efficiency figures "
                                                                 "of the line (absent: this run's own rank-0 rate stands in for it)")
    return ap.parse_args(argv)


# ---- N > 1 without a launcher: start the ranks ourselves -------------------------------------------------------

def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus() -> int:
    """number of GPUs without initialising the HIP runtime in this process (device_count() does not, on this image)"""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(args) -> int:
    """parent of an N-rank run: never touches the GPU itself (no exec of a process that has: children are plain
    subprocesses), one child per GPU, the children's stdout/stderr are ours"""
    n = args.gpus
    have = visible_gpus()
    rehearsal = os.environ.get("PG_DIST_BACKEND", "nccl") != "nccl"  # gloo: several ranks may share a GPU
    if have < n and not rehearsal:
        print(f"bench.py --gpus {n}: needs {n} GPUs, this box has {have} -- refusing to run fewer ranks than asked "
              f"(PG_DIST_BACKEND=gloo rehearses N ranks on fewer GPUs)", file=sys.stderr)
        return 2
    if have < 1:
        print("bench.py needs a gfx950 GPU (the engine has no CPU path)", file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "PG_BENCH_SPAWNED": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        for p in procs:
            rc = p.wait()
            worst = worst or rc
            if rc != 0:  # a dead rank would leave the others waiting at a barrier
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return worst


# ---- inputs ------------------------------------------------------------------------------------------------------

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
    witness below its bound for every other item, else a uniform field element"""
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


def cpu_baseline(workload: str, sample: int):
    """oracle/ (C restatement of the reference algorithm incl. its per-bit pow), single thread, same workload shape"""
    from oracle import pyoracle as po
    from plonk_gadgets_amd import synth
    po.lib()
    if workload == "c2_values":  # (the CPU side has no refresh: the reference rebuilds every row, which is what C2's baseline times)
        workload = "c2"
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
        # best-case CPU beside the faithful port: oracle/fast.c (mont(2^i) table, flat arrays, closed-form offsets) on
        # the host's cores -- 16, 64 and all of them (nproc printed), the best reported with its thread count
        ncpu = os.cpu_count() or 1
        nwit = 32768  # 7.3 GB of columns in host memory: enough work per thread for the timing to mean something
        wit = synth.random_scalars(nwit, seed=synth.SEED)
        sweep = []
        for threads in sorted({min(16, ncpu), min(64, ncpu), ncpu}):
            # (three passes over fresh arrays, the last one timed: the first lets every thread first-touch the output pages
            # it writes -- timed from the main thread's zero-fill instead, 256 threads were slower than 16: page placement)
            best_of = po.range_check_fast(synth.mont(0), synth.mont(2**254), wit, threads=threads, timed_passes=3)
            sweep.append({"cores": threads, "value": best_of["n_gates"] / best_of["seconds"], "seconds": round(best_of["seconds"], 4)})
        best = max(sweep, key=lambda r: r["value"])
        base["fast_variant"] = {"value": best["value"], "unit": "constraints/s", "cores": best["cores"],
                                "kind": "port (table-driven, threaded: oracle/fast.c)", "nproc": ncpu, "thread_sweep": sweep,
                                "sample": f"{nwit} witnesses, {best_of['n_gates']} rows, {best['seconds']:.3f} s with {best['cores']} "
                                          "threads (7.3 GB written to host memory; the third pass over arrays the threads touched first themselves)"}
    return base


# ---- one workload on this rank's GPU ---------------------------------------------------------------------------

class Workload:
    """set-up of one BASELINE config on `dev`: launch(c) emits chunk c of a step"""

    def __init__(self, name, eng, dev, rank, world, log2_batch, log2_chunk):
        import numpy as np
        import torch
        import plonk_gadgets_amd as pg
        from plonk_gadgets_amd import synth
        self.name, self.eng, self.dev = name, eng, dev
        batch = 1 << log2_batch
        free, _ = torch.cuda.mem_get_info(dev)

        def to_dev(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

        chunk = batch if log2_chunk < 0 else min(batch, 1 << log2_chunk)
        if name == "c2":
            mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
            self.mn, self.mx = mn, mx
            G, V = 1031, 1034
            per_item = G * BYTES_PER_GATE + V * BYTES_PER_VAR  # 222 792 B
            # rank r owns witnesses [r*batch, (r+1)*batch) of the global stream
            wit = to_dev(synth.random_scalars(batch * world, seed=synth.SEED)[rank * batch:(rank + 1) * batch])
            self.wit = wit
            while log2_chunk < 0 and chunk > 1 and chunk * (per_item + 8) > 0.85 * free:
                chunk >>= 1
            lay = eng.range_check_layout(mn, mx, chunk)
            assert (lay.gates_per_item, lay.vars_per_item) == (G, V)
            cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
            res = torch.empty((chunk,), dtype=torch.int64, device=dev)
            self.rows_per_launch, self.vars_per_launch = lay.n_gates, lay.n_vars
            self.read_bytes = chunk * 32

            def launch(c):
                first = rank * batch + c * chunk  # global numbering of this rank's shard
                eng.range_check_batch(mn, mx, wit[c * chunk:(c + 1) * chunk], 3 + first * G, 5 + first * V, out=cols,
                                      result_vars=res)
            self.kernel = "pg::emit_kernel<pg::RangeCheckGD>"
            self.desc = ("C2: 2^%d witnesses/GPU x (allocate + range_check(min=0,max=2^254)), n=255, 1031 rows + 1034 "
                         "vars per witness" % log2_batch)
        elif name == "c2_values":
            # the witness refresh of C2's circuit (prover.clear_witness() and the same calls on other witnesses,
            # /root/reference/tests/scalar_gadgets_tests.rs:108-119): the variable table alone, 1034 x 32 B per witness
            mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
            G, V = 1031, 1034
            wit = to_dev(synth.random_scalars(batch, seed=synth.SEED + 1))
            lay = eng.range_check_layout(mn, mx, chunk)
            table = torch.empty((lay.n_vars, 4), dtype=torch.int64, device=dev)
            self.rows_per_launch, self.vars_per_launch = lay.n_gates, lay.n_vars  # (rows of the circuit: found in place, not written)
            self.rows_written_per_launch = 0
            self.read_bytes = chunk * 32
            cols, res = table, None

            def launch(c):
                eng.range_check_values_batch(mn, mx, wit[c * chunk:(c + 1) * chunk], table)
            self.kernel = "pg::emit_kernel<pg::RangeCheckGD, EMIT_VALUES> (pg_range_check_values_batch)"
            self.desc = ("C2 witness refresh: 2^%d witnesses/GPU x the assignments of (allocate + range_check(min=0,max=2^254)), "
                         "1034 variables per witness, no rows" % log2_batch)
        elif name == "c3":
            ins = [to_dev(x) for x in mix_inputs(batch, seed=0xC3 + rank)]
            _, roff, voff = eng.ragged_buffers(chunk)
            self.plan_buffers = (None, roff, voff)  # (what the launch's plan writes: the exhaustive parity test reads them)
            lay, nerr = eng.scalar_mix_plan(ins[0][:chunk], roff, voff)
            assert nerr == 0 and (lay.n_gates, lay.n_vars) == (10 * chunk, 15 * chunk)
            self.rows_per_launch, self.vars_per_launch = lay.n_gates, lay.n_vars
            # a circuit of 2.4 GB: its five selector columns, written in lock step, go 24 GiB apart in one slab -- a 99-GiB allocation.
            # Where nine allocations in a row happen to land decides 0.51 ... 0.59 ms per step otherwise (DESIGN.md section 2,
            # tools/c3_instances.py, tools/placement_sweep.py); round 5, five boxes, slab / nine allocations: 0.595 / 0.585, 0.589 /
            # 0.595, 0.595 / 0.577, 0.562 / 0.583, 0.585 / 0.516 of peak -- level on most, but only the plain layout has the bad draws.
            # The line reports both (`one_slab` / `nine_allocations`); PG_BENCH_SPREAD_GIB=0 makes the plain layout the headline one
            self.spread_gib = float(os.environ.get("PG_BENCH_SPREAD_GIB", "24"))
            cols = self.allocate_columns(lay.n_gates, lay.n_vars, self.spread_gib)
            res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
            self.read_bytes = chunk * 160

            def launch(c):
                # the plan (zero test + prefix sums) is part of the pass: the inputs decide the ragged layout
                part = [t[c * chunk:(c + 1) * chunk] for t in ins]
                # no host round trip: the buffers hold the worst case
                if os.environ.get("PG_C3_SEPARATE_PLAN") == "1":  # (A/B: the plan as its own call ahead of the emit call)
                    eng.scalar_mix_plan_async(part[0], roff, voff)
                    eng.scalar_mix_emit(*part, roff, voff, self.cols, res, 3, 5, 0)
                else:
                    eng.scalar_mix_planned(*part, roff, voff, self.cols, res, None, 3, 5, 0)
            self.kernel = ("one step (pg_scalar_mix_planned_batch): pg::scalar_mix_vars_kernel<true> (prefix sums, inversions, "
                           "variable table), then pg::rows_periodic_kernel<pg::ScalarMixGD> (+ the generic rows launch for "
                           "tiles with a failing item)")
            self.desc = ("C3: 2^%d items/GPU x (5 add_input + is_non_zero + conditionally_select_one + maybe_equal), "
                         "one emit launch, 10 rows + 15 vars per item" % log2_batch)
        else:
            assert batch == chunk, "c4 is timed as one launch over the whole batch"
            mr_np, wt_np = c4_inputs(batch, seed=0xC4 + rank)
            mr, wt = to_dev(mr_np), to_dev(wt_np)
            nb, roff, voff = eng.ragged_buffers(chunk)
            self.plan_buffers = (nb, roff, voff)
            lay = eng.max_bound_ragged_plan(mr, nb, roff, voff)
            self.rows_per_launch, self.vars_per_launch = lay.n_gates, lay.n_vars
            # columns of 17 GB each lie far apart by themselves; 32 GiB between the selector columns (16-GiB gaps) is still worth
            # 1.5 % over nine allocations in a row (ten of each in turn: 17.34-17.71 against 17.64-17.98 ms, tools/c4_instances.py)
            self.spread_gib = float(os.environ.get("PG_BENCH_C4_SPREAD_GIB", "32"))
            cols = self.allocate_columns(lay.n_gates, lay.n_vars, self.spread_gib)
            res = torch.empty((chunk,), dtype=torch.int64, device=dev)
            self.read_bytes = chunk * 64

            def launch(c):
                eng.max_bound_ragged_plan_async(mr, nb, roff, voff)
                eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, self.cols, res, 3, 5)
            self.kernel = "pg::emit_kernel<pg::MaxBoundGD<true>> (+ the plan kernel and the inversion pre-pass)"
            self.desc = ("C4: 2^%d items/GPU x (allocate + max_bound(random 253-bit bound)), data-dependent ladder "
                         "length, ragged rows" % log2_batch)
        self.launch, self.cols, self.res = launch, cols, res
        self.batch, self.chunk, self.n_chunks = batch, chunk, batch // chunk
        rows_written = getattr(self, "rows_written_per_launch", self.rows_per_launch)
        self.algo_bytes_per_launch = rows_written * BYTES_PER_GATE + self.vars_per_launch * BYTES_PER_VAR

    def allocate_columns(self, n_gates, n_vars, spread_gib):
        """the nine output arrays: ONE slab with the selector columns spread_gib GiB apart, or -- spread_gib =
        0, or a card without the room -- nine allocations in a row; self.spread_gib / self.slab_bytes say which it became"""
        import torch
        import plonk_gadgets_amd as pg
        self.slab_bytes = 0
        if spread_gib > 0:
            try:
                cols = pg.Columns.allocate(n_gates, n_vars, self.dev, spread_gib=spread_gib)
                self.spread_gib, self.slab_bytes = spread_gib, cols.slab.numel() * 8
                return cols
            except torch.OutOfMemoryError:  # a card that does not have the room: nine allocations, and the line says so
                pass
        self.spread_gib = 0.0
        return pg.Columns.allocate(n_gates, n_vars, self.dev)

    def reallocate_columns(self, spread_gib):
        """the same workload into other arrays (c3 / c4: the launch writes self.cols)"""
        import torch
        n_gates, n_vars = self.cols.q_m.shape[0], self.cols.var_values.shape[0]
        self.cols = None
        torch.cuda.empty_cache()
        self.cols = self.allocate_columns(n_gates, n_vars, spread_gib)

    def layout_note(self):
        if getattr(self, "spread_gib", 0):
            return {"column_layout": "one slab, selector columns %g GiB apart" % self.spread_gib, "slab_total_bytes": self.slab_bytes}
        return {"column_layout": "nine allocations in a row"}

    def release(self):
        self.launch = self.cols = self.res = self.plan_buffers = None


def hip_runtime():
    """the HIP runtime this process already has (torch's, which the library binds to as well: the loader returns the loaded
    object for the soname)"""
    import ctypes as C
    import torch  # noqa: F401
    path = None
    try:  # the very object the process has mapped (torch's), whatever its soname
        for line in open("/proc/self/maps"):
            if "libamdhip64.so" in line:
                path = line.split()[-1]
                break
    except OSError:
        pass
    hip = None
    for cand in ([path] if path else []) + ["libamdhip64.so.7", "libamdhip64.so"]:
        try:
            hip = C.CDLL(cand)
            break
        except OSError:
            continue
    if hip is None:
        raise RuntimeError("the HIP runtime (libamdhip64.so) is not loadable")
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventDestroy.argtypes = [C.c_void_p]
    return hip


class TimingEvent:
    """a HIP event that only measures time: hipEventDisableSystemFence (hip_runtime_api.h: "for events that are only being used
    to measure timing ... avoiding the cost of cache writeback and invalidation, and the performance impact of those actions on
    the execution of following work").  After every call of the short C3 step a default event (torch.cuda.Event) costs 4.8 us,
    this one 3.8 (tools/event_cost.py); the timed region itself is bracketed by device synchronisation, not by events."""
    DISABLE_SYSTEM_FENCE = 0x20000000

    def __init__(self, hip):
        import ctypes as C
        self.hip, self.h = hip, C.c_void_p()
        if hip.hipEventCreateWithFlags(C.byref(self.h), self.DISABLE_SYSTEM_FENCE) != 0:
            raise RuntimeError("hipEventCreateWithFlags failed")

    def record(self, stream_handle):
        if self.hip.hipEventRecord(self.h, stream_handle) != 0:
            raise RuntimeError("hipEventRecord failed")
        return self

    def elapsed_ms(self, later):
        import ctypes as C
        ms = C.c_float()
        if self.hip.hipEventElapsedTime(C.byref(ms), self.h, later.h) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value

    def __del__(self):
        try:
            if self.h:
                self.hip.hipEventDestroy(self.h)
        except Exception:  # (interpreter shutdown: the runtime may be gone already)
            pass


def measure(wl: Workload, steps: int, warmup: int, sync_all):
    """W untimed steps, then exactly K timed steps bracketed by sync_all() (barrier + device synchronise); HIP events on
    the launch stream around every launch give the per-launch duration the roofline is computed from"""
    import ctypes as C
    import torch
    stream = C.c_void_p(torch.cuda.current_stream(wl.dev).cuda_stream)
    hip = hip_runtime()

    def step(events=None):
        # ONE event per launch boundary (launch i runs from boundary i to boundary i + 1): an event record is a packet of its own
        # on the stream -- with a pair around every launch, consecutive launches stood two records apart, ~15 us per C3 step
        for c in range(wl.n_chunks):
            wl.launch(c)
            if events is not None:
                events.append(TimingEvent(hip).record(stream))

    for _ in range(warmup):
        step()
    sync_all()
    events = [TimingEvent(hip)]
    t0 = time.perf_counter()
    events[0].record(stream)
    for _ in range(steps):
        step(events)
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_ms(b) for a, b in zip(events[:-1], events[1:])]
    return elapsed, kernel_ms


def bare_fill(wl: Workload, reps: int = 5):
    """"a bare fill-kernel ceiling measured on the same box" -- in this process, on the very arrays the workload has
    just written.  `columns`: pg_fill_columns, the emitters' store stream with nothing behind it (five selector columns in lock
    step, the wires, the variable table; tiles of 32768 rows) = the workload's store ceiling ON THESE ARRAYS; `one_window` /
    `one_stream`: pg_fill_bytes over the variable table alone (short-lived workgroups of 8 KiB, two per CU = one moving window of a few
    MiB, the shape that does not care where the table lies; long-lived ones);
    `torch_fill`: torch's fill_ over every array in turn.  GB/s, medians of `reps` launches (HIP events on the launch stream)."""
    import ctypes as C
    import torch
    stream = C.c_void_p(torch.cuda.current_stream(wl.dev).cuda_stream)
    hip = hip_runtime()
    eng = wl.eng

    def med_ms(fn):
        fn()
        ev = [TimingEvent(hip).record(stream)]
        for _ in range(reps):
            fn()
            ev.append(TimingEvent(hip).record(stream))
        torch.cuda.synchronize(wl.dev)
        ms = sorted(a.elapsed_ms(b) for a, b in zip(ev[:-1], ev[1:]))
        return ms[len(ms) // 2]

    out = {"what": "bare store streams over the arrays this workload writes, timed in this process: GB/s"}
    cols = wl.cols
    if isinstance(cols, torch.Tensor):  # (the witness refresh writes ONE array: the variable table)
        arrays, table = [cols], cols
    else:
        arrays = [getattr(cols, k) for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")]
        table = cols.var_values
        nbytes = cols.nbytes()
        out["columns"] = nbytes / med_ms(lambda: eng.fill_columns(cols)) / 1e6
        out["columns_kernel"] = "pg::fill_columns_kernel (pg_fill_columns): the emitters' sweeps, one constant, no table, no arithmetic"
    tb = table.numel() * 8
    out["one_window"] = tb / med_ms(lambda: eng.fill_bytes(table, 0)) / 1e6
    out["one_stream"] = tb / med_ms(lambda: eng.fill_bytes(table, 1)) / 1e6

    def torch_fill():
        for a in arrays:
            a.fill_(0x5A)
    out["torch_fill"] = sum(a.numel() * 8 for a in arrays) / med_ms(torch_fill) / 1e6
    out["best"] = max(v for k, v in out.items() if isinstance(v, float))
    return out


def roofline_with_fill(wl: Workload, kernel_ms):
    """the roofline object plus the same-process bare-fill ceiling (measured AFTER the timed steps: it overwrites the outputs)"""
    r = roofline_of(wl, kernel_ms)
    try:
        r["bare_fill"] = bare_fill(wl)
        r["frac_of_bare_fill"] = r["achieved"] / r["bare_fill"]["best"]
    except Exception as ex:  # a comparison point must never cost the line
        r["bare_fill"] = {"error": repr(ex)}
    return r


def f_rows_of(comp, dev, steps: int, note: dict):
    """pg_composer_materialize (per row 328 B written -- seven constant columns, w_4, three wire-value columns -- and 32 B read per
    Variable, each once) and pg_composer_permutation (four sigma columns of 8 B written per PADDED row) on a filled composer: median / min /
    max of a few synchronous calls, outputs allocated once ahead of them, rows/s and the HBM roofline of each"""
    import ctypes as C
    import torch
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    n = comp.circuit_size()
    padded = 1 << (n - 1).bit_length()
    names = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add", "w_4_value", "w_l_value",
             "w_r_value", "w_o_value")
    t = {k: torch.empty((n, 4), dtype=torch.int64, device=dev) for k in names}
    t["w_4"] = torch.empty((n,), dtype=torch.int64, device=dev)
    sigma = torch.empty((4, padded), dtype=torch.int64, device=dev)
    fc = _lib.FullColumnsC(**{k: v.data_ptr() for k, v in t.items()})

    def timed(fn):
        fn()  # warm-up (scratch of the permutation is grow-only)
        ms = []
        for _ in range(max(3, min(steps, 10))):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            assert fn() == 0
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
        ms.sort()
        return ms[len(ms) // 2], ms[0], ms[-1]

    out = {"rows": n, "sigma_padded_to": padded}
    med, lo, hi = timed(lambda: lib.pg_composer_materialize(comp._h, C.byref(fc)))
    # algorithmic bytes: 328 B written per row; read: every assignment ONCE (32 B per Variable -- the rows of a batched call take their
    # item's Variables from an LDS window read linearly, and the wires of the closed-form kinds are computed, not read back: csrc/materialize.hpp)
    # -- but for the 256 bit Variables per bound block of a uniform ladder gadget, which are made from the block's T (note["bit_variables"]).
    wr, rd = 328 * n, 32 * (comp.num_variables() - note.get("bit_variables", 0))
    out["materialize"] = {"ms": {"min": lo, "median": med, "max": hi}, "rows_per_s": n / (med / 1e3),
                          "roofline": {"bound": "hbm", "achieved": (wr + rd) / (med / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": (wr + rd) / (med / 1e3) / 1e9 / HBM_PEAK_GBPS,
                                       "algorithmic_bytes": {"written": wr, "read": rd},
                                       "frac_counting_writes_only": wr / (med / 1e3) / 1e9 / HBM_PEAK_GBPS,
                                       "kernel": note["materialize"]}}
    med, lo, hi = timed(lambda: lib.pg_composer_permutation(comp._h, padded, sigma.data_ptr()))
    # algorithmic bytes: four sigma columns of 8 B per PADDED row written; the rows of the batched appends are linked in closed form
    # (nothing read: csrc/permutation.hpp, perm_ladder_kernel / perm_template_kernel)
    pb = 32 * padded
    out["permutation"] = {"ms": {"min": lo, "median": med, "max": hi}, "rows_per_s": n / (med / 1e3),
                          "roofline": {"bound": "hbm", "achieved": pb / (med / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": pb / (med / 1e3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": pb,
                                       "kernel": note["permutation"]}}
    del t, sigma
    torch.cuda.empty_cache()
    return out


def next_rows_secondary(eng, dev, log2_batch: int, steps: int):
    """f1 pg_composer_materialize and f2 pg_composer_permutation on three composers -- C2-shaped (2^log2_batch x
    range_check(0, 2^254)), C4-shaped (twice as many max_bound items with a 253-bit bound of their own: about the same number of rows)
    and C3-shaped (2^(log2_batch + 4) fused scalar items) -- each filled by ONE batched append, as tests/test_gpu_frows_exhaustive.py
    verifies them word for word; then the same two calls on a circuit built one call at a time (single_calls)."""
    import numpy as np
    import torch
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import synth
    batch = 1 << log2_batch

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    comp = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    comp.range_check_batch(pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254), to_dev(synth.random_scalars(batch, seed=synth.SEED + 2)))
    out = f_rows_of(comp, dev, steps, {
        "bit_variables": 512 * batch,
        "materialize": "pg::materialize_items_kernel<MAT_SELF, WIRES_RANGE_CHECK> (one launch per batched call: constant columns, w_4, three "
                       "wire-value columns from an LDS window of the items' Variables that a wave of its own fetches ahead, wires in closed form, "
                       "the bits' assignments from the blocks' T)",
        "permutation": "pg::perm_ladder_kernel<false> (sigma of the ladder gadgets' rows in closed form) + perm_identity_kernel (the padding)"})
    out["config"] = {"workload": "composer of 2^%d x (allocate + range_check(0, 2^254)): %d rows, sigma padded to %d" % (log2_batch, out["rows"], out["sigma_padded_to"])}
    comp.close()
    del comp
    torch.cuda.empty_cache()
    try:  # C4's shape: per-item public bounds -- rows and Variables by the call's prefix sums
        mr, wt = c4_inputs(2 * batch, seed=0xC4 + 9)
        comp = pg.StandardComposer(eng, 3 + 2 * batch * 515 + 8, 5 + 2 * batch * 517 + 8)
        comp.max_bound_ragged_batch(to_dev(mr), to_dev(wt))
        r = f_rows_of(comp, dev, steps, {
            "bit_variables": 256 * 2 * batch,
            "materialize": "pg::materialize_items_kernel<MAT_SELF, WIRES_MAX_BOUND, true> (ladder lengths from the call's prefix sums, through the loader wave)",
            "permutation": "pg::perm_ladder_kernel<true> (an item's ladder length and place from a window of the call's prefix sums) + perm_identity_kernel"})
        r["config"] = {"workload": "composer of 2^%d x (allocate + max_bound(a 253-bit bound per item)): %d rows, sigma padded to %d" % (log2_batch + 1, r["rows"], r["sigma_padded_to"])}
        out["c4_shaped"] = r
        comp.close()
        del comp
    except Exception as ex:  # (a side figure)
        out["c4_shaped"] = {"error": repr(ex)}
    torch.cuda.empty_cache()
    try:  # C3's shape: ten rows and fifteen Variables per item
        nmix = 16 * batch
        comp = pg.StandardComposer(eng, 3 + nmix * 10 + 8, 5 + nmix * 15 + 8)
        comp.scalar_mix_batch(*[to_dev(x) for x in mix_inputs(nmix, seed=0xC3 + 9)])
        r = f_rows_of(comp, dev, steps, {
            "materialize": "pg::materialize_items_kernel<MAT_SELF, WIRES_MIX> (the item's wires from a 30-entry table)",
            "permutation": "pg::perm_template_kernel<false> (own Variables' cycles from the kind's wire table) + perm_identity_kernel"})
        r["config"] = {"workload": "composer of 2^%d fused scalar items (C3's item): %d rows, sigma padded to %d" % (log2_batch + 4, r["rows"], r["sigma_padded_to"])}
        out["c3_shaped"] = r
        comp.close()
        del comp
    except Exception as ex:
        out["c3_shaped"] = {"error": repr(ex)}
    torch.cuda.empty_cache()
    try:  # the same two calls on a circuit built the reference's way: ONE allocate + range_check at a time (tests/range_gadgets_tests.rs:29-44)
        out["single_calls"] = next_rows_of_single_calls(eng, dev, 4096)
    except Exception as ex:  # (a side figure)
        out["single_calls"] = {"error": repr(ex)}
    return out


def next_rows_of_single_calls(eng, dev, calls: int):
    """f1 / f2 on a composer filled by `calls` x (AllocatedScalar::allocate, range_check(0, 2^254)) issued one by one through the command
    queue: a flushed run of identical calls leaves the batched append's footprint, so both calls take the closed forms"""
    import ctypes as C
    import torch
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    S = pg.BlsScalar.from_int
    comp = pg.StandardComposer(eng, 3 + calls * 1031 + 8, 5 + calls * 1034 + 8)
    comp.queue(True)
    mn, mx = S(0), S(2**254)
    scalars = [S(1000 + i) for i in range(calls)]
    t0 = time.perf_counter()
    for sc in scalars:
        pg.range_check(comp, mn, mx, pg.AllocatedScalar.allocate(comp, sc))
    comp.sync()
    append_s = time.perf_counter() - t0
    n = comp.circuit_size()
    padded = 1 << (n - 1).bit_length()
    t = comp.materialize()  # (allocates its outputs; timed below into the same arrays)
    fc = _lib.FullColumnsC(**{k: v.data_ptr() for k, v in t.items()})
    sigma = torch.empty((4, padded), dtype=torch.int64, device=dev)
    lib = comp._lib

    def med(fn):
        fn()
        ms = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            assert fn() == 0
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t1) * 1e3)
        return sorted(ms)[2]
    m_ms = med(lambda: lib.pg_composer_materialize(comp._h, C.byref(fc)))
    p_ms = med(lambda: lib.pg_composer_permutation(comp._h, padded, sigma.data_ptr()))
    return {"workload": "%d x (allocate, range_check(0, 2^254)) one call at a time, command queue on: %d rows, sigma padded to %d" % (calls, n, padded),
            "append_us_per_pair_from_python": append_s / calls * 1e6,
            "materialize": {"ms": m_ms, "rows_per_s": n / (m_ms / 1e3)}, "permutation": {"ms": p_ms, "rows_per_s": n / (p_ms / 1e3)}}


def kernel_sources_sha256() -> str:
    """what the library is built from (plonk_gadgets_amd.build.kernel_sources_sha256): the PMC passes under profiles/ say
    which sources they were collected for"""
    from plonk_gadgets_amd import build as pg_build
    return pg_build.kernel_sources_sha256()


def pmc_traffic(workload: str, chunk: int):
    """HBM bytes per launch of this workload from the tracked PMC passes (offline: rocprofv3 cannot run inside bench.py).
    Only if they were collected for the kernel sources this run's library is built from: a summary that has gone stale
    yields null and says why."""
    if not os.path.exists(PMC_SUMMARY):
        return None, "no profiles/pmc_summary.json"
    try:
        summ = json.load(open(PMC_SUMMARY))
        ent = summ.get(workload, {}).get(str(chunk))
        if not ent:
            return None, "profiles/pmc_summary.json has no entry for this workload and launch size"
        have, want = ent.get("kernel_sources_sha256"), kernel_sources_sha256()
        if have != want:
            return None, ("profiles/pmc_summary.json (%s) was collected for other kernel sources (%s..., this run: %s...): "
                          "stale, not reported" % (ent.get("round", "?"), str(have)[:12], want[:12]))
        return ent["hbm_bytes_per_launch"], ("profiles/pmc_summary.json (%s, offline rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE "
                                             "passes of this command at these kernel sources, %s...; not measured by this run)"
                                             % (ent.get("round", "?"), want[:12]))
    except Exception as exc:  # a summary that cannot be read is no reason to lose the run
        return None, "profiles/pmc_summary.json unreadable: %r" % (exc,)


def roofline_of(wl: Workload, kernel_ms):
    """achieved = algorithmic bytes per launch / the MEDIAN launch duration (HIP events on the launch stream around every
    timed launch); min / median / max are in the line so that a slow box can be told from a slow build"""
    ms = sorted(kernel_ms)
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    achieved = wl.algo_bytes_per_launch / (med / 1e3) / 1e9
    traffic, src = pmc_traffic(wl.name, wl.chunk)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src, "kernel": wl.kernel,
            "algorithmic_bytes_per_launch": wl.algo_bytes_per_launch, "input_bytes_per_launch": wl.read_bytes,
            "launch_ms": {"min": ms[0], "median": med, "max": ms[-1]},
            "avg_launch_ms": sum(ms) / len(ms), "launches_timed": len(ms)}


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus and not (world == 1 and os.environ.get("PG_FORCE_DIST") == "1"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it with --nproc-per-node {args.gpus} "
                         f"(or plainly, and it starts the ranks itself)")

    # stdout carries ONE JSON line: libraries that talk on it (RCCL prints its version banner there when a communicator is
    # created) are sent to stderr -- fd 1 is pointed at fd 2, and the line goes to a duplicate of the real stdout
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(out_fd, (json.dumps(obj) + "\n").encode())

    emit_lock = threading.Lock()
    emitted = []

    def emit_once(obj):
        """True if this call wrote the line (exactly one line is ever written)"""
        with emit_lock:
            if emitted:
                return False
            emitted.append(True)
            emit(obj)
            return True

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PG_FORCE_DIST=1: take the multi-rank code path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)
    distributed = world > 1 or os.environ.get("PG_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # PG_DIST_BACKEND=gloo is a rehearsal aid (several ranks sharing one GPU, where RCCL refuses duplicate devices)
    backend = os.environ.get("PG_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if ngpu < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (the engine has no CPU path)")
    if backend == "nccl" and world > ngpu:
        raise SystemExit(f"bench.py --gpus {world}: needs {world} GPUs, this box has {ngpu}")
    local_rank %= ngpu
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    import plonk_gadgets_amd as pg

    eng = pg.Engine(local_rank)
    free, total = torch.cuda.mem_get_info(dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x: float) -> float:
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- N > 1: what this rank will hold, said before anything is allocated ---------------------------------------------
    # the gather-inclusive sample runs with the native collective (pg_comm: RCCL, or whatever PG_RCCL_LIB names -- the test-only
    # stand-in of the one-GPU rehearsals, with torch.distributed on gloo for the rendezvous)
    native_gather = distributed and args.workload == "c2" and args.allgather_chunks > 0 and (
        backend == "nccl" or bool(os.environ.get("PG_RCCL_LIB")))
    hbm_budget = None
    if distributed:
        per_item = 1031 * BYTES_PER_GATE + 1034 * BYTES_PER_VAR
        sharing = max(1, -(-world // ngpu)) if backend != "nccl" else 1  # (a rehearsal: several ranks on one card)
        gchunk_bytes = (1 << args.allgather_log2_chunk) * per_item
        want_cols = (1 << args.log2_batch) * (per_item + 40) if args.log2_chunk < 0 else (1 << min(args.log2_batch, args.log2_chunk)) * (per_item + 40)
        cols_bytes = min(want_cols, int(0.85 * free / sharing))  # (the workload halves its launch until it fits)
        # two slots, each the rank's own packed chunk and every rank's gathered chunk; the variables-only form regenerates the
        # other ranks' rows into the same slots
        pipe_bytes = 2 * (world + 1) * gchunk_bytes if native_gather else 0
        hbm_budget = {"columns_bytes": cols_bytes, "gather_pipeline_bytes": pipe_bytes, "free_bytes": free, "ranks_sharing_the_card": sharing,
                      "note": "per rank: the step's nine arrays, then (after they are released) two slots x (world + 1) packed chunks"}
        if rank == 0:
            print("bench.py rank budget: columns %.1f GB, gather pipeline %.2f GB (2 slots x %d chunks of %.2f GB), %.1f GB free, %d rank(s) "
                  "on this card" % (cols_bytes / 1e9, pipe_bytes / 1e9, world + 1, gchunk_bytes / 1e9, free / 1e9, sharing), file=sys.stderr)
        if pipe_bytes * sharing > 0.9 * free:
            raise SystemExit("bench.py --gpus %d: the gather pipeline alone needs %.1f GB per rank (2 slots x %d chunks of 2^%d witnesses) and "
                             "%.1f GB are free for %d rank(s): lower --allgather-log2-chunk" % (
                                 world, pipe_bytes / 1e9, world + 1, args.allgather_log2_chunk, free / 1e9, sharing))

    # ---- headline ------------------------------------------------------------------------------------------
    wl = Workload(args.workload, eng, dev, rank, world, args.log2_batch, args.log2_chunk)
    elapsed, kernel_ms = measure(wl, args.steps, args.warmup, sync_all)
    own_elapsed = elapsed  # (this rank's own K steps: at N > 1 rank 0's stands in for an N = 1 run that was not given)
    elapsed = max_over_ranks(elapsed)
    if os.environ.get("PG_BENCH_VERBOSE") and rank == 0:
        print("launch ms:", " ".join("%.2f" % t for t in kernel_ms), file=sys.stderr)
    constraints = world * wl.rows_per_launch * wl.n_chunks * args.steps
    value = constraints / elapsed
    # a headline whose step is short (--workload c3: 0.5 ms) has had W x 0.5 ms of warm-up: the card is still on its way up from the idle of
    # the set-up (profiles/NOTES_r05.md section 5).  The contract's figure stands as measured; the same K steps after 180 ms of work
    # (what W = 5 steps of the default headline last) are reported beside it
    steady = None
    if world == 1 and elapsed / args.steps < 5e-3 and not args.no_secondary:
        warm = max(args.warmup, min(2000, int(math.ceil(0.18 / (elapsed / args.steps)))))
        el_s, ms_s = measure(wl, args.steps, warm, sync_all)
        steady = {"warmup": warm, "warmup_note": "180 ms of work, as long as five steps of the default headline (c2)",
                  "ms_per_step": el_s / args.steps * 1e3, "value": constraints / el_s, "frac": roofline_of(wl, ms_s)["frac"]}
    roofline = roofline_with_fill(wl, kernel_ms) if world == 1 and not args.no_fill else roofline_of(wl, kernel_ms)
    config = {"workload": wl.desc, "items_per_gpu": wl.batch, "items_per_launch": wl.chunk,
              "launches_per_step": wl.n_chunks,
              "sharding": "contiguous witness ranges per rank at global numbering, no data-path collective"}
    if args.workload in ("c3", "c4"):
        config.update(wl.layout_note())
    wit = getattr(wl, "wit", None)
    mn, mx = getattr(wl, "mn", None), getattr(wl, "mx", None)
    wl.release()
    torch.cuda.empty_cache()
    headline_step_s = elapsed / args.steps

    # ---- N = 1: the other single-GPU BASELINE configs, same process, same steps ------------------------------
    secondary = None
    if world == 1 and args.workload == "c2" and not args.no_secondary:
        secondary = {}
        for name in ("c3", "c4", "c2_values"):
            try:
                w2 = Workload(name, eng, dev, rank, world, args.log2_batch, -1)
                el2, ms2 = measure(w2, args.steps, args.warmup, sync_all)
                # W steps of a 0.5-ms workload are 2.5 ms: the card is still on its way up from the idle of the set-up (C3: 0.54 ms
                # per launch falling to 0.50 over the first 30, profiles/NOTES_r05.md section 5).  The headline's W steps last
                # W x 36 ms; a secondary workload gets the same warm-up TIME, and its figure after W steps alone stays beside it
                first = None
                warm = max(args.warmup, min(2000, int(math.ceil(args.warmup * headline_step_s / max(el2 / args.steps, 1e-6)))))
                if warm > args.warmup:
                    r_first = roofline_of(w2, ms2)
                    first = {"warmup": args.warmup, "ms_per_step": el2 / args.steps * 1e3, "frac": r_first["frac"],
                             "launch_ms": r_first["launch_ms"]}
                    el2, ms2 = measure(w2, args.steps, warm, sync_all)
                secondary[name] = {"metric": f"gadget constraints/sec ({name})" if name != "c2_values" else
                                             "constraints whose witnesses are refreshed /sec (range_check 256-bit, values only)",
                                   "value": w2.rows_per_launch * w2.n_chunks * args.steps / el2, "unit": "constraints/s",
                                   "ms_per_step": el2 / args.steps * 1e3, "steps": args.steps, "warmup": warm,
                                   **({"warmup_note": "as long as the headline's %d steps (%.0f ms)" % (args.warmup, args.warmup * headline_step_s * 1e3),
                                       "after_the_headlines_step_count_only": first} if first else {}),
                                   "config": {"workload": w2.desc, "items_per_gpu": w2.batch, "items_per_launch": w2.chunk,
                                              **(w2.layout_note() if name in ("c3", "c4") else {})},
                                   "roofline": roofline_with_fill(w2, ms2)}
                if name in ("c3", "c4"):
                    # the same launches into the OTHER layout: nine plain allocations, wherever the driver puts them, against one slab
                    # with the selector columns tens of GiB apart (the placement effect, driver-observed: DESIGN.md section 2)
                    slab_first = bool(getattr(w2, "spread_gib", 0))
                    w2.reallocate_columns(0 if slab_first else (24 if name == "c3" else 32))
                    if slab_first or w2.spread_gib:  # (a card without the room for the slab: nothing to compare)
                        el3, ms3 = measure(w2, args.steps, warm, sync_all)
                        r3 = roofline_with_fill(w2, ms3)
                        secondary[name]["nine_allocations" if slab_first else "one_slab"] = {
                            "ms_per_step": el3 / args.steps * 1e3, "value": w2.rows_per_launch * w2.n_chunks * args.steps / el3,
                            "frac": r3["achieved"] / HBM_PEAK_GBPS, "achieved": r3["achieved"], "launch_ms": r3["launch_ms"],
                            "bare_fill": r3.get("bare_fill"), "frac_of_bare_fill": r3.get("frac_of_bare_fill"), **w2.layout_note()}
                w2.release()
                del w2
                torch.cuda.empty_cache()
            except Exception as ex:  # a secondary figure must never cost the headline line
                secondary[name] = {"error": repr(ex)}
        try:  # the rows that come next: materialisation and sigma on a 270 M-row composer
            secondary["next_rows"] = next_rows_secondary(eng, dev, min(18, args.log2_batch), args.steps)
        except Exception as ex:
            secondary["next_rows"] = {"error": repr(ex)}

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
        "config": config,
        **({"after_180_ms_of_warm_up": steady} if steady else {}),
        "roofline": roofline,
        "hbm_free_gb_at_start": free / 1e9, "hbm_total_gb": total / 1e9,
    }
    if hbm_budget:
        line["hbm_budget"] = hbm_budget
    if secondary:
        line["secondary"] = secondary

    # ---- N > 1: gather-inclusive rate of the chunked all-gather pipeline (bounded sample) -----------------
    allgather = None
    watchdog = None
    finished = threading.Event()
    if native_gather:
        # the headline is measured; a collective that hangs (a link, a rank that died) must not cost it: after the limit
        # rank 0 prints the line it has, and every rank leaves
        def bail():
            # exactly one line is ever written (emit_once), and a run whose collective hung does not exit 0: code 3 = "the
            # headline was measured and printed, the gather-inclusive sample was abandoned".  A timer that fires when the
            # run has in fact finished (the complete line is out, or about to be) changes nothing.
            if finished.is_set():
                return
            if rank == 0 and not emit_once(dict(line, allgather={"error": f"the gather-inclusive sample did not finish within "
                                                                          f"{args.allgather_timeout} s; abandoned"})):
                return  # the complete line was written first
            os._exit(3)
        watchdog = threading.Timer(args.allgather_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        from plonk_gadgets_amd import distributed as pd
        G = 1031
        gchunk = 1 << args.allgather_log2_chunk
        per_rank = gchunk * args.allgather_chunks

        def timed(pipe):
            pipe.run(wit[:2 * gchunk], 2 * gchunk)  # warm-up (communicator set-up)
            sync_all()
            t1 = time.perf_counter()
            pipe.run(wit[:per_rank], per_rank)
            sync_all()
            return max_over_ranks(time.perf_counter() - t1)
        # (gloo rendezvous + a library named by PG_RCCL_LIB: the pipelines are told to use the native collective themselves)
        coll = None if backend == "nccl" else pd.NativeCollective(eng)
        try:
            pipe = pd.GatherPipeline(eng, mn, mx, gchunk, collective=coll)
            dt = timed(pipe)
            allgather = {"value": world * per_rank * G / dt, "unit": "constraints/s (every rank ends with every shard)",
                         "witnesses_per_rank": per_rank, "witnesses_per_chunk": gchunk,
                         "bytes_per_rank_per_chunk": pipe.bytes_per_chunk(),
                         "ingest_gbps_per_gpu": (world - 1) * pipe.bytes_per_chunk() * args.allgather_chunks / dt / 1e9,
                         "collective": pipe.collective_name() + ", one per packed chunk, double-buffered"}
            del pipe
            torch.cuda.empty_cache()
        except Exception as ex:
            allgather = {"error": repr(ex)}
        # the same stream with only the variable tables on the links; the other ranks' rows are regenerated locally
        try:
            vpipe = pd.VariablesOnlyPipeline(eng, mn, mx, gchunk, collective=coll)
            dtv = timed(vpipe)
            allgather["variables_only"] = {
                "value": world * per_rank * G / dtv, "unit": "constraints/s (every rank ends with every shard)",
                "bytes_per_rank_per_chunk": vpipe.bytes_on_the_links_per_chunk(),
                "ingest_gbps_per_gpu": (world - 1) * vpipe.bytes_on_the_links_per_chunk() * args.allgather_chunks / dtv / 1e9,
                "collective": vpipe.collective_name() + " of the variable tables per chunk; selectors and wire indices of "
                              "the other ranks' chunks regenerated locally (pg_range_check_structure_batch)"}
            del vpipe
        except Exception as ex:
            allgather["variables_only"] = {"error": repr(ex)}

    finished.set()
    if watchdog is not None:
        watchdog.cancel()  # (a timer that fired before `finished` was set wins the lock below and exits with the short line)
    if rank == 0:
        final = dict(line)
        if allgather:
            final["allgather"] = allgather
        if world > 1:
            # SURVEY 8(e) asks for both curves: generation only (no data-path collective: by construction close to 1) and the
            # gather-inclusive one (bound by xGMI's ingest, the figure that matters).  Each against N x the N = 1 value: the one given
            # with --n1-value (the driver's N = 1 run of this same command), else this run's own rank-0 rate
            n1 = args.n1_value if args.n1_value > 0 else wl.rows_per_launch * wl.n_chunks * args.steps / own_elapsed
            final["n1_value"] = {"value": n1, "source": "--n1-value" if args.n1_value > 0 else "this run: rank 0's own steps, not the maximum over ranks"}
            final["efficiency_generation"] = value / (world * n1)
            if allgather and "value" in allgather:
                final["efficiency_gather_inclusive"] = allgather["value"] / (world * n1)
                if isinstance(allgather.get("variables_only"), dict) and "value" in allgather["variables_only"]:
                    final["efficiency_gather_inclusive_variables_only"] = allgather["variables_only"]["value"] / (world * n1)
            else:
                final["efficiency_gather_inclusive"] = None
        if world == 1 and not args.no_cpu:
            final["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample)
        emit_once(final)
    if distributed:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
