"""Multi-GPU: shard a batch of independent witnesses over ranks (one process per GPU).

The path shards naturally: items are independent and, for uniform gadgets, the global
numbering is closed-form -- item i owns rows [gate_base + i*G, +G) and variables [var_base + i*V, +V) -- so every
rank emits its contiguous witness range straight at its final global indices with NO exchange.  The only
collective is the optional all-gather that gives every rank every shard's columns:

  * gather_columns():      the nine arrays of equal shards in one grouped launch -> globally contiguous columns
                           (rank order == witness order);
  * GatherPipeline:        the streaming form for batches that do not fit (2^23 x 223 KB = 1.87 TB): chunks are
                           emitted into ONE packed buffer (all 9 arrays back to back) so that each chunk is a single
                           all-gather, double-buffered so chunk k+1 is generated while chunk k is on the links;
  * VariablesOnlyPipeline: only the variable tables travel; the other ranks' rows are regenerated locally.

Everything here is a thin host layer over the C ABI (include/plonk_gadgets_hip.h, "multi-GPU"): the shard arithmetic
(pg_shard_range, pg_range_check_shard_layout), the sharded emission (pg_range_check_sharded_batch), the packed layout
(pg_packed_layout) and the collective itself (pg_comm_* / pg_allgather_*: RCCL called from the library on the caller's
stream) are the entry points a Rust host would bind; torch.distributed is used for the rendezvous (handing rank 0's
communicator id to the others, barriers) and, with the "gloo" backend on CPU, as the stand-in collective of the tests.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a full gather makes every GPU ingest (N-1)/N of all bytes,
i.e. <= ~1.07 TB/s per GPU, well below one GPU's ~6.9 TB/s emission rate -- so the gather, not HBM, bounds a
gather-inclusive pipeline; bench.py reports both rates.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
import torch.distributed as dist

from . import _lib
from .engine import Columns, PgError


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous witness range [lo, hi) of `rank`; the first (total % world) ranks get one extra item
    (pg_shard_range)"""
    lo, hi = C.c_uint64(), C.c_uint64()
    st = _lib.load().pg_shard_range(total, rank, world, C.byref(lo), C.byref(hi))
    if st != 0:
        raise PgError(st, "pg_shard_range")
    return int(lo.value), int(hi.value)


@dataclass
class ShardInfo:
    rank: int
    world: int
    lo: int
    hi: int
    gate_base: int  # global index of this shard's first row
    var_base: int   # global index of this shard's first variable
    n_gates: int = 0
    n_vars: int = 0


def _rank_world(group=None):
    return (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)


def range_check_shard_layout(min_range, max_range, total: int, rank: int, world: int, gate_base: int = 0,
                             var_base: int = 0) -> ShardInfo:
    """placement of rank's shard of a `total`-item range_check batch (pg_range_check_shard_layout; host arithmetic)"""
    s = _lib.ShardC()
    st = _lib.load().pg_range_check_shard_layout(C.byref(min_range.c), C.byref(max_range.c), total, rank, world, gate_base,
                                                 var_base, C.byref(s))
    if st != 0:
        raise PgError(st, "pg_range_check_shard_layout")
    return ShardInfo(s.rank, s.world, s.lo, s.hi, s.gate_base, s.var_base, s.n_gates, s.n_vars)


def range_check_sharded(engine, min_range, max_range, witness_local: torch.Tensor, total: int, gate_base: int = 0,
                        var_base: int = 0, group=None, out: Columns | None = None):
    """Emit this rank's shard of `total` range_check items at its global numbering.  `witness_local` holds the
    witnesses of shard_range(total, rank, world).  No communication.  Returns (Columns, result_vars, ShardInfo)."""
    rank, world = _rank_world(group)
    info = range_check_shard_layout(min_range, max_range, total, rank, world, gate_base, var_base)
    assert witness_local.shape[0] == info.hi - info.lo, (witness_local.shape, info.lo, info.hi)
    if hasattr(engine, "range_check_sharded_batch"):  # the gfx950 engine: pg_range_check_sharded_batch
        cols, res = engine.range_check_sharded_batch(min_range, max_range, witness_local, total, rank, world, gate_base,
                                                     var_base, out=out)
    else:  # a stand-in engine (CPU tests): same placement, its own emission
        cols, res = engine.range_check_batch(min_range, max_range, witness_local, info.gate_base, info.var_base, out=out)
    return cols, res, info


# ---- the collective ----------------------------------------------------------------------------------------------

class TorchCollective:
    """torch.distributed's all_gather_into_tensor (RCCL through ProcessGroupNCCL, or gloo on CPU)"""
    name = "torch.distributed.all_gather_into_tensor"

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = _rank_world(group)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        """returns an object with .wait(): the current stream is then ordered after the collective"""
        if not dist.is_initialized():
            out.view(1, -1)[0].copy_(inp.view(-1))
            return _Done()
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def close(self):
        pass


class NativeCollective:
    """pg_comm: RCCL called by the library itself (the path a host without torch takes).  The communicator id travels
    over torch.distributed's rendezvous when one is initialised; world = 1 needs none.  Collectives run on a
    communication stream of their own, ordered after the emission by an event, so that chunk k+1 is generated while
    chunk k is on the links."""
    name = "pg_allgather_bytes (ncclAllGather called by libplonk_gadgets_hip on a communication stream)"

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group
        self.rank, self.world = _rank_world(group)
        lib = _lib.load()
        self._lib = lib
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            st = lib.pg_comm_unique_id(ident)
            if st != 0:
                raise PgError(st, "pg_comm_unique_id")
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        st = lib.pg_comm_create(engine._h, ident, self.rank, self.world, C.byref(h))
        if st != 0:
            raise PgError(st, "pg_comm_create")
        self._h = h
        assert lib.pg_comm_world(h) == self.world and lib.pg_comm_rank(h) == self.rank
        self.stream = torch.cuda.Stream(device=engine.device)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        assert out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()
        nbytes = inp.numel() * inp.element_size()
        assert out.numel() * out.element_size() == nbytes * self.world
        cur = torch.cuda.current_stream(self.engine.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.stream.wait_event(ready)  # the collective reads what the compute stream has emitted
        st = self._lib.pg_allgather_bytes(self._h, inp.data_ptr(), out.data_ptr(), nbytes, C.c_void_p(self.stream.cuda_stream))
        if st != 0:
            raise PgError(st, "pg_allgather_bytes")
        inp.record_stream(self.stream)
        out.record_stream(self.stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return _EventWork(done, self.engine.device)

    def all_gather_columns(self, local: Columns, gathered: Columns):
        """the nine arrays of equal shards, one grouped launch on the current stream (pg_allgather_columns)"""
        lc, gc = local.as_c(), gathered.as_c()
        st = self._lib.pg_allgather_columns(self._h, C.byref(lc), local.q_m.shape[0], local.var_values.shape[0], C.byref(gc),
                                            self.engine._stream())
        if st != 0:
            raise PgError(st, "pg_allgather_columns")

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.engine.device)
            self._lib.pg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _EventWork:
    def __init__(self, ev, device):
        self.ev, self.device = ev, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.ev)
        return True


class _Done:
    def wait(self):
        return True


def default_collective(engine, group=None, native: bool | None = None):
    """native (pg_comm) for the gfx950 engine when the ranks each own a GPU (backend nccl, or a single rank); the
    torch collective otherwise (gloo rehearsals, the CPU stand-in engine of the tests)"""
    is_gpu_engine = hasattr(engine, "_h") and getattr(engine, "device", torch.device("cpu")).type == "cuda"
    if native is None:
        native = is_gpu_engine and (not dist.is_initialized() or dist.get_backend(group) == "nccl")
    return NativeCollective(engine, group) if native else TorchCollective(group)


def exchange_totals(n_gates: int, n_vars: int, device, group=None):
    """the one real exchange of a ragged sharded batch: every rank learns every shard's (rows, variables) totals,
    16 bytes per rank, and derives its bases by an exclusive prefix sum.  Returns (gates_per_rank, vars_per_rank)."""
    if not dist.is_initialized():
        return [n_gates], [n_vars]
    world = dist.get_world_size(group)
    mine = torch.tensor([n_gates, n_vars], dtype=torch.int64, device=device)
    allt = torch.empty((world * 2,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allt, mine, group=group)
    allt = allt.view(world, 2).cpu()
    return [int(x) for x in allt[:, 0]], [int(x) for x in allt[:, 1]]


def max_bound_ragged_sharded(engine, max_range_local: torch.Tensor, witness_local: torch.Tensor, gate_base: int = 0,
                             var_base: int = 0, group=None, collective=None):
    """Ragged max_bound (one public bound per item) over ranks: every rank plans its contiguous shard, the shard
    totals are all-gathered (exchange_totals), and each rank emits at the global numbering its prefix gives it.
    Returns (Columns, result_vars, ShardInfo, gates_per_rank, vars_per_rank)."""
    rank, world = _rank_world(group)
    batch = witness_local.shape[0]
    nb, roff, voff = engine.ragged_buffers(batch)
    if isinstance(collective, NativeCollective):  # plan + exchange inside the library (pg_max_bound_ragged_sharded_plan)
        lib, s = _lib.load(), _lib.ShardC()
        gates, vars_ = (C.c_uint64 * world)(), (C.c_uint64 * world)()
        st = lib.pg_max_bound_ragged_sharded_plan(collective._h, max_range_local.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(),
                                                  voff.data_ptr(), gate_base, var_base, C.byref(s), gates, vars_, engine._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_sharded_plan")
        cols = Columns.allocate(s.n_gates, s.n_vars, witness_local.device, s.gate_base, s.var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=witness_local.device)
        engine.max_bound_ragged_emit(max_range_local, witness_local, nb, roff, voff, cols, res, s.gate_base, s.var_base)
        return cols, res, ShardInfo(rank, world, -1, -1, s.gate_base, s.var_base, s.n_gates, s.n_vars), list(gates), list(vars_)
    lay = engine.max_bound_ragged_plan(max_range_local, nb, roff, voff)
    gates, vars_ = exchange_totals(lay.n_gates, lay.n_vars, witness_local.device, group)
    g0, v0 = gate_base + sum(gates[:rank]), var_base + sum(vars_[:rank])
    cols = Columns.allocate(lay.n_gates, lay.n_vars, witness_local.device, g0, v0)
    res = torch.empty((batch,), dtype=torch.int64, device=witness_local.device)
    engine.max_bound_ragged_emit(max_range_local, witness_local, nb, roff, voff, cols, res, g0, v0)
    return cols, res, ShardInfo(rank, world, -1, -1, g0, v0, lay.n_gates, lay.n_vars), gates, vars_


def _gather_1d(local: torch.Tensor, counts: list[int], group) -> torch.Tensor:
    """all-gather of per-rank tensors whose leading sizes are `counts` (equal -> one all_gather_into_tensor;
    ragged -> pad to the maximum, gather, strip)"""
    world = len(counts)
    tail = tuple(local.shape[1:])
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(counts)
    padded = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.view((world, mx) + tail)
    return torch.cat([out[r, :counts[r]] for r in range(world)], dim=0)


def gather_columns(cols: Columns, result_vars: torch.Tensor | None, gates_per_rank: list[int], vars_per_rank: list[int],
                   group=None, collective=None):
    """all-gather every column so each rank holds the whole batch's columns, contiguous in witness order.  With a
    NativeCollective and equal shards: ONE grouped RCCL launch (pg_allgather_columns)."""
    world = len(gates_per_rank)
    if isinstance(collective, NativeCollective) and len(set(gates_per_rank)) == 1 and len(set(vars_per_rank)) == 1:
        full = Columns.allocate(world * gates_per_rank[0], world * vars_per_rank[0], cols.q_m.device)
        collective.all_gather_columns(cols, full)
        res = None
        if result_vars is not None:
            res = torch.empty((world * result_vars.shape[0],), dtype=torch.int64, device=result_vars.device)
            collective.all_gather(res, result_vars).wait()
        return full, res
    g = {}
    for name in Columns.SCALAR_COLS + Columns.WIRE_COLS:
        g[name] = _gather_1d(getattr(cols, name), gates_per_rank, group)
    g["var_values"] = _gather_1d(cols.var_values, vars_per_rank, group)
    full = Columns(g["q_m"], g["q_l"], g["q_r"], g["q_o"], g["q_c"], g["w_l"], g["w_r"], g["w_o"], g["var_values"])
    res = None
    if result_vars is not None:
        counts = [torch.zeros(1, dtype=torch.int64, device=result_vars.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([result_vars.shape[0]], dtype=torch.int64, device=result_vars.device),
                        group=group)
        res = _gather_1d(result_vars, [int(c.item()) for c in counts], group)
    return full, res


# ---- packed chunks: one collective per chunk -----------------------------------------

def packed_layout(n_gates: int, n_vars: int):
    """offsets (in int64 words) of the 9 arrays inside one packed chunk buffer; every section starts on a 128-byte line
    (pg_packed_layout)"""
    p = _lib.PackedC()
    st = _lib.load().pg_packed_layout(n_gates, n_vars, C.byref(p))
    if st != 0:
        raise PgError(st, "pg_packed_layout")
    off, sizes = {}, {}
    for i, name in enumerate(Columns.SCALAR_COLS):
        off[name], sizes[name] = int(p.q_words[i]), n_gates * 4
    for i, name in enumerate(Columns.WIRE_COLS):
        off[name], sizes[name] = int(p.w_words[i]), n_gates
    off["var_values"], sizes["var_values"] = int(p.var_words), n_vars * 4
    return off, sizes, int(p.total_words)


def columns_in(flat: torch.Tensor, n_gates: int, n_vars: int) -> Columns:
    """Columns whose arrays are views into the packed buffer `flat` (int64[packed words])"""
    off, sizes, total = packed_layout(n_gates, n_vars)
    assert flat.numel() >= total and flat.dtype == torch.int64
    v = {k: flat[off[k]:off[k] + sizes[k]] for k in off}
    return Columns(*[v[k].view(n_gates, 4) for k in Columns.SCALAR_COLS], *[v[k] for k in Columns.WIRE_COLS],
                   v["var_values"].view(n_vars, 4))


class _DevMem:
    """device memory owned by the library, seen by torch through __cuda_array_interface__"""

    def __init__(self, ptr: int, nwords: int):
        self.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def _dev_words(ptr: int, nwords: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevMem(ptr, nwords), device=device)


class _NativePipeline:
    """pg_range_check_gather_pipeline_*: the whole double-buffered emit-while-gather loop runs inside the library (the path
    a host without torch takes -- examples/c5_rank.c); this class only hands it the witnesses and turns the chunks it
    delivers (device pointers) into tensors for a Python consumer."""

    def __init__(self, engine, coll: "NativeCollective", min_range, max_range, chunk: int, variables_only: bool):
        self.engine, self.coll, self.chunk = engine, coll, chunk
        self._lib = _lib.load()
        h = C.c_void_p()
        st = self._lib.pg_range_check_gather_pipeline_create(coll._h, C.byref(min_range.c), C.byref(max_range.c), chunk,
                                                             1 if variables_only else 0, C.byref(h))
        if st != 0:
            raise PgError(st, "pg_range_check_gather_pipeline_create")
        self._h = h

    def bytes_per_chunk(self) -> int:
        return int(self._lib.pg_range_check_gather_pipeline_bytes_per_chunk(self._h))

    def run(self, witness_local: torch.Tensor, total_per_rank: int, gate_base: int, var_base: int, on_chunk):
        """on_chunk(parts: list of (ColumnsC, n_gates, n_vars) per rank, chunk_index) is called from inside the run"""
        failure = []

        def cb(_user, k, world, parts, n_gates, n_vars, _stream):
            if failure or on_chunk is None:
                return
            try:
                on_chunk([parts[r] for r in range(world)], int(n_gates), int(n_vars), int(k))
            except BaseException as ex:  # an exception must not unwind through the C frames
                failure.append(ex)

        fn = _lib.CHUNK_CONSUMER(cb)
        st = self._lib.pg_range_check_gather_pipeline_run(self._h, witness_local.data_ptr(), total_per_rank, gate_base, var_base,
                                                          fn, None, self.engine._stream())
        if failure:
            raise failure[0]
        if st != 0:
            raise PgError(st, "pg_range_check_gather_pipeline_run")

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.engine.device)
            self._lib.pg_range_check_gather_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _columns_from_c(pc, n_gates: int, n_vars: int, device) -> Columns:
    sc = [_dev_words(getattr(pc, n), n_gates * 4, device).view(n_gates, 4) for n in Columns.SCALAR_COLS]
    wc = [_dev_words(getattr(pc, n), n_gates, device) for n in Columns.WIRE_COLS]
    return Columns(*sc, *wc, _dev_words(pc.var_values, n_vars * 4, device).view(n_vars, 4))


class GatherPipeline:
    """Streaming sharded range_check with a single all-gather per chunk (packed buffer), double-buffered:
    while chunk k is on the links, chunk k+1 is being emitted.  `consume(gathered, chunk_index)` sees
    gathered[r] = rank r's packed chunk (use columns_in to view it)."""

    def __init__(self, engine, min_range, max_range, chunk: int, group=None, collective=None):
        self.engine, self.mn, self.mx, self.chunk, self.group = engine, min_range, max_range, chunk, group
        self.coll = collective if collective is not None else default_collective(engine, group)
        self.world, self.rank = self.coll.world, self.coll.rank
        self.lay = engine.range_check_layout(min_range, max_range, chunk)
        _, _, self.words = packed_layout(self.lay.n_gates, self.lay.n_vars)
        dev = engine.device
        self.native = None
        if isinstance(self.coll, NativeCollective):  # the loop itself is the library's (pg_range_check_gather_pipeline_run)
            self.native = _NativePipeline(engine, self.coll, min_range, max_range, chunk, variables_only=False)
            return
        self.local = [torch.empty(self.words, dtype=torch.int64, device=dev) for _ in range(2)]
        self._gflat = [torch.empty(self.world * self.words, dtype=torch.int64, device=dev) for _ in range(2)]
        self.gathered = [g.view(self.world, self.words) for g in self._gflat]
        self.cols = [columns_in(b, self.lay.n_gates, self.lay.n_vars) for b in self.local]
        self.res = torch.empty((chunk,), dtype=torch.int64, device=dev)

    def bytes_per_chunk(self) -> int:
        return self.words * 8

    def collective_name(self) -> str:
        return self.coll.name

    def run(self, witness_local: torch.Tensor, total_per_rank: int, gate_base: int = 0, var_base: int = 0, consume=None):
        """witness_local: this rank's total_per_rank witnesses (a multiple of chunk).  Global numbering: rank r's
        item i is item r*total_per_rank + i of the whole batch."""
        assert total_per_rank % self.chunk == 0
        if self.native is not None:
            def on_chunk(parts, n_gates, n_vars, k):
                # rank r's packed chunk starts at its q_m (the first section of the packed layout)
                flat = _dev_words(parts[0].q_m, self.world * self.words, self.engine.device)
                consume(flat.view(self.world, self.words), k)
            self.native.run(witness_local, total_per_rank, gate_base, var_base, on_chunk if consume is not None else None)
            return
        G, V = self.lay.gates_per_item, self.lay.vars_per_item
        pending = None
        for k in range(total_per_rank // self.chunk):
            b = k & 1
            first = self.rank * total_per_rank + k * self.chunk
            self.engine.range_check_batch(self.mn, self.mx, witness_local[k * self.chunk:(k + 1) * self.chunk],
                                          gate_base + first * G, var_base + first * V, out=self.cols[b],
                                          result_vars=self.res)
            if pending is not None:
                work, kb, kk = pending
                work.wait()
                if consume is not None:
                    consume(self.gathered[kb], kk)
            # ordered after the emission (torch's ProcessGroup / the collective's own event); chunk k+1 is emitted
            # while this one travels
            work = self.coll.all_gather(self._gflat[b], self.local[b])
            pending = (work, b, k)
        if pending is not None:
            work, kb, kk = pending
            work.wait()
            if consume is not None:
                consume(self.gathered[kb], kk)


class VariablesOnlyPipeline:
    """The same stream of chunks as GatherPipeline, but only what depends on the witnesses travels (SURVEY.md section
    8e): a rank emits its own chunk in full, regenerates the selectors and wire indices of the other ranks' chunks
    locally (engine.range_check_structure_batch: a function of the public bounds and the numbering alone) and
    all-gathers just the variable tables -- 32 B per variable instead of 184 B per row + 32 B per variable, i.e. 33 KB
    of the 223 KB a 256-bit range_check item weighs.  Per chunk and rank: (world - 1) extra structure launches (local
    HBM writes at ~7 TB/s) against a 6.7 x smaller transfer over the ~1 TB/s a GPU can ingest from its xGMI links.

    `consume(parts, chunk_index)`: parts[r] is a Columns view of rank r's chunk (all nine arrays complete)."""

    def __init__(self, engine, min_range, max_range, chunk: int, group=None, collective=None):
        self.engine, self.mn, self.mx, self.chunk, self.group = engine, min_range, max_range, chunk, group
        self.coll = collective if collective is not None else default_collective(engine, group)
        self.world, self.rank = self.coll.world, self.coll.rank
        self.lay = engine.range_check_layout(min_range, max_range, chunk)
        G, V = self.lay.n_gates, self.lay.n_vars
        dev = engine.device
        self.native = None
        if isinstance(self.coll, NativeCollective):  # the loop itself is the library's (pg_range_check_gather_pipeline_run)
            self.native = _NativePipeline(engine, self.coll, min_range, max_range, chunk, variables_only=True)
            return
        # per double-buffer slot: rows of every rank's chunk, one gathered variable table, this rank's own table
        self.rows = [[Columns.allocate(G, 0, dev) for _ in range(self.world)] for _ in range(2)]
        self._vflat = [torch.empty(self.world * V * 4, dtype=torch.int64, device=dev) for _ in range(2)]
        self.vars = [v.view(self.world, V, 4) for v in self._vflat]
        self.own = [torch.empty((V, 4), dtype=torch.int64, device=dev) for _ in range(2)]
        self.res = torch.empty((chunk,), dtype=torch.int64, device=dev)

    def bytes_on_the_links_per_chunk(self) -> int:
        return self.lay.n_vars * 32

    def collective_name(self) -> str:
        return self.coll.name

    def _parts(self, b):
        out = []
        for r in range(self.world):
            c = self.rows[b][r]
            out.append(Columns(c.q_m, c.q_l, c.q_r, c.q_o, c.q_c, c.w_l, c.w_r, c.w_o, self.vars[b][r]))
        return out

    def run(self, witness_local: torch.Tensor, total_per_rank: int, gate_base: int = 0, var_base: int = 0, consume=None):
        """numbering as in GatherPipeline.run: rank r's item i is item r * total_per_rank + i of the whole batch"""
        assert total_per_rank % self.chunk == 0
        if self.native is not None:
            def on_chunk(parts, n_gates, n_vars, k):
                consume([_columns_from_c(pc, n_gates, n_vars, self.engine.device) for pc in parts], k)
            self.native.run(witness_local, total_per_rank, gate_base, var_base, on_chunk if consume is not None else None)
            return
        G, V = self.lay.gates_per_item, self.lay.vars_per_item
        pending = None
        for k in range(total_per_rank // self.chunk):
            b = k & 1
            for r in range(self.world):
                first = r * total_per_rank + k * self.chunk
                c = self.rows[b][r]
                if r == self.rank:
                    mine = Columns(c.q_m, c.q_l, c.q_r, c.q_o, c.q_c, c.w_l, c.w_r, c.w_o, self.own[b])
                    self.engine.range_check_batch(self.mn, self.mx, witness_local[k * self.chunk:(k + 1) * self.chunk],
                                                  gate_base + first * G, var_base + first * V, out=mine, result_vars=self.res)
                else:
                    self.engine.range_check_structure_batch(self.mn, self.mx, self.chunk, gate_base + first * G,
                                                            var_base + first * V, c)
            if pending is not None:
                work, kb, kk = pending
                work.wait()
                if consume is not None:
                    consume(self._parts(kb), kk)
            work = self.coll.all_gather(self._vflat[b], self.own[b].view(-1))
            pending = (work, b, k)
        if pending is not None:
            work, kb, kk = pending
            work.wait()
            if consume is not None:
                consume(self._parts(kb), kk)
