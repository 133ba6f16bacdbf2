"""Multi-GPU: shard a batch of independent witnesses over ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for tests).

The path shards naturally: items are independent and, for uniform gadgets, the global
numbering is closed-form -- item i owns rows [gate_base + i*G, +G) and variables [var_base + i*V, +V) -- so every
rank emits its contiguous witness range straight at its final global indices with NO exchange.  The only
collective is the optional all-gather that gives every rank every shard's columns:

  * gather_columns():      one all_gather_into_tensor per column -> globally contiguous columns (rank order ==
                           witness order), 9 collectives;
  * GatherPipeline:        the streaming form for batches that do not fit (2^23 x 223 KB = 1.87 TB): chunks are
                           emitted into ONE packed buffer (all 9 arrays back to back) so that each chunk is a single
                           all-gather, double-buffered so chunk k+1 is generated while chunk k is on the links.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a full gather makes every GPU ingest (N-1)/N of all bytes,
i.e. <= ~1.07 TB/s per GPU, well below one GPU's ~6.5 TB/s emission rate -- so the gather, not HBM, bounds a
gather-inclusive pipeline; bench.py reports both rates.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist

from .engine import Columns


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous witness range [lo, hi) of `rank`; the first (total % world) ranks get one extra item"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class ShardInfo:
    rank: int
    world: int
    lo: int
    hi: int
    gate_base: int  # global index of this shard's first row
    var_base: int   # global index of this shard's first variable


def range_check_sharded(engine, min_range, max_range, witness_local: torch.Tensor, total: int, gate_base: int = 0,
                        var_base: int = 0, group=None, out: Columns | None = None):
    """Emit this rank's shard of `total` range_check items at its global numbering.  `witness_local` holds the
    witnesses of shard_range(total, rank, world).  No communication.  Returns (Columns, result_vars, ShardInfo)."""
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    lo, hi = shard_range(total, rank, world)
    assert witness_local.shape[0] == hi - lo, (witness_local.shape, lo, hi)
    lay = engine.range_check_layout(min_range, max_range, hi - lo)
    info = ShardInfo(rank, world, lo, hi, gate_base + lo * lay.gates_per_item, var_base + lo * lay.vars_per_item)
    cols, res = engine.range_check_batch(min_range, max_range, witness_local, info.gate_base, info.var_base, out=out)
    return cols, res, info


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
                             var_base: int = 0, group=None):
    """Ragged max_bound (one public bound per item) over ranks: every rank plans its contiguous shard, the shard
    totals are all-gathered (exchange_totals), and each rank emits at the global numbering its prefix gives it.
    Returns (Columns, result_vars, ShardInfo, gates_per_rank, vars_per_rank)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    batch = witness_local.shape[0]
    nb, roff, voff = engine.ragged_buffers(batch)
    lay = engine.max_bound_ragged_plan(max_range_local, nb, roff, voff)
    gates, vars_ = exchange_totals(lay.n_gates, lay.n_vars, witness_local.device, group)
    g0, v0 = gate_base + sum(gates[:rank]), var_base + sum(vars_[:rank])
    cols = Columns.allocate(lay.n_gates, lay.n_vars, witness_local.device, g0, v0)
    res = torch.empty((batch,), dtype=torch.int64, device=witness_local.device)
    engine.max_bound_ragged_emit(max_range_local, witness_local, nb, roff, voff, cols, res, g0, v0)
    return cols, res, ShardInfo(rank, world, -1, -1, g0, v0), gates, vars_


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
                   group=None):
    """all-gather every column so each rank holds the whole batch's columns, contiguous in witness order"""
    g = {}
    for name in Columns.SCALAR_COLS + Columns.WIRE_COLS:
        g[name] = _gather_1d(getattr(cols, name), gates_per_rank, group)
    g["var_values"] = _gather_1d(cols.var_values, vars_per_rank, group)
    full = Columns(g["q_m"], g["q_l"], g["q_r"], g["q_o"], g["q_c"], g["w_l"], g["w_r"], g["w_o"], g["var_values"])
    res = None
    if result_vars is not None:
        world = len(gates_per_rank)
        counts = [torch.zeros(1, dtype=torch.int64, device=result_vars.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([result_vars.shape[0]], dtype=torch.int64, device=result_vars.device),
                        group=group)
        res = _gather_1d(result_vars, [int(c.item()) for c in counts], group)
    return full, res


# ---- packed chunks: one collective per chunk -----------------------------------------

def packed_layout(n_gates: int, n_vars: int):
    """offsets (in int64 words) of the 9 arrays inside one packed chunk buffer; every section 16-byte aligned"""
    def up(x):
        return (x + 1) & ~1
    off, sizes = {}, {}
    cur = 0
    for name in Columns.SCALAR_COLS:
        off[name], sizes[name] = cur, n_gates * 4
        cur += up(n_gates * 4)
    for name in Columns.WIRE_COLS:
        off[name], sizes[name] = cur, n_gates
        cur += up(n_gates)
    off["var_values"], sizes["var_values"] = cur, n_vars * 4
    cur += up(n_vars * 4)
    return off, sizes, cur


def columns_in(flat: torch.Tensor, n_gates: int, n_vars: int) -> Columns:
    """Columns whose arrays are views into the packed buffer `flat` (int64[packed words])"""
    off, sizes, total = packed_layout(n_gates, n_vars)
    assert flat.numel() >= total and flat.dtype == torch.int64
    v = {k: flat[off[k]:off[k] + sizes[k]] for k in off}
    return Columns(*[v[k].view(n_gates, 4) for k in Columns.SCALAR_COLS], *[v[k] for k in Columns.WIRE_COLS],
                   v["var_values"].view(n_vars, 4))


class GatherPipeline:
    """Streaming sharded range_check with a single all-gather per chunk (packed buffer), double-buffered:
    while chunk k is on the links, chunk k+1 is being emitted.  `consume(gathered, chunk_index)` sees
    gathered[r] = rank r's packed chunk (use columns_in to view it)."""

    def __init__(self, engine, min_range, max_range, chunk: int, group=None):
        self.engine, self.mn, self.mx, self.chunk, self.group = engine, min_range, max_range, chunk, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lay = engine.range_check_layout(min_range, max_range, chunk)
        _, _, self.words = packed_layout(self.lay.n_gates, self.lay.n_vars)
        dev = engine.device
        self.local = [torch.empty(self.words, dtype=torch.int64, device=dev) for _ in range(2)]
        self._gflat = [torch.empty(self.world * self.words, dtype=torch.int64, device=dev) for _ in range(2)]
        self.gathered = [g.view(self.world, self.words) for g in self._gflat]
        self.cols = [columns_in(b, self.lay.n_gates, self.lay.n_vars) for b in self.local]
        self.res = torch.empty((chunk,), dtype=torch.int64, device=dev)

    def bytes_per_chunk(self) -> int:
        return self.words * 8

    def run(self, witness_local: torch.Tensor, total_per_rank: int, gate_base: int = 0, var_base: int = 0, consume=None):
        """witness_local: this rank's total_per_rank witnesses (a multiple of chunk).  Global numbering: rank r's
        item i is item r*total_per_rank + i of the whole batch."""
        assert total_per_rank % self.chunk == 0
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
            if dist.is_initialized():
                # the collective is ordered after the emission on the compute stream by torch's ProcessGroup
                work = dist.all_gather_into_tensor(self._gflat[b], self.local[b], group=self.group, async_op=True)
            else:
                self.gathered[b][0].copy_(self.local[b])
                work = _Done()
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

    def __init__(self, engine, min_range, max_range, chunk: int, group=None):
        self.engine, self.mn, self.mx, self.chunk, self.group = engine, min_range, max_range, chunk, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lay = engine.range_check_layout(min_range, max_range, chunk)
        G, V = self.lay.n_gates, self.lay.n_vars
        dev = engine.device
        # per double-buffer slot: rows of every rank's chunk, one gathered variable table, this rank's own table
        self.rows = [[Columns.allocate(G, 0, dev) for _ in range(self.world)] for _ in range(2)]
        self._vflat = [torch.empty(self.world * V * 4, dtype=torch.int64, device=dev) for _ in range(2)]
        self.vars = [v.view(self.world, V, 4) for v in self._vflat]
        self.own = [torch.empty((V, 4), dtype=torch.int64, device=dev) for _ in range(2)]
        self.res = torch.empty((chunk,), dtype=torch.int64, device=dev)

    def bytes_on_the_links_per_chunk(self) -> int:
        return self.lay.n_vars * 32

    def _parts(self, b):
        out = []
        for r in range(self.world):
            c = self.rows[b][r]
            out.append(Columns(c.q_m, c.q_l, c.q_r, c.q_o, c.q_c, c.w_l, c.w_r, c.w_o, self.vars[b][r]))
        return out

    def run(self, witness_local: torch.Tensor, total_per_rank: int, gate_base: int = 0, var_base: int = 0, consume=None):
        """numbering as in GatherPipeline.run: rank r's item i is item r * total_per_rank + i of the whole batch"""
        assert total_per_rank % self.chunk == 0
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
            if dist.is_initialized():
                work = dist.all_gather_into_tensor(self._vflat[b], self.own[b].view(-1), group=self.group, async_op=True)
            else:
                self.vars[b][0].copy_(self.own[b])
                work = _Done()
            pending = (work, b, k)
        if pending is not None:
            work, kb, kk = pending
            work.wait()
            if consume is not None:
                consume(self._parts(kb), kk)


class _Done:
    def wait(self):
        return True
