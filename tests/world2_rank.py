"""One rank of the world-2-on-ONE-GPU rehearsal of the C-ABI multi-GPU slice (started twice by tests/test_gpu_world2.py).

    python tests/world2_rank.py RANK WORLD PORT

Both ranks use device 0.  The library's communicator (pg_comm) is created on tests/cpp/libfake_rccl.so -- the test-only
collective PG_RCCL_LIB points at -- so that every branch of csrc/capi_dist.inc a communicator of one rank leaves dead runs:
other ranks' parts of a chunk, the regeneration of their rows, the totals exchange of ragged shards, bases of rank > 0,
pg_allgather_columns with more than one part.  Every array a rank ends up with is compared with the CPU oracle's rows of the
WHOLE batch at the global numbering.  The gloo group only carries the communicator id (NativeCollective's rendezvous).
Prints "rank R OK" at the end; any mismatch is an AssertionError (exit code 1)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
MN, MX = 50_000, 250_000


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert os.environ.get("PG_RCCL_LIB", "").endswith("libfake_rccl.so"), "this rehearsal runs on the test-only collective"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    from plonk_gadgets_amd import _lib, distributed as pd, synth
    lib = _lib.load()
    eng = pg.Engine(0)
    coll = pd.NativeCollective(eng)
    assert (coll.rank, coll.world) == (rank, world)
    assert lib.pg_comm_library().decode().endswith("libfake_rccl.so"), lib.pg_comm_library()
    dev = "cuda:0"
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)

    def witnesses(total, seed):
        inside = synth.scalars_from_ints([MN + int(v) % (MX - MN) for v in synth.splitmix64(total // 2, seed)])
        both = np.concatenate([inside, synth.random_scalars(total - total // 2, seed + 1)])
        return np.ascontiguousarray(both[np.argsort(synth.splitmix64(total, seed + 2), kind="stable")])  # in / out of range interleaved

    # ---- 1. raw bytes: rank r sends r * 2^32 + i; several exchanges larger than the collective's piece size ------------------------
    n = 3 * (1 << 17) + 40  # 3 MiB + 320 B per rank: FAKE_RCCL_PIECE_BYTES = 1 MiB in the test -> four pieces
    src = torch.arange(n, dtype=torch.int64, device=dev) + (rank << 32)
    dst = torch.full((world * n,), -1, dtype=torch.int64, device=dev)
    coll.all_gather(dst, src).wait()
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(dst[r * n:(r + 1) * n], torch.arange(n, dtype=torch.int64, device=dev) + (r << 32)), ("bytes", r)

    # ---- 2. equal shards of range_check, pg_allgather_columns (nine grouped gathers) -----------------------------------
    total = 12 * world
    wit = witnesses(total, 51)
    ora = po.range_check_batch(synth.mont(MN), synth.mont(MX), wit)
    assert ora["satisfied"] and (ora["gate_base"], ora["var_base"]) == (3, 5)
    lo, hi = pd.shard_range(total, rank, world)
    cols, res, info = pd.range_check_sharded(eng, mn, mx, torch.from_numpy(wit[lo:hi].view(np.int64)).to(dev), total, 3, 5)
    G, V = info.n_gates // (hi - lo), info.n_vars // (hi - lo)
    assert (info.gate_base, info.var_base) == (3 + lo * G, 5 + lo * V)
    full, full_res = pd.gather_columns(cols, res, [info.n_gates] * world, [info.n_vars] * world, collective=coll)
    torch.cuda.synchronize()
    got = full.to_numpy()
    for name in COLS:
        assert np.array_equal(got[name], ora[name]), ("allgather_columns", name)
    assert np.array_equal(full_res.cpu().numpy().view(np.uint64), ora["result_vars"])

    # ---- 3. ragged shards of max_bound: plan, totals through the collective, emission at the exchanged bases -------------------
    batch = 301
    cut = [0, 170, batch] if world == 2 else [batch * r // world for r in range(world + 1)]
    mr_np, wt_np = bench.c4_inputs(batch, seed=0xC4)
    orag = po.max_bound_batch(mr_np, wt_np)
    rows = 2 * orag["num_bits"].astype(np.int64) + 5
    a, b = cut[rank], cut[rank + 1]
    mr, wt = (torch.from_numpy(np.ascontiguousarray(x[a:b]).view(np.int64)).to(dev) for x in (mr_np, wt_np))
    c2, r2, sinfo, gates, vars_ = pd.max_bound_ragged_sharded(eng, mr, wt, 3, 5, collective=coll)
    torch.cuda.synchronize()
    assert sum(gates) == orag["n_gates"] and sum(vars_) == orag["n_vars"], (gates, vars_)
    assert (sinfo.gate_base, sinfo.var_base) == (3 + sum(gates[:rank]), 5 + sum(vars_[:rank]))
    assert gates == [int(rows[cut[r]:cut[r + 1]].sum()) for r in range(world)]
    g0, v0 = sum(gates[:rank]), sum(vars_[:rank])
    got = c2.to_numpy()
    for name in COLS:
        first, cnt = (v0, vars_[rank]) if name == "var_values" else (g0, gates[rank])
        assert np.array_equal(got[name], orag[name][first:first + cnt]), ("ragged", name)
    assert np.array_equal(r2.cpu().numpy().view(np.uint64), orag["result_vars"][a:b])
    # the same through the one-call form (pg_max_bound_ragged_sharded_batch), into worst-case buffers
    nb, roff, voff = eng.ragged_buffers(b - a)
    wc = pg.Columns.allocate(515 * (b - a), 517 * (b - a), dev, 3, 5)
    wres = torch.empty((b - a,), dtype=torch.int64, device=dev)
    cc, s = wc.as_c(), _lib.ShardC()
    st = lib.pg_max_bound_ragged_sharded_batch(coll._h, mr.data_ptr(), wt.data_ptr(), b - a, nb.data_ptr(), roff.data_ptr(),
                                               voff.data_ptr(), 3, 5, C.byref(cc), wres.data_ptr(), C.byref(s), eng._stream())
    assert st == 0, lib.pg_last_error()
    torch.cuda.synchronize()
    assert (s.rank, s.world, s.gate_base, s.var_base, s.n_gates, s.n_vars) == (rank, world, 3 + g0, 5 + v0, gates[rank], vars_[rank])
    got = wc.to_numpy()
    for name in COLS:
        first, cnt = (v0, vars_[rank]) if name == "var_values" else (g0, gates[rank])
        assert np.array_equal(got[name][:cnt], orag[name][first:first + cnt]), ("ragged one-call", name)
    # a rank whose plan fails (a NULL bounds array) still joins the exchange: EVERY rank gets an error, nobody hangs
    st = lib.pg_max_bound_ragged_sharded_plan(coll._h, None if rank == world - 1 else mr.data_ptr(), b - a, nb.data_ptr(),
                                              roff.data_ptr(), voff.data_ptr(), 3, 5, C.byref(s), None, None, eng._stream())
    assert st != 0, "a failed plan on one rank must fail the call on every rank"

    # ---- 4. config 5's pipeline, both modes: every part of every chunk == the oracle's rows of those items ------------------
    per_rank, chunk = 48, 16
    wit = witnesses(world * per_rank, 77)
    ora = po.range_check_batch(synth.mont(MN), synth.mont(MX), wit)
    mine = torch.from_numpy(wit[rank * per_rank:(rank + 1) * per_rank].view(np.int64)).to(dev)

    def check_part(part, r, k, what):
        first = r * per_rank + k * chunk
        for name in COLS:
            per = V if name == "var_values" else G
            got = getattr(part, name).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, ora[name][first * per:(first + chunk) * per]), (what, "chunk", k, "rank's part", r, name)

    seen = []
    pipe = pd.GatherPipeline(eng, mn, mx, chunk, collective=coll)
    assert pipe.world == world and pipe.native is not None

    def consume_packed(gathered, k):
        assert gathered.shape == (world, pipe.words)
        for r in range(world):
            check_part(pd.columns_in(gathered[r], pipe.lay.n_gates, pipe.lay.n_vars), r, k, "packed")
        seen.append(k)

    pipe.run(mine, per_rank, 3, 5, consume=consume_packed)
    assert seen == list(range(per_rank // chunk))
    # no host synchronisation between the chunks: the last two must still be intact at the end
    kept = []
    pipe.run(mine, per_rank, 3, 5, consume=lambda g, k: kept.append((k, g.clone())) if k >= per_rank // chunk - 2 else None)
    torch.cuda.synchronize()
    for k, g in kept:
        for r in range(world):
            check_part(pd.columns_in(g[r], pipe.lay.n_gates, pipe.lay.n_vars), r, k, "packed, unsynchronised")
    pipe.native.close()

    seen = []
    vpipe = pd.VariablesOnlyPipeline(eng, mn, mx, chunk, collective=coll)

    def consume_vars(parts, k):
        assert len(parts) == world
        for r in range(world):
            check_part(parts[r], r, k, "variables only")
        seen.append(k)

    vpipe.run(mine, per_rank, 3, 5, consume=consume_vars)
    assert seen == list(range(per_rank // chunk))
    assert vpipe.native.bytes_per_chunk() == chunk * V * 32
    vpipe.native.close()

    coll.close()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK", flush=True)


if __name__ == "__main__":
    main()
