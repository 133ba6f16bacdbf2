"""GPU: the C-ABI multi-GPU slice with a world of TWO ranks on the box's ONE GPU.

No box of the pool has two GPUs, and with a communicator of one rank the interesting branches of csrc/capi_dist.inc are
dead code.  Here two fresh processes share device 0 and talk through tests/cpp/fake_rccl.c, a test-only collective the
library loads through PG_RCCL_LIB (stream-ordered bounce through POSIX shared memory): a rehearsal of ordering and
arithmetic -- other ranks' parts of a chunk, their regenerated rows, the totals exchange, bases of rank > 0 -- never a
measurement.  BASELINE.json config 5's shape: contiguous witness shards, one all-gather per chunk, bit-exact vs the CPU
oracle at the global numbering."""
import os
import socket
import subprocess
import sys

import pytest

import test_c_example as ce

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_rccl():
    sys.path.insert(0, os.path.join(ROOT, "tests", "cpp"))
    import build as cpp_build
    return cpp_build.build_fake_rccl()


def rank_env(extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PG_RCCL_LIB=fake_rccl(), LOCAL_RANK="0", FAKE_RCCL_TIMEOUT_S="90", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra or {})
    return env


def run_ranks(cmds, env, timeout=420):
    """starts one fresh child process per rank (nothing is re-exec'd), waits for all, kills the others when one fails"""
    procs = [subprocess.Popen(c, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for c in cmds]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.communicate()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        print(f"--- rank {r}: exit {p.returncode}\n{so}\n{se[-3000:]}")
    return [(p.returncode, so, se) for p, (so, se) in zip(procs, outs)]


@pytest.mark.parametrize("variables_only", [0, 1])
@pytest.mark.parametrize("delay_us", [0, 20000])
def test_c5_two_ranks_from_plain_c(tmp_path, variables_only, delay_us):
    """examples/c5_rank.c twice (`0 2 ...` and `1 2 ...`), packed and variables-only: BOTH ranks print the digest the CPU
    oracle gives for the two shards' chunks in (chunk, rank, array) order.  With a delay inside the collective (the data is
    published 20 ms late) a consumer that did not wait for its chunk would fold stale bytes."""
    binary = ce.C5_BIN if os.path.exists(ce.C5_BIN) else (ce.build() and ce.C5_BIN)
    world, total, chunk, bits = 2, 192, 64, 18
    ident = str(tmp_path / "comm.id")
    res = run_ranks([[binary, str(r), str(world), ident, str(total), str(chunk), str(variables_only), str(bits)] for r in range(world)],
                    rank_env({"FAKE_RCCL_DELAY_US": str(delay_us), "FAKE_RCCL_PIECE_BYTES": str(1 << 20)}))
    want, words = ce.c5_oracle_digest(world, total, chunk, bits)
    for r, (code, so, se) in enumerate(res):
        assert code == 0, (r, se[-2000:])
        line = so.strip().splitlines()[-1]
        assert line.startswith(f"rank {r} of {world}: {total // chunk} chunks, {words} words"), line
        assert line.endswith("digest %016x" % want), (r, line, "%016x" % want)


@pytest.mark.parametrize("variables_only", [0, 1])
def test_c5_four_ranks_from_plain_c(tmp_path, variables_only):
    """the same with FOUR ranks on the one GPU (four fresh child processes; the box allows six): every rank folds the other
    three's parts of every chunk in rank order -- world = 2 cannot tell rank order from "the other one" -- and the collective
    publishes its data 5 ms late"""
    binary = ce.C5_BIN if os.path.exists(ce.C5_BIN) else (ce.build() and ce.C5_BIN)
    world, total, chunk, bits = 4, 128, 32, 18
    ident = str(tmp_path / "comm4.id")
    res = run_ranks([[binary, str(r), str(world), ident, str(total), str(chunk), str(variables_only), str(bits)] for r in range(world)],
                    rank_env({"FAKE_RCCL_DELAY_US": "5000", "FAKE_RCCL_PIECE_BYTES": str(1 << 20)}))
    want, words = ce.c5_oracle_digest(world, total, chunk, bits)
    for r, (code, so, se) in enumerate(res):
        assert code == 0, (r, se[-2000:])
        line = so.strip().splitlines()[-1]
        assert line.startswith(f"rank {r} of {world}: {total // chunk} chunks, {words} words"), line
        assert line.endswith("digest %016x" % want), (r, line, "%016x" % want)


def test_c5_rank_default_bits(tmp_path):
    """the example's default MAX_BITS (advisor finding: 254 made ~9 % of the witnesses non-canonical and the program exit 1)"""
    binary = ce.C5_BIN if os.path.exists(ce.C5_BIN) else (ce.build() and ce.C5_BIN)
    p = subprocess.run([binary, "0", "1", str(tmp_path / "comm.id"), "64", "32", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    want, words = ce.c5_oracle_digest(1, 64, 32, 252)
    assert p.stdout.strip().endswith("digest %016x" % want), p.stdout
    p = subprocess.run([binary, "0", "1", str(tmp_path / "comm2.id"), "64", "32", "1", "254"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "at most 253" in p.stderr


def test_two_ranks_through_the_python_layer():
    """tests/world2_rank.py twice: pg_allgather_bytes in several pieces, pg_allgather_columns with two parts, ragged shards
    (uneven: 170 + 131 items; rank 1's bases = rank 0's totals; a failing plan on one rank fails both), and both forms of
    config 5's pipeline with a consumer that compares every rank's part of every chunk with the oracle"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = run_ranks([[sys.executable, os.path.join(ROOT, "tests", "world2_rank.py"), str(r), "2", str(port)] for r in range(2)],
                    rank_env({"FAKE_RCCL_PIECE_BYTES": str(1 << 20), "FAKE_RCCL_DELAY_US": "2000"}), timeout=900)
    for r, (code, so, se) in enumerate(res):
        assert code == 0 and f"rank {r} OK" in so, (r, se[-3000:])


def test_three_ranks_through_the_python_layer():
    """the same script with THREE ranks (three processes on the GPU): a world that is not a power of two, ragged shards cut in
    thirds -- rank 2's bases = the totals of ranks 0 AND 1 --, three parts in pg_allgather_columns and in every chunk"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = run_ranks([[sys.executable, os.path.join(ROOT, "tests", "world2_rank.py"), str(r), "3", str(port)] for r in range(3)],
                    rank_env({"FAKE_RCCL_PIECE_BYTES": str(1 << 20), "FAKE_RCCL_DELAY_US": "2000"}), timeout=900)
    for r, (code, so, se) in enumerate(res):
        assert code == 0 and f"rank {r} OK" in so, (r, se[-3000:])


def _run_bench(args, extra_env, timeout=600):
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=rank_env(dict(PG_DIST_BACKEND="gloo", **extra_env)),
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_bench_five_ranks_rehearsal_with_the_gather_leg():
    """`bench.py --gpus 5`, the EXACT N > 1 code path of config 5 at tiny sizes: a GPU-less parent starts five fresh rank
    processes (this box admits six on its one card, this test runner being one of them: the driver's N = 8 is the same
    code with three more children), torch.distributed on gloo for the rendezvous, the library's own collective (pg_comm) bound to
    the test-only stand-in through PG_RCCL_LIB.  The line must carry the headline at n_gpus = 5, the per-rank HBM budget, and BOTH
    gather-inclusive figures -- GatherPipeline (every packed chunk gathered) and VariablesOnlyPipeline (tables gathered, the other
    ranks' rows regenerated) -- with the watchdog armed and cancelled."""
    world, lg = 5, 8
    p, lines = _run_bench(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--log2-batch", str(lg), "--no-cpu", "--allgather-log2-chunk", "5",
                           "--allgather-chunks", "3", "--allgather-timeout", "240", "--n1-value", "1e9"], {"FAKE_RCCL_PIECE_BYTES": str(1 << 20)})
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["config"]["items_per_gpu"] == 1 << lg
    rows = world * 2 * (1 << lg) * 1031  # ranks x steps x witnesses x rows
    assert abs(line["value"] * line["ms_per_step"] * 2 / 1e3 - rows) < 1e-6 * rows
    ag = line["allgather"]
    assert "error" not in ag and ag["value"] > 0 and ag["witnesses_per_rank"] == 3 * 32 and ag["witnesses_per_chunk"] == 32
    assert ag["bytes_per_rank_per_chunk"] >= 32 * 222_792
    vo = ag["variables_only"]
    assert "error" not in vo and vo["value"] > 0 and vo["bytes_per_rank_per_chunk"] == 32 * 1034 * 32
    b = line["hbm_budget"]
    assert b["ranks_sharing_the_card"] == world and b["gather_pipeline_bytes"] == 2 * (world + 1) * 32 * 222_792
    assert "rank budget" in p.stderr
    # both curves SURVEY 8(e) asks for, each against N x the N = 1 value the run was given
    assert line["n1_value"] == {"value": 1e9, "source": "--n1-value"}
    assert abs(line["efficiency_generation"] - line["value"] / (world * 1e9)) < 1e-12
    assert abs(line["efficiency_gather_inclusive"] - ag["value"] / (world * 1e9)) < 1e-12
    assert abs(line["efficiency_gather_inclusive_variables_only"] - vo["value"] / (world * 1e9)) < 1e-12


def test_bench_watchdog_abandons_a_gather_that_hangs():
    """a collective that does not come back (here: the stand-in publishes every exchange 3 s late, the limit is 2 s) must not
    cost the headline: rank 0 prints the line it has -- exactly one -- with the reason in `allgather.error`, and the run exits
    3, not 0; no rank is re-exec'd or restarted"""
    p, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "8", "--no-cpu", "--allgather-log2-chunk", "5",
                           "--allgather-chunks", "2", "--allgather-timeout", "2"], {"FAKE_RCCL_DELAY_US": "3000000"}, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-3000:])
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0 and "did not finish" in lines[0]["allgather"]["error"]


def test_bench_refuses_a_gather_pipeline_that_cannot_fit():
    """the per-rank budget is checked before anything is allocated: two slots x (world + 1) packed chunks of 2^22 witnesses are
    5.6 TB -- refused with the figures, exit code non-zero, no line"""
    p, lines = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-batch", "8", "--no-cpu", "--allgather-log2-chunk", "22"], {},
                          timeout=300)
    assert p.returncode != 0 and not lines
    assert "gather pipeline alone needs" in p.stderr and "--allgather-log2-chunk" in p.stderr
