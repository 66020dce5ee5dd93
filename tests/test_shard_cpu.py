"""The N > 1 path on CPU: two gloo ranks shard a batch of pairs with no
data-path collective, time it with the bench contract's barrier + max-over-
ranks, and collect the maps on rank 0."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from stereomatching_amd import shard  # noqa: E402


def test_pairs_for_rank_partitions_the_batch():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            parts = [shard.pairs_for_rank(total, r, world) for r in range(world)]
            flat = sorted(j for p in parts for j in p)
            assert flat == list(range(total))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
            for r, p in enumerate(parts):
                assert all(j % world == r for j in p)     # pair j -> rank j mod world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total_pairs, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from tests import oracle
    r, lr, w = shard.init("gloo")
    assert (r, w) == (rank, world)
    got = shard.broadcast_params({"config": "C3", "pairs": 3} if rank == 0 else {"config": "?"})
    assert got == {"config": "C3", "pairs": 3}
    mine = shard.pairs_for_rank(total_pairs, rank, world)
    # every rank "processes" its own pairs with the CPU oracle standing in for the
    # device (this test is about the sharding / collection logic, not the kernels)
    rng_maps = []
    for j in mine:
        rng = np.random.default_rng(j)
        le = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        re = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        rng_maps.append(oracle.hot_path(le, re, 8, 3)[1])
    local = torch.from_numpy(np.stack(rng_maps)) if rng_maps else torch.zeros((0, 12, 20), dtype=torch.int32)
    shard.barrier()
    t = shard.max_over_ranks(0.5 + rank)        # max over ranks = slowest rank
    assert t == pytest.approx(0.5 + world - 1)
    gathered = shard.gather_maps(local, total_pairs, rank, world)
    if rank == 0:
        np.save(Path(out_dir) / "gathered.npy", gathered.numpy())
    else:
        assert gathered is None
    dist.destroy_process_group()


@pytest.mark.parametrize("total_pairs", [4, 5])
def test_two_rank_gloo_shard_and_gather(tmp_path, total_pairs):
    from tests import oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total_pairs, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == (total_pairs, 12, 20)
    for j in range(total_pairs):
        rng = np.random.default_rng(j)
        le = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        re = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        assert np.array_equal(got[j], oracle.hot_path(le, re, 8, 3)[1]), j


@pytest.mark.parametrize("total_pairs", [7, 2])
def test_three_rank_gloo_gather_with_uneven_counts(tmp_path, total_pairs):
    """gather_maps' send / receive pairs at world 3: 7 pairs = shares of 3, 2, 2; 2 pairs = shares of
    1, 1, 0 (a rank with nothing to send takes no part in the transfer and still passes the barriers)"""
    from tests import oracle
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), total_pairs, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == (total_pairs, 12, 20)
    for j in range(total_pairs):
        rng = np.random.default_rng(j)
        le = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        re = rng.integers(0, 2, (12, 20), dtype=np.uint8)
        assert np.array_equal(got[j], oracle.hot_path(le, re, 8, 3)[1]), j


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True)
    assert p.returncode == 2 and "--nproc-per-node 2" in p.stderr


def _bench(args, **env):
    import json
    import subprocess
    e = {k: v for k, v in os.environ.items()
         if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=e, capture_output=True,
                       text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, [json.loads(l) for l in lines]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` alone (the command the driver issues): the parent
    starts two child ranks, they rendezvous (gloo here), pass the barriers and the
    max-over-ranks, and rank 0's single JSON line comes back through the parent.
    SM_BENCH_DRYRUN replaces the GPU step by a sleep -- this container has no GPU."""
    p, lines = _bench(["--gpus", "2", "--config", "C4", "--pairs", "8", "--steps", "6", "--warmup", "2"],
                      SM_BENCH_DRYRUN="1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    # and NOTHING else on stdout: gloo announces its connections on file descriptor 1
    assert len(p.stdout.strip().splitlines()) == 1, p.stdout[:500]
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["dry_run"] is True and out["value"] == 0.0       # can never pass for a measurement
    assert out["ms_per_step"] >= 0.5
    # the c4 object (SURVEY 8e's workload: 64 x 1080p, pair j -> rank j mod N, maps collected on rank 0)
    # is part of every N > 1 line; the dry run exercises its sharding, barriers and collection
    c4 = out["c4"]
    assert c4["dry_run"] is True and c4["value"] == 0.0
    assert c4["total_pairs"] == 64 and c4["pairs_per_rank"] == 32 and c4["maps_in_pair_order"] is True
    assert c4["gather_ms"] > 0


def test_bench_propagates_a_failing_rank():
    p, lines = _bench(["--gpus", "2", "--config", "NOPE", "--steps", "2", "--warmup", "1"],
                      SM_BENCH_DRYRUN="1")
    assert p.returncode != 0 and not lines


def test_bench_timing_stride_samples_enough_launches():
    sys.path.insert(0, str(ROOT))
    import bench
    for steps in (1, 5, 10, 20, 64, 200, 1000):
        every = bench.timing_stride(steps)
        assert 1 <= every <= 8
        assert (steps + every - 1) // every >= min(steps, 10)


def test_bench_fails_fast_when_a_rank_never_shows_up():
    """Round 5 (first-run readiness): a rank whose peers do not join -- a missing device, a communicator that does not
    come up -- must not hang the job: the rendezvous gives up after SM_BENCH_INIT_TIMEOUT seconds, rank 0 still prints
    ONE line, with `error` and value 0, and exits 4."""
    import time
    t0 = time.time()
    p, lines = _bench(["--gpus", "2", "--config", "C4", "--steps", "2", "--warmup", "1"], SM_BENCH_DRYRUN="1",
                      RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
                      SM_BENCH_INIT_TIMEOUT="4")
    assert p.returncode == 4, (p.returncode, p.stderr[-1500:])
    assert time.time() - t0 < 120
    assert len(lines) == 1 and lines[0]["value"] == 0.0 and lines[0]["verified"] is False
    assert "did not come up within 4 s" in lines[0]["error"] and "InitError" in lines[0]["error"]


def test_bench_under_the_drivers_own_launcher():
    """The command the driver issues for N > 1, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` (torchrun sets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*; bench.py must not start ranks of its own then).  Dry run: no GPU here."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SM_BENCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["dry_run"] is True
    assert out["c4"]["pairs_per_rank"] == 32 and out["c4"]["maps_in_pair_order"] is True
