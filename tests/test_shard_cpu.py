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


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True)
    assert p.returncode == 2 and "launch with torchrun" in p.stderr
