"""Batch sharding across the GPUs of a node.

Stereo pairs are independent, so the hot path shards with NO data-path
collective: pair j belongs to rank j mod world.  torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for
the barrier / max-over-ranks of the timing contract, the broadcast of rank 0's
job description and, optionally, to collect the result maps on rank 0.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


class InitError(RuntimeError):
    """this rank could not join the job (no such device, rendezvous or communicator start-up failed or timed out)"""


def init(backend: str | None = None, force: bool = False, timeout_s: float = 120.0):
    """Join the job described by the torchrun environment.  With the nccl (= RCCL) backend
    the rank's GPU is selected BEFORE the process group exists: every collective, the
    object broadcast included, runs on the current device, and two ranks on one device
    is an RCCL error.  `force` creates the group even for a single rank (self-tests).
    Fails FAST and by name (InitError) instead of hanging: a rank whose device does not exist says so before the
    rendezvous, the rendezvous and every later collective give up after `timeout_s`."""
    import datetime
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if local_rank >= n_dev:
                raise InitError(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) are visible "
                                f"(one process per GPU: --gpus / --nproc-per-node must not exceed the node's GPUs)")
            try:
                torch.cuda.set_device(local_rank)
            except Exception as exc:          # noqa: BLE001
                raise InitError(f"rank {rank}: cannot select GPU {local_rank}: {exc}") from exc
        try:
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s))
        except Exception as exc:              # noqa: BLE001
            raise InitError(f"rank {rank}: {backend} process group of {world} rank(s) did not come up within "
                            f"{timeout_s:.0f} s ({type(exc).__name__}: {exc})") from exc
    return rank, local_rank, world


def pairs_for_rank(total_pairs: int, rank: int, world: int) -> list[int]:
    """Indices of the pairs rank `rank` processes: j with j mod world == rank."""
    return list(range(rank, total_pairs, world))


def broadcast_params(params: dict, src: int = 0) -> dict:
    """Rank `src`'s parameter block (configuration, threshold, batch size ...) to every
    rank, so that all shards run the same job: the 'broadcast' half of the north star's
    'RCCL broadcast / gather only for result collection'.  A few hundred bytes."""
    if not dist.is_initialized():
        return params
    box = [params]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            # name the device explicitly: one process per GPU, the current one
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(seconds: float, device="cpu") -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_maps(local: torch.Tensor, total_pairs: int, rank: int, world: int):
    """Collect every rank's (pairs_r, H, W) maps on rank 0 in pair order: each rank
    r > 0 sends exactly its own maps to rank 0 (point-to-point over RCCL / xGMI, all
    transfers in flight together); nobody else receives anything.  Ranks may hold
    different counts (total_pairs % world != 0).  Returns (total_pairs, H, W) on rank
    0, None elsewhere."""
    if world == 1:
        return local
    if not dist.is_initialized():
        raise RuntimeError("gather_maps: no process group (call shard.init first)")
    shape = tuple(local.shape[1:])
    counts = [len(pairs_for_rank(total_pairs, r, world)) for r in range(world)]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} maps, its share is {counts[rank]}")
    if rank != 0:
        if counts[rank]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), 0)]):
                req.wait()
        return None
    blocks = {0: local}
    ops = []
    for r in range(1, world):
        if counts[r]:
            blocks[r] = torch.empty((counts[r],) + shape, dtype=local.dtype, device=local.device)
            ops.append(dist.P2POp(dist.irecv, blocks[r], r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out = torch.empty((total_pairs,) + shape, dtype=local.dtype, device=local.device)
    for r, blk in blocks.items():
        out[pairs_for_rank(total_pairs, r, world)] = blk
    return out


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
