"""Batch sharding of the encode/decode path across the GPUs of one node (one process per GPU).

Images are independent units on the eval path (no cross-sample op anywhere; weights and codebook are
read-only replicas), so the path shards by batch with NO data-path collective inside encode or decode.
The single exchange is an all-gather of the token ids after encode ([B_local,K] int32: 128 KiB per
rank at B_local=64) so that every rank holds the full id matrix -- RCCL over xGMI through
torch.distributed's "nccl" backend on ROCm; "gloo" on CPU for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)     # bind the communicator to this rank's GPU up front
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:                                      # older torch without device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous slice [lo, hi) of `total` images owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_ids(ids_local: torch.Tensor) -> torch.Tensor:
    """[B_local,K] token ids (any int dtype) -> [B_total,K] on every rank, in rank order.
    The payload travels as int32 (ids < 2^15); uneven shards are padded to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ids_local
    world = dist.get_world_size()
    K = ids_local.shape[1]
    out_device = ids_local.device
    if dist.get_backend() == "gloo" and ids_local.is_cuda:      # CPU collective (tests / single-GPU dry runs of the N>1 flow)
        ids_local = ids_local.cpu()
    n_local = torch.tensor([ids_local.shape[0]], dtype=torch.int64, device=ids_local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    send = torch.zeros(nmax, K, dtype=torch.int32, device=ids_local.device)
    send[: ids_local.shape[0]] = ids_local.to(torch.int32)
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(bufs, send)                       # ncclAllGather over xGMI on the GPU box
    return torch.cat([bufs[r][: counts[r]] for r in range(world)]).to(ids_local.dtype).to(out_device)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """in-place SUM over ranks (RCCL all-reduce on the GPU box; the training-side VQ's bins / embed_sum); no-op with one rank"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.broadcast(c, src=src)
            t.copy_(c)
        else:
            dist.broadcast(t, src=src)
    return t


def backend_name():
    """'nccl' (= RCCL on ROCm) / 'gloo' / None when there is a single rank"""
    return dist.get_backend() if dist.is_initialized() and dist.get_world_size() > 1 else None


def all_gather_ids_timed(ids_local: torch.Tensor):
    """all_gather_ids + its duration in ms: HIP events on the current stream for RCCL (the collective is enqueued on the
    stream, the host does not wait), host clock for gloo.  (ids_all, 0.0) with a single rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ids_local, 0.0
    if dist.get_backend() == "nccl" and ids_local.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = all_gather_ids(ids_local)
        e1.record()
        e1.synchronize()
        return out, float(e0.elapsed_time(e1))
    import time
    t0 = time.perf_counter()
    out = all_gather_ids(ids_local)
    return out, 1000.0 * (time.perf_counter() - t0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
