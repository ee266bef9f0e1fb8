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


# A single rank needs no exchange, so every helper below returns early at world_size 1 -- which also means that a one-GPU box never
# executes the communicator set-up, the side stream / event join or a collective.  `force_single_rank(True)` (tests, smoke()) makes
# the helpers take their collective path whenever a process group exists, so the RCCL device path runs with one rank too.
_FORCE_SINGLE = False


def force_single_rank(on: bool = True) -> bool:
    """take the collective path even with ONE rank (needs an initialised process group); returns the previous setting"""
    global _FORCE_SINGLE
    prev, _FORCE_SINGLE = _FORCE_SINGLE, bool(on)
    return prev


def _active() -> bool:
    """is there an exchange to perform: more than one rank, or a forced single-rank group"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_SINGLE)


def init_from_env(backend: str = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group if world > 1 (or, with
    `single_rank_group`, also for one rank: the communicator is then created and bound exactly as for N ranks)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
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


class IdGatherer:
    """The path's ONE exchange step: every rank contributes its [B_local,K] token ids and receives the whole [B_total,K] matrix.

    One collective per step -- `all_gather_into_tensor` (ncclAllGather on RCCL) of an int32 payload (ids < 2^15) into a
    preallocated [world * B_max, K] buffer.  The per-rank row counts are exchanged ONCE, here in the constructor (they are a
    property of the sharding, not of a step), so a step has no second collective and no host synchronisation.  On a GPU the
    collective is enqueued on a side stream behind an event of the producing stream: the caller's stream is free to start decoding
    the local shard (each rank only needs ITS ids for that; the gathered matrix is the API result) and joins in `wait()`."""

    def __init__(self, b_local: int, K: int, device, counts=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = _active()                 # False: one rank and nothing forced -> launch/wait pass the ids through
        self.rank = dist.get_rank() if self.active else 0
        self.K, self.b_local = int(K), int(b_local)
        self.device = torch.device(device)
        self.collectives = 0                    # collectives issued by step calls (tests assert one per step)
        self._cpu_pg = self.active and dist.get_backend() == "gloo"
        self._buf_dev = torch.device("cpu") if self._cpu_pg else self.device
        if counts is None:
            if self.active:                     # setup-time exchange of the shard sizes (once per sharding, not per step)
                mine = torch.tensor([self.b_local], dtype=torch.int64, device=self._buf_dev)
                allc = torch.empty(self.world, dtype=torch.int64, device=self._buf_dev)
                dist.all_gather_into_tensor(allc, mine)
                counts = [int(c) for c in allc.cpu().tolist()]
            else:
                counts = [self.b_local]
        assert len(counts) == self.world and counts[self.rank] == self.b_local, (counts, self.rank, self.b_local)
        self.counts = list(counts)
        self.bmax = max(self.counts) if self.counts else 0
        self.even = all(c == self.bmax for c in self.counts)
        self.send = torch.zeros(self.bmax, self.K, dtype=torch.int32, device=self._buf_dev)
        self.recv = torch.empty(self.world * self.bmax, self.K, dtype=torch.int32, device=self._buf_dev)    # rank-major, concatenated along dim 0
        self.stream = torch.cuda.Stream(device=self.device) if (self.active and self._buf_dev.type == "cuda") else None
        self._done = None
        self._t0 = self._t1 = None
        self._host_ms = 0.0

    @property
    def payload_bytes(self) -> int:
        return int(self.recv.numel() * 4) if self.active else 0

    def launch(self, ids_local: torch.Tensor, timed: bool = False) -> None:
        """enqueue the all-gather of this step's ids; returns at once (GPU: nothing waits on the host)"""
        assert tuple(ids_local.shape) == (self.b_local, self.K), (tuple(ids_local.shape), (self.b_local, self.K))
        self._dtype, self._out_dev, self._local = ids_local.dtype, ids_local.device, ids_local
        if not self.active:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))       # ids are produced on the caller's stream
            with torch.cuda.stream(self.stream):
                if timed:
                    self._t0, self._t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    self._t0.record()
                self.send[: self.b_local].copy_(ids_local, non_blocking=True)     # int64 -> int32 cast rides in the copy
                dist.all_gather_into_tensor(self.recv, self.send)                 # THE collective of the step
                if timed:
                    self._t1.record()
                self._done = torch.cuda.Event()
                self._done.record()
            ids_local.record_stream(self.stream)
        else:                                                                     # gloo (CPU tests / one-GPU dry runs): host collective
            import time
            t0 = time.perf_counter()
            self.send[: self.b_local].copy_(ids_local)
            dist.all_gather_into_tensor(self.recv, self.send)
            self._host_ms = 1000.0 * (time.perf_counter() - t0)
        self.collectives += 1

    def wait(self) -> torch.Tensor:
        """[B_total,K] ids of every rank in rank order, in the dtype / on the device of the ids handed to launch()"""
        if not self.active:
            return self._local
        if self._done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._done)         # device-side join, no host wait
        if self.even:
            out = self.recv
        else:
            out = torch.cat([self.recv[r * self.bmax: r * self.bmax + self.counts[r]] for r in range(self.world)])
        return out.to(device=self._out_dev, dtype=self._dtype, copy=True)     # `recv` is overwritten by the next launch

    def last_ms(self) -> float:
        """duration of the last launch(timed=True): HIP events on the side stream for RCCL (synchronises on them), host clock for gloo"""
        if not self.active:
            return 0.0
        if self._t1 is not None:
            self._t1.synchronize()
            return float(self._t0.elapsed_time(self._t1))
        return self._host_ms


_gatherers = {}


def id_gatherer(b_local: int, K: int, device) -> IdGatherer:
    """build the IdGatherer of a sharding: a COLLECTIVE call (the shard sizes are exchanged here, once) that every rank makes
    together.  Nothing is cached behind the caller's back: the object is valid for as long as every rank's shard size stays what it
    was when it was built (bench.py builds one per run)."""
    return IdGatherer(b_local, K, device)


def _exchange_counts(b_local: int, device) -> tuple:
    """every rank's shard size, agreed by all ranks (one tiny collective)"""
    buf_dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)
    mine = torch.tensor([int(b_local)], dtype=torch.int64, device=buf_dev)
    allc = torch.empty(dist.get_world_size(), dtype=torch.int64, device=buf_dev)
    dist.all_gather_into_tensor(allc, mine)
    return tuple(int(c) for c in allc.cpu().tolist())


def _gatherer_for_call(ids_local: torch.Tensor) -> IdGatherer:
    """the convenience entry points may be called with a different batch every time, and a rank cannot know from ITS shard size
    whether another rank's changed (total 5 -> (3,2), total 6 -> (3,3): rank 0 sees 3 both times).  So they exchange the sizes on
    EVERY call and key the buffer cache on the whole agreed vector -- all ranks then take the same branch by construction."""
    counts = _exchange_counts(ids_local.shape[0], ids_local.device)
    key = (counts, int(ids_local.shape[1]), str(ids_local.device), _FORCE_SINGLE)
    g = _gatherers.get(key)
    if g is None:
        if len(_gatherers) >= 16:               # buffers of shardings no longer in use
            _gatherers.clear()
        g = _gatherers[key] = IdGatherer(ids_local.shape[0], ids_local.shape[1], ids_local.device, counts=list(counts))
    return g


def all_gather_ids(ids_local: torch.Tensor) -> torch.Tensor:
    """[B_local,K] token ids (any int dtype) -> [B_total,K] on every rank, in rank order.  Two collectives per call: the shard sizes
    (8 bytes per rank) and the int32 payload; uneven shards are padded to the largest shard inside the payload.  A loop over one
    fixed sharding should hold an `IdGatherer` instead (one collective per step, no host synchronisation)."""
    if not _active():
        return ids_local
    g = _gatherer_for_call(ids_local)
    g.launch(ids_local)
    return g.wait()


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """in-place SUM over ranks (RCCL all-reduce on the GPU box; the training-side VQ's bins / embed_sum); no-op with one rank"""
    if _active():
        if dist.get_backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_rows(local: torch.Tensor, counts) -> torch.Tensor:
    """[counts[rank], ...] rows of every rank -> [sum(counts), ...] in rank order; the shard sizes are known to every rank (no size
    exchange).  One all_gather_into_tensor of shards padded to the largest (the reference: all_gather_variably_sized_v2,
    vector_quantize_pytorch.py:262)."""
    if not _active():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [int(c) for c in counts]
    assert len(counts) == world and local.shape[0] == counts[rank], (counts, rank, tuple(local.shape))
    cmax = max(counts)
    cpu_pg = dist.get_backend() == "gloo" and local.is_cuda
    src = local.cpu() if cpu_pg else local
    send = torch.zeros((cmax,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    send[: counts[rank]] = src
    recv = torch.empty((world * cmax,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(recv, send)
    out = torch.cat([recv[r * cmax: r * cmax + counts[r]] for r in range(world)])
    return out.to(local.device)


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if _active():
        if dist.get_backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.broadcast(c, src=src)
            t.copy_(c)
        else:
            dist.broadcast(t, src=src)
    return t


def backend_name():
    """'nccl' (= RCCL on ROCm) / 'gloo' / None when there is a single rank"""
    return dist.get_backend() if _active() else None


def all_gather_ids_timed(ids_local: torch.Tensor):
    """all_gather_ids + its duration in ms (HIP events on the collective's stream for RCCL, host clock for gloo); (ids, 0.0) with
    a single rank.  Synchronises on the end event: benchmarks that overlap the gather with decode use IdGatherer directly."""
    if not _active():
        return ids_local, 0.0
    g = _gatherer_for_call(ids_local)
    g.launch(ids_local, timed=True)
    out = g.wait()
    return out, g.last_ms()


def barrier():
    if _active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    _gatherers.clear()
    if dist.is_initialized():
        dist.destroy_process_group()
