"""not gpu: the N>1 path (batch sharding + id all-gather) with world_size 2 over gloo on CPU."""
import os

import torch
import torch.multiprocessing as mp

from selftoktokenizer_amd import dist as D, synth


def test_shard_range_partitions():
    for total, world in ((512, 8), (10, 4), (3, 2), (64, 1)):
        spans = [D.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    lo, hi = D.shard_range(total, r, w)
    all_ids = torch.from_numpy(synth.synthetic_token_ids(total))          # what a single process would produce
    mine = all_ids[lo:hi]                                                  # "encode" of my shard (images are independent)
    gathered = D.all_gather_ids(mine)
    t = D.max_over_ranks(float(rank + 1), "cpu")
    D.barrier()
    q.put((rank, bool(torch.equal(gathered, all_ids)), gathered.dtype == torch.int64, t))
    torch.distributed.destroy_process_group()


def _run(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_all_gather_ids_world2_gloo_even_and_uneven():
    for total in (4, 5):
        for rank, same, is64, tmax in _run(total):
            assert same and is64 and tmax == 2.0
