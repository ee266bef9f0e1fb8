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


def _step_worker(rank, world, port, total, q):
    """a step issues exactly ONE collective (all_gather_into_tensor on a preallocated int32 buffer); the shard
    sizes are exchanged once when the gatherer is built; the list-form all_gather is never used"""
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    calls = {"into": 0, "list": 0, "other": 0}
    real_into, real_list = dist.all_gather_into_tensor, dist.all_gather

    def into(*a, **k):
        calls["into"] += 1
        return real_into(*a, **k)

    def lst(*a, **k):
        calls["list"] += 1
        return real_list(*a, **k)
    dist.all_gather_into_tensor, dist.all_gather = into, lst
    for name in ("all_reduce", "broadcast", "all_to_all", "gather", "reduce_scatter", "all_gather_object"):
        real = getattr(dist, name)
        setattr(dist, name, (lambda real: lambda *a, **k: (calls.__setitem__("other", calls["other"] + 1), real(*a, **k))[1])(real))
    lo, hi = D.shard_range(total, r, w)
    g = D.id_gatherer(hi - lo, 512, "cpu")
    setup = dict(calls)
    ok = setup == {"into": 1, "list": 0, "other": 0} and g.counts == [D.shard_range(total, i, w)[1] - D.shard_range(total, i, w)[0] for i in range(w)]
    per_step = []
    for step in range(3):
        all_ids = torch.from_numpy(synth.synthetic_token_ids(total, first_index=7 * step))
        before = dict(calls)
        g.launch(all_ids[lo:hi])
        out = g.wait()
        per_step.append({k: calls[k] - before[k] for k in calls})
        ok = ok and bool(torch.equal(out, all_ids)) and out.dtype == torch.int64
        before = dict(calls)
        ok = ok and bool(torch.equal(D.all_gather_ids(all_ids[lo:hi].to(torch.int32)), all_ids.to(torch.int32)))
        # the convenience call owns its own buffers and re-agrees the shard sizes every time: sizes + payload
        ok = ok and {k: calls[k] - before[k] for k in calls} == {"into": 2, "list": 0, "other": 0}
    ok = ok and all(s == {"into": 1, "list": 0, "other": 0} for s in per_step) and g.collectives == 3
    ok = ok and g.payload_bytes == w * g.bmax * 512 * 4
    q.put((rank, ok, per_step))
    D.shutdown()


def test_one_collective_per_step_world2_gloo():
    for total in (6, 5):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 33500 + (os.getpid() % 2000) + total
        procs = [ctx.Process(target=_step_worker, args=(r, 2, port, total, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(ok for _, ok, _ in res), res


def test_one_collective_per_step_world8_gloo():
    """the world size the driver launches (8 ranks: configs[4] = 8 x 64 images): shard sizes exchanged once, ONE collective per step, even
    (16 = 8 x 2) and uneven (13 = 5 x 2 + 3 x 1) shards, every rank ends with the full id matrix"""
    for total in (16, 13):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 37500 + (os.getpid() % 2000) + total
        procs = [ctx.Process(target=_step_worker, args=(r, 8, port, total, q)) for r in range(8)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert len(res) == 8 and all(ok for _, ok, _ in res), res


def _resharding_worker(rank, world, port, totals, q):
    """the global batch changes between calls such that ONE rank's shard keeps its size and the other's does not
    (5 -> (3,2), 6 -> (3,3), 4 -> (2,2), 5 again).  Every rank must issue the same collectives and get the right matrix each time."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    ok = True
    for j, total in enumerate(totals):
        lo, hi = D.shard_range(total, r, w)
        all_ids = torch.from_numpy(synth.synthetic_token_ids(total, first_index=3 * j))
        out = D.all_gather_ids(all_ids[lo:hi])
        ok = ok and tuple(out.shape) == (total, 512) and bool(torch.equal(out, all_ids))
        out2, ms = D.all_gather_ids_timed(all_ids[lo:hi])
        ok = ok and bool(torch.equal(out2, all_ids)) and ms >= 0.0
    q.put((rank, ok))
    D.shutdown()


def test_all_gather_ids_survives_a_changing_global_batch_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_resharding_worker, args=(r, 2, port, (5, 6, 4, 5, 7, 1), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _single_rank_worker(port, q):
    """the forced single-rank group: every helper takes its collective path with ONE rank (what the -m gpu test does over RCCL)"""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert D.init_from_env("gloo") == (0, 1, 0) and not torch.distributed.is_initialized()      # one rank: no group by default
    D.init_from_env("gloo", single_rank_group=True)
    ids = torch.from_numpy(synth.synthetic_token_ids(3))
    ok = D.all_gather_ids(ids) is ids and D.backend_name() is None                              # not forced: pass-through
    prev = D.force_single_rank(True)
    g = D.IdGatherer(3, 512, "cpu")
    g.launch(ids, timed=True)
    out = g.wait()
    ok = ok and prev is False and g.active and g.collectives == 1 and out is not ids and bool(torch.equal(out, ids))
    ok = ok and g.payload_bytes == 3 * 512 * 4 and D.backend_name() == "gloo"
    ok = ok and bool(torch.equal(D.all_gather_ids(ids.to(torch.int16)), ids.to(torch.int16)))
    t = torch.arange(4.0)
    ok = ok and bool(torch.equal(D.all_reduce_sum_(t.clone()), t)) and bool(torch.equal(D.all_gather_rows(t[:, None], [4]), t[:, None]))
    ok = ok and bool(torch.equal(D.broadcast_(t.clone()), t)) and D.max_over_ranks(2.5, "cpu") == 2.5
    D.barrier()
    D.force_single_rank(False)
    D.shutdown()
    q.put(ok)


def test_forced_single_rank_group_takes_the_collective_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(36500 + (os.getpid() % 2000), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


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


def _bench(*argv, env=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_n_spawns_n_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun around it, the form VERDICT r1 found broken) must start 2 ranks, gather ids
    over the process group and say so in its JSON; `--selftest-dist` stops before any GPU work."""
    rc, line, err = _bench("--gpus", "2", "--selftest-dist", "--batch", "3", env={"SELFTOK_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo" and line["allgather_ok"] is True
    assert line["allgather_bytes"] == 2 * 3 * 512 * 4 and line["allgather_ms"] > 0
    rc, line, _ = _bench("--gpus", "1", "--selftest-dist", "--batch", "2")
    assert rc == 0 and line["n_gpus"] == 1 and line["backend"] is None
    # the driver's world size: 8 ranks respawned by `python bench.py --gpus 8`, one gather with world = 8
    rc, line, err = _bench("--gpus", "8", "--selftest-dist", "--batch", "2", env={"SELFTOK_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["allgather_ok"] is True and line["allgather_bytes"] == 8 * 2 * 512 * 4


def test_bench_refuses_world_size_mismatch():
    """a launcher that starts fewer ranks than --gpus says must fail loudly, not print n_gpus: 1"""
    rc, line, err = _bench("--gpus", "2", "--selftest-dist", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=1" in err


def test_bench_launch_command_is_one_rank_per_gpu():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cmd = m.launch_command(8, ["--gpus", "8", "--steps", "3"], 29999)
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")


def _vq_worker(rank, world, port, q):
    """data-parallel codebook update: each rank holds half of the batch, all-reduces bins / embed_sum / one-hot means over gloo
    (what distributed.all_reduce does in the reference, vector_quantize_pytorch.py:573,588,594) -> every rank must end with the
    state a single process computes from the whole batch"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env("gloo")
    from oracle import vq_train as VT
    C, Dm, K, B = 512, 16, 8, 8
    embed0 = VT.l2norm(synth.hash_normalish(0xE0, (C, Dm)))
    x = VT.l2norm(synth.hash_normalish(0xE1, (B, K, Dm)))
    st = VT.new_state(embed0, K)
    lo, hi = D.shard_range(B, rank, world)
    ids = VT.train_step(st, x[lo:hi], 0.99, world=world, all_reduce=D.all_reduce_sum_)
    full = VT.new_state(embed0, K)
    ids_full = VT.train_step(full, x, 0.99)
    ok = bool(torch.equal(ids, ids_full[lo:hi]))
    for name in ("embed", "embed_avg", "cluster_size", "timestep_p_over_c"):
        ok = ok and float((st[name] - full[name]).abs().max()) < 1e-6
    t = torch.full((3,), float(rank))
    D.broadcast_(t, src=1)
    ok = ok and bool((t == 1.0).all()) and D.world_size() == world
    q.put((rank, ok))
    torch.distributed.destroy_process_group()


def test_codebook_update_is_data_parallel_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_vq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _eval_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch
    from selftoktokenizer_amd import dist as D, evaluate as E
    D.init_from_env("gloo")

    class FakePipe:                                            # the harness only needs .device, .encoding, .decoding
        device = torch.device("cpu")

        def encoding(self, imgs, device=None):
            return (imgs.reshape(imgs.shape[0], -1)[:, :8] * 1000).long()

        def decoding(self, ids, device=None, noise=None):
            return ((self._last + 1.0) / 2.0 * 0.9).to(torch.bfloat16)

    pipe = FakePipe()

    def load(lo, hi):
        g = torch.Generator().manual_seed(123)
        allx = torch.rand(11, 3, 8, 8, generator=g) * 2 - 1
        pipe._last = allx[lo:hi]
        return allx[lo:hi]
    res = E.evaluate(pipe, load, 11, batch=2, seed=None)
    q.put((rank, res))
    D.shutdown()


def test_eval_harness_shards_and_gathers_over_ranks():
    """evaluate(): 11 images over 2 gloo ranks (shards of 6 and 5, batches of 2) = the single-process per-image PSNR list, on every rank"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_eval_worker, args=(0, 1, _free_port(), q1))
    p1.start()
    single = q1.get(timeout=300)[1]
    p1.join(timeout=60)
    assert got[0]["shard"] == [0, 6] and got[1]["shard"] == [6, 11] and got[0]["ranks"] == 2
    assert got[0]["diffusion"]["psnr_each_dB"] == got[1]["diffusion"]["psnr_each_dB"] == single["diffusion"]["psnr_each_dB"]
    assert len(single["diffusion"]["psnr_each_dB"]) == 11
