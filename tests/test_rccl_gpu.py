"""-m gpu: RCCL itself, on the one GPU a test box has.  A single rank needs no exchange, so dist.py's helpers pass through at
world_size 1 and the communicator set-up, the side stream + event join, the int64 -> int32 cast and the collectives had only ever run
under gloo on CPU buffers (missing 2).  Here a ONE-rank "nccl" (= RCCL) group is created exactly as an N-rank one
(`device_id=` binding, 127.0.0.1 rendezvous) and `force_single_rank` makes every helper take its device path: the first N-GPU run
then executes nothing for the first time except the xGMI transport."""
import os
import socket

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import dist as D, synth

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_single_rank():
    import torch.distributed as dist
    assert not dist.is_initialized()
    saved = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") == "0"
    rank, world, local = D.init_from_env("nccl", single_rank_group=True)
    assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
    prev = D.force_single_rank(True)
    yield dist
    D.force_single_rank(prev)
    D.shutdown()
    assert not dist.is_initialized()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_id_gatherer_device_path_over_rccl(rccl_single_rank):
    dev = torch.device("cuda", 0)
    ids = torch.from_numpy(synth.synthetic_token_ids(64)).to(dev)                 # [64,512] int64, what encoding() returns
    g = D.id_gatherer(64, 512, dev)                                               # collective set-up: shard sizes over RCCL
    assert g.active and g.stream is not None and g.counts == [64] and g.recv.is_cuda and g.recv.dtype == torch.int32
    assert g.payload_bytes == 64 * 512 * 4 and D.backend_name() == "nccl"
    for step in range(3):                                                         # side stream, event join, buffers re-used
        cur = (ids + step) % 32768
        g.launch(cur, timed=True)
        busy = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)     # the caller's stream works meanwhile
        out = g.wait()
        assert out.dtype == torch.int64 and out.device == cur.device and out.data_ptr() != cur.data_ptr()
        assert torch.equal(out, cur)
        ms = g.last_ms()
        assert 0.0 < ms < 1000.0, ms
        del busy
    assert g.collectives == 3
    # the convenience entry points (sizes re-agreed per call) and narrower wire dtypes
    assert torch.equal(D.all_gather_ids(ids.to(torch.int32)), ids.to(torch.int32))
    out, ms = D.all_gather_ids_timed(ids[:5])
    assert torch.equal(out, ids[:5]) and ms > 0.0
    assert torch.equal(D.all_gather_ids(ids[:7]), ids[:7])                        # another shard size on the same group


def test_barrier_reductions_and_row_gather_over_rccl(rccl_single_rank):
    dev = torch.device("cuda", 0)
    D.barrier()                                                                   # barrier(device_ids=[...])
    t = torch.arange(1024, device=dev, dtype=torch.float32)
    assert torch.equal(D.all_reduce_sum_(t.clone()), t)
    rows = torch.randn(37, 16, device=dev)
    assert torch.equal(D.all_gather_rows(rows, [37]), rows)
    assert torch.equal(D.broadcast_(t.clone(), src=0), t)
    assert D.max_over_ranks(1.25, dev) == 1.25
    torch.cuda.synchronize()


def test_codebook_training_step_over_rccl(rccl_single_rank):
    """f4: the data-parallel code-book update issues its id gather and both all-reduces on the RCCL group; with one rank the state
    must equal the un-distributed step (bit for bit where the arithmetic is order independent)"""
    from selftoktokenizer_amd.vq_train import CodebookEMA, l2norm
    C, K, B = 2048, 32, 16
    embed0 = l2norm(synth.hash_normalish(0xE0, (C, 16))).cuda()
    z = (synth.hash_normalish(0x7A11, (B, K, 16)) * 1.5).cuda()
    with_group = CodebookEMA(embed0, K, decay=0.99, threshold_ema_dead_code=0.0)
    q1, ids1, _ = with_group.step(z)
    prev = D.force_single_rank(False)                                             # the same step with the helpers passing through
    try:
        alone = CodebookEMA(embed0, K, decay=0.99, threshold_ema_dead_code=0.0)
        q2, ids2, _ = alone.step(z)
    finally:
        D.force_single_rank(prev)
    assert torch.equal(ids1, ids2) and torch.equal(q1, q2)
    for name in ("cluster_size", "timestep_p_over_c"):                            # integer-valued sums: exact whatever the order
        assert torch.equal(getattr(with_group, name), getattr(alone, name)), name
    for name in ("embed", "embed_avg"):                                           # fp32 scatter-adds (atomics): order noise between two runs
        torch.testing.assert_close(getattr(with_group, name), getattr(alone, name), rtol=0, atol=2e-6)
    assert np.isfinite(float(with_group.delta_embed))


def test_bench_under_torchrun_one_rank_takes_the_rccl_path():
    """the command the driver issues for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), with the ONE rank a test
    box has: torchrun's env is read, the RCCL group is created (`backend: "nccl"`), every step issues its all-gather on the device path
    (`allgather_ms` non-null) and rank 0 prints the JSON line.  What an 8-GPU run adds is the xGMI transport inside ncclAllGather."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--decode-steps", "2", "--force-collective",
           "--no-cpu-baseline", "--no-token-check", "--no-kernel-roofs", "--no-latency", "--no-other-gemm", "--tune-gemm", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print("\n[bench under torchrun, 1 rank, RCCL path forced]", {k: line[k] for k in ("n_gpus", "ranks", "backend", "allgather_bytes", "allgather_ms", "ms_per_step", "value")})
    assert line["n_gpus"] == 1 and line["ranks"] == 1 and line["backend"] == "nccl"
    assert line["allgather_bytes"] == 4 * 512 * 4 and line["allgather_ms"] is not None and 0.0 < line["allgather_ms"] < 1000.0
    assert line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0 and "INVALID" in line["config"]     # --decode-steps: a debug line, marked as such
