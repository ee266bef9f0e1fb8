#!/usr/bin/env python
"""bench.py -- images/sec of the Selftok encode + 50-step decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input per GPU: `encoding` (SD3-VAE encode ->
Q-Former encoder -> VQ nearest-code) then `decoding` (50-step rectified-flow MMDiT loop -> SD3-VAE decode) of
B = 64 images of 256x256 with the 512-token tokenizer (BASELINE.json configs[1]).  Inputs (images, decode noise,
weights) are resident in HBM before the timed region.  Weights are hash-generated (no checkpoints offline) with
the architecture of the published 512-token model; arithmetic is the parity path: fp32 tokenizer/DiT, bf16 VAE.

N > 1: batch sharding (weak scaling, 64 images per GPU), one RCCL all-gather of the token ids per step.

Extra objects on the JSON line:
  roofline     : the VQ nearest-code kernel (vq_mfma_kernel), timed live with HIP events on its launch stream.
  cpu_baseline : oracle/ (our CPU restatement, verified equal to the reference) on this box's host cores, rank 0,
                 N=1 only, on a bounded sample (see "sample").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--tokens", type=int, default=512, choices=[512, 1024])
    ap.add_argument("--decoder", default="diffusion", choices=["diffusion", "renderer"])
    ap.add_argument("--decode-steps", type=int, default=None, help="debug only: truncate the 50-step loop (marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm", default=None, choices=["fp32", "f16x2"], help="arithmetic of the MMDiT block Linears")
    return ap.parse_args()


def host_cores() -> int:
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota) -- the GPU boxes expose 256 hardware
    threads but cap the container at 16 CPUs; spawning 256 threads there is 50x slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sd_gpu, vsd_gpu, cfg, K):
    """oracle/ on the host cores: B=1, full encode (VAE-enc + Q-Former + VQ), 2 of the 50 decode steps (every step has
    identical FLOPs up to the shrinking context, extrapolated x25), full VAE decode."""
    from oracle import model as OM, schedule as OS
    from selftoktokenizer_amd import synth, weights as W
    torch.set_num_threads(host_cores())
    sd = {k: v.detach().cpu() for k, v in sd_gpu.items()}
    vsd = {k: v.detach().cpu() for k, v in vsd_gpu.items()}
    images = synth.synthetic_images(1)
    stages, kps = OS.parse_stages(cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    with torch.no_grad():
        enc_tables = OM.encoder_tables(sd, K)
        dit_tables = OM.dit_ctx_tables(sd, K)
        t0 = time.perf_counter()
        ids = OM.pipeline_encode(sd, vsd, images, enc_tables)
        t_enc = time.perf_counter() - t0
        noise = synth.synthetic_noise(1)
        t0 = time.perf_counter()
        lat = OM.decode_latent(sd, ids, noise, stages, kps, 50, dit_tables, max_steps=2)
        t_2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        OM.norm_ip(OM.vae_decode(vsd, OM.process_out(lat).to(torch.bfloat16)))
        t_vd = time.perf_counter() - t0
    total = t_enc + 25.0 * t_2 + t_vd
    return {"value": round(1.0 / total, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B=1 256x256 K={K}: full encode {t_enc:.2f}s + 2 of 50 decode steps {t_2:.2f}s (x25 extrapolated) "
                      f"+ VAE decode {t_vd:.2f}s on {torch.get_num_threads()} host threads; oracle/ = lean restatement "
                      "(no redundant encoder passes / table recomputes of the reference)"}


def fp32_flops_per_image(K: int, k_table, renderer: bool) -> float:
    """analytic fp32 matrix FLOPs of one image through encode + decode as THIS implementation executes them
    (context truncated to k+1 live tokens, adaLN tables precomputed); the bf16 VAE (0.89 TFLOP/img) is not included."""
    H, Hq, Hx, nx = 1536, 512, 64, 256
    enc_blk = (nx * (Hx * 3 * Hx + Hx * 2 * Hq + Hx * Hx + 2 * Hx * 4 * Hx) + K * (Hq * 3 * Hq + Hq * Hq + 2 * Hq * 4 * Hq)) * 2 \
        + 2 * 2 * 4 * nx * nx * 16 + 2 * 2 * 8 * K * (nx + K) * 64
    enc = 16 * enc_blk + K * Hq * 16 * 2 + 2.0 * K * 32768 * 16

    def dit_pass(n):
        lin = 23 * n * 12 * H * H * 2 + n * 2 * H * H * 2 + 24 * nx * 12 * H * H * 2
        ada = (24 * 6 + 2 + 2) * H * H * 2 + 2 * H * H * 2
        att = 23 * 4 * 24 * 64 * (n + nx) * (n + nx) + 4 * 24 * 64 * nx * (n + nx)
        return lin + ada + att + nx * H * 64 * 2 + nx * 64 * H * 2
    dec = dit_pass(K) if renderer else sum(dit_pass(int(k) + 1) for k in k_table)
    return float(enc + dec)


def main():
    args = parse()
    from selftoktokenizer_amd import dist as D, ops, synth, weights as W
    from selftoktokenizer_amd.config import default_config
    from selftoktokenizer_amd.pipeline import SelftokPipeline

    # SELFTOK_DIST_BACKEND=gloo + SELFTOK_ONE_GPU=1: dry run of the N>1 flow with every rank on GPU 0 (single-GPU boxes)
    one_gpu = os.environ.get("SELFTOK_ONE_GPU") == "1"
    if one_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = D.init_from_env(os.environ.get("SELFTOK_DIST_BACKEND", "nccl"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    K, B = args.tokens, args.batch
    renderer = args.decoder == "renderer"
    cfg = default_config(K, renderer=renderer)
    sd = W.synthetic_state_dict(W.expected_shapes(K, renderer=renderer), device=dev)
    vsd = W.synthetic_vae_state_dict(device=dev)
    pipe = SelftokPipeline(cfg, None, None, device=dev, state_dict=sd, vae_state_dict=vsd, verbose=False, gemm=args.gemm)

    images = synth.synthetic_images(B, device=dev, first_index=rank * B)          # resident in HBM
    noise = synth.synthetic_noise(B, device=dev, first_index=rank * B)
    torch.cuda.synchronize()

    # ---- live timing of the dominant hand-written kernel of the north star (VQ argmax) ----
    vq_events = []

    def encode_tokens(x0):
        z = pipe.model.encoder.features(x0)
        ids, launch_main, launch_fin = ops.vq_encode_split_launch(z, pipe.model.encoder.codebook_packed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # torch's current stream == the stream the kernel is launched on
        launch_main()
        e1.record()
        launch_fin()
        vq_events.append((e0, e1))
        return ids

    def step():
        x0 = pipe.encode_latents(images)
        ids = encode_tokens(x0)                                  # [B,K] int64 on device
        ids_all = D.all_gather_ids(ids)                          # RCCL all-gather (no-op at N=1)
        lo = rank * B
        mine = ids_all[lo:lo + B]
        if renderer:
            return pipe.decoding_with_renderer(mine)
        return pipe.decoding(mine, noise=noise, max_steps=args.decode_steps)

    for _ in range(args.warmup):
        step()
    vq_events.clear()
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)

    if rank != 0:
        D.shutdown()
        return
    n_vq = B * K
    vq_ms = float(np.mean([a.elapsed_time(b) for a, b in vq_events])) if vq_events else float("nan")
    flops = 2.0 * n_vq * 32768 * 16
    alg_bytes = 4.0 * n_vq * 16 + 4.0 * 32768 * 16 + 8.0 * n_vq          # z + codebook (once) + int64 ids
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "vq_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"N{n_vq}")
        except Exception:
            traffic = None
    roof = {"kernel": "vq_mfma_kernel", "bound": "mfma", "achieved": round(flops / (vq_ms * 1e-3) / 1e12, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / (vq_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic, "avg_launch_ms": round(vq_ms, 4), "launches": len(vq_events),
            "algorithmic_flops": flops, "algorithmic_bytes": alg_bytes,
            "hbm_achieved_GBs": round(alg_bytes / (vq_ms * 1e-3) / 1e9, 2), "hbm_frac": round(alg_bytes / (vq_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
            "note": "N*C*D = %d x 32768 x 16 fp32 FMA chain is ~7.9 kFLOP/B: matrix-core bound, not HBM bound (SURVEY.md 8d)" % n_vq}
    line = {
        "metric": "images/sec encode+decode, 256x256 %d-token" % K, "value": round(world * B * args.steps / elapsed, 4),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: batch %d x 256x256 per GPU, %d-token encode + %s decode"
                               % (3 if renderer else (2 if K == 1024 else 1), B, K, "one-step renderer" if renderer else "50-step diffusion"),
                   "global_batch": world * B, "tokens": K, "decode_steps": 1 if renderer else (args.decode_steps or 50),
                   "arithmetic": "fp32 Q-Former/VQ/MMDiT, bf16 SD3-VAE (reference dtypes)", "parallelism": "batch-shard x%d" % world,
                   "weights": "hash-generated, architecture of tokenizer_512_ckpt"},
        "roofline": roof,
    }
    fl_img = fp32_flops_per_image(K, pipe.k_table[: (args.decode_steps or 50)], renderer)
    job_tf = fl_img * world * B * args.steps / elapsed / 1e12
    line["job_roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "fp32_tflop_per_image": round(fl_img / 1e12, 2),
                            "achieved": round(job_tf, 1), "peak": FP32_MFMA_PEAK_TFLOPS * world,
                            "frac": round(job_tf / (FP32_MFMA_PEAK_TFLOPS * world), 4),
                            "note": "fp32 matrix FLOPs actually executed (encoder + MMDiT, context truncated to live tokens) / wall time; "
                                    "bf16 VAE work (0.89 TFLOP/img) excluded"}
    if args.decode_steps is not None and not renderer:
        line["config"]["INVALID"] = "decode loop truncated with --decode-steps (debug run)"
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd, vsd, cfg, K)
    print(json.dumps(line), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
