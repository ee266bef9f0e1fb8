#!/usr/bin/env python
"""bench.py -- images/sec of the Selftok encode + 50-step decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input per GPU through the PUBLIC drop-in API:
`tokens = pipe.encoding(images)` (SD3-VAE encode -> Q-Former encoder -> VQ nearest-code), the id all-gather, then
`pipe.decoding(tokens.cpu().numpy())` (50-step rectified-flow MMDiT loop -> SD3-VAE decode) of B = 64 images of
256x256 with the 512-token tokenizer (BASELINE.json configs[1]).  Images are resident in HBM before the timed
region; the decode noise is drawn inside `decoding` from the global CPU generator exactly as the reference does
(SelftokPipeline.py:264).  Weights are hash-generated (no checkpoints offline) with the architecture of the
published 512-token model; arithmetic is the parity path: fp32 tokenizer/DiT, bf16 VAE.

N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per GPU); the driver's
own torchrun launch is used as is.  Batch sharding (weak scaling, 64 images per GPU), one RCCL all-gather of the
token ids per step (event-timed, reported), every rank decodes its slice of the GATHERED id matrix.

Extra objects on the JSON line:
  roofline          : the VQ nearest-code kernel (vq_f16_kernel, the dominant launch of the VQ path), timed live with HIP events on
                      its launch stream inside the timed steps: frac = EXECUTED f16-MFMA FLOPs / kernel time / the f16 matrix peak.
  roofline_kernels  : the other kernels a step is made of (attention, LN+modulate, fp32 GEMM, f16x2-split GEMM), event-timed
                      stand-alone at the shapes of the timed step.
  token_match       : token-id exact match of the timed batch against the CPU oracle (kernel boundary + end to end).
  parity_16         : 16 images against the REFERENCE's own pipeline run (tests/golden/pipeline_b16.npz): id match, the reference
                      top-1/top-2 gap of every flip, reconstruction-PSNR deltas (end to end and same-decoder), next to the
                      CPU-oracle-vs-reference numbers on the same images.
  parity_64         : configs[1] at its configured batch: the 64 ids + noise of the reference's own one-batch run through gemm fp32 AND f16x2 (the modes the
                      images/s are measured in): final-latent deltas, per-image |delta PSNR| end to end and same-decoder, the metric's floor beside them.
  latency_b1        : (--latency) BASELINE configs[0]-style single image: encode + 50-step decode, eager vs hipGraph replay.
  gemm_modes        : the same step with the MMDiT Linears on the other GEMM arithmetic (fp32 library <-> f16x2 split).
  cpu_baseline      : oracle/ (our CPU restatement, verified equal to the reference) on this box's host cores, rank 0,
                      N=1 only, on a bounded sample (see "sample").
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: bf16/f16 MFMA, dense
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--tokens", type=int, default=512, choices=[512, 1024])
    ap.add_argument("--decoder", default="diffusion", choices=["diffusion", "renderer"])
    ap.add_argument("--decode-steps", type=int, default=None, help="debug only: truncate the 50-step loop (marks the line invalid)")
    ap.add_argument("--gemm", default=None, choices=["fp32", "f16x2", "exact"], help="arithmetic of the MMDiT of the headline number (exact: every operation in the "
                    "reference's torch-CPU order -- pixels bit-equal to the reference's, the parity mode)")
    ap.add_argument("--no-exact", action="store_true", help="skip the timed step and the 16-image pixel-equality check in the exact-order MMDiT mode")
    ap.add_argument("--vae", default=None, choices=["exact", "parity", "miopen", "fast"], help="VAE arithmetic (vae.AutoencoderKLGPU); default: the pipeline's (exact-order encoder at 256 x 256)")
    ap.add_argument("--tune-gemm", type=int, default=1, choices=[0, 1], help="fp32 Linears: hipBLASLt kernel chosen per shape family by measurement (gemm_tune.py, opt-in in the "
                    "pipeline; the bench asks for it explicitly, before the warm-up, and reports the kernels in config.fp32_linear_kernels); 0: hipBLASLt's own choice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-token-check", action="store_true")
    ap.add_argument("--no-kernel-roofs", action="store_true")
    ap.add_argument("--latency", action="store_true", help="also run the B = 1 eager / hipGraph latency leg (BASELINE configs[0]-style single image; off by default since "
                    "round 6: the default line carries the 64-image parity leg instead and stays inside the driver's time)")
    ap.add_argument("--no-latency", action="store_true", help=argparse.SUPPRESS)     # accepted for old command lines: the leg is opt-in now
    ap.add_argument("--no-parity16", action="store_true", help="skip the 16-image parity leg against the reference pipeline's run")
    ap.add_argument("--no-parity64", action="store_true", help="skip the 64-image parity leg (configs[1] at its configured batch, gemm fp32 / f16x2 vs the reference's one-batch run)")
    ap.add_argument("--vae-decode", default=None, choices=["exact", "parity"], help="VAE decoder arithmetic alone (default: the pipeline's -- parity, and exact in gemm='exact')")
    ap.add_argument("--no-other-gemm", action="store_true", help="skip the second measurement on the other GEMM arithmetic")
    ap.add_argument("--encoder", default=None, choices=["exact", "fast"], help="Q-Former encoder arithmetic; default: the pipeline's (exact: the reference's torch-CPU orders)")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: also run the second-arithmetic / kernel-roofline / parity / latency legs (default at N > 1: only the "
                    "timed steps -- rank 0 would otherwise work for minutes after the other ranks have left the group)")
    ap.add_argument("--force-collective", action="store_true", help="N = 1 under torchrun: create the RCCL group for the one rank and take the all-gather's device path "
                    "(communicator bound to the GPU, int32 cast, side stream, event join) exactly as an N-rank run does -- the 8-GPU line minus the xGMI transport")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="no GPU work: initialise the ranks, all-gather synthetic ids, print the JSON skeleton (CPU test of the N>1 entry)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# N > 1 entry: `python bench.py --gpus N` spawns N ranks itself
# ---------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n: int, argv, port: int):
    """the torchrun command line `python bench.py --gpus n ...` re-executes itself under (one process per GPU)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_respawn(args, argv) -> None:
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this host driver
        rc = subprocess.call(launch_command(args.gpus, argv, free_port()), env=env)
        sys.exit(rc)


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
    """oracle/ on the host cores.  Headline: B=1, full encode (VAE-enc + Q-Former + VQ), 2 of the 50 decode steps (every
    step has identical FLOPs up to the shrinking context, extrapolated x25), full VAE decode.  Breadth:
    encode at B=8, the VQ nearest-code lookup alone on one thread (N = 512, 8192) and on all threads (N = B K of configs[1..3]).  About 27 s of CPU work."""
    import torch
    from oracle import clib, model as OM, schedule as OS
    from selftoktokenizer_amd import synth
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
        t0 = time.perf_counter()
        OM.pipeline_encode(sd, vsd, synth.synthetic_images(8), enc_tables)
        t_enc8 = time.perf_counter() - t0
        # (B = 64, configs[1]'s batch, was timed here through round 6's first runs: 1.691 images/s against 1.689 at B = 8 -- 38 s of a 79 s leg for the same figure; dropped)
        cb = sd["encoder.quantizer._codebook.embed"][0].numpy()
        vq, vq_mt = {}, {}
        for n in (512, 8192):                                                          # one thread: 1730 / 1756 rows/s at N = 512 / 32768 (round 6): the rate does not depend on N
            z = synth.synthetic_vq_rows(n, seed=0xBE0C).numpy()
            t0 = time.perf_counter()
            clib.vq_encode(z, cb)
            vq[f"N{n}_rows_per_s"] = round(n / (time.perf_counter() - t0), 1)
        for n in (32768, 65536, 131072):                                               # N = B K of configs[1], [2], [3]: all host threads over row chunks
            z = synth.synthetic_vq_rows(n, seed=0xBE0C).numpy()
            t0 = time.perf_counter()
            clib.vq_encode_mt(z, cb)
            vq_mt[f"N{n}_rows_per_s"] = round(n / (time.perf_counter() - t0), 1)
    total = t_enc + 25.0 * t_2 + t_vd
    # the same work through the build's C ABI compiled for the CPU (oracle/libselftok_cpu.so) + torch-CPU GEMMs: the product's own host
    # modules with every kernel call going to the CPU twin (SURVEY 8d's "libselftok_cpu.so + torch-CPU GEMMs"; oracle/twin_host.py)
    twin = None
    try:
        from oracle import twin_host as TH
        with torch.no_grad(), TH.on_cpu_twin():
            enc, dit, flow, ktab = TH.build(sd, K)
            x0 = OM.process_in(OM.vae_encode_mean(vsd, images.to(torch.bfloat16))).to(torch.float32)
            t0 = time.perf_counter()
            _, ids_t = enc(x0, d=None)
            t_tok = time.perf_counter() - t0
            ehs = enc.codes_ln(ids_t)
            t0 = time.perf_counter()
            flow.p_sample_loop(dit, noise, ehs, ktab, context_see_xt=True, max_steps=1)
            t_step = time.perf_counter() - t0
        twin = {"kind": "port (include/selftok_hip.h compiled for the CPU: oracle/libselftok_cpu.so, scalar C + OpenMP, + torch-CPU GEMMs)",
                "tokenizer_encode_B1_s": round(t_tok, 2), "one_decode_step_B1_s": round(t_step, 2),
                "ids_equal_to_the_torch_port": int((ids_t.numpy() == ids.numpy()).sum()), "of": int(ids.numel()),
                "note": "slower than the torch port above (the twin's attention / LayerNorm kernels are checkers, not tuned CPU code): the torch port stays "
                        "the baseline `value`; both are this build's restatement, not the reference"}
    except Exception as e:                        # noqa: BLE001 -- a baseline leg must never take the bench line down
        twin = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"value": round(1.0 / total, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B=1 256x256 K={K}: full encode {t_enc:.2f}s + 2 of 50 decode steps {t_2:.2f}s (x25 extrapolated) "
                      f"+ VAE decode {t_vd:.2f}s on {torch.get_num_threads()} host threads; oracle/ = lean restatement "
                      "(no redundant encoder passes / table recomputes of the reference)",
            "encode_images_per_s": {"B1": round(1.0 / t_enc, 3), "B8": round(8.0 / t_enc8, 3)},
            "vq_lookup_scalar_C_1_thread": vq, "vq_lookup_scalar_C_all_threads": vq_mt, "twin": twin,
            "kinds": "every figure here is the build's own CPU restatement (oracle/: torch-CPU GEMMs / convolutions + oracle/libselftok_oracle.so for the "
                     "VQ lookup) on this box's host cores -- 'port'; the reference itself was timed only in the build container (cpu_baseline_reference_survey)"}


REFERENCE_SURVEY_BASELINE = {
    "value": round(1.0 / 166.5, 5), "unit": "images/s", "cores": 8, "kind": "reference",
    "sample": "the reference itself (mimogpt.infer.SelftokPipeline, fp32 DiT/bf16 VAE, incl. its 50 redundant encoder passes and per-call "
              "table recomputes) timed ONCE in the build container (8 vCPU) for SURVEY.md section 6 / BASELINE.md: encoding B=1 0.47 s, "
              "one MMDiT.forward 2.82 s, VAE decode 0.35 s -> 166.5 s per image.  It cannot travel to the GPU box; not re-measured here."}


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


def event_time_ms(fn, n=40, warm=5):      # 40 launches after 5 warm-ups: the clock of a cold chip drifts by several % over the first launches
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


VQ_TRAFFIC_SOURCES = ("selftoktokenizer_amd/csrc/vq.hip", "selftoktokenizer_amd/csrc/common.h")


def source_stamp(rel_paths=VQ_TRAFFIC_SOURCES) -> str:
    """sha256 (16 hex) over the kernel sources a PMC measurement belongs to: a traffic figure is only quoted for the build it was taken on"""
    import hashlib
    h = hashlib.sha256()
    for rel in rel_paths:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_vq_traffic(n_vq: int, coarse: bool, path=None):
    """HBM-side bytes per VQ launch (main + finalize kernel) from the rocprofv3 PMC passes of THIS build
    (tools/pmc_vq_traffic.sh -> profiles/vq_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, per-kernel correction factors
    calibrated by tools/microbench/fetch_calib.hip in the same passes).  Returns (bytes or None, note): None when there is no
    measurement for this (N, kernel) or when the file's source stamp is not the stamp of the sources in this tree."""
    path = path or os.path.join(ROOT, "profiles", "vq_traffic.json")
    key = f"N{n_vq}_f16" if coarse else f"N{n_vq}"
    try:
        d = json.load(open(path))
    except Exception as e:                       # noqa: BLE001
        return None, f"no traffic measurement ({type(e).__name__})"
    now = source_stamp()
    if d.get("source_stamp") != now:
        return None, f"profiles/vq_traffic.json was measured on sources {d.get('source_stamp')}, this tree is {now}: stale, not quoted"
    if key not in d:
        return None, f"profiles/vq_traffic.json has no entry {key}"
    return int(d[key]), f"rocprofv3 PMC, sources {now}: {d.get('method', '')}"


def vq_roofline(n_vq, C, Dm, main_ms, fin_ms, launches, traffic, traffic_note, fp32_main_ms=None, fp32_fin_ms=None, ids_bytes=8, mfmas=3):
    """the `roofline` object of the JSON line, from measured kernel times.  Dominant kernel = vq_f16_kernel (the coarse pass: 3
    v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 block of the score matrix).  `achieved` / `frac` (= `frac_algorithmic`): the ALGORITHMIC
    FLOPs per launch (2NCD) / its average launch duration against the dense f16 matrix peak, as the contract defines them;
    `achieved_executed` / `frac_executed`: the f16-MFMA FLOPs the kernel issues (3 x 2NCD) over the same time = pipe utilisation."""
    flops = 2.0 * n_vq * C * Dm                                   # the reference's fp32 score matrix
    executed = float(mfmas) * flops                               # f16 MFMA FLOPs the coarse kernel issues: 1 (hi*hi) or 3 (hi*hi + hi*lo + lo*hi) per product
    alg_bytes = 4.0 * n_vq * Dm + 4.0 * C * Dm + float(ids_bytes) * n_vq   # z + codebook (once) + ids
    ach_exec = executed / (main_ms * 1e-3) / 1e12
    ach = flops / (main_ms * 1e-3) / 1e12
    both = main_ms + fin_ms
    roof = {"kernel": "vq_f16_kernel<RT,%d> (f16 coarse pass of the cosine argmax, %d MFMA%s per 32 x 32 scores; vq_finalize_f16_kernel re-scores the candidates "
                      "inside the proven error window in canonical fp32: ids and top-1 score bits equal the fp32 kernels')" % (mfmas, mfmas, "" if mfmas == 1 else "s"),
            "bound": "mfma", "achieved": round(ach, 2), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F16_MFMA_PEAK_TFLOPS, 4),
            "frac_algorithmic": round(ach / F16_MFMA_PEAK_TFLOPS, 4), "achieved_executed": round(ach_exec, 2), "frac_executed": round(ach_exec / F16_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_note": traffic_note,
            "traffic_over_algorithmic_bytes": (round(traffic / alg_bytes, 2) if traffic else None),
            "avg_launch_ms": round(main_ms, 4), "finalize_kernel_ms": round(fin_ms, 4), "both_launches_ms": round(both, 4), "launches": launches,
            "executed_flops_per_launch": executed, "mfmas_per_product": mfmas, "algorithmic_flops": flops, "algorithmic_bytes": alg_bytes,
            "timing": "HIP events on the launch stream around each of the two launches of every VQ call inside the timed steps",
            "fp32_equivalent": {"tflops": round(flops / (both * 1e-3) / 1e12, 2), "fp32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
                                "note": "2NCD of the reference's fp32 score matrix / (both launches); NOT a roofline fraction of this kernel (it runs on "
                                        "the f16 matrix cores) -- kept for comparison with the fp32 kernel below"},
            "hbm": {"achieved_GBs": round(alg_bytes / (both * 1e-3) / 1e9, 2), "frac_of_8TBs": round(alg_bytes / (both * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "note": "algorithmic bytes / both launches: ~7.7 kFLOP per byte at D = 16, the path is matrix bound by three orders of magnitude"},
            "note": "frac = frac_algorithmic = 2NCD (SURVEY 8d: the reference's fp32 score matrix) / avg_launch_ms / 2500 TFLOP/s, the f16 matrix peak the "
                    "kernel runs on; frac_executed = the f16 MFMA FLOPs it issues (%s) / the same "
                    "time and peak = pipe utilisation (N*C*D = %d x %d x %d); recompute from profiles/*kernel_stats.csv: the vq_f16_kernel row's average duration"
                    % ("1 per product: hi*hi; the exact fp32 re-score of the candidates keeps ids bit-exact" if mfmas == 1 else "3 per product: hi*hi + hi*lo + lo*hi", n_vq, C, Dm)}
    if mfmas == 1:
        # what binds the one-MFMA pass is its VALU issue, not the matrix pipe (measured OFFLINE in round 5, profiles/r5_vq_bound.txt; constants, not live):
        # per 32 x 32 scores one 32-cycle MFMA and 12 VALU instructions (48 issue cycles); the pattern sustains 59 cycles per tile with four waves per SIMD
        # (tools/microbench/mfma_f16_valu.hip) and the chip clocks 1.66 GHz under this kernel (in-kernel stamps, tools/sweep_vq_f16.py)
        tiles_per_simd = (n_vq / 32.0) * (C / 32.0) / 1024.0
        pattern_ms = tiles_per_simd * 59.0 / 1.66e9 * 1e3
        roof["issue_roof"] = {"valu_instructions_per_tile": 12, "mfma_cycles_per_tile": 32, "pattern_cycles_per_tile_4_waves_per_simd": 59, "shader_ghz_under_kernel": 1.66,
                              "pattern_ms": round(pattern_ms, 4), "frac_of_pattern": round(pattern_ms / main_ms, 4),
                              "ceiling_frac_of_f16_peak": round(32.0 / 59.0 * 1.66 / 2.4, 4),
                              "note": "offline constants (profiles/r5_vq_bound.txt), live avg_launch_ms: the kernel's own roof is the VALU issue of its tile-maximum scan; "
                                      "`frac` above stays against the f16 matrix peak the contract names"}
    if fp32_main_ms:
        a32 = flops / (fp32_main_ms * 1e-3) / 1e12
        roof["fp32_mfma_kernel"] = {"kernel": "vq_mfma_kernel<RT> (round-1 kernel: exact fp32 products on v_mfma_f32_32x32x2_f32; same ids, bit for bit)",
                                    "bound": "mfma(fp32)", "avg_launch_ms": round(fp32_main_ms, 4), "finalize_kernel_ms": round(fp32_fin_ms or 0.0, 4),
                                    "achieved": round(a32, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(a32 / FP32_MFMA_PEAK_TFLOPS, 4)}
    return roof


GOLD16 = os.path.join(ROOT, "tests", "golden", "pipeline_b16.npz")


def parity_16(pipe, exact_leg=True):
    """16 images against the REFERENCE's own SelftokPipeline run on the same synthetic weights (tests/golden/pipeline_b16.npz, made by
    tools/oracle/gen_golden.py pipeline16): token ids from pixels (through the bf16 VAE), the reference's top-1/top-2 gap of every
    flipped token, and reconstruction PSNR -- end to end, and with the reference's final latents and ours through the SAME decoder
    call -- next to what the CPU oracle (another implementation of the same bf16 VAE) gets on the same images."""
    import numpy as np
    import torch
    from selftoktokenizer_amd import synth
    g = np.load(GOLD16)
    ref = g["tokens"].astype(np.int64)
    B = ref.shape[0]
    dev = pipe.device
    dit_mode = pipe.model.model.gemm
    imgs = synth.synthetic_images(B, device=dev)
    x0 = pipe.encode_latents(imgs)
    z = pipe.model.encoder.features(x0)
    ids = pipe.model.encoder(x0, d=None)[1].cpu().numpy()
    mism = ids != ref
    x0_ref = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    dx = x0.cpu() - x0_ref
    zn = torch.nn.functional.normalize(z.cpu().reshape(-1, 16), dim=-1)
    zr = torch.nn.functional.normalize(torch.from_numpy(g["z"]).reshape(-1, 16), dim=-1)
    dz = (zn - zr).norm(dim=-1).reshape(B, -1).numpy()              # |delta of the unit feature| per token: a score moves by at most this
    mo = g["tokens_oracle"].astype(np.int64) != ref
    out = {"images": B, "reference": "mimogpt.infer.SelftokPipeline on CPU (fp32 tokenizer, bf16 SDVAE mirror), tests/golden/pipeline_b16.npz",
           "vae_mode": pipe.vae.mode, "vae_latents_bit_equal_to_reference": bool(torch.equal(x0.cpu(), torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float())),
           "ids_match_vs_reference": round(float(1.0 - mism.mean()), 6), "flips": int(mism.sum()), "tokens": int(mism.size),
           "flip_gaps_reference_top1_minus_top2": [round(float(v), 8) for v in np.sort(g["gap"][mism])],
           "flips_to_the_reference_runner_up": int((ids[mism] == g["id2"].astype(np.int64)[mism]).sum()),
           "unit_feature_delta_at_flips": [round(float(v), 8) for v in dz[mism][np.argsort(g["gap"][mism])]],
           "unit_feature_delta_median_max": [round(float(np.median(dz)), 8), round(float(dz.max()), 8)],
           "tokens_with_gap_below_2x_own_feature_delta": int((g["gap"] < 2.0 * dz).sum()),
           "vae_latent_delta_vs_reference_max_rms": [round(float(dx.abs().max()), 5), round(float(dx.pow(2).mean().sqrt()), 6)],
           "second_cpu_implementation_vs_reference": {"what": "oracle/ with the VAE attention projections as diffusers' F.linear instead of the mirror's 1x1 convolutions "
                                                              "(the oracle's default formulation is bit-identical to the reference)",
                                                      "ids_match": round(float(1.0 - mo.mean()), 6), "flips": int(mo.sum()),
                                       "flip_gaps": [round(float(v), 8) for v in np.sort(g["gap"][mo])]}}
    if not pipe.model.model.renderer:
        orig = (synth.synthetic_images(B) + 1.0) / 2.0

        def psnr_each(px):
            mse = ((px.float().cpu() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
            return (10.0 * torch.log10(1.0 / mse)).numpy()
        rec, lat = pipe.decoding(ref, noise=synth.synthetic_noise(B), return_latent=True)        # the reference's ids and noise, 50 steps
        lat_ref = torch.from_numpy(g["lat"]).to(dev)
        d_e2e = np.abs(psnr_each(rec) - g["psnr_ref"])
        both = pipe._to_pixels(torch.cat([lat_ref, lat]))                                        # one decoder call: only the latents differ
        d_same = np.abs(psnr_each(both[:B]) - psnr_each(both[B:]))
        d_or = np.abs(g["psnr_oracle"] - g["psnr_ref"])
        out["psnr"] = {"unit": "dB, reconstruction PSNR vs the original image, |ours - reference| per image", "reference_mean_dB": round(float(g["psnr_ref"].mean()), 4),
                       "end_to_end_delta_mean_max": [round(float(d_e2e.mean()), 6), round(float(d_e2e.max()), 6)],
                       "same_decoder_delta_mean_max": [round(float(d_same.mean()), 7), round(float(d_same.max()), 7)],
                       "second_cpu_implementation_vs_reference_delta_mean_max": [round(float(d_or.mean()), 6), round(float(d_or.max()), 6)],
                       "final_latent_maxdiff_vs_reference": round(float((lat - lat_ref).abs().max()), 8),
                       "note": "end to end = our latents through our bf16 VAE decoder (csrc/conv.hip + the exact-order attention block) vs the reference's pixels; same decoder = the reference's "
                               "final latents and ours through ONE call of our decoder (north star: 1e-3 dB); second cpu implementation = the CPU bf16 VAE with "
                               "diffusers' Linear attention projections on the reference's latents vs the reference's pixels: the spread between two CPU "
                               "implementations of the same bf16 network"}
        if exact_leg and dit_mode != "exact":
            import zlib
            gd = os.path.join(os.path.dirname(GOLD16), "decode_b16.npz")
            pipe.set_gemm("exact")
            try:
                rec_x, lat_x = pipe.decoding(ref, noise=synth.synthetic_noise(B), return_latent=True)
            finally:
                pipe.set_gemm(dit_mode)
            bits = rec_x.cpu().view(torch.int16).numpy().view(np.uint16)
            crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(B)], dtype=np.uint32)
            out["exact_mode"] = {"gemm": "exact", "final_latent_elements_differing_from_the_reference": int((lat_x != lat_ref).sum()),
                                 "images_with_pixels_bit_equal_to_the_reference": (int((crc == np.load(gd)["crc"]).sum()) if os.path.exists(gd) else None), "images": B,
                                 "psnr_delta_max_dB": float(np.abs(psnr_each(rec_x) - g["psnr_ref"]).max()),
                                 "note": "the same 16 ids and noise through the exact-order MMDiT + exact VAE decoder: crc32 of every image's bf16 pixels against the reference "
                                         "pipeline run's (tests/golden/decode_b16.npz)"}
    return out


GOLD64 = (os.path.join(ROOT, "tests", "golden", "pipeline_b64.npz"), os.path.join(ROOT, "tests", "golden", "encode_b64.npz"))


def parity_64(pipe, modes=("fp32", "f16x2")):
    """BASELINE configs[1] AT ITS CONFIGURED BATCH, for the arithmetics the reported images/s are measured in: the 64 ids of the reference's own one-batch run
    (tests/golden/encode_b64.npz) + the hash noise through 50 steps in gemm='fp32' and 'f16x2' (reference call: SelftokPipeline.py:227-294), against that
    run's goldens (tests/golden/pipeline_b64.npz: crc32 of every image's final latents and pixels, PSNR of every image).  The golden holds crc32s, not the
    4 MB of latents: the reference's latents are re-derived here by the gemm='exact' mode and PROVEN to be the reference's by those crc32s (64 / 64), then used
    as the comparison tensor.  Per mode: final-latent max |delta| (per image), per-image |delta PSNR| vs the reference's `psnr_ref` end to end (the mode's own
    decoder) and through the same decoder (the reference's latents and ours through the same deterministic decoder, 64 images per call), with the metric's own
    floor beside it (the reference's latents x (1 + 2^-22) through that decoder: the decoder starts by rounding the latents to bf16)."""
    import zlib
    import numpy as np
    import torch
    from selftoktokenizer_amd import evaluate as E, synth
    g, e = np.load(GOLD64[0]), np.load(GOLD64[1])
    ids = e["tokens"].astype(np.int64)
    B = ids.shape[0]
    imgs = synth.synthetic_images(B)
    noise = synth.synthetic_noise(B)
    main = pipe.model.model.gemm

    def stats(d):
        i = int(np.argmax(d))
        return {"mean_dB": round(float(d.mean()), 7), "max_dB": round(float(d.max()), 7), "image_of_max": i, "images_above_1e-3_dB": int((d >= 1e-3).sum())}
    out = {"images": B, "reference": "mimogpt.infer.SelftokPipeline.decoding of 64 id rows in ONE batch (CPU, build container; tests/golden/pipeline_b64.npz)",
           "gates": "north star: |delta PSNR| < 1e-3 dB on every image, mean < 5e-4 (tests/test_parity16_gpu.py::test_psnr_64_images_headline_modes_vs_reference)"}
    try:
        pipe.set_gemm("exact")
        rec_x, lat_ref = pipe.decoding(ids, noise=noise, return_latent=True)
        lx = lat_ref.float().cpu().contiguous().numpy()
        bx = rec_x.cpu().view(torch.int16).numpy().view(np.uint16)
        lat_ok = int(sum(zlib.crc32(lx[i].tobytes()) == int(g["lat_crc"][i]) for i in range(B)))
        pix_ok = int(sum(zlib.crc32(np.ascontiguousarray(bx[i]).tobytes()) == int(g["crc"][i]) for i in range(B)))
        out["exact"] = {"gemm": "exact", "vae_decode": pipe.vae.decode_mode, "images_with_final_latents_bit_equal_to_the_reference": lat_ok,
                        "images_with_pixels_bit_equal_to_the_reference": pix_ok,
                        "psnr_identical_to_the_reference": bool(np.array_equal(E.psnr_each(rec_x, imgs), g["psnr_ref"])),
                        "note": "the exact modes reproduce the reference's run bit for bit; their latents are the comparison tensor of the legs below"}
        if lat_ok != B:
            out["error"] = "the exact mode's latents are not the reference's: no comparison tensor"
            return out
        del rec_x
        for mode in modes:
            if pipe.set_gemm(mode) != mode:
                out[mode] = {"error": "mode refused (a weight outside the fp16 range)"}
                continue
            rec, lat = pipe.decoding(ids, noise=noise, return_latent=True)
            dl = (lat - lat_ref).abs().reshape(B, -1).amax(dim=1).cpu().numpy()
            d_e2e = np.abs(E.psnr_each(rec, imgs) - g["psnr_ref"])
            leg = {"gemm": mode, "vae_decode": pipe.vae.decode_mode,
                   "final_latent_max_abs_delta": round(float(dl.max()), 9), "final_latent_max_abs_delta_mean_over_images": round(float(dl.mean()), 9),
                   "end_to_end": stats(d_e2e)}
            for dec in ("parity", "exact"):
                prev = pipe.vae.decode_mode
                pipe.vae.set_decode_mode(dec)
                try:
                    p_ref, p_our = E.psnr_each(pipe._to_pixels(lat_ref), imgs), E.psnr_each(pipe._to_pixels(lat), imgs)
                    p_flo = E.psnr_each(pipe._to_pixels(lat_ref * (1.0 + 2.0 ** -22)), imgs)
                finally:
                    pipe.vae.set_decode_mode(prev)
                d_same, d_floor = np.abs(p_our - p_ref), np.abs(p_flo - p_ref)
                st = stats(d_same)
                st["floor"] = dict(stats(d_floor), at_the_image_of_max_dB=round(float(d_floor[st["image_of_max"]]), 7))
                st["end_to_end_with_this_decoder"] = stats(np.abs(p_our - g["psnr_ref"]))
                leg[f"same_decoder_{dec}"] = st
            out[mode] = leg
    finally:
        pipe.set_gemm(main)
    return out


def latency_b1(pipe, n=3):
    """BASELINE configs[0]-style single image through the public API: encode + 50-step decode, launched eagerly (~21 k kernel launches) and
    replayed from the hipGraph `decoding(use_graph=True)` captures once per shape"""
    import numpy as np
    import torch
    from selftoktokenizer_amd import synth
    img = synth.synthetic_images(1, device=pipe.device)

    def once(graph):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = pipe.encoding(img)
        t1 = time.perf_counter()
        pipe.decoding(tok.cpu().numpy(), use_graph=graph)
        torch.cuda.synchronize()
        return 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)
    res = {}
    for name, graph in (("eager", False), ("hipgraph", True)):
        once(graph)                                    # warm-up (and the capture)
        runs = [once(graph) for _ in range(n)]
        res[name] = {"encode_ms": round(float(np.median([r[0] for r in runs])), 2), "encode_plus_decode_ms": round(float(np.median([r[1] for r in runs])), 2)}
    main = pipe.model.model.gemm
    alt = "f16x2" if main == "fp32" else "fp32"
    if pipe.set_gemm(alt) == alt:                      # the same image on the other Linear arithmetic
        for name, graph in (("eager", False), ("hipgraph", True)):
            once(graph)
            runs = [once(graph) for _ in range(n)]
            res[f"{name}_{alt}"] = {"encode_ms": round(float(np.median([r[0] for r in runs])), 2), "encode_plus_decode_ms": round(float(np.median([r[1] for r in runs])), 2)}
    pipe.set_gemm(main)
    res["note"] = ("B = 1, 256x256, 512 tokens, 50 steps; eager / hipgraph in the gemm mode of the headline, <..>_<other> in the other; encode_ms is host time to the "
                   "returned (asynchronous) id tensor's launch end.  fp32: 35 TFLOP of block Linears per image on hipBLASLt at 45-65 % of the fp32 matrix peak for M = 256 "
                   "rows (222 ms at 100 %).  f16x2: the block Linears take the small-M split-K entry points (ops.f16x2_ksplit; 767 ms with the single-pass kernels)")
    return res


def kernel_roofs(pipe, B, K, k_table):
    """stand-alone, event-timed launches of the kernels a decode step is made of, at the shapes of the timed step
    (context rows = the mean live length over the 50 steps)."""
    import torch
    import torch.nn.functional as F
    from selftoktokenizer_amd import ops
    dev = pipe.device
    H, NH = 1536, 24
    n = int(round(float(sum(int(k) + 1 for k in k_table)) / len(k_table)))
    n = min((int(k) + 1 for k in k_table), key=lambda v: abs(v - n))          # a context length the step really has: its Linears then run the kernels gemm_tune installed
    out = []
    # attention: 23 of 24 blocks have both streams as queries
    cq = torch.randn(B, n, 3 * H, device=dev)
    xq = torch.randn(B, 256, 3 * H, device=dev)
    oc, ox = torch.empty(B, n, H, device=dev), torch.empty(B, 256, H, device=dev)
    seg0 = (cq[..., :H], cq[..., H:2 * H], cq[..., 2 * H:], oc)
    seg1 = (xq[..., :H], xq[..., H:2 * H], xq[..., 2 * H:], ox)
    ms = event_time_ms(lambda: ops.attention(seg0, seg1, NH, 64))
    fl = 4.0 * B * NH * 64 * (n + 256) * (n + 256)
    out.append({"kernel": "attn64_dma_kernel (LDS-DMA staged K/V; bit-identical to attn64_kernel)", "bound": "mfma(fp32)", "shape": f"B={B} heads=24 S={n}+256", "avg_launch_ms": round(ms, 4),
                "achieved": round(fl / ms / 1e9, 1), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4),
                "launches_per_step": 24 * 50})
    ms = event_time_ms(lambda: ops.attention(seg0, seg1, NH, 64, mode=ops.ATTN_F16X2))
    out.append({"kernel": "attn64_f16x2_kernel", "bound": "valu (softmax + operand split beside 24 f16 MFMAs per 32-key tile)",
                "shape": f"B={B} heads=24 S={n}+256", "avg_launch_ms": round(ms, 4), "achieved": round(3 * fl / ms / 1e9, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s (f16 MFMA: 3 per fp32 product)", "frac": round(3 * fl / ms / 1e9 / F16_MFMA_PEAK_TFLOPS, 4),
                "fp32_equivalent_TFLOPs": round(fl / ms / 1e9, 1), "launches_per_step": 24 * 50})
    # fused residual + LayerNorm + modulate on the context stream
    x, y = torch.randn(B, n, H, device=dev), torch.randn(B, n, H, device=dev)
    tab = torch.randn(n, 6 * H, device=dev)
    ms = event_time_ms(lambda: ops.residual_ln_mod(x, y=y, gate=tab[:, 2 * H:3 * H], shift=tab[:, 3 * H:4 * H], scale=tab[:, 4 * H:5 * H]))
    by = 4.0 * B * n * H * 4 + 3.0 * n * H * 4
    out.append({"kernel": "residual_ln_mod_walk_kernel", "bound": "hbm", "shape": f"[{B},{n},{H}] ctx stream", "avg_launch_ms": round(ms, 4),
                "achieved": round(by / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                "launches_per_step": 4 * 24 * 50})
    # the block Linears: qkv of the context stream as the representative shape
    a = torch.randn(B * n, H, device=dev)
    w = torch.randn(3 * H, H, device=dev) * 0.02
    b = torch.randn(3 * H, device=dev)
    fl = 2.0 * B * n * 3 * H * H
    # what the STEP runs: inside the pipeline's gemm_tune.enabled() block the family's measured kernel serves this row count (it is one of the step's); outside, the library's
    # own choice.  (Round 5's line quoted the default here -- 0.79-0.83 at this shape -- while the timed steps ran the tuned kernel.)
    import contextlib
    from selftoktokenizer_amd import gemm_tune as _gt
    tuned = bool(getattr(pipe, "gemm_tune_report", None)) and getattr(pipe, "tune_gemm", False)
    ms_default = event_time_ms(lambda: F.linear(a, w, b))
    with (_gt.enabled() if tuned else contextlib.nullcontext()):
        ms = event_time_ms(lambda: F.linear(a, w, b))
        fam = []
        for nm, (Nn, Kk) in zip(("qkv", "proj", "fc1", "fc2"), _gt.FAMILIES):
            for rows_ in (B * n, B * 256):
                aa, ww, bb = torch.randn(rows_, Kk, device=dev), torch.randn(Nn, Kk, device=dev) * 0.02, torch.randn(Nn, device=dev)
                t_ = event_time_ms(lambda: F.linear(aa, ww, bb), n=20, warm=3)
                fam.append({"linear": nm, "shape": f"[{rows_},{Kk}]x[{Kk},{Nn}]", "ms": round(t_, 4), "frac": round(2.0 * rows_ * Nn * Kk / t_ / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4)})
                del aa, ww, bb
    out.append({"kernel": "hipBLASLt fp32 GEMM (PyTorch-ROCm; kernel per shape family picked by gemm_tune.py when it is on)", "bound": "mfma(fp32)", "shape": f"[{B * n},{H}]x[{H},{3 * H}]",
                "avg_launch_ms": round(ms, 4), "achieved": round(fl / ms / 1e9, 1), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4),
                "measured": "inside gemm_tune.enabled(): the kernel the timed steps run for this row count" if tuned else "hipBLASLt's own choice (tune_gemm off)",
                "library_default_ms": round(ms_default, 4), "library_default_frac": round(fl / ms_default / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4),
                "all_four_families_at_the_median_context_and_the_image_rows": fam})
    packed = ops.linear_f16x2_pack(w)
    # the practical ceiling of the f16 matrix cores on THIS box: the vendor's plain fp16 GEMM with the same number of MFMAs (K tripled),
    # random data -- the chip is power-limited there (effective shader clock ~1.4 GHz under matrix + LDS load)
    a16 = torch.randn(B * n, 3 * H, device=dev, dtype=torch.float16)
    w16 = torch.randn(3 * H, 3 * H, device=dev, dtype=torch.float16)
    ms_lib16 = event_time_ms(lambda: F.linear(a16, w16))
    del a16, w16
    a_s = ops.split_f16x2(a)
    for name, fn in (("linear_f16x2_pre_kernel (activations pre-split by their producer, LDS-DMA staged, ping-pong; the kernel of the f16x2 step)",
                      lambda: ops.linear_f16x2_split(a_s, packed, b, 3 * H)),
                     ("linear_f16x2_kernel (fp32 activations split in-kernel)", lambda: ops.linear_f16x2(a, packed, b, 3 * H))):
        ms = event_time_ms(fn)
        out.append({"kernel": name, "bound": "mfma(f16), power-limited", "shape": f"[{B * n},{H}]x[{H},{3 * H}]", "avg_launch_ms": round(ms, 4),
                    "achieved": round(3 * fl / ms / 1e9, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (f16 MFMA: 3 per fp32 product)",
                    "frac": round(3 * fl / ms / 1e9 / F16_MFMA_PEAK_TFLOPS, 4), "fp32_equivalent_TFLOPs": round(fl / ms / 1e9, 1),
                    "vendor_fp16_gemm_same_mfma_count_ms": round(ms_lib16, 4), "frac_of_vendor_fp16_gemm_rate": round(ms_lib16 / ms, 4)})
    del a, w, a_s, packed
    # the VAE's convolution (csrc/conv.hip) on its two heaviest decoder shapes, and the vendor bf16 GEMM of the same size beside it
    for (Hh, C, note) in ((256, 128, "decoder up_blocks.3 resnet conv, 6 per decode + 4 per encode"), (64, 512, "decoder up_blocks.1 resnet conv")):
        xx = torch.randn(B, Hh, Hh, C, device=dev).to(torch.bfloat16)
        pc = ops.PackedConv(torch.randn(C, C, 3, 3, device=dev).to(torch.bfloat16) * 0.02, torch.randn(C, device=dev).to(torch.bfloat16))
        ms = event_time_ms(lambda: ops.conv2d_nhwc(xx, pc), n=10, warm=2)
        fl = 2.0 * B * Hh * Hh * C * C * 9
        ag = torch.randn(B * Hh * Hh, 9 * C, device=dev).to(torch.bfloat16) if Hh <= 64 else None       # im2col-sized GEMM operand: 2.4 GB at 64 x 64 x 512
        wg = torch.randn(C, 9 * C, device=dev).to(torch.bfloat16)
        ms_lib = event_time_ms(lambda: F.linear(ag, wg), n=5, warm=2) if ag is not None else None
        out.append({"kernel": "conv3x3_rows_kernel<1> (implicit GEMM, fp32 accumulate incl. bias, one rounding; csrc/conv.hip)", "bound": "mfma(bf16), power-limited",
                    "shape": f"[{B},{Hh},{Hh},{C}] -> {C}, 3x3 ({note})", "avg_launch_ms": round(ms, 4), "achieved": round(fl / ms / 1e9, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / F16_MFMA_PEAK_TFLOPS, 4),
                    "vendor_bf16_gemm_same_flops_ms": None if ms_lib is None else round(ms_lib, 4)})
        del xx, pc, ag, wg
    # the exact-order VAE encoder's convolution (csrc/vae_exact.hip): oneDNN's AMX chunk order on the fp32 matrix cores
    xx = torch.randn(B, 128, 128, 256, device=dev).to(torch.bfloat16)
    ww = (torch.randn(256, 3, 3, 256, device=dev) * 0.02).to(torch.bfloat16)
    bb = torch.randn(256, device=dev).to(torch.bfloat16)
    ms = event_time_ms(lambda: ops.vx_conv2d(xx, ww, bb), n=5, warm=2)
    fl = 2.0 * B * 128 * 128 * 256 * 256 * 9
    out.append({"kernel": "xconv_kernel<2,2,2,false> (exact-order convolution: even / odd fp32 chains per 32-channel chunk on v_mfma_f32_32x32x1_2b_f32, csrc/vae_exact.hip)",
                "bound": "mfma(fp32)", "shape": f"[{B},128,128,256] -> 256, 3x3 (encoder down_blocks.1 resnet conv, 3 per encode)", "avg_launch_ms": round(ms, 4),
                "achieved": round(fl / ms / 1e9, 1), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4)})
    del xx, ww, bb
    return out


def token_match(pipe, images, tokens_gpu, sd_gpu, vsd_gpu, K, n_check=2, first_index=0):
    """token-id exact match of the timed batch (rank 0's shard):
       vs_reference    : against the REFERENCE's own `encoding` of the same 64 images in one batch (tests/golden/encode_b64.npz; present when the timed
                         batch is that batch: K = 512, images 0..63) -- ALL images of the batch, every flip listed with the reference's gap;
       kernel_boundary : the HIP VQ kernel vs the C oracle on the SAME features, every row of the batch (must be 1.0);
       e2e_same_latents: GPU encoder+VQ vs the bit-exact CPU twin (oracle/encoder_exact.c) from the same fp32 latents, first n_check images: the
                         FEATURES must be equal bit for bit (`features_bits_differing` 0);
       e2e_vs_oracle   : from pixels, i.e. incl. the bf16 VAE (oracle/vae_exact.c = the reference's run bit for bit), first n_check images."""
    import numpy as np
    import torch
    from oracle import clib, model as OM, encoder_exact as EX
    from selftoktokenizer_amd.encoder import encoder_pos_embedding
    torch.set_num_threads(host_cores())
    enc = pipe.model.encoder
    with torch.no_grad():
        x0 = pipe.encode_latents(images)
        z = enc.features(x0)
        ids_gpu = tokens_gpu.cpu().numpy()
        cb = enc.codebook.cpu().numpy()
        ids_c, _ = clib.vq_encode_mt(z.reshape(-1, 16).cpu().numpy(), cb)          # every row, threaded over row chunks
        kb = float((ids_c.reshape(ids_gpu.shape) == ids_gpu).mean())
        ref = None
        gold = os.path.join(ROOT, "tests", "golden", "encode_b64.npz")
        if K == 512 and first_index == 0 and ids_gpu.shape[0] <= 64 and os.path.exists(gold):
            g = np.load(gold)
            n = ids_gpu.shape[0]
            rt = g["tokens"][:n].astype(np.int64)
            mism = rt != ids_gpu
            x0_ref = torch.from_numpy(g["x0_bf16"][:n]).view(torch.bfloat16).float()
            ref = {"images_checked": int(n), "tokens": int(rt.size), "equal": int(rt.size - mism.sum()), "match": round(float(1.0 - mism.mean()), 8),
                   "flips": [{"image": int(b), "token": int(k), "reference_id": int(rt[b, k]), "ours": int(ids_gpu[b, k]), "reference_gap": float(g["gap"][b, k]),
                              "reference_runner_up": int(g["id2"][b, k])} for b, k in np.argwhere(mism)[:32]],
                   "latents_bits_differing": int((x0.cpu() != x0_ref).sum()),
                   "features_bits_differing": int((z.cpu().numpy().view(np.uint32) != g["z"][:n].view(np.uint32)).sum()),
                   "reference_tokens_with_gap_below_1e-5": int((g["gap"][:n] < 1e-5).sum()), "smallest_reference_gap": float(g["gap"][:n].min()),
                   "source": "tests/golden/encode_b64.npz = mimogpt.infer.SelftokPipeline.encoding on these 64 images in ONE batch (CPU, build container; "
                             "tools/oracle/gen_golden.py encode64)"}
        sd = {k: v.detach().cpu() for k, v in sd_gpu.items() if k.startswith("encoder.")}
        vsd = {k: v.detach().cpu() for k, v in vsd_gpu.items() if k.startswith("encoder.") or k.startswith("quant_conv")}
        pos = encoder_pos_embedding(K).numpy()
        tables = EX.encoder_tables(sd, K, pos)
        z_o = EX.encoder_features(sd, x0[:n_check].cpu().numpy(), pos, tables=tables)
        feat_diff = int((z_o.view(np.uint32) != z[:n_check].cpu().numpy().view(np.uint32)).sum())
        ids_same = OM.vq_ids(sd, torch.from_numpy(z_o)).numpy()
        # the VAE of the CPU checker: oracle/vae_exact.c, the bit-for-bit restatement of the reference's torch-CPU run (host independent --
        # torch's own bf16 convolution sums in another order on a host without AMX); ~10 s per image on 8 cores
        from oracle import vae_exact as VX
        mom = VX.encode_moments(VX.pack_weights(vsd), VX.bf16_bits(images[:n_check].cpu().to(torch.bfloat16).permute(0, 2, 3, 1)))
        x0_e2e = OM.process_in(VX.bits_to_torch(mom[..., :16]).permute(0, 3, 1, 2).contiguous()).to(torch.float32)
        z_e2e = torch.from_numpy(EX.encoder_features(sd, x0_e2e.numpy(), pos, tables=tables))
        ids_e2e = OM.vq_ids(sd, z_e2e).numpy()

        def gaps(ids_o, z_feat):
            mism = ids_o != ids_gpu[:n_check]
            if not mism.any():
                return []
            xn = torch.nn.functional.normalize(torch.as_tensor(z_feat).reshape(-1, 16), dim=-1)[torch.from_numpy(mism.reshape(-1))]
            top2 = (xn @ torch.from_numpy(cb).T).topk(2, dim=-1).values
            return sorted(round(float(g), 8) for g in (top2[:, 0] - top2[:, 1]))
    return {"vs_reference": ref, "kernel_boundary": kb, "kernel_boundary_rows": int(ids_gpu.size),
            "e2e_same_latents": round(float((ids_same == ids_gpu[:n_check]).mean()), 6), "features_bits_differing": feat_diff, "mismatch_gaps_same_latents": gaps(ids_same, z_o),
            "e2e_vs_oracle": round(float((ids_e2e == ids_gpu[:n_check]).mean()), 6), "mismatch_gaps_e2e": gaps(ids_e2e, z_e2e), "images_checked": n_check,
            "encoder_mode": enc.mode, "vae_mode": pipe.vae.mode,
            "note": "vs_reference covers EVERY image of the timed batch against the reference's own run; the CPU-twin legs (oracle/encoder_exact.c + oracle/vae_exact.c: "
                    "bit-for-bit restatements of the reference's torch-CPU arithmetic, ~13 s per image on 8 cores) re-derive the first images_checked images on this box. "
                    "gap = oracle top-1 minus top-2 cosine score of each mismatching token"}


T_PROCESS_START = time.perf_counter()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    maybe_respawn(args, argv)

    import numpy as np
    import torch
    from selftoktokenizer_amd import dist as D, synth

    # SELFTOK_DIST_BACKEND=gloo + SELFTOK_ONE_GPU=1: dry run of the N>1 flow with every rank on GPU 0 (single-GPU boxes)
    one_gpu = os.environ.get("SELFTOK_ONE_GPU") == "1"
    backend_req = os.environ.get("SELFTOK_DIST_BACKEND", "gloo" if args.selftest_dist else "nccl")
    if one_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = D.init_from_env(backend_req, single_rank_group=args.force_collective)
    if args.force_collective:
        D.force_single_rank(True)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)"
    backend = D.backend_name()
    if world > 1 and not args.selftest_dist and os.environ.get("SELFTOK_DIST_BACKEND") is None:
        assert backend == "nccl", f"N>1 must run over RCCL (torch.distributed backend 'nccl'), got {backend}"

    if args.selftest_dist:
        B, K = args.batch, args.tokens
        ids = torch.from_numpy(synth.synthetic_token_ids(B, K, first_index=rank * B))
        gathered, ag_ms = D.all_gather_ids_timed(ids)
        ok = bool(torch.equal(gathered, torch.from_numpy(synth.synthetic_token_ids(world * B, K))))
        ok_all = D.max_over_ranks(0.0 if ok else 1.0, "cpu") == 0.0
        D.barrier()
        if rank == 0:
            print(json.dumps({"selftest": "dist", "n_gpus": world, "ranks": world, "backend": backend, "allgather_ok": ok_all,
                              "allgather_bytes": int(world * B * K * 4), "allgather_ms": round(ag_ms, 4)}), flush=True)
        D.shutdown()
        return

    from selftoktokenizer_amd import ops, weights as W
    from selftoktokenizer_amd.config import default_config
    from selftoktokenizer_amd.pipeline import SelftokPipeline

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if not one_gpu:
        assert torch.cuda.device_count() > local, f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} visible GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    K, B = args.tokens, args.batch
    renderer = args.decoder == "renderer"
    cfg = default_config(K, renderer=renderer)
    sd = W.synthetic_state_dict(W.expected_shapes(K, renderer=renderer), device=dev)
    vsd = W.synthetic_vae_state_dict(device=dev)
    pipe = SelftokPipeline(cfg, None, None, device=dev, state_dict=sd, vae_state_dict=vsd, verbose=False, gemm=args.gemm, vae_mode=args.vae,
                           tune_gemm=bool(args.tune_gemm), encoder_mode=args.encoder, vae_decode_mode=args.vae_decode)
    if world > 1 and not args.all_legs:                # the N > 1 line = the timed steps + the cheap checks; the deep dives belong to the N = 1 line
        args.no_other_gemm = args.no_kernel_roofs = args.no_parity16 = args.no_parity64 = True
        args.latency = False
    gemm_main = pipe.model.model.gemm
    if args.tune_gemm and gemm_main == "fp32":
        pipe.tune_linears(B, renderer=renderer)       # explicit, outside the timed region (~4 s): nothing is measured inside a decoding() call

    images = synth.synthetic_images(B, device=dev, first_index=rank * B)          # resident in HBM
    torch.cuda.synchronize()

    ag = {"ms": [], "bytes": 0}
    last = {}
    gather = D.id_gatherer(B, K, dev)          # shard sizes exchanged once, here; a step then issues exactly one collective
    ag["bytes"] = gather.payload_bytes

    def step():
        tokens = pipe.encoding(images)                            # [B,K] int64 on device (public API)
        gather.launch(tokens, timed=True)                         # ONE RCCL all-gather (int32 payload) on a side stream; no-op at N=1
        mine = tokens.cpu().numpy()                               # a rank decodes its own shard: it needs only its own ids, as the host array the API takes
        last["tokens"] = tokens
        if renderer:
            out = pipe.decoding_with_renderer(mine)
        else:
            out = pipe.decoding(mine, max_steps=args.decode_steps)   # noise: torch.randn on the CPU generator, as the reference
        last["ids_all"] = gather.wait()                           # the API result [world*B, K]: joined on the device, behind the decode
        ag["ms"].append(gather.last_ms())
        return out

    def timed(nsteps, nwarm):
        torch.manual_seed(1234 + rank)
        for _ in range(nwarm):
            step()
        ops.VQ_EVENTS = []
        ag["ms"].clear()
        torch.cuda.synchronize()
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        D.barrier()
        el = D.max_over_ranks(time.perf_counter() - t0, dev)
        ev, ops.VQ_EVENTS = ops.VQ_EVENTS, None
        return el, ev

    elapsed, vq_events = timed(args.steps, args.warmup)

    # ---- the same step on the other GEMM arithmetic (all ranks take part: barriers inside) ----
    other = None
    if not args.no_other_gemm:
        alt = "f16x2" if gemm_main == "fp32" else "fp32"
        if pipe.set_gemm(alt) == alt:
            n_alt = min(args.steps, 3)
            el_alt, _ = timed(n_alt, 1)
            other = {"gemm": alt, "value": round(world * B * n_alt / el_alt, 4), "unit": "images/s", "steps": n_alt, "warmup": 1,
                     "ms_per_step": round(1000.0 * el_alt / n_alt, 2)}
        pipe.set_gemm(gemm_main)
    def time_exact(note):
        if pipe.set_gemm("exact") != "exact":
            return None
        try:
            el_ex, _ = timed(1, 0)
        finally:
            pipe.set_gemm(gemm_main)
        return {"gemm": "exact", "value": round(world * B / el_ex, 4), "unit": "images/s", "steps": 1, "warmup": 0, "ms_per_step": round(1000.0 * el_ex, 2), "warm": note}
    exact = None
    want_exact = not args.no_exact and not renderer and gemm_main != "exact" and not (world > 1 and not args.all_legs)
    # N = 1: the exact step is timed AFTER the parity legs below, whose exact-mode decodes have already loaded its kernels and built its tables (one step, no
    # warm-up step of its own: 25 s saved); N > 1 (--all-legs): here, with every rank taking part in the barriers
    if want_exact and world > 1:
        exact = time_exact("first exact-mode call of the process")

    if rank != 0:
        D.shutdown()
        return
    n_vq = B * K
    vq_main = float(np.mean([a.elapsed_time(b) for a, b, _ in vq_events])) if vq_events else float("nan")
    vq_fin = float(np.mean([b.elapsed_time(c) for _, b, c in vq_events])) if vq_events else float("nan")
    # the fp32-input MFMA kernel of round 1 on the same features, for reference (same ids, bit for bit)
    zf = pipe.model.encoder.features(pipe.encode_latents(images))
    _, lm, lf = ops.vq_encode_split_launch(zf, pipe.model.encoder.codebook_packed, coarse=False)
    fp32_main, fp32_fin = event_time_ms(lm, n=20), event_time_ms(lf, n=20)
    traffic, traffic_note = measured_vq_traffic(n_vq, ops.VQ_DEFAULT_COARSE)
    roof = vq_roofline(n_vq, 32768, 16, vq_main, vq_fin, len(vq_events), traffic, traffic_note, fp32_main, fp32_fin, mfmas=ops.VQ_COARSE_MFMAS)
    arith = {"fp32": "fp32 Q-Former/VQ/MMDiT (hipBLASLt fp32 GEMMs), bf16 SD3-VAE (reference dtypes)",
             "exact": "every operation of Q-Former, VQ, MMDiT and VAE as the sequence of fp32 / bf16 operations the reference's torch-CPU run executes (MKL / oneDNN / ATen / Sleef orders "
                      "on chained fp32 MFMAs): ids, latents and pixels bit-equal to the reference's",
             "f16x2": "fp32 Q-Former/VQ/MMDiT with the MMDiT block Linears and joint attention as f16x2-split products on the f16 matrix cores "
                      "(fp32-equivalent: error vs fp64 below the fp32 kernels', tests/test_gemm_gpu.py, test_kernels_gpu.py), bf16 SD3-VAE"}
    from selftoktokenizer_amd import gemm_tune as _gt
    gemm_tune_cuts = dict(_gt.LAST_CUTS)
    line = {
        "metric": "images/sec encode+decode, 256x256 %d-token" % K, "value": round(world * B * args.steps / elapsed, 4),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: batch %d x 256x256 per GPU, %d-token encode + %s decode"
                               % (3 if renderer else (2 if K == 1024 else 1), B, K, "one-step renderer" if renderer else "50-step diffusion"),
                   "global_batch": world * B, "tokens": K, "decode_steps": 1 if renderer else (args.decode_steps or 50),
                   "gemm": gemm_main, "arithmetic": arith[gemm_main], "vae": pipe.vae.mode, "vae_encode": pipe.vae.mode, "vae_decode": pipe.vae.decode_mode,
                   "encoder": pipe.model.encoder.mode,
                   "tune_gemm": bool(args.tune_gemm), "tune_gemm_note": "opt-in extension of the pipeline (default off there): hipBLASLt's kernel per Linear shape family chosen by a "
                                                                        "~4 s measurement before the warm-up; --tune-gemm 0 measures hipBLASLt's own choice",
                   "fp32_linear_kernels": (None if not pipe.gemm_tune_report else {f"{n}x{k}": {"kernel": b or "hipBLASLt default", "ms_default": t0, "ms_chosen": t1, "rows_up_to_this_keep_the_default": gemm_tune_cuts.get((n, k), 0)}
                                                                                       for (n, k), (b, t0, t1) in pipe.gemm_tune_report.items()}),
                   "parallelism": "batch-shard x%d" % world,
                   "api": "pipe.encoding(images) -> id all-gather -> pipe.decoding(ids.cpu().numpy()) (noise from the CPU generator, as the reference)",
                   "weights": "hash-generated, architecture of tokenizer_512_ckpt"},
        "ranks": world, "backend": backend if (world > 1 or args.force_collective) else None,
        "allgather_bytes": ag["bytes"], "allgather_ms": round(float(np.mean(ag["ms"])), 4) if (world > 1 or args.force_collective) and ag["ms"] else None,
        "roofline": roof,
    }
    fl_img = fp32_flops_per_image(K, pipe.k_table[: (args.decode_steps or 50)], renderer)
    job_tf = fl_img * world * B * args.steps / elapsed / 1e12
    line["job_roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "fp32_tflop_per_image": round(fl_img / 1e12, 2),
                            "achieved": round(job_tf, 1), "peak": FP32_MFMA_PEAK_TFLOPS * world,
                            "frac": round(job_tf / (FP32_MFMA_PEAK_TFLOPS * world), 4),
                            "note": "fp32-equivalent matrix FLOPs actually executed (encoder + MMDiT, context truncated to live tokens) / wall time, "
                                    "against the fp32 matrix peak; bf16 VAE work (0.89 TFLOP/img) excluded.  In f16x2 GEMM mode the Linears run on the "
                                    "16x faster f16 matrix cores (3 MFMAs per fp32 product), so this fraction may exceed 1"}
    if other is not None:
        line["gemm_modes"] = {gemm_main: {"value": line["value"], "ms_per_step": line["ms_per_step"]}, other["gemm"]: other,
                              "note": "same step, MMDiT block Linears on the other arithmetic; 'value' of this line is the '%s' run" % gemm_main}
    if want_exact and world == 1:
        pending_exact = True
    else:
        pending_exact = False
    if exact is not None:
        line.setdefault("gemm_modes", {gemm_main: {"value": line["value"], "ms_per_step": line["ms_per_step"]}})["exact"] = dict(
            exact, note="the parity mode: every Linear / LayerNorm / GELU / SiLU / attention of the MMDiT in the summation order torch-CPU executes for the reference "
                        "(csrc/gemm_fp32.hip + csrc/encoder_exact.hip; MKL's K-blocking on chained fp32 MFMAs, the joint attention fused on the full masked key sequence) "
                        "-- final latents and pixels bit-equal to the reference pipeline's (parity_16.exact_mode, parity_64.exact)")
    if args.decode_steps is not None and not renderer:
        line["config"]["INVALID"] = "decode loop truncated with --decode-steps (debug run)"
    legs = {"timed_region": round(elapsed * (args.steps + args.warmup) / max(args.steps, 1), 1),
            "other_gemm_steps": round(other["ms_per_step"] * (other["steps"] + other["warmup"]) / 1e3, 1) if other else 0.0}        # seconds each part of this run took (the run's own wall clock, rank 0)

    def leg(name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); legs[name] = round(time.perf_counter() - t0, 1)
        return out
    if not args.no_kernel_roofs and not renderer:
        line["roofline_kernels"] = leg("roofline_kernels", lambda: kernel_roofs(pipe, B, K, pipe.k_table))
    if not args.no_token_check:
        line["token_match"] = leg("token_match", lambda: token_match(pipe, images, last["tokens"], sd, vsd, K, first_index=rank * B))
        if K == 512 and os.path.exists(GOLD16) and not args.no_parity16:
            line["parity_16"] = leg("parity_16", lambda: parity_16(pipe, exact_leg=not args.no_exact))
        if K == 512 and not renderer and all(os.path.exists(f) for f in GOLD64) and not args.no_parity64 and pipe.vae.mode in ("exact", "parity"):
            line["parity_64"] = leg("parity_64", lambda: parity_64(pipe))
    if pending_exact:
        exact = leg("exact_step", lambda: time_exact("after the exact-mode parity legs of this run (kernels loaded, tables built)" if ("parity_64" in line or "parity_16" in line) else
                           "first exact-mode call of the process"))
        if exact is not None:
            line.setdefault("gemm_modes", {gemm_main: {"value": line["value"], "ms_per_step": line["ms_per_step"]}})["exact"] = dict(
                exact, note="the parity mode: every Linear / LayerNorm / GELU / SiLU / attention of the MMDiT in the summation order torch-CPU executes for the reference "
                            "(csrc/gemm_fp32.hip + csrc/encoder_exact.hip; MKL's K-blocking on chained fp32 MFMAs, the joint attention fused on the full masked key sequence) "
                            "-- final latents and pixels bit-equal to the reference pipeline's (parity_16.exact_mode, parity_64.exact)")
    if args.latency and not renderer:
        line["latency_b1"] = leg("latency_b1", lambda: latency_b1(pipe))
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = leg("cpu_baseline", lambda: cpu_baseline(sd, vsd, cfg, K))
        line["cpu_baseline_reference_survey"] = REFERENCE_SURVEY_BASELINE
    legs["process_so_far"] = round(time.perf_counter() - T_PROCESS_START, 1)
    line["leg_seconds"] = legs
    print(json.dumps(line), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
