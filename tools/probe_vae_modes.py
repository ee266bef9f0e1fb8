"""GPU: how the bf16 SD3-VAE behaves under different MIOpen solver-selection modes.

(The first version of this probe compared MIOpen's find modes / immediate mode / the deterministic flag on the rounds 1-2 VAE:
profiles/r3_vae_modes_first.txt.  Find mode and immediate mode change first-call time only; the deterministic flag makes the
latents bit-stable; the big deviation from the reference turned out to be the separate bf16 bias add, see vae.py.)
For every mode the VAE + Q-Former encoder + VQ (no MMDiT: it is not needed) run in TWO fresh processes on the 16 images of
tests/golden/pipeline_b16.npz.  Reported per (mode, run): first-call and steady times, whether latents / ids / pixels are
bit-stable inside the process and across processes (hashes), the token match against the reference's ids with the reference
top-1/top-2 gap of every flip, the latent deviation from the reference's CPU bf16 VAE, and the reconstruction-PSNR delta of
the VAE decoder on the reference's own final latents.

    python tools/probe_vae_modes.py                 # parent: all modes -> gpurun_out/vae_modes.json + table
    python tools/probe_vae_modes.py --child MODE    # one measurement, one JSON line
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# mode -> (environment, torch switches applied in the child before the first convolution)
MODES = {
    "product: bias fold + deterministic + FAST find": ({}, {}),
    "bias fold + FAST find, solver free (non-deterministic flag off)": ({}, {"nondet": True}),
    "bias fold + MIOpen default find (DYNAMIC_HYBRID), solver free": ({"MIOPEN_FIND_MODE": "DYNAMIC_HYBRID"}, {"nondet": True}),
    "separate bias add (rounds 1-2) + deterministic + FAST find": ({}, {"sepbias": True}),
    # which solver family is the inaccurate one?  (MIOPEN_DEBUG_CONV_* switch whole families off; the Find then picks among the rest)
    "bias fold, solver free, Winograd off": ({"MIOPEN_DEBUG_CONV_WINOGRAD": "0"}, {"nondet": True}),
    "bias fold, solver free, Winograd + direct off": ({"MIOPEN_DEBUG_CONV_WINOGRAD": "0", "MIOPEN_DEBUG_CONV_DIRECT": "0"}, {"nondet": True}),
    "bias fold, solver free, only implicit GEMM": ({"MIOPEN_DEBUG_CONV_WINOGRAD": "0", "MIOPEN_DEBUG_CONV_DIRECT": "0", "MIOPEN_DEBUG_CONV_GEMM": "0", "MIOPEN_DEBUG_CONV_FFT": "0"}, {"nondet": True}),
    "bias fold, default find (hybrid), Winograd off": ({"MIOPEN_FIND_MODE": "DYNAMIC_HYBRID", "MIOPEN_DEBUG_CONV_WINOGRAD": "0"}, {"nondet": True}),
}


def h(t):
    return hashlib.sha256(t.detach().cpu().contiguous().view(-1).view(dtype=__import__("torch").uint8).numpy().tobytes()).hexdigest()[:12]


def child(mode):
    import numpy as np
    import torch
    sw = MODES[mode][1]
    from selftoktokenizer_amd import ops, synth, weights as W
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.vae import AutoencoderKLGPU
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_b16.npz"))
    B = g["tokens"].shape[0]
    enc_sd = W.synthetic_state_dict({k: s for k, s in W.expected_shapes(512).items() if k.startswith("encoder.")}, device=dev)
    vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(device=dev), dev)
    if sw.get("nondet"):
        vae.deterministic = False
    if sw.get("sepbias"):       # the rounds 1-2 arithmetic: MIOpen conv rounded to bf16, bias added as a second bf16 op
        import torch.nn.functional as F
        vae._conv = lambda name, x, stride=1, padding=1: F.conv2d(x, vae.w[name + ".weight"], vae.w[name + ".bias"], stride=stride, padding=padding)
    enc = QformerEncoderGPU(enc_sd, dev, 512)
    imgs = synth.synthetic_images(B, device=dev)
    orig = ((synth.synthetic_images(B) + 1.0) / 2.0)

    def encode(x):
        m = vae.encode_moments(x.to(torch.bfloat16))
        return ops.latent_process_in(m.contiguous(), m.shape[1] // 2, 0.0609, 1.5305)

    def decode(lat):
        return ops.clamp01_(vae.decode(ops.latent_process_out(lat, 0.0609, 1.5305))[0].contiguous())

    def timed(fn, *a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a)
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0
    out = {"mode": mode}
    x0, out["first_encode_s"] = timed(encode, imgs)
    lat = torch.from_numpy(g["lat"]).to(dev)
    rec, out["first_decode_s"] = timed(decode, lat)
    x0s, recs = [x0], [rec]
    te, td = [], []
    for _ in range(3):
        a, t = timed(encode, imgs); x0s.append(a); te.append(t)
        b, t = timed(decode, lat); recs.append(b); td.append(t)
    out["steady_encode_ms_B16"], out["steady_decode_ms_B16"] = round(1e3 * min(te), 2), round(1e3 * min(td), 2)
    out["x0_stable_in_process"] = all(torch.equal(x0s[0], a) for a in x0s[1:])
    out["rec_stable_in_process"] = all(torch.equal(recs[0], a) for a in recs[1:])
    out["x0_hash"], out["rec_hash"] = h(x0), h(rec)
    ids = enc(x0, d=None)[1]
    out["ids_hash"] = h(ids)
    ref = g["tokens"].astype(np.int64)
    mism = ids.cpu().numpy() != ref
    out["ids_match_ref"], out["flips"] = round(float(1.0 - mism.mean()), 6), int(mism.sum())
    out["flip_gaps"] = [round(float(v), 8) for v in g["gap"][mism]]
    out["flips_to_runner_up"] = int((ids.cpu().numpy()[mism] == g["id2"].astype(np.int64)[mism]).sum())
    x0_ref = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    d = (x0.cpu() - x0_ref)
    out["x0_maxdiff_ref"], out["x0_rms_ref"] = round(float(d.abs().max()), 5), round(float(d.pow(2).mean().sqrt()), 6)
    mse = ((rec.float().cpu() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
    p = (10.0 * torch.log10(1.0 / mse)).numpy()
    dp = np.abs(p - g["psnr_ref"])
    out["psnr_delta_mean"], out["psnr_delta_max"] = round(float(dp.mean()), 6), round(float(dp.max()), 6)
    if os.environ.get("PROBE_SKIP_B64") == "1":
        print("@@" + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
        return
    # the bench batch size too (MIOpen picks per shape): first call + steady
    imgs64 = synth.synthetic_images(64, device=dev)
    _, out["first_encode_s_B64"] = timed(encode, imgs64)
    _, t = timed(encode, imgs64)
    out["steady_encode_ms_B64"] = round(1e3 * t, 2)
    lat64 = lat.repeat(4, 1, 1, 1)
    _, out["first_decode_s_B64"] = timed(decode, lat64)
    _, t = timed(decode, lat64)
    out["steady_decode_ms_B64"] = round(1e3 * t, 2)
    for k in list(out):
        if isinstance(out[k], float):
            out[k] = round(out[k], 4)
    print("@@" + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    modes = sys.argv[1:] or list(MODES)
    rows = []
    for m in modes:
        for run in range(2):
            env = dict(os.environ)
            env.update(MODES[m][0])
            tag = str(abs(hash(m)) % 100000)
            env["MIOPEN_USER_DB_PATH"] = f"/tmp/miopen_userdb_{tag}"      # every mode starts from an empty user find-db; run 1 re-uses run 0's
            env["MIOPEN_CUSTOM_CACHE_DIR"] = f"/tmp/miopen_cache_{tag}"
            if run == 1:
                env["PROBE_SKIP_B64"] = "1"
            t0 = time.time()
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", m], env=env, capture_output=True, text=True, timeout=420)
                line = [l for l in r.stdout.splitlines() if l.startswith("@@")]
                row = json.loads(line[-1][2:]) if line else {"mode": m, "error": (r.stderr or r.stdout)[-400:]}
            except subprocess.TimeoutExpired:
                row = {"mode": m, "error": "timeout"}
            row["run"], row["process_wall_s"] = run, round(time.time() - t0, 1)
            rows.append(row)
            print(json.dumps(row), flush=True)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rows, open(os.path.join(out, "vae_modes.json"), "w"), indent=1)
    print("\nmode | run | first enc/dec s | steady enc/dec ms (B16) | B64 first enc/dec s, steady ms | in-process stable x0/rec | x0 hash | ids hash | rec hash | ids match (flips) | psnr delta mean/max")
    for r in rows:
        if "error" in r or "skipped" in r:
            print(r)
            continue
        print(f"{r['mode']} | {r['run']} | {r['first_encode_s']}/{r['first_decode_s']} | {r['steady_encode_ms_B16']}/{r['steady_decode_ms_B16']} | "
              f"{r.get('first_encode_s_B64')}/{r.get('first_decode_s_B64')}, {r.get('steady_encode_ms_B64')}/{r.get('steady_decode_ms_B64')} | "
              f"{r['x0_stable_in_process']}/{r['rec_stable_in_process']} | {r['x0_hash']} | {r['ids_hash']} | {r['rec_hash']} | "
              f"{r['ids_match_ref']} ({r['flips']}) | {r['psnr_delta_mean']}/{r['psnr_delta_max']}")


if __name__ == "__main__":
    main()
