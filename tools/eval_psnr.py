"""Reconstruction PSNR of a Selftok tokenizer checkpoint over an image folder, on MI355X (one command = the PSNR column of the reference's
README.md:89-94; see selftoktokenizer_amd/evaluate.py).  Sharded over ranks when launched under torchrun:

    python tools/eval_psnr.py --images <dir> --yml-path configs/res256/256-eval.yml --pretrained tokenizer_512_ckpt.pth --sd3_pretrained <sd3 dir>
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/eval_psnr.py --images <dir> ... --renderer-yml configs/renderer/renderer-eval.yml --renderer-pretrained renderer_512_ckpt.pth
    python tools/eval_psnr.py --synthetic 16          # no checkpoint reachable: hash-generated weights / images / noise = the reference pipeline's golden run

Prints ONE JSON line on rank 0: per-image and mean PSNR for the 50-step `decoding` and, when a renderer checkpoint is given, `decoding_with_renderer`."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimogpt.infer.infer_utils import parse_args_from_yaml
from mimogpt.infer.SelftokPipeline import SelftokPipeline
from selftoktokenizer_amd import dist as D, evaluate as E, synth, weights as W
from selftoktokenizer_amd.config import default_config

ap = argparse.ArgumentParser()
ap.add_argument("--images", default=None, help="folder of images (searched recursively, sorted)")
ap.add_argument("--synthetic", type=int, default=0, help="N hash-generated images + synthetic weights + hash noise instead of files")
ap.add_argument("--limit", type=int, default=0, help="evaluate only the first N images of the folder")
ap.add_argument("--yml-path", default=None)
ap.add_argument("--pretrained", default=None)
ap.add_argument("--sd3_pretrained", default=None)
ap.add_argument("--renderer-yml", default=None)
ap.add_argument("--renderer-pretrained", default=None)
ap.add_argument("--data_size", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--gemm", default=None, choices=["fp32", "f16x2", "exact"], help="exact: every operation in the reference's torch-CPU order (bit-equal pixels, the parity mode)")
ap.add_argument("--out", default=None, help="also write the JSON line to this file (rank 0)")
a = ap.parse_args()

rank, world, local = D.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
cfg = parse_args_from_yaml(a.yml_path) if a.yml_path else default_config(512)
K = int(cfg.tokenizer.params.k)
kw = {}
if a.pretrained is None:
    kw = dict(state_dict=W.synthetic_state_dict(W.expected_shapes(K), device=dev), vae_state_dict=W.synthetic_vae_state_dict(device=dev))
pipe = SelftokPipeline(cfg=cfg, ckpt_path=a.pretrained, sd3_path=a.sd3_pretrained, datasize=a.data_size, device=dev, gemm=a.gemm, verbose=False, **kw)
decoders, rpipe = ["diffusion"], None
if a.renderer_pretrained or (a.renderer_yml and a.pretrained is None):
    rcfg = parse_args_from_yaml(a.renderer_yml) if a.renderer_yml else default_config(K, renderer=True)
    rkw = {} if a.renderer_pretrained else dict(state_dict=W.synthetic_state_dict(W.expected_shapes(K, renderer=True), device=dev), vae_state_dict=W.synthetic_vae_state_dict(device=dev))
    rpipe = SelftokPipeline(cfg=rcfg, ckpt_path=a.renderer_pretrained, sd3_path=a.sd3_pretrained, datasize=a.data_size, device=dev, gemm=a.gemm, verbose=False, **rkw)
    decoders.append("renderer")
if a.synthetic:
    n = a.synthetic
    load = lambda lo, hi: synth.synthetic_images(hi - lo, first_index=lo)
    noise = lambda lo, hi: synth.synthetic_noise(hi - lo, first_index=lo)
    src = f"{n} hash-generated images (selftoktokenizer_amd.synth)"
else:
    assert a.images, "--images <dir> or --synthetic N"
    paths = E.list_images(a.images)
    if a.limit:
        paths = paths[:a.limit]
    assert paths, f"no image files under {a.images}"
    n, load, noise, src = len(paths), E.folder_loader(paths, a.data_size), None, f"{len(paths)} files under {a.images}"
res = E.evaluate(pipe, load, n, batch=a.batch, decoders=decoders, noise_fn=noise, seed=a.seed, renderer_pipe=rpipe, verbose=True)
D.barrier()
if rank == 0:
    line = {"tool": "eval_psnr", "source": src, "data_size": a.data_size, "tokens": K, "gemm": pipe.model.model.gemm, "vae": pipe.vae.mode, "encoder": pipe.model.encoder.mode,
            "checkpoint": a.pretrained or "synthetic (hash-generated)", "renderer_checkpoint": a.renderer_pretrained, **res,
            "readme_reference_dB": {"tokenizer_512_ckpt": 21.86, "renderer_512_ckpt": 24.14, "tokenizer_1024_ckpt": 23.06, "renderer_1024_ckpt": 26.30,
                                    "note": "README.md:89-94 (256 x 256), needs the published weights"}}
    print(json.dumps(line), flush=True)
    if a.out:
        open(a.out, "w").write(json.dumps(line) + "\n")
D.shutdown()
