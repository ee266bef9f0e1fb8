"""-m gpu: the way a real user enters (reference test.py:24-43, SelftokPipeline.py:154-207): a YAML file, a flat
tokenizer `.pth`, a diffusers-layout `<sd3>/vae/diffusion_pytorch_model.safetensors` -- all on disk -- then
`SelftokPipeline(parse_args_from_yaml(yml), ckpt, sd3_dir)`, `encoding`, `np.save` / `np.load`, `decoding`.
Everything must be BIT-equal to the pipeline the other GPU tests build from in-memory state dicts; plus `ema_decoder=True`
from a file that holds no `model.*` keys at all (the MMDiT then can only have come from `ema_state_dict`).

The published weights are not reachable offline, so the files hold the synthetic parameters (full size: an 8.7 GB `.pth`)."""
import gc
import os
import shutil

import numpy as np
import pytest
import torch
import yaml

from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.config import default_config

pytestmark = pytest.mark.gpu


def _reference_style_yaml(path, K=512):
    """a config file with the key layout of the shipped eval configs (sections `common` / `model` / `optimize` / `tokenizer`,
    training-only keys the path must ignore, and PyYAML turning a bare `None` into the STRING 'None')"""
    hot = default_config(K)
    p = dict(hot["tokenizer"]["params"])
    p.update({"gradient_checkpointing": False, "ema_enc": False, "enc_decay": 0.99, "L2_lr": 0.0, "two_part_losses": False})
    p["quantizer_config"] = dict(p["quantizer_config"], w_diversity=1.0, ema_entropy_ratio=0.8, w_commit=1.0, decay=0.99,
                                 dead_code_threshold=0.2, reset_cluster_size=0.2, smart_react=True, continuous=False, reg=[0.1, 0.3])
    p["decoder_config"] = dict(p["decoder_config"], sd3_cond_pooling="None", class_dropout_prob=0.1, train_filter="all",
                               freeze_filter="", init_method="None")
    p = {k: (dict(v) if isinstance(v, dict) else v) for k, v in p.items()}
    doc = {
        "common": {"output_path": "output", "use_bf16": 1, "use_fp16": 0, "random_seed": 123, "task": "selftokenc", "is_eval": True,
                   "vae_path": "/nowhere/sd3_medium.pt", "pre_encode": False},
        "model": {"pretrain_model": ""},
        "optimize": {"max_epochs": 1000, "grad_norm": 0.0, "lr_scheduler": {"dit_lr": 1.0e-5, "token_lr": 5.0e-5}},
        "tokenizer": {"is_text_tokenized": False, "pretrained_dit_path": "/nowhere/sd3_medium.pt", "params": p},
    }
    text = yaml.safe_dump(doc, sort_keys=False).replace("'None'", "None")      # the shipped files write a bare None
    with open(path, "w") as fd:
        fd.write(text)
    assert "sd3_cond_pooling: None" in text and yaml.safe_load(text)["tokenizer"]["params"]["decoder_config"]["init_method"] == "None"


def _free():
    gc.collect()
    torch.cuda.empty_cache()


def _roundtrip(pipe, images, noise, tmp_path, tag):
    """reference test.py:36-41: ids leave as a .npy file and come back from it"""
    tokens = pipe.encoding(images, device="cuda")
    f = str(tmp_path / f"token_{tag}.npy")
    np.save(f, tokens.detach().cpu().numpy())
    back = np.load(f)
    assert back.dtype == np.int64 and back.shape == (images.shape[0], 512)
    rec, lat = pipe.decoding(back, device="cuda", noise=noise, return_latent=True)
    return back, rec.float().cpu(), lat.cpu()


def test_pipeline_from_yaml_pth_safetensors_equals_in_memory_construction(tmp_path):
    from safetensors.torch import save_file
    from mimogpt.infer.infer_utils import parse_args_from_yaml
    from mimogpt.infer.SelftokPipeline import SelftokPipeline

    free = shutil.disk_usage(str(tmp_path)).free
    assert free > 12e9, f"needs 12 GB of scratch disk for a full-size checkpoint, {free / 1e9:.1f} GB free under {tmp_path}"
    yml = str(tmp_path / "256-eval.yml")
    _reference_style_yaml(yml)
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    vsd32 = {k: v.float() for k, v in W.synthetic_vae_state_dict(device="cpu").items()}    # the published VAE file is fp32; `.to(bf16)` happens after loading (:163)
    ckpt = str(tmp_path / "tokenizer_512_ckpt.pth")
    torch.save({k: v.cpu() for k, v in sd.items()}, ckpt)
    sd3 = tmp_path / "sd3-diffusers"
    (sd3 / "vae").mkdir(parents=True)
    save_file(vsd32, str(sd3 / "vae" / "diffusion_pytorch_model.safetensors"))
    (sd3 / "vae" / "config.json").write_text('{"_class_name": "AutoencoderKL", "latent_channels": 16}')

    images = synth.synthetic_images(2).cuda()
    noise = synth.synthetic_noise(2)

    cfg = parse_args_from_yaml(yml)
    assert cfg.tokenizer.params.decoder_config.sd3_cond_pooling == "None" and cfg.common.is_eval is True
    from_files = SelftokPipeline(cfg=cfg, ckpt_path=ckpt, sd3_path=str(sd3), datasize=256, device="cuda")    # test.py:25
    assert cfg.tokenizer.params.noise_schedule_config.is_eval is True                         # the ctor's side effect on cfg (:167)
    ids_f, rec_f, lat_f = _roundtrip(from_files, images, noise, tmp_path, "files")
    n_params = sum(p.numel() for p in from_files.model.parameters())
    del from_files
    _free()

    in_memory = SelftokPipeline(default_config(512), ckpt_path=None, sd3_path=None, device="cuda", state_dict=sd,
                                vae_state_dict={k: v.cuda() for k, v in vsd32.items()})
    assert sum(p.numel() for p in in_memory.model.parameters()) == n_params
    ids_m, rec_m, lat_m = _roundtrip(in_memory, images, noise, tmp_path, "memory")
    assert np.array_equal(ids_f, ids_m), "token ids differ between the from-files and the in-memory construction"
    assert torch.equal(lat_f, lat_m) and torch.equal(rec_f, rec_m), "decoded pixels differ between the two constructions"
    assert float(rec_f.min()) >= 0.0 and float(rec_f.max()) <= 1.0 and rec_f.shape == (2, 3, 256, 256)
    del in_memory
    _free()

    # ema_decoder=True (reference :172-174, 193-198): encoder from the flat keys, MMDiT from state_dict['ema_state_dict'] (bare keys).
    # The file holds NO model.* key and the EMA copy is a different set of numbers than sd's MMDiT.
    os.remove(ckpt)
    ema = {k[len("model."):]: (v * 0.5 if v.dim() >= 2 and "embed" not in k else v) for k, v in sd.items() if k.startswith("model.")}
    enc_only = {k: v for k, v in sd.items() if not k.startswith("model.")}
    del sd
    ckpt_ema = str(tmp_path / "tokenizer_512_ema.pth")
    torch.save(dict({k: v.cpu() for k, v in enc_only.items()}, ema_state_dict={k: v.cpu() for k, v in ema.items()}), ckpt_ema)
    p_files = SelftokPipeline(cfg=parse_args_from_yaml(yml), ckpt_path=ckpt_ema, sd3_path=str(sd3), ema_decoder=True, device="cuda")
    ids_e, rec_e, lat_e = _roundtrip(p_files, images, noise, tmp_path, "ema_files")
    del p_files
    _free()
    os.remove(ckpt_ema)
    p_mem = SelftokPipeline(default_config(512), None, None, ema_decoder=True, device="cuda", state_dict=dict(enc_only, ema_state_dict=ema),
                            vae_state_dict={k: v.cuda() for k, v in vsd32.items()})
    ids_e2, rec_e2, lat_e2 = _roundtrip(p_mem, images, noise, tmp_path, "ema_memory")
    assert np.array_equal(ids_e, ids_e2) and torch.equal(lat_e, lat_e2) and torch.equal(rec_e, rec_e2)
    assert np.array_equal(ids_e, ids_f), "the encoder does not depend on the EMA switch"
    assert not torch.equal(lat_e, lat_f), "the EMA MMDiT (other weights) produced the non-EMA latents: ema_state_dict was not used"
    del p_mem
    _free()
    # a checkpoint without ema_state_dict + ema_decoder=True: the reference raises KeyError('ema_state_dict') (:194)
    small = str(tmp_path / "enc_only.pth")
    torch.save({k: v.cpu() for k, v in enc_only.items()}, small)
    with pytest.raises(KeyError):
        SelftokPipeline(cfg=parse_args_from_yaml(yml), ckpt_path=small, sd3_path=str(sd3), ema_decoder=True, device="cuda")
    with pytest.raises(FileNotFoundError):
        SelftokPipeline(cfg=parse_args_from_yaml(yml), ckpt_path=small, sd3_path=str(tmp_path / "missing-sd3"), device="cuda")
