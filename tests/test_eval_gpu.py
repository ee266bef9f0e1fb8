"""-m gpu: the PSNR evaluation harness (selftoktokenizer_amd/evaluate.py, tools/eval_psnr.py) against the REFERENCE pipeline's own run:
the 16 images of tests/golden/pipeline_b16.npz through encode -> 50-step decode -> PSNR must give the reference's `psnr_ref` within the
north star's 1e-3 dB per image; an image folder goes through the reference's Resize -> CenterCrop -> NormalizeToTensor flow (test.py:27-31)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import evaluate as E, synth, weights as W
from selftoktokenizer_amd.config import default_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "pipeline_b16.npz")


@pytest.fixture(scope="module")
def pipe():
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    return SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False)


def test_psnr_of_the_16_golden_images_equals_the_reference_run(pipe):
    g = np.load(GOLD)
    res = E.evaluate(pipe, lambda lo, hi: synth.synthetic_images(hi - lo, first_index=lo), 16, batch=16,
                     noise_fn=lambda lo, hi: synth.synthetic_noise(hi - lo, first_index=lo))
    mine = np.array(res["diffusion"]["psnr_each_dB"])
    d = np.abs(mine - g["psnr_ref"])
    print(f"\nPSNR of 16 images, harness vs the reference pipeline run: mean {mine.mean():.5f} vs {g['psnr_ref'].mean():.5f} dB; |delta| mean {d.mean():.2e} max {d.max():.2e} dB")
    assert d.max() < 1e-3 and abs(res["diffusion"]["psnr_mean_dB"] - float(g["psnr_ref"].mean())) < 5e-4
    # batches of 5 (ragged last batch): the same per-image values up to the fp32 sampler's batch-shape noise
    res5 = E.evaluate(pipe, lambda lo, hi: synth.synthetic_images(hi - lo, first_index=lo), 16, batch=5,
                      noise_fn=lambda lo, hi: synth.synthetic_noise(hi - lo, first_index=lo))
    assert np.abs(np.array(res5["diffusion"]["psnr_each_dB"]) - g["psnr_ref"]).max() < 1e-3
    assert res["token_ids_first_image"] == g["tokens"][0, :8].astype(int).tolist()


def test_image_folder_flow(pipe, tmp_path):
    """files -> Resize(256) -> CenterCrop(256) -> NormalizeToTensor -> encode -> decode: runs, deterministic with a seed, sorted order"""
    from PIL import Image
    rng = np.random.default_rng(0)
    for name, (w, h) in (("b.png", (300, 280)), ("a.jpg", (256, 400)), ("sub/c.png", (512, 512))):
        os.makedirs(os.path.dirname(tmp_path / name), exist_ok=True)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / name)
    paths = E.list_images(str(tmp_path))
    assert [os.path.relpath(p, tmp_path) for p in paths] == ["a.jpg", "b.png", "sub/c.png"]
    load = E.folder_loader(paths, 256)
    assert tuple(load(0, 3).shape) == (3, 3, 256, 256) and float(load(0, 3).abs().max()) <= 1.0
    pipe._steps_backup = None
    r1 = E.evaluate(pipe, load, 3, batch=2, seed=7)
    r2 = E.evaluate(pipe, load, 3, batch=2, seed=7)
    assert r1["diffusion"]["psnr_each_dB"] == r2["diffusion"]["psnr_each_dB"] and len(r1["diffusion"]["psnr_each_dB"]) == 3
    assert all(np.isfinite(v) and 3.0 < v < 60.0 for v in r1["diffusion"]["psnr_each_dB"])
