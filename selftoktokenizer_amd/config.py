"""YAML -> attribute dict, the drop-in of `mimogpt.infer.infer_utils.parse_args_from_yaml`
(reference infer_utils.py:12-19,165-168 builds an EasyDict; easydict is not a dependency here)."""
from __future__ import annotations

import yaml


class AttrDict(dict):
    """dict with attribute access, recursively (EasyDict semantics for the keys the pipeline touches)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(AttrDict(x) if isinstance(x, dict) and not isinstance(x, AttrDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def parse_args_from_yaml(yml_path: str) -> AttrDict:
    with open(yml_path, "r") as fd:
        return AttrDict(yaml.load(fd.read(), Loader=yaml.FullLoader))


def default_config(K: int = 512, renderer: bool = False) -> AttrDict:
    """The two shipped configs (configs/res256/256-eval.yml, configs/renderer/renderer-eval.yml) reduced to the
    keys the hot path reads, plus the assumed K=1024 split (the reference ships no 1024-token config;
    SURVEY.md section 7: k_per_stage doubled)."""
    if renderer:
        stages, kps = "1000", str(K)
    elif K == 512:
        stages, kps = "200,400,600,800,1000", "192,184,72,48,16"
    else:
        f = K // 512
        stages, kps = "200,400,600,800,1000", ",".join(str(int(v) * f) for v in "192,184,72,48,16".split(","))
    return AttrDict({
        "common": {"is_eval": True},
        "tokenizer": {"params": {
            "image_size": 256, "k": K, "stages": stages, "k_per_stage": kps, "in_channels": 16,
            "encoder_hidden_size": 16, "diffusion_type": "flow",
            "noise_schedule_config": {"schedule": "log_norm", "parameterization": "velocity", "force_recon": False, "m": 0.0, "s": 1.0},
            "enc": "Enc-Qformer-Uni-XL/2", "enable_enc_variable_size": True,
            "encoder_config": {"time_adaln": True, "qformer_mode": "dual", "pre_norm": False, "post_norm": True,
                               "xavier_init": False, "qk_norm": False, "attn_mask": False},
            "quantizer_config": {"codebook_size": 32768, "code_dim": 16, "K": K},
            "model": "MMDiT_XL_Renderer" if renderer else "MMDiT_XL", "context_see_xt": not renderer,
            "decoder_config": {"time_adaln": "pos_emb"},
        }},
    })
