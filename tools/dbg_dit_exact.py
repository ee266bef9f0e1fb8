"""debug helper (GPU box): block-0 intermediates of the exact-order MMDiT against tests/golden/_dbg_dit.npz (made in the build container by a script that is
not part of the repo); prints the first tensor that differs"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops, synth, weights as W
from selftoktokenizer_amd.encoder import QformerEncoderGPU
from selftoktokenizer_amd.mmdit import MMDiTGPU, DIT_HIDDEN as H
from selftoktokenizer_amd.pipeline import _Flow
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "_dbg_dit.npz"))
sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
dev = torch.device("cuda", 0)
enc = QformerEncoderGPU(sd, dev, 512, mode="exact")
dit = MMDiTGPU(sd, dev, 512, gemm="exact")
B = 16


def cmp(name, t, ref):
    a = t.detach().cpu().numpy().reshape(ref.shape)
    bad = (a.view(np.uint32) != ref.view(np.uint32)) & ~((a == 0) & (ref == 0))
    print(f"{name:8s} differing {int(bad.sum()):8d} of {a.size:8d}   max abs diff {np.abs(a - ref).max():.3e}", flush=True)


ehs = enc.codes_ln(torch.from_numpy(synth.synthetic_token_ids(B)).cuda())
cmp("ehs", ehs[:2], d["ehs"])
x = synth.synthetic_noise(B, device="cuda")
flow = _Flow(50, 1.0, dev)
tf = flow.t_freq_exact[0:1].expand(B, -1).contiguous()
cmp("tfreq", tf[:1], d["tfreq"])
c = dit.time_embed(tf)
cmp("c", c, d["c"])
xe = dit.embed_image(x)
cmp("xe", xe[:2], d["xe"])
ctx = dit.embed_context(ehs)
cmp("ctx", ctx[:2], d["ctx"])
cmp("tab0", dit.ctx_tables[0], d["tab0"])
mods_x, mods_c_last, mods_f = dit.modulations(c, True)
cmp("modx", mods_x[0], d["modx"])
t0 = dit.ctx_tables[0]
_, cn = dit._ln(None, ctx, shift=t0[:, 0:H], scale=t0[:, H:2 * H])
cmp("cn", cn[:1], d["cn"])
cqkv = dit.lin("model.joint_blocks.0.context_block.attn.qkv", cn)
cmp("cqkv", cqkv[:1], d["cqkv"])
_, xn = dit._ln(None, xe, shift=mods_x[0][:, 0:H], scale=mods_x[0][:, H:2 * H], per_sample=True)
cmp("xn", xn[:1], d["xn"])
xqkv = dit.lin("model.joint_blocks.0.x_block.attn.qkv", xn)
cmp("xqkv", xqkv[:1], d["xqkv"])
oc = ops.ex_attention(cqkv[..., :H], cqkv[..., H:2 * H], cqkv[..., 2 * H:], 24, xqkv[..., H:2 * H], xqkv[..., 2 * H:], slots1=512)
ox = ops.ex_attention(xqkv[..., :H], cqkv[..., H:2 * H], cqkv[..., 2 * H:], 24, xqkv[..., H:2 * H], xqkv[..., 2 * H:], slots1=512)
cmp("ca", oc[:1], d["ca"]); cmp("xa", ox[:1], d["xa"])
mx = mods_x[0]
ctx1, _ = dit._res_ln(None, ctx, "model.joint_blocks.0.context_block.attn.proj", oc, gate=t0[:, 2 * H:3 * H], gate_per_sample=False, shift=t0[:, 3 * H:4 * H], scale=t0[:, 4 * H:5 * H])
cmp("ctx1", ctx1[:1], d["ctx1"])
x1, xn2 = dit._res_ln(None, xe, "model.joint_blocks.0.x_block.attn.proj", ox, gate=mx[:, 2 * H:3 * H], gate_per_sample=True, shift=mx[:, 3 * H:4 * H], scale=mx[:, 4 * H:5 * H], per_sample=True)
cmp("x1", x1[:1], d["x1"])
h = dit.lin("model.joint_blocks.0.x_block.mlp.fc1", xn2, gelu=True)
cmp("xh", h[:1, :, :1024], d["xh"])
x2, _ = dit._res_ln(None, x1, "model.joint_blocks.0.x_block.mlp.fc2", h, gate=mx[:, 5 * H:6 * H], gate_per_sample=True, shift=mx[:, 0:H], scale=mx[:, H:2 * H], per_sample=True)
cmp("x2", x2[:1], d["x2"])
