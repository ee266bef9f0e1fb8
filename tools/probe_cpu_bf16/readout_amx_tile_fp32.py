"""exact fp32 readout of one AMX tile product (32 channels) through a cancelling second chunk"""
import torch, torch.nn.functional as F, numpy as np
torch.manual_seed(5)
def bf(x): return x.bfloat16().float()
def trunc_bf16(x):
    return (x.view(torch.int32) & ~0xFFFF).view(torch.float32)
H, OC = 128, 16
P = H*H
xs = bf(torch.randn(P, 32)); ws = bf(torch.randn(32)*0.3)
xd, wd = xs.double(), ws.double()
def chain(idx):
    t = torch.zeros(P)
    for k in idx: t = (t.double() + xd[:, k]*wd[k]).float()
    return t
te, to = chain(range(0,32,2)), chain(range(1,32,2))
hyp = {"even+odd": (te.double()+to.double()).float(),
       "seq": chain(range(32)),
       }
t_em = hyp["even+odd"]
a_hi = trunc_bf16(t_em); r = t_em - a_hi; a_lo = trunc_bf16(r*256.0)
x = torch.zeros(P, 64); x[:, :32] = xs; x[:, 32] = -a_hi; x[:, 33] = -a_lo
w = torch.zeros(OC, 64); w[:, :32] = ws[None]; w[:, 32] = 1.0; w[:, 33] = 2.0**-8
xi = x.reshape(1, H, H, 64).permute(0,3,1,2).contiguous()
y = F.conv2d(xi.bfloat16(), w.reshape(OC,64,1,1).bfloat16(), torch.zeros(OC).bfloat16()).float()[0,0].reshape(-1)
t16 = a_hi.double() + a_lo.double()/256.0
t_true = (y.double() + t16)
print("readout is fp32-representable:", (t_true.float().double() == t_true).float().mean().item())
tt = t_true.float()
for k,v in hyp.items():
    d = (tt != v)
    print(k, "mismatch", d.float().mean().item(), "max ulps", ((tt.view(torch.int32) - v.view(torch.int32)).abs().max().item()))
torch.save({"xs": xs, "ws": ws, "t_true": tt}, "readout32.pt")
