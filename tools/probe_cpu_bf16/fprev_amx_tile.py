import torch, torch.nn.functional as F, sys
IC = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
OC = 16
M = 2.0**30
n = IC*ks*ks
# pixels: one per (i,j) pair; for 1x1 conv each pixel independent
pairs = [(i,j) for i in range(n) for j in range(n) if i != j]
P = len(pairs)
x = torch.ones(P, n)
for p,(i,j) in enumerate(pairs):
    x[p,i] = M; x[p,j] = -M
W = 64
H = (P + W - 1)//W
xp = torch.ones(H*W, n); xp[:P] = x
xi = xp.reshape(1, H, W, IC).permute(0,3,1,2).contiguous()
w = torch.ones(OC, IC, 1, 1)
y = F.conv2d(xi.bfloat16(), w.bfloat16(), torch.zeros(OC).bfloat16()).float()
r = y[0,0].reshape(-1)[:P]
import numpy as np
mat = np.full((n,n), -1, dtype=int)
for p,(i,j) in enumerate(pairs): mat[i,j] = int(r[p].item())
np.set_printoptions(linewidth=400, threshold=100000)
print(mat)
