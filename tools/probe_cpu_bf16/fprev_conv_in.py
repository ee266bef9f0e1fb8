"""element-level FPRev for conv_in (3 -> 128, 3x3, pad 1) at 256x256"""
import torch, torch.nn.functional as F, sys, numpy as np
IC, OC, H, ks = 3, 128, 256, 3
M = 2.0**60
leaves = [(kh,kw,c) for kh in range(ks) for kw in range(ks) for c in range(IC)]
n = len(leaves)
pairs = [(i,j) for i in range(n) for j in range(n) if i != j]
cells = [(oy, ox) for oy in range(1, H-1, 3) for ox in range(1, H-1, 3)]
x = torch.zeros(1, IC, H, H); w = torch.ones(OC, IC, ks, ks)
for (oy,ox),(i,j) in zip(cells, pairs):
    for li,(kh,kw,c) in enumerate(leaves):
        v = 1.0
        if li == i: v = M
        if li == j: v = -M
        x[0, c, oy-1+kh, ox-1+kw] = v
y = F.conv2d(x.bfloat16(), w.bfloat16(), torch.zeros(OC).bfloat16(), padding=1).float()
mat = np.full((n,n), -1, dtype=int)
for (oy,ox),(i,j) in zip(cells, pairs): mat[i,j] = int(y[0,0,oy,ox].item())
np.set_printoptions(linewidth=500, threshold=1000000)
print("leaf index = (kh*3+kw)*3 + ic; value = #units outside LCA")
print(mat)
