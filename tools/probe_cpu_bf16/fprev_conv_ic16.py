"""element-level FPRev of the decoder's conv_in (16 -> 512, 3x3, bf16) on this CPU: leaves = (kh, kw, ic), 144 of them; value = unit summands
added after the +M / -M pair cancelled.  Prints the inferred chain structure."""
import torch, torch.nn.functional as F, numpy as np, sys
IC, OC, H = 16, 512, 32
leaves = [(kh, kw, ic) for kh in range(3) for kw in range(3) for ic in range(IC)]
n = len(leaves); M = 2.0 ** 60
pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
cells = [(oy, ox) for oy in range(1, H - 1, 3) for ox in range(1, H - 1, 3)]          # 100 non-overlapping receptive fields per image
B = (len(pairs) + len(cells) - 1) // len(cells)
x = torch.zeros(B, IC, H, H); w = torch.ones(OC, IC, 3, 3)
for c, (i, j) in enumerate(pairs):
    b, r = divmod(c, len(cells)); oy, ox = cells[r]
    for li, (kh, kw, ic) in enumerate(leaves):
        x[b, ic, oy - 1 + kh, ox - 1 + kw] = M if li == i else (-M if li == j else 1.0)
y = F.conv2d(x.bfloat16(), w.bfloat16(), torch.zeros(OC).bfloat16(), padding=1).float()
mat = np.full((n, n), -1, int)
for c, (i, j) in enumerate(pairs):
    b, r = divmod(c, len(cells)); oy, ox = cells[r]
    mat[i, j] = mat[j, i] = int(y[b, 0, oy, ox])
np.set_printoptions(linewidth=250, threshold=100000)
print("row 0 (pair (0, j)):", mat[0])
print("row 1:", mat[1])
print("row 16:", mat[16])
print("row 32:", mat[32])
np.save("/tmp/p/fprev_ic16.npy", mat)
