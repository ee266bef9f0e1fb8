"""block boundaries of MKL sgemm's K loop behind F.linear (fp32, torch CPU): rows r hold a +2^40 at k=0 and -2^40 at k=r+1 among
unit summands; the output = unit summands added after the pair cancels.  Inside the first block that is K_block_end-1-j (+ the later
blocks); across blocks the pair cancels when the partial sums are combined.
    python fprev_blocks.py K N M [threads]"""
import sys, numpy as np, torch, torch.nn.functional as F
K, N, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4: torch.set_num_threads(int(sys.argv[4]))
BIG = 2.0 ** 40
x = torch.ones(M, K)
js = [(r % (K - 1)) + 1 for r in range(M)]
for r, j in enumerate(js):
    x[r, 0], x[r, j] = BIG, -BIG
y = F.linear(x, torch.ones(N, K), torch.zeros(N))
assert bool((y == y[:, :1]).all())
got = y[:, 0].numpy().astype(np.int64)
# all rows with the same j must agree (row-position independence)
per_j = {}
for r, j in enumerate(js):
    per_j.setdefault(j, set()).add(int(got[r]))
multi = {j: v for j, v in per_j.items() if len(v) > 1}
prof = np.array([min(per_j[j]) for j in range(1, K)])
# boundaries: where prof stops decreasing by exactly 1
d = np.diff(prof)
cuts = [int(j + 2) for j in np.nonzero(d != -1)[0]]
print(f"K={K} N={N} M={M} threads={torch.get_num_threads()}: row-dependent results at {len(multi)} positions; block starts (k index) {cuts[:20]}; got(0,1)={prof[0]} got(0,K-1)={prof[-1]}")
