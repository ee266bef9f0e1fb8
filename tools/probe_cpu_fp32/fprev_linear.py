"""FPRev-style probe of the summation tree of torch-CPU `F.linear` in fp32 (MKL sgemm behind at::addmm) on this CPU.
x rows hold unit summands with a +2^40 / -2^40 pair at (i, j); w = ones: the output counts the unit summands that are added
AFTER the pair has cancelled = K - |leaves of the lowest common subtree of i and j|.
    python fprev_linear.py K N rows [bias]
rows = number of x rows in the call (the pairs are cycled through them; MKL picks its kernel by the matrix shape)."""
import sys, numpy as np, torch, torch.nn.functional as F
K, N, ROWS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
use_bias = len(sys.argv) > 4 and sys.argv[4] == "bias"
BIG = 2.0 ** 40
pairs = [(i, j) for i in range(K) for j in range(i + 1, K)]
rng = np.random.default_rng(0)
if len(pairs) > ROWS:
    sel = rng.choice(len(pairs), ROWS, replace=False)
    # always include neighbours and block-boundary pairs
    pairs = [pairs[s] for s in sel]
x = torch.ones(ROWS, K)
for r in range(ROWS):
    i, j = pairs[r % len(pairs)]
    x[r, i], x[r, j] = BIG, -BIG
w = torch.ones(N, K)
b = torch.zeros(N) if use_bias else None
y = F.linear(x, w, b)
assert bool((y == y[:, :1]).all()), "columns differ"
got = y[:, 0].numpy().astype(np.int64)
seq = np.array([K - 1 - max(p) for p in (pairs[r % len(pairs)] for r in range(ROWS))])
print(f"K={K} N={N} rows={ROWS} bias={use_bias}: sequential-chain model matches {int((got == seq).sum())} of {ROWS}")
if (got != seq).any():
    bad = np.nonzero(got != seq)[0][:40]
    for r in bad:
        print("   pair", pairs[r % len(pairs)], "got", got[r], "sequential would give", seq[r])
