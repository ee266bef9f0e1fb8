"""not gpu: the split-activation chunk layout (include/selftok_hip.h, csrc/common.h split_blk_index) as the host side sees it:
SplitAct.planes() must invert the documented index formula, for ragged row counts and multi-dimensional logical shapes."""
import numpy as np
import pytest
import torch

from selftoktokenizer_amd.ops import SplitAct


def blk_index(row, k, plane, KT):          # the formula of include/selftok_hip.h, in halfs
    return (((row // 16) * KT + k // 32) * 2 + plane) * 512 + (row % 16) * 32 + k % 32


@pytest.mark.parametrize("shape", [(16, 32), (5, 64), (3, 7, 96), (2, 300, 1536), (1, 1, 32)])
def test_planes_invert_the_chunk_index(shape):
    sa = SplitAct(shape, "cpu", zero=True)
    K = shape[-1]
    rows = int(np.prod(shape[:-1]))
    assert sa.rows == rows and sa.data.shape == ((rows + 15) // 16, K // 32, 2, 16, 32) and sa.dtype == torch.float16
    assert sa.data.numel() * 2 == ((rows + 15) // 16) * 16 * K * 4            # selftok_split_f16x2_bytes
    flat = sa.data.reshape(-1)
    rng = np.random.default_rng(0)
    picks = [(int(rng.integers(rows)), int(rng.integers(K)), int(rng.integers(2))) for _ in range(200)] + [(rows - 1, K - 1, 1), (0, 0, 0)]
    for i, (r, k, p) in enumerate(picks):
        flat[blk_index(r, k, p, K // 32)] = float(i % 1000 + 1)
    planes = sa.planes()
    assert planes.shape == (2, *shape)
    pl = planes.reshape(2, rows, K)
    last = {}
    for i, (r, k, p) in enumerate(picks):
        last[(r, k, p)] = float(i % 1000 + 1)
    for (r, k, p), v in last.items():
        assert float(pl[p, r, k]) == v
    assert int(torch.count_nonzero(pl)) == len(last)


def test_k_must_be_a_multiple_of_32():
    with pytest.raises(AssertionError):
        SplitAct((4, 48), "cpu")
