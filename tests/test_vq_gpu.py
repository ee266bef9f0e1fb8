"""-m gpu: the HIP VQ kernels (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch

from selftoktokenizer_amd import ops, synth, weights as W

pytestmark = pytest.mark.gpu


def _codebook():
    return W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous()


@pytest.fixture(scope="module")
def cb():
    return _codebook()


def _check(z, cb, packed):
    """packed=True checks BOTH packed kernels: the f16 coarse pass + exact re-score (default) and the fp32-input MFMA kernel"""
    from oracle import clib
    ids_ref, best_ref = clib.vq_encode(z.numpy(), cb.numpy())
    zc, cbc = z.cuda(), cb.cuda()
    cbk = ops.vq_pack_codebook(cbc) if packed else cbc
    for coarse in ((3, 1, False) if packed else (None,)):
        ids, best = ops.vq_encode(zc, cbk, packed=packed, return_best=True, coarse=coarse)
        torch.cuda.synchronize()
        ids, best = ids.cpu().numpy(), best.cpu().numpy()
        assert ids.dtype == np.int64
        np.testing.assert_array_equal(ids, ids_ref, err_msg=f"coarse={coarse}")
        nan = np.isnan(best_ref)
        np.testing.assert_array_equal(np.isnan(best), nan)
        np.testing.assert_array_equal(best[~nan].view(np.uint32), best_ref[~nan].view(np.uint32), err_msg=f"coarse={coarse}")


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("n", [1, 63, 512, 2048 + 17])
def test_vq_bit_exact_vs_oracle(cb, n, packed):
    z = synth.synthetic_vq_rows(n, seed=0xC0DE + n)
    _check(z, cb, packed)


@pytest.mark.parametrize("packed", [False, True])
def test_vq_edge_rows(cb, packed):
    """ties -> lowest index, all-zero row -> id 0, NaN / inf rows -> first NaN (id 0)."""
    cb2 = cb.clone()
    cb2[12345] = cb2[77]            # exact duplicate codes
    cb2[31000] = cb2[77]
    z = synth.synthetic_vq_rows(64, seed=99)
    z[0] = cb2[77] * 3.0            # best match is the duplicated code -> must return 77
    z[1] = 0.0                      # zero row: every score 0 -> id 0
    z[2, 5] = float("nan")
    z[3, 0] = float("inf")
    z[4] = -cb2[0]                  # negative scores around
    cb2[64 + 5] = cb2[64 + 2]       # duplicates inside ONE 32-code tile, held by different wave halves
    cb2[64 + 30] = cb2[64 + 2]
    z[5] = cb2[64 + 2] * 0.7        # -> 66
    cb2[20000 + 9] = cb2[20000 + 12]   # same tile, the higher slot sits in the lower half
    z[6] = cb2[20000 + 12] * 2.0    # -> 20009
    _check(z, cb2, packed)
    zc = z.cuda()
    cbk = ops.vq_pack_codebook(cb2.cuda()) if packed else cb2.cuda()
    ids = ops.vq_encode(zc, cbk, packed=packed).cpu()
    assert ids[0].item() == 77 and ids[1].item() == 0 and ids[2].item() == 0 and ids[3].item() == 0
    assert ids[5].item() == 66 and ids[6].item() == 20009


@pytest.mark.parametrize("packed", [False, True])
def test_vq_nan_code(cb, packed):
    """a NaN inside the codebook: that code's score is NaN for every row -> every id is that code."""
    cb2 = cb.clone()
    cb2[4000, 3] = float("nan")
    cb2[9000, 1] = float("nan")
    z = synth.synthetic_vq_rows(130, seed=5)
    _check(z, cb2, packed)


def test_vq_coarse_pass_error_bound_and_adversarial_near_ties(cb):
    """The f16 coarse pass is exact only because its error stays inside the window the finalize re-scores (F16_EPS = 2^-17 in
    csrc/vq.hip).  (i) measure |coarse - canonical| with the same arithmetic through hipBLASLt (fp16 hi/lo of the 2^7-scaled
    operands, three products, fp32 accumulate); (ii) rows built to sit ON decision boundaries: midpoints of code pairs in the
    same and in different tiles / halves / splits, so that the top two canonical scores differ by a few ulps or tie exactly."""
    from oracle import clib
    x = torch.nn.functional.normalize(synth.synthetic_vq_rows(4096, seed=0xB0D), dim=-1).cuda()
    e = cb.cuda()

    def split(t):
        ts = t * 128.0
        hi = ts.half()
        return hi, (ts - hi.float()).half()
    x0, x1 = split(x)
    e0, e1 = split(e)
    coarse = (torch.mm(x0, e0.t(), out_dtype=torch.float32) + torch.mm(x0, e1.t(), out_dtype=torch.float32)
              + torch.mm(x1, e0.t(), out_dtype=torch.float32)) / 16384.0
    exact = torch.from_numpy(clib.vq_scores(x.cpu().numpy(), cb.numpy()[:4096])).cuda()
    err = float((coarse[:, :4096] - exact).abs().max())
    print(f"max |coarse - canonical| over 1.7e7 scores: {err:.3e}  (F16_EPS = 2^-17 = 7.6e-6)")
    assert err < 2.0 ** -17 / 8
    # the one-MFMA pass (SELFTOK_VQ_F16COARSE1): hi x hi only, window constant F16_EPS1 = 17 x 2^-14
    coarse1 = torch.mm(x0, e0.t(), out_dtype=torch.float32) / 16384.0
    err1 = float((coarse1[:, :4096] - exact).abs().max())
    print(f"max |hi x hi - canonical| over 1.7e7 scores: {err1:.3e}  (F16_EPS1 = 17 x 2^-14 = 1.04e-3; worst-case bound 9.9e-4)")
    assert err1 < 17 * 2.0 ** -14
    # adversarial rows
    pairs = [(5, 9), (5, 5 + 4), (7, 33), (100, 32000), (31, 32), (0, 32767), (12345, 12345 + 16), (2048, 2048 + 1024)]
    rows = []
    for a, b in pairs:
        mid = cb[a] + cb[b]
        rows += [mid, mid * 0.37, mid + 1e-7 * cb[a], mid - 1e-7 * cb[a], mid + 3e-6 * cb[b]]
    cb2 = cb.clone()
    cb2[20001] = cb2[77]                                   # exact duplicates far apart: lowest index must win through the re-score
    rows += [cb2[77] * 2.0, cb2[77] + cb2[20001]]
    z = torch.stack(rows)
    for book in (cb, cb2):
        ids_ref, best_ref = clib.vq_encode(z.numpy(), book.numpy())
        pk = ops.vq_pack_codebook(book.cuda())
        for sp in (0, 1, 2, 64):
            for nm in (3, 1):
                ids, best = ops.vq_encode(z.cuda(), pk, packed=True, return_best=True, coarse=nm, split=sp)
                np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref, err_msg=f"split={sp} mfmas={nm}")
                np.testing.assert_array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32))
    # rows whose two best codes sit INSIDE the one-MFMA window (gaps 1e-5 .. 1e-3: invisible to the 2^-17 window, candidates for the
    # wide one) in the same tile / another tile of the stream (whole-stream re-scan) / another stream
    rows = []
    for a, b in ((5, 9), (5, 5 + 32 * 3), (7, 7 + 2048 * 5), (100, 100 + 4), (31, 32)):
        for t in (1e-5, 1e-4, 5e-4, 9e-4, 2e-3):
            rows.append(torch.nn.functional.normalize(cb[a] + cb[b], dim=-1) + t * cb[a])
    # three (four) codes of ONE stream and wave half in different tiles, all inside the window: second-tile re-score and whole-stream walk
    for trio in ((5, 5 + 32, 5 + 64), (9, 9 + 32 * 7, 9 + 32 * 40, 9 + 32 * 63), (2048 + 17, 2048 + 17 + 32 * 2, 2048 + 17 + 32 * 9)):
        for t in (0.0, 2e-4, 1e-3):
            rows.append(torch.nn.functional.normalize(sum(cb[c] for c in trio), dim=-1) + t * cb[trio[-1]])
    z = torch.stack(rows)
    ids_ref, best_ref = clib.vq_encode(z.numpy(), cb.numpy())
    pk = ops.vq_pack_codebook(cb.cuda())
    for nm in (3, 1):
        for sp in (0, 1, 4, 16):
            ids, best = ops.vq_encode(z.cuda(), pk, packed=True, return_best=True, coarse=nm, split=sp)
            np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref, err_msg=f"near-window rows, split={sp} mfmas={nm}")
            np.testing.assert_array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32))


def test_vq_int32_ids_and_batch_shape(cb):
    z = synth.synthetic_vq_rows(4 * 512, seed=3).reshape(4, 512, 16).cuda()
    cbc = cb.cuda()
    a = ops.vq_encode(z, cbc)
    b = ops.vq_encode(z, ops.vq_pack_codebook(cbc), packed=True, ids_dtype=torch.int32)
    assert a.shape == (4, 512) and b.dtype == torch.int32
    assert torch.equal(a, b.long())


def test_vq_full_size_properties(cb):
    """BASELINE config 2 size (N = 64*512): VALU and MFMA kernels agree bit-for-bit, the top-1 score
    equals the oracle's on a strided sample, and re-encoding a chosen code returns the same vector."""
    n = 64 * 512
    z = synth.synthetic_vq_rows(n, seed=0xBA5E).cuda()
    cbc = cb.cuda()
    ids_v, best_v = ops.vq_encode(z, cbc, return_best=True)
    ids_m, best_m = ops.vq_encode(z, ops.vq_pack_codebook(cbc), packed=True, return_best=True)
    assert torch.equal(ids_v, ids_m)
    assert torch.equal(best_v.view(torch.int32), best_m.view(torch.int32))
    # idempotence: codes are unit-norm, so encoding a code vector must return (a duplicate of) itself
    again = ops.vq_encode(cbc[ids_m], cbc)
    assert torch.equal(cbc[again], cbc[ids_m])
    # strided sample against the CPU oracle
    from oracle import clib
    sel = torch.arange(0, n, 16)
    ids_ref, best_ref = clib.vq_encode(z[sel].cpu().numpy(), cb.numpy())
    np.testing.assert_array_equal(ids_m[sel].cpu().numpy(), ids_ref)
    np.testing.assert_array_equal(best_m[sel].cpu().numpy().view(np.uint32), best_ref.view(np.uint32))


_ORACLE_CACHE = {}


def _oracle_full(n, cb):
    """every row through the C oracle (threaded over row chunks), cached per N for the launch-shape sweep"""
    if n not in _ORACLE_CACHE:
        from oracle import clib
        z = synth.synthetic_vq_rows(n, seed=0x5EED + n)
        z[n // 3] = 0.0                                   # an all-zero row and an exact-code row inside every size
        z[n // 2] = cb[(7 * n) % 32768] * 1.5
        _ORACLE_CACHE[n] = (z,) + tuple(clib.vq_encode_mt(z.numpy(), cb.numpy()))
    return _ORACLE_CACHE[n]


@pytest.mark.parametrize("n", [8192, 12288, 16384, 32768, 65536, 131072])
def test_vq_mfma_every_launch_shape_full_n(cb, n):
    """BASELINE row counts (B*K for configs[1..3]) and the sizes in between, EVERY row against the oracle, for every wave-tile
    variant vq_mfma_kernel<1|2|4> (the automatic choice is <2> for 8192 <= N < 32768) and several code-split counts
    (1 = no split, odd, maximal); the inline-asm max3 scan behind the MFMAs must give the same bits in all of them."""
    z, ids_ref, best_ref = _oracle_full(n, cb)
    zc = z.cuda()
    pk = ops.vq_pack_codebook(cb.cuda())
    combos = [(0, 0)] + [(rt, sp) for rt in (1, 2, 4) for sp in (0, 1, 7, 64)]
    if n > 32768:                                         # keep the big sizes to the variants that differ in code path
        combos = [(0, 0), (1, 0), (2, 7), (4, 64), (4, 1)]
    for coarse in (3, 1, False):                          # f16 coarse pass (3 / 1 MFMAs) + exact re-score / fp32-input MFMA kernel
        for rt, sp in combos:
            ids, best = ops.vq_encode(zc, pk, packed=True, return_best=True, rt=rt, split=sp, coarse=coarse)
            torch.cuda.synchronize()
            assert np.array_equal(ids.cpu().numpy(), ids_ref), f"ids differ at N={n} rt={rt} split={sp} coarse={coarse}"
            assert np.array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32)), f"score bits differ at N={n} rt={rt} split={sp} coarse={coarse}"
    ids_v = ops.vq_encode(zc, cb.cuda())                  # the generic VALU kernel on the same rows
    assert np.array_equal(ids_v.cpu().numpy(), ids_ref)


def test_code_gather_ln(cb):
    ids = torch.from_numpy(synth.synthetic_token_ids(3, 512)).cuda()
    cbc = cb.cuda()
    w = synth.hash_uniform(1, (16,), 0.9, 1.1).cuda()
    b = synth.hash_uniform(2, (16,), -0.1, 0.1).cuda()
    out = ops.code_gather_ln(ids, cbc, w, b)
    ref = torch.nn.functional.layer_norm(cb[ids.cpu()], (16,), w.cpu(), b.cpu(), 1e-6)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)
    plain = ops.code_gather_ln(ids.int(), cbc)
    assert torch.equal(plain.cpu(), cb[ids.cpu()])
    # every integer wire format reads the same codes (a uint16 / int16 buffer must not be read as int64)
    for dt in (torch.int16, torch.uint8, torch.int64):
        small = (ids % 200).to(dt) if dt != torch.int64 else ids % 200
        assert torch.equal(ops.code_gather_ln(small, cbc).cpu(), cb[(ids % 200).cpu()])
    u16 = torch.from_numpy((ids.cpu().numpy() % 32768).astype(np.uint16)).cuda()
    assert torch.equal(ops.code_gather_ln(u16, cbc).cpu(), cb[ids.cpu()])
    with pytest.raises(TypeError):
        ops.code_gather_ln(ids.float(), cbc)


def test_vq_empty_and_bad_arguments(cb):
    cbc = cb.cuda()
    pk = ops.vq_pack_codebook(cbc)
    for packed, c in ((False, cbc), (True, pk)):
        ids = ops.vq_encode(torch.zeros(0, 16, device="cuda"), c, packed=packed)
        assert ids.shape == (0,) and ids.dtype == torch.int64
    assert ops.code_gather_ln(torch.zeros(0, 512, dtype=torch.int64, device="cuda"), cbc).shape == (0, 512, 16)
    from selftoktokenizer_amd._lib import SelftokHipError
    with pytest.raises(SelftokHipError):
        ops.vq_encode(torch.zeros(4, 8, device="cuda"), cbc)            # wrong code dim
    with pytest.raises(SelftokHipError):
        ops.vq_pack_codebook(cbc[:100])                                  # C % 32 != 0 for the MFMA layout
    ids = ops.vq_encode(synth.synthetic_vq_rows(7).cuda(), cbc[:100])    # ... but the generic kernel takes any C
    from oracle import clib
    np.testing.assert_array_equal(ids.cpu().numpy(), clib.vq_encode(synth.synthetic_vq_rows(7).numpy(), cb[:100].numpy())[0])


def test_vq_coarse_path_guards_its_unit_norm_premise(cb):
    """the f16 coarse pass is exact only for |x|, |e| <= 1 (F16_EPS is derived for unit vectors).  A code book that
    is not l2-normalised, or caller-normalised rows (prenormed=True) that are not unit length, must not silently return a wrong id:
    the pack step / the kernel flag them and the exact scan runs.  Checked against the C oracle, ids and score bits."""
    from oracle import clib
    z = synth.synthetic_vq_rows(700, seed=0xBADC0DE)
    # (a) code book of norm ~10 (rows of very different norms), rows normalised by the kernel
    scale = (1.0 + 9.0 * synth.hash_uniform(0x5CA1E, (cb.shape[0], 1), 0.0, 1.0))
    cb10 = (cb * scale).contiguous()
    ids_ref, best_ref = clib.vq_encode(z.numpy(), cb10.numpy())
    pk = ops.vq_pack_codebook(cb10.cuda())
    for coarse in (3, 1, False):
        ids, best = ops.vq_encode(z.cuda(), pk, packed=True, return_best=True, coarse=coarse)
        np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref, err_msg=f"norm-10 code book, coarse={coarse}")
        np.testing.assert_array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32))
    # (b) unit code book, prenormed=True with rows that are NOT unit length (norm ~4 .. 40): the coarse error scales with |x|
    zz = (z * 10.0).contiguous()
    ids_ref, best_ref = clib.vq_encode(zz.numpy(), cb.numpy(), normalize=False)
    pk1 = ops.vq_pack_codebook(cb.cuda())
    for coarse in (3, 1, False):
        ids, best = ops.vq_encode(zz.cuda(), pk1, packed=True, return_best=True, prenormed=True, coarse=coarse)
        np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref, err_msg=f"non-unit prenormed rows, coarse={coarse}")
        np.testing.assert_array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32))
    # (c) honest prenormed rows (unit length by the canonical l2norm) still take the fast path and agree
    zn = torch.from_numpy(clib.l2norm16(z.numpy()))
    ids_ref, best_ref = clib.vq_encode(zn.numpy(), cb.numpy(), normalize=False)
    ids, best = ops.vq_encode(zn.cuda(), pk1, packed=True, return_best=True, prenormed=True, coarse=True)
    np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref)
    np.testing.assert_array_equal(best.cpu().numpy().view(np.uint32), best_ref.view(np.uint32))
