"""not gpu: the error bound the f16 coarse pass of the VQ lookup rests on (csrc/vq.hip: |s' - s| < F16_EPS = 2^-17 between the
three-MFMA coarse score and the canonical fp32 score, for unit-norm rows and codes), checked on a numpy emulation of that
arithmetic -- operands x 2^7, fp16 hi + unscaled fp16 lo, s' = x0 e0 + x0 e1 + x1 e0 accumulated in fp32 -- including adversarial
inputs (components of very different magnitude, near-duplicates of a code).  The GPU test measures the real kernels; this one
pins the arithmetic the proof is about, with no GPU."""
import numpy as np

F16_EPS = 2.0 ** -17
PRESCALE = np.float32(128.0)


def unit(v):
    v = v.astype(np.float32)
    n = np.sqrt((v.astype(np.float64) ** 2).sum(-1, keepdims=True)).astype(np.float32)
    return (v / np.maximum(n, np.float32(1e-12))).astype(np.float32)


def hi_lo(x):
    xs = x * PRESCALE
    x0 = xs.astype(np.float16)
    x1 = (xs - x0.astype(np.float32)).astype(np.float16)             # unscaled residual
    return x0.astype(np.float32), x1.astype(np.float32)


def coarse(z, e):
    z0, z1 = hi_lo(z)
    e0, e1 = hi_lo(e)
    s = z0 @ e0.T                                                     # chained fp32 accumulation of exact fp16 products
    s = s + z0 @ e1.T
    s = s + z1 @ e0.T
    return s / (PRESCALE * PRESCALE)


def canonical(z, e):                                                  # the oracle's k-ordered fp32 FMA chain, emulated in fp64 -> fp32 per step
    acc = np.zeros((z.shape[0], e.shape[0]), np.float32)
    for k in range(z.shape[1]):
        acc = (acc.astype(np.float64) + z[:, k:k + 1].astype(np.float64) * e[:, k].astype(np.float64)[None, :]).astype(np.float32)
    return acc


def test_coarse_score_within_the_window_random_and_adversarial():
    rng = np.random.default_rng(0)
    D, C = 16, 4096
    e = unit(rng.standard_normal((C, D)))
    rows = [unit(rng.standard_normal((512, D)))]
    skew = rng.standard_normal((256, D)) * np.exp(rng.uniform(-12, 0, (256, D)))          # components spread over 5 decades
    rows.append(unit(skew))
    rows.append(unit(e[:256] + 1e-4 * rng.standard_normal((256, D)).astype(np.float32)))  # near-duplicates of codes: scores ~ 1
    onehot = np.zeros((16, D), np.float32)
    onehot[np.arange(16), np.arange(16)] = 1.0
    rows.append(onehot)
    z = np.concatenate(rows)
    err = np.abs(coarse(z, e).astype(np.float64) - canonical(z, e).astype(np.float64))
    print("max |coarse - canonical| = %.3e (window F16_EPS = %.3e)" % (err.max(), F16_EPS))
    assert err.max() < F16_EPS / 3          # the analysis says < 2.4e-6 worst case; the window is 3x that
    # and the selection rule built on it: the true argmax is always among the codes within 2 eps of the coarse maximum
    s_c, s_t = coarse(z, e), canonical(z, e)
    true_arg = s_t.argmax(1)
    assert (s_c[np.arange(len(z)), true_arg] >= s_c.max(1) - 2 * F16_EPS).all()


F16_EPS1 = 17 * 2.0 ** -14          # csrc/vq.hip: the window constant of the one-MFMA coarse pass (hi x hi only)


def coarse1(z, e):
    z0, _ = hi_lo(z)
    e0, _ = hi_lo(e)
    return (z0 @ e0.T) / (PRESCALE * PRESCALE)


def test_one_mfma_coarse_score_within_its_window_random_and_adversarial():
    """SELFTOK_VQ_F16COARSE1: |x.e - x0.e0| <= sum |x_k e_k| (2^-11 + 2^-11 (1 + 2^-11)) <= 2^-10 (1 + 2^-12) |x| |e|: 9.9e-4 at norm^2 = 1.01;
    worst-case-ish inputs push every component's fp16 rounding error in the same direction (components just above a power of two plus 3/4 ulp)"""
    rng = np.random.default_rng(1)
    D, C = 16, 4096
    e = unit(rng.standard_normal((C, D)))
    rows = [unit(rng.standard_normal((1024, D))), unit(rng.standard_normal((256, D)) * np.exp(rng.uniform(-12, 0, (256, D)))),
            unit(e[:256] + 1e-4 * rng.standard_normal((256, D)).astype(np.float32))]
    # adversarial: every component of x and of the matching code rounds DOWN by almost half an fp16 ulp -> errors add up with one sign
    base = np.full((64, D), 0.25, np.float32) * (1 + (1023.49 / 1024) * 2.0 ** -11 * np.arange(1, 65, dtype=np.float32)[:, None] / 64)
    rows.append(unit(base))
    z = np.concatenate(rows)
    e_adv = np.concatenate([e, unit(base)])
    err = np.abs(coarse1(z, e_adv).astype(np.float64) - canonical(z, e_adv).astype(np.float64))
    print("max |hi x hi - canonical| = %.3e (window constant F16_EPS1 = %.3e, analytic bound 9.9e-4)" % (err.max(), F16_EPS1))
    assert err.max() < 9.9e-4 < F16_EPS1
    s_c, s_t = coarse1(z, e_adv), canonical(z, e_adv)
    true_arg = s_t.argmax(1)
    assert (s_c[np.arange(len(z)), true_arg] >= s_c.max(1) - 2 * F16_EPS1).all()
