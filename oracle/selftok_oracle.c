/*
 * oracle/selftok_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, scalar) of the integer/bit-exact pieces of the Selftok
 * encode/decode hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported baseline.
 *
 * Pinning: tools/oracle/gen_golden.py imports the reference (read-only) in the build
 * container and checks these functions bit-for-bit against
 *   mimogpt/models/selftok/vector_quantize_pytorch.py (CosineSimCodebook.forward eval,
 *   l2norm, gumbel_sample eval, get_codes_from_indices) on torch-CPU; the resulting vectors
 * are committed under tests/golden/ (vq_*.npz) and re-checked by tests/test_oracle_vq.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; fmaf() is used explicitly where the
 * reference arithmetic is a fused multiply-add, never implicitly).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define SELFTOK_ORACLE_D 16

/* l2norm = F.normalize(t, p=2, dim=-1, eps=1e-12)
 * reference: vector_quantize_pytorch.py:51-52 (l2norm), called at :854.
 * torch-CPU's vectorised norm kernel over a 16-float row is, bit for bit, 8 interleaved
 * accumulators a_j = fma(z[j+8], z[j+8], z[j]*z[j]) combined left to right, then
 * x = z / max(sqrt(s), eps)  (probe: SURVEY.md section 8a row a10; re-verified by gen_golden.py). */
void selftok_oracle_l2norm16(const float *z, float *x, int64_t n)
{
    for (int64_t r = 0; r < n; ++r) {
        const float *zr = z + r * SELFTOK_ORACLE_D;
        float a[8];
        for (int j = 0; j < 8; ++j)
            a[j] = fmaf(zr[j + 8], zr[j + 8], zr[j] * zr[j]);
        float s = a[0];
        for (int j = 1; j < 8; ++j)
            s = s + a[j];
        float nrm = sqrtf(s);
        if (!(nrm > 1e-12f)) nrm = 1e-12f;   /* clamp_min(eps); NaN stays NaN via the division below */
        if (s != s) nrm = s;
        for (int k = 0; k < SELFTOK_ORACLE_D; ++k)
            x[r * SELFTOK_ORACLE_D + k] = zr[k] / nrm;
    }
}

/* score(x, e) = einsum('h n d,h c d->h n c')  -- reference vector_quantize_pytorch.py:561.
 * MKL sgemm with K=16 evaluates each element as a k=0..15 sequential fp32 FMA chain from 0
 * (independent of row count / thread count; probe in SURVEY.md 8a, re-verified by gen_golden.py). */
static inline float score16(const float *x, const float *e)
{
    float s = 0.0f;
    for (int k = 0; k < SELFTOK_ORACLE_D; ++k)
        s = fmaf(x[k], e[k], s);
    return s;
}

/* ids = argmax_c score  -- reference gumbel_sample eval branch, vector_quantize_pytorch.py:125-143:
 * torch.argmax returns the FIRST maximal index; a NaN score compares as the maximum and the
 * first NaN wins.
 * z: [n,16] pre-normalisation features (output of project_in, :844); codebook: [C,16].
 * ids: [n] int64 (the reference's dtype); best: optional [n] top-1 score (may be NULL). */
void selftok_oracle_vq_encode(const float *z, const float *codebook, int64_t *ids, float *best,
                              int64_t n, int64_t C, int normalize)
{
    for (int64_t r = 0; r < n; ++r) {
        float x[SELFTOK_ORACLE_D];
        if (normalize)
            selftok_oracle_l2norm16(z + r * SELFTOK_ORACLE_D, x, 1);
        else
            memcpy(x, z + r * SELFTOK_ORACLE_D, sizeof(x));
        float bv = score16(x, codebook);
        int64_t bi = 0;
        int isnan_b = (bv != bv);
        for (int64_t c = 1; c < C && !isnan_b; ++c) {
            float s = score16(x, codebook + c * SELFTOK_ORACLE_D);
            if (s != s) { bv = s; bi = c; isnan_b = 1; break; }
            if (s > bv) { bv = s; bi = c; }
        }
        ids[r] = bi;
        if (best) best[r] = bv;
    }
}

/* full score row (for small-case debugging / gap statistics): out[n,C] */
void selftok_oracle_vq_scores(const float *x, const float *codebook, float *out, int64_t n, int64_t C)
{
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < C; ++c)
            out[r * C + c] = score16(x + r * SELFTOK_ORACLE_D, codebook + c * SELFTOK_ORACLE_D);
}

/* codes = codebook[ids]  -- reference get_codes_from_indices, vector_quantize_pytorch.py:787-794
 * (project_out is Identity because codebook_dim == output_dim == 16, :680-681). */
void selftok_oracle_code_gather(const int64_t *ids, const float *codebook, float *out, int64_t n)
{
    for (int64_t r = 0; r < n; ++r)
        memcpy(out + r * SELFTOK_ORACLE_D, codebook + ids[r] * SELFTOK_ORACLE_D, SELFTOK_ORACLE_D * sizeof(float));
}

/* torch.linspace(start, end, steps) fp32 on CPU -- reference rectified_flow.py:67.
 * ATen's CPU kernel: step = (end-start)/(steps-1); first half counts up from start with a fused
 * multiply-add, second half counts down from end (probe SURVEY.md 8a row a13; re-verified). */
void selftok_oracle_linspace(float start, float end, int steps, float *out)
{
    float step = (end - start) / (float)(steps - 1);
    int half = steps / 2;
    for (int i = 0; i < steps; ++i) {
        if (i < half)
            out[i] = fmaf(step, (float)i, start);
        else
            out[i] = end - step * (float)(steps - 1 - i);
    }
}

/* DiTi_cont.to_indices on integer timesteps -- reference diti_utils.py:73-103.
 * t is `timestep_map[i].long()` (rectified_flow.py:203): ind starts at 0 (int64) and every
 * segment with t-low >= 0 overwrites it with trunc(slope*(t-low)) + base; slope is a Python
 * float (double), slope*xp is computed by torch as int64 tensor * double scalar -> promoted to
 * float32 (default dtype) before the truncating cast back to int64. */
int64_t selftok_oracle_diti_index(int64_t t, const int *stages /*[n+1], stages[0]=0*/, const int *k_per_stage,
                                  int n_stages, int K)
{
    int64_t ind = 0, acc = 0;
    for (int i = 0; i < n_stages; ++i) {
        int64_t xp = t - stages[i];
        double slope = (double)k_per_stage[i] / (double)(stages[i + 1] - stages[i]);
        if (xp >= 0) {
            float v = (float)xp * (float)slope;   /* int64 tensor * python float -> fp32 tensor */
            ind = (int64_t)v + acc;
        }
        acc += k_per_stage[i];
    }
    if (ind < 0) ind = 0;
    if (ind > K - 1) ind = K - 1;
    return ind;
}
