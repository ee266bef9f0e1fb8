/* candidate models of MKL sgemm's per-element arithmetic behind torch-CPU F.linear (fp32); probe helper, build container only.
   out[m][n] = bias[n] (+) sum_k x[m][k] w[n][k], K split into blocks whose partial sums are sequential fmaf chains. */
#include <math.h>
#include <stdint.h>
/* blocks: KC = 384; a remainder in (384, 768) is split into two halves (first = ceil half rounded to `halfround`) */
static int next_block(int rem, int tail_rule) {
    if (rem <= 384) return rem;
    if (rem < 768) { if (tail_rule == 0) return (rem + 1) / 2; if (tail_rule == 1) return 384; }
    return 384;
}
void linear_model(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, int variant, int tail_rule) {
#pragma omp parallel for
    for (int m = 0; m < M; m++)
        for (int n = 0; n < N; n++) {
            const float *a = x + (long)m * K, *b = w + (long)n * K;
            float c = bias ? bias[n] : 0.0f;
            int k0 = 0, first = 1;
            while (k0 < K) {
                int kb = next_block(K - k0, (k0 == 0 && K < 768) ? 0 : tail_rule);
                float acc = (variant == 1) ? c : 0.0f;
                if (variant == 2 && !first) acc = c;
                for (int k = k0; k < k0 + kb; k++) acc = fmaf(a[k], b[k], acc);
                if (variant == 1 || (variant == 2 && !first)) c = acc; else c = c + acc;
                k0 += kb; first = 0;
            }
            out[(long)m * N + n] = c;
        }
}
