/* eh_oracle_fast.c -- the TIMED form of the plain-C CPU port  --  TEST INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg).
 *
 * eh_oracle.c is the fp32 CHECKER: one sample at a time, libm calls, double accumulators -- written to be read against the reference
 * (forward src/models/GenericHybridModel.jl:370-431, masked MSE src/losses/loss_fn.jl:61-63, hand VJP SURVEY.md section 8a), and a poor
 * stand-in for "the reference's CPU path on all host cores": 8 GFLOP/s on 128 threads.  This file is the same
 * step -- same layer order, same formulas, same flat-theta layout -- written the way a CPU wants it: SIXTEEN SAMPLES PER BLOCK on the
 * SIMD lanes (`#pragma omp simd` over the sample index; every weight is a broadcast scalar), tanh as the rational Lux itself evaluates
 * for Float32 (NNlib.tanh_fast; no libm call), exp / log through the vector math library, float accumulators per thread folded in
 * double at the end.  Checked against the checker in tests/test_oracle_selfcheck.py (1e-5); what bench.py times.
 *
 * Registry models with a closed inline form here: rbq10 (0), expo (1); others return -1 and the caller falls back to eh_oracle.c.
 * Build (oracle/c_oracle.py): gcc -O3 -march=x86-64-v3 -fopenmp -ffast-math -c   (AVX2: the library travels prebuilt)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXH 4
#define MAXP 8
#define MAXW 256
#define NB 16

typedef struct {          /* == eho_spec of eh_oracle.c */
    int P, NL, hidden[MAXH], K, G;
    int act, scale_nn, mech, n_par;
    int par_kind[MAXP], par_idx[MAXP];
    float lo[MAXP], hi[MAXP], def[MAXP];
    int F, forc_col[4];
    int T, targ_out[4];
} eho_spec;

long eho_n_theta(const eho_spec* s);

static inline int is_nan_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return (u & 0x7fffffffu) > 0x7f800000u; }

/* NNlib.tanh_fast for Float32: x * n(x^2) / d(x^2), sign(x) beyond x^2 = 66 (the function Lux's Dense evaluates; csrc/eh_device.hpp eh_tanh) */
static inline float tanh_fast(float x) {
    const float x2 = x * x;
    const float n = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.587199e-8f, 2.2332108e-5f), 0.0035974074f), 0.1346604f), 1.0f);
    const float d = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 8.7767893e-7f, 0.0003453992f), 0.026262015f), 0.4679937f), 1.0f);
    const float r = x * (n / d);
    return x2 < 66.0f ? r : copysignf(1.0f, x);
}

float eho_loss_and_grad_fast(const eho_spec* s, const float* theta, const float* X, const float* const* forc, const float* const* targ,
                             long B, float* grad, long* n_valid, int nthreads) {
    if (s->mech > 1 || s->T != 1 || s->act > 2) return -1.0f;      /* rbq10 / expo, one target, tanh / sigmoid / relu */
    const long nth = eho_n_theta(s);
    int woff[MAXH + 1], boff[MAXH + 1], dims[MAXH + 2];
    dims[0] = s->P;
    long off = 0;
    for (int l = 0; l <= s->NL; ++l) {
        dims[l + 1] = l < s->NL ? s->hidden[l] : s->K;
        if (dims[l + 1] > MAXW || dims[l] > MAXW) return -1.0f;
        woff[l] = (int)off; off += (long)dims[l + 1] * dims[l];
        boff[l] = (int)off; off += dims[l + 1];
    }
    const long goff = off;
    if (nthreads < 1) nthreads = 1;
    /* valid count of the whole batch first (the mean is over it), in parallel */
    long cnt = 0;
    const float* const y0 = targ[0];
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cnt) num_threads(nthreads) schedule(static)
#endif
    for (long i = 0; i < B; ++i) cnt += !is_nan_bits(y0[i]);
    if (n_valid) n_valid[0] = cnt;
    float phi[MAXP], dphi[MAXP];
    for (int j = 0; j < s->n_par; ++j) {
        phi[j] = s->def[j]; dphi[j] = 0;
        if (s->par_kind[j] == 1) { const float sg = 1.0f / (1.0f + expf(-theta[goff + s->par_idx[j]])); phi[j] = s->lo[j] + (s->hi[j] - s->lo[j]) * sg; dphi[j] = (s->hi[j] - s->lo[j]) * sg * (1 - sg); }
    }
    if (cnt == 0) { memset(grad, 0, (size_t)nth * sizeof(float)); return NAN; }
    const float inv_n = 1.0f / (float)cnt;
    const long nblk = (B + NB - 1) / NB;
    double* const gsum = (double*)calloc((size_t)(nth + 1), sizeof(double));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* const g = (float*)calloc((size_t)(nth + 1), sizeof(float));
        float (*z)[MAXW][NB] = malloc(sizeof(float) * (MAXH + 1) * MAXW * NB);
        float (*h)[MAXW][NB] = malloc(sizeof(float) * (MAXH + 2) * MAXW * NB);
        float (*d)[NB] = malloc(sizeof(float) * MAXW * NB);
        float (*dn)[NB] = malloc(sizeof(float) * MAXW * NB);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long blk = 0; blk < nblk; ++blk) {
            const long i0 = blk * NB;
            const int nb = (int)(B - i0 < NB ? B - i0 : NB);
            for (int c = 0; c < s->P; ++c)
                for (int k = 0; k < NB; ++k) h[0][c][k] = k < nb ? X[(i0 + k) * s->P + c] : 0.0f;
            for (int l = 0; l <= s->NL; ++l) {
                const int o = dims[l + 1], in = dims[l];
                const float* W = theta + woff[l]; const float* b = theta + boff[l];
                for (int r = 0; r < o; ++r) {
                    float a[NB];
#pragma omp simd
                    for (int k = 0; k < NB; ++k) a[k] = b[r];
                    for (int c = 0; c < in; ++c) {
                        const float w = W[r + (long)o * c];
#pragma omp simd
                        for (int k = 0; k < NB; ++k) a[k] = fmaf(w, h[l][c][k], a[k]);
                    }
                    if (l < s->NL) {
                        if (s->act == 0) {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) { z[l][r][k] = a[k]; h[l + 1][r][k] = tanh_fast(a[k]); }
                        } else if (s->act == 1) {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) { z[l][r][k] = a[k]; h[l + 1][r][k] = 1.0f / (1.0f + expf(-a[k])); }
                        } else {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) { z[l][r][k] = a[k]; h[l + 1][r][k] = a[k] > 0.0f ? a[k] : 0.0f; }
                        }
                    } else {
#pragma omp simd
                        for (int k = 0; k < NB; ++k) h[l + 1][r][k] = a[k];
                    }
                }
            }
            /* parameters of the mechanistic model (two of them for the models handled here), sigma-scaled where neural */
            float par[2][NB], sg[2][NB];
            for (int j = 0; j < 2; ++j) {
                if (s->par_kind[j] == 0) {
                    const float* ov = h[s->NL + 1][s->par_idx[j]];
                    const float lo = s->lo[j], sc = s->hi[j] - s->lo[j];
                    if (s->scale_nn) {
#pragma omp simd
                        for (int k = 0; k < NB; ++k) { const float q = 1.0f / (1.0f + expf(-ov[k])); par[j][k] = lo + sc * q; sg[j][k] = sc * q * (1.0f - q); }
                    } else {
#pragma omp simd
                        for (int k = 0; k < NB; ++k) { par[j][k] = ov[k]; sg[j][k] = 1.0f; }
                    }
                } else {
#pragma omp simd
                    for (int k = 0; k < NB; ++k) { par[j][k] = phi[j]; sg[j][k] = 1.0f; }
                }
            }
            const float* const fr = forc[s->forc_col[0]] + i0;
            float dp0[NB], dp1[NB], lsum = 0.0f;
            /* rbq10: y = rb Q10^(0.1 (ta - 15)); expo: y = R0 exp(k T).  d y / d par, the masked residual and d loss / d y in one pass */
#pragma omp simd reduction(+ : lsum)
            for (int k = 0; k < NB; ++k) {
                const float f0 = k < nb ? fr[k] : 0.0f, yo = k < nb ? y0[i0 + k] : NAN;
                float e, p;
                if (s->mech == 0) { e = 0.1f * (f0 - 15.0f); p = expf(e * logf(par[1][k])); }
                else { e = f0; p = expf(par[1][k] * f0); }
                const float y = par[0][k] * p;
                const int ok = !is_nan_bits(yo);
                const float r = ok ? y - yo : 0.0f;
                lsum += r * r * inv_n;
                const float dy = 2.0f * r * inv_n;
                dp0[k] = dy * p;
                dp1[k] = s->mech == 0 ? dy * y * e / par[1][k] : dy * y * f0;
            }
            g[nth] += lsum;
            for (int r = 0; r < s->K; ++r)
                for (int k = 0; k < NB; ++k) d[r][k] = 0.0f;
            for (int j = 0; j < 2; ++j) {
                const float* dp = j == 0 ? dp0 : dp1;
                if (s->par_kind[j] == 0) {
#pragma omp simd
                    for (int k = 0; k < NB; ++k) d[s->par_idx[j]][k] = dp[k] * sg[j][k];
                } else if (s->par_kind[j] == 1) {
                    float t = 0.0f;
#pragma omp simd reduction(+ : t)
                    for (int k = 0; k < NB; ++k) t += dp[k];
                    g[goff + s->par_idx[j]] += t * dphi[j];
                }
            }
            for (int l = s->NL; l >= 0; --l) {
                const int o = dims[l + 1], in = dims[l];
                const float* W = theta + woff[l];
                for (int c = 0; c < in; ++c)
                    for (int k = 0; k < NB; ++k) dn[c][k] = 0.0f;
                for (int r = 0; r < o; ++r) {
                    float t = 0.0f;
#pragma omp simd reduction(+ : t)
                    for (int k = 0; k < NB; ++k) t += d[r][k];
                    g[boff[l] + r] += t;
                    for (int c = 0; c < in; ++c) {
                        const float w = W[r + (long)o * c];
                        float u = 0.0f;
#pragma omp simd reduction(+ : u)
                        for (int k = 0; k < NB; ++k) { u += d[r][k] * h[l][c][k]; dn[c][k] = fmaf(w, d[r][k], dn[c][k]); }
                        g[woff[l] + r + (long)o * c] += u;
                    }
                }
                if (l > 0) {
                    for (int c = 0; c < in; ++c) {
                        if (s->act == 0) {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) d[c][k] = dn[c][k] * (1.0f - h[l][c][k] * h[l][c][k]);
                        } else if (s->act == 1) {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) d[c][k] = dn[c][k] * h[l][c][k] * (1.0f - h[l][c][k]);
                        } else {
#pragma omp simd
                            for (int k = 0; k < NB; ++k) d[c][k] = z[l - 1][c][k] > 0.0f ? dn[c][k] : 0.0f;
                        }
                    }
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (long k = 0; k <= nth; ++k) gsum[k] += (double)g[k];
        free(g); free(z); free(h); free(d); free(dn);
    }
    for (long k = 0; k < nth; ++k) grad[k] = (float)gsum[k];
    const float loss = (float)gsum[nth];
    free(gsum);
    return loss;
}

float eho_loss_and_grad(const eho_spec* s, const float* theta, const float* X, const float* const* forc, const float* const* targ,
                        long B, float* grad, long* n_valid, int nthreads);
void eho_adam(float* theta, float* m, float* v, float* bt, const float* g, long n, float lr, float b1, float b2, float eps);

/* eho_train_steps of eh_oracle.c on the blocked form above (the checker's where the model has no blocked form) */
float eho_train_steps_fast(const eho_spec* s, float* theta, float* m, float* v, float* bt, const float* X, const float* const* forc,
                           const float* const* targ, long N, long batch, long nsteps, float lr, int nthreads) {
    const long nth = eho_n_theta(s);
    float* g = (float*)malloc((size_t)nth * sizeof(float));
    const float* fp[4]; const float* tp[4];
    float loss = NAN;
    long first = 0;
    for (long it = 0; it < nsteps; ++it) {
        if (first + batch > N) first = 0;
        for (int f = 0; f < s->F; ++f) fp[f] = forc[f] + first;
        for (int t = 0; t < s->T; ++t) tp[t] = targ[t] + first;
        long nv[4] = {0, 0, 0, 0};
        loss = eho_loss_and_grad_fast(s, theta, X + first * s->P, fp, tp, batch, g, nv, nthreads);
        if (loss == -1.0f) loss = eho_loss_and_grad(s, theta, X + first * s->P, fp, tp, batch, g, nv, nthreads);
        long tot = 0; for (int t = 0; t < s->T; ++t) tot += nv[t];
        if (tot > 0) eho_adam(theta, m, v, bt, g, nth, lr, 0.9f, 0.999f, 1e-8f);
        first += batch;
    }
    free(g);
    return loss;
}
