/* eh_oracle.c -- plain-C restatement of the EasyHybrid training step  --  TEST INFRASTRUCTURE ONLY.
 *
 * Second, independent CPU restatement (the first is oracle/hybrid_oracle.py) of
 *   forward            src/models/GenericHybridModel.jl:370-431, src/models/NNModels.jl:220-231
 *   masked MSE, agg=sum src/losses/loss_fn.jl:61-63, src/losses/compute_loss.jl:50-53,115-126
 *   hand VJP           SURVEY.md section 8(a)   (the reference differentiates with Zygote)
 *   Adam               Optimisers.jl rule (third party; reference default Adam(0.01), TrainingConfig.jl:43)
 * in fp32, one sample at a time, OpenMP over samples.  It is (a) checked against the NumPy oracle
 * in tests/test_oracle_selfcheck.py and (b) timed by bench.py as `cpu_baseline` (kind "port": the
 * Julia reference cannot run on the box).  Nothing under easyhybrid.jl_amd/ links or loads it.
 * Parity status: pinned on the reference's numeric known-answers through the NumPy oracle it must
 * agree with; gradients / Adam trajectory are PARITY UNPINNED by the reference's own tests.
 *
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC oracle/eh_oracle.c -o oracle/libeh_oracle.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXH 4
#define MAXP 8
#define MAXW 256

typedef struct {
    int P, NL, hidden[MAXH], K, G;
    int act, scale_nn, mech, n_par;
    int par_kind[MAXP], par_idx[MAXP];          /* 0 neural, 1 global, 2 fixed */
    float lo[MAXP], hi[MAXP], def[MAXP];
    int F, forc_col[4];
    int T, targ_out[4];
} eho_spec;

static float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

static float act_f(int a, float z) {
    switch (a) {
        case 0: return tanhf(z);
        case 1: return sigm(z);
        case 2: return z > 0 ? z : 0;
        case 3: return z * sigm(z);
        default: return z;
    }
}
static float act_d(int a, float z, float h) {
    switch (a) {
        case 0: return 1.0f - h * h;
        case 1: return h * (1.0f - h);
        case 2: return z > 0 ? 1.0f : 0.0f;
        case 3: { float s = sigm(z); return s * (1.0f + z * (1.0f - s)); }
        default: return 1.0f;
    }
}

/* mech models: y (one output) and d par given dy */
static float mech_fwd(int mech, const float* par, const float* frc) {
    switch (mech) {
        case 0: return par[0] * powf(par[1], 0.1f * (frc[0] - 15.0f));
        case 1: return par[0] * expf(par[1] * frc[0]);
        case 2: return par[0] * frc[0] + par[1];
        case 3: return par[0] * expf(par[1] * frc[0]) + par[2] * expf(par[3] * frc[0]);
        case 4: { float e = 0.1f * (frc[0] - 15.0f), t = 0; for (int c = 0; c < 3; ++c) t += par[c] * powf(par[3 + c], e); return t; }
    }
    return 0;
}
static void mech_bwd(int mech, const float* par, const float* frc, float y, float dy, float* dpar) {
    switch (mech) {
        case 0: { float e = 0.1f * (frc[0] - 15.0f); float p = powf(par[1], e); dpar[0] = dy * p; dpar[1] = dy * y * e / par[1]; } break;
        case 1: { float ex = expf(par[1] * frc[0]); dpar[0] = dy * ex; dpar[1] = dy * y * frc[0]; } break;
        case 2: dpar[0] = dy * frc[0]; dpar[1] = dy; break;
        case 3: { float ea = expf(par[1] * frc[0]), eb = expf(par[3] * frc[0]);
                  dpar[0] = dy * ea; dpar[1] = dy * par[0] * ea * frc[0]; dpar[2] = dy * eb; dpar[3] = dy * par[2] * eb * frc[0]; } break;
        case 4: { float e = 0.1f * (frc[0] - 15.0f);
                  for (int c = 0; c < 3; ++c) { float p = powf(par[3 + c], e); dpar[c] = dy * p; dpar[3 + c] = dy * par[c] * p * e / par[3 + c]; } } break;
    }
}

long eho_n_theta(const eho_spec* s) {
    long n = 0; int in = s->P;
    for (int l = 0; l <= s->NL; ++l) { int o = l < s->NL ? s->hidden[l] : s->K; n += (long)o * in + o; in = o; }
    return n + s->G;
}

/* X: P x B column-major (one record of P per sample); forc[f][B]; targ[t][B] with NaN = missing.
 * Returns the loss (NaN if no valid target); grad[n_theta]; n_valid[T]. */
float eho_loss_and_grad(const eho_spec* s, const float* theta, const float* X, const float* const* forc, const float* const* targ,
                        long B, float* grad, long* n_valid, int nthreads) {
    const long nth = eho_n_theta(s);
    int woff[MAXH + 1], boff[MAXH + 1], dims[MAXH + 2];
    dims[0] = s->P;
    long off = 0;
    for (int l = 0; l <= s->NL; ++l) {
        dims[l + 1] = l < s->NL ? s->hidden[l] : s->K;
        woff[l] = (int)off; off += (long)dims[l + 1] * dims[l];
        boff[l] = (int)off; off += dims[l + 1];
    }
    const long goff = off;
    /* pass 1: valid counts (the mean is over the valid samples of the whole batch) */
    double cnt[4] = {0, 0, 0, 0};
    for (int t = 0; t < s->T; ++t) { long c = 0; for (long i = 0; i < B; ++i) c += !isnan(targ[t][i]); cnt[t] = (double)c; if (n_valid) n_valid[t] = c; }
    float phi[MAXP], dphi[MAXP];
    for (int j = 0; j < s->n_par; ++j) {
        phi[j] = s->def[j]; dphi[j] = 0;
        if (s->par_kind[j] == 1) { float sg = sigm(theta[goff + s->par_idx[j]]); phi[j] = s->lo[j] + (s->hi[j] - s->lo[j]) * sg; dphi[j] = (s->hi[j] - s->lo[j]) * sg * (1 - sg); }
    }
    if (nthreads < 1) nthreads = 1;
    double* gpart = (double*)calloc((size_t)nthreads * (nth + 1), sizeof(double));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* g = gpart + (size_t)tid * (nth + 1);
        float z[MAXH + 1][MAXW], h[MAXH + 2][MAXW], d[MAXW], dn[MAXW];
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long i = 0; i < B; ++i) {
            for (int k = 0; k < s->P; ++k) h[0][k] = X[i * s->P + k];
            for (int l = 0; l <= s->NL; ++l) {
                const int o = dims[l + 1], in = dims[l];
                const float* W = theta + woff[l]; const float* b = theta + boff[l];
                for (int r = 0; r < o; ++r) {
                    float a = b[r];
                    for (int c = 0; c < in; ++c) a += W[r + (long)o * c] * h[l][c];
                    z[l][r] = a;
                    h[l + 1][r] = l < s->NL ? act_f(s->act, a) : a;
                }
            }
            float par[MAXP], sg[MAXP], frc[4], dpar[MAXP];
            for (int j = 0; j < s->n_par; ++j) {
                par[j] = phi[j]; sg[j] = 1;
                if (s->par_kind[j] == 0) {
                    float ov = h[s->NL + 1][s->par_idx[j]];
                    if (s->scale_nn) { float q = sigm(ov); par[j] = s->lo[j] + (s->hi[j] - s->lo[j]) * q; sg[j] = (s->hi[j] - s->lo[j]) * q * (1 - q); }
                    else par[j] = ov;
                }
            }
            for (int f = 0; f < 4; ++f) frc[f] = s->forc_col[f] >= 0 ? forc[s->forc_col[f]][i] : 0;
            const float y = mech_fwd(s->mech, par, frc);
            float dy = 0;
            for (int t = 0; t < s->T; ++t) {
                const float yo = targ[t][i];
                if (!isnan(yo) && cnt[t] > 0) { const float r = y - yo; g[nth] += (double)(r * r) / cnt[t]; dy += 2.0f * r / (float)cnt[t]; }
            }
            if (dy == 0) continue;
            mech_bwd(s->mech, par, frc, y, dy, dpar);
            for (int r = 0; r < s->K; ++r) d[r] = 0;
            for (int j = 0; j < s->n_par; ++j) {
                if (s->par_kind[j] == 0) d[s->par_idx[j]] = dpar[j] * sg[j];
                else if (s->par_kind[j] == 1) g[goff + s->par_idx[j]] += dpar[j] * dphi[j];
            }
            for (int l = s->NL; l >= 0; --l) {
                const int o = dims[l + 1], in = dims[l];
                const float* W = theta + woff[l];
                for (int c = 0; c < in; ++c) dn[c] = 0;
                for (int r = 0; r < o; ++r) {
                    const float dr = d[r];
                    g[boff[l] + r] += dr;
                    for (int c = 0; c < in; ++c) { g[woff[l] + r + (long)o * c] += dr * h[l][c]; dn[c] += W[r + (long)o * c] * dr; }
                }
                if (l > 0) for (int c = 0; c < in; ++c) d[c] = dn[c] * act_d(s->act, z[l - 1][c], h[l][c]);
            }
        }
    }
    double loss = 0;
    for (long k = 0; k < nth; ++k) { double a = 0; for (int t = 0; t < nthreads; ++t) a += gpart[(size_t)t * (nth + 1) + k]; grad[k] = (float)a; }
    for (int t = 0; t < nthreads; ++t) loss += gpart[(size_t)t * (nth + 1) + nth];
    free(gpart);
    double ct = 0; for (int t = 0; t < s->T; ++t) ct += cnt[t];
    return ct > 0 ? (float)loss : NAN;
}

/* Optimisers.Adam, fp32 op for op; bt = running (beta1^t, beta2^t), starts at (beta1, beta2) */
void eho_adam(float* theta, float* m, float* v, float* bt, const float* g, long n, float lr, float b1, float b2, float eps) {
    for (long k = 0; k < n; ++k) {
        m[k] = b1 * m[k] + (1.0f - b1) * g[k];
        v[k] = b2 * v[k] + (1.0f - b2) * (g[k] * g[k]);
        theta[k] -= m[k] / (1.0f - bt[0]) / (sqrtf(v[k] / (1.0f - bt[1])) + eps) * lr;
    }
    bt[0] *= b1; bt[1] *= b2;
}

/* nsteps train steps over contiguous batches of `batch` samples cycling through N; returns last loss */
float eho_train_steps(const eho_spec* s, float* theta, float* m, float* v, float* bt, const float* X, const float* const* forc,
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
        long nv[4];
        loss = eho_loss_and_grad(s, theta, X + first * s->P, fp, tp, batch, g, nv, nthreads);
        long tot = 0; for (int t = 0; t < s->T; ++t) tot += nv[t];
        if (tot > 0) eho_adam(theta, m, v, bt, g, nth, lr, 0.9f, 0.999f, 1e-8f);
        first += batch;
    }
    free(g);
    return loss;
}
