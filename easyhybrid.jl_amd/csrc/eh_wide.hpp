// Wide-layer variant of the fused step kernel (hidden widths up to 16*NBH = 128).
//
// Same math, data layout, parameter image and slab contract as eh_step_kernel (eh_device.hpp), but a
// different decomposition: with 128-wide layers neither the weight-gradient accumulators
// (128x128 floats = 256 registers per lane) nor a per-wave activation workspace fit, so the four
// waves of a workgroup share ONE macro-tile of 16*NT samples and split every layer by OUTPUT ROWS:
// wave w owns feature blocks [w*MB, (w+1)*MB), MB = NBH/4.  Each wave computes its slice of every
// layer's outputs (B operands = the previous layer's activations, read from the shared LDS image in
// MFMA operand order), of every dH, and owns the matching slice of every weight gradient, so no
// gradient ever has to be summed across waves; the price is a workgroup barrier per layer.  The
// output layer is split over K (each wave contracts its own feature blocks, the four partial
// outputs meet in the one-sample-per-lane mechanistic stage, which wave 0 runs).
#pragma once
#include "eh_device.hpp"

// per-wave accumulator order for the host-built scatter map (cmap[wave][k][lane][r])
struct EhWideLayout { int mb, kw0, kwh, kwo, kb, kbo, na; };
__host__ __device__ constexpr EhWideLayout eh_wide_layout(int nbi, int nbh, int nl) {
    EhWideLayout L{};
    L.mb = nbh / 4;
    L.kw0 = 0;
    L.kwh = L.kw0 + L.mb * nbi;
    L.kwo = L.kwh + (nl - 1) * L.mb * nbh;
    L.kb = L.kwo + L.mb;
    L.kbo = L.kb + nl * L.mb;
    L.na = L.kbo + 1;
    return L;
}

template <int NBI, int NBH, int NL, int NT>
struct EhWideGeom : EhGeom<NBI, NBH, NL, NT, 1> {
    using B = EhGeom<NBI, NBH, NL, NT, 1>;
    static_assert(NBH % 4 == 0, "four waves split the feature blocks");
    static_assert(4 * 16 * B::SR <= B::HP * B::SR, "the split-K output partials alias the delta image");
    static constexpr int TOTAL_FLOATS = B::IMG_FLOATS + B::WAVE_WS;     // one shared workspace per workgroup
};

template <int NBI, int NBH, int NL, int NT, int ACT, int MODE>
__global__ __launch_bounds__(256, 1) void eh_wide_kernel(const EhNet net, const EhStepArgs a) {
    using G = EhWideGeom<NBI, NBH, NL, NT>;
    constexpr int MT = G::MT, SR = G::SR, HP = G::HP, S0 = G::S0, SH = G::SH, MB = NBH / 4;
    constexpr bool TRAIN = MODE == EH_MODE_TRAIN;
    constexpr EhWideLayout WL = eh_wide_layout(NBI, NBH, NL);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    float* const ws = smem + G::IMG_FLOATS;
    float* const XS = ws + G::XS_OFF;
    float* const HS = ws + G::HS_OFF;
    float* const DZ = ws + G::DZ_OFF;
    float* const OS = ws + G::OS_OFF;
    float* const OSP = DZ;                      // [4 waves][16][SR] partial outputs of the split-K output layer
    const float* const meta = wl + G::PHI_OFF;
    const int m0 = wave * MB;
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    const bool mechw = wave == 0 && lane < MT;   // the lanes that own one sample each in the mechanistic stage

    // one sample record per mechanistic lane, fetched one tile ahead
    constexpr int NX4 = (G::IP + 3) / 4;
    struct { f32x4 x[NX4]; float frc[EH_MAX_FORC]; float y[EH_MAX_TARG]; } nx;
    const int count = (int)a.count, first = (int)a.first;
    const int ntiles = (count + MT - 1) / MT;
    auto fetch = [&](int tile) {
        const int n_loc = tile * MT + lane;
        const bool live = mechw && (tile < ntiles) && (n_loc < count);
        const int n_glb = live ? (a.idx ? a.idx[first + n_loc] : first + n_loc) : 0;
        const float* const rec = a.recs + (long long)n_glb * a.C;
        if ((a.C & 3) == 0) {
#pragma unroll
            for (int q = 0; q < NX4; ++q) nx.x[q] = (live && 4 * q < net.P) ? *(const f32x4*)(rec + 4 * q) : f32x4{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int q = 0; q < NX4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) nx.x[q][e] = (live && 4 * q + e < net.P) ? rec[4 * q + e] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) {
            const unsigned col = (net.forc_col >> (8 * f)) & 0xFFu;
            nx.frc[f] = (col != 0xFFu && live) ? rec[net.P + col] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) nx.y[t] = (t < net.T && live) ? rec[net.P + net.F + t] : __builtin_nanf("");
    };
    fetch((int)blockIdx.x);

    for (int e = 4 * tid; e < G::IMG_FLOATS; e += 4 * 256) *(f32x4*)&wl[e] = *(const f32x4*)&a.image[e];
    for (int e = tid; e < G::IP * SR; e += 256) XS[e] = 0.0f;
    for (int e = tid; e < 16 * SR; e += 256) OS[e] = 0.0f;          // rows >= K of the output / dO image stay 0
    __syncthreads();
    if (a.bn_part) {       // input BatchNorm, train mode: statistics of this minibatch (see eh_step_kernel)
        if (tid < net.P) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const float m = (float)count, c0 = a.bn_part[a.bn_nblk * 64 + tid];
            const float d = s1 / m, var = fmaxf(s2 / m - d * d, 0.0f), mu = c0 + d;
            wl[G::PHI_OFF + EH_IMG_BNM + tid] = mu;
            wl[G::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(var + EH_BN_EPS);
            if (a.bn_update && blockIdx.x == 0) {
                const float rm = (1.0f - EH_BN_MOMENTUM) * a.bn_run[tid] + EH_BN_MOMENTUM * mu;
                const float rv = (1.0f - EH_BN_MOMENTUM) * a.bn_run[32 + tid] + EH_BN_MOMENTUM * (m > 1.0f ? m / (m - 1.0f) : 1.0f) * var;
                a.bn_run[tid] = rm; a.bn_run[32 + tid] = rv;
                a.image_out[G::PHI_OFF + EH_IMG_BNM + tid] = rm;
                a.image_out[G::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(rv + EH_BN_EPS);
            }
        }
        __syncthreads();
    }

    // accumulators: this wave's row slice of every weight gradient
    f32x4 aW0[MB][NBI], aWh[NL > 1 ? NL - 1 : 1][MB][NBH], aWo[MB], aB[NL][MB], aBo = f32x4{0, 0, 0, 0};
    float gacc[EH_MAX_PARAMS], lacc = 0.0f, syacc = 0.0f, syyacc = 0.0f, cacc[EH_MAX_TARG], est[EH_MAX_TARG][EH_EVAL_STATS];
#pragma unroll
    for (int mm = 0; mm < MB; ++mm) {
#pragma unroll
        for (int n = 0; n < NBI; ++n) aW0[mm][n] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int l = 0; l < NL - 1; ++l)
#pragma unroll
            for (int n = 0; n < NBH; ++n) aWh[l][mm][n] = f32x4{0, 0, 0, 0};
        aWo[mm] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int l = 0; l < NL; ++l) aB[l][mm] = f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) gacc[j] = 0.0f;
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        cacc[t] = 0.0f;
#pragma unroll
        for (int k = 0; k < EH_EVAL_STATS; ++k) est[t][k] = 0.0f;
    }
    const int ksteps0 = (net.P + 3) / 4;
    const int ksK = net.K < 4 ? net.K : 4;

    // B operand of an MFMA whose k index runs over the features of block q: lane (c, g), step s <-> feature 16q+4g+s, sample 16t+c
    auto load_b = [&](const float* img, int q, f32x4 (&bq)[NT], bool unswish) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v = img[(16 * q + 4 * g + s) * SR + 16 * t + c];
                if (ACT == EH_ACT_SWISH && unswish) v = v * eh_sigmoid(v);      // the image holds z for swish
                bq[t][s] = v;
            }
    };

    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int n_loc = tile * MT + lane;
        const bool live = mechw && (n_loc < count);
        // ---- 1. records -> normalised [feature][sample] image (wave 0), next tile's record in flight
        float frc[EH_MAX_FORC], yobs[EH_MAX_TARG];
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) frc[f] = nx.frc[f];
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) yobs[t] = nx.y[t];
        if (mechw) {
#pragma unroll
            for (int q = 0; q < NX4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < net.P) XS[(4 * q + e) * SR + lane] = (nx.x[q][e] - meta[EH_IMG_BNM + 4 * q + e]) * meta[EH_IMG_BNR + 4 * q + e];
        }
        fetch(tile + (int)gridDim.x);
        __syncthreads();

        // ---- 2. layer 0, this wave's output blocks ----------------------------------------------
        {
            f32x4 acc[MB][NT];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                const int m = m0 + mm;
                const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + 16 * m + 4 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mm][t] = bias;
            }
            for (int ks = 0; ks < ksteps0; ++ks) {
                float bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = XS[(4 * ks + g) * SR + 16 * t + c];
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const float av = wl[G::W0_OFF + (16 * (m0 + mm) + c) * S0 + 4 * ks + g];
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[mm][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[mm][t][r];
                        HS[(16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c] = (ACT == EH_ACT_SWISH && TRAIN) ? z : eh_act<ACT>(z);
                    }
        }
        __syncthreads();
        // ---- 3. hidden layers -------------------------------------------------------------------
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            const float* Hp = HS + (l - 1) * HP * SR;
            float* Hl = HS + l * HP * SR;
            f32x4 acc[MB][NT];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + l * HP + 16 * (m0 + mm) + 4 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mm][t] = bias;
            }
#pragma unroll
            for (int q = 0; q < NBH; ++q) {
                f32x4 bq[NT];
                load_b(Hp, q, bq, TRAIN);
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const f32x4 a4 = *(const f32x4*)&W[(16 * (m0 + mm) + c) * SH + 16 * q + 4 * g];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], bq[t][s], acc[mm][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[mm][t][r];
                        Hl[(16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c] = (ACT == EH_ACT_SWISH && TRAIN) ? z : eh_act<ACT>(z);
                    }
            __syncthreads();
        }
        // ---- 4. output layer, split over K: this wave contracts its own feature blocks -------------
        {
            const float* W = wl + G::WO_OFF;
            const float* Hp = HS + (NL - 1) * HP * SR;
            f32x4 o[NT];
            const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + NL * HP + 4 * g];
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = wave == 0 ? bias : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int qq = 0; qq < MB; ++qq) {
                const int q = m0 + qq;
                f32x4 bq[NT];
                load_b(Hp, q, bq, TRAIN);
                const f32x4 a4 = *(const f32x4*)&W[c * SH + 16 * q + 4 * g];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], bq[t][s], o[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) OSP[(wave * 16 + 4 * g + r) * SR + 16 * t + c] = o[t][r];
        }
        __syncthreads();
        // ---- 5. mechanistic model + masked loss: wave 0, one sample per lane -----------------------
        if (mechw) {
            float par[EH_MAX_PARAMS], sg[EH_MAX_PARAMS], dydp[EH_MAX_PARAMS];
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                par[j] = meta[EH_IMG_PHI + j]; sg[j] = 1.0f; dydp[j] = 0.0f;
                if (j < net.n_par && pkind(j) == EH_PAR_NEURAL) {
                    const int k = pidx(j);
                    const float ov = (OSP[k * SR + lane] + OSP[(16 + k) * SR + lane]) + (OSP[(32 + k) * SR + lane] + OSP[(48 + k) * SR + lane]);
                    if (net.scale_nn) {
                        const float s = eh_sigmoid(ov), sc = meta[EH_IMG_SC + j];
                        par[j] = fmaf(sc, s, meta[EH_IMG_LO + j]);
                        sg[j] = sc * s * (1.0f - s);
                    } else {
                        par[j] = ov;
                    }
                }
            }
            const float y0 = eh_mech_eval(net.mech, par, frc, dydp);
            float yx[2] = {0.0f, 0.0f}, Jx[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
            if (net.n_out > 1) eh_mech_extra(net.mech, par, frc, yx, Jx);
            float dy = 0.0f, dyx[2] = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t) {
                if (t < net.T) {
                    const int ot = (int)((net.targ_out >> (2 * t)) & 3u);
                    const float y = ot == 0 ? y0 : (ot == 1 ? yx[0] : yx[1]);
                    const bool valid = live && !__builtin_isnan(yobs[t]);
                    const float r = valid ? y - yobs[t] : 0.0f;
                    if constexpr (TRAIN) {
                        const float w = a.inv_n ? a.inv_n[t] : 1.0f;
                        const float cy = valid ? yobs[t] - a.shift[t] : 0.0f;
                        float d;
                        if (net.loss == EH_LOSS_MAE) { lacc += w * fabsf(r); d = r > 0.0f ? w : (r < 0.0f ? -w : 0.0f); }
                        else { lacc += w * r * r; d = 2.0f * w * r; }
                        dy += ot == 0 ? d : 0.0f; dyx[0] += ot == 1 ? d : 0.0f; dyx[1] += ot == 2 ? d : 0.0f;
                        cacc[t] += valid ? 1.0f : 0.0f;
                        syacc += cy; syyacc += cy * cy;
                    } else if (valid) {
                        const float cy = yobs[t] - a.shift[t], ch = y - a.shift[t];
                        est[t][0] += r * r; est[t][1] += cy; est[t][2] += cy * cy; est[t][3] += 1.0f;
                        est[t][4] += ch; est[t][5] += ch * ch; est[t][6] += ch * cy; est[t][7] += fabsf(r);
                    }
                }
            }
            if constexpr (!TRAIN) {
                if (live) {
                    if (a.yhat)
                        for (int t = 0; t < net.T; ++t) {
                            const int o = (int)((net.targ_out >> (2 * t)) & 3u);
                            a.yhat[(long long)t * a.yld + n_loc] = o == 0 ? y0 : (o == 1 ? yx[0] : yx[1]);
                        }
                    if (a.pout)
                        for (int j = 0; j < net.n_par; ++j) a.pout[(long long)j * a.yld + n_loc] = par[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                    if (j < net.n_par) {
                        float dp = dy * dydp[j];
                        if (j < 3) dp += dyx[0] * Jx[0][j] + dyx[1] * Jx[1][j];
                        dp = live ? dp : 0.0f;
                        const int kd = pkind(j);
                        if (kd == EH_PAR_NEURAL) OS[pidx(j) * SR + lane] = dp * sg[j];
                        else if (kd == EH_PAR_GLOBAL) gacc[j] += dp;
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (!TRAIN) continue;

        // ---- 6. backward through the output layer ------------------------------------------------
        f32x4 dzr[MB][NT];
        {
            f32x4 dO[NT], aT[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[t][r] = OS[(4 * g + r) * SR + 16 * t + c];
                aT[t] = *(const f32x4*)&OS[c * SR + 16 * t + 4 * g];
                if (wave == 0) aBo += dO[t];
            }
            const float* Hl = HS + (NL - 1) * HP * SR;
            const float* W = wl + G::WO_OFF;
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                const int m = m0 + mm;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 b4 = *(const f32x4*)&Hl[(16 * m + c) * SR + 16 * t + 4 * g];
                    if (ACT == EH_ACT_SWISH) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[s] = b4[s] * eh_sigmoid(b4[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) aWo[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aWo[mm], 0, 0, 0);
                }
                f32x4 dh[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dh[t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < ksK) {
                        const float av = W[(4 * g + s) * SH + 16 * m + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t) dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dO[t][s], dh[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dzr[mm][t][r] = dh[t][r] * eh_dact<ACT>(Hl[(16 * m + 4 * g + r) * SR + 16 * t + c]);
                    aB[NL - 1][mm] += dzr[mm][t];
                }
            }
        }
        // the split-K partials (aliasing DZ) were last read in step 5: safe to overwrite now
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) DZ[(16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c] = dzr[mm][t][r];
        __syncthreads();
        // ---- 7. hidden layers backward -------------------------------------------------------------
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            const float* Hp = HS + (l - 1) * HP * SR;
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            // dW_l[own rows][all columns] += dZ_l (own rows) * H_{l-1}^T
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                f32x4 aT[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) aT[t] = *(const f32x4*)&DZ[(16 * (m0 + mm) + c) * SR + 16 * t + 4 * g];
#pragma unroll
                for (int n = 0; n < NBH; ++n)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 b4 = *(const f32x4*)&Hp[(16 * n + c) * SR + 16 * t + 4 * g];
                        if (ACT == EH_ACT_SWISH) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) b4[s] = b4[s] * eh_sigmoid(b4[s]);
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            aWh[l - 1][mm][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aWh[l - 1][mm][n], 0, 0, 0);
                    }
            }
            // dH_{l-1}[own rows] = W_l^T dZ_l  (k runs over ALL rows of dZ_l: the shared image)
            f32x4 dn[MB][NT];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) dn[mm][t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < NBH; ++q) {
                f32x4 bq[NT];
                load_b(DZ, q, bq, false);
#pragma unroll
                for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float av = W[(16 * q + 4 * g + s) * SH + 16 * (m0 + mm) + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t) dn[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[t][s], dn[mm][t], 0, 0, 0);
                    }
            }
            __syncthreads();                         // every wave is done reading dZ_l
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c;
                        const float d = dn[mm][t][r] * eh_dact<ACT>(Hp[ad]);
                        dzr[mm][t][r] = d;
                        DZ[ad] = d;
                    }
                    aB[l - 1][mm] += dzr[mm][t];
                }
            __syncthreads();
        }
        // ---- 8. layer 0: dW0[own rows] += dZ_0 * X^T -------------------------------------------------
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            f32x4 aT[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) aT[t] = *(const f32x4*)&DZ[(16 * (m0 + mm) + c) * SR + 16 * t + 4 * g];
#pragma unroll
            for (int n = 0; n < NBI; ++n)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 b4 = *(const f32x4*)&XS[(16 * n + c) * SR + 16 * t + 4 * g];
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW0[mm][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aW0[mm][n], 0, 0, 0);
                }
        }
        __syncthreads();                             // XS / OS / images are rewritten by the next tile
    }

    // ---- 9. one partial per workgroup: the waves own disjoint entries, so they go straight to the slab
    float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
    if constexpr (!TRAIN) {
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t)
#pragma unroll
                for (int k = 0; k < EH_EVAL_STATS; ++k) {
                    const float v = eh_wave_sum(est[t][k]);
                    if (t < net.T && lane == 0) out[t * EH_EVAL_STATS + k] = v;
                }
        }
        return;
    }
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) aB[l][mm][r] = eh_row16_sum(aB[l][mm][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) aBo[r] = eh_row16_sum(aBo[r]);
    struct I4 { int x, y, z, w; };
    const I4* const cm = reinterpret_cast<const I4*>(a.cmap) + (long long)wave * WL.na * 64;
    auto putc = [&](int k, const f32x4& v) {
        const I4 ix = cm[k * 64 + lane];
        if (ix.x >= 0) out[ix.x] = v[0];
        if (ix.y >= 0) out[ix.y] = v[1];
        if (ix.z >= 0) out[ix.z] = v[2];
        if (ix.w >= 0) out[ix.w] = v[3];
    };
#pragma unroll
    for (int mm = 0; mm < MB; ++mm) {
#pragma unroll
        for (int n = 0; n < NBI; ++n) putc(WL.kw0 + mm * NBI + n, aW0[mm][n]);
#pragma unroll
        for (int l = 0; l < NL - 1; ++l)
#pragma unroll
            for (int n = 0; n < NBH; ++n) putc(WL.kwh + (l * MB + mm) * NBH + n, aWh[l][mm][n]);
        putc(WL.kwo + mm, aWo[mm]);
#pragma unroll
        for (int l = 0; l < NL; ++l) putc(WL.kb + l * MB + mm, aB[l][mm]);
    }
    putc(WL.kbo, aBo);                                // only wave 0's map has entries here
    if (wave == 0) {
        lacc = eh_wave_sum(lacc); syacc = eh_wave_sum(syacc); syyacc = eh_wave_sum(syyacc);
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t)
            if (t < net.T) cacc[t] = eh_wave_sum(cacc[t]);
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j)
            if (j < net.n_par) gacc[j] = eh_wave_sum(gacc[j]) * meta[EH_IMG_DPHI + j];
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j)
                if (j < net.n_par && pkind(j) == EH_PAR_GLOBAL) out[net.g_off + pidx(j)] = gacc[j];
            out[net.n_theta] = lacc;
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t)
                if (t < net.T) out[net.n_theta + 1 + t] = cacc[t];
            out[net.n_theta + 1 + net.T] = syacc;
            out[net.n_theta + 2 + net.T] = syyacc;
        }
    }
}
