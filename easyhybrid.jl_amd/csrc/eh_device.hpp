// Device-side description of a hybrid model and the fused training-step / evaluation kernel.
//
// Hot path replaced (reference = /root/reference, pure Julia):
//   Lux.Training.single_train_step!(AutoZygote(), loss_fn, batch, train_state)   src/training/epoch.jl:20-26
//     -> compute_loss (train branch)                                            src/losses/compute_loss.jl:20-35
//     -> (m::SingleNNHybridModel)(ds_k, ps, st)                                 src/models/GenericHybridModel.jl:370-431
//     -> loss_fn(yhat, y, mask, Val(:mse))                                      src/losses/loss_fn.jl:61-63
//     -> Zygote pullback of all of the above (hand-derived here)
//
// Kernel shape (gfx950): one wave owns a macro-tile of MT = 16*NT samples.  The MLP runs on the
// f32 MFMA v_mfma_f32_16x16x4_f32 with the SAMPLE index on the MFMA N dimension, so every layer's
// 16x16 output block (C/D layout: lane (c = lane&15, g = lane>>4), reg r  <->  feature 4g+r,
// sample c) is already the B operand of the next layer's MFMAs (k index permuted to 4g+s, the
// weight operand is read from LDS with the same permutation): activations never leave
// registers in the forward and dX-backward chains.  Weight gradients contract over samples, so
// they need the operands transposed (sample on K): each layer's activations and deltas are
// parked once in a wave-private LDS image [feature][sample] and read back with ds_read_b128.
// dW accumulators stay in registers across all macro-tiles of the wave; one partial per
// workgroup is written to a slab that eh_reduce_apply sums (deterministic, no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "easyhybrid_hip.h"   // public enums and EH_MAX_* limits

#define EH_EVAL_STATS 8   // per target: S=sum m(yh-y)^2, sum(y-c), sum(y-c)^2, n, sum(yh-c), sum(yh-c)^2, sum(yh-c)(y-c), sum|yh-y|

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EH_MODE_TRAIN = 0, EH_MODE_EVAL = 1 };

struct EhNet {
    int P, K, NL, G;                 // predictors, NN outputs (neural params), hidden layers, global params
    int width[EH_MAX_HIDDEN];        // hidden widths
    int w_off[EH_MAX_HIDDEN + 1];    // canonical flat-theta offset of layer l's weight (column-major (out,in))
    int b_off[EH_MAX_HIDDEN + 1];    // ... and bias
    int g_off;                       // offset of the raw global parameters
    int n_theta;                     // n_nn + G
    int act, scale_nn;
    int mech, n_par;
    int par_kind[EH_MAX_PARAMS], par_idx[EH_MAX_PARAMS];
    float par_lo[EH_MAX_PARAMS], par_hi[EH_MAX_PARAMS], par_def[EH_MAX_PARAMS];
    int F, forc_col[EH_MAX_FORC];    // number of forcing columns; column (0..F) feeding the mech model's f-th forcing
    int T, targ_out[EH_MAX_TARG];    // number of targets; mech output feeding target t
};

struct EhStepArgs {
    const float* recs;    // dataset, one record of C = P+F+T floats per sample: [predictors | forcings | targets (NaN = missing)]
    int C;
    const int* idx;       // optional gather indices (shuffled epoch): sample = idx[first + i]; nullptr = contiguous window
    long long first, count;
    const float* theta;   // canonical flat parameters
    float* slab;          // [gridDim.x][n_acc] per-workgroup partials
    int n_acc;            // train: n_theta + 1 + T ; eval: EH_EVAL_STATS*T
    const float* inv_n;   // train: per-target 1/n_t (device) or nullptr = deferred normalisation (weight 1)
    float* yhat;          // eval (optional): [T][yld] predictions for samples first..first+count
    float* pout;          // eval (optional): [n_par][yld] physical parameters per sample
    long long yld;
    float shift[EH_MAX_TARG];   // eval: metric shift c_t
};

// ------------------------------------------------------------------------------------------
// scalar math
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float eh_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// tanh as the degree-(4,4) rational in x^2 that Lux's Dense actually evaluates for Float32
// (LuxLib swaps tanh -> NNlib.tanh_fast); max relative error 1.8e-7 + rounding.
__device__ __forceinline__ float eh_tanh(float x) {
    const float x2 = x * x;
    const float n = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.587199e-8f, 2.2332108e-5f), 0.0035974074f), 0.1346604f), 1.0f);
    const float d = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 8.7767893e-7f, 0.0003453992f), 0.026262015f), 0.4679937f), 1.0f);
    const float r = x * (n * __builtin_amdgcn_rcpf(d));
    return x2 < 66.0f ? r : copysignf(1.0f, x);
}

__device__ __forceinline__ float eh_act(int act, float z) {
    switch (act) {
        case EH_ACT_TANH: return eh_tanh(z);
        case EH_ACT_SIGMOID: return eh_sigmoid(z);
        case EH_ACT_RELU: return fmaxf(z, 0.0f);
        case EH_ACT_SWISH: return z * eh_sigmoid(z);
        default: return z;
    }
}
// derivative from the stored value: h for tanh/sigmoid/relu/identity, z for swish
__device__ __forceinline__ float eh_dact(int act, float s) {
    switch (act) {
        case EH_ACT_TANH: return 1.0f - s * s;
        case EH_ACT_SIGMOID: return s * (1.0f - s);
        case EH_ACT_RELU: return s > 0.0f ? 1.0f : 0.0f;
        case EH_ACT_SWISH: { const float g = eh_sigmoid(s); return g * (1.0f + s * (1.0f - g)); }
        default: return 1.0f;
    }
}

__device__ __forceinline__ float eh_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float eh_row16_sum(float v) {   // over lanes sharing lane>>4
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// mechanistic models: y[o] and, for training, d par given d y.  One sample per lane.
// par[] in the registry's canonical order (see include/easyhybrid_hip.h).
// ------------------------------------------------------------------------------------------
template <bool GRAD>
__device__ __forceinline__ void eh_mech_eval(int mech, const float* par, const float* frc, float* y, const float* dy, float* dpar) {
    switch (mech) {
        case EH_MECH_RBQ10: {   // reco = rb * Q10^(0.1 (ta - 15))    test/test_split_data_train.jl:36-39
            const float e = 0.1f * (frc[0] - 15.0f);
            const float p = powf(par[1], e);
            y[0] = par[0] * p;
            if (GRAD) { dpar[0] = dy[0] * p; dpar[1] = dy[0] * y[0] * e / par[1]; }
        } break;
        case EH_MECH_EXPO: {    // Resp_obs = Resp0 * exp(k T)        projects/ExpoHybrid/ExpoHybridEstim.jl:83
            const float ex = expf(par[1] * frc[0]);
            y[0] = par[0] * ex;
            if (GRAD) { dpar[0] = dy[0] * ex; dpar[1] = dy[0] * y[0] * frc[0]; }
        } break;
        case EH_MECH_LINEAR: {  // obs = alpha x + beta               src/models/LinearHM.jl:65
            y[0] = par[0] * frc[0] + par[1];
            if (GRAD) { dpar[0] = dy[0] * frc[0]; dpar[1] = dy[0]; }
        } break;
        case EH_MECH_EXPO2POOL: {   // build-defined: R0a exp(ka T) + R0b exp(kb T)   (BASELINE.json config 3)
            const float ea = expf(par[1] * frc[0]), eb = expf(par[3] * frc[0]);
            y[0] = par[0] * ea + par[2] * eb;
            if (GRAD) {
                dpar[0] = dy[0] * ea; dpar[1] = dy[0] * par[0] * ea * frc[0];
                dpar[2] = dy[0] * eb; dpar[3] = dy[0] * par[2] * eb * frc[0];
            }
        } break;
        case EH_MECH_RS_COMPONENTS: {   // R_soil = sum_c Rb_c Q10_c^(0.1 (ta-15))   src/models/Rs_components.jl:45-55
            const float e = 0.1f * (frc[0] - 15.0f);
            float tot = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p = powf(par[3 + c], e);
                const float r = par[c] * p;
                tot += r;
                if (GRAD) { dpar[c] = dy[0] * p; dpar[3 + c] = dy[0] * r * e / par[3 + c]; }
            }
            y[0] = tot;
        } break;
        default: y[0] = 0.0f; break;
    }
}

// ------------------------------------------------------------------------------------------
// LDS geometry (floats).  Shared by host (size query) and device.
// ------------------------------------------------------------------------------------------
template <int NBI, int NBH, int NL, int NT>
struct EhGeom {
    static constexpr int MT = 16 * NT;          // samples per macro-tile
    static constexpr int SR = MT + 4;           // row stride of the [feature][sample] images (== 4 mod 8: conflict-free C-layout access)
    static constexpr int HP = 16 * NBH;         // padded hidden width
    static constexpr int IP = 16 * NBI;         // padded input width
    static constexpr int S0 = IP + 4;           // weight row strides
    static constexpr int SH = HP + 4;
    static constexpr int W0_OFF = 0;
    static constexpr int WH_OFF = W0_OFF + HP * S0;                  // NL-1 hidden->hidden matrices
    static constexpr int WO_OFF = WH_OFF + (NL - 1) * HP * SH;       // output layer, 16 padded rows
    static constexpr int B_OFF = WO_OFF + 16 * SH;                   // biases: NL * HP + 16
    static constexpr int WTOTAL = ((B_OFF + NL * HP + 16 + 3) / 4) * 4;
    // per-wave workspace
    static constexpr int XS_OFF = 0;
    static constexpr int HS_OFF = XS_OFF + IP * SR;                  // NL images of HP rows
    static constexpr int DZ_OFF = HS_OFF + NL * HP * SR;
    static constexpr int OS_OFF = DZ_OFF + HP * SR;                  // 16 rows
    static constexpr int WAVE_WS = OS_OFF + 16 * SR;
    static constexpr int TOTAL_FLOATS = WTOTAL + 4 * WAVE_WS;
};

// ------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------
template <int NBI, int NBH, int NL, int NT, int MODE>
__global__ __launch_bounds__(256, 1) void eh_step_kernel(const EhNet net, const EhStepArgs a) {
    using G = EhGeom<NBI, NBH, NL, NT>;
    constexpr int MT = G::MT, SR = G::SR, HP = G::HP, S0 = G::S0, SH = G::SH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    float* const ws = smem + G::WTOTAL + wave * G::WAVE_WS;
    float* const XS = ws + G::XS_OFF;
    float* const HS = ws + G::HS_OFF;
    float* const DZ = ws + G::DZ_OFF;
    float* const OS = ws + G::OS_OFF;
    const int act = net.act;

    // ---- stage the (zero-padded) weights and biases into LDS -----------------------------------
    {
        const int out0 = net.width[0];
        for (int e = tid; e < HP * G::IP; e += 256) {
            const int row = e % HP, col = e / HP;
            wl[G::W0_OFF + row * S0 + col] = (row < out0 && col < net.P) ? a.theta[net.w_off[0] + row + out0 * col] : 0.0f;
        }
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const int outl = net.width[l], inl = net.width[l - 1];
            for (int e = tid; e < HP * HP; e += 256) {
                const int row = e % HP, col = e / HP;
                wl[G::WH_OFF + (l - 1) * HP * SH + row * SH + col] = (row < outl && col < inl) ? a.theta[net.w_off[l] + row + outl * col] : 0.0f;
            }
        }
        {
            const int inl = net.width[NL - 1];
            for (int e = tid; e < 16 * HP; e += 256) {
                const int row = e % 16, col = e / 16;
                wl[G::WO_OFF + row * SH + col] = (row < net.K && col < inl) ? a.theta[net.w_off[NL] + row + net.K * col] : 0.0f;
            }
        }
        for (int e = tid; e < NL * HP + 16; e += 256) {
            const int l = e / HP, row = e % HP;
            float v = 0.0f;
            if (l < NL) { if (row < net.width[l]) v = a.theta[net.b_off[l] + row]; }
            else if (row < net.K) v = a.theta[net.b_off[NL] + row];
            wl[G::B_OFF + e] = v;
        }
        // zero the wave-private X image once (rows >= P must stay 0)
        for (int e = lane; e < G::IP * SR; e += 64) XS[e] = 0.0f;
    }
    // global physical parameters phi_g = lo + (hi-lo) sigmoid(raw)   (GenericHybridModel.jl:348-352)
    float phi[EH_MAX_PARAMS], dphi[EH_MAX_PARAMS];
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) {
        phi[j] = 0.0f; dphi[j] = 0.0f;
        if (j < net.n_par) {
            if (net.par_kind[j] == EH_PAR_GLOBAL) {
                const float s = eh_sigmoid(a.theta[net.g_off + net.par_idx[j]]);
                phi[j] = net.par_lo[j] + (net.par_hi[j] - net.par_lo[j]) * s;
                dphi[j] = (net.par_hi[j] - net.par_lo[j]) * s * (1.0f - s);
            } else if (net.par_kind[j] == EH_PAR_FIXED) {
                phi[j] = net.par_def[j];
            }
        }
    }
    __syncthreads();

    // ---- accumulators (registers, live across the tile loop) ------------------------------------
    f32x4 aW0[NBH][NBI], aWh[NL > 1 ? NL - 1 : 1][NBH][NBH], aWo[NBH];
    f32x4 aB[NL][NBH], aBo;
    float gacc[EH_MAX_PARAMS];
    float lacc = 0.0f;
    float cacc[EH_MAX_TARG];
    float est[EH_MAX_TARG][EH_EVAL_STATS];
    if (MODE == EH_MODE_TRAIN) {
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
#pragma unroll
            for (int n = 0; n < NBI; ++n) aW0[m][n] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                for (int n = 0; n < NBH; ++n) aWh[l][m][n] = f32x4{0, 0, 0, 0};
            aWo[m] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < NL; ++l) aB[l][m] = f32x4{0, 0, 0, 0};
        }
        aBo = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) gacc[j] = 0.0f;
    }
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        cacc[t] = 0.0f;
#pragma unroll
        for (int k = 0; k < EH_EVAL_STATS; ++k) est[t][k] = 0.0f;
    }

    const long long ntiles = (a.count + MT - 1) / MT;
    const int ksteps0 = (net.P + 3) / 4;   // k-steps of layer 0 that hold real features (natural k order 4s+g)

    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long n_loc = tile * MT + lane;                  // sample of this lane in the mech stage
        const bool live = (lane < MT) && (n_loc < a.count);

        // ---- 1. load the sample record: predictors -> [feature][sample] image, forcings / targets -> registers
        const long long n_glb = live ? (a.idx ? (long long)a.idx[a.first + n_loc] : a.first + n_loc) : 0;
        const float* const rec = a.recs + n_glb * a.C;
        if ((a.C & 3) == 0) {           // 16-byte records (RbQ10: exactly one dwordx4 per sample)
#pragma unroll
            for (int q = 0; q < (G::IP + 3) / 4; ++q) {
                if (4 * q < net.P) {
                    const f32x4 v = live ? *(const f32x4*)(rec + 4 * q) : f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e < net.P && lane < MT) XS[(4 * q + e) * SR + lane] = v[e];
                }
            }
        } else {
#pragma unroll 4
            for (int f = 0; f < net.P; ++f) {
                const float v = live ? rec[f] : 0.0f;
                if (lane < MT) XS[f * SR + lane] = v;
            }
        }
        float frc[EH_MAX_FORC], yobs[EH_MAX_TARG];
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) {
            frc[f] = 0.0f;
            if (net.forc_col[f] >= 0 && live) frc[f] = rec[net.P + net.forc_col[f]];
        }
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) {
            yobs[t] = __builtin_nanf("");
            if (t < net.T && live) yobs[t] = rec[net.P + net.F + t];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 2. forward, layer 0 : z = W0 x + b0 ------------------------------------------------
        f32x4 h[NBH][NT];
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
            const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + 16 * m + 4 * g];
#pragma unroll
            for (int t = 0; t < NT; ++t) h[m][t] = bias;
            for (int ks = 0; ks < ksteps0; ++ks) {
                const float av = wl[G::W0_OFF + (16 * m + c) * S0 + 4 * ks + g];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float bv = XS[(4 * ks + g) * SR + 16 * t + c];
                    h[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, h[m][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = h[m][t][r];
                    const float hv = eh_act(act, z);
                    h[m][t][r] = hv;
                    if (MODE == EH_MODE_TRAIN) HS[(16 * m + 4 * g + r) * SR + 16 * t + c] = (act == EH_ACT_SWISH) ? z : hv;
                }
        }
        // ---- 3. hidden layers -------------------------------------------------------------------
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            float* Hl = HS + l * HP * SR;
            f32x4 hn[NBH][NT];
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + l * HP + 16 * m + 4 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t) hn[m][t] = bias;
#pragma unroll
                for (int q = 0; q < NBH; ++q) {
                    const f32x4 a4 = *(const f32x4*)&W[(16 * m + c) * SH + 16 * q + 4 * g];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            hn[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], h[q][t][s], hn[m][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < NBH; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = hn[m][t][r];
                        const float hv = eh_act(act, z);
                        h[m][t][r] = hv;
                        if (MODE == EH_MODE_TRAIN) Hl[(16 * m + 4 * g + r) * SR + 16 * t + c] = (act == EH_ACT_SWISH) ? z : hv;
                    }
        }
        // ---- 4. output layer (K <= 16 rows, zero padded) ----------------------------------------
        {
            const float* W = wl + G::WO_OFF;
            f32x4 o[NT];
            const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + NL * HP + 4 * g];
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = bias;
#pragma unroll
            for (int q = 0; q < NBH; ++q) {
                const f32x4 a4 = *(const f32x4*)&W[c * SH + 16 * q + 4 * g];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], h[q][t][s], o[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) OS[(4 * g + r) * SR + 16 * t + c] = o[t][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 5. mechanistic model + masked loss, one sample per lane -----------------------------
        {
            float par[EH_MAX_PARAMS], sg[EH_MAX_PARAMS], dpar[EH_MAX_PARAMS];
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                par[j] = phi[j]; sg[j] = 1.0f; dpar[j] = 0.0f;
                if (j < net.n_par && net.par_kind[j] == EH_PAR_NEURAL) {
                    const float ov = (lane < MT) ? OS[net.par_idx[j] * SR + lane] : 0.0f;
                    if (net.scale_nn) {
                        const float s = eh_sigmoid(ov);
                        par[j] = net.par_lo[j] + (net.par_hi[j] - net.par_lo[j]) * s;
                        sg[j] = (net.par_hi[j] - net.par_lo[j]) * s * (1.0f - s);
                    } else {
                        par[j] = ov;
                    }
                }
            }
            float y[EH_MAX_TARG] = {0, 0, 0, 0}, dy[EH_MAX_TARG] = {0, 0, 0, 0}, yh[EH_MAX_TARG];
            eh_mech_eval<false>(net.mech, par, frc, y, dy, dpar);
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t) {
                yh[t] = 0.0f;
                if (t < net.T) {
#pragma unroll
                    for (int o = 0; o < EH_MAX_TARG; ++o)
                        if (net.targ_out[t] == o) yh[t] = y[o];
                    const bool valid = live && !__builtin_isnan(yobs[t]);
                    const float r = valid ? yh[t] - yobs[t] : 0.0f;
                    if (MODE == EH_MODE_TRAIN) {
                        const float w = a.inv_n ? a.inv_n[t] : 1.0f;
                        lacc += w * r * r;
                        cacc[t] += valid ? 1.0f : 0.0f;
#pragma unroll
                        for (int o = 0; o < EH_MAX_TARG; ++o)
                            if (net.targ_out[t] == o) dy[o] += 2.0f * w * r;
                    } else if (valid) {
                        const float cy = yobs[t] - a.shift[t], ch = yh[t] - a.shift[t];
                        est[t][0] += r * r; est[t][1] += cy; est[t][2] += cy * cy; est[t][3] += 1.0f;
                        est[t][4] += ch; est[t][5] += ch * ch; est[t][6] += ch * cy; est[t][7] += fabsf(r);
                    }
                }
            }
            if (MODE == EH_MODE_EVAL) {
                if (live) {
                    if (a.yhat)
                        for (int t = 0; t < net.T; ++t) a.yhat[(long long)t * a.yld + n_loc] = yh[t];
                    if (a.pout)
                        for (int j = 0; j < net.n_par; ++j) a.pout[(long long)j * a.yld + n_loc] = par[j];
                }
                continue;
            }
            eh_mech_eval<true>(net.mech, par, frc, y, dy, dpar);
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                if (j < net.n_par) {
                    if (net.par_kind[j] == EH_PAR_NEURAL) {
                        if (lane < MT) OS[net.par_idx[j] * SR + lane] = live ? dpar[j] * sg[j] : 0.0f;
                    } else if (net.par_kind[j] == EH_PAR_GLOBAL) {
                        gacc[j] += live ? dpar[j] : 0.0f;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 6. backward ------------------------------------------------------------------------
        f32x4 dz[NBH][NT];
        {
            // output layer: dWo += dO * H_last^T ; dbo += dO ; dH = Wo^T dO
            f32x4 dO[NT], aT[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[t][r] = OS[(4 * g + r) * SR + 16 * t + c];
                aT[t] = *(const f32x4*)&OS[c * SR + 16 * t + 4 * g];
                aBo += dO[t];
            }
            const float* Hl = HS + (NL - 1) * HP * SR;
#pragma unroll
            for (int n = 0; n < NBH; ++n)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 b4 = *(const f32x4*)&Hl[(16 * n + c) * SR + 16 * t + 4 * g];
                    if (act == EH_ACT_SWISH) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[s] = b4[s] * eh_sigmoid(b4[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) aWo[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aWo[n], 0, 0, 0);
                }
            const float* W = wl + G::WO_OFF;
            const int ksK = net.K < 4 ? net.K : 4;
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
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
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * m + 4 * g + r) * SR + 16 * t + c;
                        const float d = dh[t][r] * eh_dact(act, Hl[ad]);
                        dz[m][t][r] = d;
                        DZ[ad] = d;
                    }
                    aB[NL - 1][m] += dz[m][t];
                }
            }
        }
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // dW_l += dZ_l * H_{l-1}^T
            const float* Hp = HS + (l - 1) * HP * SR;
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                f32x4 aT[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) aT[t] = *(const f32x4*)&DZ[(16 * m + c) * SR + 16 * t + 4 * g];
#pragma unroll
                for (int n = 0; n < NBH; ++n)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 b4 = *(const f32x4*)&Hp[(16 * n + c) * SR + 16 * t + 4 * g];
                        if (act == EH_ACT_SWISH) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) b4[s] = b4[s] * eh_sigmoid(b4[s]);
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            aWh[l - 1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aWh[l - 1][m][n], 0, 0, 0);
                    }
            }
            // dH_{l-1} = W_l^T dZ_l ; dZ_{l-1} = dH ⊙ act'
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            f32x4 dn[NBH][NT];
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
#pragma unroll
                for (int t = 0; t < NT; ++t) dn[m][t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < NBH; ++q)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float av = W[(16 * q + 4 * g + s) * SH + 16 * m + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            dn[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dz[q][t][s], dn[m][t], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < NBH; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * m + 4 * g + r) * SR + 16 * t + c;
                        const float d = dn[m][t][r] * eh_dact(act, Hp[ad]);
                        dz[m][t][r] = d;
                        DZ[ad] = d;
                    }
                    aB[l - 1][m] += dz[m][t];
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // layer 0: dW0 += dZ_0 * X^T
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
            f32x4 aT[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) aT[t] = *(const f32x4*)&DZ[(16 * m + c) * SR + 16 * t + 4 * g];
#pragma unroll
            for (int n = 0; n < NBI; ++n)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 b4 = *(const f32x4*)&XS[(16 * n + c) * SR + 16 * t + 4 * g];
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW0[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aW0[m][n], 0, 0, 0);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- 7. workgroup reduction -> one partial per workgroup -------------------------------------
    __syncthreads();
    float* const RED = smem + G::WTOTAL;       // aliases the wave workspaces (dead now); n_acc floats
    for (int e = tid; e < a.n_acc; e += 256) RED[e] = 0.0f;
    __syncthreads();
    if (MODE == EH_MODE_EVAL) {
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t)
#pragma unroll
            for (int k = 0; k < EH_EVAL_STATS; ++k) {
                const float v = eh_wave_sum(est[t][k]);
                if (t < net.T && lane == 0) atomicAdd(&RED[t * EH_EVAL_STATS + k], v);
            }
    } else {
        // bias sums over the 16 samples held by the lanes of a row
#pragma unroll
        for (int m = 0; m < NBH; ++m)
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int r = 0; r < 4; ++r) aB[l][m][r] = eh_row16_sum(aB[l][m][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) aBo[r] = eh_row16_sum(aBo[r]);
        lacc = eh_wave_sum(lacc);
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) cacc[t] = eh_wave_sum(cacc[t]);
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) gacc[j] = eh_wave_sum(gacc[j]) * dphi[j];
        for (int w = 0; w < 4; ++w) {           // fixed wave order: deterministic sums
            if (wave == w) {
                const int out0 = net.width[0];
#pragma unroll
                for (int m = 0; m < NBH; ++m) {
#pragma unroll
                    for (int n = 0; n < NBI; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * m + 4 * g + r, col = 16 * n + c;
                            if (row < out0 && col < net.P) RED[net.w_off[0] + row + out0 * col] += aW0[m][n][r];
                        }
#pragma unroll
                    for (int l = 1; l < NL; ++l) {
                        const int outl = net.width[l], inl = net.width[l - 1];
#pragma unroll
                        for (int n = 0; n < NBH; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * m + 4 * g + r, col = 16 * n + c;
                                if (row < outl && col < inl) RED[net.w_off[l] + row + outl * col] += aWh[l - 1][m][n][r];
                            }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * g + r, col = 16 * m + c;
                        if (row < net.K && col < net.width[NL - 1]) RED[net.w_off[NL] + row + net.K * col] += aWo[m][r];
                    }
                    if (c == 0) {
#pragma unroll
                        for (int l = 0; l < NL; ++l)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * m + 4 * g + r;
                                if (row < net.width[l]) RED[net.b_off[l] + row] += aB[l][m][r];
                            }
                    }
                }
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < net.K) RED[net.b_off[NL] + 4 * g + r] += aBo[r];
                }
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < EH_MAX_PARAMS; ++j)
                        if (j < net.n_par && net.par_kind[j] == EH_PAR_GLOBAL) RED[net.g_off + net.par_idx[j]] += gacc[j];
                    RED[net.n_theta] += lacc;
#pragma unroll
                    for (int t = 0; t < EH_MAX_TARG; ++t)
                        if (t < net.T) RED[net.n_theta + 1 + t] += cacc[t];
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
    for (int e = tid; e < a.n_acc; e += 256) out[e] = RED[e];
}
