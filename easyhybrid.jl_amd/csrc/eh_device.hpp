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
// workgroup is written to a slab that eh_reduce_kernel sums (deterministic, no float atomics).
//
// Parameters reach the kernel as a zero-padded "image" in exactly the LDS layout (EhGeom), kept
// up to date by the optimiser kernel, so staging is a straight 16-byte copy; the image also
// carries the physical value and sigmoid slope of every global parameter (computed once per
// step instead of once per thread).
#pragma once
#ifndef __HIPCC_RTC__      // (this header is also compiled at run time, by hiprtc, for the recorded closures: eh_jit.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "easyhybrid_hip.h"   // public enums and EH_MAX_* limits

#define EH_BN_SELF_MAX 512   // input BatchNorm: minibatches up to this size get their statistics inside the per-wave step kernel (no eh_bn_stats_kernel launch)
#define EH_EVAL_STATS 8   // per target: S=sum m(yh-y)^2, sum(y-c), sum(y-c)^2, n, sum(yh-c), sum(yh-c)^2, sum(yh-c)(y-c), sum|yh-y|

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EH_MODE_TRAIN = 0, EH_MODE_EVAL = 1, EH_MODE_TRAIN_P2P = 2, EH_MODE_TRAIN_MULTI = 3 };   // TRAIN_P2P: fused-update step that exchanges its sums with the peer GPUs itself (EhP2P)

struct EhNet {
    int P, K, G, T, F;               // predictors, NN outputs (neural params), global params, targets, forcing columns
    int n_theta, g_off;              // n_nn + G ; offset of the raw global parameters in flat theta
    int scale_nn, mech, n_par;
    int loss;                        // eh_loss (training loss)
    int n_out;                       // outputs of the mechanistic model
    unsigned targ_out;               // 2 bits per target: which output it is compared with
    unsigned par_kind;               // 2 bits per canonical mech parameter: eh_param_kind
    unsigned par_idx;                // 4 bits per canonical mech parameter: NN output row / global index
    unsigned forc_col;               // 8 bits per canonical forcing: column among the F forcing columns (0xFF unused)
    unsigned loss_t;                 // 4 bits per target: its training loss (eh_loss; PerTarget, src/losses/compute_loss.jl:128-145).  One target: == loss.
};
__device__ __forceinline__ bool eh_target_mae(unsigned loss_t, int t) { return ((loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_MAE; }
__device__ __forceinline__ bool eh_target_prog(unsigned loss_t, int t) { return ((loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_PROGRAM; }
// losses whose d loss / d yhat_i needs batch statistics of yhat: pearson / kge / pbkge (loss_fn.jl:75-77,105-174), and rmse where its
// scale cannot be applied after the pass (multi-target models).  Two forward passes leave, per target, the coefficients of
// d loss / d yhat_i = k0 + k1 (yhat_i - centre) + k2 (y_i - shift) in the per-target table (eh_moment_coef_kernel).
__device__ __forceinline__ bool eh_target_two_pass(unsigned loss_t, int t, int T) {
    const unsigned k = (loss_t >> (4 * t)) & 15u;
    return (k >= (unsigned)EH_LOSS_PEARSONLOSS && k <= (unsigned)EH_LOSS_PBKGELOSS) || (k == (unsigned)EH_LOSS_RMSE && T > 1);
}
// EhStepArgs::inv_n is a table of EH_TT floats per target: [0] weight of the target's residual terms (1 / n_t, 1 / sum (y - ybar)^2)
// [1] centre of yhat (two-pass losses)  [4] k0  [5] k1  [6] k2  [7] the target's loss value (two-pass losses)
enum { EH_TT = 8 };
// Per-layer offsets / widths and the (lower, upper-lower) table travel in the parameter image
// (EhGeom::PHI_OFF block) instead of the kernarg: they are read from LDS where they are used,
// which keeps them out of the scalar register file during the tile loop.
enum { EH_IMG_PHI = 0, EH_IMG_DPHI = 8, EH_IMG_LO = 16, EH_IMG_SC = 24, EH_IMG_WOFF = 32, EH_IMG_BOFF = 37, EH_IMG_WIDTH = 42, EH_IMG_GPAR = 48, EH_IMG_BNM = 56, EH_IMG_BNR = 88, EH_IMG_META = 120 };

struct EhOpt {
    int rule;
    float lr, b1, b2, eps, wd;
};

// Optimisers.jl rules, fp32 op for op.  bt = {beta1^t, beta2^t} running products (Optimisers keeps
// them in Float32: 1 - Float32(0.999) != 1e-3, which matters at 1e-5 in the first steps).
// (no contraction into fused multiply-adds here: "op for op" is what Optimisers.jl's broadcasts do and what the oracle's NumPy does, and
//  left to the compiler the choice depended on the kernel the rule was inlined into -- Descent's `theta - eta * g` came out as one fma in
//  the multi-step kernel and as two operations in the single-step one: one ulp per step apart, which plain SGD on single-sample
//  minibatches amplifies to O(1) within a few hundred steps, tests/test_gpu_fuzz.py seed 296)
__device__ __forceinline__ void eh_opt_update(const EhOpt& o, float g, float bt1, float bt2, float& th, float& m, float& v) {
#pragma clang fp contract(off)
    if (o.rule == EH_OPT_ADAM || o.rule == EH_OPT_ADAMW) {
        m = o.b1 * m + (1.0f - o.b1) * g;
        v = o.b2 * v + (1.0f - o.b2) * (g * g);
        float upd = m / (1.0f - bt1) / (sqrtf(v / (1.0f - bt2)) + o.eps) * o.lr;
        if (o.rule == EH_OPT_ADAMW) upd += o.lr * o.wd * th;     // AdamW(couple = true)
        th -= upd;
    } else if (o.rule == EH_OPT_RMSPROP) {                       // RMSProp(eta, rho = b1, eps)
        v = o.b1 * v + (1.0f - o.b1) * (g * g);
        th -= g * (o.lr / (sqrtf(v) + o.eps));
    } else {                                                     // Descent(eta)
        th -= o.lr * g;
    }
}

// From the batch sums to the loss value and the factor that turns the accumulated un-normalised
// gradient (sum of 2 r dyhat, or of sign(r) dyhat for MAE) into the gradient of the loss
// (src/losses/loss_fn.jl:58-86).  S = sum r^2 (sum |r| for MAE), n = valid count, Sy / Syy = shifted
// target sums.  n == 0 -> scale 0, loss NaN: the batch is skipped (src/training/epoch.jl:17-19).
// agg: the factor `agg` puts on the data loss -- 1 for sum, 1 / (T (1 + #extra terms)) for mean (eh_set_option "agg";
// src/losses/compute_loss.jl:31-34,50-53) -- on the value and on the gradient alike.
__device__ __forceinline__ void eh_loss_finish(int kind, float S, float n, float Sy, float Syy, float& scale, float& loss, float agg = 1.0f) {
    if (!(n > 0.0f)) { scale = 0.0f; loss = __builtin_nanf(""); return; }
    if (kind == EH_LOSS_RMSE) {
        loss = sqrtf(S / n);
        scale = 1.0f / (2.0f * n * loss);
    } else if (kind == EH_LOSS_NSELOSS) {
        const float D = Syy - Sy * Sy / n;       // sum (y - mean y)^2
        loss = S / D;
        scale = 1.0f / D;
    } else {                                     // MSE, MAE
        scale = 1.0f / n;
        loss = S * scale;
    }
    if (agg != 1.0f) { scale *= agg; loss *= agg; }
}

#define EH_GSHARDS 8   // gradient accumulators are sharded 8 ways (blockIdx & 7) to spread the float atomics

// "fused update" mode: ONE kernel per training step.  The step kernel's prologue applies the
// optimiser update of the PREVIOUS step (every workgroup recomputes the new theta into its LDS
// image from theta/m/v and the accumulated gradient; workgroup 0 also stores it), and its epilogue
// adds this step's partial sums into the sharded accumulator with float atomics instead of
// writing a slab row.  Saves the reduce kernel and its launch boundary; the sums are no longer
// bitwise reproducible (atomic arrival order), so it is opt-in.
// Cross-GPU exchange without a collective call (data parallel, fused-update mode).  Every rank owns a receive
// buffer recv[3 slots][EH_GSHARDS ranks][n_acc] of 64-bit words {float value, 32-bit sequence number} in uncached
// device memory, exported over IPC and mapped by every peer; shard r of a slot is written by RANK r only.  The
// workgroups of a step add their partial sums into a local staging copy; the last workgroup to finish folds the
// staging shards and stores {value, seq} into shard `rank` of every peer's slot -- one 8-byte store per element
// carries its own arrival flag, so no fence and no separate flag round is needed (the "LL" idea of RCCL).  The
// next step's prologue reads the shards of all ranks, retrying the words whose sequence number is still old.
struct EhP2P {
    unsigned long long* peer_recv[EH_GSHARDS];   // receive buffers of every rank (own one included), mapped into this process
    float* stage;                        // local [3][EH_GSHARDS][n_acc] staging accumulators
    unsigned* counter;                   // [0] top-level ticket of the current launch (groups whose workgroups have all finished accumulating); [32 (1 + g)] group g's ticket
    int* err;                            // set when a wait ran into its deadline
    int world, rank;
    int mode;                            // 0: the last workgroup of a step to finish folds and publishes (two-level ticket election); 1: no election --
                                         // workgroup 0 of the NEXT kernel on the stream (the next step, or the flush) folds and publishes in its prologue
};

struct EhFused {
    float* gacc;           // nullptr = two-kernel (deterministic) mode; else [3][EH_GSHARDS][n_acc] rotating accumulators
    float* pset;           // [2][3][n_theta] parameter sets {theta, m, v}, then [2][2] running beta products
    const int* imap;       // canonical index -> image offset
    float* loss_slot;      // where the previous step's loss goes (nullable)
    int gslot;             // this step accumulates into gacc[gslot], applies gacc[(gslot+2)%3] when pending, clears gacc[(gslot+1)%3]
    int cur;               // reads parameter set `cur`, writes set `cur ^ 1`
    int sc_sel;            // same for the beta products
    int pending;
    EhOpt opt;
    float agg_a;           // factor of `agg` on the data loss (eh_loss_finish); multi-target steps carry it in their per-target weights
};

struct EhStepArgs {
    const float* recs;    // dataset, one record of C = P+F+T floats per sample: [predictors | forcings | targets (NaN = missing)]
    int C;
    const int* idx;       // optional gather indices (shuffled epoch): sample = idx[first + i]; nullptr = contiguous window
    long long first, count;
    const float* image;   // padded parameter image (EhGeom layout, IMG_FLOATS floats)
    float* slab;          // [gridDim.x][n_acc] per-workgroup partials
    int n_acc;            // train: n_theta + 1 + T ; eval: EH_EVAL_STATS*T
    const float* inv_n;   // train: the per-target table ([EH_TT] floats per target, see EH_TT) or nullptr = deferred normalisation (weight 1)
    float* yhat;          // eval (optional): [T][yld] predictions for samples first..first+count
    const int* rmap;      // train: canonical index -> (position | lanes<<24) among the parked accumulators (v2 / v3 workgroup reduction; row-split staging)
    float* pout;          // eval (optional): [n_par][yld] physical parameters per sample
    long long yld;
    float shift[EH_MAX_TARG];   // eval: metric shift c_t
    unsigned long long* stamps;   // diagnostic builds (-DEH_STAMPS) only: [16][2] (shader clock, 100 MHz wall clock)
    EhFused fz;
    // input BatchNorm, train mode: per-workgroup partial sums of the batch from eh_bn_stats_kernel
    const float* bn_part;   // [bn_nblk][64] (sum (x-c), sum (x-c)^2 per predictor); nullptr = use the image's statistics
    int bn_nblk;            // -1 (with bn_part == nullptr): the step kernel takes the statistics of the minibatch itself (count <= EH_BN_SELF_MAX)
    const float* bn_c;      // [32] the shift c the partial sums were taken around
    const float* bn_n;      // number of samples behind the sums when it is not `count` (cross-GPU statistics), else nullptr
    int bn_update;          // workgroup 0 also updates the running statistics (a real training step)
    float* bn_run;          // [2][32] running mean, running var
    float* image_out;       // global parameter image: its normalisation block follows the running statistics
    // (kept last: the single-GPU kernels never read them and the layout of everything above stays put)
    const EhP2P* p2p;     // EH_MODE_TRAIN_P2P only
    unsigned p2p_seq;     // sequence number this step publishes; its prologue waits for p2p_seq - 1
    EhP2P p2pv;           // EH_MODE_TRAIN_P2P: the same descriptor by value -- the kernels read it from the kernarg segment
                          // instead of chasing `p2p` (one dependent memory round trip less in front of the exchange)
    const unsigned* prog; // EH_MECH_PROGRAM kernels only: [0] length, [1] outputs, [2..4] output slots, [8..23] constants, [24..] code
    const unsigned* lprog; // recorded training losses where no kernel is compiled at run time (the layer-wise form): one program per target,
                           // EH_LPROG_WORDS apart, in the same layout -- slot 0 = yhat, slot 1 = y, [2] = the slot of l(yhat, y); interpreted
    // EH_MODE_TRAIN_MULTI only: ms_nsteps consecutive fused-update steps of ONE workgroup in one launch -- step k trains on the window
    // [first + k * ms_batch, ...) of at most ms_batch samples that ends at ms_end at the latest; the loss of step k goes to ms_loss[k]
    int ms_nsteps, ms_batch;
    int ms_keep;            // set by the multi-step kernel for its steps after the first: the parameter image is already in LDS
    int ms_direct;          // set by the multi-step kernel: ONE workgroup, so the step that produced the gradient applies the optimiser itself (see eh_ms_apply)
    long long ms_end;
    float* ms_loss;
    long long pf_first, pf_count;     // set by the multi-step kernel: the NEXT step's window (pf_count == 0: none) -- its records are fetched behind this step's compute
};
enum { EH_LPROG_WORDS = 24 + EH_MAX_PROG };

#define EH_BN_EPS 1e-5f
#define EH_BN_MOMENTUM 0.1f

// ------------------------------------------------------------------------------------------
// scalar math (hardware transcendental units; every path stays well inside the 1e-5 budget)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float eh_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// tanh as the degree-(4,4) rational in x^2 that Lux's Dense actually evaluates for Float32
// (LuxLib swaps tanh -> NNlib.tanh_fast: `ifelse(x^2 < 66, x * (n / d), sign(x))`); max relative error 1.8e-7 + rounding.
// The switch to sign(x) is part of the function: a saturated unit hands on EXACTLY +-1, so its 1 - h^2 is exactly 0 and nothing
// flows back through it.  (Rounds 1-3 clamped x to +-8.125 instead and evaluated the rational there: 1 - 1e-7, i.e. a derivative of
// 2e-7 where the reference has 0 -- on the headline inputs, raw sw_pot ~ 50 with most first-layer units saturated, Adam's
// normalisation turned those phantom gradients into full lr-sized steps: 0.038 off the fp64 trajectory after 20 steps where the
// plain-C fp32 port is 1e-6 off; found by anchoring the trajectory test in fp64, round 4.)
__device__ __forceinline__ float eh_tanh(float x) {
    const float x2 = x * x;
    const float n = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.587199e-8f, 2.2332108e-5f), 0.0035974074f), 0.1346604f), 1.0f);
    const float d = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 8.7767893e-7f, 0.0003453992f), 0.026262015f), 0.4679937f), 1.0f);
#ifdef EH_TEST_SKEW      // tests only (EH_JIT_DEFINES=EH_TEST_SKEW): a run-time compiled kernel that computes something else -- what the check
    const float r = 1.001f * x * (n * __builtin_amdgcn_rcpf(d));      // against the kernel built ahead of time (jit_verify, eh_api.hip) has to catch
#else
    const float r = x * (n * __builtin_amdgcn_rcpf(d));
#endif
    return x2 < 66.0f ? r : __builtin_copysignf(1.0f, x);
}

template <int ACT>
__device__ __forceinline__ float eh_act(float z) {
    if (ACT == EH_ACT_TANH) return eh_tanh(z);
    if (ACT == EH_ACT_SIGMOID) return eh_sigmoid(z);
    if (ACT == EH_ACT_RELU) return fmaxf(z, 0.0f);
    if (ACT == EH_ACT_SWISH) return z * eh_sigmoid(z);
    return z;
}
// Four values at once: tanh's polynomial work goes through packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32, two lanes of
// a register pair per instruction) -- the same IEEE operations in the same order as eh_tanh, at half the issue slots.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 eh_tanh2(f32x2 x) {
    const f32x2 x2 = x * x;
    auto fma2 = [](f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); };
    const f32x2 one = {1.0f, 1.0f};
    f32x2 n = fma2(x2, f32x2{1.587199e-8f, 1.587199e-8f}, 2.2332108e-5f);
    n = fma2(x2, n, 0.0035974074f); n = fma2(x2, n, 0.1346604f); n = __builtin_elementwise_fma(x2, n, one);
    f32x2 d = fma2(x2, f32x2{8.7767893e-7f, 8.7767893e-7f}, 0.0003453992f);
    d = fma2(x2, d, 0.026262015f); d = fma2(x2, d, 0.4679937f); d = __builtin_elementwise_fma(x2, d, one);
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#ifdef EH_TEST_SKEW
    const f32x2 t = f32x2{1.001f, 1.001f} * x * (n * r);
#else
    const f32x2 t = x * (n * r);
#endif
    return f32x2{x2[0] < 66.0f ? t[0] : __builtin_copysignf(1.0f, x[0]), x2[1] < 66.0f ? t[1] : __builtin_copysignf(1.0f, x[1])};      // sign(x) beyond x^2 = 66, as eh_tanh
}
template <int ACT>
__device__ __forceinline__ f32x4 eh_act4(f32x4 z) {
    if (ACT == EH_ACT_TANH) {
        const f32x2 a = eh_tanh2(f32x2{z[0], z[1]}), b = eh_tanh2(f32x2{z[2], z[3]});
        return f32x4{a[0], a[1], b[0], b[1]};
    }
    return f32x4{eh_act<ACT>(z[0]), eh_act<ACT>(z[1]), eh_act<ACT>(z[2]), eh_act<ACT>(z[3])};
}
// derivative from the stored value: h for tanh/sigmoid/relu/identity, z for swish
template <int ACT>
__device__ __forceinline__ float eh_dact(float s) {
    if (ACT == EH_ACT_TANH) return 1.0f - s * s;
    if (ACT == EH_ACT_SIGMOID) return s * (1.0f - s);
    if (ACT == EH_ACT_RELU) return s > 0.0f ? 1.0f : 0.0f;
    if (ACT == EH_ACT_SWISH) { const float g = eh_sigmoid(s); return g * (1.0f + s * (1.0f - g)); }
    return 1.0f;
}

// EH_ACT_PER_NET (MultiNNHybridModel with activation::NamedTuple, GenericHybridModel.jl:168-176): the nets sit side by side in
// the block-diagonal MLP, so the activation is a function of (layer, row).  Such kernels exist only compiled at run time: the
// generated header defines eh_row_act(layer, row) -> eh_activation from the descriptor's net widths.  The images hold the
// pre-activation z for every row (like swish), so value and derivative are both taken from z.
#ifdef EH_JIT_ROWACT
#include "eh_jit_rowact.inc"
#else
__device__ __forceinline__ int eh_row_act(int, int) { return EH_ACT_IDENTITY; }
#endif
template <int ACT> struct EhStoresZ { static constexpr bool value = ACT == EH_ACT_SWISH || ACT == EH_ACT_PER_NET; };
// A value that goes to LDS straight from an MFMA accumulator (the pre-activations the swish / per-net kernels keep) may sit in an AGPR;
// two such stores merged into one ds_write2_b32 whose other data operand is a VGPR is an instruction some ROCm compilers then refuse
// ("Illegal instruction detected: both data operands should be VGPR or AGPR" -- the hiprtc PyTorch bundles, on the per-net-activation
// kernels, found by the fuzz in round 3; the system compiler of ROCm 7.2 builds the same source).  Pinning the value to a VGPR costs
// at most the v_accvgpr_read the store would need on older parts anyway.
__device__ __forceinline__ float eh_vgpr(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float eh_act_id(int id, float z) {
    switch (id) {
        case EH_ACT_TANH: return eh_tanh(z);
        case EH_ACT_SIGMOID: return eh_sigmoid(z);
        case EH_ACT_RELU: return fmaxf(z, 0.0f);
        case EH_ACT_SWISH: return z * eh_sigmoid(z);
        default: return z;
    }
}
__device__ __forceinline__ float eh_dact_z_id(int id, float z) {       // act'(z) from the pre-activation
    switch (id) {
        case EH_ACT_TANH: { const float t = eh_tanh(z); return 1.0f - t * t; }
        case EH_ACT_SIGMOID: { const float g = eh_sigmoid(z); return g * (1.0f - g); }
        case EH_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
        case EH_ACT_SWISH: { const float g = eh_sigmoid(z); return g * (1.0f + z * (1.0f - g)); }
        default: return 1.0f;
    }
}
// the three things the kernels do with an activation, by (layer, row): z -> h; stored value -> h; stored value -> act'
template <int ACT>
__device__ __forceinline__ f32x4 eh_act4_rows(f32x4 z, int l, int row0) {      // rows row0 .. row0 + 3
    if constexpr (ACT == EH_ACT_PER_NET)
        return f32x4{eh_act_id(eh_row_act(l, row0), z[0]), eh_act_id(eh_row_act(l, row0 + 1), z[1]),
                     eh_act_id(eh_row_act(l, row0 + 2), z[2]), eh_act_id(eh_row_act(l, row0 + 3), z[3])};
    else return eh_act4<ACT>(z);
}
template <int ACT>
__device__ __forceinline__ float eh_hval(float s, int l, int row) {
    if constexpr (ACT == EH_ACT_PER_NET) return eh_act_id(eh_row_act(l, row), s);
    else if constexpr (ACT == EH_ACT_SWISH) return s * eh_sigmoid(s);
    else return s;
}
template <int ACT>
__device__ __forceinline__ float eh_dact_row(float s, int l, int row) {
    if constexpr (ACT == EH_ACT_PER_NET) return eh_dact_z_id(eh_row_act(l, row), s);
    else return eh_dact<ACT>(s);
}

// cross-lane sums on the DPP network (no LDS round trips): rotate-and-add inside each row of 16
// lanes, then combine the four rows through scalar lane reads.
template <int CTRL>
__device__ __forceinline__ float eh_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float eh_row16_sum(float v) {   // every lane gets the sum over the 16 lanes sharing lane>>4
    v += eh_dpp<0x121>(v);   // row_ror:1
    v += eh_dpp<0x122>(v);   // row_ror:2
    v += eh_dpp<0x124>(v);   // row_ror:4
    v += eh_dpp<0x128>(v);   // row_ror:8
    return v;
}
__device__ __forceinline__ float eh_wave_sum(float v) {
    const int r = __builtin_bit_cast(int, eh_row16_sum(v));
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48)));
}
// The eight shards of an accumulator element are folded in ONE order wherever they are folded (step prologue, flush kernel, the
// workgroup that publishes a rank's sums to its peers): a rank's own value and the value its peers receive must be the same bits.
// The tree is what three butterfly steps over eight adjacent lanes compute (eh_fold8_lanes).
__device__ __forceinline__ float eh_fold8(float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7) {
    return ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}
__device__ __forceinline__ float eh_fold8_lanes(float v) {   // lanes 8k + sh hold shard sh: every lane of the group gets eh_fold8 of the group
    v += eh_dpp<0xB1>(v);    // quad_perm:[1,0,3,2]
    v += eh_dpp<0x4E>(v);    // quad_perm:[2,3,0,1]
    v += eh_dpp<0x141>(v);   // row_half_mirror (the quads hold their sums in every lane by now)
    return v;
}
__device__ __forceinline__ float eh_pow(float b, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(b)); }   // b > 0

// ------------------------------------------------------------------------------------------
// mechanistic models: output y and its partial derivatives dydp[j] = dy/dpar_j.  One sample per
// lane.  par[] in the registry's canonical order (see include/easyhybrid_hip.h).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float eh_mech_eval(int mech, const float* par, const float* frc, float* dydp) {
    // the partials live in named scalars inside the switch: writing dydp[] from its arms makes the
    // compiler keep the array in scratch memory (and every scratch read drains vmcnt, i.e. waits for
    // the record prefetch)
    float y = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f, d4 = 0.0f, d5 = 0.0f;
    const float p0 = par[0], p1 = par[1], f0 = frc[0];
    switch (mech) {
        case EH_MECH_RBQ10: {   // reco = rb * Q10^(0.1 (ta - 15))    test/test_split_data_train.jl:36-39
            const float e = 0.1f * (f0 - 15.0f);
            const float p = eh_pow(p1, e);
            y = p0 * p;
            d0 = p; d1 = y * e * __builtin_amdgcn_rcpf(p1);
        } break;
        case EH_MECH_EXPO: {    // Resp_obs = Resp0 * exp(k T)        projects/ExpoHybrid/ExpoHybridEstim.jl:83
            const float ex = __expf(p1 * f0);
            y = p0 * ex;
            d0 = ex; d1 = y * f0;
        } break;
        case EH_MECH_LINEAR: {  // obs = alpha x + beta               src/models/LinearHM.jl:65
            y = p0 * f0 + p1;
            d0 = f0; d1 = 1.0f;
        } break;
        case EH_MECH_EXPO2POOL: {   // build-defined: R0a exp(ka T) + R0b exp(kb T)   (BASELINE.json config 3)
            const float p2 = par[2], p3 = par[3];
            const float ea = __expf(p1 * f0), eb = __expf(p3 * f0);
            y = p0 * ea + p2 * eb;
            d0 = ea; d1 = p0 * ea * f0;
            d2 = eb; d3 = p2 * eb * f0;
        } break;
        case EH_MECH_RS_COMPONENTS: {   // R_soil = sum_c Rb_c Q10_c^(0.1 (ta-15))   src/models/Rs_components.jl:45-55
            const float p2 = par[2], p3 = par[3], p4 = par[4], p5 = par[5];
            const float e = 0.1f * (f0 - 15.0f);
            const float q0 = eh_pow(p3, e), q1 = eh_pow(p4, e), q2 = eh_pow(p5, e);
            const float r0 = p0 * q0, r1 = p1 * q1, r2 = p2 * q2;
            y = (r0 + r1) + r2;
            d0 = q0; d1 = q1; d2 = q2;
            d3 = r0 * e * __builtin_amdgcn_rcpf(p3); d4 = r1 * e * __builtin_amdgcn_rcpf(p4); d5 = r2 * e * __builtin_amdgcn_rcpf(p5);
        } break;
        case EH_MECH_RS_COMPONENTS3F: {   // build-defined (BASELINE.json config 5): R_het + sw_in R_root + vpd R_myc, pools as Rs_components.jl:45-55
            const float p2 = par[2], p3 = par[3], p4 = par[4], p5 = par[5];
            const float e = 0.1f * (f0 - 15.0f);
            const float q0 = eh_pow(p3, e), q1 = frc[1] * eh_pow(p4, e), q2 = frc[2] * eh_pow(p5, e);
            const float r0 = p0 * q0, r1 = p1 * q1, r2 = p2 * q2;
            y = (r0 + r1) + r2;
            d0 = q0; d1 = q1; d2 = q2;
            d3 = r0 * e * __builtin_amdgcn_rcpf(p3); d4 = r1 * e * __builtin_amdgcn_rcpf(p4); d5 = r2 * e * __builtin_amdgcn_rcpf(p5);
        } break;
        case EH_MECH_FLUXPART: {    // output 0: NEE = RECO - GPP   src/models/FluxPartModel_Q10_Lux.jl:66-74
            const float p2 = par[2], f1 = frc[1];
            const float e = 0.1f * (f1 - 15.0f);
            const float p = eh_pow(p2, e);
            const float gq = f0 * (1.0f / 12.011f);
            const float reco = p1 * p;
            y = reco - gq * p0;
            d0 = -gq; d1 = p; d2 = reco * e * __builtin_amdgcn_rcpf(p2);
        } break;
        default: break;
    }
    dydp[0] = d0; dydp[1] = d1; dydp[2] = d2; dydp[3] = d3; dydp[4] = d4; dydp[5] = d5;
    return y;
}

// EH_MECH_PROGRAM: a user closure recorded as a straight-line program (include/easyhybrid_hip.h, eh_prog_op).  One
// sample per lane; the value slots are a per-lane array indexed by the (wave-uniform) operand fields, i.e. they live in
// scratch memory -- these kernels are their own instantiations (FAST bit 2), the registry models never carry it.
// The host has checked every operand slot against the instruction's position (eh_create), so no index leaves val[].
#define EH_PROG_HDR 24
__device__ __forceinline__ void eh_prog_forward(const unsigned* __restrict__ prog, const float* par, const float* frc, float* val) {
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) val[EH_PROG_SLOT_PAR + j] = par[j];
#pragma unroll
    for (int f = 0; f < EH_MAX_FORC; ++f) val[EH_PROG_SLOT_FORC + f] = frc[f];
#pragma unroll
    for (int k = 0; k < EH_MAX_PROG_CONST; ++k) val[EH_PROG_SLOT_CONST + k] = __uint_as_float(prog[8 + k]);
    const int n = (int)prog[0];
    for (int i = 0; i < n; ++i) {
        const unsigned w = prog[EH_PROG_HDR + i];
        const float x = val[(w >> 8) & 255u], y = val[(w >> 16) & 255u], z = val[w >> 24];
        float r;
        switch (w & 255u) {
            case EH_OP_ADD: r = x + y; break;
            case EH_OP_SUB: r = x - y; break;
            case EH_OP_MUL: r = x * y; break;
            case EH_OP_DIV: r = x / y; break;
            case EH_OP_NEG: r = -x; break;
            case EH_OP_EXP: r = __expf(x); break;
            case EH_OP_LOG: r = __logf(x); break;
            case EH_OP_POW: r = eh_pow(x, y); break;
            case EH_OP_SQRT: r = sqrtf(x); break;
            case EH_OP_TANH: r = eh_tanh(x); break;
            case EH_OP_SIGMOID: r = eh_sigmoid(x); break;
            case EH_OP_MAX: r = fmaxf(x, y); break;
            case EH_OP_MIN: r = fminf(x, y); break;
            case EH_OP_ABS: r = fabsf(x); break;
            case EH_OP_SIN: r = sinf(x); break;
            case EH_OP_COS: r = cosf(x); break;
            case EH_OP_SELECT: r = x > 0.0f ? y : z; break;
            case EH_OP_GT: r = x > y ? 1.0f : 0.0f; break;
            default: r = 0.0f; break;
        }
        val[EH_PROG_SLOT_INSTR + i] = r;
    }
}
// reverse sweep: adj[] comes in holding the output seeds (zero elsewhere); on return adj[j], j < 8, is d loss / d parameter j
__device__ __forceinline__ void eh_prog_reverse(const unsigned* __restrict__ prog, const float* val, float* adj) {
    const int n = (int)prog[0];
    for (int i = n - 1; i >= 0; --i) {
        const unsigned w = prog[EH_PROG_HDR + i];
        const unsigned ia = (w >> 8) & 255u, ib = (w >> 16) & 255u, ic = w >> 24;
        const float x = val[ia], y = val[ib], r = val[EH_PROG_SLOT_INSTR + i], gr = adj[EH_PROG_SLOT_INSTR + i];
        float ga = 0.0f, gb = 0.0f, gc = 0.0f;
        switch (w & 255u) {
            case EH_OP_ADD: ga = gr; gb = gr; break;
            case EH_OP_SUB: ga = gr; gb = -gr; break;
            case EH_OP_MUL: ga = gr * y; gb = gr * x; break;
            case EH_OP_DIV: ga = gr / y; gb = -(gr / y) * r; break;
            case EH_OP_NEG: ga = -gr; break;
            case EH_OP_EXP: ga = gr * r; break;
            case EH_OP_LOG: ga = gr / x; break;
            case EH_OP_POW: ga = gr * y * r / x; gb = gr * r * __logf(x); break;
            case EH_OP_SQRT: ga = gr * 0.5f / r; break;
            case EH_OP_TANH: ga = gr * (1.0f - r * r); break;
            case EH_OP_SIGMOID: ga = gr * r * (1.0f - r); break;
            case EH_OP_MAX: ga = x >= y ? gr : 0.0f; gb = x >= y ? 0.0f : gr; break;
            case EH_OP_MIN: ga = x <= y ? gr : 0.0f; gb = x <= y ? 0.0f : gr; break;
            case EH_OP_ABS: ga = x > 0.0f ? gr : (x < 0.0f ? -gr : 0.0f); break;
            case EH_OP_SIN: ga = gr * cosf(x); break;
            case EH_OP_COS: ga = -gr * sinf(x); break;
            case EH_OP_SELECT: gb = x > 0.0f ? gr : 0.0f; gc = x > 0.0f ? 0.0f : gr; break;
            default: break;
        }
        // sequential read-modify-writes: two operands may name the same slot (x * x)
        adj[ia] += ga;
        adj[ib] += gb;
        adj[ic] += gc;
    }
}

// The same program as generated straight-line code: eh_jit.hip writes eh_jit_fwd / eh_jit_rev (every slot a named value, so
// the tape lives in registers) and compiles this header with hiprtc when a model with a recorded closure is created.
#ifdef EH_JIT_MECH
#include "eh_jit_mech.inc"
#endif
// a recorded custom training loss (eh_set_loss_program): eh_jit_loss(yhat, y, dl) -> l, dl = d l / d yhat.  Run-time builds only.
#ifdef EH_JIT_LOSS
#include "eh_jit_loss.inc"
#endif

// outputs 1.. of the multi-output models and their Jacobian rows (only FLUXPART: GPP, RECO)
__device__ __forceinline__ void eh_mech_extra(int mech, const float* par, const float* frc, float* yx, float (*Jx)[3]) {
    if (mech == EH_MECH_FLUXPART) {
        const float e = 0.1f * (frc[1] - 15.0f);
        const float p = eh_pow(par[2], e);
        const float gq = frc[0] * (1.0f / 12.011f);
        yx[0] = gq * par[0];                 // GPP
        Jx[0][0] = gq; Jx[0][1] = 0.0f; Jx[0][2] = 0.0f;
        yx[1] = par[1] * p;                  // RECO
        Jx[1][0] = 0.0f; Jx[1][1] = p; Jx[1][2] = yx[1] * e * __builtin_amdgcn_rcpf(par[2]);
    }
}

// ------------------------------------------------------------------------------------------
// cross-GPU exchange helpers (EhP2P).  Every wait has a deadline on the 100 MHz wall clock, so a
// missing peer turns into an error flag, never into a kernel that does not finish; once a wait has
// timed out the job is lost and later waits return at once (the total delay stays bounded).
// ------------------------------------------------------------------------------------------
#define EH_P2P_DEADLINE_TICKS 200000000ull      // 2 s between two steps of a running job
__device__ __forceinline__ unsigned long long eh_ll_pack(float v, unsigned seq) { return ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v); }
// N words of one reader: eh_ll_issue requests them all (no waiting -- other loads can be queued behind them),
// eh_ll_finish examines them and keeps re-reading until every word carries `seq`; words that never arrive
// read as 0.  addr(i) == nullptr -> 0 without a load.
template <int N, class A>
__device__ __forceinline__ void eh_ll_issue(A addr, unsigned seq, unsigned long long (&w)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned long long* q = addr(i);
        w[i] = q ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ((unsigned long long)seq << 32);
    }
}
template <int N, class A>
__device__ __forceinline__ void eh_ll_finish(const EhP2P* P, A addr, unsigned seq, unsigned long long (&w)[N], float (&out)[N],
                                             unsigned long long deadline = EH_P2P_DEADLINE_TICKS) {
    unsigned long long t0 = 0;
    bool timing = false;
    while (true) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && ((unsigned)(w[i] >> 32) == seq);
        if (all) break;
        if (__hip_atomic_load(P->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (!timing) { t0 = wall_clock64(); timing = true; }
        else if (wall_clock64() - t0 > deadline) { __hip_atomic_store(P->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(2);
        eh_ll_issue(addr, seq, w);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = ((unsigned)(w[i] >> 32) == seq) ? __uint_as_float((unsigned)w[i]) : 0.0f;
}
// one workgroup: this rank's staging shards of `slot` folded and stored, every element with its arrival stamp, into shard `rank` of every
// rank's receive buffer (its own included).  Whoever calls it knows that every add into the shards has landed: the last workgroup of the
// step by ticket (eh_p2p_publish, mode 0), or workgroup 0 of the next kernel on the stream (mode 1: the kernel boundary says so).
// (mode 1 stores to the PEERS only: every workgroup of this rank takes the rank's own sums straight from the staging shards, as the
//  single-GPU step takes them from its accumulators -- nothing of the rank's own waits for a publication)
__device__ __forceinline__ void eh_p2p_fold_store(const EhP2P* P, int slot, unsigned seq, int n_acc, int tid, int nthr, bool peers_only = false) {
    if (peers_only && P->world == 1) return;
    const float* st = P->stage + (long long)slot * EH_GSHARDS * n_acc;
    for (int i = tid; i < n_acc; i += nthr) {
        float sv[EH_GSHARDS];
#pragma unroll
        for (int sh = 0; sh < EH_GSHARDS; ++sh) sv[sh] = __hip_atomic_load(&st[sh * n_acc + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        static_assert(EH_GSHARDS == 8, "eh_fold8");
        const unsigned long long w = eh_ll_pack(eh_fold8(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]), seq);
        for (int r = 0; r < P->world; ++r)
            if (!peers_only || r != P->rank)
                __hip_atomic_store(&P->peer_recv[r][((long long)slot * EH_GSHARDS + P->rank) * n_acc + i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
enum { EH_P2P_GROUPS = 16 };     // first-level ticket counters at counter[32 (1 + g)], g < 16 (one 128-byte line each)
// called by every thread of every workgroup once its sums are in the staging shards
__device__ __forceinline__ void eh_p2p_publish(const EhP2P* P, int slot, unsigned seq, int n_acc, int tid, int nthr) {
    __shared__ unsigned eh_p2p_last;
    // this thread's atomic adds are acknowledged by the memory side (they never live in an L2): no cache write-back,
    // which a __threadfence() would add 256 times per launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        // Two-level ticket: 256 returning atomics on ONE address serialise at the memory side (~13 ns each: the workgroups of a
        // launch retire within a microsecond of each other, so the last one used to wait out most of the queue).  Sixteen group
        // counters on lines of their own take at most 16 tickets each; a group's last workgroup takes one of 16 top-level tickets.
        const unsigned ng = gridDim.x < EH_P2P_GROUPS ? gridDim.x : EH_P2P_GROUPS, gi = blockIdx.x % EH_P2P_GROUPS;
        const unsigned gsz = (gridDim.x - gi + EH_P2P_GROUPS - 1) / EH_P2P_GROUPS;
        unsigned* const gc = P->counter + 32 * (1 + gi);
        unsigned last = 0u;
        if (atomicAdd(gc, 1u) == gsz - 1) {
            __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (atomicAdd(P->counter, 1u) == ng - 1) ? 1u : 0u;
        }
        eh_p2p_last = last;
    }
    __syncthreads();
    if (!eh_p2p_last) return;
    eh_p2p_fold_store(P, slot, seq, n_acc, tid, nthr);
    if (tid == 0) __hip_atomic_store(P->counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// LDS / image geometry (floats).  Shared by host (image packing, size query) and device.
// ------------------------------------------------------------------------------------------
// Order of the per-lane f32x4 gradient accumulators as the step kernel parks them in LDS for the
// end-of-kernel reduction ("v2" layout: region[k][lane][r], then 16 scalars).  Shared with the host,
// which builds the canonical-index -> region-position map (rmap) from it.
struct EhAccLayout { int kw0, nw0, kwh, nwh, kwo, kb, kbo, na, rw; };
__host__ __device__ constexpr EhAccLayout eh_acc_layout(int nbi, int nbh, int nl, int fast) {
    EhAccLayout L{};
    L.kw0 = 0; L.nw0 = (fast & 2) ? nbh * 4 : nbh * nbi;
    L.kwh = L.kw0 + L.nw0; L.nwh = (nl - 1) * nbh * nbh;
    L.kwo = L.kwh + L.nwh;
    L.kb = L.kwo + nbh;
    L.kbo = L.kb + nl * nbh;
    L.na = L.kbo + ((fast & 1) ? 0 : 1);
    L.rw = L.na * 256 + 16;      // tail scalars: [0..7] global-param sums, [8] loss, [9..12] counts, [13] output bias (K1)
    return L;
}

template <int NBI, int NBH, int NL, int NT, int NW>
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
    static constexpr int PHI_OFF = B_OFF + NL * HP + 16;             // EH_IMG_* block: phi[8], dphi[8], lo[8], hi-lo[8], int w_off[5], b_off[5], width[4], gpar[8], input-normalisation mean[32], 1/std[32]
    static constexpr int IMG_FLOATS = PHI_OFF + EH_IMG_META;         // multiple of 4
    // per-wave workspace
    static constexpr int XS_OFF = 0;
    static constexpr int HS_OFF = XS_OFF + IP * SR;                  // NL images of HP rows; in the backward pass layer l's delta replaces its activations in place
    static constexpr int OS_OFF = HS_OFF + NL * HP * SR;             // 16 rows
    static constexpr int WS_MIN = OS_OFF + 16 * SR;
    // the end-of-kernel reduction parks every lane's raw accumulators in the wave's workspace ("v2", EhAccLayout::rw floats) where
    // that fits within one more image of HP rows -- the room a separate delta image used to take
    static constexpr int RW0 = eh_acc_layout(NBI, NBH, NL, 0).rw;
    static constexpr int WAVE_WS = (RW0 > WS_MIN && RW0 <= WS_MIN + HP * SR) ? RW0 : WS_MIN;
    static constexpr int TOTAL_FLOATS = IMG_FLOATS + NW * WAVE_WS;
    // forward / evaluation passes (MODE == EVAL) keep the activations in registers and write no hidden image: their work space is the X
    // image and the 16 output rows -- a third to a half of the training kernel's LDS, so twice the workgroups fit a CU (the passes are
    // bound by the latency of their vector-ALU chains: 60 registers, eight waves per SIMD fit the register file)
    static constexpr int OS_OFF_EVAL = XS_OFF + IP * SR;
    static constexpr int WS_EVAL = OS_OFF_EVAL + 16 * SR;
    static constexpr int TOTAL_FLOATS_EVAL = IMG_FLOATS + NW * WS_EVAL;
    static constexpr int WS_EVAL_K1 = IP * SR;                      // single NN output (FAST bit 0): the output stays in a register, no output rows either
    static constexpr int TOTAL_FLOATS_EVAL_K1 = IMG_FLOATS + NW * WS_EVAL_K1;
};

#ifdef EH_STAMPS
#define EH_STAMP_RAW(i)                                                                  \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                           \
            a.stamps[2 * (i)] = __builtin_readcyclecounter();                            \
            a.stamps[2 * (i) + 1] = wall_clock64();                                      \
        }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define EH_STAMP_RAW(i)
#endif
// EH_STAMPS_PROLOGUE (with EH_STAMPS): slots 2..7 and 11..15 stamp the parts of the step's prologue instead of the tile loop's segments
// (tools/stamps_p2p.py)
#if defined(EH_STAMPS) && defined(EH_STAMPS_PROLOGUE)
#define EH_STAMP(i) do { if ((i) <= 1 || ((i) >= 8 && (i) <= 10)) EH_STAMP_RAW(i); } while (0)
#define EH_STAMP_PRO(i) EH_STAMP_RAW(i)
#else
#define EH_STAMP(i) EH_STAMP_RAW(i)
#define EH_STAMP_PRO(i)
#endif
#ifdef EH_STAMPS_FINE
#define EH_STAMP_FINE(i) EH_STAMP(i)
#else
#define EH_STAMP_FINE(i)
#endif

// Nothing moves across: keeps a block of LDS requests in front of the MFMA chain it is meant to hide behind.
#ifdef EH_NO_SCHED_FENCE
#define EH_SCHED_FENCE()
#else
#define EH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef EH_KEEPH_MAX
#define EH_KEEPH_MAX 64
#endif
// Orders one wave's LDS traffic: what its lanes wrote before is visible to what any of its lanes reads after (the hardware
// runs a wave's LDS operations in order; the fence is for the compiler, and it is acquire + release: a release-only fence
// would leave later READS free to move up across it).
#ifndef EH_SYNC_ORDER
#define EH_SYNC_ORDER __ATOMIC_ACQ_REL
#endif
#define EH_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(EH_SYNC_ORDER, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

// ------------------------------------------------------------------------------------------
// the fused kernel
//   FAST bit 0 (K1): exactly one NN output -> output layer, its weight gradient and the first
//                    backward step run on the vector ALU and the hand-over to the one-sample-per-lane
//                    mechanistic stage needs no LDS round trip
//   FAST bit 1 (PS): P <= 4 predictors -> the first layer's weight gradient runs on the vector ALU
// ------------------------------------------------------------------------------------------
// Multi-step launch (EH_MODE_TRAIN_MULTI: ONE workgroup): the step that produced the gradient applies the optimiser itself.  `gsum` holds
// the step's sums (LDS, one writer per element, written just before the call); thread i owns parameter i: theta / m / v of the CURRENT set
// (the one step 0's prologue wrote, z.cur ^ 1) are updated in place in LDS, the new value goes straight into the LDS parameter image, the
// loss into the step's own slot, the sums back to zero.  Op for op the arithmetic of the deferred form (the next step's prologue: 8 shards
// of which 7 are zero, three rotating accumulator sets, LDS atomics, a second parameter set), without what exists to hand sums between
// workgroups.
typedef __attribute__((address_space(3))) float eh_lds_f;
typedef __attribute__((address_space(3))) int eh_lds_i;
template <class G, class NET>
__device__ __forceinline__ void eh_ms_apply(const NET& net, const EhStepArgs& a, float* gsum_, float* wl_, int tid, int nthr) {
    const EhFused& z = a.fz;
    const int nth = net.n_theta;
    // every pointer here is LDS (the multi-step kernel redirected the step's state there): as generic pointers each access would be a FLAT
    // load / store -- hundreds of cycles apiece, five of them in a dependent chain (measured: 3 k cycles of a 11 k-cycle step)
    eh_lds_f* const gsum = (eh_lds_f*)gsum_;
    eh_lds_f* const wl = (eh_lds_f*)wl_;
    eh_lds_f* const P = (eh_lds_f*)(z.pset + (z.cur ^ 1) * 3 * nth);
    eh_lds_f* const sc = (eh_lds_f*)(z.pset + 6 * nth + 2 * (z.sc_sel ^ 1));
    const eh_lds_i* const imap = (const eh_lds_i*)z.imap;
    const eh_lds_f* const meta = wl + G::PHI_OFF;
    // (requested before the barrier: none of it is written by the step's epilogue)
    const int idx0 = tid < nth ? tid : 0;
    float th0 = P[idx0], mm0 = P[nth + idx0], vv0 = P[2 * nth + idx0];
    const int mp0 = idx0 < net.g_off ? imap[idx0] : 0;
    const float bt1 = sc[0], bt2 = sc[1];
    __syncthreads();                                   // every sum of the step is in gsum
    EH_STAMP_FINE(11);
    const eh_lds_f* const tail = gsum + nth;           // [S | n | Sy | Syy] (one target: the multi-step launch takes no other)
    const float S = tail[0], cnt = tail[1], Sy = tail[2], Syy = tail[3];
    const bool upd = cnt > 0.0f;
    float inv = 0.0f, lossv = __builtin_nanf("");
    if (upd) eh_loss_finish(net.loss, S, cnt, Sy, Syy, inv, lossv, z.agg_a);
    for (int idx = tid; idx < nth; idx += nthr) {
        float th, mm, vv; int mp;
        if (idx == tid) { th = th0; mm = mm0; vv = vv0; mp = mp0; }
        else { th = P[idx]; mm = P[nth + idx]; vv = P[2 * nth + idx]; mp = idx < net.g_off ? imap[idx] : 0; }
        if (upd) {
            eh_opt_update(z.opt, gsum[idx] * inv, bt1, bt2, th, mm, vv);
            P[idx] = th; P[nth + idx] = mm; P[2 * nth + idx] = vv;
            if (idx < net.g_off) wl[mp] = th;
            else {
                const int j = __float_as_int(meta[EH_IMG_GPAR + idx - net.g_off]);
                const float sg = 1.0f / (1.0f + expf(-th)), scl = meta[EH_IMG_SC + j];
                wl[G::PHI_OFF + EH_IMG_PHI + j] = meta[EH_IMG_LO + j] + scl * sg;
                wl[G::PHI_OFF + EH_IMG_DPHI + j] = scl * sg * (1.0f - sg);
            }
        }
    }
    // (no second barrier: the beta products were read before the barrier above, and the sums are overwritten -- plain stores, one writer per
    //  element -- only behind the next step's barriers; the multi-step kernel zeroes the array once, when the launch ends)
    if (tid == 0) {
        if (upd) { sc[0] = bt1 * z.opt.b1; sc[1] = bt2 * z.opt.b2; }
        if (a.ms_loss) *a.ms_loss = lossv;
    }
}

// one sample record per lane as the step body holds it between the fetch and the forward pass; EhCarry hands the NEXT step's records
// from one step of a multi-step launch to the following one
template <int NX4>
struct EhRec { f32x4 x[NX4]; float frc[EH_MAX_FORC]; float y[EH_MAX_TARG]; };
template <int NX4>
struct EhCarry { EhRec<NX4> rec; bool live; bool valid; };
// (the exchange's scalar sums on their way from the eight threads that fetch them to everybody; static LDS only in the instantiations that exchange)
template <bool ON>
__device__ __forceinline__ float* eh_px_table() {
    if constexpr (ON) { __shared__ __attribute__((aligned(16))) float T[8]; return T; }
    else return nullptr;
}
template <int NBI, int NBH, int NL, int NT, int NW, int ACT, int MODE, int FAST>
__device__ __forceinline__ void eh_step_body(const EhNet& net_rt, const EhStepArgs& a, EhCarry<(EhGeom<NBI, NBH, NL, NT, NW>::IP + 3) / 4>* carry = nullptr) {
    // Run-time compiled kernels (eh_jit.hip) know the model: the descriptor is a compile-time constant there and the generality
    // below -- per-parameter kinds, per-target switches, the mechanistic switch -- folds away.
#ifdef EH_SPEC_NET
    constexpr EhNet net = {EH_SPEC_NET};
#else
    const EhNet& net = net_rt;
#endif
    using G = EhGeom<NBI, NBH, NL, NT, NW>;
    constexpr int MT = G::MT, SR = G::SR, HP = G::HP, S0 = G::S0, SH = G::SH, NTHR = 64 * NW;
    constexpr bool TRAIN = MODE != EH_MODE_EVAL;
    constexpr bool P2PM = MODE == EH_MODE_TRAIN_P2P;      // its own instantiation: the single-GPU kernel carries none of this
    constexpr bool K1 = (FAST & 1) != 0, PS = (FAST & 2) != 0;
    constexpr bool PROG = (FAST & 4) != 0;                // EH_MECH_PROGRAM: the mechanistic stage interprets a.prog
    static_assert(!PROG || FAST == 4, "the program kernels are generic kernels");
    constexpr bool KEEPH = TRAIN && !EhStoresZ<ACT>::value && NL * NBH * NT * 4 <= EH_KEEPH_MAX;   // activations stay in registers for act'
    constexpr int NHS = KEEPH ? NL : 1, NHM = KEEPH ? NBH : 1, NHT = KEEPH ? NT : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    // (the wave index as a SCALAR: derived from threadIdx it counts as divergent, and every loop / branch on it -- the tile loop
    //  first of all -- would run under an exec mask with saved / restored mask pairs instead of scalar branches)
#ifdef EH_AB_VECTOR_WAVE_INDEX
    const int wave = tid >> 6;                               // (diagnostic A/B: the round-1 form)
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    constexpr bool EVALM = MODE == EH_MODE_EVAL;
    float* const ws = smem + G::IMG_FLOATS + wave * (EVALM ? (K1 ? G::WS_EVAL_K1 : G::WS_EVAL) : G::WAVE_WS);
    float* const XS = ws + G::XS_OFF;
    float* const HS = ws + G::HS_OFF;
    float* const OS = ws + (EVALM ? G::OS_OFF_EVAL : G::OS_OFF);
    const float* const meta = wl + G::PHI_OFF;
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    // Per-parameter / per-target switches as per-lane values pinned in VGPRs: as wave-uniform
    // conditions the compiler would hoist ~30 of them out of the tile loop into SGPR masks and then
    // spill them (measured: 117 SGPR spills); as VGPRs they cost one compare where they are used.
    // (Not in the run-time specialised kernels: there the switches are compile-time constants and what they guard folds away.)
#ifdef EH_SPEC_NET
#define EH_PIN(...)
#else
#define EH_PIN(...) asm volatile("" : __VA_ARGS__)
#endif
    float kN[EH_MAX_PARAMS], kG[EH_MAX_PARAMS], tOn[EH_MAX_TARG], sclOn = net.scale_nn ? 1.0f : 0.0f;
    int oOff[EH_MAX_PARAMS], fCol[EH_MAX_FORC];
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) {
        kN[j] = (j < net.n_par && pkind(j) == EH_PAR_NEURAL) ? 1.0f : 0.0f;
        kG[j] = (j < net.n_par && pkind(j) == EH_PAR_GLOBAL) ? 1.0f : 0.0f;
        oOff[j] = pidx(j) * SR;
        EH_PIN("+v"(kN[j]), "+v"(kG[j]), "+v"(oOff[j]));
    }
    int tOut[EH_MAX_TARG];
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        tOn[t] = t < net.T ? 1.0f : 0.0f; tOut[t] = (int)((net.targ_out >> (2 * t)) & 3u);
        EH_PIN("+v"(tOn[t]), "+v"(tOut[t]));
    }
    float multiOn = net.n_out > 1 ? 1.0f : 0.0f;
    EH_PIN("+v"(multiOn));
#pragma unroll
    for (int f = 0; f < EH_MAX_FORC; ++f) {
        const unsigned col = (net.forc_col >> (8 * f)) & 0xFFu;
        fCol[f] = col == 0xFFu ? -1 : (int)(net.P + col);
        EH_PIN("+v"(fCol[f]));
    }
    float maeT[EH_MAX_TARG], twoT[EH_MAX_TARG];
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        maeT[t] = eh_target_mae(net.loss_t, t) ? 1.0f : 0.0f; twoT[t] = eh_target_two_pass(net.loss_t, t, net.T) ? 1.0f : 0.0f;
        EH_PIN("+v"(maeT[t]), "+v"(twoT[t]));
    }
    EH_PIN("+v"(sclOn));
    const int tcol0 = net.P + net.F;

    // one sample record per lane, fetched one macro-tile ahead of its use
    constexpr int NX4 = (G::IP + 3) / 4;
    EhRec<NX4> nx;
    const int count = (int)a.count, first = (int)a.first;     // N <= 2^31 - 1 (checked by eh_set_data)
    const int ntiles = (count + MT - 1) / MT;
    // (in two halves: a gathered minibatch reads its record THROUGH the epoch's permutation -- a dependent pair of loads.  The step's first
    //  fetch asks for the index up front and for the record only once everything else the prologue needs is in flight -- state, exchange
    //  words, parameter image -- so the index's round trip is theirs as well; asked for first, as one piece, the pair put two round trips
    //  in front of everything else.)
    int nx_glb = 0;
    bool nx_live = false;
    auto fetch_idx = [&](int tile) {
        const int n_loc = tile * MT + lane;
        nx_live = (tile < ntiles) && (lane < MT) && (n_loc < count);
        nx_glb = nx_live ? (a.idx ? a.idx[first + n_loc] : first + n_loc) : 0;
    };
    auto fetch_rec_into = [&](EhRec<NX4>& dst) {
        const bool live = nx_live;
        const float* const rec = a.recs + (long long)nx_glb * a.C;
        if ((a.C & 3) == 0) {          // 16-byte-multiple records (RbQ10: exactly one dwordx4 per sample)
#pragma unroll
            for (int q = 0; q < NX4; ++q) dst.x[q] = (live && 4 * q < net.P) ? *(const f32x4*)(rec + 4 * q) : f32x4{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int q = 0; q < NX4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst.x[q][e] = (live && 4 * q + e < net.P) ? rec[4 * q + e] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) dst.frc[f] = (fCol[f] >= 0 && live) ? rec[fCol[f]] : 0.0f;
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) dst.y[t] = (tOn[t] != 0.0f && live) ? rec[tcol0 + t] : __builtin_nanf("");
    };
    auto fetch_rec = [&]() { fetch_rec_into(nx); };
    auto fetch = [&](int tile) { fetch_idx(tile); fetch_rec(); };
    // (a later step of a multi-step launch: the step before it fetched this step's records behind its own compute -- a gathered record is a
    //  dependent pair of loads, index then record, and with nothing else to hide behind in a step whose image is already in LDS the pair
    //  was 0.5 us at the head of a 4.5 us step)
    const bool carried = carry && carry->valid;
    bool rec_pending = false;
    if (carried) { nx = carry->rec; nx_live = carry->live; }
    else {
        fetch_idx((int)blockIdx.x * NW + wave);
        rec_pending = a.idx != nullptr;
        if (!rec_pending) fetch_rec();               // contiguous minibatch: nothing to wait for
    }
    bool pf_issued = false;

    EH_STAMP(0);
    // ---- stage the parameter image into LDS (straight copy) ------------------------------------
    // (fused update) the first chunk of optimiser inputs is requested together with the image so
    // that everything arrives in one memory round trip
    const bool fusedm = TRAIN && a.fz.gacc != nullptr;
    float f_th = 0.0f, f_m = 0.0f, f_v = 0.0f, f_g = 0.0f, f_cnt = 0.0f, f_sse = 0.0f, f_sy = 0.0f, f_syy = 0.0f, f_bt1 = 0.0f, f_bt2 = 0.0f;
    int f_map = 0;
    // (and the reduction-map entry of the first element this thread gathers in the epilogue: its load would otherwise sit, a
    // full memory round trip, between the last barrier and the LDS reads)
    int f_rcode = 0;
    if constexpr (TRAIN) f_rcode = (a.rmap && tid < a.n_acc) ? a.rmap[tid] : 0;
    // (the running statistics of the input BatchNorm, which the workgroup that updates them would otherwise read where it needs them)
    float f_rm = 0.0f, f_rv = 0.0f;
    if ((a.bn_part || a.bn_nblk == -1) && a.bn_update && blockIdx.x == 0 && tid < net.P) { f_rm = a.bn_run[tid]; f_rv = a.bn_run[32 + tid]; }
    // (ms_direct && ms_keep: a later step of a multi-step launch -- the step before it has applied its own update and written the new
    //  parameters into the LDS image: nothing deferred to pick up here)
    const bool deferred_upd = fusedm && !(a.ms_direct && a.ms_keep);
    // (the exchange descriptor's fields once, up front, into scalar registers: read where they are used -- inside address lambdas, behind
    //  branches -- every use was a chain of dependent scalar loads of its own, rank -> peer_recv[rank] -> the word)
    const int px_mode = P2PM ? a.p2pv.mode : 0, px_rank = P2PM ? a.p2pv.rank : 0, px_world = P2PM ? a.p2pv.world : 1;
    const unsigned long long* const px_recv = P2PM ? a.p2pv.peer_recv[px_rank] + (long long)((a.fz.gslot + 2) % 3) * EH_GSHARDS * a.n_acc : nullptr;
    const float* const px_stage = P2PM ? a.p2pv.stage : nullptr;
    float px_own[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // (EhP2P mode 1: this rank's own sums of the previous step come from its staging shards, like the single-GPU step's from its accumulators)
    const bool own_direct = deferred_upd && a.fz.pending && (!P2PM || px_mode == 1);
    float f_sv = 0.0f;      // lane 8 k + sh of every wave: scalar k of shard sh
    float f_gs[EH_GSHARDS] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // this thread's element in the eight shards
    if (deferred_upd) {
        const EhFused& z = a.fz;
        const float* const g_prev = (P2PM ? px_stage : z.gacc) + ((z.gslot + 2) % 3) * (EH_GSHARDS * a.n_acc);
        const float* const pin = z.pset + z.cur * 3 * net.n_theta;
        // [S | n | Sy | Syy] (one target) or [S | n_t ... | Sy | Syy] behind the gradient in each of the eight shards: FIVE scalars either way
        // (one float of slack behind the last shard), one load per lane -- lanes 0..39 of every wave -- folded across the lanes once the
        // loads are back (below, behind the image).  No branch on the target count: a uniform branch here split the prologue's single
        // memory round trip in two (+1.7 us per headline step on the kernels built ahead of time).  (Every thread used to load all forty
        // itself: 40 loads + their addresses in each of the workgroup's waves, ~0.6 us of a prologue that is bound by the instructions it issues.)
        if (own_direct && lane < 40) f_sv = g_prev[(lane & 7) * a.n_acc + net.n_theta + (lane >> 3)];
        const float* const sc_in = z.pset + 6 * net.n_theta + 2 * z.sc_sel;
        f_bt1 = sc_in[0]; f_bt2 = sc_in[1];
        if (tid < net.n_theta) {
            f_th = pin[tid]; f_m = pin[net.n_theta + tid]; f_v = pin[2 * net.n_theta + tid];
            f_map = tid < net.g_off ? z.imap[tid] : 0;
            if (own_direct) {       // (folded below, behind the image: folded here, the loads' round trip would come before the image's)
#pragma unroll
                for (int sh = 0; sh < EH_GSHARDS; ++sh) f_gs[sh] = g_prev[sh * a.n_acc + tid];
            }
        }
    }
    // EH_MODE_TRAIN_P2P: the sums of all ranks for the previous step, every 64-bit word with its own arrival stamp.  Thread tid asks
    // for ITS element of every rank's shard (what the single-GPU prologue reads from its eight local shards), the last eight threads
    // also for one of the five scalars [S | n_1 .. | Sy | Syy] behind the gradient; requested here together with the loads above and
    // the image below, examined (and re-read while a peer is late) once the image is staged.  Everything is summed in registers, in
    // RANK order on every rank, so the replicas stay bitwise identical; the scalar sums reach the other threads through 8 floats of LDS.
    // (A first version dealt the world x (n_theta + 4) words round-robin over the threads and parked all of them in LDS: nine address
    //  computations per thread whatever the world, a table walk per rank, two barriers.)
    EH_STAMP_PRO(2);
    float* const px_T = eh_px_table<P2PM>();
    const int px_nmain = net.n_theta < NTHR ? net.n_theta : NTHR;
    // (written out rather than through eh_ll_issue / eh_ll_finish: what depends on the rank count only is a scalar branch per rank, so a
    //  small world pays for its own ranks and not for eight -- all waves of the workgroup run this code, and the prologue is bound by
    //  the instructions they issue, not by the memory it waits for)
    const unsigned px_seq = a.p2p_seq - 1u;
    const unsigned long long px_none = (unsigned long long)px_seq << 32;      // "arrived, 0.0"
    unsigned long long px_w[P2PM ? EH_GSHARDS : 1], px_s[P2PM ? EH_GSHARDS : 1];
    const int px_k = tid - (NTHR - 8);                      // the last eight threads: scalar k of every rank
    const bool px_sc = px_k >= 0 && px_k < (net.T > 3 ? 5 : 4);
    auto px_request = [&]() {
#pragma unroll
        for (int sh = 0; sh < (P2PM ? EH_GSHARDS : 1); ++sh) {
            px_w[sh] = px_none; px_s[sh] = px_none;
            if (sh < px_world && !(px_mode == 1 && sh == px_rank)) {      // (own sums in mode 1: taken from the staging shards above)
                const unsigned long long* const q = px_recv + (long long)sh * a.n_acc;
                if (tid < px_nmain) px_w[sh] = __hip_atomic_load(q + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (px_sc) px_s[sh] = __hip_atomic_load(q + net.n_theta + px_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if constexpr (P2PM) {
        // mode 1: the previous step's kernel is through (stream order), so its adds have all landed -- no ticket, no election: this workgroup
        // folds the staging shards and publishes them to the peers, then waits for the words like everybody else
        if (fusedm && a.fz.pending && px_mode == 1 && px_world > 1 && blockIdx.x == 0) eh_p2p_fold_store(&a.p2pv, (a.fz.gslot + 2) % 3, px_seq, a.n_acc, tid, NTHR, true);
        if (fusedm && a.fz.pending) px_request();
    }
    EH_STAMP_PRO(3);
    if (!a.ms_keep) {   // all loads first, then the LDS stores: one memory round trip instead of one per 16 bytes
        // (ms_keep: a later step of a multi-step launch -- the image is still in LDS from the step before, whose update wrote the new
        //  parameters into it; its constant part never changes)
        constexpr int NI = (G::IMG_FLOATS / 4 + NTHR - 1) / NTHR, NIB = NI < 16 ? NI : 16;
        for (int e0 = 4 * tid; e0 < G::IMG_FLOATS; e0 += 4 * NTHR * NIB) {
            f32x4 tmp[NIB];
#pragma unroll
            for (int u = 0; u < NIB; ++u) {
                const int e = e0 + 4 * NTHR * u;
                tmp[u] = e < G::IMG_FLOATS ? *(const f32x4*)&a.image[e] : f32x4{0, 0, 0, 0};
            }
            if (rec_pending) { fetch_rec(); rec_pending = false; }
#pragma unroll
            for (int u = 0; u < NIB; ++u) {
                const int e = e0 + 4 * NTHR * u;
                if (e < G::IMG_FLOATS) *(f32x4*)&wl[e] = tmp[u];
            }
        }
    }
    if (rec_pending) fetch_rec();
    if (own_direct) {
        f_g = eh_fold8(f_gs[0], f_gs[1], f_gs[2], f_gs[3], f_gs[4], f_gs[5], f_gs[6], f_gs[7]);
        const int svi = __builtin_bit_cast(int, eh_fold8_lanes(f_sv));
        const float S0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(svi, 0)), S1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(svi, 8)),
                    S2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(svi, 16)), S3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(svi, 24)),
                    S4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(svi, 32));
        if constexpr (P2PM) { px_own[0] = S0; px_own[1] = S1; px_own[2] = S2; px_own[3] = S3; px_own[4] = S4; }
        else {
            // Multi-target steps used exact per-target weights (a.inv_n): only "any valid sample" matters, Sy / Syy are not read.
            f_sse = S0; f_cnt = S1 + (net.T > 1 ? S2 : 0.0f) + (net.T > 2 ? S3 : 0.0f) + (net.T > 3 ? S4 : 0.0f); f_sy = S2; f_syy = S3;
        }
    }
    EH_STAMP_PRO(4);
    if constexpr (P2PM) {
        if (fusedm && a.fz.pending) {
            // every word carries its arrival stamp: re-read while a peer is late; a deadline on the wall clock (eh_ll_finish's) turns a
            // missing peer into the error flag, words that never arrive read as 0
            {
                unsigned long long t0 = 0;
                bool timing = false;
                while (true) {
                    bool all = true;
#pragma unroll
                    for (int sh = 0; sh < EH_GSHARDS; ++sh)
                        if (sh < px_world) all = all && ((unsigned)(px_w[sh] >> 32) == px_seq) && ((unsigned)(px_s[sh] >> 32) == px_seq);
                    if (all) break;
                    if (__hip_atomic_load(a.p2pv.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (!timing) { t0 = wall_clock64(); timing = true; }
                    else if (wall_clock64() - t0 > EH_P2P_DEADLINE_TICKS) { __hip_atomic_store(a.p2pv.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(2);
                    px_request();
                }
            }
            EH_STAMP_PRO(11);
            // this thread's element and (the last eight threads) scalar, summed over the ranks in RANK order; mode 1 puts the rank's own sums --
            // read from its staging shards, folded in shard order as the publishing workgroup folds them for the peers -- in their place
            const bool own1 = px_mode == 1;
            const float own_g = f_g;                             // (mode 0: nothing was read from the staging shards, 0)
            const float own_s = px_k == 0 ? px_own[0] : px_k == 1 ? px_own[1] : px_k == 2 ? px_own[2] : px_k == 3 ? px_own[3] : px_own[4];
            float fs = 0.0f;
            f_g = 0.0f;
#pragma unroll
            for (int sh = 0; sh < EH_GSHARDS; ++sh)
                if (sh < px_world) {
                    const bool own = own1 && sh == px_rank;
                    f_g += own ? own_g : ((unsigned)(px_w[sh] >> 32) == px_seq) ? __uint_as_float((unsigned)px_w[sh]) : 0.0f;
                    fs += own ? own_s : ((unsigned)(px_s[sh] >> 32) == px_seq) ? __uint_as_float((unsigned)px_s[sh]) : 0.0f;
                }
            if (px_k >= 0) px_T[px_k] = fs;
            EH_STAMP_PRO(12);
            __syncthreads();
            EH_STAMP_PRO(13);
            {
                const f32x4 t4 = *(const f32x4*)px_T;
                f_sse = t4[0];
                f_cnt = t4[1] + (net.T > 1 ? t4[2] : 0.0f) + (net.T > 2 ? t4[3] : 0.0f) + (net.T > 3 ? px_T[4] : 0.0f);
                f_sy = t4[2]; f_syy = t4[3];
            }
        }
    }
    EH_STAMP_PRO(5);
    // small minibatches (a.bn_nblk == -1, count <= EH_BN_SELF_MAX: the reference's tutorial trains on 64): every workgroup takes the
    // statistics of the whole minibatch itself -- thread (g, p) sums predictor p over the samples g, g + NTHR / 32, ... about the first
    // sample's value -- instead of waiting for a launch of eh_bn_stats_kernel in front of the step (a dependent launch costs more than
    // the step's own work at this size); same sums in every workgroup, so the replicas of the image stay identical
    const bool bn_self = a.bn_nblk == -1;
    static_assert((NTHR / 32 + 1) * 64 <= NW * G::WAVE_WS && (NTHR / 32 + 1) * 64 <= NW * G::WS_EVAL_K1, "the statistics scratch fits the waves' work space");
    float* const bn_red = smem + G::IMG_FLOATS;                  // [NTHR / 32][64], in the (not yet cleared) X images
    // One workgroup covering the whole minibatch (the multi-step launches; any single launch of at most 16 NT NW samples): its waves HOLD
    // the minibatch -- one record per lane, fetched above for the forward pass.  Every wave sums its own lanes about its first sample's
    // value on the DPP network, {centre, sum, sum of squares, count} per wave and predictor go through LDS, and thread p merges the
    // waves' sums about wave 0's centre (the pairwise update of a variance: S2 = sum_w [s2_w + 2 (c_w - c) s1_w + n_w (c_w - c)^2]).
    // The gather below reads the minibatch a second time through the permutation -- index, then record, then the centre's own pair:
    // dependent round trips that were 1.6 us of a 6.1 us step of the tutorial's model.
    const bool bn_regs = bn_self && gridDim.x == 1 && ntiles <= NW;
    // thread p < P: sums of (x - c0), (x - c0)^2 over the minibatch -> mean and 1 / std into the LDS image, the running statistics updated
    auto bn_to_image = [&](float s1, float s2, float c0) {
        const float m = a.bn_n ? *a.bn_n : (float)count;
        const float d = s1 / m, var = fmaxf(s2 / m - d * d, 0.0f), mu = c0 + d;
        wl[G::PHI_OFF + EH_IMG_BNM + tid] = mu;
        wl[G::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(var + EH_BN_EPS);
        if (a.bn_update && blockIdx.x == 0) {
            const float rm = (1.0f - EH_BN_MOMENTUM) * f_rm + EH_BN_MOMENTUM * mu;
            const float rv = (1.0f - EH_BN_MOMENTUM) * f_rv + EH_BN_MOMENTUM * (m > 1.0f ? m / (m - 1.0f) : 1.0f) * var;
            a.bn_run[tid] = rm; a.bn_run[32 + tid] = rv;
            a.image_out[G::PHI_OFF + EH_IMG_BNM + tid] = rm;                       // what forward / eval (test mode) will use
            a.image_out[G::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(rv + EH_BN_EPS);
        }
    };
    static_assert(NW * G::IP * 4 <= NW * G::WS_EVAL_K1, "the per-wave statistics fit the waves' work space");
    float bn_s1r = 0.0f, bn_s2r = 0.0f;
    if (bn_regs) {
        const float n_w = (float)__popcll(__ballot(nx_live));
#pragma unroll
        for (int q = 0; q < NX4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < net.P) {
                    const float xv = nx.x[q][e];
                    const float cw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), 0));      // (live lanes are a prefix)
                    const float d = nx_live ? xv - cw : 0.0f;
                    const float s1 = eh_wave_sum(d), s2 = eh_wave_sum(d * d);
                    if (lane == 0) *(f32x4*)&bn_red[(wave * G::IP + 4 * q + e) * 4] = f32x4{cw, s1, s2, n_w};
                }
        EH_STAMP_PRO(11);
        __syncthreads();
        if (tid < net.P) {
            const float c = bn_red[tid * 4];
#pragma unroll
            for (int w = 0; w < NW; ++w) {       // (unrolled: the eight reads go out together -- one after the other they were a chain of LDS round trips)
                const f32x4 t = *(const f32x4*)&bn_red[(w * G::IP + tid) * 4];
                const float dc = t[0] - c;
                bn_s1r += t[1] + t[3] * dc;
                bn_s2r += t[2] + 2.0f * dc * t[1] + t[3] * dc * dc;
            }
            bn_to_image(bn_s1r, bn_s2r, c);      // (here, in front of the barrier the table needs anyway: a block and a barrier of its own further down otherwise)
        }
        EH_STAMP_PRO(12);
        __syncthreads();                                         // (the table sits where the X images are about to be cleared)
    } else if (bn_self) {
        const int p = tid & 31, grp = tid >> 5;
        float s1 = 0.0f, s2 = 0.0f;
        if (p < net.P) {
            const long long n0 = a.idx ? (long long)a.idx[a.first] : a.first;
            const float c0 = a.recs[n0 * a.C + p];
#pragma unroll 8
            for (int i = grp; i < count; i += NTHR / 32) {
                const long long n = a.idx ? (long long)a.idx[a.first + i] : a.first + i;
                const float d = a.recs[n * a.C + p] - c0;
                s1 += d; s2 += d * d;
            }
        }
        bn_red[grp * 64 + p] = s1; bn_red[grp * 64 + 32 + p] = s2;
        __syncthreads();
        if (tid < net.P) {
            s1 = 0.0f; s2 = 0.0f;
            for (int b = 0; b < NTHR / 32; ++b) { s1 += bn_red[b * 64 + tid]; s2 += bn_red[b * 64 + 32 + tid]; }
            bn_red[(NTHR / 32) * 64 + tid] = s1; bn_red[(NTHR / 32) * 64 + 32 + tid] = s2;      // parked across the clearing of the X images below
        }
        __syncthreads();
    }
    float bn_s1 = bn_s1r, bn_s2 = bn_s2r;
    if (bn_self && !bn_regs) {          // (uniform; the barrier only where the scratch is in use: the headline step has no BatchNorm and no barrier to spare)
        if (tid < net.P) { bn_s1 = bn_red[(NTHR / 32) * 64 + tid]; bn_s2 = bn_red[(NTHR / 32) * 64 + 32 + tid]; }
        __syncthreads();
    }
    for (int e = lane; e < G::IP * SR; e += 64) XS[e] = 0.0f;   // rows >= P of the X image stay 0
    // (the X image is the wave's own: the workgroup barrier here is for what ELSE sits in front of the tile loop -- the parameter image
    //  staged above by all threads.  A later step of a multi-step launch stages nothing: its image was completed behind a barrier by the
    //  step before, and the statistics block above ends with a barrier of its own)
    if (a.ms_keep) EH_WAVE_SYNC(); else __syncthreads();
    EH_STAMP_PRO(13);
    if ((a.bn_part || bn_self) && !bn_regs) {
        // input BatchNorm, train mode (Lux BatchNorm, affine = false): statistics of THIS minibatch
        if (tid < net.P) {
            float s1 = bn_s1, s2 = bn_s2;
            if (!bn_self) for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const float c0 = bn_self ? a.recs[(a.idx ? (long long)a.idx[a.first] : a.first) * a.C + tid] : a.bn_c[tid];
            bn_to_image(s1, s2, c0);
        }
        __syncthreads();
    }
    EH_STAMP_PRO(6);
    if (deferred_upd) {
        // fused update: apply the previous step's optimiser update straight into the LDS image
        const EhFused& z = a.fz;
        const int nth = net.n_theta;
        const float* const g_prev = z.gacc + ((z.gslot + 2) % 3) * (EH_GSHARDS * a.n_acc);
        // the accumulators the NEXT step adds into: the receive buffer itself, or the local staging copy under EhP2P
        float* const g_zero = (P2PM ? const_cast<float*>(px_stage) : z.gacc) + ((z.gslot + 1) % 3) * (EH_GSHARDS * a.n_acc);
        const float* const pin = z.pset + z.cur * 3 * nth;
        float* const pout = z.pset + (z.cur ^ 1) * 3 * nth;
        const bool upd = z.pending && f_cnt > 0.0f;
        float inv = 0.0f, lossv = __builtin_nanf("");
        if (upd) {
            if (a.inv_n) { inv = 1.0f; lossv = f_sse; }      // multi-target: per-target weights from the counting pre-pass, the sums are final
            else eh_loss_finish(net.loss, f_sse, f_cnt, f_sy, f_syy, inv, lossv, z.agg_a);
        }
        // (which workgroup stores an element: its index modulo the largest power of two within the grid -- a mask; the remainder by the grid
        //  size itself was a 32-bit division in every thread of every workgroup)
        const unsigned gmask = (1u << (31 - __builtin_clz(gridDim.x))) - 1u;
        for (int idx = tid; idx < nth; idx += NTHR) {
            float th, mm, vv, gs = 0.0f;
            int mp;
            if (idx == tid) { th = f_th; mm = f_m; vv = f_v; gs = f_g; mp = f_map; }
            else {
                th = pin[idx]; mm = pin[nth + idx]; vv = pin[2 * nth + idx];
                mp = idx < net.g_off ? z.imap[idx] : 0;
                if (upd) {
                    if constexpr (P2PM) {
                        auto ad = [&](int sh) -> const unsigned long long* {
                            if (px_mode == 1 && sh == px_rank) return nullptr;
                            return sh < px_world ? px_recv + (long long)sh * a.n_acc + idx : nullptr;
                        };
                        float own = 0.0f;
                        if (px_mode == 1) {
                            const float* const gst = px_stage + ((z.gslot + 2) % 3) * (EH_GSHARDS * a.n_acc) + idx;
                            own = eh_fold8(gst[0], gst[a.n_acc], gst[2 * a.n_acc], gst[3 * a.n_acc], gst[4 * a.n_acc], gst[5 * a.n_acc], gst[6 * a.n_acc], gst[7 * a.n_acc]);
                        }
                        unsigned long long w8[EH_GSHARDS];
                        float got[EH_GSHARDS];
                        eh_ll_issue(ad, a.p2p_seq - 1u, w8);
                        eh_ll_finish(&a.p2pv, ad, a.p2p_seq - 1u, w8, got);
#pragma unroll
                        for (int sh = 0; sh < EH_GSHARDS; ++sh) gs += (px_mode == 1 && sh == px_rank) ? own : got[sh];      // (rank order on every rank)
                    } else {
                        const float* const gq = g_prev + idx;
                        gs = eh_fold8(gq[0], gq[a.n_acc], gq[2 * a.n_acc], gq[3 * a.n_acc], gq[4 * a.n_acc], gq[5 * a.n_acc], gq[6 * a.n_acc], gq[7 * a.n_acc]);
                    }
                }
            }
            if (upd) eh_opt_update(z.opt, gs * inv, f_bt1, f_bt2, th, mm, vv);
            if (((unsigned)idx & gmask) == blockIdx.x) { pout[idx] = th; pout[nth + idx] = mm; pout[2 * nth + idx] = vv; }   // every workgroup holds the same values: spread the stores
            if (idx < net.g_off) {
                wl[mp] = th;
            } else {
                const int j = __float_as_int(meta[EH_IMG_GPAR + idx - net.g_off]);
                const float sg = 1.0f / (1.0f + expf(-th)), sc = meta[EH_IMG_SC + j];
                wl[G::PHI_OFF + EH_IMG_PHI + j] = meta[EH_IMG_LO + j] + sc * sg;
                wl[G::PHI_OFF + EH_IMG_DPHI + j] = sc * sg * (1.0f - sg);
            }
        }
        EH_STAMP_PRO(7);
        if (blockIdx.x == 0 && tid == 0) {
            float* const sc_out = z.pset + 6 * nth + 2 * (z.sc_sel ^ 1);
            sc_out[0] = upd ? f_bt1 * z.opt.b1 : f_bt1;
            sc_out[1] = upd ? f_bt2 * z.opt.b2 : f_bt2;
            if (z.loss_slot && z.pending) *z.loss_slot = lossv;
        }
        for (int e = blockIdx.x * NTHR + tid; e < EH_GSHARDS * a.n_acc; e += gridDim.x * NTHR) g_zero[e] = 0.0f;
        __syncthreads();
    }
    EH_STAMP(1);

    // ---- accumulators (registers, live across the tile loop) ------------------------------------
    f32x4 aW0[PS ? 1 : NBH][PS ? 1 : NBI], aWh[NL > 1 ? NL - 1 : 1][NBH][NBH], aWo[K1 ? 1 : NBH];
    f32x4 aW0V[PS ? NBH : 1][4];      // PS: [m][p][r]  (p = predictor)
    f32x4 aWoV[K1 ? NBH : 1];         // K1: d/dWo[16m + 4g + r]
    f32x4 aB[NL][NBH], aBo;
    float aBoS = 0.0f;
    float gacc[EH_MAX_PARAMS];
    float lacc = 0.0f, syacc = 0.0f, syyacc = 0.0f;
    float cacc[EH_MAX_TARG];
    float est[EH_MAX_TARG][EH_EVAL_STATS];
    if constexpr (TRAIN) {
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
            if constexpr (!PS) {
#pragma unroll
                for (int n = 0; n < NBI; ++n) aW0[m][n] = f32x4{0, 0, 0, 0};
            } else {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) aW0V[m][pp] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                for (int n = 0; n < NBH; ++n) aWh[l][m][n] = f32x4{0, 0, 0, 0};
            if constexpr (K1) aWoV[m] = f32x4{0, 0, 0, 0}; else aWo[m] = f32x4{0, 0, 0, 0};
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
    const int ksteps0 = (net.P + 3) / 4;   // k-steps of layer 0 that hold real features (natural k order 4s+g)

    for (int tile = (int)blockIdx.x * NW + wave; tile < ntiles; tile += (int)gridDim.x * NW) {
        const int n_loc = tile * MT + lane;                        // sample of this lane in the mech stage
        const bool live = (lane < MT) && (n_loc < count);

        EH_STAMP_FINE(2);
        // ---- 1. the record was fetched one iteration ahead: predictors -> [feature][sample] image
        float frc[EH_MAX_FORC], yobs[EH_MAX_TARG];
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) frc[f] = nx.frc[f];
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) yobs[t] = nx.y[t];
#pragma unroll
        for (int q = 0; q < NX4; ++q) {
            // (all four components of a fetched quad stay allocated up to here, used or not: the register allocator hands the unused ones of a
            //  16-byte load to the very next address computation, and a write to the destination of a load in flight waits for the load --
            //  in the step's prologue that put the record's whole round trip in front of the state loads behind it)
            asm volatile("" ::"v"(nx.x[q][0]), "v"(nx.x[q][1]), "v"(nx.x[q][2]), "v"(nx.x[q][3]));
        }
#pragma unroll
        for (int q = 0; q < NX4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < net.P && lane < MT)      // (x - mean) / std of the input BatchNorm; mean 0, 1/std 1 without it
                    XS[(4 * q + e) * SR + lane] = (nx.x[q][e] - meta[EH_IMG_BNM + 4 * q + e]) * meta[EH_IMG_BNR + 4 * q + e];
        {
            const int tnext = tile + (int)gridDim.x * NW;
            if (carry && tnext >= ntiles) {      // (multi-step launch, this wave's last tile of the step: the index of its sample in the NEXT step's window)
                if (a.pf_count > 0) {
                    const int n_nxt = ((int)blockIdx.x * NW + wave) * MT + lane;
                    nx_live = (lane < MT) && (n_nxt < (int)a.pf_count);
                    nx_glb = nx_live ? (a.idx ? a.idx[(int)a.pf_first + n_nxt] : (int)a.pf_first + n_nxt) : 0;
                    pf_issued = true;
                }
            } else fetch(tnext);             // next tile's record: in flight behind this tile's compute
        }
        EH_WAVE_SYNC();

        EH_STAMP_FINE(3);
        // ---- 2. forward, layer 0 : z = W0 x + b0 ------------------------------------------------
        f32x4 h[NBH][NT];
        f32x4 hs[NHS][NHM][NHT];       // KEEPH: every layer's activations for the backward pass
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
            for (int t = 0; t < NT; ++t) {
                const f32x4 z4 = h[m][t], hv4 = eh_act4_rows<ACT>(z4, 0, 16 * m + 4 * g);
                h[m][t] = hv4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (TRAIN && (NL > 1 || !K1 || !KEEPH)) HS[(16 * m + 4 * g + r) * SR + 16 * t + c] = EhStoresZ<ACT>::value ? eh_vgpr(z4[r]) : hv4[r];
                if constexpr (KEEPH) hs[0][m][t] = h[m][t];
            }
        }
        EH_STAMP_FINE(4);
        // ---- 3. hidden layers -------------------------------------------------------------------
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            float* Hl = HS + l * HP * SR;
            f32x4 hn[NBH][NT];
            // (the A operands of row block m + 1 are requested before the MFMA chain of block m is issued: a wave issues in
            //  order, so a load placed right in front of its use exposes the whole LDS latency once per block)
            f32x4 a4c[NBH], a4n[NBH];
#pragma unroll
            for (int q = 0; q < NBH; ++q) a4c[q] = *(const f32x4*)&W[c * SH + 16 * q + 4 * g];
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                const f32x4 bias = *(const f32x4*)&wl[G::B_OFF + l * HP + 16 * m + 4 * g];
                if (m + 1 < NBH) {
#pragma unroll
                    for (int q = 0; q < NBH; ++q) a4n[q] = *(const f32x4*)&W[(16 * (m + 1) + c) * SH + 16 * q + 4 * g];
                }
                EH_SCHED_FENCE();
#pragma unroll
                for (int t = 0; t < NT; ++t) hn[m][t] = bias;
#pragma unroll
                for (int q = 0; q < NBH; ++q) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            hn[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4c[q][s], h[q][t][s], hn[m][t], 0, 0, 0);
                }
                EH_SCHED_FENCE();
#pragma unroll
                for (int q = 0; q < NBH; ++q) a4c[q] = a4n[q];
            }
#pragma unroll
            for (int m = 0; m < NBH; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 z4 = hn[m][t], hv4 = eh_act4_rows<ACT>(z4, l, 16 * m + 4 * g);
                    h[m][t] = hv4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)      // the last layer's image is only read back for act' / dWo when those do not have it in registers
                        if (TRAIN && (l < NL - 1 || !K1 || !KEEPH)) Hl[(16 * m + 4 * g + r) * SR + 16 * t + c] = EhStoresZ<ACT>::value ? eh_vgpr(z4[r]) : hv4[r];
                    if constexpr (KEEPH) hs[l < NHS ? l : 0][m < NHM ? m : 0][t < NHT ? t : 0] = h[m][t];
                }
        }
        EH_STAMP_FINE(5);
        // ---- 4. output layer (K <= 16 rows, zero padded) ----------------------------------------
        float om = 0.0f;               // K1: this lane's sample's single NN output
        if constexpr (K1) {
            float part[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) part[t] = 0.0f;
#pragma unroll
            for (int q = 0; q < NBH; ++q) {
                const f32x4 w4 = *(const f32x4*)&wl[G::WO_OFF + 16 * q + 4 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[t] = fmaf(w4[r], h[q][t][r], part[t]);
            }
            const float bo = wl[G::B_OFF + NL * HP];
#pragma unroll
            for (int t = 0; t < NT; ++t) {           // sum over the four feature quads g (lanes c, c+16, c+32, c+48)
                part[t] += __shfl_xor(part[t], 16, 64);
                part[t] += __shfl_xor(part[t], 32, 64);
                if (g == t) om = part[t] + bo;       // lane 16t + c owns sample 16t + c
            }
        } else {
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
            EH_WAVE_SYNC();
        }

        EH_STAMP_FINE(6);
        // ---- 5. mechanistic model + masked loss, one sample per lane -----------------------------
        float dOm = 0.0f;              // K1: d loss / d (this lane's NN output)
        {
            float par[EH_MAX_PARAMS], sg[EH_MAX_PARAMS], dydp[EH_MAX_PARAMS];
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                par[j] = meta[EH_IMG_PHI + j]; sg[j] = 1.0f; dydp[j] = 0.0f;
                if (kN[j] != 0.0f) {
                    const float ov = K1 ? om : ((lane < MT) ? OS[oOff[j] + lane] : 0.0f);
                    if (sclOn != 0.0f) {
                        const float s = eh_sigmoid(ov), sc = meta[EH_IMG_SC + j];
                        par[j] = fmaf(sc, s, meta[EH_IMG_LO + j]);
                        sg[j] = sc * s * (1.0f - s);
                    } else {
                        par[j] = ov;
                    }
                }
            }
            // Settle the next tile's record prefetch here, mid-tile, rather than at the loop top where its
            // registers are consumed: measured 2 % faster at 16 tiles per wave (the wait is almost always
            // free at this point and the next prefetch then issues without a stall).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float y0, yx[2] = {0.0f, 0.0f}, Jx[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
#ifdef EH_JIT_MECH
            EhJitTape jtape;
            if constexpr (PROG) eh_jit_fwd(par, frc, jtape, y0, yx[0], yx[1]);
            else {
#else
            float pval[PROG ? EH_PROG_SLOTS : 1];
            if constexpr (PROG) {
                eh_prog_forward(a.prog, par, frc, pval);
                y0 = pval[a.prog[2]];
                if (net.n_out > 1) yx[0] = pval[a.prog[3]];
                if (net.n_out > 2) yx[1] = pval[a.prog[4]];
            } else {
#endif
                y0 = eh_mech_eval(net.mech, par, frc, dydp);
                if (multiOn != 0.0f) eh_mech_extra(net.mech, par, frc, yx, Jx);
            }
            float dy = 0.0f, dyx[2] = {0.0f, 0.0f};          // d loss / d output 0, outputs 1..2
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t) {
                if (tOn[t] != 0.0f) {
                    const float y = tOut[t] == 0 ? y0 : (tOut[t] == 1 ? yx[0] : yx[1]);
                    const bool valid = live && !__builtin_isnan(yobs[t]);
                    const float r = valid ? y - yobs[t] : 0.0f;
                    if constexpr (TRAIN) {
                        const float* const tt = a.inv_n + EH_TT * t;
                        const float w = a.inv_n ? tt[0] : 1.0f;
                        const float cy = valid ? yobs[t] - a.shift[t] : 0.0f;
                        float d;
                        if (maeT[t] != 0.0f) { lacc += w * fabsf(r); d = r > 0.0f ? w : (r < 0.0f ? -w : 0.0f); }
#ifdef EH_JIT_LOSS
                        else if (eh_target_prog(net.loss_t, t)) {
                            float dl;
                            const float lv = eh_jit_loss(t, y, valid ? yobs[t] : y, dl);
                            lacc += valid ? w * lv : 0.0f;
                            d = valid ? w * dl : 0.0f;
                        }
#endif
                        else if ((FAST & 3) == 0 && twoT[t] != 0.0f) {      // two-pass losses (generic kernels only: the host drops the fast paths for them): d loss / d yhat = k0 + k1 (yhat - centre) + k2 (y - c), k from the batch moments (eh_moment_coef_kernel)
                            d = valid ? fmaf(tt[6], cy, fmaf(tt[5], y - tt[1], tt[4])) : 0.0f;
                        }
                        else { lacc += w * r * r; d = 2.0f * w * r; }
                        dy += tOut[t] == 0 ? d : 0.0f; dyx[0] += tOut[t] == 1 ? d : 0.0f; dyx[1] += tOut[t] == 2 ? d : 0.0f;
                        cacc[t] += valid ? 1.0f : 0.0f;
                        syacc += cy; syyacc += cy * cy;
                    } else if (valid) {
                        // (moment pass of the pearson / kge losses: yhat is centred on ITS OWN mean, found by a first pass --
                        // a common centre would cancel catastrophically when the predictions sit far from the targets)
                        const float cy = yobs[t] - a.shift[t], ch = y - (a.inv_n ? a.inv_n[EH_TT * t + 1] : a.shift[t]);
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
                continue;
            }
#ifdef EH_JIT_MECH
            float padj[EH_MAX_PARAMS];
            if constexpr (PROG) eh_jit_rev(par, frc, jtape, dy, dyx[0], dyx[1], padj);
#else
            float padj[PROG ? EH_PROG_SLOTS : 1];
            if constexpr (PROG) {
                const int nslot = EH_PROG_SLOT_INSTR + (int)a.prog[0];
                for (int i = 0; i < nslot; ++i) padj[i] = 0.0f;
                padj[a.prog[2]] += dy;
                if (net.n_out > 1) padj[a.prog[3]] += dyx[0];
                if (net.n_out > 2) padj[a.prog[4]] += dyx[1];
                eh_prog_reverse(a.prog, pval, padj);
            }
#endif
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) {
                float dp;
                if constexpr (PROG) dp = padj[j];
                else {
                    dp = dy * dydp[j];
                    if (j < 3) dp += dyx[0] * Jx[0][j] + dyx[1] * Jx[1][j];     // zero for the single-output models
                }
                dp = live ? dp : 0.0f;
                if (kN[j] != 0.0f) {
                    if constexpr (K1) dOm = dp * sg[j];
                    else if (lane < MT) OS[oOff[j] + lane] = dp * sg[j];
                }
                gacc[j] = fmaf(kG[j], dp, gacc[j]);
            }
        }
        if constexpr (!K1) EH_WAVE_SYNC();

        EH_STAMP_FINE(7);
        // ---- 6. backward ------------------------------------------------------------------------
        f32x4 dz[NBH][NT];
        constexpr bool DZ_LAST = NL > 1 || !PS;       // does the last hidden layer's delta feed an MFMA weight gradient?
        if constexpr (K1) {
            float dOt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) dOt[t] = __shfl(dOm, 16 * t + c, 64);   // sample 16t + c lives in lane 16t + c
            aBoS += dOm;
            float* const Hl = HS + (NL - 1) * HP * SR;
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                const f32x4 w4 = *(const f32x4*)&wl[G::WO_OFF + 16 * m + 4 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * m + 4 * g + r) * SR + 16 * t + c;
                        const float sv = KEEPH ? hs[NL - 1 < NHS ? NL - 1 : 0][m < NHM ? m : 0][t < NHT ? t : 0][r] : Hl[ad];
                        const float hv = eh_hval<ACT>(sv, NL - 1, 16 * m + 4 * g + r);
                        aWoV[m][r] = fmaf(dOt[t], hv, aWoV[m][r]);
                        const float d = w4[r] * dOt[t] * eh_dact_row<ACT>(sv, NL - 1, 16 * m + 4 * g + r);
                        dz[m][t][r] = d;
                        if constexpr (DZ_LAST) Hl[ad] = d;         // delta replaces the activation it was derived from (same lane, same word)
                    }
                    aB[NL - 1][m] += dz[m][t];
                }
            }
        } else {
            // output layer: dWo += dO * H_last^T ; dbo += dO ; dH = Wo^T dO
            f32x4 dO[NT], aT[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[t][r] = OS[(4 * g + r) * SR + 16 * t + c];
                aT[t] = *(const f32x4*)&OS[c * SR + 16 * t + 4 * g];
                aBo += dO[t];
            }
            float* const Hl = HS + (NL - 1) * HP * SR;
#pragma unroll
            for (int n = 0; n < NBH; ++n)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 b4 = *(const f32x4*)&Hl[(16 * n + c) * SR + 16 * t + 4 * g];
                    if (EhStoresZ<ACT>::value) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[s] = eh_hval<ACT>(b4[s], NL - 1, 16 * n + c);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) aWo[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[t][s], b4[s], aWo[n], 0, 0, 0);
                }
            const float* W = wl + G::WO_OFF;
            // dH = Wo^T dO over the K real output rows only: MFMA j contracts rows 4j + g (k-slot g of the instruction), whose
            // dO values come back from the OS image in that order -- ceil(K / 4) instructions per block instead of min(K, 4)
            const int nks = (net.K + 3) >> 2;
            float dOk[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) dOk[j][t] = j < nks ? OS[(4 * j + g) * SR + 16 * t + c] : 0.0f;
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                f32x4 dh[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dh[t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nks) {
                        const float av = W[(4 * j + g) * SH + 16 * m + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t) dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dOk[j][t], dh[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * m + 4 * g + r) * SR + 16 * t + c;
                        const float sv = KEEPH ? hs[NL - 1 < NHS ? NL - 1 : 0][m < NHM ? m : 0][t < NHT ? t : 0][r] : Hl[ad];
                        const float d = dh[t][r] * eh_dact_row<ACT>(sv, NL - 1, 16 * m + 4 * g + r);
                        dz[m][t][r] = d;
                        if constexpr (DZ_LAST) Hl[ad] = d;
                    }
                    aB[NL - 1][m] += dz[m][t];
                }
            }
        }
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            EH_WAVE_SYNC();
            // dW_l += dZ_l * H_{l-1}^T
            float* const Hp = HS + (l - 1) * HP * SR;
            const float* const DZ = HS + l * HP * SR;
            // (the B operands -- the previous layer's activations in operand order -- do not depend on the row block: read once;
            //  the A operands of block m + 1 are requested ahead of block m's MFMA chain, as in the forward pass)
            f32x4 bH[NBH][NT];
#pragma unroll
            for (int n = 0; n < NBH; ++n)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bH[n][t] = *(const f32x4*)&Hp[(16 * n + c) * SR + 16 * t + 4 * g];
                    if (EhStoresZ<ACT>::value) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) bH[n][t][s] = eh_hval<ACT>(bH[n][t][s], l - 1, 16 * n + c);
                    }
                }
            f32x4 aTc[NT], aTn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) aTc[t] = *(const f32x4*)&DZ[c * SR + 16 * t + 4 * g];
            const float* W = wl + G::WH_OFF + (l - 1) * HP * SH;
            float avc[NBH][4], avn[NBH][4];
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                if (m + 1 < NBH) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) aTn[t] = *(const f32x4*)&DZ[(16 * (m + 1) + c) * SR + 16 * t + 4 * g];
                } else {
#pragma unroll
                    for (int q = 0; q < NBH; ++q)
#pragma unroll
                        for (int s = 0; s < 4; ++s) avc[q][s] = W[(16 * q + 4 * g + s) * SH + c];
                }
                EH_SCHED_FENCE();
#pragma unroll
                for (int n = 0; n < NBH; ++n)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            aWh[l - 1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aTc[t][s], bH[n][t][s], aWh[l - 1][m][n], 0, 0, 0);
                EH_SCHED_FENCE();
#pragma unroll
                for (int t = 0; t < NT; ++t) aTc[t] = aTn[t];
            }
            // dH_{l-1} = W_l^T dZ_l ; dZ_{l-1} = dH ⊙ act'
            f32x4 dn[NBH][NT];
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                if (m + 1 < NBH) {
#pragma unroll
                    for (int q = 0; q < NBH; ++q)
#pragma unroll
                        for (int s = 0; s < 4; ++s) avn[q][s] = W[(16 * q + 4 * g + s) * SH + 16 * (m + 1) + c];
                }
                EH_SCHED_FENCE();
#pragma unroll
                for (int t = 0; t < NT; ++t) dn[m][t] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < NBH; ++q)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            dn[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(avc[q][s], dz[q][t][s], dn[m][t], 0, 0, 0);
                EH_SCHED_FENCE();
#pragma unroll
                for (int q = 0; q < NBH; ++q)
#pragma unroll
                    for (int s = 0; s < 4; ++s) avc[q][s] = avn[q][s];
            }
            EH_WAVE_SYNC();
            const bool need_dz = l > 1 || !PS;
#pragma unroll
            for (int m = 0; m < NBH; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * m + 4 * g + r) * SR + 16 * t + c;
                        const float sv = KEEPH ? hs[l - 1 < NHS ? l - 1 : 0][m < NHM ? m : 0][t < NHT ? t : 0][r] : Hp[ad];
                        const float d = dn[m][t][r] * eh_dact_row<ACT>(sv, l - 1, 16 * m + 4 * g + r);
                        dz[m][t][r] = d;
                        if (need_dz) Hp[ad] = d;
                    }
                    aB[l - 1][m] += dz[m][t];
                }
        }
        // layer 0: dW0 += dZ_0 * X^T
        if constexpr (PS) {
            float xv[4][NT];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int t = 0; t < NT; ++t) xv[pp][t] = XS[pp * SR + 16 * t + c];     // rows >= P are zero
#pragma unroll
            for (int m = 0; m < NBH; ++m)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) aW0V[m][pp][r] = fmaf(dz[m][t][r], xv[pp][t], aW0V[m][pp][r]);
        } else {
            EH_WAVE_SYNC();
            const float* const DZ = HS;
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
        }
        EH_WAVE_SYNC();
    }

    // (multi-step launch: the next step's records, straight into what the launch carries from step to step -- their indices were asked for
    //  when this step's records were consumed and have long arrived; the records have the reduction and the update to arrive behind)
    if (carry) {
        carry->valid = a.pf_count > 0;
        if (pf_issued) fetch_rec_into(carry->rec);
        else if (carry->valid) carry->rec = nx;                  // (a wave without a sample in this step has none in the next: its zeros)
        carry->live = nx_live;
    }
    EH_STAMP(8);
#ifdef EH_DBG_LACC
    if (a.stamps && blockIdx.x == 0 && lane < 2) reinterpret_cast<float*>(a.stamps)[wave * 2 + lane] = lacc;      // diagnostics: lanes 0 / 1 of every wave, before the wave sums
#endif
    // ---- 7. workgroup reduction -> one partial per workgroup -------------------------------------
    constexpr EhAccLayout AL = eh_acc_layout(NBI, NBH, NL, FAST);
    constexpr bool REDV2 = TRAIN && AL.rw <= G::WAVE_WS;
    if constexpr (REDV2) {
        // v2: every lane parks its raw accumulators with unconditional 16-byte stores; the sums over
        // the 16 samples of a row, over the waves and the padding removal all happen in the final
        // gather loop through the host-built rmap (fixed order: deterministic).
        float tailv[16];
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) tailv[j] = (j < net.n_par) ? eh_wave_sum(gacc[j]) * meta[EH_IMG_DPHI + j] : 0.0f;
        tailv[8] = eh_wave_sum(lacc);
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) tailv[9 + t] = (t < net.T) ? eh_wave_sum(cacc[t]) : 0.0f;
        tailv[13] = K1 ? eh_wave_sum(aBoS) : 0.0f;
        tailv[14] = eh_wave_sum(syacc); tailv[15] = eh_wave_sum(syyacc);
        EH_STAMP(13);
        __syncthreads();                           // the wave workspaces are dead from here on
        EH_STAMP(14);
        // waves that had no tile (a minibatch smaller than the workgroup's NW tiles: the reference's batch of 64 fills two of eight) hold zeros
        // everywhere: they park nothing and the gather leaves them out -- the same sums bit for bit (x + 0), a quarter of the LDS traffic
#ifdef EH_AB_FULL_GATHER       // (diagnostic A/B: the form of rounds 1-4 -- every wave parks, the gather sums all NW)
        constexpr int nlive = NW;
        constexpr bool parks = true;
#else
        const int nlive_ = ntiles - (int)blockIdx.x * NW, nlive = nlive_ < 0 ? 0 : (nlive_ < NW ? nlive_ : NW);
        const bool parks = wave < nlive;
#endif
        float* const R = smem + G::IMG_FLOATS + wave * AL.rw;
        // region[k][g][r][c]: the 16 samples (c) of a row are contiguous, so row sums are four 16-byte reads
        auto put = [&](int k, const f32x4& v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) R[k * 256 + g * 64 + r * 16 + c] = v[r];
        };
        if (parks) {
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
            if constexpr (PS) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) put(AL.kw0 + m * 4 + pp, aW0V[m][pp]);
            } else {
#pragma unroll
                for (int n = 0; n < NBI; ++n) put(AL.kw0 + m * NBI + n, aW0[m][n]);
            }
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                for (int n = 0; n < NBH; ++n) put(AL.kwh + (l * NBH + m) * NBH + n, aWh[l][m][n]);
            if constexpr (K1) put(AL.kwo + m, aWoV[m]); else put(AL.kwo + m, aWo[m]);
#pragma unroll
            for (int l = 0; l < NL; ++l) put(AL.kb + l * NBH + m, aB[l][m]);
        }
        if constexpr (!K1) put(AL.kbo, aBo);
        if (lane < 16) {
            float tv = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tv = (lane == k) ? tailv[k] : tv;
            R[AL.na * 256 + lane] = tv;
        }
        }       // (parks)
        EH_STAMP(12);
        __syncthreads();
        EH_STAMP(9);
        const float* const R0 = smem + G::IMG_FLOATS;
        float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
        float* const gsh = a.fz.gacc ? (P2PM ? const_cast<float*>(px_stage) : a.fz.gacc) + (a.fz.gslot * EH_GSHARDS + (blockIdx.x & (EH_GSHARDS - 1))) * a.n_acc : nullptr;
        for (int e = tid; e < a.n_acc; e += NTHR) {
            const int code = e == tid ? f_rcode : a.rmap[e], pos = code & 0xFFFFFF, nlan = code >> 24;
            float sum = 0.0f;
            if (nlan == 16) {
                // (a run-time loop over the live waves -- predicated loads in an unrolled one became a branch around every load)
                if (nlive == NW) {
                    f32x4 v[NW][4];
#pragma unroll
                    for (int w = 0; w < NW; ++w)
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[w][i] = *(const f32x4*)&R0[w * AL.rw + pos + 4 * i];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const f32x4 q = (v[w][0] + v[w][1]) + (v[w][2] + v[w][3]);
                        sum += (q[0] + q[1]) + (q[2] + q[3]);
                    }
                } else {
#pragma unroll 1
                    for (int w = 0; w < nlive; ++w) {
                        const float* const Rw = R0 + w * AL.rw + pos;
                        const f32x4 v0 = *(const f32x4*)&Rw[0], v1 = *(const f32x4*)&Rw[4], v2 = *(const f32x4*)&Rw[8], v3 = *(const f32x4*)&Rw[12];
                        const f32x4 q = (v0 + v1) + (v2 + v3);
                        sum += (q[0] + q[1]) + (q[2] + q[3]);
                    }
                }
            } else {
#pragma unroll 1
                for (int w = 0; w < nlive; ++w) sum += R0[w * AL.rw + pos];
            }
            if (gsh) { if (a.ms_direct) ((eh_lds_f*)gsh)[e] = sum; else atomicAdd(&gsh[e], sum); }      // (ms_direct: one workgroup, one writer per element)
            else out[e] = sum;
        }
        if constexpr (P2PM) { if (px_mode == 0) eh_p2p_publish(&a.p2pv, a.fz.gslot, a.p2p_seq, a.n_acc, tid, NTHR); }      // (mode 1: the next kernel's workgroup 0 publishes)
        EH_STAMP_FINE(15);
        if constexpr (!P2PM) { if (a.ms_direct) eh_ms_apply<G>(net, a, gsh, wl, tid, NTHR); }
        EH_STAMP(10);
        return;
    }
    // v3 (shapes whose raw accumulators do not fit one wave workspace): as v2, but the waves park KH accumulators at a time --
    // region[kk][c][g][r], one 16-byte store per lane and accumulator, bank-conflict free -- and every canonical element is
    // gathered in the round that holds its accumulator.  (A canonical-order region per wave, "v1" below, costs sixteen-way
    // bank conflicts on every scattered store: consecutive lanes of a weight block are one output-width apart.)
    constexpr int KH = (G::WAVE_WS - 16) / 256;
    constexpr bool REDV3 = TRAIN && !REDV2 && KH >= 1;
    if constexpr (REDV3) {
        constexpr int NR = (AL.na + KH - 1) / KH;
        // The gather runs in ACCUMULATOR order: in the round that holds accumulators [rd KH, rd KH + KH) thread t sums word t of each of
        // them over the waves and stores it at its canonical index, which the host-built inverse map gives (a.rmap holds, for the shapes
        // of this branch, emap[k * 256 + word] = canonical index or -1, then 16 tail words: eh_api.hip build_maps).  (Walking the elements
        // in canonical order instead, every thread had to look at all of its ~20 elements in every one of the NR rounds -- 14 k of config
        // 3's 238 k cycles, tools/stamps_c3.py.)  Same additions in the same order as before: bit-identical sums.
        constexpr int NU = (KH * 256 + NTHR - 1) / NTHR;       // words of a round per thread: word (tid + u NTHR) of the round's KH x 256
        float tailv[16];
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) tailv[j] = (j < net.n_par) ? eh_wave_sum(gacc[j]) * meta[EH_IMG_DPHI + j] : 0.0f;
        tailv[8] = eh_wave_sum(lacc);
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) tailv[9 + t] = (t < net.T) ? eh_wave_sum(cacc[t]) : 0.0f;
        tailv[13] = K1 ? eh_wave_sum(aBoS) : 0.0f;
        tailv[14] = eh_wave_sum(syacc); tailv[15] = eh_wave_sum(syyacc);
        f32x4 vals[AL.na];
#pragma unroll
        for (int m = 0; m < NBH; ++m) {
            if constexpr (PS) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) vals[AL.kw0 + m * 4 + pp] = aW0V[m][pp];
            } else {
#pragma unroll
                for (int n = 0; n < NBI; ++n) vals[AL.kw0 + m * NBI + n] = aW0[m][n];
            }
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                for (int n = 0; n < NBH; ++n) vals[AL.kwh + (l * NBH + m) * NBH + n] = aWh[l][m][n];
            if constexpr (K1) vals[AL.kwo + m] = aWoV[m]; else vals[AL.kwo + m] = aWo[m];
#pragma unroll
            for (int l = 0; l < NL; ++l) vals[AL.kb + l * NBH + m] = aB[l][m];
        }
        if constexpr (!K1) vals[AL.kbo] = aBo;
        // accumulators that are per-lane partial sums over the 16 samples of a row (biases; the K1 / PS vectors) are summed across
        // the row here, so that the gather reads one word per wave for every kind of element
#pragma unroll
        for (int k = 0; k < AL.na; ++k) {
            const bool rowsum = k >= AL.kb || (K1 && k >= AL.kwo && k < AL.kb) || (PS && k < AL.kwh);
            if (rowsum) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vals[k][r] = eh_row16_sum(vals[k][r]);
            }
        }
        EH_STAMP(13);
        float* const R = smem + G::IMG_FLOATS + wave * G::WAVE_WS;
        const float* const R0 = smem + G::IMG_FLOATS;
        float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
        float* const gsh = a.fz.gacc ? (P2PM ? const_cast<float*>(px_stage) : a.fz.gacc) + (a.fz.gslot * EH_GSHARDS + (blockIdx.x & (EH_GSHARDS - 1))) * a.n_acc : nullptr;
#pragma unroll
        for (int rd = 0; rd < NR; ++rd) {
            __syncthreads();                       // the wave workspaces are dead / the previous round has been gathered
            if (rd == 0) EH_STAMP(14);
            int dst[NU], dtail = -1;               // (requested ahead of the parking stores and the barrier)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int wd = tid + u * NTHR;     // accumulator rd KH + wd / 256 of the kernel, its word wd % 256
                dst[u] = (wd < KH * 256 && rd * KH + wd / 256 < AL.na) ? a.rmap[rd * KH * 256 + wd] : -1;
            }
            if (rd == NR - 1 && tid < 16) dtail = a.rmap[AL.na * 256 + tid];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk)
                if (rd * KH + kk < AL.na) *(f32x4*)&R[kk * 256 + c * 16 + g * 4] = vals[rd * KH + kk];
            if (rd == NR - 1 && lane < 16) {
                float tv = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k) tv = (lane == k) ? tailv[k] : tv;
                R[KH * 256 + lane] = tv;
            }
            if (rd == 0) EH_STAMP(12);
            __syncthreads();
            if (rd == 0) EH_STAMP(9);
            {
                float sumv[NU], sumt = 0.0f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int wd = min(tid + u * NTHR, KH * 256 - 1);
                    float sum = 0.0f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += R0[w * G::WAVE_WS + wd];
                    sumv[u] = sum;
                }
                if (rd == NR - 1 && tid < 16) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) sumt += R0[w * G::WAVE_WS + KH * 256 + tid];
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (dst[u] >= 0) {
                        if (gsh) { if (a.ms_direct) ((eh_lds_f*)gsh)[dst[u]] = sumv[u]; else atomicAdd(&gsh[dst[u]], sumv[u]); }
                        else out[dst[u]] = sumv[u];
                    }
                if (dtail >= 0) {
                    if (gsh) { if (a.ms_direct) ((eh_lds_f*)gsh)[dtail] = sumt; else atomicAdd(&gsh[dtail], sumt); }
                    else out[dtail] = sumt;
                }
            }
        }
        if constexpr (P2PM) { if (px_mode == 0) eh_p2p_publish(&a.p2pv, a.fz.gslot, a.p2p_seq, a.n_acc, tid, NTHR); }      // (mode 1: the next kernel's workgroup 0 publishes)
        EH_STAMP_FINE(15);
        if constexpr (!P2PM) { if (a.ms_direct) eh_ms_apply<G>(net, a, gsh, wl, tid, NTHR); }
        EH_STAMP(10);
        return;
    }
    // forward / eval passes: the per-target metric sums, one small region per wave
    static_assert(!TRAIN || REDV2 || REDV3, "every training shape parks its accumulators the v2 / v3 way");
    EH_STAMP(13);
    __syncthreads();                               // the wave workspaces are dead from here on
    EH_STAMP(14);
    float* const RED = smem + G::IMG_FLOATS + wave * a.n_acc;
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t)
#pragma unroll
        for (int k = 0; k < EH_EVAL_STATS; ++k) {
            const float v = eh_wave_sum(est[t][k]);
            if (t < net.T && lane == 0) RED[t * EH_EVAL_STATS + k] = v;
        }
    EH_STAMP(12);
    __syncthreads();
    EH_STAMP(9);
    {
        const float* const R0 = smem + G::IMG_FLOATS;
        float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
        for (int e = tid; e < a.n_acc; e += NTHR) {
            float s = R0[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += R0[w * a.n_acc + e];
            out[e] = s;
        }
    }
    EH_STAMP(10);
}

// one training step (or one forward / eval pass) per launch
// (EH_SPEC_NS: a translation unit that bakes ONE model descriptor into its kernels ahead of time -- eh_spec.hip -- puts them in a
//  namespace of its own: the same template arguments name a different kernel there than in the generic translation units)
// The kernel arguments (some 760 bytes: twelve 64-byte lines) sit in memory the host wrote a moment ago: the first scalar load of
// each line misses every cache.  The step bodies read their arguments where they use them -- behind branches, one line after another,
// each miss a full memory round trip on the critical path of a 10 us kernel.  One load per line up front, all in flight together,
// and every later read of an argument is a scalar-cache hit.
template <int BYTES>
__device__ __forceinline__ void eh_kernarg_warm() {
#ifndef EH_NO_KERNARG_WARM
    // (written out: the compiler splits a loop of plain loads over several waits -- and every wait is one of those round trips)
    // BYTES = the explicit arguments as they are: only lines that START inside them are touched (advisor r05: rounding up to 256 read up
    // to 255 bytes past them -- inside the hidden-argument block today, an out-of-segment scalar read if the structs ever end near its end)
    static_assert(BYTES > 192 && BYTES <= 768, "four to twelve lines");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned d0, d1, d2, d3, d4 = 0u, d5 = 0u, d6 = 0u, d7 = 0u, d8 = 0u, d9 = 0u, d10 = 0u, d11 = 0u;
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
    if constexpr (BYTES > 0x100) asm volatile("s_load_dword %0, %1, 0x100" : "=&s"(d4) : "s"(ka));
    if constexpr (BYTES > 0x140) asm volatile("s_load_dword %0, %1, 0x140" : "=&s"(d5) : "s"(ka));
    if constexpr (BYTES > 0x180) asm volatile("s_load_dword %0, %1, 0x180" : "=&s"(d6) : "s"(ka));
    if constexpr (BYTES > 0x1c0) asm volatile("s_load_dword %0, %1, 0x1c0" : "=&s"(d7) : "s"(ka));
    if constexpr (BYTES > 0x200) asm volatile("s_load_dword %0, %1, 0x200" : "=&s"(d8) : "s"(ka));
    if constexpr (BYTES > 0x240) asm volatile("s_load_dword %0, %1, 0x240" : "=&s"(d9) : "s"(ka));
    if constexpr (BYTES > 0x280) asm volatile("s_load_dword %0, %1, 0x280" : "=&s"(d10) : "s"(ka));
    if constexpr (BYTES > 0x2c0) asm volatile("s_load_dword %0, %1, 0x2c0" : "=&s"(d11) : "s"(ka));
    // (the destinations stay live up to the wait: a register handed out earlier would be overwritten when its load lands)
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(d0), "s"(d1), "s"(d2), "s"(d3), "s"(d4), "s"(d5), "s"(d6), "s"(d7), "s"(d8), "s"(d9), "s"(d10), "s"(d11));
#endif
}
#ifdef EH_SPEC_NS
namespace EH_SPEC_NS {
#endif
template <int NBI, int NBH, int NL, int NT, int NW, int ACT, int MODE, int FAST>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void eh_step_kernel(const EhNet net, const EhStepArgs a) {
    eh_kernarg_warm<(int)(sizeof(EhNet) + sizeof(EhStepArgs))>();
    if constexpr (MODE == EH_MODE_TRAIN_MULTI) {
        // Minibatches that ONE workgroup covers -- the reference's default batch of 64 (src/config/TrainingConfig.jl:14) and everything up
        // to 16 NT NW samples: a step's only consumer is the same workgroup's next step, so the steps of an epoch need neither a kernel
        // boundary between them nor global memory for what one hands to the next.  One launch runs ms_nsteps fused-update steps with the
        // step-to-step state -- both parameter sets {theta, m, v}, the beta products, the three rotating gradient accumulators -- in LDS
        // behind the step body's own work space: the body is the single-step body, unchanged, with its state pointers redirected (its
        // float atomics become LDS atomics, its parameter re-load an LDS read); global memory sees the state again when the launch ends.
        // A first version kept the state in global memory with a release / acquire between two steps and was SLOWER than one launch per
        // step (10.3 against 7.4 us): every dependent round trip of a step is exposed inside one kernel.  (Staging a small data set and the
        // epoch's permutation in LDS as well was measured: 0.15-0.3 us of a 6 us step; not kept.)  Launched with ONE workgroup.
        extern __shared__ __attribute__((aligned(16))) float eh_ms_smem[];
        using G = EhGeom<NBI, NBH, NL, NT, NW>;
        constexpr int NTHR = 64 * NW;
#ifdef EH_SPEC_NET
        constexpr EhNet cnet = {EH_SPEC_NET};
        const int nth = cnet.n_theta, g_off = cnet.g_off;
#else
        const int nth = net.n_theta, g_off = net.g_off;
#endif
        const int np = 6 * nth + 4, ng = 3 * EH_GSHARDS * a.n_acc + 4;
        float* const l_pset = eh_ms_smem + G::TOTAL_FLOATS;
        float* const l_gacc = l_pset + ((np + 3) & ~3);
        int* const l_imap = reinterpret_cast<int*>(l_gacc + ((ng + 3) & ~3));      // canonical index -> image offset: read by every step's update (a global round trip per step otherwise)
        for (int i = threadIdx.x; i < np; i += NTHR) l_pset[i] = a.fz.pset[i];
        for (int i = threadIdx.x; i < ng; i += NTHR) l_gacc[i] = a.fz.gacc[i];
        for (int i = threadIdx.x; i < g_off; i += NTHR) l_imap[i] = a.fz.imap[i];
        __syncthreads();
        EhCarry<(G::IP + 3) / 4> carry;
        carry.valid = false; carry.live = false;
        for (int k = 0; k < a.ms_nsteps; ++k) {
            EhStepArgs b = a;
            b.first = a.first + (long long)k * a.ms_batch;
            const long long left = a.ms_end - b.first;
            b.count = left < (long long)a.ms_batch ? left : (long long)a.ms_batch;
            b.pf_first = b.first + a.ms_batch;
            b.pf_count = k + 1 < a.ms_nsteps ? (left - a.ms_batch < (long long)a.ms_batch ? left - a.ms_batch : (long long)a.ms_batch) : 0;
            if (b.pf_count < 0) b.pf_count = 0;
            b.fz.pset = l_pset; b.fz.gacc = l_gacc; b.fz.imap = l_imap;
            // every step applies its own update in its epilogue (eh_ms_apply): only step 0's prologue has something deferred to pick up (what was
            // pending before the launch) and flips the parameter set; the accumulator slot is used as a plain array and left zero
            b.fz.gslot = a.fz.gslot;
            b.fz.cur = a.fz.cur;
            b.fz.sc_sel = a.fz.sc_sel;
            b.fz.pending = k ? 0 : a.fz.pending;
            b.fz.loss_slot = a.fz.loss_slot;                                 // (step 0's prologue finishes the loss of the step pending before the launch)
            b.ms_loss = a.ms_loss ? a.ms_loss + k : nullptr;
            b.ms_keep = k > 0;
            b.ms_direct = 1;
            eh_step_body<NBI, NBH, NL, NT, NW, ACT, EH_MODE_TRAIN, FAST>(net, b, &carry);
            __syncthreads();
        }
        for (int i = threadIdx.x; i < a.n_acc; i += NTHR) l_gacc[(a.fz.gslot * EH_GSHARDS) * a.n_acc + i] = 0.0f;      // (the steps used shard 0 of this slot as a plain array)
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += NTHR) a.fz.pset[i] = l_pset[i];
        for (int i = threadIdx.x; i < ng; i += NTHR) a.fz.gacc[i] = l_gacc[i];
        // nothing is pending behind the launch, so no flush kernel will refresh the GLOBAL parameter image (what the forward / evaluation
        // kernels stage): the LDS image goes back as it stands -- except its normalisation block, which holds the last minibatch's statistics
        // here and the running ones there (written by the steps themselves)
        if (a.image_out)
            for (int i = threadIdx.x; i < G::IMG_FLOATS; i += NTHR) {
                const int q = i - (G::PHI_OFF + EH_IMG_BNM);
                if (q < 0 || q >= 64) a.image_out[i] = eh_ms_smem[i];
            }
    } else {
        eh_step_body<NBI, NBH, NL, NT, NW, ACT, MODE, FAST>(net, a);
    }
}
// LDS floats behind the step body's work space that the multi-step kernel keeps its state in (host side: launch size, eligibility)
__host__ __device__ inline long long eh_ms_extra_floats(int n_theta, int n_acc) { return (long long)((6 * n_theta + 4 + 3) & ~3) + ((3LL * EH_GSHARDS * n_acc + 4 + 3) & ~3LL) + n_theta + 4; }
#ifdef EH_SPEC_NS
}   // namespace EH_SPEC_NS
using namespace EH_SPEC_NS;
#endif

