// bf16-forward / fp32-accumulate variant of the row-split kernel (eh_wide.hpp): BASELINE.json configs[4]
// "MLP [32,128,128,6] ... bf16 fwd / fp32 accumulate".  NOT a mode of the reference (Float32 end to end,
// src/data/prepare_data.jl:58-60); its semantics are fixed by oracle/hybrid_oracle.py `precision = "bf16_fwd"`:
//
//   * every Dense product of the FORWARD pass takes its two operands -- weights and the layer's input (normalised predictors,
//     hidden activations) -- rounded to bfloat16 (nearest even) and accumulates in fp32 on v_mfma_f32_16x16x32_bf16
//     (16x the rate of the fp32 MFMA); biases, activations, sigma-scaling, mechanistic model and loss stay fp32;
//   * what a layer hands on IS the rounded activation, so the BACKWARD pass is the exact derivative of that function with
//     round() as the identity: dW = dZ * bf16(h)^T, dH = bf16(W)^T dZ, act' from the stored rounded activation -- in fp32.
//     Activations whose derivative needs the pre-activation (swish, per-net) are not built.
//
// The backward products run on the bf16 MFMA as well, WITHOUT giving up fp32: one operand of each (the stored activation, the
// weight) is exactly a bfloat16 already, and the other -- the fp32 delta dZ -- is split into three bfloat16 terms
// d = d1 + d2 + d3 (8 + 8 + 8 mantissa bits: the split is exact).  A bf16 x bf16 product is exact in fp32, so
// sum_p MFMA_bf16(d_p, h) accumulates exactly the products the fp32 MFMA would, at 3 x 16 cycles per 32 k-values instead of
// 8 x 32: a fifth of the matrix-pipe time, bit-for-bit the same operands.
//
// Because nothing but bf16 operands is ever needed, the LDS holds ONLY bf16 images, each stored once:
//   weights  [out feature][in feature]   (row stride k + 8 elements: 16-byte fragments, conflict-free)
//   images   [sample][feature]           activations; deltas as three planes
// A product that contracts over an image's COLUMN index (forward: features of the layer input; dH: features of dZ) reads
// 16-byte row fragments; one that contracts over its ROW index (dW: samples of dZ and of the activations; dH: out-features of
// W) reads the same image through gfx950's transposing LDS read (ds_read_b64_tr_b16, CDNA4 guide T10).  64-sample tiles
// (NT = 4).  Same slab / rmap contract as eh_wide_kernel (the accumulators are the same registers in the same order).
#pragma once
#include "eh_wide.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 eh_lds_s16x4;

__device__ __forceinline__ float eh_bf2f(__bf16 v) { return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, v) << 16); }

// d == a + b + c exactly (a, b, c bfloat16: 8 + 8 + 8 mantissa bits, round to nearest even each time; the remainders are exact in fp32)
__device__ __forceinline__ void eh_split3(float d, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)d;
    const float r1 = d - eh_bf2f(a);
    b = (__bf16)r1;
    c = (__bf16)(r1 - eh_bf2f(b));
}

// Fragment of a 16x16x32 bf16 MFMA operand whose k index runs over the ROWS of a [row][col] bf16 LDS image (row stride ld
// elements, a multiple of 4): element j of lane (c = lane & 15, g = lane >> 4) = image[row0 + 8 g + j][col0 + c] -- A[row c][k]
// or B[k][col c] alike.  Two hardware-transposed reads: lane 4q + p of a 16-lane group supplies the address of row q,
// columns 4p .. 4p+3 of the group's 4 x 16 block and receives column (lane & 15) of its four rows.  Every lane of the wave
// must be active and every address in bounds (rows row0 .. row0 + 31).
__device__ __forceinline__ bf16x8 eh_tr_frag(const __bf16* img, int ld, int row0, int col0, int lane) {
    const __bf16* const a0 = img + (row0 + 8 * (lane >> 4) + ((lane >> 2) & 3)) * ld + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((eh_lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((eh_lds_s16x4*)(a0 + 4 * ld));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// NS = bf16 terms a backward delta is carried in: 3 = exact ("bf16_fwd": bf16 forward, fp32-exact backward), 1 = rounded once
// ("bf16": bf16 operands in both passes, oracle precision = "bf16")
template <int NBI, int NBH, int NL, int NT, int NWV, int NS = 3>
struct EhBfGeom {
    using F = EhGeom<NBI, NBH, NL, NT, 1>;                 // the fp32 parameter image in global memory (what the optimiser kernel maintains)
    static_assert(NBH % NWV == 0, "the waves split the feature blocks evenly");
    static_assert(NT == 2 || NT == 4, "whole 32-sample k-steps in the weight-gradient products; one sample per lane of wave 0 in the mechanistic stage");
    static constexpr int MT = 16 * NT, SR = MT + 4, HP = 16 * NBH, IP = 16 * NBI;
    static constexpr int KP0 = 32 * ((IP + 31) / 32);      // k extent of layer 0 in whole MFMA steps (zero padded)
    static constexpr int S0B = KP0 + 8, SHB = HP + 8;      // bf16 row strides: 16-byte multiples, == 4 dwords mod 64 banks apart per row group
    static constexpr int NPART = NWV >= NT ? NWV / NT : 1; // split of the output layer's k-steps over the waves (wave w: sample block w % NT, k part w / NT)
    static_assert(NWV % NT == 0 || NWV < NT, "output-layer split");
    static_assert((HP / 32) % NPART == 0, "the output layer's k-steps divide over the wave parts");
    // LDS map, in floats (4-byte units); every offset a multiple of 4
    static constexpr int WB0_OFF = 0;                                       // bf16 [HP][S0B]
    static constexpr int WBH_OFF = WB0_OFF + HP * S0B / 2;                  // (NL-1) x bf16 [HP][SHB]
    static constexpr int WBO_OFF = WBH_OFF + (NL - 1) * HP * SHB / 2;       // bf16 [16][SHB]
    static constexpr int B_OFF = WBO_OFF + 16 * SHB / 2;                    // fp32 biases: NL * HP + 16
    static constexpr int PHI_OFF = B_OFF + NL * HP + 16;                    // fp32 EH_IMG_* block
    static constexpr int IMG_FLOATS = PHI_OFF + EH_IMG_META;
    static constexpr int XB_OFF = IMG_FLOATS;                               // bf16 [MT][S0B] normalised, rounded predictors
    static constexpr int HB_OFF = XB_OFF + MT * S0B / 2;                    // NL x bf16 [MT][SHB] rounded activations
    static constexpr int DOB = 16 + 8;                                      // row stride of the d loss / d NN output planes
    static_assert(NS == 1 || NS == 3, "delta terms");
    static constexpr int DZ_OFF = HB_OFF + NL * MT * SHB / 2;               // NS x bf16 [MT][SHB]: the terms of the current layer's delta; the fp32 split-K output partials [NPART][16][SR] alias it
    static constexpr int DZ_FLOATS = NS * MT * SHB / 2 > NPART * 16 * SR ? NS * MT * SHB / 2 : NPART * 16 * SR;      // (one plane of a 32-sample tile is smaller than the four output partials)
    static constexpr int DO_OFF = DZ_OFF + DZ_FLOATS;                       // NS x bf16 [MT][DOB]: the terms of d loss / d NN output
    static constexpr int OS_OFF = DO_OFF + NS * MT * DOB / 2;               // fp32 [16][SR] NN outputs -> physical parameters -> d loss / d output
    static constexpr int RS_OFF = OS_OFF + 16 * SR;                         // fp32 forcings (rows 0..3), targets (rows 4..7)
    static constexpr int SG_OFF = RS_OFF + (EH_MAX_FORC + EH_MAX_TARG) * SR; // fp32 [16][SR] d parameter / d output
    static constexpr int MAP_FLOATS = SG_OFF + 16 * SR;
    static constexpr int STAGE_FLOATS = NWV * eh_wide_layout(NBI, NBH, NL, NWV).na * 256;      // the end-of-kernel staging of the accumulators overlays the (then dead) map
    static constexpr int TOTAL_FLOATS = MAP_FLOATS > STAGE_FLOATS ? MAP_FLOATS : STAGE_FLOATS;
};

// sums of one wave's mechanistic stage (train: gradient of the global parameters, loss terms; eval: metric sums)
struct EhMechAcc {
    float gacc[EH_MAX_PARAMS], lacc, syacc, syyacc, cacc[EH_MAX_TARG], est[EH_MAX_TARG][EH_EVAL_STATS];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) gacc[j] = 0.0f;
        lacc = syacc = syyacc = 0.0f;
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) {
            cacc[t] = 0.0f;
#pragma unroll
            for (int k = 0; k < EH_EVAL_STATS; ++k) est[t][k] = 0.0f;
        }
    }
};

// Mechanistic model + masked loss + its pullback for ONE sample per lane (the caller's wave 0, lane = sample of the tile):
// physical parameters in OS[row k][lane] with d parameter / d NN output in SG, forcings / targets in RS -> (train) d loss / d NN
// output back into OS, sums into `acc`; (eval) predictions / parameters written out.  Same arithmetic as stage 5 of eh_wide_kernel.
template <bool TRAIN, bool PROG, bool LPROG = false, class NET>
__device__ __forceinline__ void eh_mech_stage_lane(const NET& net, const EhStepArgs& a, int lane, bool live, int n_loc, int SR, const float* RS,
                                                   float* OS, const float* SG, const float* meta, EhMechAcc& A) {
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    float par[EH_MAX_PARAMS], sg[EH_MAX_PARAMS], dydp[EH_MAX_PARAMS], frc[EH_MAX_FORC], yobs[EH_MAX_TARG];
#pragma unroll
    for (int f = 0; f < EH_MAX_FORC; ++f) {
        const unsigned col = (net.forc_col >> (8 * f)) & 0xFFu;
        frc[f] = col != 0xFFu ? RS[col * SR + lane] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) yobs[t] = t < net.T ? RS[(EH_MAX_FORC + t) * SR + lane] : __builtin_nanf("");
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) {
        par[j] = meta[EH_IMG_PHI + j]; sg[j] = 1.0f; dydp[j] = 0.0f;
        if (j < net.n_par && pkind(j) == EH_PAR_NEURAL) {
            par[j] = OS[pidx(j) * SR + lane];
            sg[j] = SG[pidx(j) * SR + lane];
        }
    }
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
        if (net.n_out > 1) eh_mech_extra(net.mech, par, frc, yx, Jx);
    }
    float dy = 0.0f, dyx[2] = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        if (t < net.T) {
            const int ot = (int)((net.targ_out >> (2 * t)) & 3u);
            const float y = ot == 0 ? y0 : (ot == 1 ? yx[0] : yx[1]);
            const bool valid = live && !__builtin_isnan(yobs[t]);
            const float r = valid ? y - yobs[t] : 0.0f;
            if constexpr (TRAIN) {
                const float* const tt = a.inv_n + EH_TT * t;
                const float w = a.inv_n ? tt[0] : 1.0f;
                const float cy = valid ? yobs[t] - a.shift[t] : 0.0f;
                float d;
                if (eh_target_mae(net.loss_t, t)) { A.lacc += w * fabsf(r); d = r > 0.0f ? w : (r < 0.0f ? -w : 0.0f); }
#ifdef EH_JIT_LOSS
                else if (eh_target_prog(net.loss_t, t)) {
                    float dl;
                    const float lv = eh_jit_loss(t, y, valid ? yobs[t] : y, dl);
                    A.lacc += valid ? w * lv : 0.0f;
                    d = valid ? w * dl : 0.0f;
                }
#else
                else if (LPROG && eh_target_prog(net.loss_t, t)) {
                    // a recorded loss function where no kernel is compiled around it (the layer-wise form): its tape interpreted, forward
                    // then the reverse sweep from d l / d l = 1 -- the arithmetic eh_jit_loss would have been generated from
                    if constexpr (LPROG) {
                        const unsigned* const lp = a.lprog + t * EH_LPROG_WORDS;
                        float lpar[EH_MAX_PARAMS], lfrc[EH_MAX_FORC], lval[EH_PROG_SLOTS], ladj[EH_PROG_SLOTS];
#pragma unroll
                        for (int j = 0; j < EH_MAX_PARAMS; ++j) lpar[j] = 0.0f;
#pragma unroll
                        for (int f = 0; f < EH_MAX_FORC; ++f) lfrc[f] = 0.0f;
                        lpar[0] = y; lpar[1] = valid ? yobs[t] : y;
                        eh_prog_forward(lp, lpar, lfrc, lval);
                        const int nslot = EH_PROG_SLOT_INSTR + (int)lp[0];
                        for (int i = 0; i < nslot; ++i) ladj[i] = 0.0f;
                        ladj[lp[2]] = 1.0f;
                        eh_prog_reverse(lp, lval, ladj);
                        A.lacc += valid ? w * lval[lp[2]] : 0.0f;
                        d = valid ? w * ladj[0] : 0.0f;
                    } else d = 0.0f;
                }
#endif
                else if (eh_target_two_pass(net.loss_t, t, net.T)) {      // two-pass losses (see eh_step_kernel)
                    d = valid ? fmaf(tt[6], cy, fmaf(tt[5], y - tt[1], tt[4])) : 0.0f;
                }
                else { A.lacc += w * r * r; d = 2.0f * w * r; }
                dy += ot == 0 ? d : 0.0f; dyx[0] += ot == 1 ? d : 0.0f; dyx[1] += ot == 2 ? d : 0.0f;
                A.cacc[t] += valid ? 1.0f : 0.0f;
                A.syacc += cy; A.syyacc += cy * cy;
            } else if (valid) {
                const float cy = yobs[t] - a.shift[t], ch = y - (a.inv_n ? a.inv_n[EH_TT * t + 1] : a.shift[t]);      // see eh_step_kernel
                A.est[t][0] += r * r; A.est[t][1] += cy; A.est[t][2] += cy * cy; A.est[t][3] += 1.0f;
                A.est[t][4] += ch; A.est[t][5] += ch * ch; A.est[t][6] += ch * cy; A.est[t][7] += fabsf(r);
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
            if (j < net.n_par) {
                float dp;
                if constexpr (PROG) dp = padj[j];
                else {
                    dp = dy * dydp[j];
                    if (j < 3) dp += dyx[0] * Jx[0][j] + dyx[1] * Jx[1][j];
                }
                dp = live ? dp : 0.0f;
                const int kd = pkind(j);
                if (kd == EH_PAR_NEURAL) OS[pidx(j) * SR + lane] = dp * sg[j];
                else if (kd == EH_PAR_GLOBAL) A.gacc[j] += dp;
            }
        }
    }
}

// (EH_SPEC_NS: a translation unit that bakes ONE model descriptor into its kernels ahead of time -- eh_spec.hip -- puts them in a
//  namespace of its own: the same template arguments name a different kernel there than in the generic translation units)
#ifdef EH_SPEC_NS
namespace EH_SPEC_NS {
#endif
template <int NBI, int NBH, int NL, int NT, int NWV, int ACT, int MODE, bool PROG = false, int NS = 3>
__global__ __launch_bounds__(64 * NWV, 1) void eh_widebf_kernel(const EhNet net_rt, const EhStepArgs a) {
    eh_kernarg_warm<(int)(sizeof(EhNet) + sizeof(EhStepArgs))>();
#ifdef EH_SPEC_NET
    constexpr EhNet net = {EH_SPEC_NET};        // see eh_step_body
#else
    const EhNet& net = net_rt;
#endif
    static_assert(!EhStoresZ<ACT>::value, "the bf16-forward kernel keeps only the rounded activation");
    using G = EhBfGeom<NBI, NBH, NL, NT, NWV, NS>;
    using F = typename G::F;
    constexpr int MT = G::MT, SR = G::SR, HP = G::HP, IP = G::IP, KP0 = G::KP0, S0B = G::S0B, SHB = G::SHB, MB = NBH / NWV, NTH = 64 * NWV;
    constexpr int NPART = G::NPART, KSH = HP / 32, KS0 = KP0 / 32;
    constexpr bool TRAIN = MODE == EH_MODE_TRAIN;
    constexpr EhWideLayout WL = eh_wide_layout(NBI, NBH, NL, NWV);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    // (the wave index as a SCALAR: derived from threadIdx it counts as divergent, and every loop / branch on it -- the tile loop
    //  first of all -- would run under an exec mask with saved / restored mask pairs instead of scalar branches)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __bf16* const WB0 = reinterpret_cast<__bf16*>(smem + G::WB0_OFF);
    __bf16* const WBH = reinterpret_cast<__bf16*>(smem + G::WBH_OFF);
    __bf16* const WBO = reinterpret_cast<__bf16*>(smem + G::WBO_OFF);
    float* const BIAS = smem + G::B_OFF;
    __bf16* const XB = reinterpret_cast<__bf16*>(smem + G::XB_OFF);
    __bf16* const HB = reinterpret_cast<__bf16*>(smem + G::HB_OFF);
    __bf16* const DZP = reinterpret_cast<__bf16*>(smem + G::DZ_OFF);      // [NS][MT][SHB]
    __bf16* const DOP = reinterpret_cast<__bf16*>(smem + G::DO_OFF);      // [NS][MT][DOB]
    float* const OSP = smem + G::DZ_OFF;        // [NPART][16][SR] partial outputs of the split-K output layer (before the deltas exist)
    float* const OS = smem + G::OS_OFF;
    float* const RS = smem + G::RS_OFF;
    float* const SG = smem + G::SG_OFF;
    const float* const meta = smem + G::PHI_OFF;
    const int m0 = wave * MB;
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    const bool mechw = wave == 0 && lane < MT;   // the lanes that own one sample each in the mechanistic stage

    // ---- records: the tile's MT records are MT*C consecutive floats (or MT gathered runs of C); thread-owned elements, loaded
    // coalesced one tile ahead, predictors normalised + rounded into the [sample][feature] bf16 image, forcings / targets into RS
    constexpr int NEL = (MT * (IP + EH_MAX_FORC + EH_MAX_TARG) + NTH - 1) / NTH;
    const int count = (int)a.count, first = (int)a.first, C = a.C;
    const int ntiles = (count + MT - 1) / MT;
    int epk[NEL], nidx[NEL];          // element k: destination (bf16 index into XB for predictors, float index into RS otherwise) | column << 16 | sample << 24 ; -1 = none
    float nx[NEL];
#pragma unroll
    for (int k = 0; k < NEL; ++k) {
        const int e = tid + k * NTH;
        epk[k] = -1; nidx[k] = 0; nx[k] = 0.0f;
        if (e < MT * C) {
            const int smp = e / C, col = e - smp * C;
            const int row = col < net.P ? -1 : (col - net.P < net.F ? col - net.P : EH_MAX_FORC + (col - net.P - net.F));
            const int dst = row < 0 ? smp * S0B + col : row * SR + smp;
            epk[k] = dst | (col << 16) | (smp << 24);
        }
    }
    // (branch-free on purpose: a dead element -- beyond the tile's C columns, the window's end or the last tile -- reads word 0 of the
    //  window, which exists whenever count > 0, and is replaced by a select; with the loads under per-element conditions the compiler
    //  built five exec-masked regions with a memory wait each: 1.6 k cycles per tile for this phase, measured with the stamps)
    auto fetch_idx = [&](int tile) {
        if (!a.idx) return;
#pragma unroll
        for (int k = 0; k < NEL; ++k) {
            const int s_loc = tile * MT + (epk[k] >> 24);
            const bool on = epk[k] >= 0 && tile < ntiles && s_loc < count;
            const int v = a.idx[first + (on ? s_loc : 0)];
            nidx[k] = on ? v : 0;
        }
    };
    auto fetch = [&](int tile) {      // data of `tile` (its gather indices are already in nidx), then the indices one tile further
        // (a dead element's word is replaced where it is consumed, one tile later: a select right here and the compiler turns it back
        //  into a branch around the load)
        if (!a.idx) {         // a window of consecutive records: one scalar base per tile, the element's constant offset beside it
            const bool in = tile < ntiles;
            const float* const tb = a.recs + (long long)(first + (in ? tile : 0) * MT) * C;
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const bool live = epk[k] >= 0 && in && tile * MT + (epk[k] >> 24) < count;
                nx[k] = tb[live ? tid + k * NTH : 0];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const int col = (epk[k] >> 16) & 0xFF, s_loc = tile * MT + (epk[k] >> 24);
                const bool live = epk[k] >= 0 && tile < ntiles && s_loc < count;
                nx[k] = a.recs[live ? (long long)nidx[k] * C + col : 0LL];
            }
        }
        fetch_idx(tile + (int)gridDim.x);
    };
    EH_STAMP(13);
    fetch_idx((int)blockIdx.x);
    fetch((int)blockIdx.x);

    // ---- parameter image (fp32, EhGeom layout, kept by the optimiser kernel) -> bf16 weights, fp32 biases + meta block
    {
        // every 16-byte piece of the three weight matrices is requested before the first one is converted: the whole image
        // arrives in one memory round trip (rows of the fp32 image are 16-byte multiples: strides IP + 4 and HP + 4 floats)
        constexpr int N0 = (HP * KP0 / 4 + NTH - 1) / NTH, NH = (HP * HP / 4 + NTH - 1) / NTH, NO = (16 * HP / 4 + NTH - 1) / NTH;
        f32x4 r0[N0], rh[NL > 1 ? NL - 1 : 1][NH], ro[NO];
        auto ld_rows = [&](const float* src, int sld, int scols, int rows, int kcols, f32x4* v, int n) {
            const int qp = kcols / 4, tot = rows * qp;
            for (int u = 0; u < n; ++u) {
                const int i = tid + u * NTH, row = i / qp, col = 4 * (i - row * qp);
                v[u] = (i < tot && col < scols) ? *(const f32x4*)&src[row * sld + col] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        auto st_rows = [&](__bf16* dst, int dld, int rows, int kcols, const f32x4* v, int n) {
            const int qp = kcols / 4, tot = rows * qp;
            for (int u = 0; u < n; ++u) {
                const int i = tid + u * NTH, row = i / qp, col = 4 * (i - row * qp);
                if (i < tot) {
                    *(bf16x2*)&dst[row * dld + col] = bf16x2{(__bf16)v[u][0], (__bf16)v[u][1]};
                    *(bf16x2*)&dst[row * dld + col + 2] = bf16x2{(__bf16)v[u][2], (__bf16)v[u][3]};
                }
            }
        };
#pragma unroll
        for (int rep = 0; rep < 1; ++rep) {
            ld_rows(a.image + F::W0_OFF, F::S0, IP, HP, KP0, r0, N0);
#pragma unroll
            for (int l = 1; l < NL; ++l) ld_rows(a.image + F::WH_OFF + (l - 1) * HP * F::SH, F::SH, HP, HP, HP, rh[l - 1], NH);
            ld_rows(a.image + F::WO_OFF, F::SH, HP, 16, HP, ro, NO);
        }
        st_rows(WB0, S0B, HP, KP0, r0, N0);
#pragma unroll
        for (int l = 1; l < NL; ++l) st_rows(WBH + (l - 1) * HP * SHB, SHB, HP, HP, rh[l - 1], NH);
        st_rows(WBO, SHB, 16, HP, ro, NO);
        for (int e = tid; e < NL * HP + 16 + EH_IMG_META; e += NTH) smem[G::B_OFF + e] = a.image[F::B_OFF + e];      // (B_OFF .. PHI_OFF + META is one run in both layouts)
    }
    for (int e = tid; e < MT * S0B / 2; e += NTH) smem[G::XB_OFF + e] = 0.0f;     // columns >= P of the predictor image stay 0
    for (int e = tid; e < 16 * SR; e += NTH) OS[e] = 0.0f;                        // rows >= K of the output / dO image stay 0
    __syncthreads();
    if (a.bn_part) {       // input BatchNorm, train mode: statistics of this minibatch (see eh_step_kernel)
        if (tid < net.P) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const float m = a.bn_n ? *a.bn_n : (float)count, c0 = a.bn_c[tid];
            const float d = s1 / m, var = fmaxf(s2 / m - d * d, 0.0f), mu = c0 + d;
            smem[G::PHI_OFF + EH_IMG_BNM + tid] = mu;
            smem[G::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(var + EH_BN_EPS);
            if (a.bn_update && blockIdx.x == 0) {
                const float rm = (1.0f - EH_BN_MOMENTUM) * a.bn_run[tid] + EH_BN_MOMENTUM * mu;
                const float rv = (1.0f - EH_BN_MOMENTUM) * a.bn_run[32 + tid] + EH_BN_MOMENTUM * (m > 1.0f ? m / (m - 1.0f) : 1.0f) * var;
                a.bn_run[tid] = rm; a.bn_run[32 + tid] = rv;
                a.image_out[F::PHI_OFF + EH_IMG_BNM + tid] = rm;
                a.image_out[F::PHI_OFF + EH_IMG_BNR + tid] = 1.0f / sqrtf(rv + EH_BN_EPS);
            }
        }
        __syncthreads();
    }

    // split-K partials -> physical parameters: element (k, sample) = tid + u*NTH, its bounds fixed for the whole launch
    constexpr int NEO = (16 * MT + NTH - 1) / NTH;
    float klo[NEO], ksc[NEO];
#pragma unroll
    for (int u = 0; u < NEO; ++u) {
        const int k = (tid + u * NTH) / MT;
        klo[u] = 0.0f; ksc[u] = 0.0f;
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j)
            if (j < net.n_par && pkind(j) == EH_PAR_NEURAL && pidx(j) == k) { klo[u] = meta[EH_IMG_LO + j]; ksc[u] = meta[EH_IMG_SC + j]; }
    }

    // accumulators: this wave's row slice of every weight gradient
    f32x4 aW0[MB][NBI], aWh[NL > 1 ? NL - 1 : 1][MB][NBH], aWo[MB], aB[NL][MB], aBo = f32x4{0, 0, 0, 0};
    EhMechAcc MA;
    MA.clear();
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
    const int ksK = net.K < 4 ? net.K : 4;
    // the output layer's k-steps split over the waves: wave w contracts k part w / NT for sample block w % NT
    const int o_t = wave % NT, o_part = wave / NT;
    const bool o_on = wave < NT * NPART;

    // one forward layer: this wave's output blocks = act(W[own rows] * in + b), rounded, into the [sample][feature] image
    auto forward_layer = [&](const __bf16* W, int ldw, int ksteps, const __bf16* in, int ldin, int layer, __bf16* out) {
        f32x4 acc[MB][NT];
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const f32x4 bias = *(const f32x4*)&BIAS[layer * HP + 16 * (m0 + mm) + 4 * g];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mm][t] = bias;
        }
        for (int kk = 0; kk < ksteps; ++kk) {
            bf16x8 bq[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bq[t] = *(const bf16x8*)&in[(16 * t + c) * ldin + 32 * kk + 8 * g];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                const bf16x8 a8 = *(const bf16x8*)&W[(16 * (m0 + mm) + c) * ldw + 32 * kk + 8 * g];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, bq[t], acc[mm][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 h = eh_act4_rows<ACT>(acc[mm][t], layer, 16 * (m0 + mm) + 4 * g);
                *(bf16x4*)&out[(16 * t + c) * SHB + 16 * (m0 + mm) + 4 * g] = bf16x4{(__bf16)h[0], (__bf16)h[1], (__bf16)h[2], (__bf16)h[3]};
            }
    };
    constexpr int DOB = G::DOB, PZ = MT * SHB, PO = MT * DOB;      // plane sizes (elements) of the delta / output-delta terms
    // a C/D block of deltas (features 16m+4g .. +3 of sample 16t+c per lane), split into its three bf16 terms, into the delta planes
    auto store_dz = [&](const f32x4& d, int m, int t) {
        __bf16* const q = DZP + (16 * t + c) * SHB + 16 * m + 4 * g;
        if constexpr (NS == 1) {
            *(bf16x4*)q = bf16x4{(__bf16)d[0], (__bf16)d[1], (__bf16)d[2], (__bf16)d[3]};
        } else {
            bf16x4 p0, p1, p2;
#pragma unroll
            for (int r = 0; r < 4; ++r) { __bf16 x, y, z; eh_split3(d[r], x, y, z); p0[r] = x; p1[r] = y; p2[r] = z; }
            *(bf16x4*)q = p0; *(bf16x4*)(q + PZ) = p1; *(bf16x4*)(q + 2 * PZ) = p2;
        }
    };
    // what a delta block adds to its bias gradient: itself (exact mode), its once-rounded value ("bf16": the operand of the products)
    auto bias_term = [&](const f32x4& d) {
        if constexpr (NS == 1) return f32x4{eh_bf2f((__bf16)d[0]), eh_bf2f((__bf16)d[1]), eh_bf2f((__bf16)d[2]), eh_bf2f((__bf16)d[3])};
        else return d;
    };
    // stored (rounded) activations of features 16m+4g .. +3 of sample 16t+c
    auto load_h4 = [&](const __bf16* img, int m, int t) {
        const bf16x4 p = *(const bf16x4*)&img[(16 * t + c) * SHB + 16 * m + 4 * g];
        return f32x4{eh_bf2f(p[0]), eh_bf2f(p[1]), eh_bf2f(p[2]), eh_bf2f(p[3])};
    };

    EH_STAMP(14);
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        EH_STAMP(0);
        const int n_loc = tile * MT + lane;
        const bool live = mechw && (n_loc < count);
        // ---- 1. records -> normalised, rounded predictor image + forcing / target rows; next tile's records in flight
        {
            // every element stores twice, once for real and once into a padding word no one reads (XB row 0 column KP0; RS row 0 column MT):
            // straight-line code, the normalisation constants of all elements read from the image in one LDS round trip
            float bm[NEL], br[NEL];
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const int col = (epk[k] >> 16) & 0xFF;
                const int cm = (epk[k] >= 0 && col < net.P) ? col : 0;
                bm[k] = meta[EH_IMG_BNM + cm]; br[k] = meta[EH_IMG_BNR + cm];
            }
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const int col = (epk[k] >> 16) & 0xFF, dst = epk[k] & 0xFFFF;
                const bool isx = epk[k] >= 0 && col < net.P, isr = epk[k] >= 0 && col >= net.P;
                const float v = tile * MT + (epk[k] >> 24) < count ? nx[k] : (col >= net.P + net.F ? __builtin_nanf("") : 0.0f);      // beyond the window's end: no sample
                XB[isx ? dst : KP0] = (__bf16)((v - bm[k]) * br[k]);
                RS[isr ? dst : MT] = v;
            }
        }
        fetch(tile + (int)gridDim.x);
        eh_lds_barrier();
        EH_STAMP(1);
        // ---- 2. / 3. forward: layer 0, hidden layers ----------------------------------------------
        forward_layer(WB0, S0B, KS0, XB, S0B, 0, HB);
        eh_lds_barrier();
        EH_STAMP(2);
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            forward_layer(WBH + (l - 1) * HP * SHB, SHB, KSH, HB + (l - 1) * MT * SHB, SHB, l, HB + l * MT * SHB);
            eh_lds_barrier();
        }
        EH_STAMP(3);
        const __bf16* const Hlast = HB + (NL - 1) * MT * SHB;
        // ---- 4. output layer (K <= 16 rows), k-steps split over the waves --------------------------
        if (o_on) {
            const f32x4 bias = *(const f32x4*)&BIAS[NL * HP + 4 * g];
            f32x4 o = o_part == 0 ? bias : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kq = 0; kq < KSH / NPART; ++kq) {
                const int kk = o_part * (KSH / NPART) + kq;
                const bf16x8 a8 = *(const bf16x8*)&WBO[c * SHB + 32 * kk + 8 * g];
                const bf16x8 b8 = *(const bf16x8*)&Hlast[(16 * o_t + c) * SHB + 32 * kk + 8 * g];
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, o, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) OSP[(o_part * 16 + 4 * g + r) * SR + 16 * o_t + c] = o[r];
        }
        eh_lds_barrier();
        // ---- 4b. all threads: sum the partials, sigmoid-scale into the parameter range (GenericHybridModel.jl:348-352)
#pragma unroll
        for (int u = 0; u < NEO; ++u) {
            const int e = tid + u * NTH, k = e / MT, smp = e % MT;
            if (k < net.K && (NEO * NTH == 16 * MT || e < 16 * MT)) {
                float ov = 0.0f;
#pragma unroll
                for (int w = 0; w < NPART; ++w) ov += OSP[(16 * w + k) * SR + smp];
                float pv = ov, sv = 1.0f;
                if (net.scale_nn) {
                    const float sgm = eh_sigmoid(ov);
                    pv = fmaf(ksc[u], sgm, klo[u]);
                    sv = ksc[u] * sgm * (1.0f - sgm);
                }
                OS[k * SR + smp] = pv;
                SG[k * SR + smp] = sv;
            }
        }
        eh_lds_barrier();
        EH_STAMP(4);
        // ---- 5. mechanistic model + masked loss: wave 0, one sample per lane -----------------------
        if (mechw) eh_mech_stage_lane<TRAIN, PROG>(net, a, lane, live, n_loc, SR, RS, OS, SG, meta, MA);
        eh_lds_barrier();
        EH_STAMP(5);
        if constexpr (!TRAIN) continue;

        // ---- 6. backward through the output layer: d loss / d NN output (fp32 in OS) -> three bf16 terms [sample][k] -------
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 dO;
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[r] = OS[(4 * g + r) * SR + 16 * t + c];
                if constexpr (NS == 1) {          // "bf16": the bias gradient sums the once-rounded delta, the operand of the products (oracle _backprop)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dO[r] = eh_bf2f((__bf16)dO[r]);
                }
                aBo += dO;
            }
        }
#pragma unroll
        for (int u = 0; u < NEO; ++u) {
            const int e = tid + u * NTH, k = e & 15, smp = e >> 4;
            if (NEO * NTH == 16 * MT || e < 16 * MT) {
                if constexpr (NS == 1) DOP[smp * DOB + k] = (__bf16)OS[k * SR + smp];
                else {
                    __bf16 x, y, z;
                    eh_split3(OS[k * SR + smp], x, y, z);
                    DOP[smp * DOB + k] = x; DOP[PO + smp * DOB + k] = y; DOP[2 * PO + smp * DOB + k] = z;
                }
            }
        }
        eh_lds_barrier();
        f32x4 dzr[MB][NT];
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const int m = m0 + mm;
            // dWo[k-out][own features] += dO * bf16(H_last)^T : both operands contract over the samples (rows of their images)
#pragma unroll
            for (int kk = 0; kk < MT / 32; ++kk) {
                const bf16x8 bfr = eh_tr_frag(Hlast, SHB, 32 * kk, 16 * m, lane);
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    aWo[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh_tr_frag(DOP + p * PO, DOB, 32 * kk, 0, lane), bfr, aWo[mm], 0, 0, 0);
            }
            // dH_last[own features][samples] = bf16(Wo)^T dO : k = the 16 (padded) output rows, the upper half of the k-step is zero
            const bf16x8 zero8 = __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
            bf16x8 afr = eh_tr_frag(WBO, SHB, -8 * (g & 2), 16 * m, lane);      // (lanes g >= 2 re-read rows 0..15: in bounds, discarded)
            afr = g < 2 ? afr : zero8;
            f32x4 dh[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) dh[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < NS; ++p)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bf16x8 bfr = *(const bf16x8*)&DOP[p * PO + (16 * t + c) * DOB + 8 * (g & 1)];
                    bfr = g < 2 ? bfr : zero8;
                    dh[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, dh[t], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 hv = load_h4(Hlast, m, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) dzr[mm][t][r] = dh[t][r] * eh_dact_row<ACT>(hv[r], NL - 1, 16 * m + 4 * g + r);
                aB[NL - 1][mm] += bias_term(dzr[mm][t]);
            }
        }
        // the split-K partials (aliasing the delta planes) were last read in step 4b: safe to overwrite now
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
#pragma unroll
            for (int t = 0; t < NT; ++t) store_dz(dzr[mm][t], m0 + mm, t);
        eh_lds_barrier();
        EH_STAMP(6);
        // ---- 7. hidden layers backward -------------------------------------------------------------
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            const __bf16* Hp = HB + (l - 1) * MT * SHB;
            const __bf16* W = WBH + (l - 1) * HP * SHB;
            // dW_l[own rows][all columns] += dZ_l (own rows) * bf16(H_{l-1})^T : contraction over the samples, both fragments transposed reads
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                constexpr int NG = NBH < 4 ? NBH : 4;           // column blocks in flight: independent accumulators
#pragma unroll
                for (int kk = 0; kk < MT / 32; ++kk) {
                    bf16x8 afr[NS];
#pragma unroll
                    for (int p = 0; p < NS; ++p) afr[p] = eh_tr_frag(DZP + p * PZ, SHB, 32 * kk, 16 * (m0 + mm), lane);
#pragma unroll
                    for (int n0 = 0; n0 < NBH; n0 += NG) {
                        bf16x8 bfr[NG];
#pragma unroll
                        for (int u = 0; u < NG; ++u) bfr[u] = eh_tr_frag(Hp, SHB, 32 * kk, 16 * (n0 + u), lane);
#pragma unroll
                        for (int p = 0; p < NS; ++p)
#pragma unroll
                            for (int u = 0; u < NG; ++u)
                                aWh[l - 1][mm][n0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[p], bfr[u], aWh[l - 1][mm][n0 + u], 0, 0, 0);
                    }
                }
            }
            // dH_{l-1}[own rows][samples] = bf16(W_l)^T dZ_l : contraction over the rows of W_l (transposed read) and the columns of dZ_l (row fragments)
            f32x4 dn[MB][NT];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) dn[mm][t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < KSH; ++kk) {
                bf16x8 afr[MB];
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) afr[mm] = eh_tr_frag(W, SHB, 32 * kk, 16 * (m0 + mm), lane);
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    bf16x8 bfr[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bfr[t] = *(const bf16x8*)&DZP[p * PZ + (16 * t + c) * SHB + 32 * kk + 8 * g];
#pragma unroll
                    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                        for (int t = 0; t < NT; ++t) dn[mm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[mm], bfr[t], dn[mm][t], 0, 0, 0);
                }
            }
            EH_STAMP(7);
            eh_lds_barrier();                         // every wave is done reading dZ_l
            EH_STAMP(8);
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 hv = load_h4(Hp, m0 + mm, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dzr[mm][t][r] = dn[mm][t][r] * eh_dact_row<ACT>(hv[r], l - 1, 16 * (m0 + mm) + 4 * g + r);
                    store_dz(dzr[mm][t], m0 + mm, t);
                    aB[l - 1][mm] += bias_term(dzr[mm][t]);
                }
            eh_lds_barrier();
        }
        EH_STAMP(9);
        // ---- 8. layer 0: dW0[own rows] += dZ_0 * bf16(X)^T -------------------------------------------
#pragma unroll
        for (int mm = 0; mm < MB; ++mm)
#pragma unroll
            for (int kk = 0; kk < MT / 32; ++kk) {
                bf16x8 bfr[NBI];
#pragma unroll
                for (int n = 0; n < NBI; ++n) bfr[n] = eh_tr_frag(XB, S0B, 32 * kk, 16 * n, lane);
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    const bf16x8 afr = eh_tr_frag(DZP + p * PZ, SHB, 32 * kk, 16 * (m0 + mm), lane);
#pragma unroll
                    for (int n = 0; n < NBI; ++n) aW0[mm][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[n], aW0[mm][n], 0, 0, 0);
                }
            }
        eh_lds_barrier();                             // the images are rewritten by the next tile
        EH_STAMP(10);
    }
    EH_STAMP(11);

    // ---- 9. one partial per workgroup (as eh_wide_kernel: the waves own disjoint entries) ------------
    float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
    if constexpr (!TRAIN) {
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t)
#pragma unroll
                for (int k = 0; k < EH_EVAL_STATS; ++k) {
                    const float v = eh_wave_sum(MA.est[t][k]);
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
    float gs[EH_MAX_PARAMS];
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) gs[j] = meta[EH_IMG_DPHI + j];
    if (a.rmap == nullptr) {
        // ONE network (SingleNN), so the canonical order is plain: layer l's weights column-major (out, in) at w_off[l], its bias behind
        // them (image meta block).  A C/D accumulator holds rows 4g .. 4g+3 of column c of a 16 x 16 block -- four CONSECUTIVE canonical
        // entries -- so every lane stores its registers straight to the slab row, 16 bytes at a time: no staging in LDS, no barrier, no
        // map.  (The staged form below -- 86 KB of map read and 86 KB written through the one 64 B/clk path of the CU, a dependent chain
        // of load, LDS read and store -- took 9.4 k cycles of a launch on config 5; round 4.)
        const int* const im = reinterpret_cast<const int*>(meta);
        int Wd[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) Wd[l] = im[EH_IMG_WIDTH + l];
#pragma unroll
        for (int mm = 0; mm < MB; ++mm) {
            const int row0 = 16 * (m0 + mm) + 4 * g;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int wo = im[EH_IMG_WOFF + l], bo = im[EH_IMG_BOFF + l], nrow = Wd[l] - row0, ncol = l == 0 ? net.P : Wd[l - 1];
                if (nrow > 0) {
                    if (l == 0) {
#pragma unroll
                        for (int n = 0; n < NBI; ++n)
                            if (16 * n + c < ncol) eh_store_upto4(out + wo + (16 * n + c) * Wd[0] + row0, aW0[mm][n], nrow);
                    } else {
#pragma unroll
                        for (int n = 0; n < NBH; ++n)
                            if (16 * n + c < ncol) eh_store_upto4(out + wo + (16 * n + c) * Wd[l] + row0, aWh[l > 0 ? l - 1 : 0][mm][n], nrow);
                    }
                    if (c == 0) eh_store_upto4(out + bo + row0, aB[l][mm], nrow);
                }
            }
            // output layer: rows = the K outputs (4g .. 4g+3 of this lane), columns = this wave's features of the last hidden layer
            const int col = 16 * (m0 + mm) + c, nk = net.K - 4 * g;
            if (col < Wd[NL - 1] && nk > 0) eh_store_upto4(out + im[EH_IMG_WOFF + NL] + col * net.K + 4 * g, aWo[mm], nk);
        }
        if (wave == 0 && c == 0 && net.K - 4 * g > 0) eh_store_upto4(out + im[EH_IMG_BOFF + NL] + 4 * g, aBo, net.K - 4 * g);
    } else {
    __syncthreads();
    float* const st = smem + (long long)wave * WL.na * 256 + lane * 4;
    auto putc = [&](int k, const f32x4& v) { *(f32x4*)&st[k * 256] = v; };
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
    putc(WL.kbo, aBo);                                // only wave 0's copy is referenced
    __syncthreads();
    constexpr int GU = 48;                            // independent map loads in flight per thread (each is an L2 round trip; the accumulators are dead by now: registers to spare.  16 -> 48: 0.5 us of a config 5 step)
    for (int i0 = tid; i0 < net.g_off; i0 += GU * NTH) {
        int pos[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) pos[u] = (i0 + u * NTH < net.g_off) ? a.rmap[i0 + u * NTH] : 0;
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (i0 + u * NTH < net.g_off) out[i0 + u * NTH] = smem[pos[u]];
    }
    }       // (staged form)
    if (wave == 0) {
        const float lacc = eh_wave_sum(MA.lacc), syacc = eh_wave_sum(MA.syacc), syyacc = eh_wave_sum(MA.syyacc);
        float cacc[EH_MAX_TARG], gacc[EH_MAX_PARAMS];
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) cacc[t] = t < net.T ? eh_wave_sum(MA.cacc[t]) : 0.0f;
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) gacc[j] = j < net.n_par ? eh_wave_sum(MA.gacc[j]) * gs[j] : 0.0f;
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
    EH_STAMP(12);
}
#ifdef EH_SPEC_NS
}   // namespace EH_SPEC_NS
using namespace EH_SPEC_NS;
#endif
