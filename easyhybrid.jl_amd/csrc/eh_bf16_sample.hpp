// Sample-owned training kernel of the bf16 modes (precision = "bf16_fwd" / "bf16", see eh_wide_bf16.hpp for their semantics): the
// round-5 rewrite of the config-5 tile.  eh_widebf_kernel splits every layer by ROWS over the eight waves of a workgroup, so a
// 64-sample tile crosses eleven workgroup barriers and every phase exposes its LDS round trip, its MFMA drain and its tanh; the
// matrix pipe was 7.8 % busy.  Here a wave OWNS sixteen samples for the whole forward / backward chain:
//
//   phase A (no barrier, nothing shared but the weight images): the MFMA puts the SAMPLE on its N dimension, so a layer's 16 x 16
//     output block (C/D layout: lane (c = sample, g), register r <-> feature 16m + 4g + r) is, converted to bf16 and packed in
//     pairs of blocks, already the B operand of the next layer's products -- k-slot (kk, g, j) <-> feature
//     32kk + 16(j >> 2) + 4g + (j & 3) -- as long as the weight fragment is read in the same order (two 8-byte reads of a natural
//     row; for the backward W^T dZ the transposing LDS read takes the row addresses per lane, so the order costs nothing).
//     Forward, sigma-scaling, mechanistic stage (one sample per lane, 16 lanes, wave-private scratch), dH and dZ of every layer
//     never leave the wave's registers.
//   phase B (row split, as before): the weight gradients contract over SAMPLES, so the bf16 activations and deltas of 64 samples at a
//     time (four waves' worth; 32 in the three-term mode) are staged in one LDS image set [sample][feature], and every wave
//     accumulates ITS rows of every dW from them through ds_read_b64_tr_b16.  Two barriers per staged set.
//
// 128-sample tiles, eight waves, four barriers per tile instead of twenty-two per 128 samples.  LDS strides: weight rows 144
// elements with an 8-element skew on odd row octets (both the 8-byte row fragments of the forward and the transposed reads of
// the backward are then conflict-free); staged images 144 / 48 / 16 (eight consecutive sample rows 8 dwords apart mod 64).
// Same slab-row contract as eh_widebf_kernel's direct-store form (one network: a.rmap == nullptr), same parameter image, same
// mechanistic stage (eh_mech_stage_lane); evaluation passes and MultiNN models stay on eh_widebf_kernel.
#pragma once
#include "eh_wide_bf16.hpp"

template <int NBI, int NBH, int NL, int NWV, int NS>
struct EhBfsGeom {
    using F = EhGeom<NBI, NBH, NL, 1, 1>;                  // the fp32 parameter image in global memory
    static_assert(NBH % NWV == 0, "the waves split the feature blocks of the weight gradients evenly");
    static_assert(NBH % 2 == 0, "activations are handed on as pairs of 16-feature blocks (one 32-deep k-step)");
    static_assert(NBI <= 2, "one k-step of predictors; the input-normalisation table holds 32");
    static_assert(NS == 1 || NS == 3, "delta terms");
    static constexpr int MT = 16 * NWV, HP = 16 * NBH, IP = 16 * NBI, KP0 = 32, KSH = HP / 32;
    static constexpr int S0B = KP0 + 8;                    // layer-0 weights: 16-byte natural-order row fragments
    static constexpr int SHW = HP + 16;                    // hidden / output weights (+ 8-element skew on rows with bit 3 set)
    static constexpr int R = (NS == 1 ? 64 : 32) < MT ? (NS == 1 ? 64 : 32) : MT;      // samples per staged set
    static constexpr int NR = MT / R, WPR = R / 16;        // staged sets per tile; owner waves per set
    static constexpr int S0S = KP0 + 16, SHS = HP + 16, DOB = 16;      // staged images: row strides
    static constexpr int SW = 17;                          // wave scratch: row stride
    // LDS map in floats; every offset a multiple of 4
    static constexpr int WB0_OFF = 0;                                       // bf16 [HP][S0B]
    static constexpr int WBH_OFF = WB0_OFF + HP * S0B / 2;                  // (NL-1) x bf16 [HP][SHW]
    static constexpr int WBO_OFF = WBH_OFF + (NL - 1) * HP * SHW / 2;       // bf16 [16][SHW]
    static constexpr int B_OFF = WBO_OFF + 16 * SHW / 2;                    // fp32 biases: NL * HP + 16
    static constexpr int PHI_OFF = B_OFF + NL * HP + 16;                    // fp32 EH_IMG_* block
    static constexpr int KT_OFF = PHI_OFF + EH_IMG_META;                    // fp32 sigma-scaling table per NN output row: lo[16], hi - lo[16]
    static constexpr int SCR_OFF = KT_OFF + 32;                             // NWV x wave scratch: OS[16][SW], SG[16][SW], RS[8][SW]
    static constexpr int SCR_WAVE = 4 * ((40 * SW + 3) / 4);
    static constexpr int MA_OFF = SCR_OFF + NWV * SCR_WAVE;                 // NWV x 16 running sums of the mechanistic stage (global-parameter gradients, loss terms, counts)
    static constexpr int XS_OFF = MA_OFF + NWV * 16;                        // bf16 [R][S0S]
    static constexpr int HS_OFF = XS_OFF + R * S0S / 2;                     // NL x bf16 [R][SHS]
    static constexpr int DS_OFF = HS_OFF + NL * R * SHS / 2;                // NL x NS x bf16 [R][SHS]
    static constexpr int DOS_OFF = DS_OFF + NL * NS * R * SHS / 2;          // NS x bf16 [R][DOB]
    static constexpr int STAGE_END = DOS_OFF + NS * R * DOB / 2;
    static constexpr int TOTAL_FLOATS = STAGE_END;
};

// Fragment of a 16x16x32 bf16 MFMA operand whose k-slot (g, j) is image row rbase + 16 (j >> 2) + 4 g + (j & 3), column col0 + c:
// the slot order in which a wave holds a pair of C/D blocks.  rbase a multiple of 32; `skew` = the image's skew of rows 4g .. 4g+3.
__device__ __forceinline__ bf16x8 eh_tr_slot(const __bf16* img, int ld, int rbase, int col0, int lane, int skew = 0) {
    const __bf16* const a0 = img + (rbase + 4 * (lane >> 4) + ((lane >> 2) & 3)) * ld + skew + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((eh_lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((eh_lds_s16x4*)(a0 + 16 * ld));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

__device__ __forceinline__ bf16x8 eh_bf_cat(bf16x4 lo, bf16x4 hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }
__device__ __forceinline__ bf16x4 eh_bf_lo(bf16x8 v) { return bf16x4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ bf16x4 eh_bf_hi(bf16x8 v) { return bf16x4{v[4], v[5], v[6], v[7]}; }
__device__ __forceinline__ bf16x4 eh_bf_pack4(const f32x4& v) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }

// The activation of a block whose result is rounded to bfloat16 straight away.  tanh: NNlib's tanh_fast (eh_tanh) switches to sign(x) at
// x^2 >= 66 so that a saturated unit hands on exactly +-1; here the rounding does that: the rational stays within [1 - 2.4e-7,
// 1 + 1.5e-6] for 8.12 <= |x| <= 9, which IS 1 in bfloat16, so clamping x to +-9 (one v_med3) replaces compare / copysign / select (three
// instructions per value) -- bit for bit the same bfloat16 as bf16(eh_tanh(x)).
template <int ACT>
__device__ __forceinline__ bf16x4 eh_act4_bf(const f32x4& z) {
    if constexpr (ACT == EH_ACT_TANH) {
        auto fma2 = [](f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); };
        f32x4 t;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 x = {__builtin_amdgcn_fmed3f(z[2 * q], -9.0f, 9.0f), __builtin_amdgcn_fmed3f(z[2 * q + 1], -9.0f, 9.0f)};
            const f32x2 x2 = x * x, one = {1.0f, 1.0f};
            f32x2 n = fma2(x2, f32x2{1.587199e-8f, 1.587199e-8f}, 2.2332108e-5f);
            n = fma2(x2, n, 0.0035974074f); n = fma2(x2, n, 0.1346604f); n = __builtin_elementwise_fma(x2, n, one);
            f32x2 d = fma2(x2, f32x2{8.7767893e-7f, 8.7767893e-7f}, 0.0003453992f);
            d = fma2(x2, d, 0.026262015f); d = fma2(x2, d, 0.4679937f); d = __builtin_elementwise_fma(x2, d, one);
            const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            const f32x2 y = x * (n * r);
            t[2 * q] = y[0]; t[2 * q + 1] = y[1];
        }
        return eh_bf_pack4(t);
    } else return eh_bf_pack4(eh_act4<ACT>(z));
}

// delta of a block from d loss / d activation and the stored (rounded) activation: dz = dh * act'(h).  tanh on register pairs:
// dh (1 - h^2) = dh - (dh h) h, two packed instructions per pair.
template <int ACT>
__device__ __forceinline__ f32x4 eh_dz4(const f32x4& dh, const bf16x4& hq) {
    if constexpr (ACT == EH_ACT_TANH) {
        f32x4 dz;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 h = {eh_bf2f(hq[2 * q]), eh_bf2f(hq[2 * q + 1])}, d = {dh[2 * q], dh[2 * q + 1]};
            const f32x2 t = d * h, r = __builtin_elementwise_fma(-t, h, d);
            dz[2 * q] = r[0]; dz[2 * q + 1] = r[1];
        }
        return dz;
    } else {
        f32x4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = dh[r] * eh_dact<ACT>(eh_bf2f(hq[r]));
        return dz;
    }
}

#ifdef EH_SPEC_NS
namespace EH_SPEC_NS {
#endif
template <int NBI, int NBH, int NL, int NWV, int ACT, bool PROG = false, int NS = 1>
__global__ __launch_bounds__(64 * NWV, 1) void eh_bfs_kernel(const EhNet net_rt, const EhStepArgs a) {
    eh_kernarg_warm<(int)(sizeof(EhNet) + sizeof(EhStepArgs))>();
#ifdef EH_SPEC_NET
    constexpr EhNet net = {EH_SPEC_NET};        // see eh_step_body
#else
    const EhNet& net = net_rt;
#endif
    static_assert(!EhStoresZ<ACT>::value, "the bf16 kernels keep only the rounded activation");
    using G = EhBfsGeom<NBI, NBH, NL, NWV, NS>;
    using F = typename G::F;
    constexpr int MT = G::MT, HP = G::HP, IP = G::IP, KP0 = G::KP0, KSH = G::KSH, S0B = G::S0B, SHW = G::SHW, MB = NBH / NWV, NTH = 64 * NWV;
    constexpr int R = G::R, NR = G::NR, WPR = G::WPR, S0S = G::S0S, SHS = G::SHS, DOB = G::DOB, SW = G::SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __bf16* const WB0 = reinterpret_cast<__bf16*>(smem + G::WB0_OFF);
    __bf16* const WBH = reinterpret_cast<__bf16*>(smem + G::WBH_OFF);
    __bf16* const WBO = reinterpret_cast<__bf16*>(smem + G::WBO_OFF);
    const float* const BIAS = smem + G::B_OFF;
    const float* const meta = smem + G::PHI_OFF;
    const float* const KT = smem + G::KT_OFF;
    float* const OSw = smem + G::SCR_OFF + wave * G::SCR_WAVE;
    float* const SGw = OSw + 16 * SW;
    float* const RSw = SGw + 16 * SW;
    float* const MAw = smem + G::MA_OFF + wave * 16;
    __bf16* const XS = reinterpret_cast<__bf16*>(smem + G::XS_OFF);
    __bf16* const HS = reinterpret_cast<__bf16*>(smem + G::HS_OFF);
    __bf16* const DS = reinterpret_cast<__bf16*>(smem + G::DS_OFF);
    __bf16* const DOS = reinterpret_cast<__bf16*>(smem + G::DOS_OFF);
    constexpr int PH = R * SHS, PO = R * DOB;             // plane sizes (elements) of the staged images
    const int skw = 8 * ((c >> 3) & 1);                   // weight images: skew of row 16m + c ...
    const int skt = 8 * (g >> 1);                         // ... and of rows 32kk + {0, 16} + 4g + q (the transposed reads)
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    const int m0 = wave * MB;

    // ---- records: lane (c, g) of wave w reads predictors 8g .. 8g+7 and columns P + g, P + 4 + g of sample tile * MT + 16 w + c,
    // one tile ahead ---------------------------------------------------------------------------------------------------------------
    const int count = (int)a.count, first = (int)a.first, C = a.C;
    const int ntiles = (count + MT - 1) / MT;
    const bool vec_ok = (C & 3) == 0 && (net.P & 7) == 0;
    float xr[8], er[2];
    auto fetch = [&](int tile) {
        const int n_loc = tile * MT + 16 * wave + c;
        const bool in = tile < ntiles && n_loc < count;
        const int s0 = in ? n_loc : 0;
        const long long smp = a.idx ? (long long)a.idx[first + s0] : (long long)(first + s0);
        const float* const rec = a.recs + smp * C;
        if (vec_ok) {
            const bool on = in && 8 * g < net.P;
            const float* const p = on ? rec + 8 * g : a.recs;
            const f32x4 u = *(const f32x4*)p, v = *(const f32x4*)(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xr[j] = on ? u[j] : 0.0f; xr[4 + j] = on ? v[j] : 0.0f; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool on = in && 8 * g + j < net.P;
                const float v = rec[on ? 8 * g + j : 0];
                xr[j] = on ? v : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = net.P + 4 * u + g;
            const bool on = in && col < C;
            const float v = rec[on ? col : 0];
            er[u] = on ? v : (col >= net.P + net.F ? __builtin_nanf("") : 0.0f);      // beyond the window's end: no sample
        }
    };
    EH_STAMP(13);
    // (a static s_setprio(1) for the younger half of the workgroup -- which reaches the tile's first barrier ~6 k cycles after the older half --
    //  was measured: 42.9 against 42.7 us per step without it; the SIMD is busy with the younger wave while the older one waits)
    if (ntiles > 0) fetch((int)blockIdx.x);

    // ---- parameter image (fp32, EhGeom layout, kept by the optimiser kernel) -> bf16 weights, fp32 biases + meta block --------
    {
        constexpr int N0 = (HP * KP0 / 4 + NTH - 1) / NTH, NH = (HP * HP / 4 + NTH - 1) / NTH, NO = (16 * HP / 4 + NTH - 1) / NTH;
        f32x4 r0[N0], rh[NL > 1 ? NL - 1 : 1][NH], ro[NO];
        auto ld_rows = [&](const float* src, int sld, int scols, int rows, int kcols, f32x4* v, int n) {
            const int qp = kcols / 4, tot = rows * qp;
            for (int u = 0; u < n; ++u) {
                const int i = tid + u * NTH, row = i / qp, col = 4 * (i - row * qp);
                v[u] = (i < tot && col < scols) ? *(const f32x4*)&src[row * sld + col] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        auto st_rows = [&](__bf16* dst, int dld, int sk, int rows, int kcols, const f32x4* v, int n) {
            const int qp = kcols / 4, tot = rows * qp;
            for (int u = 0; u < n; ++u) {
                const int i = tid + u * NTH, row = i / qp, col = 4 * (i - row * qp);
                if (i < tot) *(bf16x4*)&dst[row * dld + sk * ((row >> 3) & 1) + col] = eh_bf_pack4(v[u]);
            }
        };
        ld_rows(a.image + F::W0_OFF, F::S0, IP, HP, KP0, r0, N0);
#pragma unroll
        for (int l = 1; l < NL; ++l) ld_rows(a.image + F::WH_OFF + (l - 1) * HP * F::SH, F::SH, HP, HP, HP, rh[l - 1], NH);
        ld_rows(a.image + F::WO_OFF, F::SH, HP, 16, HP, ro, NO);
        st_rows(WB0, S0B, 0, HP, KP0, r0, N0);
#pragma unroll
        for (int l = 1; l < NL; ++l) st_rows(WBH + (l - 1) * HP * SHW, SHW, 8, HP, HP, rh[l - 1], NH);
        st_rows(WBO, SHW, 8, 16, HP, ro, NO);
        for (int e = tid; e < NL * HP + 16 + EH_IMG_META; e += NTH) smem[G::B_OFF + e] = a.image[F::B_OFF + e];      // (B_OFF .. PHI_OFF + META is one run in both layouts)
    }
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
    }
    if (tid < 16) {        // sigma-scaling of NN output row tid (GenericHybridModel.jl:348-352): lower bound, upper - lower
        float lo = 0.0f, sc = 0.0f;
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j)
            if (j < net.n_par && pkind(j) == EH_PAR_NEURAL && pidx(j) == tid) { lo = meta[EH_IMG_LO + j]; sc = meta[EH_IMG_SC + j]; }
        smem[G::KT_OFF + tid] = lo; smem[G::KT_OFF + 16 + tid] = sc;
    }
    __syncthreads();

    // accumulators: this wave's row slice of every weight gradient and bias gradient (the latter as products with a vector of ones: every
    // column of the block holds the row sums)
    f32x4 aW0[MB][NBI], aWh[NL > 1 ? NL - 1 : 1][MB][NBH], aWo[MB], aB[NL][MB], aBo = f32x4{0, 0, 0, 0};
    if (lane < 16) MAw[lane] = 0.0f;
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

    // weight fragment in k-slot order: row `row` of a hidden / output weight image, features 32kk + 4g .. +3 and 32kk + 16 + 4g .. +3
    auto w_slot = [&](const __bf16* W, int row, int kk) {
        const __bf16* const p = W + row * SHW + skw + 32 * kk + 4 * g;
        return eh_bf_cat(*(const bf16x4*)p, *(const bf16x4*)(p + 16));
    };
    const bf16x4 zero4 = __builtin_bit_cast(bf16x4, s16x4{0, 0, 0, 0});
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});
    // delta block (fp32, C/D layout) -> its NS bf16 terms
    auto split_into = [&](const f32x4& d, bf16x4* t) {
        if constexpr (NS == 1) t[0] = eh_bf_pack4(d);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { __bf16 x, y, z; eh_split3(d[r], x, y, z); t[0][r] = x; t[1][r] = y; t[2][r] = z; }
        }
    };
    // the records of the NEXT tile are requested when phase B starts and normalised / rounded / parked when it ends: phase A holds no
    // registers for them
    bf16x8 xb_next = __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
    auto consume = [&]() {
        const f32x4 m0v = *(const f32x4*)&meta[EH_IMG_BNM + 8 * g], m1v = *(const f32x4*)&meta[EH_IMG_BNM + 8 * g + 4];
        const f32x4 r0v = *(const f32x4*)&meta[EH_IMG_BNR + 8 * g], r1v = *(const f32x4*)&meta[EH_IMG_BNR + 8 * g + 4];
        f32x4 u, v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[j] = 8 * g + j < net.P ? (xr[j] - m0v[j]) * r0v[j] : 0.0f;
            v[j] = 8 * g + 4 + j < net.P ? (xr[4 + j] - m1v[j]) * r1v[j] : 0.0f;
        }
        xb_next = eh_bf_cat(eh_bf_pack4(u), eh_bf_pack4(v));
        // forcings / targets of the wave's samples -> scratch rows (forcing column f: row f; target t: row EH_MAX_FORC + t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = 4 * q + g;            // column P + col of the record
            if (net.P + col < C) RSw[(col < net.F ? col : EH_MAX_FORC + (col - net.F)) * SW + c] = er[q];
        }
    };
    EH_STAMP(14);
    if (ntiles > 0) consume();
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        EH_STAMP(0);
        const int n_loc = tile * MT + 16 * wave + c;
        const bool live = n_loc < count;
        // ================= phase A: this wave's sixteen samples, registers only ==============================================
        // Every product's weight fragments (and the block's bias) are requested one block ahead of the MFMAs that use them -- the
        // scheduling fence keeps the request in front of the previous block's MFMAs and activation, which is what hides the LDS round
        // trip: two waves per SIMD do not (measured: 125 cycles per MFMA with the reads next to their use).
        const bf16x8 xb = xb_next;
        bf16x8 hb[NL][KSH], dzb[NL][NS][KSH];
        bf16x4 dOb[NS];
        EH_STAMP(1);
        // ---- forward: layer 0 (natural k order: the predictors come from memory) ----
        {
            bf16x8 wq[2]; f32x4 bq[2]; bf16x4 t[2];
            auto ld = [&](int m) { wq[m & 1] = *(const bf16x8*)&WB0[(16 * m + c) * S0B + 8 * g]; bq[m & 1] = *(const f32x4*)&BIAS[16 * m + 4 * g]; };
            ld(0); ld(1);
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                const bf16x8 w = wq[m & 1]; const f32x4 b = bq[m & 1];
                if (m + 2 < NBH) ld(m + 2);
                EH_SCHED_FENCE();
                t[m & 1] = eh_act4_bf<ACT>(__builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb, b, 0, 0, 0));
                if (m & 1) hb[0][m >> 1] = eh_bf_cat(t[0], t[1]);
            }
        }
        EH_STAMP(2);
        // ---- hidden layers, then the output layer's fragments ride the last block's slot ----
        bf16x8 wo[KSH]; f32x4 bo;
        auto ld_out = [&]() {
#pragma unroll
            for (int kk = 0; kk < KSH; ++kk) wo[kk] = w_slot(WBO, c, kk);
            bo = *(const f32x4*)&BIAS[NL * HP + 4 * g];
        };
        if constexpr (NL == 1) ld_out();
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const __bf16* const W = WBH + (l - 1) * HP * SHW;
            bf16x8 wq[2][KSH]; f32x4 bq[2]; bf16x4 t[2];
            auto ld = [&](int m) {
#pragma unroll
                for (int kk = 0; kk < KSH; ++kk) wq[m & 1][kk] = w_slot(W, 16 * m + c, kk);
                bq[m & 1] = *(const f32x4*)&BIAS[l * HP + 16 * m + 4 * g];
            };
            ld(0);
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                if (m + 1 < NBH) ld(m + 1);
                else if (l == NL - 1) ld_out();
                EH_SCHED_FENCE();
                f32x4 acc = bq[m & 1];
#pragma unroll
                for (int kk = 0; kk < KSH; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[m & 1][kk], hb[l - 1][kk], acc, 0, 0, 0);
                t[m & 1] = eh_act4_bf<ACT>(acc);
                if (m & 1) hb[l][m >> 1] = eh_bf_cat(t[0], t[1]);
            }
        }
        EH_STAMP(3);
        // ---- output layer (16 padded rows), sigma-scaling into the parameter range ----
        // (s_setprio 1 / 3 over this short serial stretch -- dependent MFMAs, sigma, the 16-lane mechanistic stage -- measured: 42.3-43.0 us per step
        //  with it, 43.0 without, inside the lease's noise; not kept)
        {
            f32x4 o = bo;
#pragma unroll
            for (int kk = 0; kk < KSH; ++kk) o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[kk], hb[NL - 1][kk], o, 0, 0, 0);
            const f32x4 klo = *(const f32x4*)&KT[4 * g], ksc = *(const f32x4*)&KT[16 + 4 * g];
            EH_STAMP_FINE(15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = o[r], sv = 1.0f;
                if (net.scale_nn) {
                    const float sgm = eh_sigmoid(o[r]);
                    pv = fmaf(ksc[r], sgm, klo[r]);
                    sv = ksc[r] * sgm * (1.0f - sgm);
                }
                OSw[(4 * g + r) * SW + c] = 4 * g + r < net.K ? pv : 0.0f;
                SGw[(4 * g + r) * SW + c] = sv;
            }
        }
        EH_WAVE_SYNC();
        EH_STAMP(4);
        // ---- mechanistic model + masked loss + its pullback: one sample per lane, lanes 0 .. 15 ----
        {
            // (the stage's sums live in registers only across the stage: folded over the sixteen lanes and added to the wave's LDS row)
            EhMechAcc MA;
            MA.clear();
            if (lane < 16) eh_mech_stage_lane<true, PROG>(net, a, lane, live, n_loc, SW, RSw, OSw, SGw, meta, MA);
            float v[16];
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j) v[j] = j < net.n_par ? MA.gacc[j] : 0.0f;
            v[8] = MA.lacc; v[9] = MA.syacc; v[10] = MA.syyacc;
#pragma unroll
            for (int t = 0; t < EH_MAX_TARG; ++t) v[11 + t] = t < net.T ? MA.cacc[t] : 0.0f;
            v[15] = 0.0f;
            float mine = 0.0f;
#pragma unroll
            for (int j = 0; j < 15; ++j) {
                const float s = eh_row16_sum(v[j]);
                mine = lane == j ? s : mine;
            }
            if (lane < 15) MAw[lane] += mine;
        }
        EH_WAVE_SYNC();
        EH_STAMP(5);
        // ---- backward: last hidden layer  dH = bf16(Wo)^T dO  (k-slot (g, j < 4) <-> output row 4g + j; the upper half of the k-step is zero)
        {
            bf16x4 wq[2], t[2][NS];
            auto ld = [&](int m) {
                const __bf16* const a0 = WBO + (4 * g + ((lane >> 2) & 3)) * SHW + skt + 16 * m + 4 * (lane & 3);
                wq[m & 1] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((eh_lds_s16x4*)a0));
            };
            ld(0); ld(1);
            {
                f32x4 dO;
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[r] = OSw[(4 * g + r) * SW + c];       // d loss / d NN output rows 4g .. 4g+3 (rows >= K: 0)
                split_into(dO, dOb);
            }
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                const bf16x8 afr = eh_bf_cat(wq[m & 1], zero4);
                if (m + 2 < NBH) ld(m + 2);
                EH_SCHED_FENCE();
                f32x4 dh = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int p = 0; p < NS; ++p) dh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, eh_bf_cat(dOb[p], zero4), dh, 0, 0, 0);
                const bf16x4 hq = (m & 1) ? eh_bf_hi(hb[NL - 1][m >> 1]) : eh_bf_lo(hb[NL - 1][m >> 1]);
                const f32x4 dz = eh_dz4<ACT>(dh, hq);
                split_into(dz, t[m & 1]);
                if (m & 1) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) dzb[NL - 1][p][m >> 1] = eh_bf_cat(t[0][p], t[1][p]);
                }
            }
        }
        EH_STAMP(6);
        // ---- hidden layers backward: dH_{l-1} = bf16(W_l)^T dZ_l ----
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            const __bf16* const W = WBH + (l - 1) * HP * SHW;
            bf16x8 wq[2][KSH]; bf16x4 t[2][NS];
            auto ld = [&](int m) {
#pragma unroll
                for (int kk = 0; kk < KSH; ++kk) wq[m & 1][kk] = eh_tr_slot(W, SHW, 32 * kk, 16 * m, lane, skt);
            };
            ld(0);
#pragma unroll
            for (int m = 0; m < NBH; ++m) {
                if (m + 1 < NBH) ld(m + 1);
                EH_SCHED_FENCE();
                f32x4 dn = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < KSH; ++kk)
#pragma unroll
                    for (int p = 0; p < NS; ++p) dn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[m & 1][kk], dzb[l][p][kk], dn, 0, 0, 0);
                const bf16x4 hq = (m & 1) ? eh_bf_hi(hb[l - 1][m >> 1]) : eh_bf_lo(hb[l - 1][m >> 1]);
                const f32x4 dz = eh_dz4<ACT>(dn, hq);
                split_into(dz, t[m & 1]);
                if (m & 1) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) dzb[l - 1][p][m >> 1] = eh_bf_cat(t[0][p], t[1][p]);
                }
            }
        }
        EH_STAMP(7);
        // ================= phase B: weight gradients, R samples at a time through the staged image set ===========================
        fetch(tile + (int)gridDim.x);
#pragma unroll 1
        for (int rd = 0; rd < NR; ++rd) {
            if (wave / WPR == rd) {
                const int row = 16 * (wave % WPR) + c;
                *(bf16x8*)&XS[row * S0S + 8 * g] = xb;
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int kk = 0; kk < KSH; ++kk) {
                        __bf16* const q = HS + l * PH + row * SHS + 32 * kk + 4 * g;
                        *(bf16x4*)q = eh_bf_lo(hb[l][kk]); *(bf16x4*)(q + 16) = eh_bf_hi(hb[l][kk]);
#pragma unroll
                        for (int p = 0; p < NS; ++p) {
                            __bf16* const d = DS + (l * NS + p) * PH + row * SHS + 32 * kk + 4 * g;
                            *(bf16x4*)d = eh_bf_lo(dzb[l][p][kk]); *(bf16x4*)(d + 16) = eh_bf_hi(dzb[l][p][kk]);
                        }
                    }
#pragma unroll
                for (int p = 0; p < NS; ++p) *(bf16x4*)&DOS[p * PO + row * DOB + 4 * g] = dOb[p];
            }
            eh_lds_barrier();
            EH_STAMP(8);
#pragma unroll
            for (int kk = 0; kk < R / 32; ++kk)
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    const int m = m0 + mm;
                    // every fragment of the k-step is requested before the first product (the fence): one LDS round trip per k-step, not per MFMA
                    bf16x8 ah[NL > 1 ? NL - 1 : 1][NS], bh[NL > 1 ? NL - 1 : 1][NBH], a0f[NS], b0f[NBI], aof[NS], bof;
#pragma unroll
                    for (int l = NL - 1; l >= 1; --l) {
#pragma unroll
                        for (int p = 0; p < NS; ++p) ah[l - 1][p] = eh_tr_slot(DS + (l * NS + p) * PH, SHS, 32 * kk, 16 * m, lane);
#pragma unroll
                        for (int n = 0; n < NBH; ++n) bh[l - 1][n] = eh_tr_slot(HS + (l - 1) * PH, SHS, 32 * kk, 16 * n, lane);
                    }
#pragma unroll
                    for (int p = 0; p < NS; ++p) { a0f[p] = eh_tr_slot(DS + p * PH, SHS, 32 * kk, 16 * m, lane); aof[p] = eh_tr_slot(DOS + p * PO, DOB, 32 * kk, 0, lane); }
#pragma unroll
                    for (int n = 0; n < NBI; ++n) b0f[n] = eh_tr_slot(XS, S0S, 32 * kk, 16 * n, lane);
                    bof = eh_tr_slot(HS + (NL - 1) * PH, SHS, 32 * kk, 16 * m, lane);
                    EH_SCHED_FENCE();
                    // dW_l[own rows][all columns] += dZ_l (own rows) * bf16(H_{l-1})^T : contraction over the staged samples
#pragma unroll
                    for (int l = NL - 1; l >= 1; --l)
#pragma unroll
                        for (int n = 0; n < NBH; ++n)
#pragma unroll
                            for (int p = 0; p < NS; ++p) aWh[l - 1][mm][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[l - 1][p], bh[l - 1][n], aWh[l - 1][mm][n], 0, 0, 0);
                    // layer 0: dW0[own rows] += dZ_0 * bf16(X)^T ; output layer: dWo[k-out][own features] += dO * bf16(H_last)^T
#pragma unroll
                    for (int n = 0; n < NBI; ++n)
#pragma unroll
                        for (int p = 0; p < NS; ++p) aW0[mm][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0f[p], b0f[n], aW0[mm][n], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < NS; ++p) aWo[mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aof[p], bof, aWo[mm], 0, 0, 0);
                    // bias gradients: the row sums of the same delta fragments (products with a vector of ones: exact sums of what the weight
                    // gradients multiply -- the three terms in the exact mode, the once-rounded delta in the "bf16" mode)
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
#pragma unroll
                        for (int l = NL - 1; l >= 1; --l) aB[l][mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[l - 1][p], ones8, aB[l][mm], 0, 0, 0);
                        aB[0][mm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0f[p], ones8, aB[0][mm], 0, 0, 0);
                        if (mm == 0 && wave == 0) aBo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aof[p], ones8, aBo, 0, 0, 0);
                    }
                }
            eh_lds_barrier();                         // the set is rewritten by the next owners
            EH_STAMP(9);
        }
        consume();
        EH_STAMP(10);
    }
    EH_STAMP(11);

    // ---- one partial row per workgroup: weight- and bias-gradient blocks straight from the registers (eh_widebf_kernel's direct form: ONE
    // network, canonical order plain column-major per layer); the scalar sums of the mechanistic stage folded over the waves through LDS
    float* const out = a.slab + (long long)blockIdx.x * a.n_acc;
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
        const int col = 16 * (m0 + mm) + c, nk = net.K - 4 * g;
        if (col < Wd[NL - 1] && nk > 0) eh_store_upto4(out + im[EH_IMG_WOFF + NL] + col * net.K + 4 * g, aWo[mm], nk);
    }
    if (wave == 0 && c == 0 && net.K - 4 * g > 0) eh_store_upto4(out + im[EH_IMG_BOFF + NL] + 4 * g, aBo, net.K - 4 * g);
    __syncthreads();          // (every wave's last addition to its row of sums has landed)
    if (tid < 16) {           // [0..7] global-parameter sums, [8] loss, [9] / [10] target sums, [11..14] counts
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) s += smem[G::MA_OFF + w * 16 + tid];
        if (tid < EH_MAX_PARAMS) {
            if (tid < net.n_par && pkind(tid) == EH_PAR_GLOBAL) out[net.g_off + pidx(tid)] = s * meta[EH_IMG_DPHI + tid];
        } else if (tid == 8) out[net.n_theta] = s;
        else if (tid == 9) out[net.n_theta + 1 + net.T] = s;
        else if (tid == 10) out[net.n_theta + 2 + net.T] = s;
        else if (tid - 11 < net.T) out[net.n_theta + 1 + (tid - 11)] = s;
    }
    EH_STAMP(12);
}
#ifdef EH_SPEC_NS
}   // namespace EH_SPEC_NS
using namespace EH_SPEC_NS;
#endif
