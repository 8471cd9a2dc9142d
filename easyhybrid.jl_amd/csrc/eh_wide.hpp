// Wide-layer variant of the fused step kernel (hidden widths up to 16*NBH = 128).
//
// Same math, data layout, parameter image and slab contract as eh_step_kernel (eh_device.hpp), but a
// different decomposition: with 128-wide layers neither the weight-gradient accumulators
// (128x128 floats = 256 registers per lane) nor a per-wave activation workspace fit, so the NWV
// (4 or 8) waves of a workgroup share ONE macro-tile of 16*NT samples and split every layer by OUTPUT
// ROWS: wave w owns feature blocks [w*MB, (w+1)*MB), MB = NBH/NWV.  Each wave computes its slice of every
// layer's outputs (B operands = the previous layer's activations, read from the shared LDS image in
// MFMA operand order), of every dH, and owns the matching slice of every weight gradient, so no
// gradient ever has to be summed across waves; the price is a workgroup barrier per layer.  The
// output layer is split over K (each wave contracts its own feature blocks, the NWV partial
// outputs meet in the one-sample-per-lane mechanistic stage, which wave 0 runs).
#pragma once
#include "eh_device.hpp"

// per-wave accumulator order; the host maps canonical index -> staging position ((wave*na + k)*64 + lane)*4 + r (rmap)
struct EhWideLayout { int mb, kw0, kwh, kwo, kb, kbo, na; };
__host__ __device__ constexpr EhWideLayout eh_wide_layout(int nbi, int nbh, int nl, int nwv) {
    EhWideLayout L{};
    L.mb = nbh / nwv;
    L.kw0 = 0;
    L.kwh = L.kw0 + L.mb * nbi;
    L.kwo = L.kwh + (nl - 1) * L.mb * nbh;
    L.kb = L.kwo + L.mb;
    L.kbo = L.kb + nl * L.mb;
    L.na = L.kbo + 1;
    return L;
}

template <int NBI, int NBH, int NL, int NT, int NWV>
struct EhWideGeom : EhGeom<NBI, NBH, NL, NT, 1> {
    using B = EhGeom<NBI, NBH, NL, NT, 1>;
    static_assert(NBH % NWV == 0, "the waves split the feature blocks evenly");
    static_assert(NWV * 16 <= B::HP, "the split-K output partials alias the delta image");
    static constexpr int DZ_OFF = B::WS_MIN;                            // the delta image (the waves share the tile: not in place as in the per-wave kernel)
    static constexpr int RS_OFF = DZ_OFF + B::HP * B::SR;               // forcings (rows 0..3) and targets (rows 4..7) of the tile
    static constexpr int SG_OFF = RS_OFF + (EH_MAX_FORC + EH_MAX_TARG) * B::SR;   // d(parameter)/d(network output), 16 rows
    static constexpr int WS_FLOATS = SG_OFF + 16 * B::SR;
    static constexpr int TOTAL_FLOATS = B::IMG_FLOATS + WS_FLOATS;      // one shared workspace per workgroup
    static_assert(NWV * eh_wide_layout(NBI, NBH, NL, NWV).na * 256 <= TOTAL_FLOATS, "the end-of-kernel staging of the accumulators overlays image + workspace");
};

// the first n (1..4; more: all four) floats of v to p, a 4-byte aligned address in global memory
__device__ __forceinline__ void eh_store_upto4(float* p, const f32x4& v, int n) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    if (n >= 4) *reinterpret_cast<f32x4u*>(p) = v;
    else { if (n > 0) p[0] = v[0]; if (n > 1) p[1] = v[1]; if (n > 2) p[2] = v[2]; }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would
// make every barrier wait for the next tile's records (global loads issued a tile ahead on purpose).
__device__ __forceinline__ void eh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (EH_SPEC_NS: a translation unit that bakes ONE model descriptor into its kernels ahead of time -- eh_spec.hip -- puts them in a
//  namespace of its own: the same template arguments name a different kernel there than in the generic translation units)
#ifdef EH_SPEC_NS
namespace EH_SPEC_NS {
#endif
template <int NBI, int NBH, int NL, int NT, int NWV, int ACT, int MODE, bool PROG = false>
__global__ __launch_bounds__(64 * NWV, 1) void eh_wide_kernel(const EhNet net_rt, const EhStepArgs a) {
    eh_kernarg_warm<(int)(sizeof(EhNet) + sizeof(EhStepArgs))>();
#ifdef EH_SPEC_NET
    constexpr EhNet net = {EH_SPEC_NET};        // see eh_step_body
#else
    const EhNet& net = net_rt;
#endif
    using G = EhWideGeom<NBI, NBH, NL, NT, NWV>;
    constexpr int MT = G::MT, SR = G::SR, HP = G::HP, S0 = G::S0, SH = G::SH, MB = NBH / NWV, NTH = 64 * NWV;
    constexpr bool TRAIN = MODE == EH_MODE_TRAIN;
    constexpr EhWideLayout WL = eh_wide_layout(NBI, NBH, NL, NWV);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    // (the wave index as a SCALAR: derived from threadIdx it counts as divergent, and every loop / branch on it -- the tile loop
    //  first of all -- would run under an exec mask with saved / restored mask pairs instead of scalar branches)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ws = smem + G::IMG_FLOATS;
    float* const XS = ws + G::XS_OFF;
    float* const HS = ws + G::HS_OFF;
    float* const DZ = ws + G::DZ_OFF;
    float* const OS = ws + G::OS_OFF;
    float* const OSP = DZ;                      // [NWV waves][16][SR] partial outputs of the split-K output layer
    const float* const meta = wl + G::PHI_OFF;
    const int m0 = wave * MB;
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    const bool mechw = wave == 0 && lane < MT;   // the lanes that own one sample each in the mechanistic stage

    // The tile's MT records are MT*C consecutive floats (or MT gathered runs of C): every thread owns
    // the elements e = tid + k*NTH, loads them coalesced one tile ahead and drops them transposed into
    // the [feature][sample] image (predictors) or the forcing / target rows.
    float* const RS = ws + G::RS_OFF;
    constexpr int NEL = (MT * (G::IP + EH_MAX_FORC + EH_MAX_TARG) + NTH - 1) / NTH;
    const int count = (int)a.count, first = (int)a.first, C = a.C;
    const int ntiles = (count + MT - 1) / MT;
    int epk[NEL], nidx[NEL];          // element k: LDS offset | column << 16 | sample << 24 ; -1 = none
    float nx[NEL];
#pragma unroll
    for (int k = 0; k < NEL; ++k) {
        const int e = tid + k * NTH;
        epk[k] = -1; nidx[k] = 0; nx[k] = 0.0f;
        if (e < MT * C) {
            const int smp = e / C, col = e - smp * C;
            const int row = col < net.P ? -1 : (col - net.P < net.F ? col - net.P : EH_MAX_FORC + (col - net.P - net.F));
            const int dst = row < 0 ? G::XS_OFF + col * SR + smp : G::RS_OFF + row * SR + smp;
            epk[k] = dst | (col << 16) | (smp << 24);
        }
    }
    // (straight-line code on purpose, as in eh_wide_bf16.hpp: a dead element -- beyond the tile's C columns, the window's end or the last
    //  tile -- reads word 0 of the window, which exists whenever count > 0, and is replaced where it is consumed, one tile later)
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

    {   // all loads first, then the LDS stores: one memory round trip instead of one per 16 bytes
        constexpr int NI = (G::IMG_FLOATS / 4 + NTH - 1) / NTH, NIB = NI < 16 ? NI : 16;
        for (int e0 = 4 * tid; e0 < G::IMG_FLOATS; e0 += 4 * NTH * NIB) {
            f32x4 tmp[NIB];
#pragma unroll
            for (int u = 0; u < NIB; ++u) {
                const int e = e0 + 4 * NTH * u;
                tmp[u] = e < G::IMG_FLOATS ? *(const f32x4*)&a.image[e] : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < NIB; ++u) {
                const int e = e0 + 4 * NTH * u;
                if (e < G::IMG_FLOATS) *(f32x4*)&wl[e] = tmp[u];
            }
        }
    }
    for (int e = tid; e < G::IP * SR; e += NTH) XS[e] = 0.0f;
    for (int e = tid; e < 16 * SR; e += NTH) OS[e] = 0.0f;          // rows >= K of the output / dO image stay 0
    __syncthreads();
    if (a.bn_part) {       // input BatchNorm, train mode: statistics of this minibatch (see eh_step_kernel)
        if (tid < net.P) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const float m = a.bn_n ? *a.bn_n : (float)count, c0 = a.bn_c[tid];
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

    // split-K partials -> physical parameters: element (k, sample) = tid + u*NTH, its bounds fixed for the whole launch
    float* const SG = ws + G::SG_OFF;
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
    auto load_b = [&](const float* img, int q, f32x4 (&bq)[NT], bool unswish, int layer) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v = img[(16 * q + 4 * g + s) * SR + 16 * t + c];
                if (EhStoresZ<ACT>::value && unswish) v = eh_hval<ACT>(v, layer, 16 * q + 4 * g + s);      // the image holds z for swish / per-net activations
                bq[t][s] = v;
            }
    };

    EH_STAMP(14);
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        EH_STAMP(0);
        const int n_loc = tile * MT + lane;
        const bool live = mechw && (n_loc < count);
        // ---- 1. records -> normalised [feature][sample] image + forcing / target rows; next tile's records in flight
        {
            // the normalisation constants of all elements in one LDS round trip; an element this thread does not own stores into a padding
            // word (row 0 of the predictor image, column MT) that no one reads
            float bm[NEL], br[NEL];
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const int col = (epk[k] >> 16) & 0xFF;
                const int cm = (epk[k] >= 0 && col < net.P) ? col : 0;
                bm[k] = meta[EH_IMG_BNM + cm]; br[k] = meta[EH_IMG_BNR + cm];
            }
#pragma unroll
            for (int k = 0; k < NEL; ++k) {
                const int col = (epk[k] >> 16) & 0xFF;
                const float v = tile * MT + (epk[k] >> 24) < count ? nx[k] : (col >= net.P + net.F ? __builtin_nanf("") : 0.0f);      // beyond the window's end: no sample
                ws[epk[k] >= 0 ? (epk[k] & 0xFFFF) : G::XS_OFF + MT] = (epk[k] >= 0 && col < net.P) ? (v - bm[k]) * br[k] : v;
            }
        }
        fetch(tile + (int)gridDim.x);
        eh_lds_barrier();
        EH_STAMP(1);

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
                for (int t = 0; t < NT; ++t) {
                    const f32x4 z4 = acc[mm][t], hv4 = (EhStoresZ<ACT>::value && TRAIN) ? z4 : eh_act4_rows<ACT>(z4, 0, 16 * (m0 + mm) + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) HS[(16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c] = (EhStoresZ<ACT>::value && TRAIN) ? eh_vgpr(hv4[r]) : hv4[r];      // (eh_vgpr: see there)
                }
        }
        eh_lds_barrier();
        EH_STAMP(2);
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
                load_b(Hp, q, bq, TRAIN, l - 1);
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
                for (int t = 0; t < NT; ++t) {
                    const f32x4 z4 = acc[mm][t], hv4 = (EhStoresZ<ACT>::value && TRAIN) ? z4 : eh_act4_rows<ACT>(z4, l, 16 * (m0 + mm) + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Hl[(16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c] = (EhStoresZ<ACT>::value && TRAIN) ? eh_vgpr(hv4[r]) : hv4[r];
                }
            eh_lds_barrier();
        }
        EH_STAMP(3);
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
                load_b(Hp, q, bq, TRAIN, NL - 1);
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
        eh_lds_barrier();
        // ---- 4b. all threads: sum the partials, sigmoid-scale into the parameter range (GenericHybridModel.jl:348-352)
#pragma unroll
        for (int u = 0; u < NEO; ++u) {
            const int e = tid + u * NTH, k = e / MT, smp = e % MT;
            if (k < net.K && (NEO * NTH == 16 * MT || e < 16 * MT)) {
                float ov = 0.0f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) ov += OSP[(16 * w + k) * SR + smp];
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
        if (mechw) {
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
                        if (eh_target_mae(net.loss_t, t)) { lacc += w * fabsf(r); d = r > 0.0f ? w : (r < 0.0f ? -w : 0.0f); }
#ifdef EH_JIT_LOSS
                        else if (eh_target_prog(net.loss_t, t)) {
                            float dl;
                            const float lv = eh_jit_loss(t, y, valid ? yobs[t] : y, dl);
                            lacc += valid ? w * lv : 0.0f;
                            d = valid ? w * dl : 0.0f;
                        }
#endif
                        else if (eh_target_two_pass(net.loss_t, t, net.T)) {      // two-pass losses (see eh_step_kernel)
                            d = valid ? fmaf(tt[6], cy, fmaf(tt[5], y - tt[1], tt[4])) : 0.0f;
                        }
                        else { lacc += w * r * r; d = 2.0f * w * r; }
                        dy += ot == 0 ? d : 0.0f; dyx[0] += ot == 1 ? d : 0.0f; dyx[1] += ot == 2 ? d : 0.0f;
                        cacc[t] += valid ? 1.0f : 0.0f;
                        syacc += cy; syyacc += cy * cy;
                    } else if (valid) {
                        const float cy = yobs[t] - a.shift[t], ch = y - (a.inv_n ? a.inv_n[EH_TT * t + 1] : a.shift[t]);      // see eh_step_kernel
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
                        else if (kd == EH_PAR_GLOBAL) gacc[j] += dp;
                    }
                }
            }
        }
        eh_lds_barrier();
        EH_STAMP(5);
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
                    if (EhStoresZ<ACT>::value) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[s] = eh_hval<ACT>(b4[s], NL - 1, 16 * m + c);
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
                    for (int r = 0; r < 4; ++r) dzr[mm][t][r] = dh[t][r] * eh_dact_row<ACT>(Hl[(16 * m + 4 * g + r) * SR + 16 * t + c], NL - 1, 16 * m + 4 * g + r);
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
        eh_lds_barrier();
        EH_STAMP(6);
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
                        if (EhStoresZ<ACT>::value) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) b4[s] = eh_hval<ACT>(b4[s], l - 1, 16 * n + c);
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
                load_b(DZ, q, bq, false, 0);
#pragma unroll
                for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float av = W[(16 * q + 4 * g + s) * SH + 16 * (m0 + mm) + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t) dn[mm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[t][s], dn[mm][t], 0, 0, 0);
                    }
            }
            EH_STAMP(7);
            eh_lds_barrier();                         // every wave is done reading dZ_l
            EH_STAMP(8);
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ad = (16 * (m0 + mm) + 4 * g + r) * SR + 16 * t + c;
                        const float d = dn[mm][t][r] * eh_dact_row<ACT>(Hp[ad], l - 1, 16 * (m0 + mm) + 4 * g + r);
                        dzr[mm][t][r] = d;
                        DZ[ad] = d;
                    }
                    aB[l - 1][mm] += dzr[mm][t];
                }
            eh_lds_barrier();
        }
        EH_STAMP(9);
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
        eh_lds_barrier();                             // XS / OS / images are rewritten by the next tile
        EH_STAMP(10);
    }

    EH_STAMP(11);
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
    // Stage the accumulators in LDS (image and workspace are dead now) as [wave][k][lane][4], then all
    // threads write the canonical-order partial coalesced through the host-built position map (rmap);
    // scattering them to the slab directly costs one 64-byte line per float.
    float gs[EH_MAX_PARAMS];
#pragma unroll
    for (int j = 0; j < EH_MAX_PARAMS; ++j) gs[j] = meta[EH_IMG_DPHI + j];
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
    constexpr int GU = 16;                                    // independent map loads in flight per thread (each is an L2 round trip)
    for (int i0 = tid; i0 < net.g_off; i0 += GU * NTH) {
        int pos[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) pos[u] = (i0 + u * NTH < net.g_off) ? a.rmap[i0 + u * NTH] : 0;
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (i0 + u * NTH < net.g_off) out[i0 + u * NTH] = smem[pos[u]];
    }
    if (wave == 0) {
        lacc = eh_wave_sum(lacc); syacc = eh_wave_sum(syacc); syyacc = eh_wave_sum(syyacc);
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t)
            if (t < net.T) cacc[t] = eh_wave_sum(cacc[t]);
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j)
            if (j < net.n_par) gacc[j] = eh_wave_sum(gacc[j]) * gs[j];
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
