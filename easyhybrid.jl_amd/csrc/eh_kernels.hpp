// The small kernels around the fused step kernel: partial-slab reduction + optimiser update, the stand-alone mechanistic stage,
// valid counts, moment coefficients, input BatchNorm statistics, epoch permutation, record packing.  Included by eh_api.hip only
// (the kernels have external linkage: one translation unit).
#pragma once
#include "eh_internal.hpp"

// --------------------------------------------------------------------------------------------
// small kernels
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool eh_is_weight(const EhImg& im, int idx) { return idx < im.g_off && (im.imap ? im.imap[idx] < im.b_off : im.wflag[idx] != 0); }
// d(extra loss) / d theta_idx = 2 * this * theta_idx
__device__ __forceinline__ float eh_l2_coef(const EhImg& im, int idx) { return im.l2s * (im.l2w ? im.l2w[idx] : (eh_is_weight(im, idx) ? im.l2c : 0.0f)); }

// the extra loss of the CURRENT parameters (before the optimiser kernel touches them): l2c * sum of squared Dense weights, or sum_i l2w[i] theta_i^2
__global__ __launch_bounds__(256) void eh_weight_l2_kernel(const float* theta, EhImg im, float* out) {
    __shared__ float red[4];
    float s = 0.0f;
    if (im.l2w) {
        for (int i = threadIdx.x; i < im.n_theta; i += 256) { const float w = theta[i]; s = fmaf(im.l2w[i] * w, w, s); }
    } else {
        for (int i = threadIdx.x; i < im.g_off; i += 256)
            if (eh_is_weight(im, i)) { const float w = theta[i]; s += w * w; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = im.l2s * (im.l2w ? 1.0f : im.l2c) * ((red[0] + red[1]) + (red[2] + red[3]));
}

__device__ __forceinline__ void eh_image_store(const EhImg& im, int idx, float th) {
    if (idx < im.g_off) {
        if (im.imap) im.image[im.imap[idx]] = th;      // (layer-wise form: the kernels read the canonical theta itself)
    } else {   // raw global -> physical value and sigmoid slope (GenericHybridModel.jl:348-352)
        const int g = idx - im.g_off, j = im.glob_par[g];
        const float s = 1.0f / (1.0f + expf(-th));
        im.image[im.phi_off + EH_IMG_PHI + j] = im.glo[g] + (im.ghi[g] - im.glo[g]) * s;
        im.image[im.phi_off + EH_IMG_DPHI + j] = (im.ghi[g] - im.glo[g]) * s * (1.0f - s);
    }
}

__global__ void eh_image_kernel(const float* theta, int n_theta, EhImg im) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_theta) eh_image_store(im, idx, theta[idx]);
}


// ---------------------------------------------------------------------------------------------------------------------
// Mechanistic stage on its own (eh_mech_loss_vjp): o = NN outputs of an MLP that lives OUTSIDE this library -> physical
// parameters (sigma-scaling) -> M(par, forcings) -> masked MSE summed over the targets -> d loss / d o and the gradient of
// the raw global parameters.  Pure streaming: 4 (K + F + T) bytes read and 4 K written per sample, ~20-60 flop -- the one
// stage of the path that the HBM roof bounds.  Every lane owns V consecutive samples (V = 4: 16-byte
// loads and stores, a wave covers 1 KiB runs of every array); sums leave a workgroup once, as one row of partials.
// ---------------------------------------------------------------------------------------------------------------------
struct EhMechArgs {
    const float* o;                          // [K][ld] NN outputs (raw)
    const float* frc[EH_MAX_FORC];           // the F forcing arrays in eh_set_data order
    const float* y[EH_MAX_TARG];             // the T target arrays (NaN = missing)
    float* d_o;                              // [K][ld] d loss / d o
    float* yhat;                             // [T][ld] or nullptr
    long long n, ld;
    const float* meta;                       // parameter image, PHI block (values / d value d raw of the global and fixed parameters, lo, hi - lo)
    const unsigned long long* counts;        // valid samples per target (whole call), counted on the device ...
    unsigned long long counts_v[EH_MAX_TARG];   // ... or handed in by the caller (use_v)
    int use_v;
    float* part;                             // [gridDim.x][EH_MECH_PART] partial sums
    const unsigned* prog;                    // EH_MECH_PROGRAM: the recorded closure (EhStepArgs::prog layout)
    float agg_a;                             // factor of `agg` on the data loss (EhImg::agg_a)
    int tiles;                               // > 0: workgroup b owns the `tiles` consecutive 256 V-sample tiles from b * tiles (a front that moves through memory
                                             // with the dispatch order); 0: grid-stride trips (a capped grid)
};
enum { EH_MECH_PART = 16 };                  // [dL/dpar_j (8) | S_t (4) | pad]
enum { EH_MECH_MAXROWS = 65536 };            // rows of partials (workgroups of the streaming kernel) at most: 4 MiB

template <int V>
__global__ __launch_bounds__(256) void eh_count_valid_kernel(EhMechArgs a, int T, unsigned long long* counts) {
    unsigned c[EH_MAX_TARG] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256 * V;
    for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * V; i0 < a.n; i0 += 4 * stride)
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t)
            if (t < T) {
                if constexpr (V == 4) {
                    f32x4 v[4];                                  // four trips requested before the first is examined
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = i0 + u * stride < a.n ? *(const f32x4*)(a.y[t] + i0 + u * stride) : f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * stride < a.n) c[t] += !__builtin_isnan(v[u][0]) + !__builtin_isnan(v[u][1]) + !__builtin_isnan(v[u][2]) + !__builtin_isnan(v[u][3]);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * stride < a.n) c[t] += !__builtin_isnan(a.y[t][i0 + u * stride]);
                }
            }
    __shared__ unsigned sh[4][EH_MAX_TARG];
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG; ++t) {
        unsigned v = c[t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < T) atomicAdd(&counts[threadIdx.x], (unsigned long long)(sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// (parameters, forcings, outputs) of a registry model: compile-time array sizes, so that a two-parameter model keeps a dozen
// values per sample in registers, not the 8 + 4 + 4 of the largest one (occupancy is what a streaming kernel lives on)
// (EH_MECH_PROGRAM, a recorded closure run by the interpreter of eh_device.hpp: the limits of the program format)
constexpr int eh_mech_np(int m) { return m == EH_MECH_PROGRAM ? EH_MAX_PARAMS : m == EH_MECH_EXPO2POOL ? 4 : (m == EH_MECH_RS_COMPONENTS || m == EH_MECH_RS_COMPONENTS3F) ? 6 : m == EH_MECH_FLUXPART ? 3 : 2; }
constexpr int eh_mech_nf(int m) { return m == EH_MECH_PROGRAM ? EH_MAX_FORC : m == EH_MECH_FLUXPART ? 2 : m == EH_MECH_RS_COMPONENTS3F ? 3 : 1; }
constexpr int eh_mech_no(int m) { return (m == EH_MECH_PROGRAM || m == EH_MECH_FLUXPART) ? 3 : 1; }

// (RbQ10: 68 VGPRs, seven waves per SIMD.  Forcing eight -- amdgpu_waves_per_eu(8): 64 VGPRs -- spills two registers, and a kernel
// with ANY scratch pays for the allocation at every wave launch, which a launch of tens of thousands of short workgroups cannot afford.)
template <int V, int MECH>
__global__ __launch_bounds__(256) void eh_mech_vjp_kernel(const EhNet net, EhMechArgs a) {
    constexpr int NP = eh_mech_np(MECH), NF = eh_mech_nf(MECH), NO = eh_mech_no(MECH), NTG = NO > 1 ? EH_MAX_TARG : 1;
    float cpar[NP], lo[NP], sc[NP], w[NTG];
    int row[NP];                                                 // NN output row of a neural parameter, -1 otherwise
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        cpar[j] = a.meta[EH_IMG_PHI + j]; lo[j] = a.meta[EH_IMG_LO + j]; sc[j] = a.meta[EH_IMG_SC + j];
        row[j] = (((net.par_kind >> (2 * j)) & 3u) == EH_PAR_NEURAL) ? (int)((net.par_idx >> (4 * j)) & 15u) : -1;
    }
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
        const unsigned long long c = a.use_v ? a.counts_v[t] : a.counts[t];
        w[t] = (t < net.T && c > 0) ? a.agg_a / (float)c : 0.0f;
    }
    float gp[NP], S[NTG];
    const bool mae = net.loss == EH_LOSS_MAE;
#pragma unroll
    for (int j = 0; j < NP; ++j) gp[j] = 0.0f;
#pragma unroll
    for (int t = 0; t < NTG; ++t) S[t] = 0.0f;
    const long long i_first = a.tiles > 0 ? ((long long)blockIdx.x * a.tiles * 256 + threadIdx.x) * V : ((long long)blockIdx.x * 256 + threadIdx.x) * V;
    const long long i_step = a.tiles > 0 ? 256ll * V : (long long)gridDim.x * 256 * V;
    // (no std::min here: it takes references, and a reference to a member of the by-value kernarg struct makes the compiler copy the
    // whole struct to scratch -- 200 bytes per lane and 24 VGPRs more, measured)
    const long long n_all = a.n, tile_end = (long long)(blockIdx.x + 1) * a.tiles * 256 * V;
    const long long i_end = (a.tiles > 0 && tile_end < n_all) ? tile_end : n_all;
    for (long long i = i_first; i < i_end; i += i_step) {
        float ov[NP][V], fv[NF][V], yv[NTG][V], dov[NP][V], yh[NTG][V];
        auto ld = [&](const float* p, float (&dst)[V]) {
            if constexpr (V == 4) { const f32x4 v = *(const f32x4*)(p + i); dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
            else dst[0] = p[i];
        };
        auto st = [&](float* p, const float (&src)[V]) {
            if constexpr (V == 4) *(f32x4*)(p + i) = f32x4{src[0], src[1], src[2], src[3]};
            else p[i] = src[0];
        };
        // every load of the tile is requested before the first use
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (row[j] >= 0) ld(a.o + (long long)row[j] * a.ld, ov[j]);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const unsigned col = (net.forc_col >> (8 * f)) & 255u;
            if (col != 255u) ld(a.frc[col], fv[f]);
            else
#pragma unroll
                for (int e = 0; e < V; ++e) fv[f][e] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t)
            if (t < net.T) ld(a.y[t], yv[t]);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float par[EH_MAX_PARAMS], sg[NP], dydp[EH_MAX_PARAMS], frc[EH_MAX_FORC];
#pragma unroll
            for (int j = NP; j < EH_MAX_PARAMS; ++j) { par[j] = 0.0f; dydp[j] = 0.0f; }
#pragma unroll
            for (int f = NF; f < EH_MAX_FORC; ++f) frc[f] = 0.0f;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                par[j] = cpar[j]; sg[j] = 0.0f; dydp[j] = 0.0f;
                if (row[j] >= 0) {
                    if (net.scale_nn) { const float s = eh_sigmoid(ov[j][e]); par[j] = fmaf(sc[j], s, lo[j]); sg[j] = sc[j] * s * (1.0f - s); }
                    else { par[j] = ov[j][e]; sg[j] = 1.0f; }
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) frc[f] = fv[f][e];
            float yx[2] = {0.0f, 0.0f}, Jx[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
            constexpr bool PROG = MECH == EH_MECH_PROGRAM;
            float pval[PROG ? EH_PROG_SLOTS : 1], y0;
            if constexpr (PROG) {
                eh_prog_forward(a.prog, par, frc, pval);
                y0 = pval[a.prog[2]];
                if (net.n_out > 1) yx[0] = pval[a.prog[3]];
                if (net.n_out > 2) yx[1] = pval[a.prog[4]];
            } else {
                y0 = eh_mech_eval(MECH, par, frc, dydp);
                if constexpr (NO > 1) eh_mech_extra(MECH, par, frc, yx, Jx);
            }
            float dy = 0.0f, dyx[2] = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < NTG; ++t)
                if (t < net.T) {
                    const int oi = NO > 1 ? (int)((net.targ_out >> (2 * t)) & 3u) : 0;
                    const float y = oi == 0 ? y0 : (oi == 1 ? yx[0] : yx[1]);
                    yh[t][e] = y;
                    const float r = __builtin_isnan(yv[t][e]) ? 0.0f : y - yv[t][e];
                    float d;
                    if (mae) { S[t] += fabsf(r); d = r > 0.0f ? w[t] : (r < 0.0f ? -w[t] : 0.0f); }      // mean |r| (loss_fn.jl:64-66)
                    else { S[t] = fmaf(r, r, S[t]); d = 2.0f * w[t] * r; }                              // mean r^2 (loss_fn.jl:61-63)
                    dy += oi == 0 ? d : 0.0f; dyx[0] += oi == 1 ? d : 0.0f; dyx[1] += oi == 2 ? d : 0.0f;
                }
            float padj[PROG ? EH_PROG_SLOTS : 1];
            if constexpr (PROG) {                                // reverse sweep over the tape (what Zygote derives from the closure)
                const int nslot = EH_PROG_SLOT_INSTR + (int)a.prog[0];
                for (int q = 0; q < nslot; ++q) padj[q] = 0.0f;
                padj[a.prog[2]] += dy;
                if (net.n_out > 1) padj[a.prog[3]] += dyx[0];
                if (net.n_out > 2) padj[a.prog[4]] += dyx[1];
                eh_prog_reverse(a.prog, pval, padj);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float dp = PROG ? padj[PROG ? j : 0] : dy * dydp[j];
                if (!PROG && NO > 1 && j < 3) dp += dyx[0] * Jx[0][j] + dyx[1] * Jx[1][j];
                gp[j] += row[j] >= 0 ? 0.0f : dp;
                dov[j][e] = dp * sg[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (row[j] >= 0) st(a.d_o + (long long)row[j] * a.ld, dov[j]);
        if (a.yhat)
#pragma unroll
            for (int t = 0; t < NTG; ++t)
                if (t < net.T) st(a.yhat + (long long)t * a.ld, yh[t]);
    }
    __shared__ float sh[4][EH_MECH_PART];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float v = eh_wave_sum(k < 8 ? (k < NP ? gp[k < NP ? k : 0] : 0.0f) : (k - 8 < NTG ? S[k - 8 < NTG ? k - 8 : 0] : 0.0f));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) a.part[(long long)blockIdx.x * EH_MECH_PART + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// rows of partials -> out[0] = loss, out[1 + j] = d loss / d raw global parameter j (canonical parameter order), fixed order:
// deterministic.  Up to 2 048 rows: eh_mech_finish_kernel alone (one workgroup, 64 row groups x 16 columns, all of a thread's loads
// -- rows written on other XCDs, each a trip to memory -- requested before the first add).  More rows: eh_mech_fold_kernel first,
// whose workgroup g folds rows [g rows_per, (g + 1) rows_per) into row g of a second, <= 64-row table for the finish kernel.  (One
// launch with a ticket for the last workgroup was measured and lost: 13.6 us at 4 096 rows, 37 us at 65 536 -- the release in front
// of the ticket writes back an L2 -- against 5.1-5.7 us for the plain one-workgroup kernel; profiles/r03/mech_stage_ab.txt.)
__device__ __forceinline__ float eh_mech_fold_rows(const float* part, int r0, int r1, float (*red)[EH_MECH_PART]) {
    const int k = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int rb = r0 + grp; rb < r1; rb += 64 * 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = rb + 64 * u < r1 ? part[(long long)(rb + 64 * u) * EH_MECH_PART + k] : 0.0f;
#pragma unroll
        for (int u = 0; u < 32; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
    }
    red[grp][k] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float s = 0.0f;
    if (grp == 0)
        for (int g2 = 0; g2 < 64; ++g2) s += red[g2][k];
    return s;                                                    // (threads 0..15 hold column k's sum)
}
__global__ __launch_bounds__(1024) void eh_mech_fold_kernel(const float* part, int nblk, int rows_per, float* part2) {
    __shared__ float red[64][EH_MECH_PART];
    const int r0 = blockIdx.x * rows_per;
    const float s = eh_mech_fold_rows(part, r0, min(nblk, r0 + rows_per), red);
    if (threadIdx.x < EH_MECH_PART) part2[blockIdx.x * EH_MECH_PART + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void eh_mech_finish_kernel(const float* part, int nblk, const EhNet net, const float* meta, const EhMechArgs a, float* out) {
    __shared__ float red[64][EH_MECH_PART];
    const int k = threadIdx.x & 15;
    const float s = eh_mech_fold_rows(part, 0, nblk, red);
    if (threadIdx.x >= EH_MECH_PART) return;
    __shared__ float St[EH_MAX_TARG];
    if (k >= 8 && k < 12) {
        const unsigned long long c = a.use_v ? a.counts_v[k - 8] : a.counts[k - 8];
        St[k - 8] = (k - 8 < net.T && c > 0) ? a.agg_a * s / (float)c : 0.0f;
    }
    else if (k < 8) out[1 + k] = (k < net.n_par && ((net.par_kind >> (2 * k)) & 3u) == EH_PAR_GLOBAL) ? s * meta[EH_IMG_DPHI + k] : 0.0f;
    EH_WAVE_SYNC();                                              // (the 16 threads left are lanes of one wave)
    if (k == 0) out[0] = (St[0] + St[1]) + (St[2] + St[3]);
}

// Sum the per-workgroup partials of the step kernel (fixed order: deterministic), normalise by the
// valid count when the step ran with deferred normalisation, and (APPLY) update theta and its
// image in place.  Block = CW columns x 256/CW row groups (CW = 16 for small models: many blocks; CW = 64 for
// big gradients: 256-byte runs per slab row); every load of a thread is independent, so
// the whole slab read costs about one L2 round trip.  gradbuf = [grad | loss | counts].
template <bool APPLY, int CW, int NC = 1, bool VEC = false>
__global__ __launch_bounds__(256) void eh_reduce_kernel(const float* __restrict__ slab, int nblk, int n_acc, int n_theta, int T, int deferred,
                                                        float* __restrict__ gradbuf, float* theta, float* m, float* v, const float* sc_in,
                                                        float* sc_out, EhOpt o, float* loss_slot, EhImg im, int loss_kind, const float* mom, const float* l2val, unsigned tp_mask) {
    // NC > 1 (layer-wise form: hundreds of thousands of columns under a few rows): NC columns per thread -- a quarter of the
    // workgroups, so a quarter of the per-workgroup part (counts, barriers), and NC times the loads in flight per thread.
    // VEC (NC == 4): the thread's columns are four CONSECUTIVE ones, moved as 16-byte pieces wherever all four are parameters (slab
    // rows are only 4-byte aligned: n_acc is any number -- global memory takes unaligned 16-byte accesses); otherwise CW apart.
    constexpr int NQ = 256 / CW;
    static_assert(NC == 1 || NQ == 1, "several columns per thread: one row group");
    static_assert(!VEC || NC == 4, "16-byte pieces");
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    __shared__ float part[NQ][CW + 1];
    __shared__ float wsum[4][EH_MAX_TARG + 3];
    const int tid = threadIdx.x, p = tid % CW, q = tid / CW;
    constexpr int CS = VEC ? 1 : CW;                       // distance between a thread's columns
    const int idx0 = VEC ? (blockIdx.x * CW + p) * NC : blockIdx.x * (CW * NC) + p;
    const bool full = VEC && idx0 + NC - 1 < n_theta;     // all of the thread's columns are parameters: the 16-byte path
    // optimiser inputs are independent of the slab: request them first so they arrive together (only the state the rule keeps:
    // RMSProp has no first moment, Descent none at all)
    const bool use_m = o.rule == EH_OPT_ADAM || o.rule == EH_OPT_ADAMW, use_v = use_m || o.rule == EH_OPT_RMSPROP;
    float th[NC], mm[NC], vv[NC], bt1 = 0.0f, bt2 = 0.0f;
    int mp[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) { th[j] = 0.0f; mm[j] = 0.0f; vv[j] = 0.0f; mp[j] = -1; }
    if (VEC && full) {
        if constexpr (VEC) {
            if (APPLY || l2val) { const f32x4u t4 = *reinterpret_cast<const f32x4u*>(theta + idx0); th[0] = t4[0]; th[1] = t4[1]; th[2] = t4[2]; th[3] = t4[3]; }
            if (APPLY && use_m) { const f32x4u t4 = *reinterpret_cast<const f32x4u*>(m + idx0); mm[0] = t4[0]; mm[1] = t4[1]; mm[2] = t4[2]; mm[3] = t4[3]; }
            if (APPLY && use_v) { const f32x4u t4 = *reinterpret_cast<const f32x4u*>(v + idx0); vv[0] = t4[0]; vv[1] = t4[1]; vv[2] = t4[2]; vv[3] = t4[3]; }
            if (APPLY && im.imap) {
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    if (idx0 + j < im.g_off) mp[j] = im.imap[idx0 + j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int idx = idx0 + CS * j;
            if (APPLY && q == 0 && idx < n_theta) {
                th[j] = theta[idx];
                if (use_m) mm[j] = m[idx];
                if (use_v) vv[j] = v[idx];
                if (idx < im.g_off && im.imap) mp[j] = im.imap[idx];
            }
            if (!APPLY && l2val && q == 0 && idx < n_theta) th[j] = theta[idx];
        }
    }
    if (APPLY && q == 0) { bt1 = sc_in[0]; bt2 = sc_in[1]; }
    // (all of a thread's rows in flight at once: the step is one memory round trip, not nblk / NQ / 16 of them)
    float s[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) s[j] = 0.0f;
    if (VEC && full) {
        if constexpr (VEC) {
            constexpr int U = 8;
            const float* col = slab + idx0;
            int r = 0;
            for (; r + U - 1 < nblk; r += U) {
                f32x4u t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = *reinterpret_cast<const f32x4u*>(col + (size_t)u * n_acc);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < NC; ++j) s[j] += t[u][j];
                col += (size_t)U * n_acc;
            }
            for (; r < nblk; ++r) {
                const f32x4u t4 = *reinterpret_cast<const f32x4u*>(col);
#pragma unroll
                for (int j = 0; j < NC; ++j) s[j] += t4[j];
                col += n_acc;
            }
        }
    } else if (idx0 < n_acc) {
        constexpr int U = 32 / NC;
        const float* col[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) col[j] = slab + (size_t)q * n_acc + min(idx0 + CS * j, n_acc - 1);       // (columns past the end: clamped, dropped below)
        const size_t rstride = (size_t)NQ * n_acc;
        int r = q;
        for (; r + (U - 1) * NQ < nblk; r += U * NQ) {
            float t[U][NC];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NC; ++j) t[u][j] = col[j][u * rstride];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NC; ++j) s[j] += t[u][j];
#pragma unroll
            for (int j = 0; j < NC; ++j) col[j] += U * rstride;
        }
#pragma unroll 4
        for (; r < nblk; r += NQ) {
#pragma unroll
            for (int j = 0; j < NC; ++j) { s[j] += *col[j]; col[j] += rstride; }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (idx0 + CS * j >= n_acc) s[j] = 0.0f;
    }
    if constexpr (NC == 1) part[q][p] = s[0];
    // valid counts: every block needs them (nblk <= 256: one row per thread)
    // wsum columns: [0..3] n_valid per target, [4] S, [5] Sy, [6] Syy
    // (a row's scalars [S | n_t ... | Sy | Syy] are 3 + T consecutive floats behind the gradient: two wide loads per row -- one
    //  instruction per column made every workgroup walk 7 x 256 cache lines and cost 2-3 us of a launch, tools/ubench/reduce.hip;
    //  4-byte aligned only, and up to 3 floats past a row's end when T < 4: the slab is allocated with that much slack)
    float cs[EH_MAX_TARG + 3];
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG + 3; ++t) cs[t] = 0.0f;
    for (int r = tid; r < nblk; r += 256) {
        const float* const row = slab + (size_t)r * n_acc + n_theta;
        const f32x4u lo = *reinterpret_cast<const f32x4u*>(row);
        const f32x3u hi = *reinterpret_cast<const f32x3u*>(row + 4);
        const float f[7] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2]};
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) cs[t] += t < T ? f[1 + t] : 0.0f;
        cs[EH_MAX_TARG] += f[0];
#pragma unroll
        for (int t = 1; t <= EH_MAX_TARG; ++t)
            if (t == T) { cs[EH_MAX_TARG + 1] += f[1 + t]; cs[EH_MAX_TARG + 2] += f[2 + t]; }
    }
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG + 3; ++t) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cs[t] += __shfl_xor(cs[t], off, 64);
        if ((tid & 63) == 0) wsum[tid >> 6][t] = cs[t];
    }
    __syncthreads();
    float cnts[EH_MAX_TARG + 3], ntot = 0.0f;
#pragma unroll
    for (int t = 0; t < EH_MAX_TARG + 3; ++t) {
        cnts[t] = (wsum[0][t] + wsum[1][t]) + (wsum[2][t] + wsum[3][t]);
        if (t < EH_MAX_TARG) ntot += cnts[t];
    }
    float dscale = 1.0f, dloss = 0.0f;
    if (deferred) eh_loss_finish(loss_kind, cnts[EH_MAX_TARG], cnts[0], cnts[EH_MAX_TARG + 1], cnts[EH_MAX_TARG + 2], dscale, dloss, im.agg_a);
    if (deferred && mom && cnts[0] > 0.0f) { dscale = 1.0f; dloss = mom[7]; }      // moment-based loss: per-sample weights were exact, value from eh_moment_coef_kernel
    float tp_loss = 0.0f;                    // multi-target: the targets whose loss came out of the coefficient kernel (the others' terms are in the loss sum)
    if (!deferred && mom) {
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) tp_loss += (t < T && ((tp_mask >> t) & 1u)) ? mom[EH_TT * t + 7] : 0.0f;
    }
    if (VEC && full) {
        if constexpr (VEC) {
            const float scale = deferred ? dscale : 1.0f;
            f32x4u g4;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                float g = s[j] * scale;
                if (l2val && ntot > 0.0f) { const float c2 = eh_l2_coef(im, idx0 + j); if (c2 != 0.0f) g = fmaf(2.0f * c2, th[j], g); }
                g4[j] = g;
                if (APPLY && ntot > 0.0f) eh_opt_update(o, g, bt1, bt2, th[j], mm[j], vv[j]);
            }
            *reinterpret_cast<f32x4u*>(gradbuf + idx0) = g4;
            if (APPLY && ntot > 0.0f) {
                *reinterpret_cast<f32x4u*>(theta + idx0) = f32x4u{th[0], th[1], th[2], th[3]};
                if (use_m) *reinterpret_cast<f32x4u*>(m + idx0) = f32x4u{mm[0], mm[1], mm[2], mm[3]};
                if (use_v) *reinterpret_cast<f32x4u*>(v + idx0) = f32x4u{vv[0], vv[1], vv[2], vv[3]};
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if (idx0 + j < im.g_off) { if (mp[j] >= 0) im.image[mp[j]] = th[j]; }
                    else eh_image_store(im, idx0 + j, th[j]);
                }
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int idx = idx0 + CS * j;
        if (q == 0 && idx < n_acc) {
            float tot = 0.0f;
            if constexpr (NC == 1) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) tot += part[k][p];
            } else tot = s[j];
            const float scale = deferred ? dscale : 1.0f;
            if (idx < n_theta) {
                float g = tot * scale;
                if (l2val && ntot > 0.0f) { const float c2 = eh_l2_coef(im, idx); if (c2 != 0.0f) g = fmaf(2.0f * c2, th[j], g); }      // + d/dw (l2c * sum w^2)
                gradbuf[idx] = g;
                if (APPLY && ntot > 0.0f) {
                    eh_opt_update(o, g, bt1, bt2, th[j], mm[j], vv[j]);
                    theta[idx] = th[j];
                    if (use_m) m[idx] = mm[j];
                    if (use_v) v[idx] = vv[j];
                    if (idx < im.g_off) { if (mp[j] >= 0) im.image[mp[j]] = th[j]; }
                    else eh_image_store(im, idx, th[j]);
                }
            } else if (idx == n_theta) {
                const float loss = ntot > 0.0f ? (deferred ? dloss : tot + tp_loss) + (l2val ? *l2val : 0.0f) : __builtin_nanf("");      // agg = sum([loss, extra...]), compute_loss.jl:31-34
                gradbuf[idx] = loss;
                if (loss_slot) *loss_slot = loss;
            } else {
                gradbuf[idx] = tot;
            }
        }
    }
    }
    if (APPLY && blockIdx.x == 0 && tid == 0) {
        sc_out[0] = ntot > 0.0f ? sc_in[0] * o.b1 : sc_in[0];
        sc_out[1] = ntot > 0.0f ? sc_in[1] * o.b2 : sc_in[1];
    }
}

// fused-update mode: apply the still-pending gradient (sharded accumulator g_prev) in place
__global__ __launch_bounds__(256) void eh_fused_flush_kernel(const float* g_prev, int n_acc, int n_theta, float* theta, float* m, float* v,
                                                             const float* sc_in, float* sc_out, EhOpt o, float* loss_slot, EhImg im, int loss_kind,
                                                             const EhP2P* p2p, int slot, unsigned seq, int T) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float cnt = 0.0f, sse = 0.0f, sy = 0.0f, syy = 0.0f, gs_p2p = 0.0f;
    // everything this thread will need is requested before the first value is looked at: ONE memory round trip (the update used to wait
    // for the counts before it asked for the parameter, and the image store for its map entry after that: three -- this kernel sits
    // between the last step and the host's synchronisation, 5.7 us of a 20-step timed run)
    const bool own = idx < n_theta;
    float gsv[EH_GSHARDS], th0 = 0.0f, mm0 = 0.0f, vv0 = 0.0f;
    int mp0 = -1;
#pragma unroll
    for (int sh = 0; sh < EH_GSHARDS; ++sh) gsv[sh] = (own && !p2p) ? g_prev[sh * n_acc + idx] : 0.0f;
    if (own) { th0 = theta[idx]; mm0 = m[idx]; vv0 = v[idx]; if (idx < im.g_off && im.imap) mp0 = im.imap[idx]; }
    const float sc0 = sc_in[0], sc1 = sc_in[1];
    if (p2p) {       // every rank's sums of the last step, straight from the receive shards (see EhP2P)
        // mode 1: nobody has published the last step yet -- block 0 does it for the peers; this rank's own sums come from its staging shards
        const bool own_direct = p2p->mode == 1;
        if (own_direct && blockIdx.x == 0) eh_p2p_fold_store(p2p, slot, seq, n_acc, (int)threadIdx.x, 256, true);
        // (five scalars behind the gradient, [S | n_1 .. | Sy | Syy], as the step's prologue reads them; the fifth only where it is a count;
        //  the own sums folded as the publishing workgroup folds them -- eh_fold8)
        float ownv[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (own_direct) {
            const float* const st = p2p->stage + (long long)slot * EH_GSHARDS * n_acc;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k == 4 && T <= 3) continue;
                if (k == 5 && !own) continue;
                const float* q = st + (k < 5 ? n_theta + k : idx);
                ownv[k] = eh_fold8(q[0], q[n_acc], q[2 * n_acc], q[3 * n_acc], q[4 * n_acc], q[5 * n_acc], q[6 * n_acc], q[7 * n_acc]);
            }
        }
        auto ad = [&](int i) -> const unsigned long long* {
            const int sh = i / 6, k = i % 6;
            if (sh >= p2p->world || (k == 5 && idx >= n_theta) || (k == 4 && T <= 3) || (own_direct && sh == p2p->rank)) return nullptr;
            const unsigned long long* base = p2p->peer_recv[p2p->rank] + ((long long)slot * EH_GSHARDS + sh) * n_acc;
            return k < 5 ? base + n_theta + k : base + idx;
        };
        unsigned long long w[6 * EH_GSHARDS];
        float got[6 * EH_GSHARDS];
        eh_ll_issue(ad, seq, w);
        eh_ll_finish(p2p, ad, seq, w, got);
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
#pragma unroll
        for (int sh = 0; sh < EH_GSHARDS; ++sh) {          // (rank order on every rank: bitwise-identical replicas)
            const bool me = own_direct && sh == p2p->rank;
            sse += me ? ownv[0] : got[6 * sh]; s1 += me ? ownv[1] : got[6 * sh + 1]; s2 += me ? ownv[2] : got[6 * sh + 2];
            s3 += me ? ownv[3] : got[6 * sh + 3]; s4 += me ? ownv[4] : got[6 * sh + 4]; gs_p2p += me ? ownv[5] : got[6 * sh + 5];
        }
        cnt = s1 + (T > 1 ? s2 : 0.0f) + (T > 2 ? s3 : 0.0f) + (T > 3 ? s4 : 0.0f); sy = s2; syy = s3;
    } else {
        // the eight shards of every scalar in the order the step's prologue folds them (eh_fold8): a step applied here and the same step
        // applied by the next step's prologue are the same bits, so a trajectory does not depend on when the host drains (advisor r05)
        static_assert(EH_GSHARDS == 8, "eh_fold8");
        float S[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k == 4 && T <= 3) continue;
            const float* q = g_prev + n_theta + k;
            S[k] = eh_fold8(q[0], q[n_acc], q[2 * n_acc], q[3 * n_acc], q[4 * n_acc], q[5 * n_acc], q[6 * n_acc], q[7 * n_acc]);
        }
        sse = S[0]; cnt = S[1] + (T > 1 ? S[2] : 0.0f) + (T > 2 ? S[3] : 0.0f) + (T > 3 ? S[4] : 0.0f); sy = S[2]; syy = S[3];
    }
    float inv = 0.0f, lossv = 0.0f;
    if (T == 1) eh_loss_finish(loss_kind, sse, cnt, sy, syy, inv, lossv, im.agg_a);
    else { inv = cnt > 0.0f ? 1.0f : 0.0f; lossv = cnt > 0.0f ? sse : __builtin_nanf(""); }      // multi-target: the step used exact per-target weights
    float th = th0;
    if (own && cnt > 0.0f) {
        const float gs = p2p ? gs_p2p : eh_fold8(gsv[0], gsv[1], gsv[2], gsv[3], gsv[4], gsv[5], gsv[6], gsv[7]);
        float mm = mm0, vv = vv0;
        eh_opt_update(o, gs * inv, sc0, sc1, th, mm, vv);
        theta[idx] = th; m[idx] = mm; v[idx] = vv;
    }
    if (own) {
        if (idx < im.g_off) { if (mp0 >= 0) im.image[mp0] = th; }
        else eh_image_store(im, idx, th);
    }
    if (idx == 0) {
        sc_out[0] = cnt > 0.0f ? sc0 * o.b1 : sc0;
        sc_out[1] = cnt > 0.0f ? sc1 * o.b2 : sc1;
        if (loss_slot) *loss_slot = lossv;
    }
}


// Moment-based training losses (pearsonLoss, kgeLoss, pbkgeLoss; src/losses/loss_fn.jl:75-77,105-174): from the batch
// moments the forward-only pass left in the slab ([blocks][EH_EVAL_STATS], shifted by c) to the loss value and the
// coefficients of  d loss / d yhat_i = k0 + k1 (yhat_i - cu) + k2 (y_i - c).  out = [1, cu, -, -, k0, k1, k2, loss].
// With u = yhat - cu (cu = batch mean of yhat, from a first forward pass), w = y - c:  r = Suw_c / sqrt(Suu_c Sww_c),  alpha = sqrt(Suu_c / Sww_c) (the n-1 of std cancels),
// beta = mean(yhat) / mean(y);  dr/du_i = (w_i - mw) / sqrt(Suu_c Sww_c) - r (u_i - mu) / Suu_c,
// dalpha/du_i = (u_i - mu) / (alpha Sww_c),  dbeta/du_i = 1 / (n mean(y)).
// stage 0: the centre of yhat for the moment pass proper = its batch mean (sum (yhat - c) is accurate; the squares are not)
struct EhShift4 { float c[EH_MAX_TARG]; };
// (one workgroup per target; targets whose loss needs no batch statistics of yhat are left alone)
__device__ __forceinline__ bool eh_kind_two_pass(unsigned kind, int T) {
    return (kind >= (unsigned)EH_LOSS_PEARSONLOSS && kind <= (unsigned)EH_LOSS_PBKGELOSS) || (kind == (unsigned)EH_LOSS_RMSE && T > 1);
}
__global__ __launch_bounds__(64) void eh_moment_centre_kernel(const float* slab, int nblk, int T, unsigned loss_t, EhShift4 shift, float* tt) {
    __shared__ double tot[EH_EVAL_STATS];
    const int tid = threadIdx.x, t = blockIdx.x;
    if (!eh_kind_two_pass((loss_t >> (4 * t)) & 15u, T)) return;
    if (tid < EH_EVAL_STATS) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)slab[(b * T + t) * EH_EVAL_STATS + tid];
        tot[tid] = s;
    }
    __syncthreads();
    if (tid == 0) tt[EH_TT * t + 1] = tot[3] > 0.0 ? (float)((double)shift.c[t] + tot[4] / tot[3]) : shift.c[t];
}
// stage 1: moments with u = yhat - tt[1], w = y - shift  ->  tt[4..6] = k0, k1, k2 ; tt[7] = the target's loss value
// (rmse, the one loss without batch moments here, takes this form on multi-target models, where its scale 1 / (n rmse) has to be
// known inside the one streaming pass that serves all targets: d/dyhat_i = (yhat_i - y_i) / (n rmse), loss_fn.jl:58-60)
__global__ __launch_bounds__(64) void eh_moment_coef_kernel(const float* slab, int nblk, int T, unsigned loss_t, EhShift4 shift4, float* tt_all, float agg_a) {
    __shared__ double tot[EH_EVAL_STATS];
    const int tid = threadIdx.x, t = blockIdx.x;
    const int kind = (int)((loss_t >> (4 * t)) & 15u);
    if (!eh_kind_two_pass((unsigned)kind, T)) return;
    float* const out = tt_all + EH_TT * t;
    const float shift = shift4.c[t];
    if (tid < EH_EVAL_STATS) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += (double)slab[(b * T + t) * EH_EVAL_STATS + tid];
        tot[tid] = s;
    }
    __syncthreads();
    if (tid != 0) return;
    const double n = tot[3], Sw = tot[1], Sww = tot[2], Su = tot[4], Suu = tot[5], Suw = tot[6], cu = (double)out[1];
    float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f, loss = T > 1 ? 0.0f : __builtin_nanf("");      // (a target without a valid sample adds nothing to a multi-target loss)
    if (n > 0.0 && kind == EH_LOSS_RMSE) {
        const double rm = sqrt(tot[0] / n), q = rm > 0.0 ? 1.0 / (n * rm) : 0.0;            // d = q (yhat - y) = q (u + cu) - q (w + shift)
        k1 = (float)q; k2 = (float)-q; k0 = (float)(q * (cu - (double)shift));
        loss = (float)rm;
    } else if (n > 0.0) {
        const double mu = Su / n, mw = Sw / n;
        const double Suu_c = Suu - Su * Su / n, Sww_c = Sww - Sw * Sw / n, Suw_c = Suw - Su * Sw / n;
        const double den = sqrt(Suu_c * Sww_c), r = Suw_c / den;
        // dr = a_u (u_i - mu) + a_w (w_i - mw)
        const double a_u = -r / Suu_c, a_w = 1.0 / den;
        double g_r, g_a = 0.0, g_b = 0.0, L;
        if (kind == EH_LOSS_PEARSONLOSS) { L = 1.0 - r; g_r = -1.0; }
        else {
            const double alpha = sqrt(Suu_c / Sww_c), beta = (cu + mu) / ((double)shift + mw);
            if (kind == EH_LOSS_KGELOSS) { L = sqrt((r - 1) * (r - 1) + (alpha - 1) * (alpha - 1) + (beta - 1) * (beta - 1)); g_a = (alpha - 1) / L / (alpha * Sww_c); }
            else L = sqrt((r - 1) * (r - 1) + (beta - 1) * (beta - 1));
            g_r = (r - 1) / L;
            g_b = (beta - 1) / L / (n * ((double)shift + mw));
        }
        const double qu = g_r * a_u + g_a, qw = g_r * a_w;        // coefficients of (u_i - mu), (w_i - mw)
        k1 = (float)qu; k2 = (float)qw; k0 = (float)(g_b - qu * mu - qw * mw);
        loss = (float)L;
    }
    out[0] = 1.0f; out[4] = k0 * agg_a; out[5] = k1 * agg_a; out[6] = k2 * agg_a; out[7] = loss * agg_a;      // (agg_a: the factor of `agg` on the data loss, EhImg)
}

// data parallel, two-pass losses (eh_dp_moments): this shard's per-workgroup moment rows [nblk][T][EH_EVAL_STATS] of a forward-only
// pass -> one row [T][EH_EVAL_STATS] (EH_BUF_MOMENT), which the caller all-reduces
__global__ __launch_bounds__(64) void eh_moment_fold_kernel(const float* slab, int nblk, int T, float* out) {
    const int tid = threadIdx.x;
    if (tid >= T * EH_EVAL_STATS) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)slab[b * T * EH_EVAL_STATS + tid];
    out[tid] = (float)s;
}

// data-parallel tail: gradbuf holds the all-reduced RAW sums [grad | sse | count]
// mom / tp_mask: the per-target table and the targets with a two-pass loss (their exact in-pass weights came from the all-reduced
// moments of the GLOBAL batch, eh_dp_moments; their loss values sit in the table)
__global__ __launch_bounds__(256) void eh_apply_kernel(float* gradbuf, int n_theta, float* theta, float* m, float* v, const float* sc_in,
                                                       float* sc_out, EhOpt o, float* loss_slot, EhImg im, int loss_kind, int T, const float* l2val,
                                                       const float* mom, unsigned tp_mask) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float cnt = gradbuf[n_theta + 1];
    float scale = 0.0f, lossv = 0.0f;
    if (T == 1 && !tp_mask) eh_loss_finish(loss_kind, gradbuf[n_theta], cnt, gradbuf[n_theta + 2], gradbuf[n_theta + 3], scale, lossv, im.agg_a);
    else {      // multi-target: the shards used the weights of the global batch (eh_dp_counts): the all-reduced sums are final
        for (int t = 1; t < T; ++t) cnt += gradbuf[n_theta + 1 + t];
        scale = cnt > 0.0f ? 1.0f : 0.0f;
        lossv = cnt > 0.0f ? gradbuf[n_theta] : __builtin_nanf("");
        if (mom && cnt > 0.0f)
            for (int t = 0; t < T; ++t) lossv += ((tp_mask >> t) & 1u) ? mom[EH_TT * t + 7] : 0.0f;
    }
    if (l2val && cnt > 0.0f) lossv += *l2val;                  // + lambda * weight_l2 of the (replicated) parameters: agg = sum([loss, extra...]), compute_loss.jl:31-34
    if (idx < n_theta && cnt > 0.0f) {
        float g = gradbuf[idx] * scale;
        float th = theta[idx], mm = m[idx], vv = v[idx];
        if (l2val) { const float c2 = eh_l2_coef(im, idx); if (c2 != 0.0f) g = fmaf(2.0f * c2, th, g); }
        eh_opt_update(o, g, sc_in[0], sc_in[1], th, mm, vv);
        theta[idx] = th; m[idx] = mm; v[idx] = vv;
        eh_image_store(im, idx, th);
    }
    if (idx == 0) {
        sc_out[0] = cnt > 0.0f ? sc_in[0] * o.b1 : sc_in[0];
        sc_out[1] = cnt > 0.0f ? sc_in[1] * o.b2 : sc_in[1];
        if (loss_slot) *loss_slot = lossv;
    }
}

// Per-target weight of one batch (multi-target models: the normaliser differs per target, so it has to be known before the pass):
// the residual terms of target t enter the loss as w_t r^2 (w_t |r| for MAE) with  w_t = 1 / n_t  for mse / mae  (loss_fn.jl:61-66)
// and  w_t = 1 / sum (y - mean y)^2  for nseLoss (:79-81) -- all of it a function of the targets alone.  One workgroup per target.
__global__ __launch_bounds__(256) void eh_count_kernel(const float* recs, int C, int toff, int T, const int* idx, long long first, long long count,
                                                       float* inv_n, unsigned loss_t, EhShift4 shift, float agg_a, unsigned roles, float l2s, float* raw = nullptr) {
    __shared__ float red[3][256];
    const int t = blockIdx.x;
    float c = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (long long i = threadIdx.x; i < count; i += 256) {
        const long long n = idx ? (long long)idx[first + i] : first + i;
        const float y = recs[n * C + toff + t];
        if (!__builtin_isnan(y)) { const float d = y - shift.c[t]; c += 1.0f; s1 += d; s2 += d * d; }
    }
    red[0][threadIdx.x] = c; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; red[2][threadIdx.x] += red[2][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = red[0][0];
        if (raw) { raw[3 * t] = n; raw[3 * t + 1] = red[1][0]; raw[3 * t + 2] = red[2][0]; return; }      // data parallel: this shard's sums (EH_BUF_TCOUNT), all-reduced by the caller
        float w = n > 0.0f ? 1.0f / n : 0.0f;
        if (((loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_NSELOSS && n > 0.0f) w = 1.0f / (red[2][0] - red[1][0] * red[1][0] / n);
        // (the per-target table of EhStepArgs::inv_n, eh_device.hpp; agg_a: the factor of `agg`, EhImg.  A target that stands for an ENTRY OF
        //  THE EXTRA LOSS -- eh_set_target_roles: a recorded function of the predictions, summed or averaged over all samples -- takes the
        //  extra loss's factor instead, and no 1 / n when it is a sum)
        const unsigned role = (roles >> (2 * t)) & 3u;
        inv_n[EH_TT * t] = role == 0u ? w * agg_a : (role == 2u ? (n > 0.0f ? l2s : 0.0f) : w * l2s);
    }
}
// (data parallel) the all-reduced sums [n_t | sum (y - c) | sum (y - c)^2] of the GLOBAL batch -> the per-target weights; c is common to the ranks (eh_set_target_shift)
__global__ void eh_weights_from_counts_kernel(const float* raw, int T, unsigned loss_t, float* inv_n, float agg_a, unsigned roles, float l2s) {
    const int t = threadIdx.x;
    if (t >= T) return;
    const float n = raw[3 * t];
    float w = n > 0.0f ? 1.0f / n : 0.0f;
    if (((loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_NSELOSS && n > 0.0f) w = 1.0f / (raw[3 * t + 2] - raw[3 * t + 1] * raw[3 * t + 1] / n);
    const unsigned role = (roles >> (2 * t)) & 3u;          // (see eh_count_kernel)
    inv_n[EH_TT * t] = role == 0u ? w * agg_a : (role == 2u ? (n > 0.0f ? l2s : 0.0f) : w * l2s);
}

// input BatchNorm: per-workgroup partial sums of one minibatch, shifted by the batch's first sample
// against cancellation.  part = [gridDim][64] (sum (x-c) | sum (x-c)^2 per predictor) then c[32].
// cshift != nullptr: shift by that common vector instead (sums of different GPUs must share their shift).
__global__ __launch_bounds__(1024) void eh_bn_stats_kernel(const float* recs, int C, int P, const int* idx, int first, int count, float* part,
                                                            const float* cshift) {
    __shared__ float red[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = idx ? idx[first] : first;
    float s1[32], s2[32];
#pragma unroll
    for (int p = 0; p < 32; ++p) { s1[p] = 0.0f; s2[p] = 0.0f; }
    for (int i = blockIdx.x * 1024 + tid; i < count; i += gridDim.x * 1024) {
        const int n = idx ? idx[first + i] : first + i;
        const float* r = recs + (long long)n * C;
        const float* r0 = cshift ? cshift : recs + (long long)n0 * C;
#pragma unroll
        for (int p = 0; p < 32; ++p)
            if (p < P) { const float d = r[p] - r0[p]; s1[p] += d; s2[p] += d * d; }
    }
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        if (p < P) {
            float a = s1[p], b = s2[p];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (lane == 0) { red[wave][p] = a; red[wave][32 + p] = b; }
        }
    }
    __syncthreads();
    if (tid < 64) {
        float a = 0.0f;
        if ((tid & 31) < P)
            for (int w = 0; w < 16; ++w) a += red[w][tid];
        part[blockIdx.x * 64 + tid] = a;
        if (blockIdx.x == 0 && tid < 32) part[gridDim.x * 64 + tid] = tid < P ? (cshift ? cshift[tid] : recs[(long long)n0 * C + tid]) : 0.0f;
    }
}

// cross-GPU statistics: fold the workgroup partials of this GPU's shard into stat = [sum d (32) | sum d^2 (32) | n]
__global__ __launch_bounds__(64) void eh_bn_fold_kernel(const float* part, int nblk, int count, float* stat) {
    const int tid = threadIdx.x;
    float a = 0.0f;
    for (int b = 0; b < nblk; ++b) a += part[b * 64 + tid];
    stat[tid] = a;
    if (tid == 0) stat[64] = (float)count;
}

// keyed bijection of [0, n): 4-round Feistel network on 2*hb bits, cycle-walked into range.
__host__ __device__ inline uint32_t eh_mix32(uint32_t x, uint32_t k) {
    x ^= k; x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t eh_perm32(uint32_t i, uint32_t n, int hb, uint64_t seed) {
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t x = i;
    do {
        uint32_t L = x >> hb, R = x & mask;
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = (uint32_t)(seed >> (16 * (r & 1))) + 0x632BE5ABu * (uint32_t)(r + 1) + (uint32_t)(seed >> 32);
            const uint32_t t = L ^ (eh_mix32(R, k) & mask);
            L = R; R = t;
        }
        x = (L << hb) | R;
    } while (x >= n);
    return x;
}
// "check_idx" option (debug): how many of idx[first .. first + count) lie outside [0, n), and the position of the first one
__global__ __launch_bounds__(256) void eh_idx_check_kernel(const int* idx, long long first, long long count, long long n, unsigned* bad) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const long long v = idx[first + i];
        if (v < 0 || v >= n) { atomicAdd(&bad[0], 1u); atomicMin(&bad[1], (unsigned)i); }
    }
}
__global__ void eh_perm_kernel(int* idx, uint32_t n, int hb, uint64_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)eh_perm32(i, n, hb, seed);
}

// (P x N col-major predictors, F forcing arrays, T target arrays) -> N records of C floats
struct EhPackArgs {
    const float* x;
    const float* forc[EH_MAX_FORC];
    const float* targ[EH_MAX_TARG];
};
__global__ void eh_pack_kernel(EhPackArgs a, float* recs, long long n, int P, int F, int T, int planes) {
    const int C = P + F + T;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * C) return;
    const long long s = e / C;
    const int j = (int)(e % C);
    float v;
    if (j < P) v = planes ? a.x[(long long)j * n + s] : a.x[s * P + j];
    else if (j < P + F) v = a.forc[j - P][s];
    else v = a.targ[j - P - F][s];
    recs[e] = v;
}

