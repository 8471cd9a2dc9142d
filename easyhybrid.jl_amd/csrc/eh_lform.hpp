// Layer-wise execution form ("L-form") of the training step, for networks the fused kernels cannot hold: hidden widths above 128,
// more than three hidden layers -- e.g. the reference's own GPU tutorial net hidden_layers = [1024, 512, 256, 128, 64]
// (docs/literate/tutorials/synthetic_respiration_gpu.jl:87).  A weight image of that size (2.8 MB) does not fit a CU's LDS, so the
// activations live in HBM as [sample][feature] planes and every Dense layer is one tiled GEMM on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) with its elementwise tail fused into the epilogue:
//
//   forward    H_l  [B x out]  = act(H_{l-1} [B x in] * W_l^T + b_l)         (last layer: O^T [K][B] = ... + b, no activation)
//   mechanistic stage + masked loss + its pullback, one sample per lane (eh_mech_stage_lane, the code of the fused kernels)
//   backward   dW_l^T [in x out] = H_{l-1}^T * dZ_l      split over the samples into S partial slabs (-> eh_reduce_kernel + optimiser)
//              db_l   [out]      = column sums of dZ_l   (same split)
//              dZ_{l-1} [B x in] = (dZ_l * W_l) .* act'(H_{l-1})
//
// The canonical flat theta of the reference (ComponentArray of Lux Dense parameters: weight column-major (out, in), then bias,
// GenericHybridModel.jl:236-256) IS the [in][out] row-major operand these GEMMs want, and dW^T comes out in the same order: no
// parameter image, no index maps.  Same arithmetic as the fused kernels (fp32 products, fp32 sums; only the summation order
// differs), same slab / gradbuf contract, so reduce + optimiser, the data-parallel seam and the epoch driver are shared.
#pragma once
#include "eh_wide_bf16.hpp"      // EhMechAcc, eh_mech_stage_lane

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EH_GEPI_STORE = 0, EH_GEPI_BIAS_ACT = 1, EH_GEPI_BIAS_T = 2, EH_GEPI_DACT = 3, EH_GEPI_APPLY = 4 };

struct EhGemmArgs {
    const float* A; long long lda;       // !ATR: A[m][k] at A + m*lda + k ; ATR: A[k][m] at A + k*lda + m
    const float* B; long long ldb;       // !BTR: B[k][n] at B + k*ldb + n ; BTR: B[n][k] at B + n*ldb + k
    float* C; long long ldc;             // C[m][n] at C + m*ldc + n  (EH_GEPI_BIAS_T: C[n][m] at C + n*ldc + m)
    int M, N, K;
    int kchunk;                          // split-K: blockIdx.z covers k in [z*kchunk, min(K, (z+1)*kchunk)), C advanced by z * c_zstride
    long long c_zstride;
    const float* bias;                   // EH_GEPI_BIAS_*: [N]
    const float* H; long long ldh;       // EH_GEPI_DACT: stored activations, same shape as C
    int act;                             // eh_activation of the epilogue
    float* Z;                            // EH_GEPI_BIAS_ACT, nullable: the pre-activation, same layout as C (swish: act' needs it, the rounded h does not give it back)
    float* colsum;                       // (split-K weight gradients) nullable: colsum[z * c_zstride + n] = sum over the k chunk of B[k][n] -- the bias gradient,
                                         // taken from the B tiles the first row of workgroups stages anyway
    // (eh_fewrows_gemm_kernel only) a side job for workgroup (0, 0): the rows of partial sums the mechanistic stage left (few-rows chain
    // kernel) added up in the order of eh_lform_tail_sum -> job_out[16], so that the launch with the optimiser in its epilogues
    // (eh_dw_apply_kernel) finds the step's normalisation ready instead of every workgroup adding the rows up again
    const float* job_part; float* job_out; int job_nblk;
};

// The optimiser in the epilogue of the weight-gradient products (few rows, ONE slab row: the product IS the gradient, un-normalised):
// the element that would have gone to the slab goes through the update rule into theta / m / v instead -- no slab row written and read
// back, no reduce + optimiser launch behind the products (8.6 us of the tutorial net's 59 us step at batch 64).  EhLApply: what the
// host hands over; EhLApplyS: what a workgroup keeps in LDS once it has the step's normalisation (the sums of the mechanistic stage).
struct EhLApply {
    float* slab; float* theta; float* m; float* v;
    const float* sc_in; float* sc_out;
    EhOpt o; float* loss_slot; float* gradbuf; EhImg im;
    int loss_kind, n_theta, g_off;
    const float* part; int nblk;         // the mechanistic stage's rows of partial sums ...
    const float* tot;                    // ... or, non-null, their sums [16] (a side job of an earlier launch of the step, EhGemmArgs::job_out)
    unsigned long long* stamps; int stamp_wg;      // diagnostic builds (-DEH_STAMPS): workgroup stamp_wg stamps its phases
};
struct EhLApplyS { float* slab; float* theta; float* m; float* v; EhOpt o; float scale, bt1, bt2; int go, use_m, use_v; unsigned long long* stamps; };
#ifdef EH_STAMPS
#define EH_LSTAMP(S, i) do { __builtin_amdgcn_sched_barrier(0); if ((S)->stamps && threadIdx.x == 0) { (S)->stamps[2 * (i)] = __builtin_readcyclecounter(); (S)->stamps[2 * (i) + 1] = wall_clock64(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EH_LSTAMP(S, i)
#endif
// (EhLApplyS lives in LDS and reaches the epilogues through a generic pointer: read through it element by element, every field was a
//  flat load that had to be repeated behind every global store -- 13 k cycles for sixteen updates.  The epilogues take ONE copy into registers.)
typedef __attribute__((address_space(1))) float eh_gfloat;      // (pointers that come out of an LDS copy are generic: say that they are global)
__device__ __forceinline__ void eh_lapply_one(const EhLApplyS& S, const float* cslot, float gsum) {
    const long long idx = cslot - S.slab;
    eh_gfloat* const tp = (eh_gfloat*)S.theta; eh_gfloat* const mp = (eh_gfloat*)S.m; eh_gfloat* const vp = (eh_gfloat*)S.v;
    float th = tp[idx], mm = S.use_m ? mp[idx] : 0.0f, vv = S.use_v ? vp[idx] : 0.0f;
    eh_opt_update(S.o, gsum * S.scale, S.bt1, S.bt2, th, mm, vv);
    tp[idx] = th;
    if (S.use_m) mp[idx] = mm;
    if (S.use_v) vp[idx] = vv;
}

// N independent elements through the update rule, the rule looked at ONCE: inside eh_opt_update the rule is a branch per element, and
// sixteen unrolled copies of it keep the compiler from interleaving the sixteen dependent chains (an IEEE square root and two IEEE
// divisions each: 7.7 k cycles for the sixteen elements of a thread, one wave per SIMD).  Same arithmetic, op for op: this IS
// eh_opt_update with the rule hoisted (tests/test_gpu_lform.py trains through both and meets the same oracle trajectory).
template <int RULE, int N>
__device__ __forceinline__ void eh_opt_update_n(const EhOpt& o, const float (&g)[N], float bt1, float bt2, float (&th)[N], float (&m)[N], float (&v)[N]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (RULE == EH_OPT_ADAM || RULE == EH_OPT_ADAMW) {
            m[i] = o.b1 * m[i] + (1.0f - o.b1) * g[i];
            v[i] = o.b2 * v[i] + (1.0f - o.b2) * (g[i] * g[i]);
            float upd = m[i] / (1.0f - bt1) / (sqrtf(v[i] / (1.0f - bt2)) + o.eps) * o.lr;
            if constexpr (RULE == EH_OPT_ADAMW) upd += o.lr * o.wd * th[i];
            th[i] -= upd;
        } else if constexpr (RULE == EH_OPT_RMSPROP) {
            v[i] = o.b1 * v[i] + (1.0f - o.b1) * (g[i] * g[i]);
            th[i] -= g[i] * (o.lr / (sqrtf(v[i]) + o.eps));
        } else {
            th[i] -= o.lr * g[i];
        }
    }
}
template <int N>
__device__ __forceinline__ void eh_opt_update_all(const EhOpt& o, const float (&g)[N], float bt1, float bt2, float (&th)[N], float (&m)[N], float (&v)[N]) {
    if (o.rule == EH_OPT_ADAM) eh_opt_update_n<EH_OPT_ADAM, N>(o, g, bt1, bt2, th, m, v);
    else if (o.rule == EH_OPT_ADAMW) eh_opt_update_n<EH_OPT_ADAMW, N>(o, g, bt1, bt2, th, m, v);
    else if (o.rule == EH_OPT_RMSPROP) eh_opt_update_n<EH_OPT_RMSPROP, N>(o, g, bt1, bt2, th, m, v);
    else eh_opt_update_n<EH_OPT_DESCENT, N>(o, g, bt1, bt2, th, m, v);
}

__device__ __forceinline__ float eh_act_rt(int act, float z) {
    switch (act) {
        case EH_ACT_TANH: return eh_tanh(z);
        case EH_ACT_SIGMOID: return eh_sigmoid(z);
        case EH_ACT_RELU: return fmaxf(z, 0.0f);
        case EH_ACT_SWISH: return z * eh_sigmoid(z);
        default: return z;
    }
}
__device__ __forceinline__ float eh_dact_rt(int act, float h) {       // act' from the stored activation (swish: from the stored PRE-activation, EhGemmArgs::Z)
    switch (act) {
        case EH_ACT_SWISH: { const float sg = eh_sigmoid(h); return sg * (1.0f + h * (1.0f - sg)); }
        case EH_ACT_TANH: return 1.0f - h * h;
        case EH_ACT_SIGMOID: return h * (1.0f - h);
        case EH_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}

// C (M x N) = A (M x K) * B (K x N), fp32 in, fp32 out, on v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fmaf chain per output).
// 128 x 128 output tile per workgroup of four waves (2 x 2, 64 x 64 per wave = 2 x 2 MFMA tiles); K in steps of 16 through LDS as
// k-major tiles, so both MFMA operands are conflict-free row reads; the next step's global loads are in flight during the MFMAs.
// VEC (chosen by the host, eh_gemm_vec_ok): every operand base 16-byte aligned, leading dimensions multiples of 4, the k chunk a
// whole number of BK steps, the extent along an operand's contiguous dimension a multiple of 4.  The tiles then come in as 16-byte
// loads through pointers set up once (rows beyond the matrix are clamped to its last row: their products land in outputs the
// epilogue masks), the LDS tiles are double-buffered (one barrier per step) -- the main loop is loads, LDS traffic and MFMAs with
// a handful of vector-ALU instructions: on gfx950 an fp32 MFMA does not overlap with the vector ALU, so every
// address computation or bounds predicate in the loop is time taken from the matrix pipe.
// Four waves per SIMD (<= 128 registers: accumulators in the VGPR file, no AGPR copies): four workgroups per CU, so that e.g. the
// 2 048 tiles of the tutorial net's largest forward product are two full rounds of 1 024 resident workgroups, not 2.67 of 768.
#ifndef EH_GEMM_OCC
#define EH_GEMM_OCC 4
#endif
// BT = 64: 64 x 64 tiles (one MFMA tile per wave) for products too small to fill the chip with 128 x 128 ones.
template <bool ATR, bool BTR, int EPI, bool VEC, int BT>
__device__ __forceinline__ void eh_gemm_tile(const EhGemmArgs& g, const int bx, const int by, const int bz, const EhLApplyS* const Sp = nullptr) {
    EhLApplyS S{};
    if constexpr (EPI == EH_GEPI_APPLY) S = *Sp;
    static_assert(BT == 128 || (BT == 64 && VEC), "64 x 64 tiles exist in the 16-byte-load form only");
    constexpr int BM = BT, BN = BT, BK = 16, LDS_LD = BM + 4, TI = BT / 64, WT = BT / 2, NP = BT / 64, QT = BT / 4, RP = 256 / QT;
    __shared__ __attribute__((aligned(16))) float As[VEC ? 2 : 1][BK][LDS_LD], Bs[VEC ? 2 : 1][BK][LDS_LD];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, lh = lane >> 5;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    f32x16 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // EH_GEPI_APPLY (64 x 64 tiles): the thread's sixteen parameters and their moments are requested NOW -- they do not depend on the product
    float ap_th[16], ap_m[16], ap_v[16];
    long long ap_ix[16];
    if constexpr (EPI == EH_GEPI_APPLY) {
        static_assert(EPI != EH_GEPI_APPLY || BT == 64, "apply epilogue: one MFMA tile per wave");
        eh_gfloat* const tp = (eh_gfloat*)S.theta; eh_gfloat* const mp = (eh_gfloat*)S.m; eh_gfloat* const vp = (eh_gfloat*)S.v;
        const int n = min(n0 + wn * WT + l32, g.N - 1);
        const float* const Cb = g.C + (long long)bz * g.c_zstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wm * WT + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
            ap_ix[r] = (Cb + (long long)m * g.ldc + n) - S.slab;
            ap_th[r] = tp[ap_ix[r]];
            ap_m[r] = S.use_m ? mp[ap_ix[r]] : 0.0f;
            ap_v[r] = S.use_v ? vp[ap_ix[r]] : 0.0f;
        }
    }
    const bool do_cs_v = (EPI == EH_GEPI_STORE || EPI == EH_GEPI_APPLY) && g.colsum != nullptr && by == 0 && tid < BN;
    float cs_v = 0.0f;
    if constexpr (VEC) {
        // NP (= BT / 64) 16-byte pieces of each tile per thread and step
        //   operand contiguous along k (A: !ATR, B: BTR): piece j = row (tid >> 2) + 64 j, k quad tid & 3  -> four b32 LDS stores [4q + i][row]
        //   operand contiguous along m / n (A: ATR, B: !BTR): piece j = k row tid / QT + RP j, quad tid % QT -> one b128 LDS store [k][4 quad]
        const float* pa[NP]; const float* pb[NP];
        long long sa, sb;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (ATR) { const int mq = m0 + 4 * (tid % QT); pa[j] = g.A + (long long)(kbeg + tid / QT + RP * j) * g.lda + (mq < g.M ? mq : 0); }
            else { const int m = min(m0 + (tid >> 2) + 64 * j, g.M - 1); pa[j] = g.A + (long long)m * g.lda + kbeg + 4 * (tid & 3); }
            if (BTR) { const int n = min(n0 + (tid >> 2) + 64 * j, g.N - 1); pb[j] = g.B + (long long)n * g.ldb + kbeg + 4 * (tid & 3); }
            else { const int nq = n0 + 4 * (tid % QT); pb[j] = g.B + (long long)(kbeg + tid / QT + RP * j) * g.ldb + (nq < g.N ? nq : 0); }
        }
        sa = ATR ? (long long)BK * g.lda : BK; sb = BTR ? BK : (long long)BK * g.ldb;
        f32x4 ra[NP], rb[NP];
        auto gload = [&]() {
#pragma unroll
            for (int j = 0; j < NP; ++j) { ra[j] = *(const f32x4*)pa[j]; rb[j] = *(const f32x4*)pb[j]; pa[j] += sa; pb[j] += sb; }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (ATR) *(f32x4*)&As[buf][tid / QT + RP * j][4 * (tid % QT)] = ra[j];
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) As[buf][4 * (tid & 3) + i][(tid >> 2) + 64 * j] = ra[j][i];
                }
                if (BTR) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Bs[buf][4 * (tid & 3) + i][(tid >> 2) + 64 * j] = rb[j][i];
                } else *(f32x4*)&Bs[buf][tid / QT + RP * j][4 * (tid % QT)] = rb[j];
            }
        };
        const int nsteps = (kend - kbeg) / BK;
        // (one MFMA chain of a 64 x 64 step is ~0.25 us, a global load 1-2 us: with one tile in flight the K loop of a small-batch
        //  product -- one or two workgroups per CU, nothing else to hide behind -- ran at one memory latency per step, 17.5 us for the
        //  tutorial net's 1 024-deep forward product at B = 64 (tools/lform_trace.sh).  64 x 64 tiles keep DEPTH = 4 tiles in flight in
        //  registers; the 128 x 128 ones, whose accumulators fill the register budget and whose products fill the chip, keep one.)
        constexpr int DEPTH = BT == 64 ? 4 : 1;
        if constexpr (DEPTH == 1) {
            if (nsteps > 0) { gload(); lstore(0); }
            __syncthreads();
            for (int st = 0; st < nsteps; ++st) {
                const int cur = st & 1;
                if (st + 1 < nsteps) gload();             // in flight behind this step's MFMAs
                if (do_cs_v) {
#pragma unroll
                    for (int kk = 0; kk < BK; ++kk) cs_v += Bs[cur][kk][tid];
                }
#pragma unroll
                for (int k2 = 0; k2 < BK; k2 += 2) {
                    float av[TI], bv[TI];
#pragma unroll
                    for (int i = 0; i < TI; ++i) { av[i] = As[cur][k2 + lh][wm * WT + 32 * i + l32]; bv[i] = Bs[cur][k2 + lh][wn * WT + 32 * i + l32]; }
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
                if (st + 1 < nsteps) lstore(cur ^ 1);     // the other buffer: its readers passed the barrier that ended the previous step
                __syncthreads();
            }
        } else {
            // slot d of the ring holds tile s with s % DEPTH == d: requested DEPTH steps before it is parked in LDS
            f32x4 qa[DEPTH][NP], qb[DEPTH][NP];
            auto gload_to = [&](int d) {
#pragma unroll
                for (int j = 0; j < NP; ++j) { qa[d][j] = *(const f32x4*)pa[j]; qb[d][j] = *(const f32x4*)pb[j]; pa[j] += sa; pb[j] += sb; }
            };
            auto lstore_from = [&](int buf, int d) {
#pragma unroll
                for (int j = 0; j < NP; ++j) { ra[j] = qa[d][j]; rb[j] = qb[d][j]; }
                lstore(buf);
            };
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                if (d < nsteps) gload_to(d);
            if (nsteps > 0) lstore_from(0, 0);
            __syncthreads();
#pragma unroll 1
            for (int st0 = 0; st0 < nsteps; st0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const int st = st0 + d;
                    if (st < nsteps) {
                        const int cur = d & 1;                                  // (DEPTH is even: st & 1 == d & 1)
                        if (st + DEPTH < nsteps) gload_to(d);                  // slot d was parked at the end of the previous step
                        if (do_cs_v) {
#pragma unroll
                            for (int kk = 0; kk < BK; ++kk) cs_v += Bs[cur][kk][tid];
                        }
#pragma unroll
                        for (int k2 = 0; k2 < BK; k2 += 2) {
                            float av[TI], bv[TI];
#pragma unroll
                            for (int i = 0; i < TI; ++i) { av[i] = As[cur][k2 + lh][wm * WT + 32 * i + l32]; bv[i] = Bs[cur][k2 + lh][wn * WT + 32 * i + l32]; }
#pragma unroll
                            for (int i = 0; i < TI; ++i)
#pragma unroll
                                for (int j = 0; j < TI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                        }
                        if (st + 1 < nsteps) lstore_from(cur ^ 1, (d + 1) % DEPTH);
                        __syncthreads();
                    }
                }
            }
        }
    }
    if constexpr (EPI == EH_GEPI_APPLY) { EH_LSTAMP(Sp, 2); }
    const bool do_cs = do_cs_v;
    float cs = cs_v;
    if constexpr (!VEC) {
    constexpr int NE = BM * BK / 256;      // tile elements per thread (8)
    float ra[NE], rb[NE];
    // element e of a tile: the thread order follows the operand's contiguous dimension (coalesced global loads)
    auto tile_a = [&](int e, int& mm, int& kk) { if (ATR) { kk = e / BM; mm = e % BM; } else { mm = e / BK; kk = e % BK; } };
    auto tile_b = [&](int e, int& nn, int& kk) { if (BTR) { nn = e / BK; kk = e % BK; } else { kk = e / BN; nn = e % BN; } };
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            int mm, nn, ka, kb;
            tile_a(tid + 256 * j, mm, ka);
            tile_b(tid + 256 * j, nn, kb);
            const int m = m0 + mm, n = n0 + nn;
            ra[j] = (m < g.M && k0 + ka < kend) ? (ATR ? g.A[(long long)(k0 + ka) * g.lda + m] : g.A[(long long)m * g.lda + k0 + ka]) : 0.0f;
            rb[j] = (n < g.N && k0 + kb < kend) ? (BTR ? g.B[(long long)n * g.ldb + k0 + kb] : g.B[(long long)(k0 + kb) * g.ldb + n]) : 0.0f;
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                       // the previous step's MFMAs are done with the tiles
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            int mm, nn, ka, kb;
            tile_a(tid + 256 * j, mm, ka);
            tile_b(tid + 256 * j, nn, kb);
            As[0][ka][mm] = ra[j];
            Bs[0][kb][nn] = rb[j];
        }
        __syncthreads();
        if (k0 + BK < kend) load(k0 + BK);
        if (do_cs) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) cs += Bs[0][kk][tid];      // (k order: deterministic)
        }
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
            const float a0 = As[0][k2 + lh][wm * 64 + l32], a1 = As[0][k2 + lh][wm * 64 + 32 + l32];
            const float b0 = Bs[0][k2 + lh][wn * 64 + l32], b1 = Bs[0][k2 + lh][wn * 64 + 32 + l32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    }
    if (do_cs && n0 + tid < g.N) {
        if constexpr (EPI == EH_GEPI_APPLY) { if (S.go) eh_lapply_one(S, g.colsum + n0 + tid, cs); }
        else g.colsum[(long long)bz * g.c_zstride + n0 + tid] = cs;
    }
    // C/D layout of the 32x32 MFMA: lane -> column (lane & 31); register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* const C = g.C + (long long)bz * g.c_zstride;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int n = n0 + wn * WT + 32 * j + l32;
            const float bv = ((EPI == EH_GEPI_BIAS_ACT || EPI == EH_GEPI_BIAS_T) && n < g.N) ? g.bias[n] : 0.0f;
            if constexpr (EPI == EH_GEPI_APPLY) {
                // (all of a thread's parameters and moments requested before the first update: one round trip, not sixteen)
                if (S.go && n < g.N) {
                    // every update, then the stores -- a store between two updates made the next one wait for the store's acknowledgement
                    // (loads and stores share one counter and may complete out of order: the compiler waits for zero), sixteen round
                    // trips in a row: 13 k of the workgroup's 28 k cycles (tools/stamps_lform.py, EH_STAMP_DW)
                    eh_gfloat* const tp = (eh_gfloat*)S.theta; eh_gfloat* const mp = (eh_gfloat*)S.m; eh_gfloat* const vp = (eh_gfloat*)S.v;
                    float gg[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) gg[r] = acc[i][j][r] * S.scale;
                    EH_LSTAMP(Sp, 3);
                    eh_opt_update_all<16>(S.o, gg, S.bt1, S.bt2, ap_th, ap_m, ap_v);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * WT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (m < g.M) {
                            tp[ap_ix[r]] = ap_th[r];
                            if (S.use_m) mp[ap_ix[r]] = ap_m[r];
                            if (S.use_v) vp[ap_ix[r]] = ap_v[r];
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.M && n < g.N) {
                    float v = acc[i][j][r];
                    if (EPI == EH_GEPI_BIAS_ACT) { v += bv; if (g.Z) g.Z[(long long)m * g.ldc + n] = v; v = eh_act_rt(g.act, v); }
                    else if (EPI == EH_GEPI_BIAS_T) v += bv;
                    else if (EPI == EH_GEPI_DACT) v *= eh_dact_rt(g.act, g.H[(long long)m * g.ldh + n]);
                    if (EPI == EH_GEPI_BIAS_T) C[(long long)n * g.ldc + m] = v;
                    else C[(long long)m * g.ldc + n] = v;
                }
            }
        }
}

template <bool ATR, bool BTR, int EPI, bool VEC = false, int BT = 128>
__global__ __launch_bounds__(256, EH_GEMM_OCC) void eh_gemm_kernel(const EhGemmArgs g) {
    eh_gemm_tile<ATR, BTR, EPI, VEC, BT>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several independent products in one launch (the weight gradients of every layer of a small-batch step, once every delta is there:
// each of them alone is a handful of tiles, one dependent launch of 4-6 us each): workgroup -> product by its first-tile table.
enum { EH_GEMM_GROUP = 8 };
struct EhGemmGroup { EhGemmArgs g[EH_GEMM_GROUP]; int t0[EH_GEMM_GROUP + 1]; int gx[EH_GEMM_GROUP], gy[EH_GEMM_GROUP]; int n; };
template <bool ATR, bool BTR, int EPI, bool VEC, int BT>
__global__ __launch_bounds__(256, EH_GEMM_OCC) void eh_gemm_group_kernel(const EhGemmGroup G) {
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.t0[i + 1]) ++i;
    const int t = (int)blockIdx.x - G.t0[i], gx = G.gx[i], gy = G.gy[i];
    const EhGemmArgs g = G.g[i];
    eh_gemm_tile<ATR, BTR, EPI, VEC, BT>(g, t % gx, (t / gx) % gy, t / (gx * gy));
}

// may this product run the 16-byte-load form of eh_gemm_kernel?
inline bool eh_gemm_vec_ok(const EhGemmArgs& g, bool atr, bool btr) {
    auto al16 = [](const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; };
    if (!al16(g.A) || !al16(g.B) || (g.lda & 3) || (g.ldb & 3)) return false;
    if (g.K <= 0 || (g.K % 16) || (g.kchunk % 16)) return false;
    if (atr && (g.M & 3)) return false;          // 16-byte pieces along m
    if (!btr && (g.N & 3)) return false;         // 16-byte pieces along n
    return g.M > 0 && g.N > 0;
}

// Weight gradient of a layer with a thin side (the first layer's few predictors, the last layer's few outputs): C(col, j) =
// sum over the samples b of chunk z of wide[b][col] * thin(b, j), j < J <= 8 -- a streaming pass over `wide` (a 128 x 128 MFMA
// tile would spend 94 % and more of its work on padding: 277 us against 70 for the tutorial net's first layer at B = 65 536).
// Also the bias gradient that the tiled product takes from its B tiles: column sums of `wide` (cs_wide) or of `thin` (cs_thin).
struct EhThinArgs {
    const float* wide; long long ldw; int ncols;
    const float* thin; long long tsb, tsj; int J;
    int K, kchunk;
    float* C; long long c_col, c_j, c_z;
    float* cs_wide; float* cs_thin; long long cs_z;
};
template <bool APPLY = false>
__device__ __forceinline__ void eh_thin_gemm_tile(const EhThinArgs& a, const int bx, const int by, const EhLApplyS* const Sp = nullptr) {
    EhLApplyS S{};
    if constexpr (APPLY) S = *Sp;
    constexpr int SB = 512, U = 8;              // samples staged per round; wide loads in flight per thread
    __shared__ float sT[8][SB];                 // the thin operand of the round, [j][sample]: every lane of a wave reads the same word (broadcast)
    __shared__ float red[4][64][10];
    __shared__ float redt[4][8];
    const int tid = threadIdx.x, cl = tid & 63, q = tid >> 6, col = bx * 64 + cl, z = by;
    const int kbeg = z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const bool live = col < a.ncols;
    float acc[8], cst[8], csw = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j] = 0.0f; cst[j] = 0.0f; }
    for (int s0 = kbeg; s0 < kend; s0 += SB) {
        const int ns = min(SB, kend - s0);
        __syncthreads();
        for (int e = tid; e < a.J * SB; e += 256) {
            const int j = a.tsb == 1 ? e / SB : e % a.J, bl = a.tsb == 1 ? e % SB : e / a.J;      // follow the operand's contiguous dimension
            sT[j][bl] = bl < ns ? a.thin[(long long)(s0 + bl) * a.tsb + (long long)j * a.tsj] : 0.0f;
        }
        __syncthreads();
        for (int b0 = q; b0 < ns; b0 += 4 * U) {
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int bl = b0 + 4 * u; w[u] = (live && bl < ns) ? a.wide[(long long)(s0 + bl) * a.ldw + col] : 0.0f; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int bl = b0 + 4 * u;
                if (bl < ns) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < a.J) { const float t = sT[j][bl]; acc[j] = fmaf(t, w[u], acc[j]); cst[j] += t; }
                    csw += w[u];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[q][cl][j] = acc[j];
    red[q][cl][8] = csw;
    if (cl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) redt[q][j] = cst[j];
    }
    __syncthreads();
    if constexpr (APPLY) {
        if (S.go) {
            if (q == 0 && live) {
                // (all of the thread's parameters and moments requested before the first update, every update before the first store: see eh_gemm_tile)
                long long ix[9];
                float th[9], mm[9], vv[9];
                eh_gfloat* const tp = (eh_gfloat*)S.theta; eh_gfloat* const mp = (eh_gfloat*)S.m; eh_gfloat* const vp = (eh_gfloat*)S.v;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                    const float* const slot = j < 8 ? a.C + (long long)col * a.c_col + (long long)min(j, a.J - 1) * a.c_j : (a.cs_wide ? a.cs_wide + col : a.C);
                    ix[j] = slot - S.slab;
                    th[j] = on ? tp[ix[j]] : 0.0f;
                    mm[j] = (on && S.use_m) ? mp[ix[j]] : 0.0f;
                    vv[j] = (on && S.use_v) ? vp[ix[j]] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                    if (on) eh_opt_update(S.o, ((red[0][cl][j] + red[1][cl][j]) + (red[2][cl][j] + red[3][cl][j])) * S.scale, S.bt1, S.bt2, th[j], mm[j], vv[j]);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                    if (on) {
                        tp[ix[j]] = th[j];
                        if (S.use_m) mp[ix[j]] = mm[j];
                        if (S.use_v) vp[ix[j]] = vv[j];
                    }
                }
            }
            if (a.cs_thin && bx == 0 && tid < a.J) eh_lapply_one(S, a.cs_thin + tid, (redt[0][tid] + redt[1][tid]) + (redt[2][tid] + redt[3][tid]));
        }
        return;
    }
    if (q == 0 && live) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < a.J) a.C[(long long)z * a.c_z + (long long)col * a.c_col + (long long)j * a.c_j] = (red[0][cl][j] + red[1][cl][j]) + (red[2][cl][j] + red[3][cl][j]);
        if (a.cs_wide) a.cs_wide[(long long)z * a.cs_z + col] = (red[0][cl][8] + red[1][cl][8]) + (red[2][cl][8] + red[3][cl][8]);
    }
    if (a.cs_thin && bx == 0 && tid < a.J) a.cs_thin[(long long)z * a.cs_z + tid] = (redt[0][tid] + redt[1][tid]) + (redt[2][tid] + redt[3][tid]);
}
__global__ __launch_bounds__(256) void eh_thin_gemm_kernel(const EhThinArgs a) { eh_thin_gemm_tile<false>(a, (int)blockIdx.x, (int)blockIdx.y); }
// (several of them in one launch, as eh_gemm_group_kernel)
struct EhThinGroup { EhThinArgs a[EH_GEMM_GROUP]; int t0[EH_GEMM_GROUP + 1]; int gx[EH_GEMM_GROUP]; int n; };
__global__ __launch_bounds__(256) void eh_thin_gemm_group_kernel(const EhThinGroup G) {
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.t0[i + 1]) ++i;
    const int t = (int)blockIdx.x - G.t0[i], gx = G.gx[i];
    const EhThinArgs a = G.a[i];
    eh_thin_gemm_tile<false>(a, t % gx, t / gx);
}

// every 64-byte line of a large block of kernel arguments requested at once (eh_kernarg_warm's idea, eh_device.hpp, for blocks beyond
// its twelve lines): the table of layers is read phase by phase, and every cold line is a memory round trip on the critical path
template <int LINE0, int BYTES, class KA>
__device__ __forceinline__ void eh_kernarg_warm4(const KA ka) {
    if constexpr (64 * LINE0 < BYTES) {
        unsigned d0, d1 = 0u, d2 = 0u, d3 = 0u;
        asm volatile("s_load_dword %0, %1, %2" : "=&s"(d0) : "s"(ka), "n"(64 * LINE0));
        if constexpr (64 * (LINE0 + 1) < BYTES) asm volatile("s_load_dword %0, %1, %2" : "=&s"(d1) : "s"(ka), "n"(64 * (LINE0 + 1)));
        if constexpr (64 * (LINE0 + 2) < BYTES) asm volatile("s_load_dword %0, %1, %2" : "=&s"(d2) : "s"(ka), "n"(64 * (LINE0 + 2)));
        if constexpr (64 * (LINE0 + 3) < BYTES) asm volatile("s_load_dword %0, %1, %2" : "=&s"(d3) : "s"(ka), "n"(64 * (LINE0 + 3)));
        eh_kernarg_warm4<LINE0 + 4, BYTES>(ka);
        // (the destinations stay allocated up to the wait the outermost call ends with: a register handed out earlier would be overwritten when its load lands)
        asm volatile("" ::"s"(d0), "s"(d1), "s"(d2), "s"(d3));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
template <int BYTES>
__device__ __forceinline__ void eh_kernarg_warm_big() {
    static_assert(BYTES <= 4096, "kernel arguments");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    eh_kernarg_warm4<0, BYTES>(ka);
}
// Both groups of a small-batch step's weight gradients -- the tiled ones and the thin ones -- as ONE launch: the first workgroups run
// the thin products, the others the tiles (one dependent launch fewer; the two bodies' LDS side by side, 44 KB).
// ... and, as one more workgroup behind them, the sums of the mechanistic stage (rows of partial sums -> the tail columns of the slab:
// eh_lform_tail_sum below, the body of eh_lform_tail_kernel) when the stage ran on several workgroups (eh_lform_tailchain_kernel).
struct EhLTailJob { const float* part; int nblk; float* slab; int nrows; long long n_acc; };
__device__ __forceinline__ void eh_lform_tail_sum(const float* part, int nblk, const EhNet& net, float* slab, int nrows, long long n_acc);
__global__ __launch_bounds__(256) void eh_dw_group_kernel(const EhGemmGroup G, const EhThinGroup T, const EhLTailJob J, const EhNet net) {
    const int nthin = T.t0[T.n];
    if (J.part && (int)blockIdx.x == (int)gridDim.x - 1) { eh_lform_tail_sum(J.part, J.nblk, net, J.slab, J.nrows, J.n_acc); return; }
    if ((int)blockIdx.x < nthin) {
        int i = 0;
        while (i + 1 < T.n && (int)blockIdx.x >= T.t0[i + 1]) ++i;
        const int t = (int)blockIdx.x - T.t0[i], gx = T.gx[i];
        const EhThinArgs a = T.a[i];
        eh_thin_gemm_tile<false>(a, t % gx, t / gx);
    } else {
        const int b = (int)blockIdx.x - nthin;
        int i = 0;
        while (i + 1 < G.n && b >= G.t0[i + 1]) ++i;
        const int t = b - G.t0[i], gx = G.gx[i], gy = G.gy[i];
        const EhGemmArgs g = G.g[i];
        eh_gemm_tile<true, false, EH_GEPI_STORE, true, 64>(g, t % gx, (t / gx) % gy, t / (gx * gy));
    }
}

// Products with a degenerate dimension, as streaming kernels (a 128 x 128 MFMA tile spends 94 % and more of such a product on padding;
// at the tutorial's batch of 64 the three of them -- first layer (2 predictors), output layer (1 output) and its delta -- took
// 16 + 14 + 13 us of a 250 us step, tools/lform_trace.sh):
//   eh_thin_fwd_k_kernel : C[m][n] = act(sum_{k < K <= 8} A[m][k] B[k][n] + bias[n])          (first layer: few predictors; optional Z)
//   eh_thin_fwd_n_kernel : C[n][m] = sum_k A[m][k] B[k][n] + bias[n],  N <= 16                 (output layer, transposed output)
//   eh_thin_dact_kernel  : C[m][n] = (sum_{k < K <= 16} A[k][m] B[n][k]) * act'(H[m][n])        (delta below the output layer; A = dO^T [K][lda])
// Same sums in the same k order as the tiled kernel (a k-ordered fmaf chain per output).
__global__ __launch_bounds__(256) void eh_thin_fwd_k_kernel(const EhGemmArgs g) {
    const long long tot = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int m = (int)(e / g.N), n = (int)(e - (long long)m * g.N);
        float v = 0.0f;
        for (int k = 0; k < g.K; ++k) v = fmaf(g.A[(long long)m * g.lda + k], g.B[(long long)k * g.ldb + n], v);
        v += g.bias[n];
        if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
        g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
    }
}
__global__ __launch_bounds__(256) void eh_thin_fwd_n_kernel(const EhGemmArgs g) {
    // one wave per row m: lanes over k, all N (<= 16) outputs per lane, wave sums -- every load of A is a contiguous run
    const int lane = threadIdx.x & 63, wave = (int)(((long long)blockIdx.x * 256 + threadIdx.x) >> 6), nwave = (int)(((long long)gridDim.x * 256) >> 6);
    for (int m = wave; m < g.M; m += nwave) {
        float acc[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = 0.0f;
        for (int k = lane; k < g.K; k += 64) {
            const float a = g.A[(long long)m * g.lda + k];
#pragma unroll
            for (int n = 0; n < 16; ++n)
                if (n < g.N) acc[n] = fmaf(a, g.B[(long long)k * g.ldb + n], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < 16; ++n)
            if (n < g.N) {
                const float sum = eh_wave_sum(acc[n]);
                if (lane == 0) g.C[(long long)n * g.ldc + m] = sum + g.bias[n];
            }
    }
}
__global__ __launch_bounds__(256) void eh_thin_dact_kernel(const EhGemmArgs g) {
    const long long tot = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int m = (int)(e / g.N), n = (int)(e - (long long)m * g.N);
        float v = 0.0f;
        for (int k = 0; k < g.K; ++k) v = fmaf(g.A[(long long)k * g.lda + m], g.B[(long long)n * g.ldb + k], v);
        g.C[(long long)m * g.ldc + n] = v * eh_dact_rt(g.act, g.H[(long long)m * g.ldh + n]);
    }
}

// Split-K for products with few rows (small minibatches): M <= 256 rows give one or four 64 x 64 tiles per 64 output columns -- 8-16
// workgroups on 256 CUs, each a dependent chain of eight MFMAs per 16-deep step, 17 us for a 1 024-deep product -- so the k range is
// split over blockIdx.z into partial products (plain stores, [z][M][N]) and this pass adds them in z order (deterministic) and applies
// the epilogue the tiled kernel would have: bias + activation (+ the pre-activation for swish), or act' of the stored activation.
__global__ __launch_bounds__(256) void eh_splitk_combine_kernel(const float* part, int nz, int epi, const EhGemmArgs g) {
    const long long tot = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int m = (int)(e / g.N), n = (int)(e - (long long)m * g.N);
        float v = 0.0f;
        for (int z = 0; z < nz; ++z) v += part[(long long)z * tot + e];
        if (epi == EH_GEPI_BIAS_ACT) {
            v += g.bias[n];
            if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
            g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
        } else {
            g.C[(long long)m * g.ldc + n] = v * eh_dact_rt(g.act, g.H[(long long)m * g.ldh + n]);
        }
    }
}

// Products with few rows and a deep k without the combine pass: one 16 x 16 output tile per workgroup, the k range split over its
// (up to 16) waves.  A wave's operands go straight from global memory into registers, 64 k per round with every load of the round in
// flight at once (v_mfma_f32_16x16x4_f32: lane (r, q) supplies A[m0 + r][k] and B[k][n0 + r] for k = kb + 4 q + i in step i of a
// 16-deep group -- any partition of the k range into fours is a valid one as long as both operands use the same); the waves'
// partial tiles are added in wave order through LDS (deterministic), then the epilogue of the tiled kernel.  A is [m][k] (!ATR).
// g.kchunk = the k slice of one wave (a multiple of 16); blockDim.x = 64 * number of slices.
typedef float f32x4_lf __attribute__((ext_vector_type(4)));
template <bool BTR, int EPI>
__global__ __launch_bounds__(1024) void eh_fewrows_gemm_kernel(const EhGemmArgs g) {
    static_assert(EPI == EH_GEPI_BIAS_ACT || EPI == EH_GEPI_DACT, "epilogues of the small-batch products");
    __shared__ float red[16][256];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = (int)(blockDim.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int kbeg = wave * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nc = min(n0 + r, g.N - 1);
    const float* const pa = g.A + (long long)min(m0 + r, g.M - 1) * g.lda + 4 * q;
    const float* const pb = BTR ? g.B + (long long)nc * g.ldb + 4 * q : g.B + (long long)(4 * q) * g.ldb + nc;
    // what the epilogue needs from global memory -- the bias, or the stored activation act' is taken from -- is asked for NOW: read where it
    // is used, behind the fold, it was one more cold round trip at the end of a launch that is little more than two of them
    float epi_in = 0.0f;
    if (tid < 256) {
        const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
        if (m < g.M && n < g.N) epi_in = EPI == EH_GEPI_BIAS_ACT ? g.bias[n] : g.H[(long long)m * g.ldh + n];
    }
    f32x4_lf acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        f32x4_lf a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kb = k0 + 16 * u;
            const bool ok = kb < kend;
            a[u] = ok ? *(const f32x4_lf*)(pa + kb) : f32x4_lf{0.0f, 0.0f, 0.0f, 0.0f};
            if (BTR) b[u] = ok ? *(const f32x4_lf*)(pb + kb) : f32x4_lf{0.0f, 0.0f, 0.0f, 0.0f};
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[u][i] = ok ? pb[(long long)(kb + i) * g.ldb] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][i], acc, 0, 0, 0);
    }
    // C/D layout of the 16x16 MFMA: lane -> column (lane & 15); register i -> row 4 (lane >> 4) + i
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * q + i) * 16 + r] = acc[i];
    __syncthreads();
    if (tid < 256) {
        float v = 0.0f;
        for (int w = 0; w < nw; ++w) v += red[w][tid];
        const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
        if (m < g.M && n < g.N) {
            if (EPI == EH_GEPI_BIAS_ACT) {
                v += epi_in;
                if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
                g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
            } else {
                g.C[(long long)m * g.ldc + n] = v * eh_dact_rt(g.act, epi_in);
            }
        }
    }
    if (g.job_part && blockIdx.x == 0 && blockIdx.y == 0 && tid < 256) {
        const int col = tid & 15, grp = tid >> 4;
        float s = 0.0f;
        for (int b = grp; b < g.job_nblk; b += 16) s += g.job_part[(long long)b * 16 + col];
        red[15][tid] = s;                                                // (a thread's own column of the last wave's partials: read above by this thread only)
    }
    __syncthreads();
    if (g.job_part && blockIdx.x == 0 && blockIdx.y == 0 && tid < 16) {
        float t = 0.0f;
        for (int q = 0; q < 16; ++q) t += red[15][q * 16 + tid];
        g.job_out[tid] = t;
    }
}

// The same with a 32 x 32 output tile per workgroup (v_mfma_f32_32x32x2_f32: lane (r, h) supplies A[m0 + r][k] and B[k][n0 + r] for
// k = kb + 4 h + i in step i of an 8-deep group): a quarter of the workgroups and half the operand traffic per product -- from a
// few hundred rows on, the 16 x 16 form is thousands of workgroups that each live one memory latency (16 us for a 1 024 x 1 024 x 512
// product, tools/lform_trace.sh).  64 KB of LDS for the waves' partial tiles.
template <bool BTR, int EPI>
__global__ __launch_bounds__(1024) void eh_fewrows32_gemm_kernel(const EhGemmArgs g) {
    static_assert(EPI == EH_GEPI_BIAS_ACT || EPI == EH_GEPI_DACT, "epilogues of the small-batch products");
    __shared__ float red[16][1024];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = (int)(blockDim.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kbeg = wave * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nc = min(n0 + r, g.N - 1);
    const float* const pa = g.A + (long long)min(m0 + r, g.M - 1) * g.lda + 4 * hh;
    const float* const pb = BTR ? g.B + (long long)nc * g.ldb + 4 * hh : g.B + (long long)(4 * hh) * g.ldb + nc;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        f32x4_lf a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kb = k0 + 8 * u;
            const bool ok = kb < kend;
            a[u] = ok ? *(const f32x4_lf*)(pa + kb) : f32x4_lf{0.0f, 0.0f, 0.0f, 0.0f};
            if (BTR) b[u] = ok ? *(const f32x4_lf*)(pb + kb) : f32x4_lf{0.0f, 0.0f, 0.0f, 0.0f};
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[u][i] = ok ? pb[(long long)(kb + i) * g.ldb] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][i], acc, 0, 0, 0);
    }
    // C/D layout of the 32x32 MFMA: lane -> column (lane & 31); register i -> row (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][((i & 3) + 8 * (i >> 2) + 4 * hh) * 32 + r] = acc[i];
    __syncthreads();
    for (int e = tid; e < 1024; e += (int)blockDim.x) {
        float v = 0.0f;
        for (int w = 0; w < nw; ++w) v += red[w][e];
        const int m = m0 + (e >> 5), n = n0 + (e & 31);
        if (m < g.M && n < g.N) {
            if (EPI == EH_GEPI_BIAS_ACT) {
                v += g.bias[n];
                if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
                g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
            } else {
                g.C[(long long)m * g.ldc + n] = v * eh_dact_rt(g.act, g.H[(long long)m * g.ldh + n]);
            }
        }
    }
}

// The minibatch as the GEMMs want it: Xb [count][P] = the predictors of samples idx[first + i] (or first + i), normalised by the
// input BatchNorm when the model has one (train mode: the statistics of THIS minibatch from eh_bn_stats_kernel's partial sums, and
// block 0 advances the running statistics; test mode: the running statistics in `meta`).
struct EhLPrepArgs {
    const float* recs; int C, P;
    const int* idx; long long first; int count;
    float* Xb;
    float* meta;              // the handle's EH_IMG_* block (global memory)
    const float* bn_part; int bn_nblk; const float* bn_c; const float* bn_n; int bn_update; float* bn_run;
    int bn_self;              // train-mode statistics of a small minibatch taken here (every workgroup the same sums in the same order) -- no eh_bn_stats_kernel
    unsigned long long* stamps;     // diagnostic builds (-DEH_STAMPS, EH_STAMP_FIRST): workgroup (0, 0) stamps its phases
};
#ifdef EH_STAMPS
#define EH_FSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.stamps[2 * (i)] = __builtin_readcyclecounter(); a.stamps[2 * (i) + 1] = wall_clock64(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EH_FSTAMP(i)
#endif
// FWDK: the first Dense layer of a network with few predictors (eh_thin_fwd_k_kernel's product, K <= 8) in the same launch -- its
// inputs are the values this kernel writes to Xb, taken from the records by the same expression (other workgroups write the rows this
// one would need): g.A is not read, the network's predictors are columns c0 .. c0 + K - 1 of the minibatch matrix.
template <bool FWDK>
__global__ __launch_bounds__(256) void eh_lform_prep_kernel(const EhLPrepArgs a, const EhGemmArgs g, const int c0) {
    __shared__ float mu[32], rs[32], sred[8][64];
    const int tid = threadIdx.x;
    if (a.bn_self) {
        // thread (g = tid / 32, p = tid % 32): samples g, g + 8, ... of predictor p, shifted by the batch's first sample against cancellation
        const int p = tid & 31, grp = tid >> 5;
        const long long n0 = a.idx ? (long long)a.idx[a.first] : a.first;
        float s1 = 0.0f, s2 = 0.0f;
        if (p < a.P) {
            const float c0 = a.recs[n0 * a.C + p];
#pragma unroll 8
            for (int i = grp; i < a.count; i += 8) {
                const long long n = a.idx ? (long long)a.idx[a.first + i] : a.first + i;
                const float d = a.recs[n * a.C + p] - c0;
                s1 += d; s2 += d * d;
            }
        }
        sred[grp][p] = s1; sred[grp][32 + p] = s2;
        __syncthreads();
    }
    if (tid < 32) {
        float m = a.meta[EH_IMG_BNM + tid], r = a.meta[EH_IMG_BNR + tid];
        if ((a.bn_part || a.bn_self) && tid < a.P) {
            float s1 = 0.0f, s2 = 0.0f;
            if (a.bn_self) { for (int b = 0; b < 8; ++b) { s1 += sred[b][tid]; s2 += sred[b][32 + tid]; } }
            else for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const long long nf = a.idx ? (long long)a.idx[a.first] : a.first;
            const float cnt = a.bn_n ? *a.bn_n : (float)a.count, c0 = a.bn_self ? a.recs[nf * a.C + tid] : a.bn_c[tid];
            const float d = s1 / cnt, var = fmaxf(s2 / cnt - d * d, 0.0f);
            m = c0 + d; r = 1.0f / sqrtf(var + EH_BN_EPS);
            if (a.bn_update && blockIdx.x == 0) {
                const float rm = (1.0f - EH_BN_MOMENTUM) * a.bn_run[tid] + EH_BN_MOMENTUM * m;
                const float rv = (1.0f - EH_BN_MOMENTUM) * a.bn_run[32 + tid] + EH_BN_MOMENTUM * (cnt > 1.0f ? cnt / (cnt - 1.0f) : 1.0f) * var;
                a.bn_run[tid] = rm; a.bn_run[32 + tid] = rv;
                a.meta[EH_IMG_BNM + tid] = rm;                          // what forward / eval (test mode) will use
                a.meta[EH_IMG_BNR + tid] = 1.0f / sqrtf(rv + EH_BN_EPS);
            }
        }
        mu[tid] = m; rs[tid] = r;
    }
    __syncthreads();
    const long long tot = (long long)a.count * a.P;
    for (long long e = (long long)blockIdx.x * 256 + tid; e < tot; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / a.P), p = (int)(e - (long long)i * a.P);
        const long long n = a.idx ? (long long)a.idx[a.first + i] : a.first + i;
        const float x = a.recs[n * a.C + p];
        a.Xb[e] = p < 32 ? (x - mu[p]) * rs[p] : x;        // (the normalisation block holds 32 predictors; wider inputs come without input BatchNorm)
    }
    if constexpr (FWDK) {
        const long long totc = (long long)g.M * g.N;
        for (long long e = (long long)blockIdx.x * 256 + tid; e < totc; e += (long long)gridDim.x * 256) {
            const int m = (int)(e / g.N), n = (int)(e - (long long)m * g.N);
            const long long ng = a.idx ? (long long)a.idx[a.first + m] : a.first + m;
            float v = 0.0f;
            for (int k = 0; k < g.K; ++k) {
                const int p = c0 + k;
                const float x = a.recs[ng * a.C + p];
                v = fmaf(p < 32 ? (x - mu[p]) * rs[p] : x, g.B[(long long)k * g.ldb + n], v);
            }
            v += g.bias[n];
            if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
            g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
        }
    }
}

// Few rows, few predictors: the minibatch matrix, the input BatchNorm, the FIRST Dense layer (K = P <= 8: a handful of fused multiply-adds
// per value) and the SECOND layer's few-rows product (eh_fewrows_gemm_kernel: 16 x 16 output tile per workgroup, the k range over its
// waves) as ONE launch: the second layer's A operand -- 16 rows of the first layer's activations -- is computed where it is needed, from
// this workgroup's 16 normalised records in LDS, by the expression of eh_lform_prep_kernel<true> (the same fmaf chain: the same bits); the
// workgroups of the first column block also store it (and the normalised records) for the weight gradients and act' of the backward pass.
// Every workgroup takes the minibatch's BatchNorm statistics itself (the same sums in the same order); workgroup (0, 0) advances the running
// ones.  One launch boundary and one round of dependent loads fewer per step: the tutorial net's prep + first-layer launch was 5 us of 56.
// g: the second layer's product (g.A unused); f: the first layer (B = W_0 [P][out_0], bias, act, C = H_0, Z = its pre-activation for swish).
// KP: predictors of the first network, rounded up (2 or 4): its weight fragments are held in registers -- 16 waves leave 128 each (a first
// version sized for eight predictors spilled: 84 against 56 us per step)
template <int EPI, int KP>
__global__ __launch_bounds__(1024) void eh_fewrows_first_kernel(const EhLPrepArgs a, const EhGemmArgs f, const EhGemmArgs g, const int c0) {
    static_assert(EPI == EH_GEPI_BIAS_ACT && (KP == 2 || KP == 4), "forward product; at most four predictors");
    __shared__ float red[16][256];
    __shared__ float mu[32], rs[32], sred[32][64], xs[16][8], xraw[64][33];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = (int)(blockDim.x >> 6), nthr = (int)blockDim.x;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    EH_FSTAMP(0);
    // ---- the weights of this wave's first k round are asked for before anything else: they arrive behind the statistics ------------------
    const int kbeg = wave * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nc = min(n0 + r, g.N - 1);
    const float* const pb = g.B + (long long)(4 * q) * g.ldb + nc;
    f32x4_lf b[4], w0[4][KP], bq[4];
    auto load_round = [&](const int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kb = k0 + 16 * u;
            const bool ok = kb < kend;
            const int kk = ok ? kb + 4 * q : 0;
            bq[u] = *(const f32x4_lf*)(f.bias + kk);
#pragma unroll
            for (int pp = 0; pp < KP; ++pp) w0[u][pp] = *(const f32x4_lf*)(f.B + (long long)min(pp, f.K - 1) * f.ldb + kk);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[u][i] = ok ? pb[(long long)(kb + i) * g.ldb] : 0.0f;
        }
    };
    load_round(kbeg);
    float bias2 = 0.0f;                                  // (the epilogue's bias: asked for now, not behind the fold)
    if (tid < 256 && n0 + (tid & 15) < g.N) bias2 = g.bias[n0 + (tid & 15)];
    EH_FSTAMP(1);
    // ---- input BatchNorm statistics of the minibatch (train mode, small minibatch) or the image's (eh_lform_prep_kernel) ------------------
    // (every barrier of this kernel orders LDS traffic only: eh_lds_barrier does not wait for the stores of the first column block;
    //  what a later phase needs from global memory again -- the centre, the image's statistics, the running ones -- is asked for here)
    const int ngrp = nthr >> 5;                      // sample groups of 32 predictors each
    const bool useraw = a.bn_self && a.count <= 64 && a.P <= 32;      // the records the statistics read are kept (LDS) for the normalisation
    float m_img = 0.0f, r_img = 0.0f, run_m = 0.0f, run_v = 0.0f, cc_self = 0.0f;
    if (tid < 32) {
        m_img = a.meta[EH_IMG_BNM + tid]; r_img = a.meta[EH_IMG_BNR + tid];
        if (a.bn_update && blockIdx.x == 0 && blockIdx.y == 0 && tid < a.P) { run_m = a.bn_run[tid]; run_v = a.bn_run[32 + tid]; }
    }
    if (a.bn_self) {
        const int p = tid & 31, grp = tid >> 5;
        const long long nf = a.idx ? (long long)a.idx[a.first] : a.first;
        float s1 = 0.0f, s2 = 0.0f;
        if (p < a.P) {
            const float cc = a.recs[nf * a.C + p];
            cc_self = cc;
            for (int i = grp; i < a.count; i += ngrp) {
                const long long n = a.idx ? (long long)a.idx[a.first + i] : a.first + i;
                const float x = a.recs[n * a.C + p];
                if (useraw) xraw[i][p] = x;
                const float d = x - cc;
                s1 += d; s2 += d * d;
            }
        }
        EH_FSTAMP(2);
        sred[grp][p] = s1; sred[grp][32 + p] = s2;
        eh_lds_barrier();
    }
    EH_FSTAMP(3);
    if (tid < 32) {
        float m = m_img, rr = r_img;
        if ((a.bn_part || a.bn_self) && tid < a.P) {
            float s1 = 0.0f, s2 = 0.0f;
            if (a.bn_self) { for (int b = 0; b < ngrp; ++b) { s1 += sred[b][tid]; s2 += sred[b][32 + tid]; } }
            else for (int b = 0; b < a.bn_nblk; ++b) { s1 += a.bn_part[b * 64 + tid]; s2 += a.bn_part[b * 64 + 32 + tid]; }
            const float cnt = a.bn_n ? *a.bn_n : (float)a.count, cc = a.bn_self ? cc_self : a.bn_c[tid];      // (tid < 32: group 0, predictor tid -- its own centre)
            const float d = s1 / cnt, var = fmaxf(s2 / cnt - d * d, 0.0f);
            m = cc + d; rr = 1.0f / sqrtf(var + EH_BN_EPS);
            if (a.bn_update && blockIdx.x == 0 && blockIdx.y == 0) {
                const float rm = (1.0f - EH_BN_MOMENTUM) * run_m + EH_BN_MOMENTUM * m;
                const float rv = (1.0f - EH_BN_MOMENTUM) * run_v + EH_BN_MOMENTUM * (cnt > 1.0f ? cnt / (cnt - 1.0f) : 1.0f) * var;
                a.bn_run[tid] = rm; a.bn_run[32 + tid] = rv;
                a.meta[EH_IMG_BNM + tid] = rm;                          // what forward / eval (test mode) will use
                a.meta[EH_IMG_BNR + tid] = 1.0f / sqrtf(rv + EH_BN_EPS);
            }
        }
        mu[tid] = m; rs[tid] = rr;
    }
    eh_lds_barrier();
    EH_FSTAMP(4);
    // ---- this workgroup's 16 rows, normalised: LDS (all predictors of the first network: K = f.K <= 8) and, from the first column block, Xb
    if (tid < 16 * f.K) {
        const int rr = tid / f.K, k = tid - rr * f.K, m = min(m0 + rr, g.M - 1), pcol = c0 + k;
        float x;
        if (useraw) x = xraw[m][pcol];
        else { const long long ng = a.idx ? (long long)a.idx[a.first + m] : a.first + m; x = a.recs[ng * a.C + pcol]; }
        xs[rr][k] = pcol < 32 ? (x - mu[pcol]) * rs[pcol] : x;
    }
    if (blockIdx.x == 0) {                            // the minibatch matrix (every predictor column: networks of a MultiNN model read theirs from it)
        for (int e = tid; e < 16 * a.P; e += nthr) {
            const int rr = e / a.P, pc = e - rr * a.P, m = m0 + rr;
            if (m < g.M) {
                float x;
                if (useraw) x = xraw[m][pc];
                else { const long long ng = a.idx ? (long long)a.idx[a.first + m] : a.first + m; x = a.recs[ng * a.C + pc]; }
                a.Xb[(long long)m * a.P + pc] = pc < 32 ? (x - mu[pc]) * rs[pc] : x;
            }
        }
    }
    eh_lds_barrier();
    EH_FSTAMP(5);
    // ---- the second layer's product, its A operand made on the way ---------------------------------------------------------------------
    float xr[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) xr[k] = k < f.K ? xs[r][k] : 0.0f;
    const bool keep = blockIdx.x == 0 && m0 + r < g.M;
    f32x4_lf acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        f32x4_lf av[4];
        if (k0 != kbeg) load_round(k0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kb = k0 + 16 * u;
            const bool ok = kb < kend;
            f32x4_lf z;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = 0.0f;
#pragma unroll
                for (int pp = 0; pp < KP; ++pp)
                    if (pp < f.K) v = fmaf(xr[pp], w0[u][pp][i], v);
                z[i] = v + bq[u][i];
                av[u][i] = ok ? eh_act_rt(f.act, z[i]) : 0.0f;
            }
            if (keep && ok) {
                *(f32x4_lf*)(f.C + (long long)(m0 + r) * f.ldc + kb + 4 * q) = av[u];
                if (f.Z) *(f32x4_lf*)(f.Z + (long long)(m0 + r) * f.ldc + kb + 4 * q) = z;
            }
        }
        EH_FSTAMP(6);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], b[u][i], acc, 0, 0, 0);
    }
    EH_FSTAMP(7);
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * q + i) * 16 + r] = acc[i];
    eh_lds_barrier();
    EH_FSTAMP(8);
    if (tid < 256) {
        float v = 0.0f;
        for (int w = 0; w < nw; ++w) v += red[w][tid];
        const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
        if (m < g.M && n < g.N) {
            v += bias2;
            if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
            g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
        }
    }
    EH_FSTAMP(9);
}

// Mechanistic model + masked loss (+ its pullback), one sample per lane, between the forward and the backward GEMMs:
// O [K][ldo] raw NN outputs in -> (TRAIN) d loss / d O in place, one row of partial sums per workgroup
// [grad of the raw globals (8) | S | n_t (4) | Sy | Syy]; (eval) predictions / parameters out, metric sums per workgroup.
struct EhLMechArgs {
    float* O; long long ldo;
    float* part;              // [gridDim][EH_LMECH_PART] (train) / [gridDim][EH_EVAL_STATS * T] (eval)
    float* slab; int nrows; long long n_acc;      // (train, ONE workgroup) non-null: the work of eh_lform_tail_kernel done here
};
// the sums of the mechanistic stage -> the tail columns of the slab rows (row 0: the sums; the others: cleared)
__device__ __forceinline__ void eh_lform_tail_write(const float* tot, const EhNet& net, float* slab, int nrows, long long n_acc, int tid) {
    const int ntail = net.G + 1 + net.T + 2;
    for (int e = tid; e < nrows * ntail; e += 256) {
        const int row = e / ntail, c = e % ntail;
        float v = 0.0f;
        if (row == 0) {
            if (c < net.G) {
#pragma unroll
                for (int j = 0; j < EH_MAX_PARAMS; ++j)
                    if (j < net.n_par && ((net.par_kind >> (2 * j)) & 3u) == EH_PAR_GLOBAL && (int)((net.par_idx >> (4 * j)) & 15u) == c) v = tot[j];
            } else if (c == net.G) v = tot[8];
            else if (c <= net.G + net.T) v = tot[9 + (c - net.G - 1)];
            else v = tot[13 + (c - net.G - 1 - net.T)];
        }
        slab[(long long)row * n_acc + net.g_off + c] = v;
    }
}

enum { EH_LMECH_PART = 16 };
template <bool TRAIN, bool PROG, bool LPROG = false>
__global__ __launch_bounds__(256) void eh_lform_mech_kernel(const EhNet net, const EhStepArgs a, const EhLMechArgs m, const float* meta_g) {
    constexpr int SR = 64;
    __shared__ float OSs[4][16 * SR], SGs[4][16 * SR], RSs[4][(EH_MAX_FORC + EH_MAX_TARG) * SR], metas[EH_IMG_META], red[4][32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < EH_IMG_META; e += 256) metas[e] = meta_g[e];
    __syncthreads();
    float* const OS = OSs[wave]; float* const SG = SGs[wave]; float* const RS = RSs[wave];
    auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
    auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
    EhMechAcc MA;
    MA.clear();
    const int count = (int)a.count;
    for (int base = ((int)blockIdx.x * 4 + wave) * 64; base < count; base += (int)gridDim.x * 256) {
        const int n_loc = base + lane;
        const bool live = n_loc < count;
        const long long n_glb = live ? (a.idx ? (long long)a.idx[a.first + n_loc] : a.first + n_loc) : 0;
        const float* const rec = a.recs + n_glb * a.C;
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) RS[f * SR + lane] = (f < net.F && live) ? rec[net.P + f] : 0.0f;
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) RS[(EH_MAX_FORC + t) * SR + lane] = (t < net.T && live) ? rec[net.P + net.F + t] : __builtin_nanf("");
        for (int k = 0; k < net.K; ++k) {
            const float ov = live ? m.O[(long long)k * m.ldo + n_loc] : 0.0f;
            float pv = ov, sv = 1.0f;
            if (net.scale_nn) {
                float lo = 0.0f, sc = 0.0f;
#pragma unroll
                for (int j = 0; j < EH_MAX_PARAMS; ++j)
                    if (j < net.n_par && pkind(j) == EH_PAR_NEURAL && pidx(j) == k) { lo = metas[EH_IMG_LO + j]; sc = metas[EH_IMG_SC + j]; }
                const float sgm = eh_sigmoid(ov);
                pv = fmaf(sc, sgm, lo);
                sv = sc * sgm * (1.0f - sgm);
            }
            OS[k * SR + lane] = pv; SG[k * SR + lane] = sv;
        }
        eh_mech_stage_lane<TRAIN, PROG, LPROG>(net, a, lane, live, n_loc, SR, RS, OS, SG, metas, MA);      // (every access of a lane is to its own column: no cross-lane traffic)
        if constexpr (TRAIN) {
            if (live)
                for (int k = 0; k < net.K; ++k) m.O[(long long)k * m.ldo + n_loc] = OS[k * SR + lane];
        }
    }
    // one row of sums per workgroup (fixed order: deterministic)
    constexpr int NS = TRAIN ? EH_LMECH_PART : EH_EVAL_STATS * EH_MAX_TARG;
    float v[NS];
    if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) v[j] = (j < net.n_par && pkind(j) == EH_PAR_GLOBAL) ? MA.gacc[j] * metas[EH_IMG_DPHI + j] : 0.0f;
        v[8] = MA.lacc;
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t) v[9 + t] = MA.cacc[t];
        v[13] = MA.syacc; v[14] = MA.syyacc; v[15] = 0.0f;
    } else {
#pragma unroll
        for (int t = 0; t < EH_MAX_TARG; ++t)
#pragma unroll
            for (int k = 0; k < EH_EVAL_STATS; ++k) v[t * EH_EVAL_STATS + k] = MA.est[t][k];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float s = eh_wave_sum(v[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    const int nout = TRAIN ? EH_LMECH_PART : EH_EVAL_STATS * net.T;
    if constexpr (TRAIN) {
        if (m.slab) {             // one workgroup: its sums ARE the totals
            if (tid < EH_LMECH_PART) red[0][tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            __syncthreads();
            eh_lform_tail_write(red[0], net, m.slab, m.nrows, m.n_acc, tid);
            return;
        }
    }
    if (tid < nout) m.part[(long long)blockIdx.x * nout + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// rows of mechanistic-stage sums -> the tail of slab row 0 ([grad of the raw globals | S | n_t | Sy | Syy] in the fused kernels'
// column order); the same columns of the other slab rows are cleared.  One workgroup.
__device__ __forceinline__ void eh_lform_tail_sum(const float* part, int nblk, const EhNet& net, float* slab, int nrows, long long n_acc) {
    __shared__ float tot[EH_LMECH_PART], red[16][EH_LMECH_PART];
    const int tid = threadIdx.x, col = tid & 15, grp = tid >> 4;       // 16 row groups x 16 columns, fixed order: deterministic
    float s = 0.0f;
    for (int b = grp; b < nblk; b += 16) s += part[(long long)b * EH_LMECH_PART + col];
    red[grp][col] = s;
    __syncthreads();
    if (tid < EH_LMECH_PART) {
        float t = 0.0f;
        for (int q = 0; q < 16; ++q) t += red[q][tid];
        tot[tid] = t;
    }
    __syncthreads();
    eh_lform_tail_write(tot, net, slab, nrows, n_acc, tid);
}
__global__ __launch_bounds__(256) void eh_lform_tail_kernel(const float* part, int nblk, const EhNet net, float* slab, int nrows, long long n_acc) {
    eh_lform_tail_sum(part, nblk, net, slab, nrows, n_acc);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Few rows (B <= 256), the narrow END of the network as ONE launch: the reference's GPU tutorial net [1024,512,256,128,64] at its own
// batch of 64 ran 256->128, 128->64, 64->1, the mechanistic stage, and the deltas back down to the 256-wide layer as SEVEN dependent
// launches of 3.6-6.6 us each -- none of them has work for more than a few CUs, each is a launch boundary plus one or two memory round
// trips (profiles/r05/train_e2e_kernel_stats.csv).  Forward and delta products of a layer touch one ROW of the minibatch at a time, so
// the rows are independent all the way: workgroup b runs rows [R b, R b + R) through every layer of the suffix, the mechanistic model +
// masked loss + its pullback, and the delta chain back to the layer below the suffix, with the activations of its rows in LDS.  Products
// are row-vector x matrix on the vector ALU (a 16-row MFMA tile would be 1/16 full): the k range of a forward product is split over the
// workgroup's 1 024 threads (thread = 4 adjacent output columns x one k slice; 16-byte loads of the canonical [in][out] weights, one
// batch in flight per layer), the slices are added in k order through LDS; a delta product contracts over the contiguous dimension
// (lane = 4 adjacent columns, a lane group per k row, butterfly sum).  Every sum in a fixed order: deterministic.  What the weight-
// gradient products (grouped launch, eh_dw_group_kernel) need -- H_l, dZ_l -- goes to global memory on the way.
// Suffix: every layer from `first` on has in, out <= 256 (EH_LTAIL_MAXW); each workgroup streams its weights from L2 twice.
enum { EH_LTAIL_MAXL = EH_MAX_HIDDEN + 1, EH_LTAIL_MAXW = 256, EH_LTAIL_THREADS = 512, EH_LTAIL_RED = 4 * EH_LTAIL_THREADS, EH_LTAIL_RED2 = 1024, EH_LTAIL_PAD = 64,
       EH_LTAIL_FD = 8, EH_LTAIL_BU = 4 };      // weight fragments a thread keeps in flight: rows of its k slice (forward), passes over the k rows (delta)
struct EhLTailLayer {
    const float* W; const float* b;      // canonical [in][out] row-major weights, bias [out]
    float* H; float* Z;                  // (hidden layers) activations [B][out] out; pre-activations out when the layer's activation is swish, else null
    float* D;                            // (hidden layers) dZ_l [B][out] out
    int in, out, act, vec;               // vec: W 16-byte aligned and out % 4 == 0
    // geometry of the two products (eh_ltail_geometry, host): forward thread = (k slice tid >> lg_ng, column group tid & (ng - 1)) with
    // KS slices of kper rows, partial rows ostr = 1 << lg_ostr apart; delta product: 1 << lg_G lanes per k row
    int lg_ng, KS, kper, lg_ostr, lg_G, nloop, pad0, pad1;
};
struct EhLTailArgs {
    EhLTailLayer L[EH_LTAIL_MAXL];
    int nl, wmax, any_swish, act_below;
    const float* Hin; long long ldin;    // input of L[0]: [B][ldin]
    const float* Zin;                    // its pre-activation (the layer below is swish), else null
    float* Dbelow;                       // dZ of the layer below [B][L[0].in]; null: L[0] is the network's first layer
    float* O; long long ldo;             // this network's rows of O^T [K][ldo]: raw outputs -> d loss / d O
    float* part;                         // [gridDim][EH_LMECH_PART]
};
inline void eh_ltail_geometry(EhLTailLayer& T) {
    const int CV = T.vec ? 4 : 1;
    int lg = 0;
    while ((CV << lg) < T.out) ++lg;                      // ng = 1 << lg column groups
    T.lg_ng = lg;
    T.lg_ostr = lg + (T.vec ? 2 : 0);
    int KS = std::min(std::min((int)EH_LTAIL_THREADS >> lg, T.in), 64);          // (two levels of eight: at most 64 slices)
    KS = std::max(KS, 1);
    T.kper = (T.in + KS - 1) / KS;
    T.KS = (T.in + T.kper - 1) / T.kper;                  // slices that are not empty
    int lgG = 0;
    while (4 * (CV << lgG) < T.out && lgG < 6) ++lgG;     // a lane takes up to four pieces (of CV columns) of its k row
    T.lg_G = lgG;
    T.nloop = (T.out + (CV << lgG) - 1) / (CV << lgG);    // pieces per lane, <= 4
    T.pad0 = T.pad1 = 0;
}
inline size_t eh_ltail_lds_bytes(int R, int nl, int wmax, bool any_swish) {
    const size_t WP = (size_t)wmax + EH_LTAIL_PAD;
    return sizeof(float) * ((size_t)(nl + 1) * R * WP * (any_swish ? 3 : 2) + (size_t)R * (EH_LTAIL_RED + EH_LTAIL_RED2) + 2 * 16 * 64 +
                            (EH_MAX_FORC + EH_MAX_TARG) * 64 + ((EH_IMG_META + 3) & ~3) + (size_t)nl * wmax);
}
// sum over the 1 << lg lanes of a lane group (aligned, lg <= 6), every lane of the group gets it: DPP inside rows of 16, the crossbar for 32 / 64
__device__ __forceinline__ float eh_group_sum(float v, const int lg) {
    if (lg >= 1) v += eh_dpp<0xB1>(v);      // quad_perm:[1,0,3,2]
    if (lg >= 2) v += eh_dpp<0x4E>(v);      // quad_perm:[2,3,0,1]
    if (lg >= 3) v += eh_dpp<0x141>(v);     // row_half_mirror
    if (lg >= 4) v += eh_dpp<0x140>(v);     // row_mirror
    if (lg >= 5) v += __shfl_xor(v, 16, 64);
    if (lg >= 6) v += __shfl_xor(v, 32, 64);
    return v;
}
#ifndef EH_STAMP_L
#define EH_STAMP_L 0      // (diagnostic builds: the layer of the suffix whose phases get the inner stamps 12-14 / 6-7)
#endif
template <int R, bool PROG, bool LPROG>
__global__ __launch_bounds__(EH_LTAIL_THREADS, 1) void eh_lform_tailchain_kernel(const EhNet net, const EhStepArgs a, const EhLTailArgs t, const float* meta_g) {
    extern __shared__ __attribute__((aligned(16))) float eh_lt_smem[];
    constexpr int NTH = EH_LTAIL_THREADS, NWV = NTH / 64, SR = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = t.wmax, WP = W + EH_LTAIL_PAD, nl = t.nl;
    float* const hb = eh_lt_smem;                                        // [(nl + 1)][R][WP]: hb[0] = the input rows, hb[l + 1] = layer l's activations (last: raw outputs); zero beyond a row's width
    float* const zb = hb + (size_t)(nl + 1) * R * WP;                    // the same for pre-activations (swish layers only)
    float* const dd = zb + (t.any_swish ? (size_t)(nl + 1) * R * WP : 0); // [(nl + 1)][R][WP] deltas: dd[j] = d loss / d (input of layer j), dd[nl] = d loss / d (raw outputs); zero beyond a row's width
    float* const red = dd + (size_t)(nl + 1) * R * WP;                   // [R][EH_LTAIL_RED] k-slice partials
    float* const red2 = red + (size_t)R * EH_LTAIL_RED;                  // [R][EH_LTAIL_RED2] their sums in groups of eight
    float* const OS = red2 + (size_t)R * EH_LTAIL_RED2;                  // mechanistic stage: [16][SR] parameters -> d loss / d output
    float* const SG = OS + 16 * SR;
    float* const RS = SG + 16 * SR;
    float* const metas = RS + (EH_MAX_FORC + EH_MAX_TARG) * SR;
    float* const bb = metas + ((EH_IMG_META + 3) & ~3);                  // [nl][W] biases
    const int count = (int)a.count, row0 = (int)blockIdx.x * R;
    // The kernel is a chain of a dozen short phases; what each of them costs is a memory round trip plus the instructions every wave
    // issues (stamps of the first versions, tools/stamps_lform.py: 27 us = 65 k cycles, 7-12 k per phase -- ONE thread per output adding
    // 64 k-slice partials from LDS one after the other, butterfly sums issued one at a time, barriers waiting for global stores; then
    // 22 us: sixteen waves walking through index arithmetic with run-time divisors, a branch per predicated load, the finishing of a
    // delta product -- cross-lane sums and act' -- done by every lane for every row).  Hence: eight waves; the geometry of every
    // product comes from the host as shifts; rows in LDS are zero beyond their width, so the last k slice reads on instead of
    // branching; a lane of a delta product takes up to four pieces of its row, so that a row is eight lanes and its sum three DPP
    // steps; and the weights of the NEXT product are requested before the sums of the current one are finished, so that a phase
    // finds them in registers.  Everything that does not depend on the rows is asked for at the start, in one round of requests.
    EH_STAMP(0);
    eh_kernarg_warm_big<(int)(sizeof(EhNet) + sizeof(EhStepArgs) + sizeof(EhLTailArgs) + 8)>();
    EH_STAMP(1);
    // ---- weight fragments of the two kinds of product --------------------------------------------------------------------------------
    // forward: thread (ks, g) -- the first FD rows of its k slice, columns n0 .. n0 + 3 (or n0)
    constexpr int FD = EH_LTAIL_FD, BU = EH_LTAIL_BU;
    auto fwd_load = [&](const EhLTailLayer& L, f32x4 (&w)[FD]) {
        const int g = tid & ((1 << L.lg_ng) - 1), ks = tid >> L.lg_ng, n0 = L.vec ? 4 * g : g, k0 = ks * L.kper;
        if (n0 < L.out && ks < L.KS) {
#pragma unroll
            for (int u = 0; u < FD; ++u)
                if (u < L.kper) {                                         // (the same for every thread: a scalar branch)
                    const float* const q = L.W + (long long)min(k0 + u, L.in - 1) * L.out + n0;      // (rows past the end: the last row again, times the zeros behind the input)
                    if (L.vec) w[u] = *(const f32x4*)q; else w[u] = f32x4{*q, 0.0f, 0.0f, 0.0f};
                }
        }
    };
    // delta product: lane (kr, gl) of wave w -- rows kb + u RP + kr (u < BU, RP = rows of a pass of the workgroup), pieces gl + it G (it < nloop <= 4)
    auto bwd_load = [&](const EhLTailLayer& L, f32x4 (&w)[BU][4]) {
        const int lg_G = L.lg_G, rpw = 64 >> lg_G, gl = lane & ((1 << lg_G) - 1), kr = lane >> lg_G, CV = L.vec ? 4 : 1, kb = wave * rpw;
#pragma unroll
        for (int u = 0; u < BU; ++u)
            if (kb + u * NWV * rpw < L.in) {                              // (per wave: scalar)
                const float* const row = L.W + (long long)min(kb + u * NWV * rpw + kr, L.in - 1) * L.out;
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (it < L.nloop) {
                        const float* const q = row + min((gl + (it << lg_G)) * CV, L.out - CV);      // (columns past the end: the last ones again, times the zeros behind the delta)
                        if (L.vec) w[u][it] = *(const f32x4*)q; else w[u][it] = f32x4{*q, 0.0f, 0.0f, 0.0f};
                    }
            }
    };
    // this workgroup's rows of the input, the first product's weights, the records: all requested before anything is waited for
    float xin[R][(EH_LTAIL_MAXW + NTH - 1) / NTH], zin[R][(EH_LTAIL_MAXW + NTH - 1) / NTH];
    {
        const int in0 = t.L[0].in;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < (EH_LTAIL_MAXW + NTH - 1) / NTH; ++j) {
                const int k = tid + j * NTH, row = row0 + r;
                const bool ok = k < in0 && row < count;
                xin[r][j] = ok ? t.Hin[(long long)row * t.ldin + k] : 0.0f;
                zin[r][j] = (ok && t.Zin) ? t.Zin[(long long)row * in0 + k] : 0.0f;
            }
    }
    f32x4 wf[FD];
    if (nl > 1) fwd_load(t.L[0], wf);
    // The suffix's weights (one contiguous range of theta: [W | b] layer after layer) were rewritten by the optimiser kernel of the step
    // before: the first request for a line from this XCD goes past its L2 -- about 3 k cycles, which a fetch one phase ahead does not
    // cover.  The workgroups of an XCD (blockIdx mod 8 under round-robin dispatch; nothing depends on it) share the range between them:
    // one load per 64-byte line, summed into a value that is looked at when the kernel ends.  (Every workgroup touching the whole range
    // was tried first: it doubles the bytes a CU pulls through its 64-byte-per-clock port.)
    float touch[4];
    {
        const float* const w0 = t.L[0].W;
        const long long wn = (t.L[nl - 1].b + t.L[nl - 1].out) - w0;
        const int part8 = ((int)blockIdx.x >> 3) & 7;
        const long long seg = ((wn + 7) / 8 + 15) & ~15ll, s0 = part8 * seg, s1 = min(wn, s0 + seg);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long e = s0 + 16ll * tid + 16ll * NTH * u; touch[u] = w0[e < s1 ? e : 0]; }      // (up to 32 k floats per workgroup, 256 k per XCD)
    }
    for (int e = tid; e < (nl + 1) * R * WP * (t.any_swish ? 3 : 2); e += NTH) hb[e] = 0.0f;
    if (wave == 0) {
        const int n_loc = row0 + lane;
        const bool live = lane < R && n_loc < count;
        const long long n_glb = live ? (a.idx ? (long long)a.idx[a.first + n_loc] : a.first + n_loc) : 0;
        const float* const rec = a.recs + n_glb * a.C;
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) RS[f * SR + lane] = (f < net.F && live) ? rec[net.P + f] : 0.0f;
#pragma unroll
        for (int tt = 0; tt < EH_MAX_TARG; ++tt) RS[(EH_MAX_FORC + tt) * SR + lane] = (tt < net.T && live) ? rec[net.P + net.F + tt] : __builtin_nanf("");
    } else if (wave == 1) {
        for (int e = lane; e < EH_IMG_META; e += 64) metas[e] = meta_g[e];
    }
    for (int l = wave; l < nl; l += NWV) {
        const float* const bp = t.L[l].b; const int out = t.L[l].out;
        for (int n = lane; n < out; n += 64) bb[l * W + n] = bp[n];
    }
    eh_lds_barrier();                                                    // (the zeros are down before the rows go on top of them)
    {
        const int in0 = t.L[0].in;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < (EH_LTAIL_MAXW + NTH - 1) / NTH; ++j) {
                const int k = tid + j * NTH;
                if (k < in0) { hb[r * WP + k] = xin[r][j]; if (t.Zin) zb[r * WP + k] = zin[r][j]; }
            }
    }
    eh_lds_barrier();
    EH_STAMP(2);
    // ---- forward through the suffix ---------------------------------------------------------------------------------------------
    // (the network's output layer, at most 16 outputs, is not one of these products: it runs with the mechanistic stage below)
    f32x4 wb[BU][4];
    for (int l = 0; l + 1 < nl; ++l) {
        const EhLTailLayer L = t.L[l];
        const int in = L.in, out = L.out, KS = L.KS, kper = L.kper, lg_ostr = L.lg_ostr, ostr = 1 << lg_ostr;
        const int g = tid & ((1 << L.lg_ng) - 1), ks = tid >> L.lg_ng, n0 = L.vec ? 4 * g : g;
        const float* const hin = hb + (size_t)l * R * WP;
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
        if (n0 < out && ks < KS) {
            const int k0 = ks * kper;
            for (int kb = 0; kb < kper; kb += FD) {                       // (kper is the same for every thread: the branches below are scalar)
                if (kb > 0) {                                             // a slice of more than FD rows: the later batches are fetched here
#pragma unroll
                    for (int u = 0; u < FD; ++u)
                        if (kb + u < kper) {
                            const float* const q = L.W + (long long)min(k0 + kb + u, in - 1) * out + n0;
                            if (L.vec) wf[u] = *(const f32x4*)q; else wf[u] = f32x4{*q, 0.0f, 0.0f, 0.0f};
                        }
                }
#pragma unroll
                for (int u = 0; u < FD; ++u)
                    if (kb + u < kper) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float hv = hin[r * WP + k0 + kb + u];
                            if (L.vec) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(hv, wf[u][c], acc[r][c]);
                            } else acc[r][0] = fmaf(hv, wf[u][0], acc[r][0]);
                        }
                    }
            }
        }
        if (l == EH_STAMP_L) EH_STAMP(12);
        if (ks < KS) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (L.vec) *(f32x4*)&red[r * EH_LTAIL_RED + ks * ostr + n0] = f32x4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
                else red[r * EH_LTAIL_RED + ks * ostr + n0] = acc[r][0];
            }
        }
        // the next product's weights are on their way while this one's sums are finished
        if (l + 2 < nl) fwd_load(t.L[l + 1], wf);
        else if (l > 0 || t.Dbelow) bwd_load(t.L[l], wb);        // (the last of them: the first delta product's, across the output layer and the mechanistic stage)
        eh_lds_barrier();
        // the k slices of an output are added in two levels -- groups of eight, then the groups -- both in k order (deterministic), the
        // first by as many threads as there are groups x outputs
        const float* src = red;
        int nsrc = KS, sstr = EH_LTAIL_RED;
        if (KS > 8) {
            const int KS1 = (KS + 7) >> 3;
            for (int e = tid; e < (KS1 << lg_ostr); e += NTH) {
                const int j = e >> lg_ostr, n = e & (ostr - 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float* const pp = red + r * EH_LTAIL_RED + ((8 * j) << lg_ostr) + n;
                    float pv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = pp[min(i, KS - 1 - 8 * j) << lg_ostr];
                    float v = pv[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) v += 8 * j + i < KS ? pv[i] : 0.0f;
                    red2[r * EH_LTAIL_RED2 + e] = v;
                }
            }
            eh_lds_barrier();
            src = red2; nsrc = KS1; sstr = EH_LTAIL_RED2;
        }
        if (l == EH_STAMP_L) EH_STAMP(13);
        const bool last = l + 1 == nl;
        float* const hout = hb + (size_t)(l + 1) * R * WP;
        float* const zout = zb + (size_t)(l + 1) * R * WP;
        for (int n = tid; n < out; n += NTH) {
            const float bias = bb[l * W + n];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float pv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = src[r * sstr + (min(i, nsrc - 1) << lg_ostr) + n];
                float v = pv[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) v += i < nsrc ? pv[i] : 0.0f;
                v += bias;
                if (!last) {
                    const float hv = eh_act_rt(L.act, v);
                    hout[r * WP + n] = hv;
                    if (L.act == EH_ACT_SWISH) zout[r * WP + n] = v;
                } else hout[r * WP + n] = v;
            }
        }
        if (l == EH_STAMP_L) EH_STAMP(14);
        eh_lds_barrier();
        EH_STAMP(3 + l);
    }
    // ---- the output layer (K <= 16 outputs): an output per wave, lanes over k, the sum on the DPP network -----------------------------
    {
        const EhLTailLayer T = t.L[nl - 1];
        const float* const hin = hb + (size_t)(nl - 1) * R * WP;
        float* const oraw = hb + (size_t)nl * R * WP;
        for (int n = wave; n < T.out; n += NWV) {
            float wv[EH_LTAIL_MAXW / 64];
#pragma unroll
            for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) { const int k = lane + 64 * j; wv[j] = k < T.in ? T.W[(long long)k * T.out + n] : 0.0f; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float sacc = 0.0f;
#pragma unroll
                for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) sacc = fmaf(hin[r * WP + lane + 64 * j], wv[j], sacc);
                const float sum = eh_wave_sum(sacc);
                if (lane == 0) oraw[r * WP + n] = sum + bb[(nl - 1) * W + n];
            }
        }
    }
    eh_lds_barrier();
    EH_STAMP(3 + nl - 1);
    // ---- mechanistic model + masked loss + its pullback: wave 0, lane r = row r of this workgroup ---------------------------------
    float* const dcur0 = dd + (size_t)nl * R * WP;
    float* const psum = red;                                             // (the k-slice partials are done with) this workgroup's row of partial sums
    if (wave == 0) {
        auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
        auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
        const float* const oraw = hb + (size_t)nl * R * WP;
        const int K = t.L[nl - 1].out;
        const int n_loc = row0 + lane;
        const bool live = lane < R && n_loc < count;
        for (int k = 0; k < 16; ++k) { OS[k * SR + lane] = 0.0f; SG[k * SR + lane] = 1.0f; }
        for (int k = 0; k < K; ++k) {
            const float ov = live ? oraw[lane * WP + k] : 0.0f;
            float pv = ov, sv = 1.0f;
            if (net.scale_nn) {
                float lo = 0.0f, sc = 0.0f;
#pragma unroll
                for (int j = 0; j < EH_MAX_PARAMS; ++j)
                    if (j < net.n_par && pkind(j) == EH_PAR_NEURAL && pidx(j) == k) { lo = metas[EH_IMG_LO + j]; sc = metas[EH_IMG_SC + j]; }
                const float sgm = eh_sigmoid(ov);
                pv = fmaf(sc, sgm, lo);
                sv = sc * sgm * (1.0f - sgm);
            }
            OS[k * SR + lane] = pv; SG[k * SR + lane] = sv;
        }
        EhMechAcc MA;
        MA.clear();
        eh_mech_stage_lane<true, PROG, LPROG>(net, a, lane, live, n_loc, SR, RS, OS, SG, metas, MA);
        if (lane < R) {
            for (int k = 0; k < K; ++k) {
                const float d = live ? OS[k * SR + lane] : 0.0f;
                dcur0[lane * WP + k] = d;
            }
        }
        float v[EH_LMECH_PART];
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) v[j] = (j < net.n_par && pkind(j) == EH_PAR_GLOBAL) ? MA.gacc[j] * metas[EH_IMG_DPHI + j] : 0.0f;
        v[8] = MA.lacc;
#pragma unroll
        for (int tt = 0; tt < EH_MAX_TARG; ++tt) v[9 + tt] = MA.cacc[tt];
        v[13] = MA.syacc; v[14] = MA.syyacc; v[15] = 0.0f;
        // (rows r < R <= 4 are the only lanes with a sample: their sums in lane order, no butterfly over 64 lanes -- one store per column from lane k)
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < EH_LMECH_PART; ++k) {
            float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 0));
#pragma unroll
            for (int r = 1; r < R; ++r) s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), r));
            mine = lane == k ? s : mine;
        }
        if (lane < EH_LMECH_PART) psum[lane] = mine;
    }
    eh_lds_barrier();
    EH_STAMP(8);
    // ---- deltas back down: dZ_{l-1}[r][k] = (sum_n dZ_l[r][n] W_l[k][n]) act'(h_{l-1}[r][k]) --------------------------------------
    if (nl - 1 > 0 || t.Dbelow) {        // across the output layer: a thread per k, its K weights are adjacent
        const EhLTailLayer T = t.L[nl - 1];
        const int actp = nl - 1 > 0 ? t.L[nl - 2].act : t.act_below;
        const float* const hp = (actp == EH_ACT_SWISH ? zb : hb) + (size_t)(nl - 1) * R * WP;
        float* const dn = dd + (size_t)(nl - 1) * R * WP;
        for (int k = tid; k < T.in; k += NTH) {
            float wv[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) wv[n] = n < T.out ? T.W[(long long)k * T.out + n] : 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float sacc = 0.0f;
#pragma unroll
                for (int n = 0; n < 16; ++n) sacc = n < T.out ? fmaf(dcur0[r * WP + n], wv[n], sacc) : sacc;
                dn[r * WP + k] = sacc * eh_dact_rt(actp, hp[r * WP + k]);
            }
        }
        eh_lds_barrier();
        EH_STAMP(9);
    }
    for (int l = nl - 2; l >= 0; --l) {
        if (l == 0 && !t.Dbelow) break;
        const EhLTailLayer L = t.L[l];
        const int in = L.in, out = L.out, CV = L.vec ? 4 : 1, lg_G = L.lg_G;
        const int actp = l > 0 ? t.L[l - 1].act : t.act_below;
        const float* const dz = dd + (size_t)(l + 1) * R * WP;
        float* const dn = dd + (size_t)l * R * WP;
        const float* const hp = (actp == EH_ACT_SWISH ? zb : hb) + (size_t)l * R * WP;
        const int rpw = 64 >> lg_G;                                      // k rows per wave and pass
        const int gl = lane & ((1 << lg_G) - 1), kr = lane >> lg_G;
        for (int kb = wave * rpw; kb < in; kb += NWV * rpw * BU) {
            if (kb != wave * rpw) {                                      // more than BU passes: the later rows' weights are fetched here
                EhLTailLayer Lk = L;
                Lk.W = L.W + (long long)(kb - wave * rpw) * out; Lk.in = in - (kb - wave * rpw);
                bwd_load(Lk, wb);
            }
            float p[BU][R];
#pragma unroll
            for (int u = 0; u < BU; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) p[u][r] = 0.0f;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (it < L.nloop) {
                    const int n0 = (gl + (it << lg_G)) * CV;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (L.vec) {
                            const f32x4 d4 = *(const f32x4*)&dz[r * WP + n0];
#pragma unroll
                            for (int u = 0; u < BU; ++u)
                                if (kb + u * NWV * rpw < in) p[u][r] = fmaf(d4[3], wb[u][it][3], fmaf(d4[2], wb[u][it][2], fmaf(d4[1], wb[u][it][1], fmaf(d4[0], wb[u][it][0], p[u][r]))));
                        } else {
                            const float d1 = dz[r * WP + n0];
#pragma unroll
                            for (int u = 0; u < BU; ++u)
                                if (kb + u * NWV * rpw < in) p[u][r] = fmaf(d1, wb[u][it][0], p[u][r]);
                        }
                    }
                }
            if (l == EH_STAMP_L) EH_STAMP(6);
#pragma unroll
            for (int u = 0; u < BU; ++u)
                if (kb + u * NWV * rpw < in) {
                    const int k = kb + u * NWV * rpw + kr;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float sum = eh_group_sum(p[u][r], lg_G);
                        if (gl == 0 && k < in) dn[r * WP + k] = sum * eh_dact_rt(actp, hp[r * WP + k]);
                    }
                }
        }
        if (l == EH_STAMP_L) EH_STAMP(7);
        // the next delta product's weights while this one's rows go out
        if (l - 1 > 0 || (l - 1 == 0 && t.Dbelow)) bwd_load(t.L[l - 1], wb);
#ifdef EH_STAMP_BWD
        if (l == EH_STAMP_L) EH_STAMP(13);
#endif
        eh_lds_barrier();
        EH_STAMP(9 + (nl - 1 - l));
    }
    // ---- everything the weight-gradient products need goes out NOW, in one round of stores: a store in the middle of the chain made
    // the next wait for a prefetched weight a wait for the store's acknowledgement as well (loads and stores share one counter)
    for (int j = 0; j < nl; ++j) {
        const EhLTailLayer L = t.L[j];
        const bool hidden = j + 1 < nl;
        float* const dst = j > 0 ? t.L[j - 1].D : t.Dbelow;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (row < count) {
                if (hidden)
                    for (int n = tid; n < L.out; n += NTH) {
                        L.H[(long long)row * L.out + n] = hb[((size_t)(j + 1) * R + r) * WP + n];
                        if (L.Z) L.Z[(long long)row * L.out + n] = zb[((size_t)(j + 1) * R + r) * WP + n];
                    }
                if (dst)
                    for (int k = tid; k < L.in; k += NTH) dst[(long long)row * L.in + k] = dd[((size_t)j * R + r) * WP + k];
            }
        }
    }
    {
        const int K = t.L[nl - 1].out;
        for (int e = tid; e < R * K; e += NTH) {
            const int r = e / K, k = e - r * K, row = row0 + r;
            if (row < count) t.O[(long long)k * t.ldo + row] = dd[((size_t)nl * R + r) * WP + k];
        }
        if (tid < EH_LMECH_PART) t.part[(long long)blockIdx.x * EH_LMECH_PART + tid] = psum[tid];
    }
    if (touch[0] + touch[1] + touch[2] + touch[3] == -1.2345e-30f) t.part[(long long)blockIdx.x * EH_LMECH_PART + EH_LMECH_PART - 1] = 0.0f;      // (keeps the warming loads; column 15 is padding, 0 either way)
    EH_STAMP(15);
}


// The grouped weight-gradient launch of a few-rows step with the optimiser in its epilogues (EhLApply above): every workgroup first takes
// the step's normalisation from the mechanistic stage's partial rows (the sums in the order of eh_lform_tail_sum), then runs its product
// and puts its elements through the update rule; the last workgroup updates the global parameters, writes the loss and advances the
// running beta products -- what eh_lform_tail_kernel + eh_reduce_kernel<APPLY> did in two more launches.  One target, a loss without
// batch moments, no weight_l2 (the host checks; everything else takes the slab and the reduce kernel as before).
__global__ __launch_bounds__(256) void eh_dw_apply_kernel(const EhGemmGroup G, const EhThinGroup T, const EhNet net, const EhLApply ap) {
    __shared__ float tot[EH_LMECH_PART], red[16][EH_LMECH_PART];
    __shared__ EhLApplyS S;
    const int tid = threadIdx.x;
#ifdef EH_STAMPS
    if (ap.stamps && (int)blockIdx.x == ap.stamp_wg && tid == 0) { ap.stamps[0] = __builtin_readcyclecounter(); ap.stamps[1] = wall_clock64(); }
#endif
    if (ap.tot) {
        if (tid < EH_LMECH_PART) tot[tid] = ap.tot[tid];
    } else {
        const int col = tid & 15, grp = tid >> 4;                        // 16 row groups x 16 columns, fixed order: deterministic (eh_lform_tail_sum)
        float s = 0.0f;
        for (int b = grp; b < ap.nblk; b += 16) s += ap.part[(long long)b * EH_LMECH_PART + col];
        red[grp][col] = s;
        __syncthreads();
        if (tid < EH_LMECH_PART) {
            float t = 0.0f;
            for (int q = 0; q < 16; ++q) t += red[q][tid];
            tot[tid] = t;
        }
    }
    __syncthreads();
    float scale, loss;
    eh_loss_finish(ap.loss_kind, tot[8], tot[9], tot[13], tot[14], scale, loss, ap.im.agg_a);      // [grad of the raw globals (8) | S | n_t (4) | Sy | Syy]
    const bool go = tot[9] > 0.0f;
    if (tid == 0) {
        S.slab = ap.slab; S.theta = ap.theta; S.m = ap.m; S.v = ap.v; S.o = ap.o; S.scale = scale; S.bt1 = ap.sc_in[0]; S.bt2 = ap.sc_in[1];
        S.go = go ? 1 : 0;
        S.use_m = (ap.o.rule == EH_OPT_ADAM || ap.o.rule == EH_OPT_ADAMW) ? 1 : 0;
        S.use_v = (S.use_m || ap.o.rule == EH_OPT_RMSPROP) ? 1 : 0;
        S.stamps = (int)blockIdx.x == ap.stamp_wg ? ap.stamps : nullptr;
    }
    __syncthreads();
    EH_LSTAMP(&S, 1);
    const int nthin = T.t0[T.n], ntile = G.t0[G.n];
    if ((int)blockIdx.x == nthin + ntile) {                              // the global parameters, the loss, the beta products
        const int ng = net.n_theta - net.g_off;
        if (tid < ng && go) {
            float gsum = 0.0f;
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j)
                if (j < net.n_par && ((net.par_kind >> (2 * j)) & 3u) == EH_PAR_GLOBAL && (int)((net.par_idx >> (4 * j)) & 15u) == tid) gsum = tot[j];
            const int idx = net.g_off + tid;
            float th = ap.theta[idx], mm = S.use_m ? ap.m[idx] : 0.0f, vv = S.use_v ? ap.v[idx] : 0.0f;
            eh_opt_update(ap.o, gsum * scale, S.bt1, S.bt2, th, mm, vv);
            ap.theta[idx] = th;
            if (S.use_m) ap.m[idx] = mm;
            if (S.use_v) ap.v[idx] = vv;
            eh_image_store(ap.im, idx, th);
        }
        if (tid == 0) {
            ap.sc_out[0] = go ? S.bt1 * ap.o.b1 : S.bt1;
            ap.sc_out[1] = go ? S.bt2 * ap.o.b2 : S.bt2;
            ap.gradbuf[net.n_theta] = loss;                             // (what eh_reduce_kernel leaves behind the gradient: loss, count, Sy, Syy -- the gradient itself is not kept)
            ap.gradbuf[net.n_theta + 1] = tot[9]; ap.gradbuf[net.n_theta + 2] = tot[13]; ap.gradbuf[net.n_theta + 3] = tot[14];
            if (ap.loss_slot) *ap.loss_slot = loss;
        }
        return;
    }
    if ((int)blockIdx.x < nthin) {
        int i = 0;
        while (i + 1 < T.n && (int)blockIdx.x >= T.t0[i + 1]) ++i;
        const int t = (int)blockIdx.x - T.t0[i], gx = T.gx[i];
        const EhThinArgs a = T.a[i];
        eh_thin_gemm_tile<true>(a, t % gx, t / gx, &S);
        EH_LSTAMP(&S, 5);
    } else {
        const int b = (int)blockIdx.x - nthin;
        int i = 0;
        while (i + 1 < G.n && b >= G.t0[i + 1]) ++i;
        const int t = b - G.t0[i], gx = G.gx[i], gy = G.gy[i];
        const EhGemmArgs g = G.g[i];
        eh_gemm_tile<true, false, EH_GEPI_APPLY, true, 64>(g, t % gx, (t / gx) % gy, t / (gx * gy), &S);
        EH_LSTAMP(&S, 5);
    }
}

// ---- the same launch for minibatches of at most 64 rows (the reference's default batchsize) ------------------------------------------------
// Stamps of eh_dw_apply_kernel at batch 64 (tools/stamps_lform.py, EH_STAMP_DW): a 64 x 64 tile took 20 k cycles -- 2.5-3.5 k until the
// step's sums were in, 10.5 k for a product 64 deep (its loads asked for only then; four LDS-staged k steps, each a barrier pair), 5 k for
// sixteen updates and 32-48 stores per thread -- and a thin product three dependent round trips (thin operand -> LDS, wide loads, parameters).
// Every memory round trip of data a previous launch wrote is ~3 k cycles, so here everything a workgroup will read is requested in its
// first instructions -- the step's sums, the beta products, the operands, the parameters and their moments -- and waited for once:
//   tiled products: 32 x 32 tiles (4 x the workgroups: 698 for the tutorial net, all resident), the <= 64 samples split over the four waves
//     (<= 16 each: eight 32x32x2 MFMAs whose operands come straight from global memory in the MFMA's own lane layout -- rows of H^T and dZ
//     are contiguous along m / n, no LDS staging), the four partial tiles folded in LDS in a fixed order; four parameters per thread;
//   thin products: eh_thin_gemm_tile's arithmetic in its order (the same bits), the wide loads and the parameters requested before the thin
//     operand is staged.
// (The same kernel sized for 256 rows -- 64 + 64 operand registers per lane in flight -- was built and is not kept: 19.8 / 28.3 us per launch at
//  128 / 256 rows, no better than the staged 64 x 64 tiles (20.0) or the two-chunk products + reduction (17.2 + 8.4): 59.7 / 61.3 / 73.2 /
//  82.3 us per step at 96 / 128 / 192 / 256 rows against 57.5 / 61.3 / 70.1 / 77.5 without it.)
__device__ __forceinline__ void eh_lapply64_norm(const EhLApply& ap, const float* totl, float& scale, bool& go) {
    float loss;
    eh_loss_finish(ap.loss_kind, totl[8], totl[9], totl[13], totl[14], scale, loss, ap.im.agg_a);
    go = totl[9] > 0.0f;
}
__global__ __launch_bounds__(256) void eh_dw_apply64_kernel(const EhGemmGroup G, const EhThinGroup T, const EhNet net, const EhLApply ap) {
    __shared__ float totl[EH_LMECH_PART];
    __shared__ __attribute__((aligned(16))) float red[4][32][40];      // (row stride 40: the two half-waves of an accumulator store land in disjoint banks)
    __shared__ float csr[8][32];
    __shared__ float sT[8][64], redq[4][64][10], redt[4][8];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef EH_STAMPS
#define EH_A64(i) do { __builtin_amdgcn_sched_barrier(0); if (ap.stamps && (int)blockIdx.x == ap.stamp_wg && tid == 0) { ap.stamps[2 * (i)] = __builtin_readcyclecounter(); ap.stamps[2 * (i) + 1] = wall_clock64(); } __builtin_amdgcn_sched_barrier(0); } while (0)
    if (ap.stamps && tid == 0) atomicMin(ap.stamps + 28, (unsigned long long)wall_clock64());      // (the launch's span: first start, last end -- the host resets the pair)
#else
#define EH_A64(i)
#endif
    EH_A64(0);
    // the tables of products are looked up by a dependent chain of scalar loads (which product? -> its tile counts -> its arguments): ten cold
    // lines one after the other were 15-26 k cycles before a workgroup's first vector load went out (stamps); all lines at once: one round trip
    // (asking for every line of the 2.8 KB of arguments at once was no better: 7-8 k cycles for that round alone.)  Two rounds: the tables of
    // first tiles at fixed offsets -> which product; its arguments -> everything else
    int gt[EH_GEMM_GROUP + 1], tt[EH_GEMM_GROUP + 1];
#pragma unroll
    for (int j = 0; j <= EH_GEMM_GROUP; ++j) { gt[j] = G.t0[j]; tt[j] = T.t0[j]; }
    const int gn = G.n, tn = T.n;
    int nthin = 0, ntile = 0;
#pragma unroll
    for (int j = 1; j <= EH_GEMM_GROUP; ++j) { if (j == tn) nthin = tt[j]; if (j == gn) ntile = gt[j]; }
    EH_A64(12);
    const float tv = tid < EH_LMECH_PART ? ap.tot[tid] : 0.0f;
    const float bt1 = ap.sc_in[0], bt2 = ap.sc_in[1];
    const bool use_m = ap.o.rule == EH_OPT_ADAM || ap.o.rule == EH_OPT_ADAMW, use_v = use_m || ap.o.rule == EH_OPT_RMSPROP;
    eh_gfloat* const tp = (eh_gfloat*)ap.theta; eh_gfloat* const mp = (eh_gfloat*)ap.m; eh_gfloat* const vp = (eh_gfloat*)ap.v;
    // workgroup -> work: [0, 8 Q) tile slots, then the thin products, then one for the global parameters.  Slot s runs tile (s % 8) Q + s / 8:
    // consecutive workgroups land on consecutive XCDs, so XCD x works on tiles [x Q, (x + 1) Q) -- a contiguous range of rows of every
    // product, the same one every step: its L2 fetches an eighth of H^T (and all of dZ) instead of all of both (TCC_EA0_RDREQ 101 k -> 62 k
    // lines per launch).  What is left are the parameters and moments themselves: an XCD's L2 does not keep lines across a launch boundary
    // -- pulling them in from the delta products one and two launches earlier (a side job per workgroup, the consumer's own XCD ranges)
    // added 49 k lines to those launches and took none off this one (48.2 -> 48.9 us per step; the same for the chain kernel's weights)
    const int Q = (ntile + 7) >> 3, slots = 8 * Q;
    const int wg = (int)blockIdx.x;
    if (wg == slots + nthin) {                                           // the global parameters, the loss, the beta products (eh_dw_apply_kernel)
        if (tid < EH_LMECH_PART) totl[tid] = tv;
        __syncthreads();
        float scale, loss;
        eh_loss_finish(ap.loss_kind, totl[8], totl[9], totl[13], totl[14], scale, loss, ap.im.agg_a);
        const bool go = totl[9] > 0.0f;
        const int ng = net.n_theta - net.g_off;
        if (tid < ng && go) {
            float gsum = 0.0f;
#pragma unroll
            for (int j = 0; j < EH_MAX_PARAMS; ++j)
                if (j < net.n_par && ((net.par_kind >> (2 * j)) & 3u) == EH_PAR_GLOBAL && (int)((net.par_idx >> (4 * j)) & 15u) == tid) gsum = totl[j];
            const int idx = net.g_off + tid;
            float th = ap.theta[idx], mm = use_m ? ap.m[idx] : 0.0f, vv = use_v ? ap.v[idx] : 0.0f;
            eh_opt_update(ap.o, gsum * scale, bt1, bt2, th, mm, vv);
            ap.theta[idx] = th;
            if (use_m) ap.m[idx] = mm;
            if (use_v) ap.v[idx] = vv;
            eh_image_store(ap.im, idx, th);
        }
        if (tid == 0) {
            ap.sc_out[0] = go ? bt1 * ap.o.b1 : bt1;
            ap.sc_out[1] = go ? bt2 * ap.o.b2 : bt2;
            ap.gradbuf[net.n_theta] = loss;
            ap.gradbuf[net.n_theta + 1] = totl[9]; ap.gradbuf[net.n_theta + 2] = totl[13]; ap.gradbuf[net.n_theta + 3] = totl[14];
            if (ap.loss_slot) *ap.loss_slot = loss;
        }
        return;
    }
    if (wg >= slots) {
        const int tb = wg - slots;
        int i = 0, ti = 0;
#pragma unroll
        for (int j = 1; j < EH_GEMM_GROUP; ++j)
            if (j < tn && tb >= tt[j]) { i = j; ti = tt[j]; }
        const EhThinArgs a = T.a[i];
        const int bx = tb - ti;                                          // (one slab row: gx workgroups per product)
        const int cl = lane, q = wave, col = bx * 64 + cl;
        const bool live = col < a.ncols;
        long long ix[9];
        float th[9], mm[9], vv[9];
        if (q == 0 && live) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                const float* const slot = j < 8 ? a.C + (long long)col * a.c_col + (long long)min(j, a.J - 1) * a.c_j : (a.cs_wide ? a.cs_wide + col : a.C);
                ix[j] = slot - ap.slab;
                th[j] = on ? tp[ix[j]] : 0.0f;
                mm[j] = (on && use_m) ? mp[ix[j]] : 0.0f;
                vv[j] = (on && use_v) ? vp[ix[j]] : 0.0f;
            }
        }
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int bl = q + 4 * u; w[u] = (live && bl < a.K) ? a.wide[(long long)bl * a.ldw + col] : 0.0f; }
        for (int e = tid; e < a.J * 64; e += 256) {
            const int j = a.tsb == 1 ? e / 64 : e % a.J, bl = a.tsb == 1 ? e % 64 : e / a.J;
            sT[j][bl] = bl < a.K ? a.thin[(long long)bl * a.tsb + (long long)j * a.tsj] : 0.0f;
        }
        if (tid < EH_LMECH_PART) totl[tid] = tv;
        __syncthreads();
        float acc[8], cst[8], csw = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j] = 0.0f; cst[j] = 0.0f; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int bl = q + 4 * u;
            if (bl < a.K) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < a.J) { const float t = sT[j][bl]; acc[j] = fmaf(t, w[u], acc[j]); cst[j] += t; }
                csw += w[u];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) redq[q][cl][j] = acc[j];
        redq[q][cl][8] = csw;
        if (cl == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) redt[q][j] = cst[j];
        }
        __syncthreads();
        float scale; bool go;
        eh_lapply64_norm(ap, totl, scale, go);
        if (!go) return;
        if (q == 0 && live) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                if (on) eh_opt_update(ap.o, ((redq[0][cl][j] + redq[1][cl][j]) + (redq[2][cl][j] + redq[3][cl][j])) * scale, bt1, bt2, th[j], mm[j], vv[j]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool on = j < 8 ? j < a.J : a.cs_wide != nullptr;
                if (on) {
                    tp[ix[j]] = th[j];
                    if (use_m) mp[ix[j]] = mm[j];
                    if (use_v) vp[ix[j]] = vv[j];
                }
            }
        }
        if (a.cs_thin && bx == 0 && tid < a.J) {
            const long long idx = (a.cs_thin + tid) - ap.slab;
            float t1 = tp[idx], m1 = use_m ? mp[idx] : 0.0f, v1 = use_v ? vp[idx] : 0.0f;
            eh_opt_update(ap.o, ((redt[0][tid] + redt[1][tid]) + (redt[2][tid] + redt[3][tid])) * scale, bt1, bt2, t1, m1, v1);
            tp[idx] = t1;
            if (use_m) mp[idx] = m1;
            if (use_v) vp[idx] = v1;
        }
        return;
    }
    // ---- a 32 x 32 tile of dW_l^T [in x out] = H_{l-1}^T [in x B] * dZ_l [B x out]
    const int b = (wg & 7) * Q + (wg >> 3);
    if (b >= ntile) return;
    int i = 0, gi = 0;
#pragma unroll
    for (int j = 1; j < EH_GEMM_GROUP; ++j)
        if (j < gn && b >= gt[j]) { i = j; gi = gt[j]; }
    const EhGemmArgs g = G.g[i];
    const int t = b - gi, gx = (g.N + 31) >> 5;
    const int bx = t % gx, by = t / gx, m0 = by * 32, n0 = bx * 32;
    const int l32 = lane & 31, lh = lane >> 5;
    // this thread's four parameters: column tid & 31, rows (tid >> 5) + 8 j
    const int pc = n0 + (tid & 31), pr0 = m0 + (tid >> 5);
    const bool pcol = pc < g.N;
    long long ix[4];
    float th[4], mm[4], vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = min(pr0 + 8 * j, g.M - 1);
        ix[j] = (g.C + (long long)m * g.ldc + min(pc, g.N - 1)) - ap.slab;
        th[j] = tp[ix[j]];
        mm[j] = use_m ? mp[ix[j]] : 0.0f;
        vv[j] = use_v ? vp[ix[j]] : 0.0f;
    }
    const bool do_cs = g.colsum != nullptr && by == 0;
    long long cix = 0; float cth = 0.0f, cm = 0.0f, cv = 0.0f;
    if (do_cs && tid < 32 && pcol) {
        cix = (g.colsum + pc) - ap.slab;
        cth = tp[cix]; cm = use_m ? mp[cix] : 0.0f; cv = use_v ? vp[cix] : 0.0f;
    }
    const int kper = (((g.K + 3) >> 2) + 1) & ~1;                          // samples per wave (even, <= 16 for K <= 64)
    const int k0 = wave * kper, k1 = min(g.K, k0 + kper);
    const int am = min(m0 + l32, g.M - 1), bn = min(n0 + l32, g.N - 1);     // (rows / columns beyond the matrix: clamped, their outputs are never stored)
    float av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = k0 + 2 * s + lh;
        const bool ok = k < k1;
        av[s] = ok ? g.A[(long long)k * g.lda + am] : 0.0f;
        bv[s] = ok ? g.B[(long long)k * g.ldb + bn] : 0.0f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float csum = 0.0f;
    EH_A64(13);
#ifdef EH_STAMPS
    asm volatile("" :: "v"(tv));
    EH_A64(7);
    asm volatile("" :: "v"(th[0]), "v"(vv[0]));
    EH_A64(8);
    asm volatile("" :: "v"(th[3]), "v"(vv[3]));
    EH_A64(9);
    asm volatile("" :: "v"(av[0]), "v"(bv[0]));
    EH_A64(10);
    asm volatile("" :: "v"(av[7]), "v"(bv[7]));
    EH_A64(11);
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0); csum += bv[s]; }
    EH_A64(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][l32] = acc[r];
    EH_A64(2);
    if (do_cs) csr[2 * wave + lh][l32] = csum;
    if (tid < EH_LMECH_PART) totl[tid] = tv;
    __syncthreads();
    EH_A64(3);
    float scale; bool go;
    eh_lapply64_norm(ap, totl, scale, go);
    if (!go) return;
    float gg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = (tid >> 5) + 8 * j, cc = tid & 31;
        gg[j] = ((red[0][rr][cc] + red[1][rr][cc]) + (red[2][rr][cc] + red[3][rr][cc])) * scale;
    }
    EH_A64(4);
    eh_opt_update_all<4>(ap.o, gg, bt1, bt2, th, mm, vv);
    EH_A64(5);
    if (do_cs && tid < 32 && pcol) {
        float s = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) s += csr[p][tid];
        eh_opt_update(ap.o, s * scale, bt1, bt2, cth, cm, cv);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (pcol && pr0 + 8 * j < g.M) {
            tp[ix[j]] = th[j];
            if (use_m) mp[ix[j]] = mm[j];
            if (use_v) vp[ix[j]] = vv[j];
        }
    if (do_cs && tid < 32 && pcol) {
        tp[cix] = cth;
        if (use_m) mp[cix] = cm;
        if (use_v) vp[cix] = cv;
    }
    EH_A64(6);
#ifdef EH_STAMPS
    if (ap.stamps && tid == 0) atomicMax(ap.stamps + 30, (unsigned long long)wall_clock64());
#endif
}

// ---- the chain kernel for a suffix of one or two hidden layers + the output layer whose weights fit the REGISTERS of its 512 threads ---------
// Stamps of eh_lform_tailchain_kernel on the tutorial net (256 -> 128 -> 64 -> 1, one row per workgroup; tools/stamps_lform.py): 39 k cycles, of
// which the two delta products took 7.6 k and 4.2 k -- 4.2 k of the first just to ISSUE the sixteen 16-byte loads of the next product's
// weights: a workgroup streams the suffix's 164 KB twice (forward, delta products) through its CU's 64-byte-per-clock port, and a thread's k
// slice of more than eight rows fetched its second batch inside the phase.  Here a thread asks for ALL its fragments of both hidden layers
// (<= 16 + 8 pieces of 16 bytes), the output layer's columns in both layouts, the row and the records in its first instructions and keeps
// the fragments: the delta product of a layer is taken from the SAME registers -- thread (k slice, column group) has W[k][n0 .. n0 + 3] for
// its rows k, so it contributes dz[n0 .. n0 + 3] . W[k][n0 .. n0 + 3] to dZ_below[k]; the contributions of the column groups are written to
// LDS [k][group] and added in group order by the thread of row k.  No global load after the prologue, so the stores of H_l / dZ_l (what the
// weight-gradient launch needs) go out as the values are produced instead of waiting for the end.  Every sum in a fixed order.
inline bool eh_ltail_keep_ok(const EhLTailArgs& t, int count, int* tr_floats) {
    if (count > 256 || t.nl < 2 || t.nl > 3) return false;
    int tr = 0;
    for (int j = 0; j + 1 < t.nl; ++j) {
        const EhLTailLayer& L = t.L[j];
        if (!L.vec || L.out < 16 || L.kper > (j == 0 ? 16 : 8) || L.in > (int)EH_LTAIL_MAXW) return false;
        tr = std::max(tr, L.in * ((1 << L.lg_ng) + 4));
    }
    if (t.L[t.nl - 1].out > 16 || t.L[t.nl - 1].in > (int)EH_LTAIL_MAXW) return false;
    *tr_floats = tr;
    return true;
}
template <bool PROG, bool LPROG>
__global__ __launch_bounds__(EH_LTAIL_THREADS, 1) void eh_lform_tailkeep_kernel(const EhNet net, const EhStepArgs a, const EhLTailArgs t, const float* meta_g) {
    extern __shared__ __attribute__((aligned(16))) float eh_lt_smem[];
    constexpr int R = 1, NTH = EH_LTAIL_THREADS, NWV = NTH / 64, SR = 64, F0 = 16, F1 = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = t.wmax, WP = W + EH_LTAIL_PAD, nl = t.nl;
    float* const hb = eh_lt_smem;                                        // (the layout of eh_lform_tailchain_kernel with R = 1, the transposing area behind it)
    float* const zb = hb + (size_t)(nl + 1) * WP;
    float* const dd = zb + (t.any_swish ? (size_t)(nl + 1) * WP : 0);
    float* const red = dd + (size_t)(nl + 1) * WP;
    float* const red2 = red + EH_LTAIL_RED;
    float* const OS = red2 + EH_LTAIL_RED2;
    float* const SG = OS + 16 * SR;
    float* const RS = SG + 16 * SR;
    float* const metas = RS + (EH_MAX_FORC + EH_MAX_TARG) * SR;
    float* const bb = metas + ((EH_IMG_META + 3) & ~3);
    float* const tr = bb + (((size_t)nl * W + 3) & ~(size_t)3);          // [in][ng + 4] contributions of the column groups to a delta
    const int count = (int)a.count, row = (int)blockIdx.x;
    EH_STAMP(0);
    eh_kernarg_warm_big<(int)(sizeof(EhNet) + sizeof(EhStepArgs) + sizeof(EhLTailArgs) + 8)>();
    EH_STAMP(1);
    const bool two = nl == 3;
    const EhLTailLayer L0 = t.L[0], L1 = t.L[two ? 1 : 0], T = t.L[nl - 1];
    // ---- everything this workgroup will read from global memory, requested at once ---------------------------------------------------
    const int g0 = tid & ((1 << L0.lg_ng) - 1), ks0 = tid >> L0.lg_ng, n00 = 4 * g0, k00 = ks0 * L0.kper;
    const bool on0 = n00 < L0.out && ks0 < L0.KS;
    const int g1 = tid & ((1 << L1.lg_ng) - 1), ks1 = tid >> L1.lg_ng, n01 = 4 * g1, k01 = ks1 * L1.kper;
    const bool on1 = two && n01 < L1.out && ks1 < L1.KS;
    // order of the requests = order of their use (a wave's loads come back in order, and a wait for a late one is a wait for all before it):
    // the sample's index, the row, the small tables -- what the first barrier needs -- then the fragments layer by layer
    const bool rok = row < count;
    long long n_glb = 0;
    if (wave == 0) n_glb = rok ? (a.idx ? (long long)a.idx[a.first + row] : a.first + row) : 0;
    const float xin = (tid < L0.in && rok) ? t.Hin[(long long)row * t.ldin + tid] : 0.0f;
    const float zin = (tid < L0.in && rok && t.Zin) ? t.Zin[(long long)row * L0.in + tid] : 0.0f;
    float mt[(EH_IMG_META + 63) / 64], bv[EH_LTAIL_MAXL][EH_LTAIL_MAXW / 64];
    if (wave == 1) {
#pragma unroll
        for (int j = 0; j < (EH_IMG_META + 63) / 64; ++j) { const int e = lane + 64 * j; mt[j] = e < EH_IMG_META ? meta_g[e] : 0.0f; }
    }
#pragma unroll
    for (int i = 0; i < (EH_LTAIL_MAXL + NWV - 1) / NWV; ++i) {
        const int l = wave + NWV * i;
#pragma unroll
        for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) { const int n = lane + 64 * j; bv[i][j] = (l < nl && n < t.L[min(l, nl - 1)].out) ? t.L[min(l, nl - 1)].b[n] : 0.0f; }
    }
    f32x4 w0[F0], w1[F1];
#pragma unroll
    for (int u = 0; u < F0; ++u) w0[u] = (on0 && u < L0.kper) ? *(const f32x4*)(L0.W + (long long)min(k00 + u, L0.in - 1) * L0.out + n00) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // (wave 0: the sample's record, kept in registers until the mechanistic stage -- its address is the first request's answer)
    float rf[EH_MAX_FORC], rt[EH_MAX_TARG];
    if (wave == 0) {
        const bool live = lane < R && rok;
        const float* const rec = a.recs + n_glb * a.C;
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) rf[f] = (f < net.F && live) ? rec[net.P + f] : 0.0f;
#pragma unroll
        for (int tt = 0; tt < EH_MAX_TARG; ++tt) rt[tt] = (tt < net.T && live) ? rec[net.P + net.F + tt] : __builtin_nanf("");
    }
#pragma unroll
    for (int u = 0; u < F1; ++u) w1[u] = (on1 && u < L1.kper) ? *(const f32x4*)(L1.W + (long long)min(k01 + u, L1.in - 1) * L1.out + n01) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float wof[2][EH_LTAIL_MAXW / 64];                                    // output layer, forward: output n = wave + 8 i, lanes over k
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) { const int n = wave + NWV * i, k = lane + 64 * j; wof[i][j] = (n < T.out && k < T.in) ? T.W[(long long)k * T.out + n] : 0.0f; }
    float wvo[16];                                                       // ... and for the delta across it: row k = tid, its (<= 16) columns
#pragma unroll
    for (int n = 0; n < 16; ++n) wvo[n] = (n < T.out && tid < T.in) ? T.W[(long long)tid * T.out + n] : 0.0f;
    for (int e = tid; e < (nl + 1) * WP * (t.any_swish ? 3 : 2); e += NTH) hb[e] = 0.0f;
    if (wave == 1) {
#pragma unroll
        for (int j = 0; j < (EH_IMG_META + 63) / 64; ++j) { const int e = lane + 64 * j; if (e < EH_IMG_META) metas[e] = mt[j]; }
    }
#pragma unroll
    for (int i = 0; i < (EH_LTAIL_MAXL + NWV - 1) / NWV; ++i) {
        const int l = wave + NWV * i;
#pragma unroll
        for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) { const int n = lane + 64 * j; if (l < nl && n < t.L[min(l, nl - 1)].out) bb[l * W + n] = bv[i][j]; }
    }
    eh_lds_barrier();
    if (tid < L0.in) { hb[tid] = xin; if (t.Zin) zb[tid] = zin; }
    eh_lds_barrier();
    EH_STAMP(2);
    // ---- forward through the hidden layers of the suffix (the sums of eh_lform_tailchain_kernel in its order) -----------------------------
    auto forward = [&](const int l, const EhLTailLayer& L, const f32x4* w, const int NF, const int ks, const int n0, const int k0, const bool on) {
        const int out = L.out, KS = L.KS, lg_ostr = L.lg_ostr, ostr = 1 << lg_ostr;
        const float* const hin = hb + (size_t)l * WP;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (on) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < NF && u < L.kper) {
                    const float hv = hin[k0 + u];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(hv, w[u][c], acc[c]);
                }
            *(f32x4*)&red[ks * ostr + n0] = acc;
        }
        if (l == EH_STAMP_L) EH_STAMP(12);
        eh_lds_barrier();
        if (l == EH_STAMP_L) EH_STAMP(13);
        const float* src = red;
        int nsrc = KS;
        if (KS > 8) {
            const int KS1 = (KS + 7) >> 3;
            for (int e = tid; e < (KS1 << lg_ostr); e += NTH) {
                const int j = e >> lg_ostr, n = e & (ostr - 1);
                const float* const pp = red + ((8 * j) << lg_ostr) + n;
                float pv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = pp[min(i, KS - 1 - 8 * j) << lg_ostr];
                float v = pv[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) v += 8 * j + i < KS ? pv[i] : 0.0f;
                red2[e] = v;
            }
            eh_lds_barrier();
            src = red2; nsrc = KS1;
        }
        if (l == EH_STAMP_L) EH_STAMP(14);
        float* const hout = hb + (size_t)(l + 1) * WP;
        float* const zout = zb + (size_t)(l + 1) * WP;
        for (int n = tid; n < out; n += NTH) {
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = src[(min(i, nsrc - 1) << lg_ostr) + n];
            float v = pv[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) v += i < nsrc ? pv[i] : 0.0f;
            v += bb[l * W + n];
            const float hv = eh_act_rt(L.act, v);
            hout[n] = hv;
            if (L.act == EH_ACT_SWISH) zout[n] = v;
            if (rok) {                                                    // (what the weight-gradient launch reads)
                L.H[(long long)row * out + n] = hv;
                if (L.Z) L.Z[(long long)row * out + n] = v;
            }
        }
        if (l == EH_STAMP_L) EH_STAMP(6);
        eh_lds_barrier();
    };
    forward(0, L0, w0, F0, ks0, n00, k00, on0);
    EH_STAMP(3);
    if (two) forward(1, L1, w1, F1, ks1, n01, k01, on1);
    EH_STAMP(4);
    // ---- the output layer (<= 16 outputs): an output per wave, lanes over k -----------------------------------------------------------------
    {
        const float* const hin = hb + (size_t)(nl - 1) * WP;
        float* const oraw = hb + (size_t)nl * WP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = wave + NWV * i;
            if (n < T.out) {
                float sacc = 0.0f;
#pragma unroll
                for (int j = 0; j < EH_LTAIL_MAXW / 64; ++j) sacc = fmaf(hin[lane + 64 * j], wof[i][j], sacc);
                const float sum = eh_wave_sum(sacc);
                if (lane == 0) oraw[n] = sum + bb[(nl - 1) * W + n];
            }
        }
    }
    eh_lds_barrier();
    EH_STAMP(5);
    // ---- mechanistic model + masked loss + its pullback: wave 0, lane 0 = this workgroup's row (eh_lform_tailchain_kernel) ------------------
    float* const dcur0 = dd + (size_t)nl * WP;
    float* const psum = red;
    if (wave == 0) {
        auto pkind = [&](int j) { return (int)((net.par_kind >> (2 * j)) & 3u); };
        auto pidx = [&](int j) { return (int)((net.par_idx >> (4 * j)) & 15u); };
        const float* const oraw = hb + (size_t)nl * WP;
        const int K = T.out;
        const bool live = lane < R && rok;
#pragma unroll
        for (int f = 0; f < EH_MAX_FORC; ++f) RS[f * SR + lane] = rf[f];
#pragma unroll
        for (int tt = 0; tt < EH_MAX_TARG; ++tt) RS[(EH_MAX_FORC + tt) * SR + lane] = rt[tt];
        for (int k = 0; k < 16; ++k) { OS[k * SR + lane] = 0.0f; SG[k * SR + lane] = 1.0f; }
        for (int k = 0; k < K; ++k) {
            const float ov = live ? oraw[lane * WP + k] : 0.0f;
            float pv = ov, sv = 1.0f;
            if (net.scale_nn) {
                float lo = 0.0f, sc = 0.0f;
#pragma unroll
                for (int j = 0; j < EH_MAX_PARAMS; ++j)
                    if (j < net.n_par && pkind(j) == EH_PAR_NEURAL && pidx(j) == k) { lo = metas[EH_IMG_LO + j]; sc = metas[EH_IMG_SC + j]; }
                const float sgm = eh_sigmoid(ov);
                pv = fmaf(sc, sgm, lo);
                sv = sc * sgm * (1.0f - sgm);
            }
            OS[k * SR + lane] = pv; SG[k * SR + lane] = sv;
        }
        EhMechAcc MA;
        MA.clear();
        eh_mech_stage_lane<true, PROG, LPROG>(net, a, lane, live, row + lane, SR, RS, OS, SG, metas, MA);
        if (lane < R) {
            for (int k = 0; k < K; ++k) {
                const float d = live ? OS[k * SR + lane] : 0.0f;
                dcur0[lane * WP + k] = d;
                if (rok) t.O[(long long)k * t.ldo + row] = d;
            }
        }
        float v[EH_LMECH_PART];
#pragma unroll
        for (int j = 0; j < EH_MAX_PARAMS; ++j) v[j] = (j < net.n_par && pkind(j) == EH_PAR_GLOBAL) ? MA.gacc[j] * metas[EH_IMG_DPHI + j] : 0.0f;
        v[8] = MA.lacc;
#pragma unroll
        for (int tt = 0; tt < EH_MAX_TARG; ++tt) v[9 + tt] = MA.cacc[tt];
        v[13] = MA.syacc; v[14] = MA.syyacc; v[15] = 0.0f;
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < EH_LMECH_PART; ++k) {
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 0));
            mine = lane == k ? s : mine;
        }
        if (lane < EH_LMECH_PART) { psum[lane] = mine; t.part[(long long)blockIdx.x * EH_LMECH_PART + lane] = mine; }
    }
    eh_lds_barrier();
    EH_STAMP(8);
    // ---- deltas back down ---------------------------------------------------------------------------------------------------------------
    {   // across the output layer: a thread per k
        const int actp = t.L[nl - 2].act;
        const float* const hp = (actp == EH_ACT_SWISH ? zb : hb) + (size_t)(nl - 1) * WP;
        float* const dn = dd + (size_t)(nl - 1) * WP;
        if (tid < T.in) {
            float sacc = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; ++n) sacc = n < T.out ? fmaf(dcur0[n], wvo[n], sacc) : sacc;
            const float d = sacc * eh_dact_rt(actp, hp[tid]);
            dn[tid] = d;
            if (rok) t.L[nl - 2].D[(long long)row * T.in + tid] = d;
        }
        eh_lds_barrier();
    }
    EH_STAMP(9);
    auto backward = [&](const int l, const EhLTailLayer& L, const f32x4* w, const int NF, const int g, const int n0, const int k0, const bool on, float* const dst) {
        const int in = L.in, ngp = (1 << L.lg_ng) + 4, nga = L.out >> 2;
        const int actp = l > 0 ? t.L[l - 1].act : t.act_below;
        const float* const dz = dd + (size_t)(l + 1) * WP;
        float* const dn = dd + (size_t)l * WP;
        const float* const hp = (actp == EH_ACT_SWISH ? zb : hb) + (size_t)l * WP;
        if (on) {
            const f32x4 d4 = *(const f32x4*)&dz[n0];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < NF && u < L.kper && k0 + u < in)
                    tr[(k0 + u) * ngp + g] = fmaf(d4[3], w[u][3], fmaf(d4[2], w[u][2], fmaf(d4[1], w[u][1], d4[0] * w[u][0])));
        }
        eh_lds_barrier();
        if (tid < in) {
            const float* const q = tr + tid * ngp;
            float s = 0.0f;
            for (int j = 0; 4 * j < nga; ++j) {
                const f32x4 v = *(const f32x4*)&q[4 * j];
#pragma unroll
                for (int c = 0; c < 4; ++c) s += 4 * j + c < nga ? v[c] : 0.0f;
            }
            const float d = s * eh_dact_rt(actp, hp[tid]);
            dn[tid] = d;
            if (rok && dst) dst[(long long)row * in + tid] = d;
        }
        eh_lds_barrier();
    };
    if (two) backward(1, L1, w1, F1, g1, n01, k01, on1, L0.D);
    EH_STAMP(10);
    if (t.Dbelow) backward(0, L0, w0, F0, g0, n00, k00, on0, t.Dbelow);
    EH_STAMP(15);
}
