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

enum { EH_GEPI_STORE = 0, EH_GEPI_BIAS_ACT = 1, EH_GEPI_BIAS_T = 2, EH_GEPI_DACT = 3 };

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
};

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
__device__ __forceinline__ void eh_gemm_tile(const EhGemmArgs& g, const int bx, const int by, const int bz) {
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
    const bool do_cs_v = EPI == EH_GEPI_STORE && g.colsum != nullptr && by == 0 && tid < BN;
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
    if (do_cs && n0 + tid < g.N) g.colsum[(long long)bz * g.c_zstride + n0 + tid] = cs;
    // C/D layout of the 32x32 MFMA: lane -> column (lane & 31); register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* const C = g.C + (long long)bz * g.c_zstride;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int n = n0 + wn * WT + 32 * j + l32;
            const float bv = ((EPI == EH_GEPI_BIAS_ACT || EPI == EH_GEPI_BIAS_T) && n < g.N) ? g.bias[n] : 0.0f;
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
__device__ __forceinline__ void eh_thin_gemm_tile(const EhThinArgs& a, const int bx, const int by) {
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
    if (q == 0 && live) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < a.J) a.C[(long long)z * a.c_z + (long long)col * a.c_col + (long long)j * a.c_j] = (red[0][cl][j] + red[1][cl][j]) + (red[2][cl][j] + red[3][cl][j]);
        if (a.cs_wide) a.cs_wide[(long long)z * a.cs_z + col] = (red[0][cl][8] + red[1][cl][8]) + (red[2][cl][8] + red[3][cl][8]);
    }
    if (a.cs_thin && bx == 0 && tid < a.J) a.cs_thin[(long long)z * a.cs_z + tid] = (redt[0][tid] + redt[1][tid]) + (redt[2][tid] + redt[3][tid]);
}
__global__ __launch_bounds__(256) void eh_thin_gemm_kernel(const EhThinArgs a) { eh_thin_gemm_tile(a, (int)blockIdx.x, (int)blockIdx.y); }
// (several of them in one launch, as eh_gemm_group_kernel)
struct EhThinGroup { EhThinArgs a[EH_GEMM_GROUP]; int t0[EH_GEMM_GROUP + 1]; int gx[EH_GEMM_GROUP]; int n; };
__global__ __launch_bounds__(256) void eh_thin_gemm_group_kernel(const EhThinGroup G) {
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.t0[i + 1]) ++i;
    const int t = (int)blockIdx.x - G.t0[i], gx = G.gx[i];
    const EhThinArgs a = G.a[i];
    eh_thin_gemm_tile(a, t % gx, t / gx);
}

// Both groups of a small-batch step's weight gradients -- the tiled ones and the thin ones -- as ONE launch: the first workgroups run
// the thin products, the others the tiles (one dependent launch fewer; the two bodies' LDS side by side, 44 KB).
__global__ __launch_bounds__(256) void eh_dw_group_kernel(const EhGemmGroup G, const EhThinGroup T) {
    const int nthin = T.t0[T.n];
    if ((int)blockIdx.x < nthin) {
        int i = 0;
        while (i + 1 < T.n && (int)blockIdx.x >= T.t0[i + 1]) ++i;
        const int t = (int)blockIdx.x - T.t0[i], gx = T.gx[i];
        const EhThinArgs a = T.a[i];
        eh_thin_gemm_tile(a, t % gx, t / gx);
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
                v += g.bias[n];
                if (g.Z) g.Z[(long long)m * g.ldc + n] = v;
                g.C[(long long)m * g.ldc + n] = eh_act_rt(g.act, v);
            } else {
                g.C[(long long)m * g.ldc + n] = v * eh_dact_rt(g.act, g.H[(long long)m * g.ldh + n]);
            }
        }
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
};
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
__global__ __launch_bounds__(256) void eh_lform_tail_kernel(const float* part, int nblk, const EhNet net, float* slab, int nrows, long long n_acc) {
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
