// Does the matrix pipe overlap with the vector ALU on gfx950?  256 workgroups of 1 or 2 waves per SIMD run (M) chains of
// independent MFMAs, (V) independent v_pk_fma_f32 chains, (MV) both, one MFMA : R VALU; whole-kernel times (HIP events), so
// that the oldest-wave-first arbitration between the waves of a SIMD does not bias the answer.
// build + run (on the GPU box): hipcc --offload-arch=gfx950 -O3 -w tools/ubench/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int R, int MODE, int VOP>      // KIND 0: fp32 16x16x4, 1: bf16 16x16x32 ; MODE 0: M, 1: V, 2: MV ; VOP 0: v_pk_fma_f32, 1: v_fma_f32, 2: v_exp_f32, 3: v_add_u32
__global__ void k(float* out, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    auto mf = [&](int q) {
        if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        else acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[q], 0, 0, 0);
    };
    auto va = [&](int q) {
        if constexpr (VOP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(m), "v"(c));
        else if constexpr (VOP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q][0]) : "v"(m[0]), "v"(c[0]));
        else if constexpr (VOP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q][0]));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[q][0]) : "v"(c[0]));
    };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (MODE != 1) mf(q);
            if constexpr (MODE != 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) va((q * R + r) & 7);
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int R, int MODE, int VOP>
float one(float* out, int nthr, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, R, MODE, VOP>), dim3(256), dim3(nthr), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, R, MODE, VOP>), dim3(256), dim3(nthr), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (4.0f * iters);       // ns per (MFMA [+ R VALU]) of one wave
}
template <int KIND, int R, int VOP = 0>
void run(const char* name, int nthr) {
    float* out; hipMalloc(&out, 4 * 1024 * 256);
    const int iters = 20000;
    const float tm = one<KIND, R, 0, VOP>(out, nthr, iters), tv = one<KIND, R, 1, VOP>(out, nthr, iters), tmv = one<KIND, R, 2, VOP>(out, nthr, iters);
    static const char* vn[4] = {"v_pk_fma_f32", "v_fma_f32", "v_exp_f32", "v_add_u32"};
    printf("%-12s 1 MFMA : %d %s, %d wave(s)/SIMD: M %.1f ns | V %.1f ns | MV %.1f ns   (sum %.1f, max %.1f) -> overlap %.0f %%\n", name, R, vn[VOP], nthr / 256, tm, tv, tmv, tm + tv,
           tm > tv ? tm : tv, 100.0f * (tm + tv - tmv) / (tm < tv ? tm : tv));
    hipFree(out);
}
int main() {
    run<0, 2>("f32 16x16x4", 256); run<0, 4>("f32 16x16x4", 256); run<0, 8>("f32 16x16x4", 256);
    run<0, 2>("f32 16x16x4", 512); run<0, 4>("f32 16x16x4", 512); run<0, 8>("f32 16x16x4", 512);
    run<1, 1>("bf16 16x16x32", 256); run<1, 2>("bf16 16x16x32", 256); run<1, 4>("bf16 16x16x32", 256);
    run<1, 1>("bf16 16x16x32", 512); run<1, 2>("bf16 16x16x32", 512); run<1, 4>("bf16 16x16x32", 512);
    run<1, 2>("bf16 16x16x32", 1024); run<1, 4>("bf16 16x16x32", 1024);
    run<0, 4, 1>("f32 16x16x4", 256); run<0, 4, 1>("f32 16x16x4", 512); run<0, 8, 1>("f32 16x16x4", 512);
    run<0, 2, 2>("f32 16x16x4", 256); run<0, 2, 2>("f32 16x16x4", 512);
    run<0, 4, 3>("f32 16x16x4", 256); run<0, 4, 3>("f32 16x16x4", 512); run<0, 8, 3>("f32 16x16x4", 512);
    run<1, 4, 1>("bf16 16x16x32", 256); run<1, 4, 1>("bf16 16x16x32", 512); run<1, 2, 2>("bf16 16x16x32", 512); run<1, 4, 3>("bf16 16x16x32", 512);
    return 0;
}
