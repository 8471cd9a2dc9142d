// Does the matrix pipe overlap with the vector ALU on gfx950?  One wave per SIMD (or two), three loops timed with the shader clock:
//   (M) chains of independent MFMAs, (V) independent v_pk_fma_f32 chains, (MV) both interleaved 1 MFMA : R VALU.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o gpurun_out/overlap && ./gpurun_out/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int R>      // KIND 0: fp32 16x16x4, 1: bf16 16x16x32
__global__ void k(float* out, long long* cyc, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    long long t[4];
    auto mf = [&](int q) {
        if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        else acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[q], 0, 0, 0);
    };
    auto va = [&](int q) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(m), "v"(c)); };
    t[0] = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { mf(q); }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
    t[1] = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < R; ++r) va((q * R + r) & 7);
        }
    }
    t[2] = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mf(q);
#pragma unroll
            for (int r = 0; r < R; ++r) va((q * R + r) & 7);
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
    t[3] = __builtin_readcyclecounter();
    float s = 0;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t[1] - t[0]; cyc[1] = t[2] - t[1]; cyc[2] = t[3] - t[2]; }
}

template <int KIND, int R>
void run(const char* name, int nthr) {
    float* out; long long* cyc; long long h[3];
    hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, R>), dim3(256), dim3(nthr), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    const double n = 4.0 * iters;
    printf("%-10s R=%d waves/SIMD=%d: MFMA %.1f cyc each | %d VALU %.1f cyc (%.2f each) | interleaved %.1f cyc  (sum %.1f, max %.1f)\n", name, R, nthr / 256,
           h[0] / n, R, h[1] / n, h[1] / n / R, h[2] / n, (h[0] + h[1]) / n, (h[0] > h[1] ? h[0] : h[1]) / n);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 2>("f32 16x16x4", 256); run<0, 4>("f32 16x16x4", 256); run<0, 8>("f32 16x16x4", 256); run<0, 4>("f32 16x16x4", 512); run<0, 8>("f32 16x16x4", 512);
    run<1, 1>("bf16 x32", 256); run<1, 2>("bf16 x32", 256); run<1, 4>("bf16 x32", 256); run<1, 2>("bf16 x32", 512); run<1, 4>("bf16 x32", 512);
    return 0;
}
