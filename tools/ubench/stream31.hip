// What the memory system gives a 3-reads : 1-write streaming mix (the access pattern of eh_mech_vjp_kernel<4, rbq10>: o, ta, y in,
// d_o out; 16 B per lane and array, 1 GiB per launch -- four times the Infinity Cache), against a 1 : 1 copy, a read-only and a
// write-only sweep of the same total, for several grid shapes.  Establishes the roof the stand-alone mechanistic stage is priced
// against.   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/stream31.hip -o /tmp/stream31 && /tmp/stream31
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 3 reads 1 write; 1: copy (1 : 1); 2: 4 reads (sum kept alive); 3: 4 writes.  CHUNK: every workgroup owns one contiguous run
template <int MODE, bool CHUNK, bool NT>
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ d,
                                                 long long n, float* sink) {
    const long long per = CHUNK ? (n / 4 + gridDim.x - 1) / gridDim.x : 0;          // float4s per workgroup (CHUNK)
    long long i = CHUNK ? (long long)blockIdx.x * per + threadIdx.x : (long long)blockIdx.x * 256 + threadIdx.x;
    const long long end = CHUNK ? ((long long)(blockIdx.x + 1) * per < n / 4 ? (long long)(blockIdx.x + 1) * per : n / 4) : n / 4;
    const long long step = CHUNK ? 256 : (long long)gridDim.x * 256;
    f32x4 acc = {0, 0, 0, 0};
    auto ld = [&](const float* p, long long k) { return NT ? __builtin_nontemporal_load((const f32x4*)p + k) : ((const f32x4*)p)[k]; };
    auto st = [&](float* p, long long k, f32x4 v) { if (NT) __builtin_nontemporal_store(v, (f32x4*)p + k); else ((f32x4*)p)[k] = v; };
    for (; i < end; i += step) {
        if (MODE == 0) { const f32x4 x = ld(a, i), y = ld(b, i), z = ld(c, i); st(d, i, x * y + z); }
        else if (MODE == 1) { const f32x4 x = ld(a, i), y = ld(b, i); st((float*)c, i, x); st(d, i, y); }
        else if (MODE == 2) { acc += ld(a, i) + ld(b, i) + ld(c, i) + ld(d, i); }
        else { const f32x4 v = {1.f, 2.f, 3.f, (float)i}; st((float*)a, i, v); st((float*)b, i, v); st((float*)c, i, v); st(d, i, v); }
    }
    if (MODE == 2 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

int main() {
    const long long n = 64ll << 20;                      // floats per array: 4 x 256 MiB = 1 GiB per launch
    float *a, *b, *c, *d, *sink;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4); hipMalloc(&sink, 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4); hipMemset(d, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, int grid, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / 20;
        printf("%-44s grid %6d  %8.1f us  %6.2f TB/s  %.3f of 8 TB/s\n", name, grid, us, 4.0 * n * 4 / us / 1e6, 4.0 * n * 4 / us / 1e6 / 8.0);
    };
#define RUN(M, CH, NTF, G, NAME) timeit(NAME, G, [&] { hipLaunchKernelGGL((k_stream<M, CH, NTF>), dim3(G), dim3(256), 0, 0, a, b, c, d, n, sink); })
    for (int g : {512, 1024, 2048, 4096, 8192, 16384, 65536}) RUN(0, false, false, g, "3 reads : 1 write, grid-stride");
    for (int g : {1024, 2048, 4096, 8192, 65536}) RUN(0, true, false, g, "3 reads : 1 write, contiguous run per workgroup");
    for (int g : {2048, 4096}) RUN(0, false, true, g, "3 reads : 1 write, grid-stride, non-temporal");
    for (int g : {2048, 4096}) RUN(0, true, true, g, "3 reads : 1 write, contiguous, non-temporal");
    for (int g : {2048, 4096, 65536}) RUN(1, false, false, g, "copy 2 reads : 2 writes, grid-stride");
    for (int g : {2048, 4096, 65536}) RUN(2, false, false, g, "4 reads, grid-stride");
    for (int g : {2048, 4096, 65536}) RUN(3, false, false, g, "4 writes, grid-stride");
    return 0;
}
