// Throughput of float atomic adds into a few per-XCD accumulator rows, against plain stores of per-workgroup partial rows
// (the two ways a step kernel can hand its gradient sums to the optimiser kernel).  n = floats per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic(float* acc, int n, int shards) {
    float* row = acc + (size_t)(blockIdx.x % shards) * n;
    for (int e = threadIdx.x; e < n; e += blockDim.x) atomicAdd(&row[e], 1.0f + e);
}
__global__ void k_store(float* slab, int n) {
    float* row = slab + (size_t)blockIdx.x * n;
    for (int e = threadIdx.x; e < n; e += blockDim.x) row[e] = 1.0f + e;
}
__global__ void k_reduce(const float* slab, float* out, int n, int rows) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0;
    for (int r = 0; r < rows; ++r) s += slab[(size_t)r * n + e];
    out[e] = s;
}
int main() {
    const int WG = 256;
    for (int n : {5004, 21510}) {
        float *acc, *slab, *out;
        hipMalloc(&acc, sizeof(float) * 64 * n); hipMalloc(&slab, sizeof(float) * WG * n); hipMalloc(&out, sizeof(float) * n);
        hipMemset(acc, 0, sizeof(float) * 64 * n);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("n=%d %-34s %.2f us\n", n, name, ms * 1000 / 50);
        };
        for (int sh : {8, 16, 32, 64}) {
            char nm[64]; snprintf(nm, 64, "atomics into %d rows (512 thr)", sh);
            timeit(nm, [&] { hipLaunchKernelGGL(k_atomic, dim3(WG), dim3(512), 0, 0, acc, n, sh); });
        }
        timeit("atomics into 8 rows (256 thr)", [&] { hipLaunchKernelGGL(k_atomic, dim3(WG), dim3(256), 0, 0, acc, n, 8); });
        timeit("stores of 256 rows", [&] { hipLaunchKernelGGL(k_store, dim3(WG), dim3(512), 0, 0, slab, n); });
        timeit("reduce 256 rows", [&] { hipLaunchKernelGGL(k_reduce, dim3((n + 63) / 64), dim3(64), 0, 0, slab, out, n, 256); });
        timeit("reduce 8 rows", [&] { hipLaunchKernelGGL(k_reduce, dim3((n + 63) / 64), dim3(64), 0, 0, slab, out, n, 8); });
        timeit("empty-ish (n=64 stores)", [&] { hipLaunchKernelGGL(k_store, dim3(WG), dim3(512), 0, 0, slab, 64); });
        hipFree(acc); hipFree(slab); hipFree(out);
    }
    return 0;
}
