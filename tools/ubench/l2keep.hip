// Does an XCD's L2 keep what ITS workgroups wrote (or read) in an earlier launch?  A read-modify-write kernel over 5.6 MB, 683 workgroups x 8
// floats per thread, launched back to back: (a) the same workgroup -> address mapping every launch (workgroup i lands on XCD i % 8 every
// time), (b) the mapping shifted by one workgroup per launch (every line is touched by a different XCD than last time), (c) as (a) with an
// unrelated kernel (reads 8 MB elsewhere) between two launches; and the read-only counterparts.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2keep.hip -o tools/ubench/l2keep
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int U = 8;
template <bool WRITE>
__global__ __launch_bounds__(256) void rmw(float* p, int shift, int nwg, float* out) {
    const int b = (blockIdx.x + shift) % nwg;
    float* q = p + (long long)b * U * 256 + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = q[u * 256];
    if (WRITE) {
#pragma unroll
        for (int u = 0; u < U; ++u) q[u * 256] = v[u] * 1.0001f + 1.0f;
    } else {
        float s = 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
        if (s == 12345.678f) out[blockIdx.x] = s;
    }
}
__global__ __launch_bounds__(256) void other(const float* p, float* out) {
    const float* q = p + (long long)blockIdx.x * U * 256 + threadIdx.x;
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += q[u * 256];
    if (s == 12345.678f) out[blockIdx.x] = s;
}
template <class F> static float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f(i);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}
int main() {
    const int nwg = 683;
    float *p, *o, *out; hipMalloc(&p, 64 << 20); hipMalloc(&o, 64 << 20); hipMalloc(&out, 1 << 20); hipMemset(p, 0, 64 << 20); hipMemset(o, 0, 64 << 20);
    const float tother = timeit([&](int) { hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, o, out); }, 400);
    printf("unrelated kernel alone (8 MB, hot): %.2f us\n", tother);
    printf("read-modify-write 5.6 MB, same mapping:        %.2f us\n", timeit([&](int) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); }, 400));
    printf("read-modify-write 5.6 MB, shifted by 1 / launch: %.2f us\n", timeit([&](int i) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, i % nwg, nwg, out); }, 400));
    printf("read-modify-write 5.6 MB, shifted by 8 / launch: %.2f us (same XCD, other CU)\n", timeit([&](int i) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, (8 * i) % 680, 680, out); }, 400));
    printf("same mapping + the unrelated kernel in between:  %.2f us (minus %.2f)\n", timeit([&](int) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); hipLaunchKernelGGL(other, dim3(1024), dim3(256), 0, 0, o, out); }, 400), tother);
    printf("read only 5.6 MB, same mapping:                 %.2f us\n", timeit([&](int) { hipLaunchKernelGGL((rmw<false>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); }, 400));
    printf("read only 5.6 MB, shifted by 1 / launch:         %.2f us\n", timeit([&](int i) { hipLaunchKernelGGL((rmw<false>), dim3(nwg), dim3(256), 0, 0, p, i % nwg, nwg, out); }, 400));
    // a reader on another mapping after the writer (what the forward / delta products do with the weights), then the writer again
    printf("write (same mapping) + read-only shifted reader:  %.2f us per pair\n", timeit([&](int i) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); hipLaunchKernelGGL((rmw<false>), dim3(nwg), dim3(256), 0, 0, p, 3, nwg, out); }, 400));
    printf("write (same mapping) + read-only same-mapping reader: %.2f us per pair\n", timeit([&](int i) { hipLaunchKernelGGL((rmw<true>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); hipLaunchKernelGGL((rmw<false>), dim3(nwg), dim3(256), 0, 0, p, 0, nwg, out); }, 400));
    return 0;
}
