// How fast can a freshly launched kernel pull N bytes that a PREVIOUS kernel wrote (cold in every L2: they come from the memory side)?
// The same total (5.6 MB: the tutorial net's parameters + second moments; 16.8 MB) split over 256 ... 2048 workgroups of 256 threads, every
// thread's loads independent and issued at once (U = total / (workgroups x 1 KB) 4-byte loads, 256 B per wave and load), then summed and
// one store.  Between two reads a writer kernel rewrites the buffer (what the previous step's optimiser launch does).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/coldread.hip -o /tmp/coldread && /tmp/coldread
// prints us per (write + read) pair, per write alone, and the difference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void writer(float* p, long long n, float v) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) p[i] = v;
}
template <int U>
__global__ __launch_bounds__(256) void reader(const float* __restrict__ p, float* out) {
    const long long base = (long long)blockIdx.x * U * 256 + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[base + (long long)u * 256];
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
    if (s == 12345.678f) out[blockIdx.x] = s;
}
// the tile pattern of the weight-gradient launch: 32 rows x 128 B at a row stride of ld floats
template <int U>
__global__ __launch_bounds__(256) void reader_tile(const float* __restrict__ p, float* out, int ld, int gx) {
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx, tid = threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(long long)(by * 8 * U + (tid >> 5) + 8 * u) * ld + bx * 32 + (tid & 31)];
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
    if (s == 12345.678f) out[blockIdx.x] = s;
}
template <class F> static float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}
int main() {
    const long long nmax = 16ll << 20;       // floats
    float *p, *out; hipMalloc(&p, nmax * 4); hipMalloc(&out, 1 << 20); hipMemset(p, 0, nmax * 4);
    for (long long total : {1400000ll, 4200000ll}) {       // floats: 5.6 MB, 16.8 MB
        const float w = timeit([&] { hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, p, total, 1.0f); }, 300);
        printf("total %.1f MB: writer alone %.2f us\n", total * 4 / 1e6, w);
#define RUN(U) { const int wg = (int)(total / (256ll * U)); const float t = timeit([&] { hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, p, total, 1.0f); hipLaunchKernelGGL((reader<U>), dim3(wg), dim3(256), 0, 0, p, out); }, 300); \
                 const float t2 = timeit([&] { hipLaunchKernelGGL((reader<U>), dim3(wg), dim3(256), 0, 0, p, out); }, 300); \
                 printf("   %5d workgroups x %2d loads per thread: cold read %.2f us (pair %.2f), re-read without a writer in between %.2f us\n", wg, U, t - w, t, t2); }
        RUN(1) RUN(2) RUN(4) RUN(8) RUN(16) RUN(32)
    }
    {   // tile pattern over a 1024 x 512 matrix (2 MB) x 2 arrays worth: rows of 128 B
        const int ld = 512, gx = 16; const long long total = 1024ll * 512;
        const float w = timeit([&] { hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, p, total, 1.0f); }, 300);
        const float t = timeit([&] { hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, p, total, 1.0f); hipLaunchKernelGGL((reader_tile<4>), dim3(512), dim3(256), 0, 0, p, out, ld, gx); }, 300);
        const float tl = timeit([&] { hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, p, total, 1.0f); hipLaunchKernelGGL((reader<4>), dim3(512), dim3(256), 0, 0, p, out); }, 300);
        printf("2 MB matrix, 512 workgroups: 32 x 32 tiles cold %.2f us, linear cold %.2f us (writer %.2f)\n", t - w, tl - w, w);
    }
    return 0;
}
