// Does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N) shorten a small kernel whose first memory
// accesses depend on pointers from the kernarg segment?  256 workgroups of 512 threads, every thread loads one value through each of
// four argument pointers and stores their sum; 2 000 back-to-back launches.  Build twice (with / without the option) and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(const float* a, const float* b, const float* c, const float* d, float* out, int n) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = (a[i] + b[i]) + (c[i] + d[i]);
}
int main() {
    const int n = 256 * 512;
    float *p[5];
    for (auto& q : p) { hipMalloc(&q, 4 * n); hipMemset(q, 0, 4 * n); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, p[0], p[1], p[2], p[3], p[4], n);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, p[0], p[1], p[2], p[3], p[4], n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%.3f us per launch\n", ms * 1000 / 2000);
    }
    return 0;
}
