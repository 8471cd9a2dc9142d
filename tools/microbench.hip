// Launch-floor microbenchmark: what does the shortest possible kernel of our launch shapes cost?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int a[100]; };
__global__ void k_empty() {}
__global__ void k_empty_lds() { extern __shared__ float s[]; if (threadIdx.x == 12345) s[0] = 1; }
__global__ void k_big(Big b, float* o) { if (b.a[99] == 123456) o[0] = 1; }
__global__ void k_rw(const float* in, float* out) { out[blockIdx.x * 256 + threadIdx.x] = in[blockIdx.x * 256 + threadIdx.x] + 1.0f; }
__global__ void k_rw2(const float* in, float* out) {  // two dependent round trips
    float v = in[blockIdx.x * 256 + threadIdx.x]; int j = ((int)v) & 255; out[blockIdx.x * 256 + threadIdx.x] = in[j] + v; }
template <class F> void run(const char* name, F f) {
    for (int i = 0; i < 200; ++i) f();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) f();
    (void)hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
    printf("%-28s wall %.2f us/launch\n", name, us);
}
int main() {
    float *a, *b; (void)hipMalloc(&a, 1 << 20); (void)hipMalloc(&b, 1 << 20); (void)hipMemset(a, 0, 1 << 20);
    (void)hipFuncSetAttribute((const void*)k_empty_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    Big big{}; hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    run("empty<22,256>", [&] { hipLaunchKernelGGL(k_empty, dim3(22), dim3(256), 0, s); });
    run("empty<256,256>", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); });
    run("empty_lds96K<256,256>", [&] { hipLaunchKernelGGL(k_empty_lds, dim3(256), dim3(256), 96 * 1024, s); });
    run("bigarg<256,256>", [&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, b); });
    run("rw<22,256>", [&] { hipLaunchKernelGGL(k_rw, dim3(22), dim3(256), 0, s, a, b); });
    run("rw2<22,256>", [&] { hipLaunchKernelGGL(k_rw2, dim3(22), dim3(256), 0, s, a, b); });
    run("rw<256,256>", [&] { hipLaunchKernelGGL(k_rw, dim3(256), dim3(256), 0, s, a, b); });
    run("pair rw+rw", [&] { hipLaunchKernelGGL(k_rw, dim3(256), dim3(256), 0, s, a, b); hipLaunchKernelGGL(k_rw, dim3(22), dim3(256), 0, s, b, a); });
    return 0;
}
