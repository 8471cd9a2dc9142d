// Latencies the cross-GPU exchange of the fused step is made of ("why the loop-back cost is 2.3 us"), each measured as a
// chain of DEPENDENT operations issued by one lane of one wave on an otherwise idle GPU:
//   returning device-scope atomic (what elects the last workgroup of a launch: two of them in the two-level ticket),
//   non-returning float atomic + the wait for its acknowledgement (what a workgroup's partial sums cost before it may take a ticket),
//   plain load that misses every cache (sc1: the re-read of sums other XCDs produced), 8-byte store + dependent uncached re-load of the
//   same word (one leg of a publish -> poll hand-over on one GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic_ret(unsigned* p, int n, unsigned* out) {
    unsigned v = 0;
    for (int i = 0; i < n; ++i) v = atomicAdd(p + (v & 1u), 1u);          // the next address depends on the value returned
    *out = v;
}
__global__ void k_atomic_ack(float* p, int n) {
    for (int i = 0; i < n; ++i) { atomicAdd(p, 1.0f); __builtin_amdgcn_s_waitcnt(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}
__global__ void k_load_sc1(const unsigned* p, int n, unsigned* out) {
    unsigned v = 0;
    for (int i = 0; i < n; ++i) v = __builtin_nontemporal_load(p + (v & 1023u) * 64u) + (unsigned)i;   // a chain through a 256 KB table of zeros
    *out = v;
}
__global__ void k_store_load(unsigned long long* p, int n, unsigned* out) {
    unsigned long long v = 0;
    for (int i = 0; i < n; ++i) {
        __hip_atomic_store(p, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        do { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (v != (unsigned long long)(i + 1));
    }
    *out = (unsigned)v;
}
int main() {
    unsigned *a, *out, *tab; float* f; unsigned long long* w;
    hipMalloc(&a, 256); hipMalloc(&out, 64); hipMalloc(&f, 256); hipMalloc(&tab, 1024 * 64 * 4); hipMalloc(&w, 256);
    hipMemset(a, 0, 256); hipMemset(f, 0, 256); hipMemset(tab, 0, 1024 * 64 * 4); hipMemset(w, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipMemset(w, 0, 256);
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-66s %7.0f ns per kernel / %d\n", name, 1e6 * ms / N, N);
    };
    timeit("returning device-scope atomic", [&] { hipLaunchKernelGGL(k_atomic_ret, dim3(1), dim3(1), 0, 0, a, N, out); });
    timeit("returning device-scope atomic, 8 workgroups (8 XCDs), 1 address", [&] { hipLaunchKernelGGL(k_atomic_ret, dim3(8), dim3(1), 0, 0, a, N, out); });
    timeit("returning device-scope atomic, 256 workgroups, 1 address", [&] { hipLaunchKernelGGL(k_atomic_ret, dim3(256), dim3(1), 0, 0, a, N / 10, out); });
    timeit("float atomic add + wait for its acknowledgement", [&] { hipLaunchKernelGGL(k_atomic_ack, dim3(1), dim3(1), 0, 0, f, N); });
    timeit("dependent load past the caches (nontemporal)", [&] { hipLaunchKernelGGL(k_load_sc1, dim3(1), dim3(1), 0, 0, tab, N, out); });
    timeit("system-scope 8-byte store + re-load until it shows", [&] { hipLaunchKernelGGL(k_store_load, dim3(1), dim3(1), 0, 0, w, N, out); });
    return 0;
}
