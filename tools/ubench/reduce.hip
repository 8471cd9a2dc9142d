// Where does the slab-reduction + optimiser kernel's time go?  A stand-alone model of eh_reduce_kernel<true, 64> (256 rows of
// n_acc partial sums -> column sums, batch counts, Adam, scatter through the image map), with its parts switched off one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CW, int MODE>   // MODE bit 0: counts loop, bit 1: Adam + stores, bit 2: image scatter through imap, bit 3: imap prefetched with theta
__global__ __launch_bounds__(256) void red(const float* __restrict__ slab, int nblk, int n_acc, int n_theta, float* gradbuf, float* theta, float* m, float* v,
                                           const int* imap, float* image) {
    constexpr int NQ = 256 / CW;
    __shared__ float part[NQ][CW + 1];
    __shared__ float wsum[4][8];
    const int tid = threadIdx.x, p = tid % CW, q = tid / CW, idx = blockIdx.x * CW + p;
    float th = 0, mm = 0, vv = 0; int mp = 0;
    if ((MODE & 2) && q == 0 && idx < n_theta) { th = theta[idx]; mm = m[idx]; vv = v[idx]; if (MODE & 8) mp = imap[idx]; }
    float s = 0.0f;
    if (idx < n_acc) {
        const float* col = slab + (size_t)q * n_acc + idx;
        const size_t rstride = (size_t)NQ * n_acc;
        int r = q;
        for (; r + 31 * NQ < nblk; r += 32 * NQ) {
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = col[u * rstride];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += t[u];
            col += 32 * rstride;
        }
#pragma unroll 16
        for (; r < nblk; r += NQ) { s += *col; col += rstride; }
    }
    part[q][p] = s;
    float ntot = 1.0f;
    if (MODE & 1) {
        float cs[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            cs[t] = 0.0f;
            for (int r = tid; r < nblk; r += 256) cs[t] += slab[(size_t)r * n_acc + n_theta + t];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cs[t] += __shfl_xor(cs[t], off, 64);
            if ((tid & 63) == 0) wsum[tid >> 6][t] = cs[t];
        }
    }
    __syncthreads();
    if (MODE & 1) ntot = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
    if (q == 0 && idx < n_acc) {
        float tot = 0.0f;
#pragma unroll
        for (int k = 0; k < NQ; ++k) tot += part[k][p];
        float g = tot / ntot;
        gradbuf[idx] = g;
        if ((MODE & 2) && idx < n_theta) {
            mm = 0.9f * mm + 0.1f * g; vv = 0.999f * vv + 0.001f * g * g;
            th -= mm / (sqrtf(vv) + 1e-8f) * 0.01f;
            theta[idx] = th; m[idx] = mm; v[idx] = vv;
            if (MODE & 4) image[(MODE & 8) ? mp : imap[idx]] = th;
        }
    }
}
int main() {
    const int rows = 256;
    for (int n_theta : {4996, 21510 - 7}) {
        const int n_acc = n_theta + 7;
        float *slab, *gb, *th, *m, *v, *img; int* imap;
        hipMalloc(&slab, sizeof(float) * (size_t)rows * n_acc); hipMemset(slab, 0, sizeof(float) * (size_t)rows * n_acc);
        hipMalloc(&gb, 4 * n_acc); hipMalloc(&th, 4 * n_acc); hipMalloc(&m, 4 * n_acc); hipMalloc(&v, 4 * n_acc); hipMalloc(&img, 4 * 2 * n_acc); hipMalloc(&imap, 4 * n_acc);
        hipMemset(th, 0, 4 * n_acc); hipMemset(m, 0, 4 * n_acc); hipMemset(v, 0, 4 * n_acc);
        std::vector<int> im(n_acc); for (int i = 0; i < n_acc; ++i) im[i] = (i * 7) % (2 * n_acc);
        hipMemcpy(imap, im.data(), 4 * n_acc, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 100; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("n_acc=%d %-46s %.2f us per launch (back to back)\n", n_acc, name, ms * 10);
        };
#define GO(CW, MODE) hipLaunchKernelGGL((red<CW, MODE>), dim3((n_acc + CW - 1) / CW), dim3(256), 0, 0, slab, rows, n_acc, n_theta, gb, th, m, v, imap, img)
        timeit("column sums only", [&] { GO(64, 0); });
        timeit("+ counts loop", [&] { GO(64, 1); });
        timeit("+ Adam", [&] { GO(64, 3); });
        timeit("+ image scatter (as shipped)", [&] { GO(64, 7); });
        timeit("+ image scatter, map entry prefetched", [&] { GO(64, 15); });
        timeit("CW=32, all parts, map prefetched", [&] { GO(32, 15); });
        timeit("CW=128, all parts, map prefetched", [&] { GO(128, 15); });
        timeit("empty launch (1 block)", [&] { hipLaunchKernelGGL((red<64, 0>), dim3(1), dim3(256), 0, 0, slab, 1, 64, 0, gb, th, m, v, imap, img); });
    }
    return 0;
}
