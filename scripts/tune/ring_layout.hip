// ring_layout.hip — L-BFGS combine mix (R 2m ring vectors + g / W u): does ONE block-interleaved ring array
// ring[block][slot][B pairs] read faster than 2m separate arrays?   (the separate-array form sits at
// 5.2–5.5 TB/s under every access pattern tried, the 3-stream read-only launch at 6.4–6.8.)
// build: hipcc -O3 --offload-arch=gfx950 ring_layout.hip -o ring_layout     run: ./ring_layout [n] [slots]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int BLOCK = 256, GRID = 4096, MAXS = 24;

template <bool NT> __device__ inline d2 ld(const double *p, long long pair) {
    const d2 *q = reinterpret_cast<const d2 *>(p) + pair;
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}
template <bool NT> __device__ inline void st(double *p, long long pair, d2 v) {
    d2 *q = reinterpret_cast<d2 *>(p) + pair;
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}
struct PS { const double *s[MAXS]; const double *g; double *out; long long n2; int ns; };
struct PI { const double *ring; const double *g; double *out; long long n2; int ns; };

template <int NS, bool NT>
__global__ __launch_bounds__(BLOCK) void k_sep(PS p) {
    const long long per = (p.n2 + gridDim.x - 1) / gridDim.x;
    const long long hi = (per * blockIdx.x + per < p.n2) ? per * blockIdx.x + per : p.n2;
    for (long long i = per * blockIdx.x + threadIdx.x; i < hi; i += BLOCK) {
        d2 v[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) v[j] = ld<NT>(p.s[j], i);
        d2 r = ld<NT>(p.g, i);
#pragma unroll
        for (int j = 0; j < NS; ++j) r = r + (0.5 + j) * v[j];
        st<NT>(p.out, i, r);
    }
}
// ring[block][slot][B pairs]
template <int NS, int B, bool NT>
__global__ __launch_bounds__(BLOCK) void k_il(PI p) {
    const long long per = (p.n2 + gridDim.x - 1) / gridDim.x;
    const long long hi = (per * blockIdx.x + per < p.n2) ? per * blockIdx.x + per : p.n2;
    for (long long i = per * blockIdx.x + threadIdx.x; i < hi; i += BLOCK) {
        const long long base = (i / B) * ((long long)NS * B) + (i % B);
        d2 v[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) v[j] = ld<NT>(p.ring, base + (long long)j * B);
        d2 r = ld<NT>(p.g, i);
#pragma unroll
        for (int j = 0; j < NS; ++j) r = r + (0.5 + j) * v[j];
        st<NT>(p.out, i, r);
    }
}
__global__ void k_fill(double *p, long long n, double v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
template <class F> static double timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 14; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ts[ts.size() / 2];
}
template <int NS> static void run(long long n) {
    const long long n2 = n / 2;
    const double bytes = 8.0 * n * (NS + 2);
    std::vector<double *> bufs;
    PS ps; ps.n2 = n2; ps.ns = NS;
    for (int j = 0; j < NS + 2; ++j) { double *b; CK(hipMalloc(&b, n * 8 + 4096)); k_fill<<<1024, 256>>>(b, n, 1e-3 * j); bufs.push_back(b); }
    for (int j = 0; j < NS; ++j) ps.s[j] = bufs[j];
    ps.g = bufs[NS]; ps.out = bufs[NS + 1];
    CK(hipDeviceSynchronize());
    double t = timeit([&] { k_sep<NS, true><<<GRID, BLOCK>>>(ps); });
    printf("n=%.1e R%d+g/W1 separate arrays, nt            %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    t = timeit([&] { k_sep<NS, false><<<GRID, BLOCK>>>(ps); });
    printf("n=%.1e R%d+g/W1 separate arrays                %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    for (int j = 0; j < NS; ++j) CK(hipFree(bufs[j]));
    double *ring; CK(hipMalloc(&ring, (size_t)n * 8 * NS + (size_t)NS * 4096 * 16));
    k_fill<<<2048, 256>>>(ring, n * NS, 1e-3);
    CK(hipDeviceSynchronize());
    PI pi; pi.ring = ring; pi.g = bufs[NS]; pi.out = bufs[NS + 1]; pi.n2 = n2; pi.ns = NS;
    t = timeit([&] { k_il<NS, 64, true><<<GRID, BLOCK>>>(pi); });
    printf("n=%.1e R%d+g/W1 ring[block][slot][64 pairs], nt   %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    t = timeit([&] { k_il<NS, 256, true><<<GRID, BLOCK>>>(pi); });
    printf("n=%.1e R%d+g/W1 ring[block][slot][256 pairs], nt  %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    t = timeit([&] { k_il<NS, 256, false><<<GRID, BLOCK>>>(pi); });
    printf("n=%.1e R%d+g/W1 ring[block][slot][256 pairs]      %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    t = timeit([&] { k_il<NS, 1024, true><<<GRID, BLOCK>>>(pi); });
    printf("n=%.1e R%d+g/W1 ring[block][slot][1024 pairs], nt %8.1f us  %6.0f GB/s\n", (double)n, NS, t, bytes / t * 1e-3);
    CK(hipFree(ring)); CK(hipFree(bufs[NS])); CK(hipFree(bufs[NS + 1]));
    fflush(stdout);
}
int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 10000000LL;
    run<23>(n);   // ≈ the push + Gram launch: ring (20) + x, u, g, g⁺ read
    run<20>(n);   // the combine launch: ring + g read, u written
    run<10>(n);
    run<3>(n);
    return 0;
}
