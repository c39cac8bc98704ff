// place_pmc.hip — ONE process, one fast and one slow (x, u, D) buffer triple, the same five-stream mix on both, under
// rocprofv3 --pmc: what differs in the memory system between the two placement levels of DESIGN.md §2.5?
//
// The accept+dir+trial launch of the engine reads x, u, D and writes x, u in place.  With ten separately allocated
// buffers and every ordered triple of them as (x, u, D), that mix runs at one of three levels at n = 1e8 (≈ 635 / 715 /
// 755 µs), stable per triple.  This harness times a sample of triples with k_mix<0>, then runs k_mix<1> on the fastest
// and k_mix<2> on the slowest triple found (same code, different symbol: the counter CSV separates them by kernel name).
//
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 place_pmc.hip -o place_pmc ; run: ./place_pmc [n] [nbuf] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

struct P { double *x, *u; const double *d; long long n; double a, b; };

__device__ inline d2 ld(const double *p, long long i) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p) + i); }
__device__ inline void st(double *p, long long i, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p) + i); }

__device__ inline void body(const P &p, long long i, d2 x, d2 u, d2 d) {
    d2 xn, un;
    xn.x = x.x + p.a * u.x; xn.y = x.y + p.a * u.y;
    un.x = p.b * u.x - (d.x * xn.x) * 1e-9; un.y = p.b * u.y - (d.y * xn.y) * 1e-9;
    st(p.x, i, xn); st(p.u, i, un);
}

// the engine's pure-HBM policy: one contiguous chunk per workgroup, non-temporal, two 16-B groups per lane per trip
template <int TAG>
__global__ __launch_bounds__(256) void k_mix(P p) {
    const long long n2 = p.n >> 1;
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    long long i = per * blockIdx.x + threadIdx.x;
    const long long hi = std::min(per * blockIdx.x + per, n2);
    for (; i + 256 < hi; i += 512) {
        const d2 xa = ld(p.x, i), xb = ld(p.x, i + 256), ua = ld(p.u, i), ub = ld(p.u, i + 256), da = ld(p.d, i), db = ld(p.d, i + 256);
        body(p, i, xa, ua, da); body(p, i + 256, xb, ub, db);
    }
    for (; i < hi; i += 256) body(p, i, ld(p.x, i), ld(p.u, i), ld(p.d, i));
}

__global__ void fill(double *v, long long n, double a) {
    const long long T = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) v[i] = a + 1e-3 * (double)(i % 1000);
}

template <int TAG>
static double time_mix(hipStream_t s, hipEvent_t e0, hipEvent_t e1, P p, int reps) {
    k_mix<TAG><<<4096, 256, 0, s>>>(p);
    std::vector<float> t(reps);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        k_mix<TAG><<<4096, 256, 0, s>>>(p);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&t[r], e0, e1));
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2] * 1e3;
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 100000000LL;
    const int nbuf = argc > 2 ? atoi(argv[2]) : 10;
    const int reps = argc > 3 ? atoi(argv[3]) : 8;
    hipStream_t s; hipEvent_t e0, e1;
    CK(hipStreamCreate(&s)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double *> b(nbuf);
    for (int j = 0; j < nbuf; ++j) { CK(hipMalloc(&b[j], (size_t)n * 8)); fill<<<2048, 256, 0, s>>>(b[j], n, 1.0 + j); }
    CK(hipStreamSynchronize(s));
    struct T { int x, u, d; double us; };
    std::vector<T> ts;
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int c = 0; c < 60; ++c) {
        int x = next() % nbuf, u = next() % nbuf, d = next() % nbuf;
        if (x == u || x == d || u == d) { --c; continue; }
        P p{b[x], b[u], b[d], n, 1e-12, 1.0};
        const double us = time_mix<0>(s, e0, e1, p, 3);
        ts.push_back({x, u, d, us});
        printf("triple (%d,%d,%d)  %8.1f us  %6.1f GB/s   x=%p u=%p D=%p\n", x, u, d, us, 40.0 * n / us / 1e3, (void *)b[x], (void *)b[u], (void *)b[d]);
    }
    std::sort(ts.begin(), ts.end(), [](const T &a, const T &c) { return a.us < c.us; });
    const T f = ts.front(), w = ts.back();
    printf("FAST triple (%d,%d,%d) %.1f us   SLOW triple (%d,%d,%d) %.1f us   ratio %.3f\n", f.x, f.u, f.d, f.us, w.x, w.u, w.d, w.us, w.us / f.us);
    P pf{b[f.x], b[f.u], b[f.d], n, 1e-12, 1.0}, pw{b[w.x], b[w.u], b[w.d], n, 1e-12, 1.0};
    for (int round = 0; round < 2; ++round) {   // alternate, so that neither level owns a thermal or clock state
        const double tf = time_mix<1>(s, e0, e1, pf, reps), tw = time_mix<2>(s, e0, e1, pw, reps);
        printf("round %d: k_mix<1> (fast) %.1f us = %.1f GB/s ; k_mix<2> (slow) %.1f us = %.1f GB/s\n", round, tf, 40.0 * n / tf / 1e3, tw, 40.0 * n / tw / 1e3);
    }
    return 0;
}
