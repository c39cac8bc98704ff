// il_mix.hip — does INTERLEAVING x and u in HBM (fewer concurrent DRAM streams) beat separate arrays for the
// accept+dir+trial access mix  R x,u,D / W x,u ?   (with three arrays the mix runs at one of three
// levels depending on which physical buffers it got — bank conflicts between streams walking in lock step.)
//
//   layout 0: x[n], u[n], D[n]                          five streams  (what the engine does)
//   layout 1: XU = blocks of { x: B pairs | u: B pairs }, D[n]        three streams, every wave access contiguous
//   layout 2: XUD = blocks of { x | u | D }                            two streams
//   layout 3: XU = per pair { x0 x1 u0 u1 }, D[n]                      three streams, 32-B granules
// Same streaming policy as the engine's BIG launches: 4096 workgroups, one contiguous chunk each, 256 lanes, 16 B per
// lane and access, non-temporal loads and stores, two independent groups per trip.  The arithmetic is the mix's
// (x += a u ; u = b u − D x): nothing that could bind.
//
// build: hipcc -O3 --offload-arch=gfx950 il_mix.hip -o il_mix      run: ./il_mix [n] [allocations]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int BLOCK = 256, GRID = 4096;

__device__ inline d2 ld(const double *p, long long pair) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p) + pair); }
__device__ inline void st(double *p, long long pair, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p) + pair); }

// pair index → d2 index of x / u / D in each layout
template <int LAYOUT, int B> struct Map {
    __device__ static inline long long x(long long i) {
        if (LAYOUT == 0) return i;
        if (LAYOUT == 1) return (i / B) * (2 * B) + (i % B);
        if (LAYOUT == 2) return (i / B) * (3 * B) + (i % B);
        return 2 * i;
    }
    __device__ static inline long long u(long long i) {
        if (LAYOUT == 0) return i;
        if (LAYOUT == 1) return (i / B) * (2 * B) + B + (i % B);
        if (LAYOUT == 2) return (i / B) * (3 * B) + B + (i % B);
        return 2 * i + 1;
    }
    __device__ static inline long long d(long long i) {
        if (LAYOUT == 2) return (i / B) * (3 * B) + 2 * B + (i % B);
        return i;
    }
};

template <int LAYOUT, int B>
__global__ __launch_bounds__(BLOCK) void k_mix(double *X, double *U, const double *D, long long n2, double a, double b) {
    using M = Map<LAYOUT, B>;
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    long long i = per * blockIdx.x + threadIdx.x;
    const long long hi = (per * blockIdx.x + per < n2) ? per * blockIdx.x + per : n2;
    for (; i + BLOCK < hi; i += 2 * BLOCK) {
        const long long j = i + BLOCK;
        d2 xa = ld(X, M::x(i)), xb = ld(X, M::x(j));
        d2 ua = ld(U, M::u(i)), ub = ld(U, M::u(j));
        const d2 da = ld(D, M::d(i)), db = ld(D, M::d(j));
        xa = xa + a * ua; xb = xb + a * ub;
        ua = b * ua - da * xa; ub = b * ub - db * xb;
        st(X, M::x(i), xa); st(U, M::u(i), ua);
        st(X, M::x(j), xb); st(U, M::u(j), ub);
    }
    if (i < hi) {
        d2 xa = ld(X, M::x(i)), ua = ld(U, M::u(i));
        const d2 da = ld(D, M::d(i));
        xa = xa + a * ua; ua = b * ua - da * xa;
        st(X, M::x(i), xa); st(U, M::u(i), ua);
    }
}

__global__ void k_fill(double *p, long long n, double v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

template <int LAYOUT, int B>
static void run(const char *name, long long n, int allocs) {
    const long long n2 = n / 2;
    std::vector<double> best_of_alloc;
    for (int t = 0; t < allocs; ++t) {
        double *bx = nullptr, *bu = nullptr, *bd = nullptr;
        double *X, *U; const double *D;
        if (LAYOUT == 0) {
            CK(hipMalloc(&bx, n * 8)); CK(hipMalloc(&bu, n * 8)); CK(hipMalloc(&bd, n * 8));
            X = bx; U = bu; D = bd;
            k_fill<<<1024, 256>>>(bx, n, 1.0); k_fill<<<1024, 256>>>(bu, n, 0.5); k_fill<<<1024, 256>>>(bd, n, 1e-3);
        } else if (LAYOUT == 2) {
            CK(hipMalloc(&bx, 3 * n * 8 + 3 * B * 16));
            X = U = bx; D = bx;
            k_fill<<<1024, 256>>>(bx, 3 * n, 1e-3);
        } else {
            CK(hipMalloc(&bx, 2 * n * 8 + 2 * B * 16)); CK(hipMalloc(&bd, n * 8));
            X = U = bx; D = bd;
            k_fill<<<1024, 256>>>(bx, 2 * n, 0.5); k_fill<<<1024, 256>>>(bd, n, 1e-3);
        }
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> ts;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0));
            k_mix<LAYOUT, B><<<GRID, BLOCK>>>(X, U, D, n2, 1e-9, 0.5);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        best_of_alloc.push_back(ts[ts.size() / 2]);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        // keep the buffers of this round allocated while the next round allocates (new physical pages each time)
        static std::vector<void *> keep;
        keep.push_back(bx); keep.push_back(bu); keep.push_back(bd);
        if (keep.size() > 9) { for (int q = 0; q < 3; ++q) { if (keep[q]) CK(hipFree(keep[q])); } keep.erase(keep.begin(), keep.begin() + 3); }
    }
    std::vector<double> s = best_of_alloc; std::sort(s.begin(), s.end());
    printf("%-46s n=%.2e  median-of-10 per allocation:", name, (double)n);
    for (double v : best_of_alloc) printf(" %6.1f", v);
    printf("   | min %6.1f med %6.1f max %6.1f us  -> %.0f GB/s at the median\n", s.front(), s[s.size() / 2], s.back(), 40.0 * n / s[s.size() / 2] * 1e-3);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 100000000LL;
    const int allocs = argc > 2 ? atoi(argv[2]) : 6;
    run<0, 64>("0: x, u, D separate (engine)", n, allocs);
    run<1, 64>("1: {x|u} blocks of 64 pairs (1 KB), D", n, allocs);
    run<1, 128>("1: {x|u} blocks of 128 pairs (2 KB), D", n, allocs);
    run<1, 256>("1: {x|u} blocks of 256 pairs (4 KB), D", n, allocs);
    run<1, 4096>("1: {x|u} blocks of 4096 pairs (64 KB), D", n, allocs);
    run<2, 64>("2: {x|u|D} blocks of 64 pairs", n, allocs);
    run<2, 256>("2: {x|u|D} blocks of 256 pairs", n, allocs);
    run<3, 1>("3: {x0 x1 u0 u1} per pair, D", n, allocs);
    run<0, 64>("0: x, u, D separate (again)", n, allocs);
    return 0;
}
