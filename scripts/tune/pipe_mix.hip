// pipe_mix.hip — does a software pipeline (next trip's loads in flight during this trip's arithmetic) pay at the 8-GPU shard
// size, and how does the launch time depend on the FP64 work per element and on waves per SIMD?
//
// Same streams as the engine's accept+dir+trial launch (R x,u,D / W x,u in place, 16 B per lane, grid-stride, two groups per
// lane per trip) with WORK dependent-chain FP64 FMAs per element pair standing in for the trial points (the 7-point quadratic
// body is ≈ 210 VALU instructions per pair), REGS extra live accumulators to pin the register budget, PF = prefetch on/off.
//
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 pipe_mix.hip -o pipe_mix ; run: ./pipe_mix [n] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

struct P { double *x, *u; const double *d; long long n; double a, b; double *sink; };
__device__ inline d2 ld(const double *p, long long i) { return *(reinterpret_cast<const d2 *>(p) + i); }
__device__ inline void st(double *p, long long i, d2 v) { *(reinterpret_cast<d2 *>(p) + i) = v; }

template <int WORK, int NACC>
__device__ inline void body(const P &p, long long i, d2 x, d2 u, d2 d, double (&acc)[NACC]) {
    d2 xn, un;
    xn.x = x.x + p.a * u.x; xn.y = x.y + p.a * u.y;
    un.x = p.b * u.x - (d.x * xn.x) * 1e-9; un.y = p.b * u.y - (d.y * xn.y) * 1e-9;
    // WORK FMAs per pair spread over NACC independent accumulators (like the 7 × 7 sums of the trial points)
#pragma unroll
    for (int k = 0; k < WORK; ++k) {
        const double t = (k & 1) ? xn.y : xn.x, s = (k & 2) ? un.y : un.x;
        acc[k % NACC] = __builtin_fma(t, s, acc[k % NACC]);
    }
    st(p.x, i, xn); st(p.u, i, un);
}

template <int WORK, int NACC, bool PF>
__global__ __launch_bounds__(256) void k(P p) {
    double acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s) acc[s] = 0.0;
    const long long n2 = p.n >> 1, step = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (PF) {
        d2 xa, xb, ua, ub, da, db;
        bool have = i + step < n2;
        if (have) { xa = ld(p.x, i); xb = ld(p.x, i + step); ua = ld(p.u, i); ub = ld(p.u, i + step); da = ld(p.d, i); db = ld(p.d, i + step); }
        while (have) {
            const long long in = i + 2 * step;
            const bool more = in + step < n2;
            d2 nxa, nxb, nua, nub, nda, ndb;
            if (more) { nxa = ld(p.x, in); nxb = ld(p.x, in + step); nua = ld(p.u, in); nub = ld(p.u, in + step); nda = ld(p.d, in); ndb = ld(p.d, in + step); }
            __builtin_amdgcn_sched_barrier(0);   // the prefetch stays in front of the arithmetic
            body<WORK, NACC>(p, i, xa, ua, da, acc); body<WORK, NACC>(p, i + step, xb, ub, db, acc);
            xa = nxa; xb = nxb; ua = nua; ub = nub; da = nda; db = ndb;
            i = in; have = more;
        }
    } else {
        for (; i + step < n2; i += 2 * step) {
            const d2 xa = ld(p.x, i), xb = ld(p.x, i + step), ua = ld(p.u, i), ub = ld(p.u, i + step), da = ld(p.d, i), db = ld(p.d, i + step);
            body<WORK, NACC>(p, i, xa, ua, da, acc); body<WORK, NACC>(p, i + step, xb, ub, db, acc);
        }
    }
    if (i < n2) body<WORK, NACC>(p, i, ld(p.x, i), ld(p.u, i), ld(p.d, i), acc);
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < NACC; ++s) t += acc[s];
    if (t == 123.456) p.sink[0] = t;
}

__global__ void fill(double *v, long long n, double a) {
    const long long T = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) v[i] = a + 1e-3 * (double)(i % 1000);
}

template <int WORK, int NACC, bool PF>
static void run(hipStream_t s, hipEvent_t e0, hipEvent_t e1, P p, int grid, int reps) {
    for (int w = 0; w < 3; ++w) k<WORK, NACC, PF><<<grid, 256, 0, s>>>(p);
    std::vector<float> t(reps);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        k<WORK, NACC, PF><<<grid, 256, 0, s>>>(p);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&t[r], e0, e1));
    }
    std::sort(t.begin(), t.end());
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k<WORK, NACC, PF>)));
    printf("work %3d acc %2d %s grid %4d  regs %3d  med %7.1f us  best %7.1f us  %6.0f GB/s\n", WORK, NACC, PF ? "prefetch" : "plain   ", grid, fa.numRegs,
           t[reps / 2] * 1e3, t[0] * 1e3, 40.0 * p.n / (t[reps / 2] * 1e3) / 1e3);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 12500000LL;
    const int reps = argc > 2 ? atoi(argv[2]) : 15;
    hipStream_t s; hipEvent_t e0, e1;
    CK(hipStreamCreate(&s)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double *x, *u, *d, *sink;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&u, n * 8)); CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&sink, 64));
    fill<<<2048, 256, 0, s>>>(x, n, 1.0); fill<<<2048, 256, 0, s>>>(u, n, 2.0); fill<<<2048, 256, 0, s>>>(d, n, 3.0);
    P p{x, u, d, n, 1e-12, 1.0, sink};
    for (int grid : {256, 512, 768, 1024}) {
        run<0, 8, false>(s, e0, e1, p, grid, reps);   run<0, 8, true>(s, e0, e1, p, grid, reps);
        run<48, 8, false>(s, e0, e1, p, grid, reps);  run<48, 8, true>(s, e0, e1, p, grid, reps);
        run<100, 24, false>(s, e0, e1, p, grid, reps); run<100, 24, true>(s, e0, e1, p, grid, reps);
        run<200, 56, false>(s, e0, e1, p, grid, reps); run<200, 56, true>(s, e0, e1, p, grid, reps);
        run<200, 30, false>(s, e0, e1, p, grid, reps); run<200, 30, true>(s, e0, e1, p, grid, reps);
        run<140, 30, false>(s, e0, e1, p, grid, reps); run<140, 30, true>(s, e0, e1, p, grid, reps);
    }
    return 0;
}
