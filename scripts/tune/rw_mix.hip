// rw_mix.hip — what does MI355X HBM deliver for the READ/WRITE MIX of the accept+dir+trial launch?
//
// The dominant launch of the engine (k_cg<accept_dir_trial>) reads x, u, D and writes x, u in place:
// 3 read + 2 write streams of 16 B per lane, 40 B per element.  This harness runs that mix with next to no
// arithmetic under every streaming policy we can think of, next to the pure-read (trial launch), copy and
// out-of-place mixes, so that the engine's kernel can be priced against what the memory system gives this
// mix on the SAME box rather than against the 8 TB/s pin peak.
//
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 rw_mix.hip -o rw_mix ; run: ./rw_mix [n] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

struct P { double *x, *u; const double *d; double *x2, *u2; long long n; double a, b; double *sink; };

template <bool NT> __device__ inline d2 ld(const double *p, long long i) {
    const d2 *q = reinterpret_cast<const d2 *>(p) + i;
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}
template <bool NT> __device__ inline void st(double *p, long long i, d2 v) {
    d2 *q = reinterpret_cast<d2 *>(p) + i;
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}

// KIND 0: R x,u,D / W x,u in place (the engine's mix)   1: R x,u,D only (trial launch)
//      2: copy R x / W x2                               3: R x,u,D / W x2,u2 (out of place)
//      4: R x,u / W x,u in place                         5: W x,u only (fill)
struct V { d2 x, u, d; };
template <int KIND, bool NTL> __device__ inline void load(const P &p, long long i, V &v) {
    if (KIND != 5) v.x = ld<NTL>(p.x, i);
    if (KIND == 0 || KIND == 1 || KIND == 3 || KIND == 4) v.u = ld<NTL>(p.u, i);
    if (KIND == 0 || KIND == 1 || KIND == 3) v.d = ld<NTL>(p.d, i);
}
template <int KIND, bool NTS> __device__ inline void body(const P &p, long long i, V &v, double &acc) {
    if (KIND == 1) { acc += v.x.x * v.u.x + v.d.x; acc += v.x.y * v.u.y + v.d.y; return; }
    if (KIND == 2) { st<NTS>(p.x2, i, v.x); return; }
    if (KIND == 5) { d2 c; c.x = p.a; c.y = p.b; st<NTS>(p.x, i, c); st<NTS>(p.u, i, c); return; }
    d2 xn, un;
    xn.x = v.x.x + p.a * v.u.x; xn.y = v.x.y + p.a * v.u.y;
    if (KIND == 4) { un.x = p.b * v.u.x - xn.x * 1e-9; un.y = p.b * v.u.y - xn.y * 1e-9; }
    else { un.x = p.b * v.u.x - (v.d.x * xn.x) * 1e-9; un.y = p.b * v.u.y - (v.d.y * xn.y) * 1e-9; }
    if (KIND == 3) { st<NTS>(p.x2, i, xn); st<NTS>(p.u2, i, un); }
    else { st<NTS>(p.x, i, xn); st<NTS>(p.u, i, un); }
}

// POLICY 0: grid-stride.  1: one contiguous chunk per workgroup.  2: contiguous chunk per WAVE (each wave streams its own
// 1-KiB-granular run: fewer DRAM pages open per CU).
template <int KIND, int POLICY, int UNROLL, bool NTL, bool NTS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(P p) {
    double acc = 0.0;
    const long long n2 = p.n >> 1;
    long long i, hi, step;
    if (POLICY == 0) {
        i = (long long)blockIdx.x * THREADS + threadIdx.x; hi = n2; step = (long long)gridDim.x * THREADS;
    } else if (POLICY == 1) {
        const long long per = (n2 + gridDim.x - 1) / gridDim.x;
        i = per * blockIdx.x + threadIdx.x; hi = std::min(per * blockIdx.x + per, n2); step = THREADS;
    } else {
        const long long waves = (long long)gridDim.x * (THREADS / 64);
        const long long per = ((n2 + waves - 1) / waves + 63) / 64 * 64;
        const long long w = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
        i = per * w + (threadIdx.x & 63); hi = std::min(per * w + per, n2); step = 64;
    }
    for (; i + (UNROLL - 1) * step < hi; i += UNROLL * step) {
        V v[UNROLL];
#pragma unroll
        for (int k2 = 0; k2 < UNROLL; ++k2) load<KIND, NTL>(p, i + k2 * step, v[k2]);
#pragma unroll
        for (int k2 = 0; k2 < UNROLL; ++k2) body<KIND, NTS>(p, i + k2 * step, v[k2], acc);
    }
    for (; i < hi; i += step) { V v; load<KIND, NTL>(p, i, v); body<KIND, NTS>(p, i, v, acc); }
    if (KIND == 1 && acc == 123.456) p.sink[0] = acc;
}

// ---- many read streams → one write stream (the L-BFGS combine / Gram launches: 21–24 streams of 16 B per lane) --------
// PATTERN 0: per element group, one 16-B load from each of NS streams (what k_lbfgs_combine does).
// PATTERN 1: U groups per lane per trip, stream by stream with the next stream's loads issued before this stream's FMAs
//            (bigger contiguous bursts per stream: U·4 KiB per workgroup instead of 4 KiB).
__global__ void fill(double *v, long long n, double a, double b);
struct Env { hipStream_t st; hipEvent_t e0, e1; int reps; FILE *csv; };
struct PM { const double *s[24]; double *out; long long n; int ns; };
template <int PATTERN, int U, bool NT>
__global__ __launch_bounds__(256) void kmulti(PM p) {
    const long long n2 = p.n >> 1;
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long long hi = std::min(per * blockIdx.x + per, n2);
    long long i = per * blockIdx.x + threadIdx.x;
    if (PATTERN == 0) {
        for (; i < hi; i += 256) {
            d2 v[24];
#pragma unroll
            for (int j = 0; j < 24; ++j) if (j < p.ns) v[j] = ld<NT>(p.s[j], i);
            d2 r; r.x = 0; r.y = 0;
#pragma unroll
            for (int j = 0; j < 24; ++j) if (j < p.ns) { r.x = r.x + 0.5 * v[j].x; r.y = r.y + 0.5 * v[j].y; }
            st<NT>(p.out, i, r);
        }
    } else {
        for (; i + (U - 1) * 256 < hi; i += U * 256) {
            d2 r[U], cur[U], nxt[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { r[k].x = 0; r[k].y = 0; cur[k] = ld<NT>(p.s[0], i + k * 256); }
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                if (j < p.ns) {
                    if (j + 1 < p.ns) {
#pragma unroll
                        for (int k = 0; k < U; ++k) nxt[k] = ld<NT>(p.s[j + 1], i + k * 256);
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) { r[k].x = r[k].x + 0.5 * cur[k].x; r[k].y = r[k].y + 0.5 * cur[k].y; cur[k] = nxt[k]; }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) st<NT>(p.out, i + k * 256, r[k]);
        }
        for (; i < hi; i += 256) {
            d2 r; r.x = 0; r.y = 0;
            for (int j = 0; j < p.ns; ++j) { const d2 v = ld<NT>(p.s[j], i); r.x = r.x + 0.5 * v.x; r.y = r.y + 0.5 * v.y; }
            st<NT>(p.out, i, r);
        }
    }
}

template <int PATTERN, int U, bool NT>
static void runm(const Env &E, PM p, int grid, const char *name) {
    for (int w = 0; w < 2; ++w) kmulti<PATTERN, U, NT><<<grid, 256, 0, E.st>>>(p);
    std::vector<float> t(E.reps);
    for (int r = 0; r < E.reps; ++r) {
        CK(hipEventRecord(E.e0, E.st));
        kmulti<PATTERN, U, NT><<<grid, 256, 0, E.st>>>(p);
        CK(hipEventRecord(E.e1, E.st));
        CK(hipStreamSynchronize(E.st));
        CK(hipEventElapsedTime(&t[r], E.e0, E.e1));
    }
    std::sort(t.begin(), t.end());
    const double gb = 8.0 * p.n * (p.ns + 1) / t[E.reps / 2] / 1e6;
    printf("n=%.2e R%d/W1 %-34s U%d %s grid=%5d  med %8.1f us  %7.1f GB/s (%4.1f%% of 8 TB/s)\n", (double)p.n, p.ns, name, U, NT ? "nt" : "  ", grid,
           t[E.reps / 2] * 1e3, gb, gb / 80.0);
    fflush(stdout);
}

static void many_streams(const Env &E, long long n, int ns) {
    PM p; p.n = n; p.ns = ns;
    std::vector<double *> bufs;
    for (int j = 0; j <= ns; ++j) { double *b; CK(hipMalloc(&b, (size_t)n * 8 + 4096)); bufs.push_back(b); fill<<<2048, 256, 0, E.st>>>(b, n, 1.0, 0.1 * j); }
    for (int j = 0; j < 24; ++j) p.s[j] = bufs[j < ns ? j : 0];
    p.out = bufs[ns];
    CK(hipStreamSynchronize(E.st));
    printf("== %d read streams -> 1 write stream (L-BFGS combine mix)\n", ns);
    runm<0, 1, true>(E, p, 4096, "all streams per group");
    runm<0, 1, false>(E, p, 4096, "all streams per group");
    runm<0, 1, true>(E, p, 2048, "all streams per group");
    runm<1, 2, true>(E, p, 4096, "stream by stream, prefetch next");
    runm<1, 4, true>(E, p, 4096, "stream by stream, prefetch next");
    runm<1, 8, true>(E, p, 4096, "stream by stream, prefetch next");
    runm<1, 4, true>(E, p, 2048, "stream by stream, prefetch next");
    runm<1, 8, true>(E, p, 2048, "stream by stream, prefetch next");
    runm<1, 4, false>(E, p, 4096, "stream by stream, prefetch next");
    for (double *b : bufs) CK(hipFree(b));
}

__global__ void fill(double *v, long long n, double a, double b) {
    const long long T = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) v[i] = a + b * (double)(i % 1000) / 1000.0;
}

static double bytes_of(int kind, long long n) { const int v = kind == 0 ? 5 : kind == 1 ? 3 : kind == 2 ? 2 : kind == 3 ? 5 : kind == 4 ? 4 : 2; return 8.0 * n * v; }
static const char *kind_name(int k) { const char *nm[] = {"R3W2 in place", "R3 (trial)", "copy R1W1", "R3W2 out of place", "R2W2 in place", "W2 fill"}; return nm[k]; }


template <int KIND, int POLICY, int UNROLL, bool NTL, bool NTS, int THREADS>
static double run(const Env &E, P p, int grid) {
    const long long n2 = p.n / 2;
    if (POLICY == 0) { long long b = (n2 + (long long)THREADS * UNROLL - 1) / ((long long)THREADS * UNROLL); if (b < grid) grid = (int)std::max<long long>(b, 1); }
    for (int w = 0; w < 2; ++w) k<KIND, POLICY, UNROLL, NTL, NTS, THREADS><<<grid, THREADS, 0, E.st>>>(p);
    std::vector<float> t(E.reps);
    for (int r = 0; r < E.reps; ++r) {   // per-launch events: report the MEDIAN launch (robust to one slow launch)
        CK(hipEventRecord(E.e0, E.st));
        k<KIND, POLICY, UNROLL, NTL, NTS, THREADS><<<grid, THREADS, 0, E.st>>>(p);
        CK(hipEventRecord(E.e1, E.st));
        CK(hipStreamSynchronize(E.st));
        CK(hipEventElapsedTime(&t[r], E.e0, E.e1));
    }
    std::sort(t.begin(), t.end());
    const double med = t[E.reps / 2], best = t[0];
    const double gb = bytes_of(KIND, p.n) / med / 1e6;
    const char *pol[] = {"grid-stride", "chunk/WG", "chunk/wave"};
    printf("n=%.2e %-18s %-11s U%d thr%-4d %s%s grid=%5d  med %8.1f us  best %8.1f us  %7.1f GB/s (%4.1f%% of 8 TB/s)\n", (double)p.n,
           kind_name(KIND), pol[POLICY], UNROLL, THREADS, NTL ? "ntL " : "    ", NTS ? "ntS " : "    ", grid, med * 1e3, best * 1e3, gb, gb / 80.0);
    if (E.csv) fprintf(E.csv, "%lld,%s,%s,%d,%d,%d,%d,%d,%.2f,%.2f,%.1f\n", p.n, kind_name(KIND), pol[POLICY], UNROLL, THREADS, (int)NTL, (int)NTS, grid, med * 1e3, best * 1e3, gb);
    fflush(stdout);
    return gb;
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 100000000LL;
    Env E; E.reps = argc > 2 ? atoi(argv[2]) : 15;
    E.csv = (argc > 3 && strcmp(argv[3], "many") != 0) ? fopen(argv[3], "a") : nullptr;
    CK(hipStreamCreate(&E.st)); CK(hipEventCreate(&E.e0)); CK(hipEventCreate(&E.e1));
    if (argc > 3 && strcmp(argv[3], "many") == 0) { many_streams(E, n, 21); many_streams(E, n, 12); return 0; }
    if (argc > 3 && strcmp(argv[3], "spacer") == 0) {
        // Hypothesis from the "place" run: the two WRITTEN streams must not share a ~4-GiB physical region.  x, then a spacer of
        // S GiB, then u, then D (allocated in this order, the spacer kept or freed), the in-place mix for each S.
        const size_t N8 = (size_t)n * 8;
        for (int keep = 1; keep >= 0; --keep)
            for (double gib : {0.0, 0.5, 1.0, 2.0, 3.0, 3.5, 4.0, 5.0, 6.0, 8.0, 12.0}) {
                double *x, *u, *d; char *sp = nullptr;
                CK(hipMalloc(&x, N8));
                if (gib > 0) CK(hipMalloc(&sp, (size_t)(gib * 1073741824.0)));
                CK(hipMalloc(&u, N8));
                if (!keep && sp) { CK(hipFree(sp)); sp = nullptr; }
                CK(hipMalloc(&d, N8));
                fill<<<2048, 256, 0, E.st>>>(x, n, 1.0, 0.1); fill<<<2048, 256, 0, E.st>>>(u, n, -1.0, 0.3); fill<<<2048, 256, 0, E.st>>>(d, n, 1.0, 9.0);
                CK(hipStreamSynchronize(E.st));
                P p{x, u, d, nullptr, nullptr, n, 1e-9, 0.5, nullptr};
                printf("spacer %5.1f GiB (%s) x=%p u=%p d=%p: ", gib, keep ? "kept " : "freed", (void *)x, (void *)u, (void *)d);
                run<0, 1, 2, true, true, 256>(E, p, 4096);
                CK(hipFree(x)); CK(hipFree(u)); CK(hipFree(d)); if (sp) CK(hipFree(sp));
            }
        return 0;
    }
    if (argc > 3 && strcmp(argv[3], "place") == 0) {
        // Which physical placement of x, u, D does the mix like?  Ten separately allocated candidate buffers (sizes padded by
        // different amounts so that the allocator places them differently), every triple of them as (x, u, D).
        const size_t N8 = (size_t)n * 8;
        const int NB = 10;
        const size_t pad[NB] = {0, 0, 4096, 2u << 20, 0, 6u << 20, 1u << 20, 0, 34u << 20, 0};
        double *b[NB];
        for (int k = 0; k < NB; ++k) { CK(hipMalloc(&b[k], N8 + pad[k])); fill<<<2048, 256, 0, E.st>>>(b[k], n, 1.0 + k, 0.1); printf("buf %d = %p (pad %zu)\n", k, (void *)b[k], pad[k]); }
        CK(hipStreamSynchronize(E.st));
        E.reps = argc > 4 ? 3 : 5;
        std::vector<std::pair<float, int>> res;
        for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) for (int k2 = 0; k2 < NB; ++k2) {
            if (i == j || j == k2 || i == k2) continue;
            if (argc > 4 ? false : ((i * 7 + j * 3 + k2) % 6 != 0)) continue;   // a sixth of the 720 ordered triples (argv[4] = all)
            P p{b[i], b[j], b[k2], nullptr, nullptr, n, 1e-9, 0.5, nullptr};
            for (int w = 0; w < 1; ++w) k<0, 1, 2, true, true, 256><<<4096, 256, 0, E.st>>>(p);
            std::vector<float> t(E.reps);
            for (int r = 0; r < E.reps; ++r) {
                CK(hipEventRecord(E.e0, E.st));
                k<0, 1, 2, true, true, 256><<<4096, 256, 0, E.st>>>(p);
                CK(hipEventRecord(E.e1, E.st)); CK(hipStreamSynchronize(E.st));
                CK(hipEventElapsedTime(&t[r], E.e0, E.e1));
            }
            std::sort(t.begin(), t.end());
            res.push_back({t[E.reps / 2] * 1e3f, i * 100 + j * 10 + k2});
            if (argc > 4) printf("T %d %d %d %.1f\n", i, j, k2, t[E.reps / 2] * 1e3f);
        }
        std::sort(res.begin(), res.end());
        printf("%zu triples: fastest %.1f us (x,u,D = %03d), 10%% %.1f, median %.1f (%03d), 90%% %.1f, slowest %.1f us (%03d)\n", res.size(), res[0].first, res[0].second,
               res[res.size() / 10].first, res[res.size() / 2].first, res[res.size() / 2].second, res[res.size() * 9 / 10].first, res.back().first, res.back().second);
        for (size_t q = 0; q < res.size(); q += std::max<size_t>(1, res.size() / 24)) printf("  %.1f us  %03d\n", res[q].first, res[q].second);
        // is a fast triple fast again?  (twice, after the others ran)
        for (int rep = 0; rep < 2; ++rep) for (size_t q : {(size_t)0, res.size() - 1}) {
            const int c = res[q].second; P p{b[c / 100], b[(c / 10) % 10], b[c % 10], nullptr, nullptr, n, 1e-9, 0.5, nullptr};
            printf("again %03d: ", c); run<0, 1, 2, true, true, 256>(E, p, 4096);
        }
        return 0;
    }
    if (argc > 3 && strcmp(argv[3], "arena") == 0) {
        // Does the RELATIVE PLACEMENT of x, u, D matter?  One arena (one hipMalloc: large allocations come in large physically
        // contiguous blocks), the three vectors carved out of it with a stagger S between consecutive ones, the in-place mix
        // under the engine's policy for each S; then the same with three separate hipMallocs, allocated in two orders.
        const size_t N8 = (size_t)n * 8;
        const size_t staggers[] = {0, 256, 4096, 65536, 1u << 20, 2u << 20, (2u << 20) + 4096, 3u << 20, 8u << 20, (8u << 20) + (1u << 16), 32u << 20, 33u << 20,
                                   64u << 20, 96u << 20, 100u << 20, 128u << 20, 192u << 20, 256u << 20};
        char *arena;
        const size_t maxs = 256u << 20;
        CK(hipMalloc(&arena, 3 * (N8 + maxs) + (4u << 20)));
        for (size_t S : staggers) {
            double *x = (double *)arena, *u = (double *)(arena + N8 + S), *d = (double *)(arena + 2 * (N8 + S));
            fill<<<2048, 256, 0, E.st>>>(x, n, 1.0, 0.1); fill<<<2048, 256, 0, E.st>>>(u, n, -1.0, 0.3); fill<<<2048, 256, 0, E.st>>>(d, n, 1.0, 9.0);
            CK(hipStreamSynchronize(E.st));
            P p{x, u, d, nullptr, nullptr, n, 1e-9, 0.5, nullptr};
            printf("arena stagger %10zu B: ", S);
            run<0, 1, 2, true, true, 256>(E, p, 4096);
        }
        CK(hipFree(arena));
        for (int order = 0; order < 3; ++order) {
            double *b[3];
            for (int k = 0; k < 3; ++k) CK(hipMalloc(&b[k], N8 + (order == 2 ? (size_t)(k + 1) * (3u << 20) : 0)));
            double *x = b[order == 1 ? 2 : 0], *u = b[1], *d = b[order == 1 ? 0 : 2];
            fill<<<2048, 256, 0, E.st>>>(x, n, 1.0, 0.1); fill<<<2048, 256, 0, E.st>>>(u, n, -1.0, 0.3); fill<<<2048, 256, 0, E.st>>>(d, n, 1.0, 9.0);
            CK(hipStreamSynchronize(E.st));
            P p{x, u, d, nullptr, nullptr, n, 1e-9, 0.5, nullptr};
            printf("separate hipMallocs, order %d (x=%p u=%p d=%p): ", order, (void *)x, (void *)u, (void *)d);
            run<0, 1, 2, true, true, 256>(E, p, 4096);
            for (int k = 0; k < 3; ++k) CK(hipFree(b[k]));
        }
        return 0;
    }
    double *x, *u, *d, *x2, *u2, *sink;
    const size_t B = (size_t)n * 8 + 4096;
    CK(hipMalloc(&x, B)); CK(hipMalloc(&u, B)); CK(hipMalloc(&d, B)); CK(hipMalloc(&x2, B)); CK(hipMalloc(&u2, B)); CK(hipMalloc(&sink, 64));
    fill<<<2048, 256, 0, E.st>>>(x, n, 1.0, 0.1); fill<<<2048, 256, 0, E.st>>>(u, n, -1.0, 0.3); fill<<<2048, 256, 0, E.st>>>(d, n, 1.0, 9.0);
    CK(hipStreamSynchronize(E.st));
    P p{x, u, d, x2, u2, n, 1e-9, 0.5, sink};
    if (E.csv) fprintf(E.csv, "n,kind,policy,unroll,threads,nt_loads,nt_stores,grid,median_us,best_us,gbps\n");
    printf("== the engine's mix: R x,u,D / W x,u in place (40 B/elt)\n");
    run<0, 0, 2, false, false, 256>(E, p, 1024);
    run<0, 0, 2, false, false, 256>(E, p, 2048);
    run<0, 0, 2, true, true, 256>(E, p, 2048);
    run<0, 1, 2, false, false, 256>(E, p, 4096);
    run<0, 1, 2, true, true, 256>(E, p, 2048);
    run<0, 1, 2, true, true, 256>(E, p, 4096);     // = the engine's BIG policy
    run<0, 1, 2, true, true, 256>(E, p, 8192);
    run<0, 1, 2, true, true, 256>(E, p, 16384);
    run<0, 1, 2, false, true, 256>(E, p, 4096);
    run<0, 1, 2, true, false, 256>(E, p, 4096);
    run<0, 1, 1, true, true, 256>(E, p, 4096);
    run<0, 1, 4, true, true, 256>(E, p, 4096);
    run<0, 1, 8, true, true, 256>(E, p, 4096);
    run<0, 1, 4, true, true, 256>(E, p, 2048);
    run<0, 1, 8, true, true, 256>(E, p, 2048);
    run<0, 1, 8, true, true, 256>(E, p, 1024);
    run<0, 1, 4, true, true, 512>(E, p, 2048);
    run<0, 1, 4, true, true, 1024>(E, p, 1024);
    run<0, 1, 2, true, true, 1024>(E, p, 1024);
    run<0, 1, 2, true, true, 64>(E, p, 16384);
    run<0, 2, 2, true, true, 256>(E, p, 4096);
    run<0, 2, 4, true, true, 256>(E, p, 4096);
    run<0, 2, 8, true, true, 256>(E, p, 2048);
    run<0, 2, 4, true, true, 256>(E, p, 1024);
    run<0, 2, 4, false, false, 256>(E, p, 4096);
    printf("== reference mixes\n");
    run<1, 1, 2, true, true, 256>(E, p, 4096);
    run<1, 1, 4, true, true, 256>(E, p, 4096);
    run<1, 0, 2, false, false, 256>(E, p, 2048);
    run<2, 1, 2, true, true, 256>(E, p, 4096);
    run<2, 1, 4, true, true, 256>(E, p, 4096);
    run<2, 0, 2, false, false, 256>(E, p, 2048);
    run<3, 1, 2, true, true, 256>(E, p, 4096);
    run<3, 1, 4, true, true, 256>(E, p, 4096);
    run<4, 1, 2, true, true, 256>(E, p, 4096);
    run<4, 1, 4, true, true, 256>(E, p, 4096);
    run<5, 1, 2, true, true, 256>(E, p, 4096);
    run<5, 1, 4, false, false, 256>(E, p, 4096);
    if (E.csv) fclose(E.csv);
    return 0;
}
