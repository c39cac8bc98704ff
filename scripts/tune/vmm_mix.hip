// vmm_mix.hip — can the placement lottery of the accept+dir+trial stream be taken out with the virtual-memory API?
//
// Round 2 found the R x,u,D / W x,u mix at n = 1e8 running at one of three speeds (≈ 635 / 715 / 755 µs) depending on
// WHICH physical buffers it touches, and searches whole hipMalloc'd buffers for a fast triple (HipBackend::tune_placement):
// a lottery — some processes' pools hold no fast triple at all.  VERDICT r02 next #6: build x, u, D from physical chunks of
// OUR choosing (hipMemCreate / hipMemAddressReserve / hipMemMap at the allocation granularity), time the bare mix per
// chunk triple, map a fast set into contiguous virtual ranges.
//
// This harness: (1) the granularity; (2) the mix on plain hipMalloc triples (the baseline levels of this process);
// (3) P physical chunks of C bytes; the mix on random chunk triples mapped at three fixed virtual ranges — is "fast" a
// property of a chunk triple?  (4) full-size vectors assembled from (a) chunks in creation order, (b) per-position triples
// chosen fastest-first — against (2).
//
// Build: hipcc -O3 --offload-arch=gfx950 vmm_mix.hip -o vmm_mix ; run: ./vmm_mix [n = 1e8] [chunk MiB = 64] [pool chunks = 60]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s (line %d)\n", hipGetErrorString(e), #x, __LINE__); exit(1); } } while (0)

// the engine's pure-HBM policy: contiguous chunk per workgroup, 4096 workgroups, non-temporal, two groups per lane per trip
__global__ __launch_bounds__(256) void k_mix(double *x, double *u, const double *d, long long n, double a, double b) {
    const long long n2 = n >> 1;
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long long hi = (per * blockIdx.x + per < n2) ? per * blockIdx.x + per : n2;
    long long i = per * blockIdx.x + threadIdx.x;
    auto body = [&](long long j, d2 xv, d2 uv, d2 dv) {
        d2 xn, un;
        xn.x = xv.x + a * uv.x; xn.y = xv.y + a * uv.y;
        un.x = b * uv.x - (dv.x * xn.x) * 1e-9; un.y = b * uv.y - (dv.y * xn.y) * 1e-9;
        __builtin_nontemporal_store(xn, reinterpret_cast<d2 *>(x) + j); __builtin_nontemporal_store(un, reinterpret_cast<d2 *>(u) + j);
    };
    auto ld = [](const double *p, long long j) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p) + j); };
    for (; i + 256 < hi; i += 512) {
        const d2 xa = ld(x, i), xb = ld(x, i + 256), ua = ld(u, i), ub = ld(u, i + 256), da = ld(d, i), db = ld(d, i + 256);
        body(i, xa, ua, da); body(i + 256, xb, ub, db);
    }
    if (i < hi) body(i, ld(x, i), ld(u, i), ld(d, i));
}

static hipEvent_t e0, e1;
static double time_mix(double *x, double *u, const double *d, long long n, int reps = 5) {
    std::vector<float> t;
    for (int r = -1; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        k_mix<<<4096, 256>>>(x, u, d, n, 1e-9, 0.5);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e3;
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 100000000LL;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
    const int pool = argc > 3 ? atoi(argv[3]) : 60;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu B, recommended %zu B; chunk %zu MiB, pool %d chunks, n = %lld\n", gran_min, gran_rec, chunk >> 20, pool, n);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;

    // (2) plain hipMalloc buffers: this process's levels
    const size_t vec = (size_t)n * 8;
    {
        std::vector<double *> b(8);
        for (auto &p : b) { CK(hipMalloc(&p, vec)); CK(hipMemset(p, 0, vec)); }
        printf("== hipMalloc triples (x, u, D) at n = %lld:", n);
        double best = 1e30, worst = 0;
        for (int t = 0; t < 12; ++t) {
            const int i = (t * 3) % 8, j = (t * 3 + 1 + t / 3) % 8, k = (t * 5 + 2) % 8;
            if (i == j || j == k || i == k) continue;
            const double us = time_mix(b[i], b[j], b[k], n);
            printf(" %.0f", us); best = std::min(best, us); worst = std::max(worst, us);
        }
        printf("  → best %.1f, worst %.1f us\n", best, worst);
        for (auto p : b) CK(hipFree(p));
    }

    // (3) physical chunks, timed as triples at three fixed virtual ranges
    std::vector<hipMemGenericAllocationHandle_t> h(pool);
    for (int i = 0; i < pool; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    void *va[3];
    for (int v = 0; v < 3; ++v) CK(hipMemAddressReserve(&va[v], chunk, gran_rec, nullptr, 0));
    const long long nc = (long long)(chunk / 8);
    auto map3 = [&](int a, int b, int c) {
        const int ids[3] = {a, b, c};
        for (int v = 0; v < 3; ++v) { CK(hipMemMap(va[v], chunk, 0, h[ids[v]], 0)); CK(hipMemSetAccess(va[v], chunk, &acc, 1)); }
    };
    auto unmap3 = [&]() { for (int v = 0; v < 3; ++v) CK(hipMemUnmap(va[v], chunk)); };
    unsigned long long lcg = 12345;
    auto rnd = [&](int m) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (int)((lcg >> 33) % (unsigned)m); };
    struct T { int a, b, c; double us; };
    std::vector<T> tri;
    for (int t = 0; t < 120; ++t) {
        int a = rnd(pool), b = rnd(pool), c = rnd(pool);
        if (a == b || b == c || a == c) continue;
        map3(a, b, c);
        if (t == 0) { CK(hipMemset(va[0], 0, chunk)); CK(hipMemset(va[1], 0, chunk)); CK(hipMemset(va[2], 0, chunk)); }
        tri.push_back({a, b, c, time_mix((double *)va[0], (double *)va[1], (const double *)va[2], nc, 7)});
        unmap3();
    }
    std::sort(tri.begin(), tri.end(), [](const T &p, const T &q) { return p.us < q.us; });
    printf("== %zu random chunk triples of %zu MiB (bytes moved %.0f MB each): fastest %.1f, median %.1f, slowest %.1f us; deciles:", tri.size(), chunk >> 20,
           40.0 * nc / 1e6, tri.front().us, tri[tri.size() / 2].us, tri.back().us);
    for (int q = 0; q <= 10; ++q) printf(" %.1f", tri[std::min(tri.size() - 1, tri.size() * q / 10)].us);
    printf("\n");
    // the same triple again, and with its roles permuted: is the time a property of the triple?
    for (int r = 0; r < 3; ++r) {
        const T &f = tri[r], &s = tri[tri.size() - 1 - r];
        map3(f.a, f.b, f.c); const double f2 = time_mix((double *)va[0], (double *)va[1], (const double *)va[2], nc, 7); unmap3();
        map3(f.b, f.c, f.a); const double f3 = time_mix((double *)va[0], (double *)va[1], (const double *)va[2], nc, 7); unmap3();
        map3(s.a, s.b, s.c); const double s2 = time_mix((double *)va[0], (double *)va[1], (const double *)va[2], nc, 7); unmap3();
        printf("   fast #%d (%d,%d,%d): %.1f → again %.1f, rotated %.1f | slow #%d (%d,%d,%d): %.1f → again %.1f\n", r, f.a, f.b, f.c, f.us, f2, f3, r, s.a, s.b, s.c, s.us, s2);
    }

    // (4) full-size vectors from chunks
    const int per = (int)((vec + chunk - 1) / chunk);
    if (3 * per > pool) { printf("pool too small for full-size vectors (%d chunks needed)\n", 3 * per); return 0; }
    void *full[3];
    for (int v = 0; v < 3; ++v) CK(hipMemAddressReserve(&full[v], (size_t)per * chunk, gran_rec, nullptr, 0));
    auto map_full = [&](const std::vector<int> &ids) {   // ids[v * per + i]
        for (int v = 0; v < 3; ++v) {
            for (int i = 0; i < per; ++i) CK(hipMemMap((char *)full[v] + (size_t)i * chunk, chunk, 0, h[ids[v * per + i]], 0));
            CK(hipMemSetAccess(full[v], (size_t)per * chunk, &acc, 1));
        }
    };
    auto unmap_full = [&]() { for (int v = 0; v < 3; ++v) CK(hipMemUnmap(full[v], (size_t)per * chunk)); };
    {   // (a) creation order
        std::vector<int> ids(3 * per);
        for (int i = 0; i < 3 * per; ++i) ids[i] = i;
        map_full(ids);
        printf("== full-size vectors (%d chunks each) from chunks in creation order: %.1f us", per, time_mix((double *)full[0], (double *)full[1], (const double *)full[2], n));
        unmap_full();
        for (int i = 0; i < 3 * per; ++i) ids[i] = (i % 3) * per + i / 3;   // interleaved creation order
        map_full(ids);
        printf("; interleaved creation order: %.1f us\n", time_mix((double *)full[0], (double *)full[1], (const double *)full[2], n));
        unmap_full();
    }
    {   // (b) per position: the fastest triple among candidates drawn from the unused chunks
        std::vector<char> used(pool, 0);
        std::vector<int> ids(3 * per);
        double sum = 0;
        for (int i = 0; i < per; ++i) {
            T best = {-1, -1, -1, 1e30};
            for (int t = 0; t < 24; ++t) {
                int a = rnd(pool), b = rnd(pool), c = rnd(pool);
                if (a == b || b == c || a == c || used[a] || used[b] || used[c]) continue;
                map3(a, b, c);
                const double us = time_mix((double *)va[0], (double *)va[1], (const double *)va[2], nc, 3);
                unmap3();
                if (us < best.us) best = {a, b, c, us};
                if (us <= tri.front().us * 1.03) break;   // as good as the fastest seen
            }
            if (best.a < 0) { printf("ran out of chunks at position %d\n", i); return 0; }
            used[best.a] = used[best.b] = used[best.c] = 1;
            ids[0 * per + i] = best.a; ids[1 * per + i] = best.b; ids[2 * per + i] = best.c;
            sum += best.us;
        }
        map_full(ids);
        const double us = time_mix((double *)full[0], (double *)full[1], (const double *)full[2], n);
        printf("== full-size vectors from per-position fastest triples (Σ of their chunk times %.1f us): %.1f us = %.2f TB/s (%.1f%% of 8 TB/s)\n", sum, us,
               40.0 * n / us / 1e6, 40.0 * n / us / 1e6 / 8 * 100);
        unmap_full();
    }
    return 0;
}
