// cgo_backend_place.hip — WHERE x, u and D live decides how fast the accept+dir+trial mix runs: the buffer placement search
// (opt-in: cgo_solver_policy.placement_search) and the bare stream-mix harness it is priced against.
#include "cgo_backend_internal.hpp"

#include "cgo_kernels.hip.hpp"
#include "cgo_kernels_cg.hip.hpp"

namespace cgo {

using namespace dev;

// BIG launches of the k_cg family can write x / u out of place when a second pair of buffers fits beside the state.
// OFF unless CGO_PINGPONG=1: the no-arithmetic harness showed out-of-place 10 % ahead on one MI355X (650 vs 720 µs)
// and level on another (717 vs 719 µs), and the engine's own launch gained nothing on either (697 vs 719, 681 vs
// 683 µs; gpurun_out/r02_ab, r02_misc) — not worth 16 B/element of HBM.  Decided once, at the first such launch.
bool HipBackend::pingpong_ready() {
    if (pingpong_ >= 0) return pingpong_ == 1;
    pingpong_ = 0;
    const char *e = getenv("CGO_PINGPONG");
    if (!e || e[0] != '1') return false;
    if (!rmode_ || sys_on_) return false;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    const size_t need = 2 * sizeof(double) * (size_t)obj_->n_local;
    if (fr < need + (size_t(2) << 30)) return false;   // keep 2 GiB of headroom for ga_/gb_ and the caller
    if (x2_.alloc((size_t)obj_->n_local) != CGO_OK) { (void)hipGetLastError(); return false; }
    if (u2_.alloc((size_t)obj_->n_local) != CGO_OK) { (void)hipGetLastError(); x2_.release(); return false; }
    xalt_ = (xc_ == x_.p) ? x2_.p : x_.p;
    ualt_ = (uc_ == u_.p) ? u2_.p : u_.p;
    pingpong_ = 1;
    return true;
}

// WHERE x, u and D live decides how fast the accept+dir+trial mix runs.  On every MI355X box sampled (five), the same
// no-arithmetic kernel (k_stream_mix: R x,u,D / W x,u in place, n = 1e8) takes ≈ 635 µs on some triples of separately
// allocated buffers and 740–770 µs on others — stable per triple, three levels (≈ 640 / 715 / 755), no rule in the
// virtual addresses, a spacer between the allocations does not help (scripts/tune/rw_mix.hip "place", "spacer", "arena";
// profiles/r02_placement_*.log): DRAM channel/bank conflicts between the physical pages the allocator happened to hand
// out.  The buffers a solver gets by plain consecutive hipMallocs are usually a slow triple (engine launch 750–775 µs).
// So for pure-HBM problem sizes the solver allocates a few spare buffers, times the bare mix on the ordered pairs
// (x, u) of the pool with D where it is, then on the best pair with D moved into each remaining buffer, keeps the
// fastest triple (D is copied once, device to device) and frees the rest: ≈ 80 ms once per solver at n = 1e8, paid back
// within a few hundred iterations.  CGO_PLACE_TUNE=0 switches it off; skipped when the spare buffers do not fit.
int HipBackend::tune_placement() {
    const bool on = pol_.placement_search != 0;
    const int64_t n = obj_->n_local;
    const bool hp = obj_->uses_param();
    static const bool dbg = getenv("CGO_DEBUG_PLACE") != nullptr;
    if (dbg) fprintf(stderr, "[cgo place] on=%d rmode=%d chain=%d bytes=%.3g big=%.3g\n", (int)on, (int)rmode_, (int)chain(),
                     bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, n, hp), big_bytes(false));
    // Searched for the pure-HBM (BIG) launches only.  Round 3 measured the grid-stride launches that exceed the 256 MiB
    // Infinity Cache as well (CGO_PLACE_MIN_BYTES=2.7e8; the 8-GPU shard of config 5, n/8 = 1.25e7, and config 3 at
    // n = 1e7): there the bare mix on the launch's own policy differs by 3–5 % between triples, with no
    // "level" among 134–192 candidates (65.2 → 61.8 µs, 67.5 → 64.3 µs; config 3: 40.4 → 38.9 µs), and the engine's launch
    // does not move at all (87.0 vs 87.2 µs, 53.7 vs 54.6 µs; scripts/r03_shard.sh) — at those sizes the launch is 20 µs above
    // its own mix for other reasons (two waves per SIMD do not hide the FP64 work behind the stream).  Not worth 24 spare
    // buffers and 30–40 ms per solver: off by default below the BIG threshold.
    const double launch_bytes = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, n, hp);
    static const double min_env = [] { const char *e = getenv("CGO_PLACE_MIN_BYTES"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.0; }();   // (experiments: the search below the pure-HBM threshold)
    const double min_bytes = min_env > 0.0 ? min_env : big_bytes(false);
    const bool big = launch_bytes > big_bytes(false);
    const int mix_grid = big ? GRID_BIG : grid_cg(n, policy_points());
    if (!on || !rmode_ || chain() || launch_bytes <= min_bytes) return CGO_OK;
    HIPCHK(hipSetDevice(ctx_->device));
    if (ctx_->placed_n == n && ctx_->placed_x.p && ctx_->placed_u.p) {   // an earlier solver of this size already searched
        x_.release(); u_.release();
        std::swap(x_.p, ctx_->placed_x.p); std::swap(x_.n, ctx_->placed_x.n);
        std::swap(u_.p, ctx_->placed_u.p); std::swap(u_.n, ctx_->placed_u.n);
        place_first_us_ = ctx_->placed_first_us; place_best_us_ = ctx_->placed_best_us; place_candidates_ = ctx_->placed_candidates;
        placed_ = true;
        return CGO_OK;
    }
    if (ctx_->placed_x.p || ctx_->placed_u.p) {   // parked buffers of another size: give them back before searching
        ctx_->placed_x.release(); ctx_->placed_u.release(); ctx_->placed_n = 0;
    }
    // Spare buffers come in stages of eight, up to three stages (CGO_PLACE_STAGES): whether a process's allocations hold a
    // fast triple at all is a matter of luck — on one box three processes of four found none among 64 candidates from
    // x, u, D + 8 spares, the fourth at its 8th candidate (gpurun_out/r02_fin1) — and new allocations made while the old
    // ones are held land on other physical pages.
    constexpr int STAGE = 8, SPARE = 3 * STAGE, PER_STAGE = 64;
    const int stages = (pol_.placement_stages >= 1 && pol_.placement_stages <= 3) ? pol_.placement_stages : 3;
    const size_t vec = (size_t)n * sizeof(double);
    hipStream_t st = ctx_->stream;
    DevBuf spare[SPARE];
    int have = 0;
    std::vector<double *> pool = {x_.p, u_.p};
    // The search's TRANSIENT memory is capped: policy.placement_max_bytes, or — library policy — a quarter of what is free now
    // (never more than the 24 vectors of three stages).  Below one stage's worth it does not run.
    int max_spares = SPARE;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return CGO_OK; }
        const double cap = pol_.placement_max_bytes > 0 ? (double)pol_.placement_max_bytes : 0.25 * (double)fr;
        max_spares = (int)std::min<double>((double)SPARE, cap / (double)vec);
        place_cap_bytes_ = (double)max_spares * (double)vec;
        if (max_spares < STAGE) { if (dbg) fprintf(stderr, "[cgo place] memory cap %.3g B < one stage of spares: no search\n", cap); return CGO_OK; }
    }
    auto grow = [&]() -> int {   // one more stage of spares, as far as memory allows (ga_/gb_ and the caller need room too)
        int added = 0;
        while (have < SPARE && have < max_spares && added < STAGE) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); break; }
            if (fr < 2 * vec + (size_t(4) << 30)) break;
            if (spare[have].alloc((size_t)n) != CGO_OK) { (void)hipGetLastError(); break; }
            if (hipMemsetAsync(spare[have].p, 0, vec, st) != hipSuccess) { (void)hipGetLastError(); break; }
            pool.push_back(spare[have].p);
            ++have; ++added;
        }
        if (dbg) fprintf(stderr, "[cgo place] +%d spare buffers (%d in the pool)\n", added, (int)pool.size());
        return added;
    };
    if (grow() < 2) return CGO_OK;
    HIPCHK(hipMemsetAsync(x_.p, 0, vec, st));
    HIPCHK(hipMemsetAsync(u_.p, 0, vec, st));
    auto time_mix = [&](double *x, double *u, const double *d, double &us) -> int {
        float t[2];
        for (int r = -1; r < 2; ++r) {
            if (r >= 0) HIPCHK(hipEventRecord(ctx_->ev0, st));
            if (big) {
                if (hp) k_stream_mix<true, true><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
                else k_stream_mix<false, true><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
            } else {   // the streaming policy the engine's launch will use at this size
                if (hp) k_stream_mix<true, false><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
                else k_stream_mix<false, false><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
            }
            if (r >= 0) {
                HIPCHK(hipEventRecord(ctx_->ev1, st));
                HIPCHK(hipStreamSynchronize(st));
                HIPCHK(hipEventElapsedTime(&t[r], ctx_->ev0, ctx_->ev1));
            }
        }
        HIPCHK(hipGetLastError());
        us = (double)std::min(t[0], t[1]) * 1e3;
        return CGO_OK;
    };
    const double *d0 = hp ? obj_->p0.p : nullptr;
    double best = 0.0, first = 0.0, worst = 0.0;
    int bx = 0, bu = 1, bd = -1;   // bd = −1: D stays where it is
    // The times come in levels ≈ 10–15 % apart (≈ 640 / 715 / 755 µs at n = 1e8 — none, one, several of the three streams
    // in conflict): stop as soon as a triple sits a level below the slowest seen.  Triples (x, u, D) are drawn from the pool
    // in a fixed pseudo-random order (D may stay where it is or move into a pool buffer); at most 64 per stage are timed.
    if (int rc = time_mix(pool[0], pool[1], d0, first)) return rc;
    best = worst = first; place_candidates_ = 1;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    auto next = [&](int m) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (int)((lcg >> 33) % (unsigned)m); };
    static const double ratio = [] { const char *e = getenv("CGO_PLACE_RATIO"); double v = e ? atof(e) : 0.0; return (v > 0.0 && v < 1.0) ? v : 0.88; }();
    auto found = [&] { return place_candidates_ >= 4 && best <= ratio * worst; };
    // Round 3: "a level below the slowest seen" used to end the search at the MIDDLE level too (667–670 µs at n = 1e8: 3 of 8
    // fresh processes in profiles/r03_headline_samples.txt stopped there after 5–65 candidates, 4 reached 643–646 µs).  The
    // levels are physical — 6.2 / 6.0 / 5.3 TB/s of the five-stream mix on every box sampled — so the top one has an absolute
    // mark: inside a stage the search now goes on until a triple streams at ≥ 6.1 TB/s (or the stage's 64 candidates are
    // used up: ≈ 50 ms at n = 1e8); further stages of spares are still added only while not even the middle level is in hand.
    static const double fast_tbps = [] { const char *e = getenv("CGO_PLACE_FAST_TBPS"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 6.1; }();
    const double fast_us = (big && hp) ? 40.0 * (double)n / (fast_tbps * 1e12) * 1e6 : 0.0;
    auto done = [&] { return fast_us > 0.0 ? (place_candidates_ >= 2 && best <= fast_us) : found(); };
    for (int stage = 0; stage < stages && !found() && !done(); ++stage) {
        if (stage > 0 && grow() == 0) break;
        const int P = (int)pool.size();
        for (int it = 0; it < PER_STAGE - (stage == 0 ? 1 : 0) && !done(); ++it) {
            const int i = next(P);
            int j = next(P - 1); if (j >= i) ++j;
            int k = -1;
            // (D moves only while this solver is the objective's only user)
            if (hp && obj_->users <= 1 && next(4) != 0) { k = next(P - 2); const int lo = std::min(i, j), hi2 = std::max(i, j); if (k >= lo) ++k; if (k >= hi2) ++k; }
            double us = 0.0;
            if (int rc = time_mix(pool[i], pool[j], k >= 0 ? pool[k] : d0, us)) return rc;
            place_candidates_++;
            if (us < best) { best = us; bx = i; bu = j; bd = k; }
            if (us > worst) worst = us;
        }
    }
    place_first_us_ = first; place_best_us_ = best;
    if (dbg) fprintf(stderr, "[cgo place] %d candidates: as allocated %.1f us, best %.1f us (x=%d u=%d d=%d)\n", place_candidates_, first, best, bx, bu, bd);
    // hand the chosen buffers to x_, u_ (and the objective's parameter vector); everything else is released
    auto owner = [&](double *p) -> DevBuf * {
        if (p == x_.p) return &x_;
        if (p == u_.p) return &u_;
        for (auto &sb : spare) if (sb.p == p) return &sb;
        return nullptr;
    };
    double *px = pool[bx], *pu = pool[bu], *pd = bd >= 0 ? pool[bd] : nullptr;
    if (pd) {
        HIPCHK(hipMemcpyAsync(pd, obj_->p0.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        DevBuf *o = owner(pd);
        std::swap(o->p, obj_->p0.p); std::swap(o->n, obj_->p0.n);
    }
    if (px != x_.p) { DevBuf *o = owner(px); std::swap(o->p, x_.p); std::swap(o->n, x_.n); }
    if (pu != u_.p) { DevBuf *o = owner(pu); std::swap(o->p, u_.p); std::swap(o->n, u_.n); }
    HIPCHK(hipStreamSynchronize(st));
    placed_ = true;
    return CGO_OK;   // the spare DevBufs (now holding the rejected buffers) free themselves here
}

// The placement search, after the C API has set the launch policy (its stream mix runs on the grid the solver's launches will use).
int HipBackend::place() {
    if (int rc = tune_placement()) return rc;
    xc_ = x_.p; uc_ = u_.p;
    return CGO_OK;
}

// The read/write mix of the dominant launch without its arithmetic: median and best of `reps` launches (HIP events).
int HipBackend::bench_stream_mix(HipCtx *ctx, int64_t n, int reps, double *median_us, double *best_us) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n < 2 || reps < 1 || reps > 1000) { set_error("bench_stream_mix: n ≥ 2 and 1 ≤ reps ≤ 1000 required"); return CGO_EINVAL; }
    DevBuf x, u, d;
    if (int rc = x.alloc((size_t)n)) return rc;
    if (int rc = u.alloc((size_t)n)) return rc;
    if (int rc = d.alloc((size_t)n)) return rc;
    const int fg = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    hipStream_t st = ctx->stream;
    k_fill<<<fg, BLOCK, 0, st>>>(x.p, n, 0, 1, 1, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, st>>>(u.p, n, 0, 1, 2, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, st>>>(d.p, n, 0, 1, 3, 1.0, 10.0);
    std::vector<float> t((size_t)reps);
    for (int r = -2; r < reps; ++r) {
        if (r >= 0) HIPCHK(hipEventRecord(ctx->ev0, st));
        k_stream_mix<true><<<GRID_BIG, BLOCK, 0, st>>>(x.p, u.p, d.p, n, 1e-9, 0.5);
        if (r >= 0) {
            HIPCHK(hipEventRecord(ctx->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipEventElapsedTime(&t[(size_t)r], ctx->ev0, ctx->ev1));
        }
    }
    HIPCHK(hipGetLastError());
    std::sort(t.begin(), t.end());
    *median_us = (double)t[(size_t)reps / 2] * 1e3;
    *best_us = (double)t[0] * 1e3;
    return CGO_OK;
}

}  // namespace cgo
