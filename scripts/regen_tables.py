#!/usr/bin/env python3
"""Regenerates the MEASURED tables of DESIGN.md §4 and BASELINE.md §3 from the artefacts under profiles/ (VERDICT r02
next #8: "regenerate the prose from the artefacts").

    python3 scripts/regen_tables.py            # rewrite the blocks between the GENERATED markers in place
    python3 scripts/regen_tables.py --check    # exit 1 if a file's block differs from what the artefacts give

Every number in the block comes from a committed file: profiles/<tag>_bench_*.json (bench.py lines: HIP-event kernel
times, it/s, CPU baselines, library build id) and profiles/<tag>_*rocprofv3_kernel_stats.csv (rocprofv3 --kernel-trace
--stats of the same command).  tests/test_docs_consistency.py runs the check in the CPU tier, so a hand-edited or stale
number fails the suite.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r04"
BEGIN, END = f"<!-- GENERATED:{TAG}-measurements BEGIN (scripts/regen_tables.py — do not edit by hand) -->", f"<!-- GENERATED:{TAG}-measurements END -->"

ROWS = [  # (key, bench-line stem, rocprofv3 stem or None, label)
    ("c1", "bench_c1", "c1", "1: paired Rosenbrock n = 1000, PR-CG, strong Wolfe c2 = 0.1 (15 steps after 3)"),
    ("c1c", "bench_c1c", None, "1 (chained form): stencil objective, n = 1000"),
    ("c2", "bench_c2", "c2", "2: quadratic n = 1e6, PR-CG"),
    ("c3", "bench_c3", "c3", "3: extended Rosenbrock n = 1e7, HZ + WolfeBisection — 160 MB working set: **Infinity-Cache-assisted**"),
    ("c3big", "bench_c3big", "c3big", "3 at n = 4e7 (640 MB working set: cannot sit in the 256 MiB Infinity Cache)"),
    ("c4", "bench_c4", "c4", "4: log-sum-exp n = 1e7, L-BFGS m = 10 (1.76 GB ring)"),
    ("c5", "bench_c5", "c5", "5: quadratic n = 1e8, PR-CG, one GPU (headline; placement search on)"),
    ("c5n", "bench_c5_nosearch", "c5_nosearch", "5 without the placement search (`--no-placement-search`: buffers as allocated)"),
    ("shard", "bench_shard", "shard", "5's 8-GPU shard on one GPU: n = 1.25e7 — 300 MB working set: partly **Infinity-Cache-assisted**"),
    ("c1h", "bench_c1_hostdriven", None, "1 with CGO_RESIDENT=0 (a launch per trial)"),
    ("c2h", "bench_c2_hostdriven", None, "2 with CGO_RESIDENT=0 (a launch per trial)"),
    ("c4t", "bench_c4_twopass", None, "4 with CGO_LBFGS_SPEC=0 (two passes over the ring per iteration, fused push)"),
]


def load(stem):
    p = os.path.join(ROOT, "profiles", f"{TAG}_{stem}.json")
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p))
    except Exception:
        return None


def stats_csv(name):
    """rocprofv3 kernel stats → {short symbol: (calls, avg_us)}"""
    p = os.path.join(ROOT, "profiles", f"{TAG}_{name}rocprofv3_kernel_stats.csv")
    out = {}
    if not os.path.exists(p):
        return out
    for row in csv.DictReader(open(p)):
        sym = row["Name"].split("(")[0].replace("void ", "").replace("cgo::dev::", "").replace("cgo::", "").strip()
        out[sym] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    return out


def fmt(v, nd=0):
    if v is None:
        return "—"
    if nd == 0:
        return f"{v:,.0f}".replace(",", " ")
    return f"{v:,.{nd}f}".replace(",", " ")


def prof_row(stem):
    """rocprofv3 --kernel-trace --stats of a bench.py command + the line that very process printed → (symbol, calls, avg µs, line)"""
    p = os.path.join(ROOT, "profiles", f"{TAG}_{stem}_rocprofv3_bench_line.json")
    if not os.path.exists(p):
        return None
    line = json.load(open(p))
    st = stats_csv(stem + "_")
    sym = line["roofline"]["kernel"]
    if sym not in st:
        return None
    calls, avg = st[sym]
    return sym, calls, avg, line


def render():
    lines = [BEGIN, ""]
    heads = [load(stem) for _, stem, _, _ in ROWS]
    builds = sorted({d.get("library_build_id") for d in heads if d})
    lines.append(f"Library build(s) measured: {', '.join('`%s`' % b for b in builds if b)}; one MI355X per row; `value` = the first timed window, "
                 "median over the windows in brackets; **kernel µs (events)** = HIP events inside the un-profiled `bench.py` run (every launch from n = 3e7, every 4th below); "
                 "**kernel µs (rocprofv3)** = average duration in `profiles/" + TAG + "_<config>_rocprofv3_kernel_stats.csv` of the same command in another process "
                 "(its own bench line beside it: `…_rocprofv3_bench_line.json`); **% of 8 TB/s** = algorithmic bytes per launch ÷ the rocprofv3 duration ÷ 8 TB/s "
                 "(÷ the event duration in brackets); **PMC traffic** = HBM bytes per launch from separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes of the same build "
                 "(`profiles/" + TAG + "_pmc_summary.json`; FETCH_SIZE doubled per the guide's gfx950 correction) ÷ algorithmic bytes; "
                 "CPU = `oracle/cgo_oracle.c` on the SAME workload AT ITS FULL SIZE on the GPU box's host (iterations w+1 … w+k of one run, timed inside the oracle; "
                 "1 thread / the OpenMP build on the container's cores, never below the 1-thread figure).")
    lines.append("")
    lines.append("| config | it/s first window [median] | trials / launches per iteration | dominant kernel | kernel µs events / rocprofv3 (calls) | algorithmic B per launch | % of 8 TB/s by rocprofv3 [by events] | PMC traffic ÷ algorithmic | CPU 1 thread it/s | CPU all cores it/s (cores) |")
    lines.append("|---|---|---|---|---|---|---|---|---|---|")
    for (key, stem, pstem, label), d in zip(ROWS, heads):
        if not d:
            continue
        rf = d.get("roofline", {})
        cfg = d.get("config", {})
        cb, ca = d.get("cpu_baseline") or {}, d.get("cpu_baseline_all_cores") or {}
        pr = prof_row(pstem) if pstem else None
        hbm = key in ("c3", "c3big", "c4", "c4t", "c5", "c5n", "shard")   # HBM-fraction claims only where the launch streams more than the caches hold
        byt = rf.get("algorithmic_bytes_per_launch") or 0
        ev = rf.get("avg_launch_us")
        rp = f"{fmt(pr[2], 1)} ({pr[1]})" if pr and pr[0] == rf.get("kernel") else "—"
        frac_rp = f"**{100 * byt / pr[2] / 1e3 / 8000:.1f} %**" if (hbm and pr and pr[0] == rf.get("kernel") and byt) else None
        frac_ev = f"{100 * rf.get('frac', 0):.1f} %" if hbm and rf.get("frac") else None
        frac = (f"{frac_rp} [{frac_ev}]" if frac_rp else (f"[{frac_ev}]" if frac_ev else "— (latency-bound)"))
        tr = f"{rf['traffic'] / byt:.4f}" if (rf.get("traffic") and byt) else "—"
        lvl = (d.get("placement") or {}).get("level")
        lab = label + (f" — placement level **{lvl}** ({(d.get('placement') or {}).get('candidates')} candidates)" if lvl else "")
        lines.append(f"| {lab} | **{fmt(d['value'])}** [{fmt(d.get('value_median'))}] | {fmt(cfg.get('trials_per_iteration'), 2)} / {fmt(cfg.get('launches_per_iteration'), 2)} | "
                     f"`{rf.get('kernel', '')}` | {fmt(ev, 1)} / {rp} | {fmt(byt / 1e6, 1) + ' MB' if hbm and byt else '—'} | {frac} | {tr} | "
                     f"{fmt(cb.get('value'), 2) if cb.get('value') and cb['value'] < 100 else fmt(cb.get('value'))} | "
                     f"{(fmt(ca.get('value'), 2) if ca.get('value') and ca['value'] < 100 else fmt(ca.get('value')))} ({ca.get('cores', '—')}) |")
    lines.append("")
    # the headline kernel at BOTH placement levels: each a rocprofv3 CSV of one process with that process's own line (VERDICT r03 next #1a)
    for stem, what in (("c5", "with the placement search"), ("c5_searchmiss", "with the placement search in a process that holds no fast triple (box-dependent: 3 of 8 processes in round 3's sampling, 0 of 16 in round 4's)"),
                       ("c5_nosearch", "without it (buffers as allocated)")):
        pr = prof_row(stem)
        if not pr:
            continue
        sym, calls, avg, line = pr
        byt = line["roofline"]["algorithmic_bytes_per_launch"]
        pl = line.get("placement")
        lines.append(f"Headline kernel `{sym}` {what}: **{fmt(avg, 1)} µs average over {calls} calls** in `profiles/{TAG}_{stem}_rocprofv3_kernel_stats.csv` "
                     f"= {fmt(byt / avg / 1e3)} GB/s = **{byt / avg / 1e3 / 8000:.3f}** of 8 TB/s; the same process's HIP events: {fmt(line['roofline']['avg_launch_us'], 1)} µs "
                     f"({line['roofline']['frac']:.3f}); " + (f"placement search: {pl['candidates']} candidates, bare mix {fmt(pl['mix_as_allocated_us'], 1)} µs as allocated → "
                                                              f"{fmt(pl['mix_chosen_us'], 1)} µs chosen ({pl['mix_chosen_tbps']:.2f} TB/s: level **{pl['level']}**)" if pl else "no search") +
                     f"; PMC traffic {fmt((line['roofline'].get('traffic') or 0) / 1e9, 4)} GB vs {fmt(byt / 1e9, 4)} GB algorithmic.")
        lines.append("")
    for name, label in (("c1_", "config 1"), ("c2_", "config 2"), ("c3_", "config 3"), ("c3big_", "config 3 at n = 4e7"), ("c4_", "config 4"), ("shard_", "the 8-GPU shard size")):
        st = stats_csv(name)
        if not st:
            continue
        top = sorted(((k, v) for k, v in st.items() if k.startswith("k_")), key=lambda kv: -kv[1][0] * kv[1][1])[:5]
        lines.append(f"rocprofv3, {label} (`profiles/{TAG}_{name}rocprofv3_kernel_stats.csv`): " +
                     "; ".join(f"`{k}` {fmt(v[1], 1)} µs × {v[0]}" for k, v in top) + ".")
    lines.append("")
    g = load("gaps_fused_shard_n1p25e7")
    if g:
        ks = g.get("kernels", {})
        dom = max(ks.items(), key=lambda kv: kv[1]["total_us"]) if ks else None
        gap = g["gaps"][0] if g.get("gaps") else None
        if dom and gap:
            lines.append(f"Kernel-trace gap table at the 8-GPU shard size (`profiles/{TAG}_gaps_fused_shard_n1p25e7.json`, events off): `{dom[0]}` "
                         f"{fmt(dom[1]['avg_us'], 1)} µs average ({dom[1]['calls']} calls) + {fmt(gap['median_us'], 1)} µs median gap "
                         f"= {fmt(dom[1]['avg_us'] + gap['median_us'], 1)} µs per iteration WITH rocprofv3 attached; un-profiled the same run takes "
                         f"{fmt(1e6 / (load('bench_shard') or {}).get('value_median', float('nan')), 1)} µs per iteration (`profiles/{TAG}_bench_shard.json`, median window).")
            lines.append("")
    lh = []
    for c in ("c2", "c3", "c4", "c5"):
        d = load(f"long_horizon_{c}")
        if not d:
            continue
        g_, f_ = d.get("gpu", {}), d.get("_facts", {})
        xs = g_.get("x_rel_err", {})
        last = sorted(xs.items(), key=lambda kv: int(kv[0]))[-1] if xs else None
        refs = [d[k]["x_rel_err"].get(last[0]) for k in ("omp", "c") if k in d and last and d[k]["x_rel_err"].get(last[0]) is not None]
        lh.append(f"{c} (n = {f_.get('n', 0):.0e}, {f_.get('iters')} iterations): GPU on the arbiter's step sequence for {g_.get('iterations_in_common')} iterations"
                  + (f" (parted where the arbiter's margin was {g_['arbiter_margin_where_parted']:.2g})" if g_.get("parted") else " (never parted)")
                  + (f", iterate error at iteration {last[0]} {last[1]:.1e} vs the double oracle(s) {', '.join('%.1e' % r for r in refs)}" if last else ""))
    if lh:
        lines.append("Long-horizon parity against the arbiter (`profiles/" + TAG + "_long_horizon_c*.json`, `tests/test_long_horizon.py`): " + "; ".join(lh) + ".")
        lines.append("")
    lines.append(END)
    return "\n".join(lines)


def apply(path, block, check):
    p = os.path.join(ROOT, path)
    txt = open(p).read()
    if BEGIN not in txt or END not in txt:
        print(f"{path}: markers missing", file=sys.stderr)
        return False
    new = txt[:txt.index(BEGIN)] + block + txt[txt.index(END) + len(END):]
    if check:
        return new == txt
    if new != txt:
        open(p, "w").write(new)
    return True


def main():
    check = "--check" in sys.argv
    block = render()
    ok = all(apply(f, block, check) for f in ("DESIGN.md", "BASELINE.md"))
    if check and not ok:
        print("the generated measurement block is stale: run python3 scripts/regen_tables.py", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
