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
TAG = "r03"
BEGIN, END = f"<!-- GENERATED:{TAG}-measurements BEGIN (scripts/regen_tables.py — do not edit by hand) -->", f"<!-- GENERATED:{TAG}-measurements END -->"

ROWS = [  # (key, profiles file stem, label)
    ("c1", "bench_c1", "1: paired Rosenbrock n = 1000, PR-CG, strong Wolfe c2 = 0.1 (15 steps after 3)"),
    ("c1c", "bench_c1c", "1 (chained form): stencil objective, n = 1000"),
    ("c2", "bench_c2", "2: quadratic n = 1e6, PR-CG"),
    ("c3", "bench_c3", "3: extended Rosenbrock n = 1e7, HZ + WolfeBisection"),
    ("c4", "bench_c4", "4: log-sum-exp n = 1e7, L-BFGS m = 10"),
    ("c5", "bench_n1", "5: quadratic n = 1e8, PR-CG, one GPU (headline)"),
    ("shard", "bench_shard_n1p25e7", "5's 8-GPU shard on one GPU: n = 1.25e7"),
    ("c1h", "bench_c1_hostdriven", "1 with CGO_RESIDENT=0 (a launch per trial, as in round 2)"),
    ("c2h", "bench_c2_hostdriven", "2 with CGO_RESIDENT=0 (a launch per trial, as in round 2)"),
    ("c4t", "bench_c4_twopass", "4 with CGO_LBFGS_SPEC=0 (two passes over the ring per iteration, fused push)"),
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


def render():
    lines = [BEGIN, ""]
    heads = [load(stem) for _, stem, _ in ROWS]
    builds = sorted({d.get("library_build_id") for d in heads if d})
    lines.append(f"Library build(s) measured: {', '.join('`%s`' % b for b in builds if b)}; one MI355X per row; `value` = the first timed window, "
                 "median over the windows in brackets; kernel times = HIP events inside `bench.py` (every launch from n = 3e7, every 4th below); "
                 "CPU = `oracle/cgo_oracle.c` on the SAME workload on the GPU box's host (1 thread / all cores of the container's share).")
    lines.append("")
    lines.append("| config | it/s first window [median] | trials / launches per iteration | dominant kernel (bench events) | algorithmic GB/s | % of 8 TB/s | CPU 1 thread it/s | CPU all cores it/s (cores) |")
    lines.append("|---|---|---|---|---|---|---|---|")
    for (key, stem, label), d in zip(ROWS, heads):
        if not d:
            continue
        rf = d.get("roofline", {})
        cfg = d.get("config", {})
        cb, ca = d.get("cpu_baseline") or {}, d.get("cpu_baseline_all_cores") or {}
        kern = f"`{rf.get('kernel', '')}` {fmt(rf.get('avg_launch_us'), 1)} µs"
        hbm = key in ("c3", "c4", "c4t", "c5", "shard")   # HBM-fraction claims only where the working set leaves the Infinity Cache
        lines.append(f"| {label} | **{fmt(d['value'])}** [{fmt(d.get('value_median'))}] | {fmt(cfg.get('trials_per_iteration'), 2)} / {fmt(cfg.get('launches_per_iteration'), 2)} | {kern} | "
                     f"{fmt(rf.get('achieved')) if hbm else '—'} | {fmt(100 * rf.get('frac', 0), 1) + ' %' if hbm and rf.get('frac') else '— (latency-bound)'} | "
                     f"{fmt(cb.get('value'), 2) if cb.get('value') and cb['value'] < 100 else fmt(cb.get('value'))} | "
                     f"{(fmt(ca.get('value'), 2) if ca.get('value') and ca['value'] < 100 else fmt(ca.get('value')))} ({ca.get('cores', '—')}) |")
    lines.append("")
    # the dominant kernel of the headline run: bench events vs the rocprofv3 CSV of the same command
    h = load("bench_n1")
    st = stats_csv("")
    if h and st:
        sym = h["roofline"]["kernel"]
        if sym in st:
            calls, avg = st[sym]
            byt = h["roofline"]["algorithmic_bytes_per_launch"]
            lines.append(f"Headline kernel `{sym}`: {fmt(h['roofline']['avg_launch_us'], 1)} µs by HIP events in `bench.py` (`profiles/{TAG}_bench_n1.json`: "
                         f"{fmt(h['roofline']['achieved'])} GB/s = **{h['roofline']['frac']:.3f}** of 8 TB/s), **{fmt(avg, 1)} µs average over {calls} calls** in "
                         f"`profiles/{TAG}_rocprofv3_kernel_stats.csv` ({fmt(byt / avg / 1e3)} GB/s = {byt / avg / 1e3 / 8000:.3f}); "
                         f"PMC traffic per launch {fmt(h['roofline']['traffic'] / 1e9, 4) + ' GB' if h['roofline'].get('traffic') else 'n/a'} vs "
                         f"{fmt(byt / 1e9, 4)} GB algorithmic; placement search: {h.get('placement')}.")
            lines.append("")
    h2 = load("bench_n1_second_box")
    if h2:   # the same build and command on another box of the pool: the placement lottery of §2.5, and the PMC traffic beside it
        r2 = h2["roofline"]
        lines.append(f"The same command on a second box (`profiles/{TAG}_bench_n1_second_box.json`, build `{h2.get('library_build_id')}`): **{fmt(h2['value'])}** it/s "
                     f"[{fmt(h2.get('value_median'))}], `{r2['kernel']}` {fmt(r2['avg_launch_us'], 1)} µs = **{r2['frac']:.3f}** of 8 TB/s "
                     f"({r2.get('frac_of_measured_mix', 0):.3f} of the bare mix on its own buffers; placement search: {h2.get('placement')}); "
                     f"`roofline.traffic` {fmt((r2.get('traffic') or 0) / 1e9, 4)} GB per launch from {r2.get('traffic_source')}.")
        lines.append("")
    for name, label in (("c1_", "config 1"), ("c2_", "config 2"), ("c3_", "config 3"), ("c4_", "config 4")):
        st = stats_csv(name)
        if not st:
            continue
        top = sorted(((k, v) for k, v in st.items() if k.startswith("k_")), key=lambda kv: -kv[1][0] * kv[1][1])[:5]
        lines.append(f"rocprofv3, {label} (`profiles/{TAG}_{name}rocprofv3_kernel_stats.csv`): " +
                     "; ".join(f"`{k}` {fmt(v[1], 1)} µs × {v[0]}" for k, v in top) + ".")
    lines.append("")
    g = load("gaps_fused_shard_n1.25e7")
    if g:
        ks = g.get("kernels", {})
        dom = max(ks.items(), key=lambda kv: kv[1]["total_us"]) if ks else None
        gap = g["gaps"][0] if g.get("gaps") else None
        if dom and gap:
            lines.append(f"Kernel-trace gap table at the 8-GPU shard size (`profiles/{TAG}_gaps_fused_shard_n1.25e7.json`, events off): `{dom[0]}` "
                         f"{fmt(dom[1]['avg_us'], 1)} µs average ({dom[1]['calls']} calls) + {fmt(gap['median_us'], 1)} µs median gap "
                         f"= {fmt(dom[1]['avg_us'] + gap['median_us'], 1)} µs per iteration.")
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
