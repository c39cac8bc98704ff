#!/usr/bin/env python3
"""Condenses gpurun_out/r04_prof (scripts/profile_r04.sh) into profiles/r04_*.

  r04_pmc_summary.json            HBM bytes per launch of every kernel of every HBM-bound workload, from separate --pmc
                                  FETCH_SIZE / WRITE_SIZE passes (KiB units; FETCH_SIZE doubled: gfx950 counts the 128-B requests
                                  of a 16-B-per-lane stream as 64 B — /opt/skills/guides/MI355X_MICROARCH.md §HBM), keyed
                                  "<workload>/<kernel symbol>" with the problem size and the library build
  r04_<tag>_rocprofv3_kernel_stats.csv + r04_<tag>_rocprofv3_bench_line.json
                                  rocprofv3 --kernel-trace --stats of a bench.py command AND the line that very process printed
                                  (placement search outcome and level, its own HIP-event times): a reader recomputes every
                                  "% of 8 TB/s" from one CSV row and the algorithmic bytes of the line beside it
  r04_bench_*.json                the un-profiled bench lines (with roofline.traffic and the full-size CPU baselines)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r04"
SRC = os.path.join(ROOT, "gpurun_out", "r04_prof")
DST = os.path.join(ROOT, "profiles")
N_LOCAL = {"c5": 10**8, "shard": 12_500_000, "c3": 10**7, "c3big": 4 * 10**7, "c4": 10**7}


def symbol(name):
    return name.split("(")[0].replace("void ", "").replace("cgo::dev::", "").replace("cgo::", "").strip()


def counters(tag, which, cname):
    f = sorted(glob.glob(os.path.join(SRC, f"prof_{which}_{tag}", "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    if not f:
        return acc
    for row in csv.DictReader(open(f[-1])):
        if row["Counter_Name"] == cname:
            acc[symbol(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def meta():
    sys.path.insert(0, ROOT)
    import cgo_amd
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "conjugategradientoptim.jl_amd/csrc", "include"],
                                capture_output=True, text=True).stdout.strip())
    return dict(library_build_id=cgo_amd.build_id(), git_head=head, csrc_dirty_vs_head=dirty,
                note="collected with scripts/profile_r04.sh; FETCH_SIZE doubled (gfx950 counts 128-B requests of a 16-B/lane stream as 64 B), KiB units; "
                     "one --pmc counter per pass")


def pmc_summary():
    summary = {"_meta": meta()}
    for tag, n in N_LOCAL.items():
        fetch, write = counters(tag, "fetch", "FETCH_SIZE"), counters(tag, "write", "WRITE_SIZE")
        for k in sorted(set(fetch) & set(write)):
            if not (k.startswith("k_") and len(fetch[k]) >= 3):
                continue
            fv, wv = fetch[k][1:], write[k][1:]      # (the first launch of a kind: cold caches)
            fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
            rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
            summary[f"{tag}/{k}"] = dict(kernel_symbol=k, n_local=n, workload=tag, fetch_size_kib_raw=fk, write_size_kib=wk,
                                         read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
                                         launches_fetch_pass=len(fetch[k]), launches_write_pass=len(write[k]))
    json.dump(summary, open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w"), indent=1)
    return summary


def bench_line_of(log):
    if not os.path.exists(log):
        return None
    for ln in reversed(open(log, errors="replace").read().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                return json.loads(ln)
            except Exception:
                return None
    return None


def main():
    os.makedirs(DST, exist_ok=True)
    s = pmc_summary()
    print(json.dumps({k: (v if k == "_meta" else {a: v[a] for a in ("n_local", "hbm_bytes_per_launch", "launches_fetch_pass")}) for k, v in s.items()}, indent=1))
    if "--pmc-only" in sys.argv:
        return
    for tag in ("c5", "c5_searchmiss", "c5_nosearch", "c1", "c2", "c3", "c3big", "c4", "shard"):
        f = sorted(glob.glob(os.path.join(SRC, f"prof_stats_{tag}", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if f:
            shutil.copy(f[-1], os.path.join(DST, f"{TAG}_{tag}_rocprofv3_kernel_stats.csv"))
        line = bench_line_of(os.path.join(SRC, f"prof_stats_{tag}.log"))
        if line:
            json.dump(line, open(os.path.join(DST, f"{TAG}_{tag}_rocprofv3_bench_line.json"), "w"))
    for name in ("c5", "c5_nosearch", "c1", "c1c", "c2", "c3", "c3big", "c4", "c4_twopass", "shard", "c1_hostdriven", "c2_hostdriven",
                 "rehearsal_2ranks", "rehearsal_4ranks"):
        p = os.path.join(SRC, f"bench_{name}.json")
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(DST, f"{TAG}_bench_{name}.json"))
    p = os.path.join(SRC, "gaps_shard.json")
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, f"{TAG}_gaps_fused_shard_n1p25e7.json"))


if __name__ == "__main__":
    main()
