#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/prof_{stats,fetch,write}) into profiles/<tag>_*.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are collected in SEPARATE --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots),
are in KiB, and on gfx950 FETCH_SIZE reports exactly half of a 16-B-per-lane coalesced
stream, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

MODE_NAMES = {"16": "init", "12": "trial", "4": "trial_nobeta", "15": "accept_dir_trial", "3": "accept_dir",
              "1": "accept_only", "2": "dir", "32": "reset_dir", "64": "upg_norm", "128": "beta_partials"}


CG_MODE_NAMES = {"8": "init", "4": "trial", "7": "accept_dir_trial", "3": "accept_dir", "1": "accept_only",
                 "16": "reset_dir", "32": "upg_norm", "64": "grad", "128": "grad_trial"}


def short(name):
    if "k_cg<" in name:  # k_cg<Obj, MODE, NPTS, BIG>: the gradient-free multi-point family
        inside = name.split("k_cg<")[1].split(">")[0].split(",")
        return CG_MODE_NAMES.get(inside[1].strip(), "mode" + inside[1].strip())
    if "k_fused" in name:
        inside = name.split("k_fused<")[1].split(">")[0].split(",")
        return MODE_NAMES.get(inside[1].strip(), "mode" + inside[1].strip())
    return name.split("(")[0].replace("cgo::dev::", "").replace("void ", "")


def counters(which, cname):
    f = sorted(glob.glob(os.path.join(src, f"prof_{which}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    if not f:
        return acc
    for row in csv.DictReader(open(f[-1])):  # newest run
        if row["Counter_Name"] == cname:
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def symbol(name):
    """'void cgo::dev::k_cg<cgo::dev::ObjQuadDiag, 7, 7, true>(cgo::dev::RParams)' → 'k_cg<ObjQuadDiag, 7, 7, true>'
    — the form cgo_solver_kernel_symbol() returns, so that bench.py can match a summary entry to the running kernel."""
    return name.split("(")[0].replace("void ", "").replace("cgo::dev::", "").replace("cgo::", "").strip()


def symbols(which):
    f = sorted(glob.glob(os.path.join(src, f"prof_{which}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    out = {}
    if f:
        for row in csv.DictReader(open(f[-1])):
            out.setdefault(short(row["Kernel_Name"]), symbol(row["Kernel_Name"]))
    return out


def meta():
    import subprocess
    sys.path.insert(0, ROOT)
    import cgo_amd
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "conjugategradientoptim.jl_amd/csrc", "include"],
                                capture_output=True, text=True).stdout.strip())
    return dict(library_build_id=cgo_amd.build_id(), git_head=head, csrc_dirty_vs_head=dirty,
                note=f"collected with scripts/profile_{tag}.sh; FETCH_SIZE doubled (gfx950 counts 128-B requests of a 16-B/lane stream as 64 B), KiB units")


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
syms = symbols("fetch")
summary = {"_meta": meta()}
for k in sorted(set(fetch) | set(write)):
    fk = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else None
    wk = sum(write[k]) / len(write[k]) if write.get(k) else None
    if fk is None or wk is None:
        continue
    rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
    summary[k] = dict(kernel_symbol=syms.get(k, ""), fetch_size_kib_raw=fk, write_size_kib=wk, read_bytes_corrected=rd, write_bytes=wr,
                      hbm_bytes_per_launch=rd + wr, launches_fetch_pass=len(fetch[k]), launches_write_pass=len(write[k]))
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)

for pat, out in (("prof_stats/*/*_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats.csv"),
                 ("prof_stats_c1/*/*_kernel_stats.csv", f"{tag}_c1_rocprofv3_kernel_stats.csv"),
                 ("prof_stats_c2/*/*_kernel_stats.csv", f"{tag}_c2_rocprofv3_kernel_stats.csv"),
                 ("prof_stats_c3/*/*_kernel_stats.csv", f"{tag}_c3_rocprofv3_kernel_stats.csv"),
                 ("prof_stats_c4/*/*_kernel_stats.csv", f"{tag}_c4_rocprofv3_kernel_stats.csv")):
    f = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    if f:
        shutil.copy(f[-1], os.path.join(dst, out))
for name, out in (("bench_n1.json", f"{tag}_bench_n1.json"), ("bench_c1.json", f"{tag}_bench_c1.json"), ("bench_c1c.json", f"{tag}_bench_c1c.json"),
                  ("bench_c2.json", f"{tag}_bench_c2.json"), ("bench_c2_hostdriven.json", f"{tag}_bench_c2_hostdriven.json"),
                  ("bench_c1_hostdriven.json", f"{tag}_bench_c1_hostdriven.json"),
                  ("bench_c3.json", f"{tag}_bench_c3.json"), ("bench_c4.json", f"{tag}_bench_c4.json"), ("bench_c4_twopass.json", f"{tag}_bench_c4_twopass.json"),
                  ("bench_rehearsal_2ranks.json", f"{tag}_bench_rehearsal_2ranks_one_gpu.json"), ("bench_rehearsal_4ranks.json", f"{tag}_bench_rehearsal_4ranks_one_gpu.json"), ("bench_shard.json", f"{tag}_bench_shard_n1p25e7.json"),
                  ("gaps_c2.json", f"{tag}_gaps_fused_c2_n1e6.json"), ("gaps_shard.json", f"{tag}_gaps_fused_shard_n1.25e7.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, out))
print(json.dumps(summary, indent=1))
