#!/usr/bin/env python3
"""Kernel-trace gap table: for a rocprofv3 --kernel-trace CSV, the time between the END of one kernel and
the START of the next on the device, grouped by (previous kernel → next kernel), plus per-kernel durations.

    python3 scripts/gap_table.py <dir-or-csv> [--skip N] [--out profiles/xyz.json]

The fixed cost of a fused launch ("Next round" #2) is everything that is not the k_cg/k_fused
kernel itself: the gap rows below + the reduce/controller kernels.  --skip drops the first N dispatches
(warm-up, fills, hiprtc) so that the table describes the steady state.
"""
import collections
import csv
import glob
import json
import os
import re
import statistics
import sys


def short(name: str) -> str:
    name = name.replace("cgo::dev::", "").replace("cgo::", "").replace("void ", "")
    m = re.match(r"(k_\w+)<(.*)>", name.split("(")[0])
    if not m:
        return name.split("(")[0][:60]
    args = [a.strip() for a in m.group(2).split(",")]
    args = [a for a in args if not a.startswith("Obj")] if m.group(1) in ("k_cg", "k_fused") else args
    obj = [a for a in [x.strip() for x in m.group(2).split(",")] if a.startswith("Obj")]
    return f"{m.group(1)}<{','.join(args)}>" + (f"[{obj[0][3:]}]" if obj else "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    src = args[0]
    if os.path.isdir(src):
        fs = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
        if not fs:
            raise SystemExit(f"no *kernel_trace.csv under {src}")
        src = fs[-1]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[skip:]
    dur = collections.defaultdict(list)
    gaps = collections.defaultdict(list)
    for i, (s, e, k) in enumerate(rows):
        dur[k].append((e - s) / 1e3)
        if i:
            ps, pe, pk = rows[i - 1]
            gaps[(pk, k)].append((s - pe) / 1e3)
    span_us = (rows[-1][1] - rows[0][0]) / 1e3 if rows else 0.0
    busy_us = sum(sum(v) for v in dur.values())
    res = {"source": os.path.relpath(src), "dispatches": len(rows), "span_us": span_us, "kernel_busy_us": busy_us,
           "kernel_busy_fraction": busy_us / span_us if span_us else None, "kernels": {}, "gaps": []}
    print(f"{len(rows)} dispatches over {span_us / 1e3:.2f} ms, kernels busy {100 * busy_us / max(span_us, 1e-9):.1f} %")
    print(f"{'kernel':58s} {'calls':>7s} {'avg us':>9s} {'med us':>9s} {'total ms':>9s}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        res["kernels"][k] = dict(calls=len(v), avg_us=sum(v) / len(v), median_us=statistics.median(v), total_us=sum(v))
        print(f"{k[:58]:58s} {len(v):7d} {sum(v) / len(v):9.2f} {statistics.median(v):9.2f} {sum(v) / 1e3:9.2f}")
    print(f"\n{'gap: previous kernel -> next kernel':88s} {'count':>6s} {'med us':>8s} {'avg us':>8s} {'total ms':>9s}")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 3:
            continue
        res["gaps"].append(dict(prev=a, next=b, count=len(v), median_us=statistics.median(v), avg_us=sum(v) / len(v), total_us=sum(v)))
        print(f"{(a[:42] + ' -> ' + b[:42]):88s} {len(v):6d} {statistics.median(v):8.2f} {sum(v) / len(v):8.2f} {sum(v) / 1e3:9.2f}")
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
