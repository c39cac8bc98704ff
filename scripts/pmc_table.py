#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc output directories into one JSON table.

    python3 scripts/pmc_table.py OUT.json DIR [DIR ...] [--skip K] [--match SUBSTR]

Every DIR holds one `rocprofv3 --kernel-trace --pmc …` run (its *_counter_collection.csv).  Per kernel symbol and counter:
mean / min / max of Counter_Value over the dispatches (the first K dispatches of a kernel are skipped: warm-up), the
number of dispatches, grid and register counts, and the mean dispatch duration UNDER the counter pass (End − Start; a
profiled pass runs at lower clocks than an un-profiled one, so these durations are only compared among themselves).
Units are the counters' own: FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE to be doubled for 16-B-per-lane streams on gfx950,
/opt/skills/guides/MI355X_MICROARCH.md §HBM), SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over waves.
"""
import collections
import csv
import glob
import json
import os
import sys


def symbol(name):
    return name.split("(")[0].replace("void ", "").replace("cgo::dev::", "").replace("cgo::", "").strip()


def main():
    args = sys.argv[1:]
    skip, match = 0, None
    if "--skip" in args:
        i = args.index("--skip"); skip = int(args[i + 1]); del args[i:i + 2]
    if "--match" in args:
        i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
    out, dirs = args[0], args[1:]
    table = collections.defaultdict(lambda: dict(counters={}, info={}))
    for d in dirs:
        files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            print(f"no counter csv under {d}", file=sys.stderr)
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel → dispatch → counter → value
        info = {}
        dur = collections.defaultdict(dict)
        for row in csv.DictReader(open(files[-1])):
            k = symbol(row["Kernel_Name"])
            if match and match not in k:
                continue
            did = int(row["Dispatch_Id"])
            c = row["Counter_Name"]
            per[k][did][c] = per[k][did].get(c, 0.0) + float(row["Counter_Value"])
            info[k] = dict(grid=int(row["Grid_Size"]), workgroup=int(row["Workgroup_Size"]), vgpr=int(row["VGPR_Count"]),
                           accum_vgpr=int(row.get("Accum_VGPR_Count", 0) or 0), sgpr=int(row["SGPR_Count"]),
                           lds=int(row["LDS_Block_Size"]), scratch=int(row["Scratch_Size"]))
            try:
                dur[k][did] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            except Exception:
                pass
        for k, disp in per.items():
            ids = sorted(disp)[skip:] or sorted(disp)
            cs = collections.defaultdict(list)
            for i in ids:
                for c, v in disp[i].items():
                    cs[c].append(v)
            for c, v in cs.items():
                table[k]["counters"][c] = dict(mean=sum(v) / len(v), min=min(v), max=max(v), dispatches=len(v))
            table[k]["info"] = info[k]
            ds = [dur[k][i] for i in ids if i in dur[k]]
            if ds:
                table[k].setdefault("duration_us_under_pmc", {})[os.path.basename(d.rstrip("/"))] = sum(ds) / len(ds)
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(table.items()):
        print(k, v["info"])
        for c, s in sorted(v["counters"].items()):
            print(f"    {c:36s} {s['mean']:16.1f}  (n={s['dispatches']}, min {s['min']:.1f}, max {s['max']:.1f})")


if __name__ == "__main__":
    main()
