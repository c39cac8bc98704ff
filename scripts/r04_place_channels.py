#!/usr/bin/env python3
"""Per-instance (channel) TCC counter values of k_mix<1> (fast buffer triple) vs k_mix<2> (slow triple) from rocprofv3 json output.

    python3 scripts/r04_place_channels.py gpurun_out/r04_prof

The json keeps one record per (dispatch, counter, dimension instance); the csv of the same run sums them.  Printed per counter:
the per-instance mean over the dispatches of each kernel, its spread (max/mean, coefficient of variation) — is the slow level
a matter of a few hot channels, or of all channels alike?"""
import collections
import glob
import json
import os
import sys

import numpy as np

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "place_chan_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*results.json"), recursive=True)
    if not files:
        print(d, "no json")
        continue
    doc = json.load(open(files[0]))
    tool = doc["rocprofiler-sdk-tool"][0] if isinstance(doc.get("rocprofiler-sdk-tool"), list) else doc.get("rocprofiler-sdk-tool", doc)
    # names
    ksyms = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("demangled_kernel_name", k.get("kernel_name", ""))) for k in tool.get("kernel_symbols", [])}
    cnames = {}
    for c in tool.get("counters", []):
        cnames[c.get("id", {}).get("handle", c.get("id"))] = c.get("name")
    recs = tool.get("callback_records", {}).get("counter_collection", []) or tool.get("buffer_records", {}).get("counter_collection", [])
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # (kernel, counter) → instance → values over dispatches
    for r in recs:
        di = r.get("dispatch_data", {}).get("dispatch_info", {})
        k = ksyms.get(di.get("kernel_id"), "")
        if "k_mix<1>" in k:
            kk = "fast"
        elif "k_mix<2>" in k:
            kk = "slow"
        else:
            continue
        seen = collections.Counter()
        for rec in r.get("records", []):     # one record per (counter, instance), instances in a fixed order: 16 TCC channels × 8 XCCs
            cid = rec.get("counter_id", {}).get("handle", rec.get("counter_id"))
            inst = seen[cid]; seen[cid] += 1
            per[(kk, cnames.get(cid, str(cid)))][inst].append(float(rec.get("value", 0.0)))
    print("==", os.path.basename(d), f"({len(recs)} dispatch records)")
    for (kk, cn), insts in sorted(per.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        v = np.array([np.mean(x) for _, x in sorted(insts.items())])
        if v.size == 0:
            continue
        print(f"  {cn:34s} {kk:4s} instances {v.size:4d}  sum {v.sum():16.1f}  mean {v.mean():14.1f}  min {v.min():14.1f}  max {v.max():14.1f}  "
              f"max/mean {v.max() / max(v.mean(), 1e-30):6.3f}  cv {v.std() / max(v.mean(), 1e-30):6.3f}")
        if v.size == 128:      # per XCC (8 × 16 channels, XCC-major or channel-major: print both foldings)
            a = v.reshape(8, 16)
            print("      by 16-blocks :", " ".join(f"{x:10.0f}" for x in a.sum(axis=1)))
            print("      by stride-8  :", " ".join(f"{x:10.0f}" for x in v.reshape(16, 8).sum(axis=0)))
