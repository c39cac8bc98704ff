#!/usr/bin/env python3
"""In-kernel timeline of the LAST k_cg launch of a solve (diagnostic build: make EXTRA=-DCGO_STAMPS, CGO_LIB_PATH=…/libcgo_hip_stamps.so).

    python3 scripts/r04_stamps.py [n] [iters]

Per workgroup: entry, end of the streaming loop, end of the reduction tail (100-MHz wall clock), XCC id.  Prints when workgroups
start, how long their loops run, how their loop ends spread, and what the tail adds — per XCC as well: where do the µs
between the bare stream mix and the k_cg launch go at the 8-GPU shard size?
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgo_amd as cgo  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = cgo.Context(0)
obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0, ctx)
s = cgo.Solver(obj, cgo.setupCGConfig(1e-200, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=iters + 8), cgo.setupStrongWolfeBisection(1e-5, 0.1))
s.set_x0_fill("constant", 1.0)
s.start()
s.profile(True)
s.iterate(iters)
prof = s.profile_get()
print({k: (v["launches"], round(v["total_ms"] / v["launches"] * 1e3, 1)) for k, v in prof.items()}, s.kernel_symbol("accept_dir_trial"))
L = cgo.lib()
L.cgo_debug_stamps.restype = C.c_int
L.cgo_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = np.zeros(4096 * 4, dtype=np.uint64)
assert L.cgo_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
st = buf.reshape(4096, 4)
live = st[:, 0] > 0
t0max = st[live, 0].max()
live &= st[:, 0] + 20000 > t0max        # workgroups of the last launch (within 200 µs of its latest entry)
w = st[live]
base = w[:, 0].min()
t0 = (w[:, 0] - base) / 100.0
t1 = (w[:, 1] - base) / 100.0
t2 = (w[:, 2] - base) / 100.0
xcc = (w[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (w[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
print(f"workgroups in the last launch: {len(w)}")
q = lambda v: " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 5, 25, 50, 75, 95, 100))
print("percentiles [µs]              min      5%     25%     50%     75%     95%     max")
print("entry (after first entry)  ", q(t0))
print("loop end                   ", q(t1))
print("loop duration              ", q(t1 - t0))
print("workgroup end              ", q(t2))
print("tail (end − loop end)      ", q(t2 - t1))
print(f"last loop end {t1.max():.1f} µs → last workgroup end {t2.max():.1f} µs: the launch's own tail costs {t2.max() - t1.max():.1f} µs after the slowest loop")
print(f"mean loop end {t1.mean():.1f} µs vs max {t1.max():.1f} µs: {t1.max() - t1.mean():.1f} µs of imbalance")
for x in sorted(set(xcc)):
    m = xcc == x
    print(f"  XCC {x}: {m.sum():4d} workgroups, entry {t0[m].mean():6.1f}, loop duration mean {np.mean(t1[m] - t0[m]):6.1f} max {np.max(t1[m] - t0[m]):6.1f}, loop end mean {t1[m].mean():6.1f} max {t1[m].max():6.1f}")
# the slowest and fastest 5 % of the loops: which XCC / CU / SE?
d = t1 - t0
order = np.argsort(d)
k = max(len(d) // 20, 1)
for name, idx in (("fastest", order[:k]), ("slowest", order[-k:])):
    print(name, "5 % loops: XCC histogram", np.bincount(xcc[idx], minlength=8).tolist(), "mean", round(float(d[idx].mean()), 1), "µs")
bi = np.nonzero(live)[0]
print("slowest 5 %: blockIdx sample", bi[order[-k:]][:16].tolist(), "hw_id sample", [hex(int(h)) for h in hwid[order[-k:]][:8]])
# pairing: the workgroups that shared a CU (key: XCC, SE, SH, CU of HW_ID) — do a CU's two workgroups split into a fast and a slow one?
cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
simd = (hwid >> 4) & 3
groups = {}
for i, c in enumerate(cu):
    groups.setdefault(int(c), []).append(i)
sizes = np.bincount([len(v) for v in groups.values()])
print("workgroups per CU histogram (index = count):", sizes.tolist(), "CUs used:", len(groups))
pairs = [(d[v[0]], d[v[1]], t0[v[0]], t0[v[1]]) for v in groups.values() if len(v) == 2]
if pairs:
    pr = np.array(pairs)
    first_is_older = pr[:, 2] <= pr[:, 3]
    older = np.where(first_is_older, pr[:, 0], pr[:, 1]); younger = np.where(first_is_older, pr[:, 1], pr[:, 0])
    print(f"CUs with two workgroups: {len(pr)}; loop duration of the one that entered first: mean {older.mean():.1f} (min {older.min():.1f}, max {older.max():.1f}); "
          f"of the other: mean {younger.mean():.1f} (min {younger.min():.1f}, max {younger.max():.1f}); |difference| mean {np.abs(older - younger).mean():.1f} µs; "
          f"older faster in {int((older < younger).sum())} of {len(pr)}")
    print("per-CU mean of the pair: ", q((older + younger) / 2))
out = os.environ.get("CGO_STAMPS_OUT")
if out:
    np.savez(out, stamps=w, blockidx=bi)
s.close(); obj.close(); ctx.close()
