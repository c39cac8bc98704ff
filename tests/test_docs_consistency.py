"""CPU tier: the numbers DESIGN.md §4 and BASELINE.md §3 quote are the artefacts' numbers.

The measured tables of both files are GENERATED from profiles/ by scripts/regen_tables.py; this test regenerates them in
memory and fails if a file's block differs (a hand-edited, stale or drifted number), and cross-checks the headline
kernel's two independent timings (HIP events inside bench.py vs the rocprofv3 CSV of the same command)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_measured_tables_are_generated_from_the_committed_artefacts():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "regen_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_headline_kernel_time_agrees_between_bench_events_and_rocprofv3():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import regen_tables as R
    h = R.load("bench_n1")
    assert h, "profiles/r03_bench_n1.json missing"
    st = R.stats_csv("")
    sym = h["roofline"]["kernel"]
    assert sym in st, (sym, list(st)[:5])
    calls, avg = st[sym]
    ev = h["roofline"]["avg_launch_us"]
    # two processes on two boxes of the same pool: buffer placement decides between ≈ 640 and ≈ 760 µs, so the
    # agreement asked for is "the same kernel at a plausible level", not equality
    assert calls >= 20 and 0.8 <= avg / ev <= 1.25, (avg, ev)
    # the quoted fraction follows from the quoted time
    rf = h["roofline"]
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] / 1e3) <= 1e-6 * rf["achieved"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9 and rf["peak"] == 8000.0
    assert h["cpu_baseline"]["kind"] == "port" and h["cpu_baseline"]["cores"] == 1 and h["cpu_baseline"]["value"] > 0


def test_every_workload_line_carries_its_cpu_baseline():
    """VERDICT r02 missing #6: c1–c4 lines are kept under profiles/ with the oracle's rate on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import regen_tables as R
    for stem in ("bench_c1", "bench_c1c", "bench_c2", "bench_c3", "bench_c4", "bench_n1"):
        d = R.load(stem)
        assert d, stem
        for key in ("cpu_baseline", "cpu_baseline_all_cores"):
            assert d.get(key) and d[key]["value"] and d[key]["value"] > 0 and d[key]["kind"] == "port", (stem, key, d.get(key))
        assert d["roofline"]["kernel"] and d["library_build_id"]
