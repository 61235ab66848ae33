#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + optional PMC counters) into a small text table for profiles/.

usage: python tools/summarize_rocprof.py <rocprof output dir> [--out profiles/xxx.txt]
Handles *_kernel_trace.csv (Kernel_Name, Start_Timestamp, End_Timestamp), *_kernel_stats.csv and
*_counter_collection.csv (Kernel_Name, Counter_Name, Counter_Value)."""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--igemm-json", default=None, help="write {steps, igemm_ms_per_step, ...} of the kernel trace (bench.py reads it as "
                                                       "roofline.igemm_ms_rocprof); steps = calls of --step-kernel")
    ap.add_argument("--step-kernel", default="head_fwd_kernel", help="a kernel that runs exactly once per step")
    ap.add_argument("--replays-only", action="store_true",
                    help="keep only the periodic tail of the dispatch sequence: the hipGraph replays.  The eager warm-up step, the capture's "
                         "warm-up and the first replays (cold caches: 1 ms outliers on 23 us kernels) are cut off, so the per-step "
                         "figures are those of the steady state the HIP-event timing in bench.py sees")
    ap.add_argument("--stats-csv", default=None, help="with --replays-only: write the trimmed per-kernel statistics in rocprofv3's kernel_stats.csv columns")
    a = ap.parse_args()
    lines = []
    per_step = None
    traces = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    for t in traces:
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        with open(t) as f:
            rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(f)]
        rows.sort()
        trimmed = ""
        if a.replays_only:
            # dispatch indices of the once-per-step kernel; the replays are the tail in which they are a constant number of
            # dispatches apart (an eager step launches more kernels than its captured form: casts, packs, statistics fall-backs).
            # The window runs from the second occurrence of that tail to the last one: whole periods, whatever the phase.
            idx = [i for i, r in enumerate(rows) if r[2].startswith(a.step_kernel)]
            if len(idx) >= 4:
                period = idx[-1] - idx[-2]
                k = len(idx) - 1
                while k > 0 and idx[k] - idx[k - 1] == period:
                    k -= 1
                k = min(k + 1, len(idx) - 2)          # drop the first replay of the tail as well (cold instruction / L2 state)
                rows_all = len(rows)
                rows = rows[idx[k]:idx[-1]]
                trimmed = f"  [replays only: {len(idx) - 1 - k} steps x {period} dispatches of {rows_all} traced]"
        for s0, e0, nm in rows:
            d = (e0 - s0) / 1e3
            g = agg[nm]
            g[0] += 1; g[1] += d; g[2] = min(g[2], d); g[3] = max(g[3], d)
        tot = sum(v[1] for v in agg.values())
        lines.append(f"# kernel trace: {os.path.relpath(t, a.dir)}  total kernel time {tot / 1e3:.3f} ms{trimmed}")
        if a.replays_only and a.stats_csv:
            with open(a.stats_csv, "w") as f:
                f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
                for k_, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write(f'"{k_}",{v[0]},{int(v[1] * 1e3)},{v[1] * 1e3 / v[0]:.1f},{100 * v[1] / tot:.2f},{int(v[2] * 1e3)},{int(v[3] * 1e3)}\n')
        lines.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{v[0]:7d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {v[2]:9.2f} {v[3]:9.2f} {100 * v[1] / tot:6.2f}  {k}")
        steps = sum(v[0] for k, v in agg.items() if k.startswith(a.step_kernel))
        if steps:
            def grp(pred):
                return sum(v[1] for k, v in agg.items() if pred(k)) / steps / 1e3, sum(v[0] for k, v in agg.items() if pred(k)) // steps
            ig = grp(lambda k: k.startswith("igemm") or k.startswith("splitk_epilogue"))
            wg = grp(lambda k: k.startswith("wgrad"))
            per_step = {"traced_steps": steps, "igemm_ms_per_step": round(ig[0], 3), "igemm_launches_per_step": ig[1],
                        "wgrad_ms_per_step": round(wg[0], 3), "wgrad_launches_per_step": wg[1],
                        "all_kernels_ms_per_step": round(tot / steps / 1e3, 3),
                        "kernels_per_step": sum(v[0] for v in agg.values()) // steps,
                        "note": "rocprofv3 --kernel-trace" + (", graph replays only" if a.replays_only else "") + "; igemm = igemm_kernel* + "
                                "igemm_big_kernel + splitk_epilogue_kernel; bench.py --no-side-stream (one stream: durations are not "
                                "inflated by co-running kernels)"}
    for t in glob.glob(os.path.join(a.dir, "**", "*_results.db"), recursive=True):   # rocprofv3 default (rocpd sqlite)
        import sqlite3
        con = sqlite3.connect(t)
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        for name, dur in con.execute("select name, duration from kernels"):
            d = dur / 1e3
            g = agg[short(name)]
            g[0] += 1; g[1] += d; g[2] = min(g[2], d); g[3] = max(g[3], d)
        tot = sum(v[1] for v in agg.values())
        lines.append(f"# kernel trace: {os.path.relpath(t, a.dir)}  total kernel time {tot / 1e3:.3f} ms")
        lines.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{v[0]:7d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {v[2]:9.2f} {v[3]:9.2f} {100 * v[1] / tot:6.2f}  {k}")
    for t in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(t) as f:
            for r in csv.DictReader(f):
                g = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                g[0] += 1; g[1] += float(r["Counter_Value"])
        lines.append(f"# counters: {os.path.relpath(t, a.dir)}  (sum over dispatches; per-dispatch avg in brackets)")
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
            for c, v in sorted(cs.items()):
                lines.append(f"{c:>22} {v[1]:18.1f} [{v[1] / v[0]:14.1f} x {v[0]:5d}]  {k}")
    if a.igemm_json and per_step is not None:
        import json
        with open(a.igemm_json, "w") as f:
            json.dump(per_step, f)
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
