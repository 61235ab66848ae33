#!/usr/bin/env python3
"""Any rocprofv3 PMC counters summed per kernel name over an arbitrary command (one --pmc pass, kernel trace only).  GPU box only.
usage: python tools/pmc_cmd.py --counters TCC_HIT_sum TCC_MISS_sum [--top 12] [--match wgrad] -- python tools/bench_wgrad.py ..."""
import argparse, csv, glob, os, subprocess, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("--counters", nargs="+", required=True)
ap.add_argument("--top", type=int, default=12)
ap.add_argument("--match", default="")
ap.add_argument("cmd", nargs=argparse.REMAINDER)
a = ap.parse_args()
cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
d = "/tmp/pmc_cmd"
subprocess.run(["rm", "-rf", d])
full = ["timeout", "500", "rocprofv3", "--pmc", *a.counters, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd]
r = subprocess.run(full, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if r.returncode != 0:
    raise SystemExit(f"rocprofv3 failed ({r.returncode}):\n{r.stdout[-2000:]}")
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:70]
            if a.match and a.match not in k:
                continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == a.counters[0]:
                cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][a.counters[-1]])[:a.top]
print("kernel".ljust(72) + "launches".rjust(10) + "".join(c[-22:].rjust(24) for c in a.counters))
for k, c in rows:
    print(k.ljust(72) + f"{cnt[k]:10d}" + "".join(f"{c[n]:24.6g}" for n in a.counters))
