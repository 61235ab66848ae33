#!/usr/bin/env python3
"""Any set of rocprofv3 PMC counters summed per kernel name over one eager bench.py training step (one --pmc pass, kernel trace
only).  GPU box only.  usage: python tools/pmc_counters.py --counters SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE [--config c1p] [--top 12]"""
import argparse, csv, glob, os, subprocess, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--counters", nargs="+", required=True)
ap.add_argument("--config", default="c1p")
ap.add_argument("--batch", default="32")
ap.add_argument("--top", type=int, default=12)
a = ap.parse_args()
d = "/tmp/pmc_any"
subprocess.run(["rm", "-rf", d])
cmd = ["timeout", "500", "rocprofv3", "--pmc", *a.counters, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
       os.path.join(REPO, "bench.py"), "--config", a.config, "--batch", a.batch, "--steps", "1", "--warmup", "1", "--no-graph",
       "--no-cpu-baseline", "--no-roofline", "--no-f32"]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if r.returncode != 0:
    raise SystemExit(f"rocprofv3 failed ({r.returncode}):\n{r.stdout[-2000:]}")
agg = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]] += float(row["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1][a.counters[-1]])[:a.top]
print("kernel".ljust(72) + "".join(c[-22:].rjust(24) for c in a.counters))
for k, c in rows:
    print(k.ljust(72) + "".join(f"{c[n]:24.4g}" for n in a.counters))
