#!/usr/bin/env python3
"""MFMA utilisation of one bench.py training step from the rocprofv3 PMC counters (north_star: "rocprof HBM GB/s and MFMA
utilisation against gfx950 peak").  One --pmc pass (kernel trace only, no other trace domain) with
SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles summed over the SIMDs), GRBM_GUI_ACTIVE (GPU-active cycles of the dispatch)
and SQ_BUSY_CYCLES; utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs), the gfx94x MfmaUtil formula (ROCm 7.2 ships no
gfx950 derived-metric section, MI355X_MICROARCH.md).  GPU box only.
usage: python tools/pmc_mfma.py [--mode train|eval] [--out profiles/r02_mfma_train_bf16.json]"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    d = f"/tmp/pmc_mfma_{a.mode}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["timeout", "500", "rocprofv3", "--pmc", *COUNTERS, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--mode", a.mode, "--steps", "1", "--warmup", "1", "--no-graph",
           "--no-cpu-baseline", "--no-roofline", "--no-f32"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 failed ({r.returncode}):\n{r.stdout[-2000:]}")
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                grp = ("igemm (forward convs / linears + data gradients)" if "igemm" in k else
                       "wgrad (weight gradients)" if ("wgrad_kernel" in k or "wgrad_group_kernel" in k) else "everything else")
                agg[grp][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == COUNTERS[0]:
                    calls[grp] += 1
    out = {"source": "rocprofv3 --pmc " + " ".join(COUNTERS) + f" (one pass, kernel trace only), bench.py --mode {a.mode} --no-graph "
                     "--steps 1 --warmup 1 (all executed passes summed); tools/pmc_mfma.py",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)  (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs)", "groups": {}}
    tot_busy = tot_act = 0.0
    for grp, c in agg.items():
        busy, act = c[COUNTERS[0]], c[COUNTERS[1]]
        tot_busy += busy
        tot_act += act
        out["groups"][grp] = {"dispatches": calls[grp], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                              "mfma_util": round(busy / (act / 8.0 * 1024.0), 4) if act else None}
    out["whole_step_mfma_util"] = round(tot_busy / (tot_act / 8.0 * 1024.0), 4) if tot_act else None
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
