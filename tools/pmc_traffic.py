#!/usr/bin/env python3
"""HBM traffic of one bench.py step from the rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE in SEPARATE --pmc passes (kernel trace only), FETCH_SIZE doubled (gfx950 counts 128-byte requests of wide
coalesced reads at 64 bytes), WRITE_SIZE as reported; both counters are in KiB... (rocprofv3 reports kilobytes).
GPU box only.  usage: python tools/pmc_traffic.py --mode train|eval [--out profiles/r01_traffic_train_bf16.json]"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, mode, steps, warmup):
    d = f"/tmp/pmc_traffic_{counter}_{mode}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["timeout", "400", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--mode", mode, "--steps", str(steps), "--warmup", str(warmup),
           "--no-graph", "--no-cpu-baseline", "--no-roofline", "--no-f32"]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 {counter} failed ({r.returncode}):\n{r.stdout[-2000:]}")
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                a = agg[k]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    passes = a.steps + a.warmup + (0 if a.mode == "train" else 0)
    fetch = one_pass("FETCH_SIZE", a.mode, a.steps, a.warmup)
    write = one_pass("WRITE_SIZE", a.mode, a.steps, a.warmup)

    def total(agg, pred):
        return sum(v[1] for k, v in agg.items() if pred(k)), sum(v[0] for k, v in agg.items() if pred(k))
    is_igemm = lambda k: "igemm_kernel" in k or "igemm_big_kernel" in k
    is_wgrad = lambda k: "wgrad_kernel" in k
    everything = lambda k: True
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only), bench.py --mode {a.mode} "
                     f"--no-graph --steps {a.steps} --warmup {a.warmup}; per-step = totals / executed steps; tools/pmc_traffic.py",
           "dtype": "bf16", "batch": 32,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both in KB"}
    f_ig, n_ig = total(fetch, is_igemm)
    w_ig, _ = total(write, is_igemm)
    steps_seen = None
    # steps executed in the profiled process: bench runs warmup + steps eager passes (+1 eager warm-up pass in train mode)
    per = lambda kb: kb * 1024.0
    out["igemm_launches_total"] = n_ig
    for name, pred in (("igemm", is_igemm), ("wgrad", is_wgrad), ("all_kernels", everything)):
        f, n = total(fetch, pred)
        w, _ = total(write, pred)
        out[name] = {"launches": n, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "hbm_bytes": per(2.0 * f + w)}
    out["note"] = "divide by the number of executed passes (igemm launches per pass: 166 train / 83 eval) to get per-step bytes"
    lp = 166 if a.mode == "train" else 83
    n_pass = max(1, round(out["igemm"]["launches"] / lp))
    out["passes"] = n_pass
    out["launches_per_step"] = lp
    out["hbm_bytes_per_step"] = out["igemm"]["hbm_bytes"] / n_pass
    out["hbm_bytes_per_launch"] = out["igemm"]["hbm_bytes"] / max(1, out["igemm"]["launches"])
    out["wgrad_hbm_bytes_per_step"] = out["wgrad"]["hbm_bytes"] / n_pass
    out["all_kernels_hbm_bytes_per_step"] = out["all_kernels"]["hbm_bytes"] / n_pass
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
