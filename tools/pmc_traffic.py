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


def one_pass(counter, mode, steps, warmup, config="c1p", batch=32, dtype="bf16"):
    d = f"/tmp/pmc_traffic_{counter}_{mode}_{config}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["timeout", "600", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--mode", mode, "--steps", str(steps), "--warmup", str(warmup),
           "--config", config, "--batch", str(batch), "--dtype", dtype, "--no-graph", "--no-cpu-baseline", "--no-roofline", "--no-f32", "--no-eval-leg"]
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


def measure(mode="train", config="c1p", batch=32, dtype="bf16", steps=1, warmup=1):
    """Per-step HBM bytes of bench.py's step: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH x 2 (gfx950), totals divided by
    the passes the profiled process executed (bench.py --no-graph: one eager warm-up pass + warmup + steps)."""
    fetch = one_pass("FETCH_SIZE", mode, steps, warmup, config, batch, dtype)
    write = one_pass("WRITE_SIZE", mode, steps, warmup, config, batch, dtype)

    def total(agg, pred):
        return sum(v[1] for k, v in agg.items() if pred(k)), sum(v[0] for k, v in agg.items() if pred(k))
    is_igemm = lambda k: "igemm_kernel" in k or "igemm_big_kernel" in k
    is_wgrad = lambda k: "wgrad_kernel" in k or "wgrad_group_kernel" in k or "wgrad_reduce" in k
    everything = lambda k: True
    n_pass = steps + warmup + 1
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only), bench.py --mode {mode} "
                     f"--config {config} --no-graph --steps {steps} --warmup {warmup}; per-step = totals / {n_pass} executed passes; "
                     f"tools/pmc_traffic.py",
           "dtype": dtype, "batch": batch, "config": config,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both in KB"}
    per = lambda kb: kb * 1024.0
    for name, pred in (("igemm", is_igemm), ("wgrad", is_wgrad), ("all_kernels", everything)):
        f, n = total(fetch, pred)
        w, _ = total(write, pred)
        out[name] = {"launches": n, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "hbm_bytes": per(2.0 * f + w)}
    out["passes"] = n_pass
    out["launches_per_step"] = out["igemm"]["launches"] / n_pass
    out["all_launches_per_step"] = out["all_kernels"]["launches"] / n_pass
    out["hbm_bytes_per_step"] = out["igemm"]["hbm_bytes"] / n_pass
    out["hbm_bytes_per_launch"] = out["igemm"]["hbm_bytes"] / max(1, out["igemm"]["launches"])
    out["wgrad_hbm_bytes_per_step"] = out["wgrad"]["hbm_bytes"] / n_pass
    out["all_kernels_hbm_bytes_per_step"] = out["all_kernels"]["hbm_bytes"] / n_pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train")
    ap.add_argument("--config", default="c1p")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    txt = json.dumps(measure(a.mode, a.config, a.batch, "bf16", a.steps, a.warmup), indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
