"""bench.py's secondary configurations keep working end to end on one GPU (small batches, a couple of steps): BASELINE config
#1's plumbing (OS8, 22 classes), config #5's clip-shaped CE + ContrastLoss step, the deterministic mode and the CPU-baseline leg's
thread sweep (on a tiny sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert "capture failed" not in r.stderr
    return json.loads(lines[0])


def test_config_c1_train_line():
    d = _bench("--config", "c1", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-f32")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["dtype"] == "bf16"
    assert "OS8" in d["config"]["workload"] and "num_classes=22" in d["config"]["workload"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["step"]["fused_min_gb"] > 0
    assert d["config"]["launch"] == "hipGraph replay"


def test_config_c5_clip_step_line():
    d = _bench("--config", "c5", "--batch", "10", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["value"] > 0 and "ContrastLoss" in d["config"]["workload"] and "2 clips x 5 frames" in d["config"]["workload"]
    assert "hipGraph" in d["config"]["launch"]          # the model's part on the graphed autograd node by default
    assert d["roofline"]["step"]["algorithmic_gflop"] > 0 and 0 < d["roofline"]["step"]["frac_of_hbm_peak"] < 1
    e = _bench("--config", "c5", "--batch", "10", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph")
    assert e["value"] > 0 and "eager autograd node" in e["config"]["launch"]
    with pytest.raises(AssertionError):
        _bench("--config", "c5", "--batch", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")   # not a multiple of 5 frames


def test_deterministic_line_and_cpu_thread_sweep():
    d = _bench("--deterministic", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-f32", "--no-roofline", "--cpu-sample-batch", "2")
    assert d["config"]["deterministic"] is True and d["value"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and str(cb["cores"]) in cb["thread_sweep_frames_per_s"]
    assert cb["value"] >= 0.5 * max(cb["thread_sweep_frames_per_s"].values())   # the reported figure is the sweep's best count, re-timed


def test_trainer_loop_lines():
    """--trainer-loop: the reference trainer's call sequence through the graphed autograd node and through the eager one."""
    g = _bench("--trainer-loop", "--batch", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-f32")
    # launches inside graph replays: no per-launch view, but the whole step against both roofs (round 5)
    assert g["value"] > 0 and "hipGraph" in g["config"]["launch"] and set(g["roofline"]) == {"step"}
    st = g["roofline"]["step"]
    assert st["algorithmic_gflop"] > 0 and st["fused_min_gb"] > 0 and 0 < st["frac_of_hbm_peak"] < 1 and 0 < st["frac_of_mfma_peak"] < 1
    e = _bench("--trainer-loop", "--no-graph", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-f32")
    assert e["value"] > 0 and "eager autograd node" in e["config"]["launch"] and e["roofline"]["frac"] > 0
