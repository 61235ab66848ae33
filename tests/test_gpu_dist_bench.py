"""The data-parallel bench path end to end on ONE GPU: two ranks share cuda:0 and talk over gloo
(CAVP_BENCH_SHARE_GPU=1), which exercises everything RCCL runs would - process-group init, the two-graph capture with a
live process group, the asynchronous early all-reduce + late all-reduce around the replays, barriers, max-over-ranks
timing and the single JSON line - except the RCCL transport itself (multi-GPU boxes are the driver's)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,launcher,wire", [("train", "torchrun", "f32"), ("eval", "torchrun", "f32"), ("train", "self", "bf16")])
def test_two_rank_bench_on_one_gpu(mode, launcher, wire):
    env = dict(os.environ, CAVP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--mode", mode, "--grad-allreduce", wire]
    if launcher == "torchrun":   # as the driver runs N > 1; default flags: the roofline leg must stay rank-local
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(REPO, "bench.py")] + args
    else:                        # plain `python bench.py --gpus 2`: bench.py starts its own two ranks (main_vpo_mono.py:274-275)
        cmd = [sys.executable, os.path.join(REPO, "bench.py")] + args
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["config"]["ranks"] == 2
    assert "capture failed" not in r.stderr
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d   # per-kernel timing on rank 0 only; CPU baseline is an N=1 leg
    if mode == "train":   # the two gradient collectives, timed alone, in the line (rccl_ranks = ranks of the process group)
        c = d["collective"]
        assert c["rccl_ranks"] == 2 and c["wire_dtype"] == wire and c["early_piece_ms"] > 0 and c["late_piece_ms"] > 0
        assert abs(c["early_piece_mb"] + c["late_piece_mb"] - (240 if wire == "bf16" else 479)) < 3
    else:
        assert "collective" not in d


def test_gpus_flag_must_match_the_launcher():
    """--gpus N under a launcher that started a different number of ranks is an error, not a silent N = 1 run."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=REPO, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr
