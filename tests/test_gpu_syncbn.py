"""SyncBatchNorm (main_vpo_mono.py:130) end to end on ONE GPU: two ranks share cuda:0 over gloo, each trains on half the batch
with the converted model; result == the single-process full-batch step (tests/_syncbn_two_rank.py).  One (mean, M2) exchange
per BatchNorm layer in the forward, one sum exchange in the backward."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_syncbn_two_ranks_equal_full_batch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "_syncbn_two_rank.py")]
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    key = [ln for ln in (r.stdout + r.stderr).splitlines() if "SYNCBN_OK" in ln or "AssertionError" in ln]
    assert r.returncode == 0, (key, r.stderr[-3000:])
    assert "SYNCBN_OK" in r.stdout, r.stdout[-1500:]
    assert "SYNCBN_MISMATCH_OK" in r.stdout, r.stdout[-1500:]
    print([ln for ln in r.stdout.splitlines() if "SYNCBN_OK" in ln][-1])
