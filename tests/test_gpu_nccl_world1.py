"""RCCL itself on the single-GPU test box: the gradient collectives of the data-parallel step on backend "nccl" with one rank
(see tests/_nccl_world1.py).  Multi-GPU runs are the driver's; this pins init + collectives + graph interplay."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_backend_single_rank_training_step():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_nccl_world1.py")], cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "NCCL_WORLD1_OK backend=nccl" in r.stdout, r.stdout[-1500:]
    print(r.stdout.strip().splitlines()[-1])
