"""SURVEY.md section 8b "wrappers it must survive": DistributedDataParallel(find_unused_parameters=True) over SyncBatchNorm on
a one-rank RCCL group and nn.DataParallel(device_ids=["cuda:0"]) around the model, trainer call sequence, two iterations, eager
and graphed autograd nodes (tests/_wrappers_world1.py; main_vpo_mono.py:127-144, trainer_cavp_vpo_mono.py:166-193)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_and_dataparallel_wrappers_match_the_bare_model():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_wrappers_world1.py")], cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "WRAPPERS_OK" in r.stdout, r.stdout[-1500:]
    print(r.stdout.strip())
