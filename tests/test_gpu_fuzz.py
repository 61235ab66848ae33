"""Seeded random shapes through the conv forward / data gradient / weight gradient entry points (tools/fuzz_conv.py) against PyTorch
on the CPU: ragged extents, channel tails, strides, dilations, fused epilogues, plus shapes that reach the 256x256 tile, split-K and
the deep-ring plans."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["--cases", "60", "--seed", "11"], ["--cases", "16", "--seed", "12", "--large"]], ids=["small", "large"])
def test_fuzz_conv(args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_conv.py")] + args, cwd=REPO, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and ", 0 bad" in r.stdout, tail


def test_fuzz_model_vs_oracle():
    """Random model-level configurations (odd image extents such as 33 x 112 or 70 x 81, 2 .. 71 classes, the dilation variants):
    eval forward within 1e-3 of the CPU oracle, training loss within 2e-4, every parameter gradient's cosine >= 0.97."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_fuzz_model.py"), "--cases", "3", "--seed", "5"], cwd=REPO,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    tail = "\n".join(r.stdout.splitlines()[-8:])
    assert r.returncode == 0 and ", 0 bad" in r.stdout, tail
    print(tail)
