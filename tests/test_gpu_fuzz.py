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
