"""Run tests/test_optim.py::test_two_train_steps_vs_reference_weights N times with the second stream on / off and report the pass
rate (the test's second-step band is a noise band by construction; this separates noise from a stream race)."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cavp_amd.train as TR
from tests import test_optim as TO

for side in (True, False, True, False):
    TR._SIDE_STREAM = side
    ok = 0
    fails = []
    for i in range(6):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                TO.test_two_train_steps_vs_reference_weights()
            ok += 1
        except AssertionError as e:
            fails.append(str(e)[:120])
    print(f"side_stream={side}: {ok}/6 passed", fails)
