"""Runs last (file name): the opt-in HIP-graph replay of fp_search (FP_GRAPH=1)."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fptest_env import with_test_opts  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cap_pct", ["", "50"])
def test_graph_replay_equals_trace(cap_pct):
    """FP_GRAPH=1: from the third batch of a shape on, fp_search is one hipGraphLaunch; results must equal the traces.  With
    FP_TEST=spec_cap_pct=50 every speculative batch overflows, so no graph ever becomes valid and the fallbacks are what runs."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FP_GRAPH="1")
    if cap_pct:
        env = with_test_opts(env, spec_cap_pct=cap_pct)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "graph_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout + r.stderr
