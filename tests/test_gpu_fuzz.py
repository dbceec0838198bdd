"""Randomised end-to-end parity: 40 random (element type, N, D incl. padded rows, Nq, K, KBuild,
tau, iterations, pre-screen) configurations through the ggnn surface; bf_query and query on the
GPU-built graph must equal the oracle bit for bit on integer-valued data (tests/tools/
fuzz_engine.py; other seeds: `python tests/tools/fuzz_engine.py 80 <seed>`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_engine_against_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_engine.py"), "20",
                        str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 19
