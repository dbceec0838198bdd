import os
import sys

import numpy as np
import pytest

# The library reads its GGNN_<HOOK> environment variables only while this master switch is set
# (include/ggnn_c.h, "Test and tuning hooks"): the tests that steer a hook through monkeypatch.setenv
# rely on it; tests/test_cabi.py checks that without the switch the environment is ignored.
os.environ["GGNN_TEST_HOOKS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build_lib()
    return oracle


@pytest.fixture(autouse=True)
def _oracle_modes_reset():
    """a test that switches the oracle's float summation order never leaks it into the next one"""
    yield
    mod = sys.modules.get("oracle.oracle")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod.set_wave_order(False)


def make_int_data(N, D, seed):
    """S-int of SURVEY 8(d): integers in [0,255] stored as f32 -> every fp32 sum is exact."""
    return np.random.default_rng(seed).integers(0, 256, (N, D)).astype(np.float32)


def make_uni_data(N, D, seed):
    """S-uni: uniform [0,1) f32 (tolerance track)."""
    return np.random.default_rng(seed).random((N, D), dtype=np.float32)


@pytest.fixture(scope="session")
def small_graph(orc):
    """A complete oracle-built graph on integer-valued data, shared by the GPU parity tests."""
    N, D, K = 2048, 128, 24
    base = make_int_data(N, D, 1234)
    rng = orc.make_rng(N, 99)
    cfg, graph, tr, sel, stats = orc.build(base, K, 0.5, 1, rng=rng)
    return dict(N=N, D=D, K=K, base=base, cfg=cfg, graph=graph, tr=tr, sel=sel, stats=stats,
                rng=rng)
