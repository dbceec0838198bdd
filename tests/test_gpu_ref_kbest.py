"""Pins SURVEY row B (KBestList) against the REFERENCE's own device code: oracle/_ref/
libggnn_ref_kbest.so is include/ggnn/cuda_utils/k_best_list.cuh of the reference compiled
unchanged for gfx950 (oracle/ref_kbest_harness.hip).  The oracle's lockstep emulation must
reproduce it bit for bit on random (distance, id) streams with ties, for the block sizes the
reference instantiates; bf_query and top inherit the pin through their own oracle parity tests."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ref(orc):
    if not os.path.exists(orc.REF_KBEST_SO):
        pytest.skip("oracle/_ref was not built (reference tree not mounted at build time)")
    lib = C.CDLL(orc.REF_KBEST_SO)
    lib.ref_kbest_script.restype = C.c_int
    lib.ref_kbest_script.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_int, C.c_void_p, C.c_void_p]
    return lib


def run_ref(ref, block, best, dists, ids, check_worst):
    d_d = torch.from_numpy(dists).cuda()
    d_i = torch.from_numpy(ids).cuda()
    o_d = torch.empty(best, dtype=torch.float32, device="cuda")
    o_i = torch.empty(best, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = ref.ref_kbest_script(block, best, d_d.data_ptr(), d_i.data_ptr(), dists.size,
                              int(check_worst), o_d.data_ptr(), o_i.data_ptr())
    assert rc == 0
    return o_d.cpu().numpy(), o_i.cpu().numpy()


@pytest.mark.parametrize("block", [32, 64, 128, 256])
@pytest.mark.parametrize("best", [1, 2, 10, 24, 33, 100, 300])
@pytest.mark.parametrize("check_worst", [True, False])
def test_oracle_kbest_equals_reference_device_code(orc, ref, block, best, check_worst):
    r = np.random.default_rng(block * 1000 + best)
    for n, hi in ((5, 4), (200, 20), (1500, 1000)):
        dists = r.integers(0, hi, n).astype(np.float32)      # many exact ties
        ids = r.integers(0, 10**6, n).astype(np.int32)
        rd, ri = run_ref(ref, block, best, dists, ids, check_worst)
        od, oi = orc.kbest_script(best, block, dists, ids, check_worst)
        assert np.array_equal(rd, od), (block, best, n)
        assert np.array_equal(ri, oi), (block, best, n)


def test_engine_bf_query_equals_reference_kbest_order(orc, ref):
    """end to end: the engine's bf_query list == the reference KBestList fed with the exact
    distances in base order (ties keep the lower index first)."""
    from ggnn_amd import ops
    r = np.random.default_rng(5)
    base = r.integers(0, 6, (4000, 8)).astype(np.float32)      # tiny value range -> many ties
    q = r.integers(0, 6, (3, 8)).astype(np.float32)
    ids, d = ops.bf_query(torch.from_numpy(base).cuda(), torch.from_numpy(q).cuda(), 17)
    for n in range(3):
        dists = ((base - q[n]) ** 2).sum(1).astype(np.float32)
        rd, ri = run_ref(ref, 32, 17, dists, np.arange(4000, dtype=np.int32), True)
        assert np.array_equal(d.cpu().numpy()[n], rd)
        assert np.array_equal(ids.cpu().numpy()[n], ri)
