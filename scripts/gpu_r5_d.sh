#!/bin/bash
# round-5 GPU call D: ring-less set on the one-register kernels (headline, u8) + operating points
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 600 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_parity.py -k "query" tests/test_gpu_fuzz.py 2>&1 | tail -15) > gpurun_out/d_tests.log 2>&1
C="QUERY_EARLY=0;QUERY_GLOBAL_RING=1;QUERY_GLOBAL_RING=0"
(timeout 500 python scripts/early_probe.py 1000000 128 f32 --combos "$C" 0.85:175 0.9:175 0.85:192 0.8:192 0.9:257 0.95:257 1.0:257 0.9:280 0.95:280 1.0:280 1.0:300 2>&1 | grep -v amdgpu.ids) > gpurun_out/d_probe_f32.log 2>&1
(timeout 400 python scripts/early_probe.py 1000000 128 u8 --combos "$C" 0.85:175 0.95:257 1.0:257 0.95:280 1.0:280 2>&1 | grep -v amdgpu.ids) > gpurun_out/d_probe_u8.log 2>&1
tail -5 gpurun_out/d_tests.log; cat gpurun_out/d_probe_*.log
