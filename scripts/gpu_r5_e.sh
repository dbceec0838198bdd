#!/bin/bash
# round-5 GPU call E: unconditional speculative row + straight-line sorted scan
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 600 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_parity.py -k "query or merge" tests/test_gpu_build_parity.py tests/test_gpu_fuzz.py 2>&1 | tail -15) > gpurun_out/e_tests.log 2>&1
C="QUERY_EARLY=0;QUERY_EARLY=1"
(timeout 300 python scripts/early_probe.py 1000000 128 f32 --combos "$C" 0.85:175 2>&1 | grep -v amdgpu.ids) > gpurun_out/e_probe_f32.log 2>&1
(timeout 300 python scripts/early_probe.py 1000000 128 u8 --combos "$C" 0.85:175 2>&1 | grep -v amdgpu.ids) > gpurun_out/e_probe_u8.log 2>&1
(timeout 500 python scripts/early_probe.py 12500000 96 f32 --combos "$C" 1.0:400 0.95:300 0.95:280 2>&1 | grep -v amdgpu.ids) > gpurun_out/e_probe_c4.log 2>&1
tail -5 gpurun_out/e_tests.log; cat gpurun_out/e_probe_*.log
