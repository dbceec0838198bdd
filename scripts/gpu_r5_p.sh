#!/bin/bash
# round-5 GPU call P: ring-less tag set + early rows on long searches; uint8 bf counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_parity.py -k "query" tests/test_gpu_limits.py tests/test_gpu_fuzz.py 2>&1 | tail -8) > gpurun_out/p_tests.log 2>&1
C="QUERY_EARLY=0;QUERY_EARLY=1"
(timeout 400 python scripts/early_probe.py 1000000 128 f32 --kind lowrank24 --combos "$C" 1.0:600 1.0:800 1.0:1000 2>&1 | grep -v amdgpu.ids) > gpurun_out/p_probe_lr24.log 2>&1
(timeout 400 python scripts/early_probe.py 1000000 128 f32 --kind lowrank32 --combos "$C" 1.0:2000 1.1:2000 2>&1 | grep -v amdgpu.ids) > gpurun_out/p_probe_lr32.log 2>&1
unset GGNN_TEST_HOOKS
bash scripts/pmc_bf.sh u8 scripts/bf_time_u8.py > gpurun_out/p_pmc_u8.log 2>&1
cd "$GRAFT_REPO_ROOT"
cat gpurun_out/p_tests.log gpurun_out/p_probe_lr24.log gpurun_out/p_probe_lr32.log; tail -30 gpurun_out/p_pmc_u8.log
