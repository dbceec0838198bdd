#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 1200 python -m pytest -q -m gpu --timeout 900 tests/test_gpu_bench_modes.py tests/test_gpu_distributed.py 2>&1 | tail -30) > gpurun_out/g_tests.log 2>&1
(timeout 600 python scripts/shard8_probe.py 12500000 96 0.75:280 0.7:280 0.75:320 0.7:350 0.8:300 0.8:270 2>&1 | grep -v amdgpu.ids) > gpurun_out/g_shard8.log 2>&1
cat gpurun_out/g_tests.log gpurun_out/g_shard8.log
