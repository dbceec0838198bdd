#!/bin/bash
# round-5 GPU call C: ring-less hashed set (GR) on the C4 shape + operating-point sweep there
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 600 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_parity.py -k "query" 2>&1 | tail -15) > gpurun_out/c_tests.log 2>&1
C="QUERY_EARLY=0;QUERY_GLOBAL_RING=1;QUERY_GLOBAL_RING=0;QUERY_GLOBAL_RING=1,QUERY_LDS_PAD=768"
(timeout 700 python scripts/early_probe.py 12500000 96 f32 --combos "$C" 1.0:400 0.95:300 0.9:300 1.0:300 0.95:256 1.0:256 0.9:280 0.95:280 2>&1 | grep -v amdgpu.ids) > gpurun_out/c_probe_c4.log 2>&1
tail -5 gpurun_out/c_tests.log; cat gpurun_out/c_probe_*.log
