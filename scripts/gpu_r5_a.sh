#!/bin/bash
# round-5 GPU call A: parity of the early-rows order + new tests, then the A/B timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 900 \
   tests/test_gpu_parity.py tests/test_gpu_build_parity.py tests/test_gpu_limits.py \
   "tests/test_gpu_out_of_core.py::test_out_of_core_equals_oracle" \
   "tests/test_gpu_bf_exact.py::test_bf_mfma_chunked_many_segments_per_workgroup" \
   tests/test_gpu_fuzz.py 2>&1 | tail -40) > gpurun_out/a_tests.log 2>&1
(timeout 300 python scripts/early_probe.py 1000000 128 f32 0.85:175 0.9:175 1.0:400 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_probe_f32.log 2>&1
(timeout 300 python scripts/early_probe.py 1000000 128 u8 0.85:175 1.0:400 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_probe_u8.log 2>&1
(timeout 400 python scripts/early_probe.py 12500000 96 f32 1.0:400 0.9:300 0.9:256 0.85:256 1.0:256 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_probe_c4.log 2>&1
tail -5 gpurun_out/a_tests.log; cat gpurun_out/a_probe_*.log
