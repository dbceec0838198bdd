#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_bf_exact.py tests/test_gpu_parity.py -k "uint8 or u8" 2>&1 | tail -8) > gpurun_out/j_tests.log 2>&1
for e in 0 128 256 512 1024 2048; do
  echo "BF_I8_SEED=$e" >> gpurun_out/j_i8.log
  GGNN_BF_I8_SEED=$e timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/j_i8.log
done
echo "k=16 / k=4 / D=64 / D=64 seed 0 / D=64 noshare seed 0" >> gpurun_out/j_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 16 2>&1 | grep bf_query >> gpurun_out/j_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 4 2>&1 | grep bf_query >> gpurun_out/j_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/j_i8.log
GGNN_BF_I8_SEED=0 timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/j_i8.log
GGNN_BF_I8_SEED=0 GGNN_BF_I8_NOSHARE=1 timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/j_i8.log
cat gpurun_out/j_tests.log gpurun_out/j_i8.log
