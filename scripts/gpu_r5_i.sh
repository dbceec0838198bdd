#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_bf_exact.py tests/test_gpu_parity.py -k "uint8 or u8" 2>&1 | tail -8) > gpurun_out/i_tests.log 2>&1
for e in 16 32 64 128 256; do
  echo "default ranks, BF_I8_REFRESH=$e" >> gpurun_out/i_i8.log
  GGNN_BF_I8_REFRESH=$e timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/i_i8.log
done
echo "ranks 18 (p=1 and p=9)" >> gpurun_out/i_i8.log
GGNN_BF_I8_RANKS=18 timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/i_i8.log
echo "k=16 / k=4 / D=64 / D=96" >> gpurun_out/i_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 16 2>&1 | grep bf_query >> gpurun_out/i_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 4 2>&1 | grep bf_query >> gpurun_out/i_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/i_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 10 96 2>&1 | grep bf_query >> gpurun_out/i_i8.log
cat gpurun_out/i_tests.log gpurun_out/i_i8.log
