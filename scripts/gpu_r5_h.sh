#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_bf_exact.py tests/test_gpu_parity.py -k "uint8 or u8 or bf_" 2>&1 | tail -12) > gpurun_out/h_tests.log 2>&1
for r in -1 16 2 3 1 4 8; do
  echo "BF_I8_RANKS=$r" >> gpurun_out/h_i8.log
  GGNN_BF_I8_RANKS=$r timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/h_i8.log
done
echo "NOSHARE" >> gpurun_out/h_i8.log
GGNN_BF_I8_NOSHARE=1 timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/h_i8.log
echo "k=16 / k=4 / D=64" >> gpurun_out/h_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 16 2>&1 | grep bf_query >> gpurun_out/h_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 4 2>&1 | grep bf_query >> gpurun_out/h_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/h_i8.log
cat gpurun_out/h_tests.log gpurun_out/h_i8.log
