#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python -m pytest -q -n 4 -m gpu --timeout 600 tests/test_gpu_bf_exact.py tests/test_gpu_parity.py -k "uint8 or u8" 2>&1 | tail -8) > gpurun_out/l_tests.log 2>&1
for cfg in "X=1" "GGNN_BF_I8_RANKS=18" "GGNN_BF_I8_RANKS=3" "GGNN_BF_I8_RANKS=31" "GGNN_BF_I8_REFRESH=16" "GGNN_BF_I8_REFRESH=8" "GGNN_BF_I8_REFRESH=16 GGNN_BF_I8_RANKS=3" "GGNN_BF_I8_SEED=512"; do
  echo "== $cfg" >> gpurun_out/l_i8.log
  env $cfg timeout 120 python scripts/bf_time_u8.py 2>&1 | grep bf_query >> gpurun_out/l_i8.log
done
echo "k=16 / k=4 / D=64" >> gpurun_out/l_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 16 2>&1 | grep bf_query >> gpurun_out/l_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 4 2>&1 | grep bf_query >> gpurun_out/l_i8.log
timeout 120 python scripts/bf_time_u8.py 1000000 10 64 2>&1 | grep bf_query >> gpurun_out/l_i8.log
cat gpurun_out/l_tests.log gpurun_out/l_i8.log
