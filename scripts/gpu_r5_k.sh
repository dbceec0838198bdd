#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
export GGNN_AMD_LIB=$GRAFT_REPO_ROOT/ggnn_amd/csrc/libggnn_dbg.so
for cfg in "GGNN_BF_I8_SEED=0" "GGNN_BF_I8_SEED=512" "GGNN_BF_I8_SEED=0 GGNN_BF_I8_RANKS=16" "GGNN_BF_I8_SEED=0 GGNN_BF_I8_NOSHARE=1"; do
  echo "== $cfg" >> gpurun_out/k_stats.log
  env $cfg timeout 200 python scripts/bf_i8_stats.py 2>&1 | grep -v amdgpu >> gpurun_out/k_stats.log
done
cat gpurun_out/k_stats.log
