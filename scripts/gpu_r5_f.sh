#!/bin/bash
# round-5 GPU call F: merged-recall operating point of the 100M x 96 base (8 shards on one GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GGNN_TEST_HOOKS=1
(timeout 900 python scripts/shard8_probe.py 12500000 96 1.0:400 0.95:280 0.9:280 0.9:256 0.85:256 0.85:200 0.8:256 0.9:200 0.8:280 0.85:280 2>&1 | grep -v amdgpu.ids) > gpurun_out/f_shard8.log 2>&1
cat gpurun_out/f_shard8.log
