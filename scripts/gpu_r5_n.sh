#!/bin/bash
# round-5 GPU call N: every rocprofv3 pass behind profiles/r05_* (four shapes) + the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/profiles_r05
export GRAFT_REPO_ROOT
bash scripts/profile_round.sh r05 > gpurun_out/n_prof_head.log 2>&1
bash scripts/profile_round.sh r05 u8 --dtype u8 > gpurun_out/n_prof_u8.log 2>&1
bash scripts/profile_round.sh r05 d96 --n-base 12500000 --dim 96 > gpurun_out/n_prof_d96.log 2>&1
bash scripts/profile_round.sh r05 d960cos --dim 960 --measure cosine --tau-query 0.85 --max-iters 175 > gpurun_out/n_prof_d960.log 2>&1
cd "$GRAFT_REPO_ROOT"
(timeout 900 python bench.py > gpurun_out/profiles_r05/r05_bench_n1.json 2> gpurun_out/n_bench.err)
cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null
ls -la gpurun_out/profiles_r05 | head -50
tail -3 gpurun_out/n_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/profiles_r05/r05_bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step","recall_at_10")}, d["roofline"]["frac"], d["roofline"].get("kernel_ms"))
PY
