#!/bin/bash
# round-5 GPU call Q: the whole suite once more, then every profile of the round and the default line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/profiles_r05
export GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -q -m gpu -n 4 --dist loadfile --timeout 1200 2>&1 | tail -12) > gpurun_out/q_tests.log 2>&1
bash scripts/profile_round.sh r05 > gpurun_out/q_prof_head.log 2>&1
bash scripts/profile_round.sh r05 u8 --dtype u8 > gpurun_out/q_prof_u8.log 2>&1
bash scripts/profile_round.sh r05 d96 --n-base 12500000 --dim 96 > gpurun_out/q_prof_d96.log 2>&1
bash scripts/profile_round.sh r05 d960cos --dim 960 --measure cosine --tau-query 0.85 --max-iters 175 > gpurun_out/q_prof_d960.log 2>&1
cd "$GRAFT_REPO_ROOT"
cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null
(timeout 900 python bench.py > gpurun_out/profiles_r05/r05_bench_n1.json 2> gpurun_out/q_bench.err)
cat gpurun_out/q_tests.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/profiles_r05/r05_bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step","recall_at_10")}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "valu", d["roofline"]["secondary"]["valu_issue"].get("valu_insts_per_pop"))
print({k:(v["at_recall_0.99"] or {}).get("queries_per_s") for k,v in d["recall_targets"]["results"].items()})
print({k:d["build"][k]["roofline"]["bound"]+" %.3f"%d["build"][k]["roofline"]["frac"] for k in ("merge_kernel","sym_kernel")})
PY
