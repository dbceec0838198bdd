#!/bin/bash
# What is run on the GPU box before a round closes (one gpurun call, ~7 minutes):
#   the whole -m gpu suite (4 workers), every rocprofv3 pass behind profiles/<tag>_* for the four
#   BASELINE shapes, and the default `python bench.py` line.  The summaries land in
#   gpurun_out/profiles_<tag>/ (gpurun only merges gpurun_out/ back): copy them into profiles/.
#   The counter passes carry a fingerprint of the kernel sources: collect them AFTER the last
#   change to traversal.hpp / query.hip / merge.hip / sym.hip, or bench.py ignores them.
#     gpurun --timeout 3000 -- 'bash scripts/round_end.sh r06'
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r06}
# optional second argument: the test files to run instead of the whole -m gpu suite (when only a
# scheduling detail of a kernel changed after the last full run and GPU minutes are short)
sel=${2:-tests}
mkdir -p gpurun_out/profiles_$tag
export GRAFT_REPO_ROOT
(timeout ${TEST_TIMEOUT:-1500} python -m pytest $sel -q -m gpu ${PYTEST_DIST:--n 4 --dist loadfile} --timeout 1200 2>&1 | tail -12) > gpurun_out/q_tests.log 2>&1
bash scripts/profile_round.sh $tag > gpurun_out/q_prof_head.log 2>&1
bash scripts/profile_round.sh $tag u8 --dtype u8 > gpurun_out/q_prof_u8.log 2>&1
bash scripts/profile_round.sh $tag d96 --n-base 12500000 --dim 96 > gpurun_out/q_prof_d96.log 2>&1
bash scripts/profile_round.sh $tag d960cos --dim 960 --measure cosine --tau-query 0.85 --max-iters 175 > gpurun_out/q_prof_d960.log 2>&1
# the long-search regime (harder synthetic bases at their recall-0.99 points: the ring-less tag-set kernels)
bash scripts/profile_round.sh $tag lr24 --dataset lowrank24 --tau-query 1.0 --max-iters 750 > gpurun_out/q_prof_lr24.log 2>&1
bash scripts/profile_round.sh $tag lr32 --dataset lowrank32 --tau-query 1.1 --max-iters 2000 > gpurun_out/q_prof_lr32.log 2>&1
cd "$GRAFT_REPO_ROOT"
cp profiles/${tag}_* gpurun_out/profiles_$tag/ 2>/dev/null
(timeout 900 python bench.py > gpurun_out/profiles_$tag/${tag}_bench_n1.json 2> gpurun_out/q_bench.err)
cat gpurun_out/q_tests.log
python - <<PY
import json
d=json.load(open("gpurun_out/profiles_$tag/${tag}_bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step","recall_at_10")}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "valu", d["roofline"]["secondary"]["valu_issue"].get("valu_insts_per_pop"))
print({k:(v["at_recall_0.99"] or {}).get("queries_per_s") for k,v in d["recall_targets"]["results"].items()})
print("value_at_lid21", d.get("value_at_lid21"))
print({k:d["build"][k]["roofline"]["bound"]+" %.3f"%d["build"][k]["roofline"]["frac"] for k in ("merge_kernel","sym_kernel")})
PY
