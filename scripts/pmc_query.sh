#!/bin/bash
# SQ counters of the query kernel (two passes); run on the GPU box from the repo root.
# usage: scripts/pmc_query.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
            "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o pmc -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$tag/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_$tag/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "query_kernel" in k and "bf_" not in k:
            acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"  {name:24s} mean/launch {sum(v)/len(v):.4g}  (n={len(v)})")
PY
