#!/bin/bash
# SQ counters of the bf_mfma kernels; run on the GPU box from the repo root:
#   scripts/pmc_bf.sh <tag> <timing script> [args]     e.g.  scripts/pmc_bf.sh d960 scripts/bf_time.py 960
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_bf_$tag
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bf_$tag/p$i -o pmc -- python $R/"$@" > $R/gpurun_out/pmc_bf_$tag/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_bf_$tag/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bf_mfma" in k or "bf_i8v2" in k:
            k = k.split("(")[0] + " grid=" + row.get("Grid_Size", "?")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row:
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
out = {}
for k, c in acc.items():
    out[k] = {n: sum(v) / len(v) for n, v in c.items()}
    if dur[k]:
        out[k]["kernel_duration_s_under_profiling"] = sum(dur[k]) / len(dur[k])
    print(k)
    for name, v in sorted(out[k].items()):
        print(f"  {name:40s} {v:.5g}")
json.dump(out, open("$R/gpurun_out/pmc_bf_$tag/summary.json", "w"), indent=1)
PY
tail -2 $R/gpurun_out/pmc_bf_$tag/p3.log
