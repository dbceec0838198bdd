#!/bin/bash
# SQ counters of the bf_mfma kernel; run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_bf
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_WAVE32_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_bf/p$i -o pmc -- python $R/scripts/bf_time.py > $R/gpurun_out/pmc_bf/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_bf/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bf_mfma_kernel" in k:
            acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"  {name:32s} mean/launch {sum(v)/len(v):.4g}  (n={len(v)})")
PY
tail -2 $R/gpurun_out/pmc_bf/p3.log
