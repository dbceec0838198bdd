#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_*; run on the GPU box from the repo root:
#   scripts/profile_round.sh r01
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- \
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
short="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -o bench -- $short > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$tag -o bench -- $short > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv \
  -d $R/gpurun_out/pmc_sq1_$tag -o bench -- $short > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv \
  -d $R/gpurun_out/pmc_sq2_$tag -o bench -- $short > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
  -d $R/gpurun_out/pmc_mfma_$tag -o bench -- $short > /dev/null 2>&1
cd $R && tail -1 gpurun_out/bench_prof.json > gpurun_out/bench_prof.tmp && mv gpurun_out/bench_prof.tmp gpurun_out/bench_prof.json
find gpurun_out -name "*_counter_collection.csv" -o -name "*kernel_stats.csv" | head -20
python scripts/summarize_profile.py $tag
