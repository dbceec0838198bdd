#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_*; run on the GPU box from the repo root:
#   scripts/profile_round.sh r02
# The profiled command is `python bench.py --lean` (the headline measurement only: the default
# command also runs other datasets and the 8-shard series point through the SAME kernel template,
# which would mix workloads in the per-kernel averages).  Counter passes never combine --pmc with
# the API-trace domains (see the task statement); each --pmc set is its own run.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -o bench -- \
  python $R/bench.py --lean --steps 20 --warmup 3 > $R/gpurun_out/$tag/bench_prof.json 2> $R/gpurun_out/$tag/bench_prof.err
short="python $R/bench.py --lean --steps 3 --warmup 1"
pass() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_$n -o bench -- $short > /dev/null 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
pass sq2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pass l2a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass l2b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum
pass mfma SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
cd $R && tail -1 gpurun_out/$tag/bench_prof.json > gpurun_out/$tag/bench_prof.tmp && mv gpurun_out/$tag/bench_prof.tmp gpurun_out/$tag/bench_prof.json
python scripts/summarize_profile.py $tag gpurun_out/$tag
