#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_*; run on the GPU box from the repo root:
#   scripts/profile_round.sh r03                      headline workload, every pass
#   scripts/profile_round.sh r03 u8 --dtype u8        another BASELINE shape: kernel stats +
#                                                     FETCH/WRITE + SQ passes -> profiles/r03_u8_*
# The profiled command is `python bench.py --lean [args]` (the headline measurement only: the
# default command also runs other datasets and the multi-shard point through the SAME kernel
# template, which would mix workloads in the per-kernel averages).  Counter passes never combine
# --pmc with the API-trace domains (see the task statement); each --pmc set is its own run.
tag=${1:-r03}; shift
sub=""
if [ $# -gt 0 ]; then sub=$1; shift; fi
extra="$*"
R=$GRAFT_REPO_ROOT
name=$tag${sub:+_$sub}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$name
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name/prof -o bench -- \
  python $R/bench.py --lean --steps 20 --warmup 3 $extra > $R/gpurun_out/$name/bench_prof.json 2> $R/gpurun_out/$name/bench_prof.err
short="python $R/bench.py --lean --steps 3 --warmup 1 $extra"
pass() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name/pmc_$n -o bench -- $short > /dev/null 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
pass sq2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
if [ -z "$sub" ]; then
  pass l2a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass l2b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum
  pass mfma SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
fi
cd $R && tail -1 gpurun_out/$name/bench_prof.json > gpurun_out/$name/bench_prof.tmp && mv gpurun_out/$name/bench_prof.tmp gpurun_out/$name/bench_prof.json
python scripts/summarize_profile.py $name gpurun_out/$name "$extra"
