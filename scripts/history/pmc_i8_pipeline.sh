#!/bin/bash
# One SQ counter pass of the uint8 brute-force kernel for the library given in $1 (the product
# library, or an experiment build such as `make OBJDIR=build_exp TARGET=libggnn_exp.so
# EXTRA=-DGGNN_I8_EXP=1`: the tile pipeline without hit handling).  Run from the repo root:
#   scripts/pmc_i8_pipeline.sh <tag> [library]
tag=$1; lib=${2:-}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_i8_$tag
[ -n "$lib" ] && export GGNN_TEST_HOOKS=1 GGNN_AMD_LIB=$R/$lib
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_i8_$tag/p -o pmc -- python $R/scripts/bf_time_u8.py > $R/gpurun_out/pmc_i8_$tag/log 2>&1
tail -1 $R/gpurun_out/pmc_i8_$tag/log
