#!/bin/bash
# A/B of experimental builds of the library on ONE GPU box (kernel times differ by 1-2 % between
# boxes, so variants are only compared inside one call, alternating, REPS times):
#     gpurun -- 'bash scripts/ab_libs.sh "<n> <dim> <f32|u8>" <tau:iters> REPS libA.so libB.so ...'
# Libraries are looked up in ggnn_amd/csrc/ (built with `make TARGET=... OBJDIR=... EXTRA=...`);
# GGNN_AMD_LIB is honoured under GGNN_TEST_HOOKS=1 only (ggnn_amd/_lib.py).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
shape=$1; point=$2; reps=$3; shift 3
export GGNN_TEST_HOOKS=1
for r in $(seq 1 "$reps"); do
  for lib in "$@"; do
    GGNN_AMD_LIB=$PWD/ggnn_amd/csrc/$lib timeout 120 python scripts/early_probe.py $shape --combos "QUERY_EARLY=1" $point 2>&1 \
      | grep -E '^\{|build [0-9.]+ s$' | python -c "
import sys, json
lib = '$lib'
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = [v for v in d.values() if isinstance(v, dict) and 'ms10k' in v][0]
        print(f'{lib:28s} ms10k {k[\"ms10k\"]:.4f} ms100k {k[\"ms100k\"]:.3f} recall {d[\"recall\"][\"tune\"]:.4f} identical {d[\"identical\"]}')
    else:
        print(f'{lib:28s} {line.strip().split(\", \")[-1]}')
"
  done
done
