#!/bin/bash
# round-5 GPU call M: the whole -m gpu suite (4 workers, one file per worker at a time) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -n 4 --dist loadfile --timeout 1200 2>&1 | tail -25) > gpurun_out/m_tests.log 2>&1
(timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/m_smoke.log 2>&1
cat gpurun_out/m_tests.log gpurun_out/m_smoke.log
