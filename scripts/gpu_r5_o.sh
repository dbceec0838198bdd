#!/bin/bash
# round-5 GPU call O: the driver's N = 8 command, dry-run with all 8 ranks on the one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 --steps 10 --warmup 2 --single-device > gpurun_out/o_n8.json 2> gpurun_out/o_n8.err)
echo rc=$?
tail -5 gpurun_out/o_n8.err
python - <<'PY'
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/o_n8.json") if l.startswith("{")][-1]
    print({k:d.get(k) for k in ("value","ms_per_step","n_gpus","recall_at_10","exchange","speedup_vs_one_gpu_same_base")})
    print("one", {k:d["one_gpu_same_base"].get(k) for k in ("queries_per_s","ms_per_step","recall_at_10")} if d.get("one_gpu_same_base") else None)
    print("inproc", d.get("in_process_handle"))
except Exception as e:
    print("no line", e)
PY
