"""Dry runs of every bench.py mode on the one-GPU test box, at toy sizes:
  * `--gpus 2` under torch.distributed.run (two gloo ranks, both on GPU 0): the fixed-base series
    with blocking / saturated / pipelined figures, the one-GPU point of the same base, the
    one-handle `--in-process` child (in-engine exchange) and the secondary base;
  * the lean N = 1 line: `roofline` is the HBM own-bytes fraction, `cpu_baseline`
    and the secondaries have their contracted shape.
These check plumbing and JSON contracts, not performance."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_dry_run():
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2",
               "--steps", "4", "--warmup", "1", "--backend", "gloo", "--single-device",
               "--n-base", "20000", "--secondary-n-base", "10000", "--n-query", "400",
               "--in-process-timeout", "300"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["exchange"]["requested"] == "gloo" and out["exchange"]["fallback"] is False
    assert "160000 points" in out["config"]["workload"]
    assert out["recall_at_10"] > 0.9
    rl = out["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 16000.0 and 0 < rl["frac"] < 1
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    one = out["one_gpu_same_base"]
    assert one["queries_per_s"] > 0 and "saturated_batch" in one and "pipelined_batches" in one
    for key in ("speedup_vs_one_gpu_same_base", "saturated_speedup_vs_one_gpu_same_base",
                "pipelined_speedup_vs_one_gpu_same_base"):
        assert out[key] and out[key] > 0, key
    assert out["pipelined_batches"]["results_equal_blocking"] is True
    inproc = out["in_process_handle"]
    assert "error" not in inproc, inproc
    assert inproc["exchange"] in ("copy", "rccl") and inproc["queries_per_s"] > 0
    assert inproc["pipelined_batches"]["results_equal_blocking"] is True
    sec = out["secondary_base"]
    assert sec["n_base_per_shard"] == 10000 and sec["speedup_vs_one_gpu_same_base"]["blocking"] > 0


def test_bench_two_ranks_rccl_is_attempted_and_falls_back_loudly():
    """The driver launches `bench.py --gpus N` with the default `--backend nccl`.  On this one-GPU
    box both ranks share device 0, which RCCL refuses: the exchange group's probe collective must
    have been ATTEMPTED (requested == "nccl"), and the fallback to the host-staged gloo exchange
    must be loud -- `exchange.fallback` with the reason in the line and a WARNING on stderr -- so
    that a first run on real hardware cannot silently measure the copy path as "RCCL"."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29641", "bench.py", "--gpus", "2",
           "--steps", "3", "--warmup", "1", "--single-device", "--n-base", "20000",
           "--secondary-n-base", "0", "--n-query", "400", "--no-in-process",
           "--no-scaling-reference", "--no-pipelined"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    ex = out["exchange"]
    assert ex["requested"] == "nccl"
    if ex["fallback"]:      # one GPU: both ranks share device 0 -- found by the precondition check
        assert ex["used"].startswith("gloo") and ex["reason"]   # (no RCCL call, no timeout to sit out)
        assert "WARNING" in r.stderr and "does NOT measure RCCL" in r.stderr
        assert out["rccl_ranks"] == 0 and "share a device" in ex["reason"]
    else:                   # (a box where the probe works: then the line must say RCCL)
        assert "RCCL" in ex["used"] and out["rccl_ranks"] == 2
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["recall_at_10"] > 0.9
    # --require-rccl turns the loud fallback into a failure of the command
    r2 = subprocess.run(cmd + ["--require-rccl"], cwd=ROOT, env=env, capture_output=True, text=True,
                        timeout=600)
    if ex["fallback"]:
        assert r2.returncode != 0 and "--require-rccl" in (r2.stderr + r2.stdout)


def test_bench_lean_line_contract():
    out = run([sys.executable, "bench.py", "--lean", "--steps", "3", "--warmup", "1",
               "--n-base", "100000", "--n-query", "2000"])
    rl = out["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    assert abs(rl["achieved"] - rl["bytes_per_launch"] / (rl["kernel_ms"] * 1e-3) / 1e9) < 1e-6
    assert rl["traffic"] is None          # no committed PMC pass of this toy workload
    sec = rl["secondary"]
    assert sec["valu_issue"]["frac"] is None and "note" in sec["valu_issue"]
    assert sec["reference_algorithm_bytes"]["bytes_per_launch"] >= rl["bytes_per_launch"]
    assert out["dtype"] == "f32" and out["vs_baseline"] is None and out["cpu_baseline"] is None


def test_bench_u8_and_cosine_flags():
    out = run([sys.executable, "bench.py", "--lean", "--steps", "2", "--warmup", "1",
               "--n-base", "50000", "--n-query", "1000", "--dtype", "u8"])
    assert out["dtype"] == "u8" and "u8" in out["config"]["workload"]
    assert out["code_rows_per_query"] == 0 and out["roofline"]["frac"] > 0
    out = run([sys.executable, "bench.py", "--lean", "--steps", "2", "--warmup", "1",
               "--n-base", "30000", "--n-query", "500", "--dim", "256", "--measure", "cosine"])
    assert "cosine" in out["config"]["workload"] and out["recall_at_10"] > 0.5


def test_sift1m_real_plumbing_on_synthetic_files(tmp_path):
    """bench.sift1m_real (the real-dataset leg behind $GGNN_SIFT1M_DIR: the reference's four
    published SIFT1M settings, examples/python/sift1m_fvecs.py:19-30) on synthetic files with the
    TEXMEX names and formats -- sift_base.fvecs, sift_query.fvecs, sift_groundtruth.ivecs (100
    neighbours per query) -- so that the first box that has the real files cannot fail on
    plumbing: loaders, shapes, ground truth of the file against the engine's exact brute force."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import ggnn_amd as ggnn
    dev = torch.device("cuda", 0)
    base = bench.synthetic("lowrank16", 30_000, 128, 1234, dev)
    query = bench.synthetic("lowrank16", 500, 128, 4321, dev)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    gt100, _ = eng.bf_query(query, 100)
    ggnn.FloatDataset(base.cpu()).store(str(tmp_path / "sift_base.fvecs"))
    ggnn.FloatDataset(query.cpu()).store(str(tmp_path / "sift_query.fvecs"))
    ggnn.IntDataset(gt100.cpu()).store(str(tmp_path / "sift_groundtruth.ivecs"))
    del eng
    # the files have the TEXMEX record layout: int32 D, then D values
    raw = np.fromfile(tmp_path / "sift_groundtruth.ivecs", dtype=np.int32).reshape(500, 101)
    assert (raw[:, 0] == 100).all()
    out = bench.sift1m_real(str(tmp_path), ggnn, dev)
    assert out["N"] == 30_000 and out["Nq"] == 500 and out["graph_build_s"] > 0
    assert out["groundtruth_file"]["bf_query_agrees_at_10"] == 1.0
    assert len(out["points"]) == 4
    for name, p in out["points"].items():
        assert p["queries_per_s"] > 0 and 0.5 < p["c_at_10"] <= 1.0, name
        assert p["c_at_10_vs_groundtruth_file"] == p["c_at_10"], name
    assert out["points"]["tau=0.64,iters=400"]["c_at_10"] > 0.95      # (plumbing, not a recall claim)
