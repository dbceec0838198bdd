#!/usr/bin/env python3
"""Benchmark of the hot path: queries/sec at recall@10 on a SIFT1M-shaped base (+ build time).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: `query()` of the whole
query set (10 000 x 128 f32, k=10) against the resident graph; inputs are in HBM when the timed
region starts.

N = 1: BASELINE.json configs[1] (1M x 128 f32, k_build 24, tau_b 0.5; k = 10) on one MI355X.
N > 1: STRONG scaling on the north star's FIXED base: 100M points = 8 shards x 12.5M x 96 f32
(BASELINE configs[3], the DEEP100M shape: the base partitioned across the GPUs of one node;
--n-base / --dim change the shard, a second series on 8 x 1M x 128 is carried as
`secondary_base`).  Rank r owns shards
[r*8/N, (r+1)*8/N) as resident shards of one engine, every rank searches the full query set in its
shards, the sorted per-rank candidates are exchanged with ONE packed RCCL all-gather and merged on
the device.  `value` is queries/s (Nq / T of blocking 10k-query steps), not shard-searches; the
same line carries a saturating 100k-query batch and two batches in flight, the one-GPU point of
the SAME base in the same three modes (all 8 shards resident on rank 0's GPU, measured in the
same run) with the like-with-like speed-ups, the one-handle form (set_gpus([...]), in-engine
RCCL; a child process of rank 0) and the aggregate HBM roofline figure.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- SURVEY 8(d): the query kernel's own algorithmic bytes per launch (from its live
                  work counters) / its live HIP-event duration / the 8 TB/s HBM peak; the
                  reference algorithm's bytes (n_dist x 4D, what the kernel avoids moving), the
                  measured fabric traffic and the VALU-issue fraction (committed PMC passes, used
                  only while the kernel sources still match them) are named secondaries
  cpu_baseline -- the CPU oracle (a port of the reference algorithm; the reference has no CPU
                  path) timed on a bounded sample on this box's host cores (N=1, rank 0 only)
"""
import argparse
import glob
import hashlib
import json
import socket
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024             # 256 CUs x 4 SIMDs
F32_MFMA_PEAK = 157.3e12  # dense f32 MFMA
I8_MFMA_PEAK = 3944e12    # int8 MFMA: measured ceiling >= 3944 TOPS (MI355X_MICROARCH.md, no spec figure)
TOTAL_SHARDS = 8          # fixed base of the multi-GPU (strong scaling) series


def synthetic(kind, n, d, seed, device):
    """SIFT-like synthetic vectors: integer values in [0,255] stored as float32 (every fp32 sum
    of squared differences is exact, so GPU and oracle agree bit for bit).
      lowrank16: 16-dimensional Gaussian latent mixed into D dims (local intrinsic dimension
                 comparable to SIFT descriptors), rounded and clipped to [0,255]
      lowrankf16: the same without rounding/clipping (genuinely fractional float32 values)
      iid:       i.i.d. uniform integers (no structure; recall targets are not reachable)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if kind == "iid":
        return torch.randint(0, 256, (n, d), generator=g, device=device).float()
    if kind.startswith("lowrank"):
        fractional = kind.startswith("lowrankf")
        latent = int(kind[len("lowrankf" if fractional else "lowrank"):] or 16)
        ga = torch.Generator(device=device)
        ga.manual_seed(777)
        mix = torch.randn(latent, d, generator=ga, device=device) * (40.0 / latent ** 0.5)
        out = torch.empty((n, d), device=device)
        for lo in range(0, n, 1 << 20):
            hi = min(n, lo + (1 << 20))
            z = torch.randn(hi - lo, latent, generator=g, device=device)
            out[lo:hi] = 128 + z @ mix
            if not fractional:
                out[lo:hi].round_().clamp_(0, 255)
        return out
    raise ValueError(kind)


def local_intrinsic_dimension(eng, base, args, k=20, sample=10_000, seed=97):
    """Maximum-likelihood estimate of the local intrinsic dimension (Levina & Bickel 2004 in the
    averaged-inverse form of MacKay & Ghahramani): for `sample` base points, the k exact nearest
    neighbours (the engine's own brute force; the point itself, distance 0, is dropped) give
    m(x) = [ 1/(k-1) * sum_{j<k} ln(T_k(x) / T_j(x)) ]^-1 ; reported: the inverse of the mean of
    1/m(x).  Makes "SIFT1M-shaped" a number: published estimates for SIFT1M are around 20."""
    g = torch.Generator(device=base.device)
    g.manual_seed(seed)
    idx = torch.randperm(base.shape[0], generator=g, device=base.device)[:sample]
    pts = base[idx].contiguous()
    _, d = eng.bf_query(pts, k + 1, _measure(args))
    d = d.double()
    if args.measure == "l2":
        d = d.clamp_min(0).sqrt()
    d = d[:, 1:]                       # drop the point itself
    ok = d[:, 0] > 0                   # duplicates of the sample point carry no information
    d = d[ok]
    logs = torch.log(d[:, -1:] / d[:, :-1])
    inv_m = logs.sum(1) / (k - 1)
    return {"mle_k20": float(1.0 / inv_m.mean().item()), "points": int(ok.sum().item()),
            "k": k, "estimator": "Levina-Bickel MLE, MacKay-Ghahramani averaging, exact neighbours"}


def engine_clock_hz(device):
    """engine clock the device reports (kHz -> Hz); SQ cycle counters tick once per 4 clocks"""
    import ctypes as C
    from ggnn_amd._lib import check, lib
    hz = C.c_double()
    check(lib().ggnn_device_clock_hz(device.index or 0, C.byref(hz)))
    return float(hz.value)


KERNEL_SOURCES = ("traversal.hpp", "query.hip", "common.hpp", "prescreen.hip")
BUILD_SOURCES = ("traversal.hpp", "merge.hip", "sym.hip", "common.hpp", "prescreen.hip")


def kernel_source_sha(sources=None):
    """fingerprint of the traversal kernel sources: committed counter summaries carry the one
    they were collected with and are ignored once it no longer matches"""
    h = hashlib.sha256()
    for f in (sources or KERNEL_SOURCES):
        with open(os.path.join(ROOT, "ggnn_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recall_at_k(ids, gt):
    return (ids.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()


def host_cpu_budget():
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU boxes of
    this pool show 256 hardware threads but grant 16 CPUs: 256 busy threads then run at a
    quarter of the rate of 32), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(base, query, k, graph, cfg, stats, tau, iters, budget_s=12.0):
    """Oracle timed on the host cores: brute force (the 'reference CPU brute force' of
    BASELINE.json, a port because the reference has none: cache-blocked, 4 rows x 8 AVX lanes,
    direct form without contraction so that its sums equal the plain port's) and the traversal
    port.  Threads: twice the CPU quota when the container has one (measured on this pool:
    32 threads 1.34 TFLOP/s, 64: 1.29, 256: 0.83 on a 16-CPU quota), else every hardware thread."""
    from oracle import oracle as orc
    orc.set_fast_distance(True)  # plain loops, not the lockstep emulation used for parity
    hw, quota = host_cpu_budget()
    threads = int(min(hw, max(1, round(2 * quota)))) if quota else hw
    cores = threads
    base_h = base.cpu().numpy()
    q_h = query.cpu().numpy()
    # calibrate on a few queries, then size the sample for ~budget_s
    # (the fast port handles queries in groups of 32 per thread: probe with whole groups)
    probe = max(32, min(32 * cores, q_h.shape[0]))
    t = time.perf_counter()
    orc.bf_query(base_h, q_h[:probe], k, threads=cores)
    dt = max(time.perf_counter() - t, 1e-3)
    rate = probe / dt
    n = int(max(probe, min(q_h.shape[0], rate * budget_s)))
    t = time.perf_counter()
    orc.bf_query(base_h, q_h[:n], k, threads=cores)
    bf_s = time.perf_counter() - t
    # traversal port on the GPU-built graph
    nq_t = min(q_h.shape[0], 2000)
    start = graph["tr"][cfg["STs_offsets"][3]:cfg["STs_offsets"][3] + cfg["Ns"][3]]
    t = time.perf_counter()
    orc.query(base_h, q_h[:nq_t], graph["graph0"], start, stats, k, tau, iters, threads=cores)
    tr_s = time.perf_counter() - t
    orc.set_fast_distance(False)
    flops = 3.0 * n * base_h.shape[0] * base_h.shape[1]
    # `cores`: the CPUs the box grants (the cgroup quota where there is one: 16 on this pool);
    # `threads`: what the port ran with (twice the quota measured fastest)
    return {"value": n / bf_s, "unit": "queries/s",
            "cores": int(round(quota)) if quota else cores, "threads": threads, "kind": "port",
            "host": {"hardware_threads": hw, "cgroup_cpu_quota": quota},
            "sample": f"oracle bf_query (cache-blocked AVX2 port of bf_query_layer.cu, direct "
                      f"form, {cores} threads), first {n} of {q_h.shape[0]} queries x "
                      f"{base_h.shape[0]} base rows, {bf_s:.1f} s",
            "gflops": flops / bf_s / 1e9,
            "gflops_per_thread": flops / bf_s / 1e9 / threads,
            "note": "baseline only; one thread of this port alone reaches ~60 GFLOP/s on the "
                    "pool's EPYC 9575F (about half of what its sub/mul/add loop can issue)",
            "traversal_port_qps": nq_t / tr_s,
            "traversal_sample": f"oracle query on the GPU-built graph, {nq_t} queries, "
                                f"{tr_s:.1f} s"}


def workload_string(args):
    shape = "SIFT1M-shaped " if (args.n_base, args.dim) == (1_000_000, 128) else ""
    measure = "" if args.measure == "l2" else ", cosine"
    return (f"{args.dataset} {shape}{args.n_base}x{args.dim} {args.dtype} per GPU{measure}, "
            f"{args.n_query} queries, k={args.k}, k_build={args.k_build}, "
            f"tau_build={args.tau_build}, refine={args.refine}, "
            f"tau_query={args.tau_query}, max_iterations={args.max_iters}")


def _latest_profile(suffix, args):
    """newest committed profiles/*<suffix> collected on THIS workload (else None)"""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True):
        try:
            with open(f) as fh:
                doc = json.load(fh)
        except Exception:
            continue
        if doc.get("workload") == workload_string(args):
            doc["_file"] = os.path.relpath(f, ROOT)
            doc["_stale"] = doc.get("kernel_source_sha") != kernel_source_sha()
            doc["_stale_build"] = doc.get("build_source_sha") != kernel_source_sha(BUILD_SOURCES)
            return doc
    return None


def _query_kernel_entry(doc, prescreened):
    cands = [(name, c) for name, c in doc["kernels"].items()
             if "query_kernel" in name and "bf_query" not in name]
    for name, c in cands:
        if ("NoPrescreen" in name) != prescreened:
            return name, c
    return None, None


def pmc_traffic(args, prescreened):
    """HBM-side bytes per query_kernel launch from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/*_pmc_hbm.json), corrected
    as MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE x2 on gfx950 for 16 B/lane loads).
    Only valid for the default workload; otherwise null."""
    doc = _latest_profile("_pmc_hbm.json", args)
    if not doc or doc["_stale"]:
        return None
    _, c = _query_kernel_entry(doc, prescreened)
    if not c or "FETCH_SIZE" not in c:
        return None
    wr = c.get("WRITE_SIZE", {"avg_kb": 0.0})["avg_kb"]
    return 2.0 * c["FETCH_SIZE"]["avg_kb"] * 1024.0 + wr * 1024.0


def pmc_sq(args, prescreened):
    """SQ counters of the query kernel from the committed profile of this workload"""
    doc = _latest_profile("_pmc_sq.json", args)
    if not doc:
        return None, None, "no committed SQ counter pass for this workload"
    if doc["_stale"]:
        return None, doc["_file"], ("the kernel sources changed since this pass was collected "
                                    "(kernel_source_sha differs): re-run scripts/profile_round.sh")
    name, c = _query_kernel_entry(doc, prescreened)
    if not c:
        return None, doc["_file"], "the pass holds no entry for this kernel variant"
    c = dict(c, _kernel=name)
    return c, doc["_file"], None


def _measure(args):
    import ggnn_amd as ggnn
    return ggnn.DistanceMeasure.Cosine if args.measure == "cosine" else ggnn.DistanceMeasure.Euclidean


def make_data(args, kind, n, seed, device):
    x = synthetic(kind, n, args.dim, seed, device)
    return x.to(torch.uint8) if args.dtype == "u8" else x


def measure_point(eng, query, gt, args, steps, tau=None, iters=None, warm=2):
    """kernel time (HIP events inside the engine) and recall of one operating point"""
    tau = args.tau_query if tau is None else tau
    iters = args.max_iters if iters is None else iters
    m = _measure(args)
    for _ in range(warm):
        eng.query(query, args.k, tau, iters, m)
    ms = []
    for _ in range(steps):
        ids, _ = eng.query(query, args.k, tau, iters, m)
        ms.append(eng.last_timing_ms()["query_ms"])
    t = float(np.mean(ms))
    return {"query_kernel_ms": t, "queries_per_s": query.shape[0] / (t * 1e-3),
            "recall_at_10": recall_at_k(ids, gt)}


# Operating points (tau_query, max_iterations): `value` is quoted AT recall@10 >= 0.99, so the point is
# part of the result.  Each was chosen from a sweep on the tuning query set (seed 4321) with the
# recall confirmed on held-out sets (scripts/point_sweep.py, scripts/early_probe.py,
# scripts/shard8_probe.py; tables in DESIGN.md 4).  --tau-query / --max-iters override.
OPERATING_POINTS = {
    # (shards searched per query, points per shard, D, dtype, measure): (tau, iterations, recall)
    (1, 1_000_000, 128, "f32", "l2"): (0.85, 175, "0.9914-0.9917 on three query sets"),
    (1, 1_000_000, 128, "u8", "l2"): (0.85, 175, "0.9913-0.9917"),
    # one DEEP100M shard alone: 0.95 / 280 (a 512-key cache: sorted part of 32 keys, no ring)
    (1, 12_500_000, 96, "f32", "l2"): (0.95, 280, "0.9913 / 0.9918 (tuning / held-out queries)"),
    # the 100M x 96 base in 8 shards, MERGED recall (every true neighbour only has to be found by
    # the one shard that holds it): 0.8 / 280 -> 0.9936 / 0.9928
    (8, 12_500_000, 96, "f32", "l2"): (0.8, 280, "merged 0.9936 / 0.9928"),
}
FALLBACK_POINT = (0.85, 175)


def resolve_operating_point(args, shards):
    """fills args.tau_query / args.max_iters from OPERATING_POINTS unless given on the command line"""
    key = (shards, args.n_base, args.dim, args.dtype, args.measure)
    tau, iters = OPERATING_POINTS.get(key, (FALLBACK_POINT + ("",)))[:2]
    args.operating_point_source = ("command line" if args.tau_query is not None or
                                   args.max_iters is not None else
                                   ("OPERATING_POINTS[%r]" % (key,) if key in OPERATING_POINTS
                                    else "fallback (untuned shape)"))
    if args.tau_query is None:
        args.tau_query = tau
    if args.max_iters is None:
        args.max_iters = iters


SEARCH_TAUS = (0.5, 0.64, 0.8, 0.9, 1.0, 1.1, 1.2, 1.5, 2.0, 2.5)
SEARCH_ITERS = (100, 175, 250, 400, 600, 700, 750, 800, 1000, 1500, 1750, 2000)


def cheapest_point_at_recall(eng, query, gt, args, target=0.99):
    """BASELINE's metric is queries/s AT recall@10 >= 0.99: the fastest (tau_query,
    max_iterations) of a bounded grid that reaches the target on this base, against its own exact
    ground truth.  The kernel time grows with both parameters, so rows of the grid are left as
    soon as a point is slower than the best one found."""
    best, top, tried = None, None, 0
    for iters in SEARCH_ITERS:
        for tau in SEARCH_TAUS:
            r = measure_point(eng, query, gt, args, 2, tau, iters, warm=1)
            tried += 1
            r.update(tau_query=tau, max_iterations=iters)
            if top is None or r["recall_at_10"] > top["recall_at_10"]:
                top = r
            if best is not None and r["query_kernel_ms"] >= best["query_kernel_ms"]:
                break       # larger tau in this row only costs more
            if r["recall_at_10"] >= target:
                best = r
                break
    if best is not None:
        tau, iters = best["tau_query"], best["max_iterations"]
        best = measure_point(eng, query, gt, args, 5, tau, iters)
        best.update(tau_query=tau, max_iterations=iters)
    return best, top, tried


def recall_target_sweep(args, device, ggnn, own_eng, own_base, own_query, own_gt):
    """queries/s at recall@10 >= 0.99 per synthetic base (16 / 24 / 32-dimensional latent with
    integer values, and the 16-dimensional one with genuinely fractional float32 values, whose
    pre-screen codes are lossy), each with its own graph, exact ground truth and operating point"""
    out = {}
    for kind in ("lowrank16", "lowrank24", "lowrank32", "lowrankf16"):
        if args.dtype == "u8" and kind == "lowrankf16":
            continue
        if kind == args.dataset:
            eng, query, gt, build_s = own_eng, own_query, own_gt, None
        else:
            base = make_data(args, kind, args.n_base, 1234, device)
            query = make_data(args, kind, args.n_query, 4321, device)
            eng = ggnn.GGNN()
            eng.set_base_reference(base)
            eng.set_return_results_on_gpu(True)
            eng.build(args.k_build, args.tau_build, args.refine, _measure(args))
            build_s = eng.last_timing_ms()["build_ms"] / 1000.0
            gt, _ = eng.bf_query(query, args.k, _measure(args))
        r = {"same_settings_as_headline": measure_point(eng, query, gt, args, 3)}
        r["local_intrinsic_dimension"] = local_intrinsic_dimension(
            eng, own_base if kind == args.dataset else base, args)
        if build_s is not None:
            r["graph_build_s"] = build_s
        best, top, tried = cheapest_point_at_recall(eng, query, gt, args)
        r["grid_points_tried"] = tried
        if best is not None:
            r["at_recall_0.99"] = best
            if kind != args.dataset:
                # the same point on a batch that fills the chip many times over: a 10 000-query
                # batch of long searches is 1.4 rounds of resident waves (7168 searches per round),
                # i.e. two wave latencies -- the saturated rate is what the kernel sustains
                try:
                    big = make_data(args, kind, 10 * args.n_query, 9876, device)
                    for _ in range(2):
                        eng.query(big, args.k, best["tau_query"], best["max_iterations"], _measure(args))
                    ms = eng.last_timing_ms()["query_ms"]
                    best["saturated_batch"] = {"n_query": int(big.shape[0]), "query_kernel_ms": ms,
                                               "queries_per_s": big.shape[0] / (ms * 1e-3)}
                    del big
                except Exception as e:   # informational
                    best["saturated_batch"] = {"error": repr(e)}
        else:
            r["at_recall_0.99"] = None
            r["not_reached_best"] = top
        out[kind] = r
        if kind != args.dataset:
            del eng, base, query, gt
            torch.cuda.empty_cache()
    return out


def bf_block(args, bf_ms, rescanned):
    """exact brute force (the recall ground truth) against the matrix-core peak of its dtype"""
    ops_ = 2.0 * args.n_query * args.n_base * args.dim
    if args.dtype == "u8" and args.measure == "l2" and args.dim <= 128:
        kernel = ("bf_i8v2_kernel (v_mfma_i32_32x32x32_i8, K-best sets in registers), exact integers"
                  if args.k <= 16 else "bf_mfma_i8_kernel (v_mfma_i32_32x32x32_i8, LDS lists), exact integers")
        peak, unit = I8_MFMA_PEAK, "TOPS"
    else:
        kernel, peak, unit = ("bf_mfma_kernel (v_mfma_f32_32x32x2_f32) + certified exact re-rank",
                              F32_MFMA_PEAK, "TFLOP/s")
    return {"ms": bf_ms, "kernel": kernel, "rate": ops_ / (bf_ms * 1e-3) / 1e12, "unit": unit,
            "peak": peak / 1e12, "mfma_frac_of_peak": ops_ / (bf_ms * 1e-3) / peak,
            "note": "end to end (norms, tile kernel, re-rank / certificate), second call",
            "queries_rescanned_by_the_exact_scan": rescanned}


def sift1m_real(directory, ggnn, device):
    """$GGNN_SIFT1M_DIR/sift_base.fvecs + sift_query.fvecs (+ sift_groundtruth.ivecs): the
    reference's four published settings (examples/python/sift1m_fvecs.py:19-30) next to their
    targets, through the engine's own fvecs loader"""
    base = ggnn.FloatDataset.load(os.path.join(directory, "sift_base.fvecs"))
    query = ggnn.FloatDataset.load(os.path.join(directory, "sift_query.fvecs"))
    b = base.view.to(device)
    q = query.view.to(device)
    eng = ggnn.GGNN()
    eng.set_base_reference(b)
    eng.set_return_results_on_gpu(True)
    eng.build(24, 0.5, 2)
    gt, _ = eng.bf_query(q, 10)
    out = {"N": int(b.shape[0]), "Nq": int(q.shape[0]),
           "graph_build_s": eng.last_timing_ms()["build_ms"] / 1e3, "points": {}}
    # the dataset's own ground truth (TEXMEX: 100 neighbours per query), when the file is there:
    # recall is quoted against IT as well, and the engine's exact brute force has to agree with it
    # (up to exact distance ties, which real SIFT has: equal rows are counted by distance)
    gt_file = None
    gt_path = os.path.join(directory, "sift_groundtruth.ivecs")
    if os.path.exists(gt_path):
        gt_file = ggnn.IntDataset.load(gt_path).view.to(device)[:, :10].contiguous()
        out["groundtruth_file"] = {"K": 10, "bf_query_agrees_at_10": recall_at_k(gt, gt_file)}
    for tau, iters, target in ((0.34, 200, "c@1 ~0.90"), (0.41, 200, "c@1 ~0.95"),
                               (0.51, 200, "c@1 ~0.99"), (0.64, 400, "c@10 ~0.99")):
        for _ in range(3):
            ids, _ = eng.query(q, 10, tau, iters)
        ms = eng.last_timing_ms()["query_ms"]
        out["points"][f"tau={tau},iters={iters}"] = {
            "published_target": target, "c_at_1": (ids[:, 0] == gt[:, 0]).float().mean().item(),
            "c_at_10": recall_at_k(ids, gt), "query_kernel_ms": ms,
            "queries_per_s": q.shape[0] / (ms * 1e-3)}
        if gt_file is not None:
            out["points"][f"tau={tau},iters={iters}"]["c_at_10_vs_groundtruth_file"] = \
                recall_at_k(ids, gt_file)
    return out


L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md: aggregate L2 bandwidth, 8 XCDs


def _pmc_kernel_total(args, needle):
    """(bytes, launches) of FETCH_SIZE x2 + WRITE_SIZE summed over every launch of the kernels
    whose name contains `needle`, from the committed PMC passes of this workload (else None)"""
    doc = _latest_profile("_pmc_hbm.json", args)
    if not doc or doc["_stale_build"]:
        return None, None
    total, launches = 0.0, 0
    for name, c in doc["kernels"].items():
        if needle in name and "FETCH_SIZE" in c:
            n = c["FETCH_SIZE"]["launches"]
            wr = c.get("WRITE_SIZE", {"avg_kb": 0.0})["avg_kb"]
            total += n * (2.0 * c["FETCH_SIZE"]["avg_kb"] + wr) * 1024.0
            launches += n
    return (total, launches) if launches else (None, None)


def _pmc_build_sq(args, kname):
    """SQ counters of one construction kernel family summed over a build, from the committed pass
    of this workload (None when absent or collected with other kernel sources)"""
    doc = _latest_profile("_pmc_build_sq.json", args)
    if not doc or doc.get("build_source_sha") != kernel_source_sha(BUILD_SOURCES):
        return None, None
    return doc["kernels"].get(kname), doc["_file"]


def build_roofline(args, ggnn, base):
    """SURVEY 8(d) for the build: per construction kernel, summed over every launch of one whole
    build -- own algorithmic bytes (live work counters of a second, counted build: rows actually
    read, graph rows per pop, the point's own row and its result row), the sum of the launches'
    HIP-event durations, the fraction of the 8 TB/s HBM peak, and the fabric traffic of the same
    launches from the committed PMC passes.  The reference algorithm's bytes (n_dist x 4D: every
    evaluation reads a float row) are a named secondary."""
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_collect_counters(True)
    eng.build(args.k_build, args.tau_build, args.refine, _measure(args))
    work = eng.last_build_work()
    counted_build_s = eng.last_timing_ms()["build_ms"] / 1e3
    del eng
    clock_hz = engine_clock_hz(base.device)
    d, k = args.dim, args.k_build
    esz = 1 if args.dtype == "u8" else 4
    code_dim = max(16, 1 << (d - 1).bit_length()) if d <= 64 else (d + 63) // 64 * 64
    out = {"counted_build_s": counted_build_s,
           "note": "one whole build (all layers and refinement passes) in the diagnostic mode that "
                   "synchronises after every merge / sym launch; `achieved` can exceed what HBM "
                   "delivers because consecutive points share neighbourhoods (rows served by the "
                   "L2s and the Infinity Cache): `traffic` is what reached the fabric"}
    for kname, w, row_bytes_per_pop, needle in (
            ("merge_kernel", work["merge"], k * 4, "::merge_kernel<"),
            ("sym_kernel", work["sym"], (k + k // 2) * 4, "::sym_kernel<")):
        if not w["launches"]:
            continue
        own = (w["float_rows"] * d * esz + w["code_rows"] * code_dim + w["pops"] * row_bytes_per_pop
               + w["points"] * (d * esz + k * 4))
        ref = w["n_dist"] * d * esz + w["pops"] * row_bytes_per_pop + w["points"] * (d * esz + k * 4)
        t = w["ms"] * 1e-3
        traffic, launches = _pmc_kernel_total(args, needle)
        if launches is not None and launches != w["launches"]:
            traffic = None
        gbs = own / t / 1e9
        # Three candidate roofs, the largest fraction names the bound (no fraction above 1):
        #   hbm   -- bytes that REACHED the fabric (committed FETCH/WRITE passes) / 8 TB/s.  The own
        #            bytes are NOT an HBM figure here: consecutive points share neighbourhoods, so
        #            rows are served by the L2s and the Infinity Cache (own / 8 TB/s exceeded 1 for
        #            sym in round 4 -- kept below as `own_bytes_over_hbm_peak`, a ratio, not a frac)
        #   l2    -- own bytes / the aggregate L2 bandwidth
        #   valu  -- SQ_ACTIVE_INST_VALU x 4 / (SIMD-cycles of the launches) from the committed SQ pass
        fracs = {"l2": gbs / L2_PEAK_GBS}
        if traffic is not None:
            fracs["hbm"] = traffic / t / 1e9 / HBM_PEAK_GBS
        sq, sq_file = _pmc_build_sq(args, kname)
        valu = None
        if sq and sq.get("SQ_ACTIVE_INST_VALU") and sq.get("duration_s_sq2_pass"):
            valu = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (sq["duration_s_sq2_pass"] * clock_hz * N_SIMD)
            fracs["valu"] = valu
        bound = max(fracs, key=fracs.get)
        peak = {"hbm": HBM_PEAK_GBS, "l2": L2_PEAK_GBS, "valu": 1.0}[bound]
        achieved = {"hbm": None if traffic is None else traffic / t / 1e9, "l2": gbs,
                    "valu": valu}[bound]
        out[kname] = {
            "launches": w["launches"], "points": w["points"], "kernel_ms_sum": w["ms"],
            "n_dist": w["n_dist"], "float_rows": w["float_rows"], "code_rows": w["code_rows"],
            "graph_rows": w["pops"],
            "roofline": {"bound": bound, "achieved": achieved, "peak": peak,
                         "unit": "fraction of the VALU issue slots" if bound == "valu" else "GB/s",
                         "frac": fracs[bound], "bytes": own, "traffic": traffic,
                         "candidates": fracs,
                         "traffic_over_algorithmic": None if traffic is None else traffic / own,
                         "own_bytes_over_hbm_peak": gbs / HBM_PEAK_GBS,
                         "sq_pass": sq_file,
                         "definition": "own bytes = float_rows x D x s + code_rows x Dc + graph_rows "
                                       "x row bytes + points x (D x s + KBuild x 4), summed over the "
                                       "launches; t = sum of their HIP-event durations; candidates: "
                                       "fabric bytes / t / 8 TB/s (hbm), own bytes / t / 34.5 TB/s "
                                       "(l2), VALU issue (valu); `bound` = the largest"},
            "reference_algorithm": {"bytes": ref, "achieved": ref / t / 1e9,
                                    "note": "n_dist x D x s instead of the rows actually read: what "
                                            "the reference's algorithm would move (bytes avoided by "
                                            "the exact pre-screen when above `bytes`)"}}
    return out


def big_base(args, n_rows, seed0, device):
    """n_rows synthetic rows generated in chunks of one shard (seed per shard, so that a shard
    holds the same rows whichever rank or process generates it)"""
    shards = n_rows // args.n_base
    base = torch.empty((n_rows, args.dim), dtype=torch.float32, device=device)
    for s_ in range(shards):
        lo = s_ * args.n_base
        for c in range(0, args.n_base, 5_000_000):
            hi = min(args.n_base, c + 5_000_000)
            base[lo + c:lo + hi] = synthetic(args.dataset, hi - c, args.dim,
                                             1234 + 1000 * (seed0 + s_) + c // 5_000_000, device)
    return base


def timed(fn, steps, sync):
    """wall time per call of fn() over `steps` calls, bracketed by sync()"""
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    sync()
    return (time.perf_counter() - t0) / steps, out


def one_gpu_reference(args, device, ggnn, steps):
    """the one-GPU point of the strong-scaling series: all TOTAL_SHARDS shards of the SAME base
    resident on this GPU (the reference would need its GPU<->CPU swapping or the 8 GPUs).
    Blocking 10k-query calls, a saturating 100k-query batch and two batches in flight."""
    base = big_base(args, TOTAL_SHARDS * args.n_base, 0, device)
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_shard_size(args.n_base)
    eng.set_return_results_on_gpu(True)
    eng.build(args.k_build, args.tau_build, args.refine)
    build_s = eng.last_timing_ms()["build_ms"] / 1000.0
    gt, _ = eng.bf_query(query, args.k)
    k = args.k

    def step(q=query):
        ids, dists = eng.query(q, k, args.tau_query, args.max_iters)
        # results on the GPU are the sorted [Nq, K * shards] rows: the answer is their head
        return ids[:, :k], dists[:, :k]

    for _ in range(2):
        step()
    el, (ids, _) = timed(step, steps, torch.cuda.synchronize)
    out = {"workload": f"{TOTAL_SHARDS} resident shards x {args.n_base} points on ONE GPU "
                       f"({TOTAL_SHARDS * args.n_base} x {args.dim} f32), {args.n_query} queries",
           "queries_per_s": args.n_query / el, "ms_per_step": el * 1e3,
           "recall_at_10": recall_at_k(ids.contiguous(), gt), "graph_build_s": build_s}
    try:
        big = synthetic(args.dataset, 10 * args.n_query, args.dim, 9876, device)
        step(big)
        el_b, _ = timed(lambda: step(big), max(2, steps // 3), torch.cuda.synchronize)
        out["saturated_batch"] = {"n_query": int(big.shape[0]), "ms_per_step": el_b * 1e3,
                                  "queries_per_s": big.shape[0] / el_b}
        del big
        eng.query_async(query, k, args.tau_query, args.max_iters, slot=0)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.query_async(query, k, args.tau_query, args.max_iters, slot=i % 2)
        eng.synchronize()
        el_p = (time.perf_counter() - t0) / steps
        out["pipelined_batches"] = {"batches_in_flight": 2, "ms_per_batch": el_p * 1e3,
                                    "queries_per_s": args.n_query / el_p}
    except Exception as e:   # secondary figures: never lose the reference point over them
        out["secondary_error"] = repr(e)
    del eng, base
    torch.cuda.empty_cache()
    return out


def scaling_reference(args, device, ggnn, steps):
    return one_gpu_reference(args, device, ggnn, steps)


def run_single(args, device, ggnn):
    measure = _measure(args)
    if args.n_base > 5_000_000:
        base = torch.empty((args.n_base, args.dim), device=device,
                           dtype=torch.uint8 if args.dtype == "u8" else torch.float32)
        for lo in range(0, args.n_base, 5_000_000):
            hi = min(args.n_base, lo + 5_000_000)
            base[lo:hi] = make_data(args, args.dataset, hi - lo, 1234 + lo, device)
    else:
        base = make_data(args, args.dataset, args.n_base, 1234, device)
    query = make_data(args, args.dataset, args.n_query, 4321, device)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_return_results_on_gpu(True)
    t0 = time.perf_counter()
    eng.build(args.k_build, args.tau_build, args.refine, measure)
    torch.cuda.synchronize()
    build_wall_s = time.perf_counter() - t0
    build_kernel_s = eng.last_timing_ms()["build_ms"] / 1000.0

    def step():
        return eng.query(query, args.k, args.tau_query, args.max_iters, measure)

    # ground truth by exact brute force on the same data (untimed)
    gt, _ = eng.bf_query(query, args.k, measure)
    # its time is reported from a second call: the first one also grows the scratch pool
    eng.bf_query(query, args.k, measure)
    bf_ms = eng.last_timing_ms()["bf_query_ms"]
    bf_rescanned = eng.last_bf_query_rescanned()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, dists = step()
        kernel_ms.append(eng.last_timing_ms()["query_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    recall = recall_at_k(ids, gt)
    c1 = (ids[:, 0] == gt[:, 0]).float().mean().item()

    # the operating point was chosen on the query set above (seed 4321); a query set it has never
    # seen tells whether the recall figure generalises
    held = make_data(args, args.dataset, args.n_query, 8642, device)
    held_gt, _ = eng.bf_query(held, args.k, measure)
    held_ids, _ = eng.query(held, args.k, args.tau_query, args.max_iters, measure)
    recall_heldout = recall_at_k(held_ids, held_gt)
    del held, held_gt, held_ids

    # work counters of one pass (untimed extra run) for the roofline figure
    eng.set_collect_counters(True)
    eng.query(query, args.k, args.tau_query, args.max_iters, measure)
    cnt = eng.last_query_counters()
    rows = eng.last_query_rows_read()
    eng.set_collect_counters(False)

    # the same kernel without the pre-screen (untimed extra runs): the HBM-bound form of the
    # traversal, every distance evaluation reads its 4D-byte row
    plain_ms = None
    if rows["code_rows"] > 0:
        eng.set_prescreen(False)
        plain = []
        for i in range(2 + min(args.steps, 10)):
            ids_plain, dists_plain = step()
            if i >= 2:
                plain.append(eng.last_timing_ms()["query_ms"])
        eng.set_prescreen(True)
        plain_ms = float(np.mean(plain))
        if not (torch.equal(ids_plain, ids) and torch.equal(dists_plain, dists)):
            raise RuntimeError("pre-screened and plain query results differ")

    # informational: throughput with a batch large enough to keep every SIMD busy to the end
    # (10k queries are 9.8 waves per SIMD with 7 resident: the tail of the launch runs at low
    # occupancy); not part of `value`
    saturated = None
    if args.saturated_batch:
        big = make_data(args, args.dataset, 10 * args.n_query, 9876, device)
        for _ in range(2):
            eng.query(big, args.k, args.tau_query, args.max_iters, measure)
        sat_ms = []
        for _ in range(3):
            eng.query(big, args.k, args.tau_query, args.max_iters, measure)
            sat_ms.append(eng.last_timing_ms()["query_ms"])
        saturated = {"n_query": int(big.shape[0]), "query_kernel_ms": float(np.mean(sat_ms)),
                     "queries_per_s": big.shape[0] / (float(np.mean(sat_ms)) * 1e-3)}
        del big

    # informational: the same batches enqueued without waiting for each other (query_async, two
    # alternating streams): the thin tail of one launch overlaps with the head of the next, which
    # is what a server with several batches in flight sees; not part of `value`
    pipelined = None
    if not args.no_pipelined:
        for slot in range(2):
            eng.query_async(query, args.k, args.tau_query, args.max_iters, measure, slot=slot)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [eng.query_async(query, args.k, args.tau_query, args.max_iters, measure, slot=i % 2)
                for i in range(args.steps)]
        eng.synchronize()
        pip_s = (time.perf_counter() - t0) / args.steps
        if not all(torch.equal(o[0], ids) and torch.equal(o[1], dists) for o in outs):
            raise RuntimeError("asynchronous and blocking query results differ")
        pipelined = {"batches_in_flight": 2, "n_query_per_batch": args.n_query,
                     "ms_per_batch": pip_s * 1e3, "queries_per_s": args.n_query / pip_s,
                     "results": "bit-identical to the blocking calls"}
        del outs

    nq, d, k = args.n_query, args.dim, args.k
    esz = 1 if args.dtype == "u8" else 4
    ms_per_step = elapsed / args.steps * 1000.0
    value = nq / (elapsed / args.steps)
    # bytes_q = D*s + n_dist*D*s + n_pop*KBuild*4 + S*4 + 8 + K*8 is what the
    # reference's algorithm moves.  With the exact pre-screen a distance evaluation
    # reads a D-byte code row and only the candidates that pass it read their 4D-byte float row:
    # the algorithmic bytes of THIS kernel are counted from its own row counters.
    fixed = nq * d * esz + cnt["n_pop"] * args.k_build * 4 + nq * (32 * 4 + 8 + k * 8)
    ref_bytes = fixed + cnt["n_dist"] * d * esz
    # prescreen_code_dim(D), common.hpp: power of two up to 64, then multiples of 64 bytes
    code_dim = max(16, 1 << (d - 1).bit_length()) if d <= 64 else (d + 63) // 64 * 64
    prescreened = rows["code_rows"] > 0
    alg_bytes = fixed + rows["float_rows"] * d * esz + rows["code_rows"] * code_dim
    if prescreened:
        alg_bytes += nq * (code_dim + 8) * 4  # per-dimension offsets + header, per query
    avg_kernel_ms = float(np.mean(kernel_ms))
    t_kernel = avg_kernel_ms * 1e-3
    alg_gbs = alg_bytes / t_kernel / 1e9
    traffic = pmc_traffic(args, prescreened)
    sq, sq_file, sq_note = pmc_sq(args, prescreened)
    default_shape = (args.dtype, args.measure, args.dim) == ("f32", "l2", 128)
    if default_shape and args.max_iters <= 512 and args.k <= 15:
        kernel_name = ("query_kernel<float,16,2,1,L2,Prescreen<8,1>,HB=1>" if prescreened
                       else "query_kernel<float,16,2,1,L2,NoPrescreen>")
    else:
        kernel_name = (f"query_kernel<{args.dtype}, D={d}, {args.measure}, "
                       f"{'Prescreen' if prescreened else 'NoPrescreen'}> (variant chosen by the engine)")
    clock = engine_clock_hz(device)
    valu = None
    if sq and "SQ_ACTIVE_INST_VALU" in sq:
        peak = N_SIMD * clock / 4.0
        valu = {"achieved": sq["SQ_ACTIVE_INST_VALU"] / t_kernel / 1e9, "peak": peak / 1e9,
                "unit": "G VALU-busy SIMD quad-cycles/s",
                "frac": sq["SQ_ACTIVE_INST_VALU"] / t_kernel / peak,
                "engine_clock_hz": clock, "counters_from": sq_file, "kernel_in_profile": sq["_kernel"],
                "valu_insts_per_pop": sq.get("SQ_INSTS_VALU", 0.0) / max(1, cnt["n_pop"]),
                "note": "SQ_ACTIVE_INST_VALU per launch (committed PMC pass of this workload, "
                        "collected with the same kernel sources) / (live kernel time x 1024 SIMDs "
                        "x reported engine clock / 4): the instruction-issue share of the kernel "
                        "time -- what actually limits the pre-screened kernel"}
    roofline = {
        "bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
        "bytes_per_launch": alg_bytes, "kernel_ms": avg_kernel_ms,
        "definition": "algorithmic bytes of this kernel per launch (rows it reads: "
                      "float_rows x D x s + code_rows x Dc, graph rows n_pop x KBuild x 4, query, "
                      "start ids, results; live work counters of this run) / average launch "
                      "duration (HIP events on the engine's stream, this run) / 8 TB/s",
        "traffic_note": ("FETCH_SIZE x2 (gfx950 16 B/lane correction) + WRITE_SIZE per launch from "
                         "the committed PMC passes of this workload" if traffic is not None else
                         "null: no committed FETCH_SIZE/WRITE_SIZE pass matches this workload and "
                         "these kernel sources"),
        "secondary": {
            "reference_algorithm_bytes": {
                "bytes_per_launch": ref_bytes, "achieved": ref_bytes / t_kernel / 1e9,
                "frac": ref_bytes / t_kernel / 1e9 / HBM_PEAK_GBS,
                "note": "SURVEY 8(d)'s formula with n_dist x D x s: what the reference's algorithm "
                        "would move for the same answers.  With the exact pre-screen most "
                        "evaluations read a D-byte code row instead, so this is bytes AVOIDED "
                        "and may exceed the peak"},
            "measured_fabric_traffic": (None if traffic is None else {
                "bytes_per_launch": traffic, "achieved": traffic / t_kernel / 1e9,
                "frac": traffic / t_kernel / 1e9 / HBM_PEAK_GBS,
                "over_algorithmic": traffic / alg_bytes}),
            "valu_issue": valu if valu is not None else {"frac": None, "note": sq_note},
        },
    }
    roofline["without_prescreen"] = (None if plain_ms is None else {
        "kernel": "query_kernel<float,16,2,1,L2,NoPrescreen>" if default_shape else "NoPrescreen variant",
        "query_kernel_ms": plain_ms,
        "queries_per_s": nq / (plain_ms * 1e-3), "bound": "hbm + infinity cache",
        "achieved": ref_bytes / (plain_ms * 1e-3) / 1e9,
        "frac": ref_bytes / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc_traffic(args, False),
        "results": "bit-identical to the pre-screened run"})

    lid = local_intrinsic_dimension(eng, base, args)
    out = {
        "metric": "queries/sec @ recall@10 (SIFT1M-shaped, k=10)",
        "value": value, "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_string(args), "parallelism": "single GPU",
                   "operating_point": {"tau_query": args.tau_query, "max_iterations": args.max_iters,
                                       "source": getattr(args, "operating_point_source", None)}},
        "recall_at_10": recall, "recall_at_10_heldout_queries": recall_heldout, "c_at_1": c1,
        "base_local_intrinsic_dimension": lid,
        "graph_build_s": build_kernel_s, "graph_build_wall_s": build_wall_s,
        "bf_query_ms": bf_ms,
        "bf_query": bf_block(args, bf_ms, bf_rescanned),
        "query_kernel_ms": avg_kernel_ms,
        "n_dist_per_query": cnt["n_dist"] / nq, "n_pop_per_query": cnt["n_pop"] / nq,
        "saturated_batch": saturated,
        "pipelined_batches": pipelined,
        "float_rows_per_query": rows["float_rows"] / nq,
        "code_rows_per_query": rows["code_rows"] / nq,
        "roofline": roofline,
    }
    plain_f32_l2 = (args.dtype, args.measure) == ("f32", "l2")
    if not args.no_build_roofline and plain_f32_l2:
        out["build"] = dict(build_roofline(args, ggnn, base), graph_build_s=build_kernel_s)
    if not args.no_datasets:
        out["recall_targets"] = {
            "target": "recall@10 >= 0.99 against each base's own exact bf_query",
            "grid": {"tau_query": list(SEARCH_TAUS), "max_iterations": list(SEARCH_ITERS)},
            "note": "queries/s AT the recall target per synthetic base: the cheapest point of "
                    "the grid that reaches it (query-kernel time, 10k-query blocking launches); "
                    "`same_settings_as_headline` is the headline's own (tau, iterations) on that "
                    "base.  The headline dataset is the easiest of these.",
            "results": recall_target_sweep(args, device, ggnn, eng, base, query, gt)}
        # the figure next to `value` for a base of SIFT1M's hardness: `value` is quoted on the
        # easiest synthetic base (local intrinsic dimension 15), published estimates for SIFT1M
        # are around 20 -- lowrank24 measures 21
        lid21 = out["recall_targets"]["results"].get("lowrank24")
        if lid21 and lid21.get("at_recall_0.99"):
            b = lid21["at_recall_0.99"]
            out["value_at_lid21"] = {
                "dataset": "lowrank24", "queries_per_s": b["queries_per_s"],
                "local_intrinsic_dimension": lid21["local_intrinsic_dimension"]["mle_k20"],
                "tau_query": b["tau_query"], "max_iterations": b["max_iterations"],
                "recall_at_10": b["recall_at_10"], "query_kernel_ms": b["query_kernel_ms"],
                "batch": f"{args.n_query} queries, blocking (as `value`)",
                "saturated_batch_queries_per_s": b.get("saturated_batch", {}).get("queries_per_s")}
        # the reference's own four SIFT1M settings (sift1m_fvecs.py:19-30 / ggnn_benchmark.cpp:
        # 196-200: tau 0.34 / 0.41 / 0.51 at 200 iterations, 0.64 at 400) on this synthetic base
        pts = {}
        for tau, iters in ((0.34, 200), (0.41, 200), (0.51, 200), (0.64, 400), (1.0, 400)):
            r = measure_point(eng, query, gt, args, 5, tau, iters)
            ids_p, _ = eng.query(query, args.k, tau, iters, measure)
            r["c_at_1"] = (ids_p[:, 0] == gt[:, 0]).float().mean().item()
            pts[f"tau={tau},iters={iters}"] = r
        out["operating_points"] = pts
    sift_dir = os.environ.get("GGNN_SIFT1M_DIR")
    if sift_dir and not args.lean:
        try:
            out["sift1m_real_files"] = sift1m_real(sift_dir, ggnn, device)
        except Exception as e:   # informational: never lose the line over it
            out["sift1m_real_files"] = {"error": repr(e)}
    if not args.no_scaling_reference and plain_f32_l2:
        out["strong_scaling_one_gpu"] = scaling_reference(args, device, ggnn, max(5, args.steps // 2))
    if not args.no_cpu_baseline:
        graph = eng.get_graph(0)
        cfg = graph.config
        g = {"graph0": graph.graph[0].view.numpy(),
             "tr": np.concatenate([t.view.numpy().reshape(-1) for t in graph.translation[1:]])}
        out["cpu_baseline"] = cpu_baseline(base, query, k, g, cfg,
                                           graph.nn1_stats.view.numpy().reshape(-1),
                                           args.tau_query, args.max_iters)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def exchange_group(args, device, world, rank, cpu_group):
    """The process group the candidate lists travel on, and what to say about it in the line.
    `--backend nccl` (the default): an RCCL group is created and ATTEMPTED with one small
    all_gather_into_tensor on the device; if that raises on ANY rank, every rank falls back to the
    host-staged gloo exchange -- LOUDLY (stderr and `exchange.fallback` in the JSON line), so that a
    run can never silently measure the copy path while the line says RCCL."""
    info = {"requested": args.backend, "fallback": False, "rccl_ranks": 0}
    if args.backend != "nccl":
        info["used"] = f"{args.backend} (candidates staged through the host)"
        return cpu_group, info
    import datetime
    err, group = "", None
    # precondition, checked on the host-side group BEFORE any RCCL call: one device per rank.  RCCL
    # refuses two ranks on a device ("Duplicate GPU detected"), and not necessarily on every rank
    # at the same moment -- the others would sit in the probe collective until its timeout
    ident = [None] * world
    props = torch.cuda.get_device_properties(device)
    dist.all_gather_object(ident, (socket.gethostname(), str(getattr(props, "uuid", device.index))),
                           group=cpu_group)
    if len(set(ident)) != world:
        err = f"ranks share a device ({len(set(ident))} distinct devices for {world} ranks)"
    else:
        try:
            # the probe runs on a group of its own with a SHORT timeout: a rank whose RCCL call
            # raised reaches the flag reduction below at once, the others after 2 minutes at most
            # (round-5 advisor finding: with the long timeout a one-sided failure was a 45-minute
            # stall in front of the "loud fallback")
            probe = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            x = torch.full((4,), float(rank), device=device)
            out = torch.empty((4 * world,), device=device)
            dist.all_gather_into_tensor(out, x, group=probe)
            torch.cuda.synchronize()
            if not torch.equal(out.view(world, 4)[:, 0].cpu(),
                               torch.arange(world, dtype=torch.float32)):
                err = "the probe all_gather returned wrong data"
        except Exception as e:  # noqa: BLE001 -- reported in the line, every rank falls back
            err = repr(e)
    flag = torch.tensor([1.0 if err else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=cpu_group)
    if flag.item() == 0:
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(minutes=45))
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        flag = torch.tensor([1.0 if err else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=cpu_group)
    if flag.item() > 0:
        info.update(fallback=True, used="gloo (candidates staged through the host)", rccl_ranks=0,
                    reason=(err or "the RCCL probe failed on another rank")[:500])
        if rank == 0:
            print("[bench] WARNING: the RCCL exchange was requested but its probe collective FAILED "
                  f"({info['reason']}); every rank falls back to the host-staged gloo exchange -- "
                  "this line does NOT measure RCCL over xGMI (rccl_ranks = 0)", file=sys.stderr,
                  flush=True)
        if args.require_rccl:
            raise SystemExit(f"--require-rccl: no RCCL group of {world} ranks ({info['reason']})")
        return cpu_group, info
    info["rccl_ranks"] = dist.get_world_size(group)
    info["used"] = "nccl = RCCL (one packed all_gather_into_tensor per batch, on the device)"
    return group, info


def sharded_case(args, device, world, rank, spg, cpu_group, xgroup=None):
    """One base of TOTAL_SHARDS x args.n_base points partitioned over the ranks (one process per
    GPU): blocking 10k-query steps (the contract's timed region), a saturating 100k-query batch
    and two batches in flight.  Times are the MAX over ranks."""
    from ggnn_amd.distributed import ShardedGGNN

    # coordination goes through a host-side (gloo) group: a rank that waits in an RCCL barrier
    # spins in a kernel on its GPU, which would disturb whatever still runs there (rank 0's
    # reference point, the one-handle child that drives all GPUs)
    def barrier():
        torch.cuda.synchronize()
        dist.barrier(group=cpu_group)
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=cpu_group)
        return [float(v) for v in t.tolist()]

    base = big_base(args, spg * args.n_base, rank * spg, device)
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    sharded = ShardedGGNN(group=xgroup)
    sharded.engine.set_base_reference(base)
    sharded.n_local = int(base.shape[0])
    sharded.set_shard_size(args.n_base)
    eng = sharded.engine
    t0 = time.perf_counter()
    sharded.build(args.k_build, args.tau_build, args.refine)
    torch.cuda.synchronize()
    build_wall_s = time.perf_counter() - t0
    build_kernel_s = eng.last_timing_ms()["build_ms"] / 1000.0

    def step(q=query):
        return sharded.query(q, args.k, args.tau_query, args.max_iters)

    gt, _ = sharded.bf_query(query, args.k)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, dists = step()
    barrier()
    elapsed, = max_over_ranks(time.perf_counter() - t0)
    blocking_parts = sharded.last_query_parts
    # the same steps as ONE batch each (the blocking mode of rounds 1-3; also the source of the
    # kernel time: the half-batches of the split run on the asynchronous lanes, which are not timed)
    sharded.split_blocking = False
    kernel_ms = []
    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(max(3, args.steps // 4)):
        ids_u, dists_u = step()
        kernel_ms.append(eng.last_timing_ms()["query_ms"])
    barrier()
    elapsed_unsplit, = max_over_ranks((time.perf_counter() - t0) / max(3, args.steps // 4))
    if not (torch.equal(ids_u, ids) and torch.equal(dists_u, dists)):
        raise RuntimeError("split and unsplit blocking query results differ")
    # work counters of one pass (untimed) for the aggregate roofline figure: bytes the kernels
    # of all ranks read by their own accounting / the slowest rank's kernel time / N x 8 TB/s
    eng.set_collect_counters(True)
    step()
    cnt, rows = eng.last_query_counters(), eng.last_query_rows_read()
    eng.set_collect_counters(False)
    sharded.split_blocking = None
    kms = float(np.mean(kernel_ms))
    tot = torch.tensor([cnt["n_dist"], cnt["n_pop"], rows["float_rows"], rows["code_rows"]],
                       dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=cpu_group)
    kmax, = max_over_ranks(kms)
    n_dist, n_pop, float_rows, code_rows = (float(v) for v in tot.tolist())
    d_, nq_ = args.dim, args.n_query
    code_dim = max(16, 1 << (d_ - 1).bit_length()) if d_ <= 64 else (d_ + 63) // 64 * 64
    shards_total = spg * world
    alg_bytes = (float_rows * d_ * 4 + code_rows * code_dim + n_pop * args.k_build * 4 +
                 shards_total * nq_ * (d_ * 4 + 32 * 4 + 8 + args.k * 8 +
                                       ((code_dim + 8) * 4 if code_rows else 0)))
    roofline = {"bound": "hbm", "achieved": alg_bytes / (kmax * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "frac": alg_bytes / (kmax * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                "bytes_per_step_all_ranks": alg_bytes, "kernel_ms_slowest_rank": kmax,
                "n_dist_per_query_all_shards": n_dist / nq_,
                "definition": "algorithmic bytes of the query kernels of ALL ranks per step (their "
                              "own row counters, as in the N=1 line) / the slowest rank's summed "
                              "kernel time (HIP events) / (N x 8 TB/s); the exchange and merge "
                              "are in ms_per_step, not in this kernel figure"}
    out = {"n_base_per_shard": args.n_base, "shards_per_gpu": spg, "roofline": roofline,
           "elapsed_s": elapsed, "ms_per_step": elapsed / args.steps * 1e3,
           "queries_per_s": args.n_query / (elapsed / args.steps),
           "blocking_half_batches_in_flight": blocking_parts,
           "blocking_as_one_batch": {"ms_per_step": elapsed_unsplit * 1e3,
                                     "queries_per_s": args.n_query / elapsed_unsplit,
                                     "results": "bit-identical to the split steps"},
           "recall_at_10": recall_at_k(ids, gt),
           "graph_build_s_per_gpu": build_kernel_s, "graph_build_wall_s": build_wall_s,
           "query_kernel_ms_sum_over_local_shards": float(np.mean(kernel_ms))}

    # informational, never part of `value`; every rank takes part in every reduction whether or
    # not its own attempt worked
    sat, err = -1.0, 0.0
    nbig = 10 * args.n_query
    try:
        big = synthetic(args.dataset, nbig, args.dim, 9876, device)
        step(big)
        barrier()
        reps = max(2, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            step(big)
        barrier()
        sat = (time.perf_counter() - t0) / reps
        del big
    except Exception as e:
        err, out["saturated_batch_error"] = 1.0, repr(e)
    sat, err = max_over_ranks(sat, err)
    if not err and sat > 0:
        out["saturated_batch"] = {"n_query": nbig, "ms_per_step": sat * 1e3,
                                  "queries_per_s": nbig / sat}
    if not args.no_pipelined:
        pip, same, err = -1.0, False, 0.0
        try:
            sharded.finish(sharded.query_async(query, args.k, args.tau_query, args.max_iters, slot=0))
            barrier()
            t0 = time.perf_counter()
            tickets, last = [], None
            for i in range(args.steps):
                tickets.append(sharded.query_async(query, args.k, args.tau_query, args.max_iters,
                                                   slot=i % 2))
                if len(tickets) == 2:
                    last = sharded.finish(tickets.pop(0))
            while tickets:
                last = sharded.finish(tickets.pop(0))
            torch.cuda.synchronize()
            pip = (time.perf_counter() - t0) / args.steps
            same = bool(torch.equal(last[0], ids) and torch.equal(last[1], dists))
        except Exception as e:
            err, out["pipelined_error"] = 1.0, repr(e)
        pip, err = max_over_ranks(pip, err)
        if not err and pip > 0:
            out["pipelined_batches"] = {"batches_in_flight": 2, "ms_per_batch": pip * 1e3,
                                        "queries_per_s": args.n_query / pip,
                                        "results_equal_blocking": same}
    del sharded, eng, base
    torch.cuda.empty_cache()
    barrier()
    return out


def speedups(case, one):
    """speed-up of every measured mode over the same mode of the one-GPU point of the same base"""
    if not one or "queries_per_s" not in one:
        return None
    r = {"blocking": case["queries_per_s"] / one["queries_per_s"]}
    for mode in ("saturated_batch", "pipelined_batches"):
        if mode in case and mode in one:
            r[mode] = case[mode]["queries_per_s"] / one[mode]["queries_per_s"]
    return r


def run_in_process(args, ggnn):
    """The reference's own multi-GPU form (examples/cpp-and-cuda/ggnn_main_multi_gpu.cpp:
    ONE handle, set_gpus([...]) + set_shard_size): host thread per GPU, in-engine RCCL
    all-gather of the packed candidates over xGMI, per-GPU slice merges.  Prints one JSON line."""
    gpus = [0] * args.gpus if args.single_device else list(range(args.gpus))
    device = torch.device("cuda", gpus[0])
    torch.cuda.set_device(device)
    base = big_base(args, TOTAL_SHARDS * args.n_base, 0, device)
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_gpus(gpus)
    eng.set_shard_size(args.n_base)
    t0 = time.perf_counter()
    eng.build(args.k_build, args.tau_build, args.refine)
    build_wall_s = time.perf_counter() - t0

    def step(q=query):
        return eng.query(q, args.k, args.tau_query, args.max_iters)

    for _ in range(max(2, args.warmup)):
        ids, dists = step()
    el, (ids, dists) = timed(step, args.steps, torch.cuda.synchronize)
    out = {"form": "one handle, set_gpus(%s), shard size %d" % (gpus, args.n_base),
           "exchange": eng.last_exchange(), "rccl_ranks": eng.rccl_ranks(),
           "queries_per_s": args.n_query / el,
           "blocking_half_batches_in_flight": eng.last_query_parts(),
           "ms_per_step": el * 1e3, "graph_build_wall_s": build_wall_s,
           "query_kernel_ms_max_over_gpus": eng.last_timing_ms()["query_ms"],
           "results": "merged [Nq, K] on the host (as the reference returns them)"}
    try:
        big = synthetic(args.dataset, 10 * args.n_query, args.dim, 9876, device)
        step(big)
        el_b, _ = timed(lambda: step(big), max(2, args.steps // 4), torch.cuda.synchronize)
        out["saturated_batch"] = {"n_query": int(big.shape[0]), "ms_per_step": el_b * 1e3,
                                  "queries_per_s": big.shape[0] / el_b}
        del big
        eng.query_async(query, args.k, args.tau_query, args.max_iters, slot=0)
        eng.synchronize()
        t0 = time.perf_counter()
        tickets = [eng.query_async(query, args.k, args.tau_query, args.max_iters, slot=i % 2)
                   for i in range(args.steps)]
        eng.synchronize()
        el_p = (time.perf_counter() - t0) / args.steps
        same = all(torch.equal(t.ids.cpu(), ids) and torch.equal(t.dists.cpu(), dists)
                   for t in tickets)
        out["pipelined_batches"] = {"batches_in_flight": 2, "ms_per_batch": el_p * 1e3,
                                    "queries_per_s": args.n_query / el_p,
                                    "results_equal_blocking": bool(same)}
    except Exception as e:
        out["secondary_error"] = repr(e)
    print(json.dumps(out), flush=True)


def in_process_child(args, n_base, timeout_s):
    """rank 0 runs the one-handle form in a child process of its own (a crash or a hang there
    must not cost the benchmark line); the other ranks idle at a barrier meanwhile"""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE",
                        "TORCHELASTIC_RUN_ID", "GGNN_EXCHANGE")}
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process", "--gpus", str(args.gpus),
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--n-base", str(n_base),
           "--n-query", str(args.n_query), "--dim", str(args.dim), "--k", str(args.k),
           "--k-build", str(args.k_build), "--tau-build", str(args.tau_build),
           "--refine", str(args.refine), "--tau-query", str(args.tau_query),
           "--max-iters", str(args.max_iters), "--dataset", args.dataset]
    if args.single_device:
        cmd.append("--single-device")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s} s"}
    except Exception as e:
        return {"error": repr(e)}


def run_sharded(args, device, ggnn, world, rank):
    """strong scaling on a fixed base of TOTAL_SHARDS shards spread over the ranks"""
    if TOTAL_SHARDS % world:
        raise SystemExit(f"--gpus must divide {TOTAL_SHARDS}")
    spg = TOTAL_SHARDS // world

    import datetime
    cpu_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=45))

    def barrier():
        torch.cuda.synchronize()
        dist.barrier(group=cpu_group)

    # optional parts are dropped (and say so) when the run is already long: the line matters more
    t_start = time.perf_counter()

    def still_time(limit_s):
        """rank 0 decides, everyone follows"""
        ok = torch.tensor([1 if time.perf_counter() - t_start < limit_s else 0], dtype=torch.int32)
        dist.broadcast(ok, src=0, group=cpu_group)
        return bool(ok.item())

    xgroup, exchange = exchange_group(args, device, world, rank, cpu_group)
    ref_steps = max(5, args.steps // 2)
    main_base, main_dim = args.n_base, args.dim
    cases = []
    skipped = []
    # secondary series: 8 shards of the N=1 configuration's shape (SIFT1M-shaped rows of 128)
    series = [(main_base, main_dim)]
    if args.secondary_n_base and (args.secondary_n_base, 128) != (main_base, main_dim):
        series.append((args.secondary_n_base, 128))
    for n_base, dim in series:
        if (n_base, dim) != (main_base, main_dim) and not still_time(args.optional_budget_s + 90):
            skipped.append(f"secondary base of 8 x {n_base} x {dim}: run already longer than "
                           f"{args.optional_budget_s + 90:.0f} s")
            break
        args.n_base, args.dim = n_base, dim
        case = sharded_case(args, device, world, rank, spg, cpu_group, xgroup)
        case["dim"] = dim
        # the one-GPU point of the series on the SAME base, measured by rank 0 while the others wait
        one = None
        if rank == 0 and not args.no_scaling_reference:
            try:
                one = one_gpu_reference(args, device, ggnn, ref_steps)
            except Exception as e:  # e.g. not enough free memory next to another job: keep the line
                one = {"error": repr(e)}
        barrier()
        # ... and the one-handle form (in-engine RCCL) in a child process of rank 0
        inproc = None
        if not args.no_in_process and (n_base, dim) == (main_base, main_dim):
            if still_time(args.optional_budget_s):
                if rank == 0:
                    inproc = in_process_child(args, n_base, args.in_process_timeout)
                barrier()
            else:
                skipped.append(f"in_process_handle: run already longer than "
                               f"{args.optional_budget_s:.0f} s")
        case["one_gpu_same_base"] = one
        case["speedup_vs_one_gpu_same_base"] = speedups(case, one)
        if inproc is not None:
            case["in_process_handle"] = inproc
            if "queries_per_s" in inproc:
                inproc["speedup_vs_one_gpu_same_base"] = speedups(inproc, one)
        cases.append(case)
    args.n_base, args.dim = main_base, main_dim

    if rank == 0:
        main, nq = cases[0], args.n_query
        total = TOTAL_SHARDS * main_base
        sp = main["speedup_vs_one_gpu_same_base"] or {}
        out = {
            "metric": "queries/sec @ recall@10 (100M-point base sharded over the GPUs, k=10)",
            "value": main["queries_per_s"], "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"fixed base of {total} points = {TOTAL_SHARDS} shards x "
                                   f"{main_base} ({total} x {args.dim} f32, {args.dataset}), "
                                   f"{nq} queries, k={args.k}, k_build={args.k_build}, "
                                   f"tau_build={args.tau_build}, refine={args.refine}, "
                                   f"tau_query={args.tau_query}, max_iterations={args.max_iters}",
                       "parallelism": f"base partitioned over {world} ranks ({spg} resident "
                                      f"shard(s) per GPU), every rank searches all queries in its "
                                      f"shards, ONE packed RCCL all-gather of the sorted candidates "
                                      f"+ device k-way merge; value = Nq / T of blocking steps"},
            "exchange": exchange,
            # ranks of the RCCL group the candidates travelled on: N, or 0 after a (loud) fallback
            "rccl_ranks": exchange.get("rccl_ranks", 0),
            "recall_at_10": main["recall_at_10"],
            "graph_build_s_per_gpu": main["graph_build_s_per_gpu"],
            "graph_build_wall_s": main["graph_build_wall_s"],
            "query_kernel_ms_sum_over_local_shards": main["query_kernel_ms_sum_over_local_shards"],
            "one_gpu_same_base": main["one_gpu_same_base"],
            "speedup_vs_one_gpu_same_base": sp.get("blocking"),
            "saturated_batch": main.get("saturated_batch"),
            "saturated_speedup_vs_one_gpu_same_base": sp.get("saturated_batch"),
            "pipelined_batches": main.get("pipelined_batches"),
            "pipelined_speedup_vs_one_gpu_same_base": sp.get("pipelined_batches"),
            "in_process_handle": main.get("in_process_handle"),
            "secondary_base": (None if len(cases) < 2 else dict(
                cases[1], note=f"the same series on {TOTAL_SHARDS} x {cases[1]['n_base_per_shard']} "
                               f"x {cases[1]['dim']} (8 shards of the N=1 configuration's shape)")),
            "roofline": main.get("roofline"), "cpu_baseline": None,
            "note": "N=1 of this command is the BASELINE single-shard configuration (1M points); "
                    "the multi-GPU series keeps the BASE fixed (north star: 100M points) instead, "
                    "so compare with one_gpu_same_base -- all 8 shards resident on one MI355X, "
                    "measured by rank 0 in this same run -- not with the N=1 `value`; the three "
                    "speed-ups compare like with like (blocking / 100k-query batch / two batches "
                    "in flight)",
        }
        for k_ in ("saturated_batch_error", "pipelined_error"):
            if k_ in main:
                out[k_] = main[k_]
        if skipped:
            out["skipped_optional_parts"] = skipped
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-base", type=int, default=None,
                    help="points per shard (default: 1M at N=1 = BASELINE configs[1]; 12.5M at "
                         "N>1 = the north star's 100M-point base in 8 shards)")
    ap.add_argument("--secondary-n-base", type=int, default=1_000_000,
                    help="N>1: a second, smaller fixed base of 8 such shards (0: skip)")
    ap.add_argument("--no-in-process", action="store_true",
                    help="N>1: skip the one-handle set_gpus([...]) form (in-engine RCCL)")
    ap.add_argument("--in-process", action="store_true",
                    help="run ONLY the one-handle form over --gpus GPUs in this process")
    ap.add_argument("--in-process-timeout", type=float, default=300.0)
    ap.add_argument("--optional-budget-s", type=float, default=330.0,
                    help="N>1: the one-handle child is skipped when the run is already longer than "
                         "this, the secondary base 90 s later (the line matters more)")
    ap.add_argument("--n-query", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=None,
                    help="default: 128 at N=1 (BASELINE configs[1], SIFT1M shape); 96 at N>1 "
                         "(BASELINE configs[3], the DEEP100M shape: 100M x 96 over the GPUs)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--k-build", type=int, default=24)
    ap.add_argument("--tau-build", type=float, default=0.5)
    ap.add_argument("--refine", type=int, default=2)
    ap.add_argument("--tau-query", type=float, default=None,
                    help="operating point (with --max-iters); default: the tuned point of the shape "
                         "(OPERATING_POINTS: the cheapest point of a tau x iterations sweep that "
                         "holds recall@10 >= 0.991 on the tuning query set AND on held-out sets; "
                         "headline 0.85 / 175)")
    ap.add_argument("--max-iters", type=int, default=None)
    ap.add_argument("--dataset", default="lowrank16")
    ap.add_argument("--dtype", default="f32", choices=("f32", "u8"),
                    help="element type of base and queries (u8: BASELINE configs[4] rows)")
    ap.add_argument("--measure", default="l2", choices=("l2", "cosine"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-datasets", action="store_true", help="skip the other synthetic bases")
    ap.add_argument("--no-build-roofline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the batches-in-flight figure (query_async)")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the 8-shards-on-one-GPU point of the strong-scaling series")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: only the headline measurement (implies the --no-* flags)")
    ap.add_argument("--saturated-batch", action="store_true",
                    help="also time a 10x larger query batch (informational; off by default so "
                         "that a rocprofv3 --stats of the default command averages only "
                         "launches of the benchmark's own batch size)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--require-rccl", action="store_true",
                    help="N > 1: exit with an error instead of the (loud) gloo fallback when the RCCL "
                         "group of N ranks cannot be created")
    ap.add_argument("--single-device", action="store_true",
                    help="testing only: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_datasets = True
        args.no_build_roofline = args.no_scaling_reference = args.no_pipelined = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.n_base is None:
        args.n_base = 12_500_000 if (world > 1 or args.in_process) else 1_000_000
    if args.dim is None:
        args.dim = 96 if (world > 1 or args.in_process) else 128
    resolve_operating_point(args, TOTAL_SHARDS if (world > 1 or args.in_process) else 1)
    if args.in_process:
        import ggnn_amd as ggnn
        ggnn.set_log_level(-1)
        run_in_process(args, ggnn)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # rank 0 measures the one-GPU point and the one-handle form while the others wait at a
        # barrier: minutes, not seconds
        patience = datetime.timedelta(minutes=45)
        # the DEFAULT group is host-side (gloo: coordination, barriers, reductions of timings); the
        # candidates travel on an RCCL group that is created and PROBED separately
        # (exchange_group), so that a failing RCCL cannot take the rendezvous down with it
        dist.init_process_group("gloo", timeout=patience)

    import ggnn_amd as ggnn
    if world > 1:
        run_sharded(args, device, ggnn, world, rank)
    else:
        run_single(args, device, ggnn)


if __name__ == "__main__":
    main()
