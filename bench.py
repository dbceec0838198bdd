#!/usr/bin/env python3
"""Benchmark of the hot path: queries/sec at recall@10 on a SIFT1M-shaped base (+ build time).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: `query()` of the whole
query set (10 000 x 128 f32, k=10) against the resident graph; inputs are in HBM when the timed
region starts.

N = 1: BASELINE.json configs[1] (1M x 128 f32, k_build 24, tau_b 0.5; k = 10) on one MI355X.
N > 1: STRONG scaling on a FIXED base of 8 shards x 1M points (BASELINE configs[3]/[4] shape:
the base partitioned across the GPUs of one node).  Rank r owns shards [r*8/N, (r+1)*8/N) as
resident shards of one engine, every rank searches the full query set in its shards, the sorted
per-rank candidates are exchanged with ONE RCCL all-gather and merged on the device.  `value` is
queries/s (Nq / T), not shard-searches; the same line carries the one-GPU figure for the SAME base
(all 8 shards resident on rank 0's GPU, measured in the same run) and the speed-up over it.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the resource that binds the query kernel (VALU issue, from the committed SQ
                  counters of this workload) next to its byte rates (algorithmic / measured)
  cpu_baseline -- the CPU oracle (a port of the reference algorithm; the reference has no CPU
                  path) timed on a bounded sample on this box's host cores (N=1, rank 0 only)
"""
import argparse
import glob
import json
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
CLOCK_HZ = 2.4e9          # nominal engine clock; SQ cycle counters tick once per 4 clocks
N_SIMD = 1024             # 256 CUs x 4 SIMDs
F32_MFMA_PEAK = 157.3e12  # dense f32 MFMA
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


def recall_at_k(ids, gt):
    return (ids.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()


def cpu_baseline(base, query, k, graph, cfg, stats, tau, iters, budget_s=12.0):
    """Oracle timed on the host cores: brute force (the 'reference CPU brute force' of
    BASELINE.json, a port because the reference has none; cache-blocked, AVX2) and the
    traversal port."""
    from oracle import oracle as orc
    orc.set_fast_distance(True)  # plain loops, not the lockstep emulation used for parity
    cores = os.cpu_count() or 1
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
    return {"value": n / bf_s, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"oracle bf_query (cache-blocked port of bf_query_layer.cu), first {n} of "
                      f"{q_h.shape[0]} queries x {base_h.shape[0]} base rows, {bf_s:.1f} s",
            "gflops": flops / bf_s / 1e9,
            "traversal_port_qps": nq_t / tr_s,
            "traversal_sample": f"oracle query on the GPU-built graph, {nq_t} queries, "
                                f"{tr_s:.1f} s"}


def workload_string(args):
    return (f"{args.dataset} SIFT1M-shaped {args.n_base}x{args.dim} f32 per GPU, "
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
            return doc
    return None


def _query_kernel_entry(doc, prescreened):
    for name, c in doc["kernels"].items():
        if "query_kernel" not in name or "bf_query" in name:
            continue
        if ("NoPrescreen" in name) == prescreened:
            continue
        return name, c
    return None, None


def pmc_traffic(args, prescreened):
    """HBM-side bytes per query_kernel launch from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/*_pmc_hbm.json), corrected
    as MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE x2 on gfx950 for 16 B/lane loads).
    Only valid for the default workload; otherwise null."""
    doc = _latest_profile("_pmc_hbm.json", args)
    if not doc:
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
        return None, None
    name, c = _query_kernel_entry(doc, prescreened)
    return (c, doc["_file"]) if c else (None, None)


def measure_point(eng, query, gt, args, steps, tau=None, iters=None):
    """kernel time (HIP events inside the engine) and recall of one operating point"""
    tau = args.tau_query if tau is None else tau
    iters = args.max_iters if iters is None else iters
    for _ in range(2):
        eng.query(query, args.k, tau, iters)
    ms = []
    for _ in range(steps):
        ids, _ = eng.query(query, args.k, tau, iters)
        ms.append(eng.last_timing_ms()["query_ms"])
    m = float(np.mean(ms))
    return {"query_kernel_ms": m, "queries_per_s": query.shape[0] / (m * 1e-3),
            "recall_at_10": recall_at_k(ids, gt)}


def dataset_sweep(args, device, ggnn):
    """the same operating point on other synthetic bases (the headline dataset is the easiest):
    recall and query-kernel rate per dataset, each with its own exact ground truth"""
    out = {}
    for kind in ("lowrank16", "lowrank24", "lowrank32", "iid"):
        if kind == args.dataset:
            continue
        base = synthetic(kind, args.n_base, args.dim, 1234, device)
        query = synthetic(kind, args.n_query, args.dim, 4321, device)
        eng = ggnn.GGNN()
        eng.set_base_reference(base)
        eng.set_return_results_on_gpu(True)
        eng.build(args.k_build, args.tau_build, args.refine)
        gt, _ = eng.bf_query(query, args.k)
        r = measure_point(eng, query, gt, args, 5)
        r["graph_build_s"] = eng.last_timing_ms()["build_ms"] / 1000.0
        # a higher-effort point as well: where the recall of the harder bases goes
        r["tau1.0_iters400"] = measure_point(eng, query, gt, args, 3, 1.0, 400)
        out[kind] = r
        del eng, base, query
        torch.cuda.empty_cache()
    return out


def build_roofline(args, eng, base):
    """merge kernel (75 % of the build): the final (3 -> 0) launch replayed on the built graph
    through the operator seam with work counters; bytes by SURVEY 8(d)'s per-kernel formula."""
    import ctypes as C
    from ggnn_amd import _lib, ops
    from ggnn_amd._lib import check, lib
    view = _lib.GraphView()
    check(lib().ggnn_get_graph(eng._h, 0, C.byref(view)))
    cfg = view.config
    N, K, D = cfg.Ns[0], cfg.KBuild, args.dim
    dev = base.device
    gb = torch.empty((N, K), dtype=torch.int32, device=dev)
    nn1 = torch.zeros(N, device=dev)
    nd = torch.zeros(N, dtype=torch.int32, device=dev)
    out = {}
    ps = ops.prescreen_encode(base)
    for label, pre in (("plain", None), ("prescreened", ps)):
        ms = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if pre is None:
                check(lib().ggnn_op_merge(base.data_ptr(), 0, 0, cfg, view.graph, view.translation,
                                          view.selection, view.nn1_stats, args.tau_build, 3, 0,
                                          gb.data_ptr(), nn1.data_ptr(), nd.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
            else:
                check(lib().ggnn_op_merge_prescreened(
                    base.data_ptr(), pre[0].data_ptr(), pre[1].data_ptr(), 0, cfg, view.graph,
                    view.translation, view.selection, view.nn1_stats, args.tau_build, 3, 0,
                    gb.data_ptr(), nn1.data_ptr(), nd.data_ptr(),
                    torch.cuda.current_stream().cuda_stream))
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ms.append(e0.elapsed_time(e1))
        t = float(np.mean(ms)) * 1e-3
        n_dist = int(nd.sum().item())
        # every evaluation reads one 4D-byte row in the reference's algorithm; graph rows and
        # translation entries are < 2 % and left out
        ref_bytes = n_dist * D * 4
        out[label] = {"ms": t * 1e3, "n_dist_per_point": n_dist / N, "points_per_s": N / t,
                      "reference_algorithm_GBs": ref_bytes / t / 1e9,
                      "frac_of_hbm_peak": ref_bytes / t / 1e9 / HBM_PEAK_GBS}
    out["note"] = ("merge_kernel (3 -> 0) on the final graph of this run, 1M points; 'plain' reads "
                   "a 4D-byte row per evaluation (rates above the HBM peak are L2 / Infinity-Cache "
                   "hits: consecutive points share neighbourhoods); 'prescreened' is the kernel "
                   "the build uses (same results, most rows replaced by D-byte code rows), its "
                   "GB/s is the reference algorithm's bytes over its time = bytes avoided")
    return out


def scaling_reference(args, device, ggnn, steps):
    """the strong-scaling series' one-GPU point: all TOTAL_SHARDS shards resident on this GPU"""
    base = torch.cat([synthetic(args.dataset, args.n_base, args.dim, 1234 + s, device)
                      for s in range(TOTAL_SHARDS)])
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    eng = ggnn.GGNN()
    eng.set_base_reference(base)
    eng.set_shard_size(args.n_base)
    eng.set_return_results_on_gpu(True)
    eng.build(args.k_build, args.tau_build, args.refine)
    build_s = eng.last_timing_ms()["build_ms"] / 1000.0
    gt, _ = eng.bf_query(query, args.k)
    from ggnn_amd import ops

    def step():
        ids, dists = eng.query(query, args.k, args.tau_query, args.max_iters)
        # results on the GPU are the sorted [Nq, K * shards] rows: the answer is their head
        return ids[:, :args.k], dists[:, :args.k]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ids, _ = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    out = {"workload": f"{TOTAL_SHARDS} resident shards x {args.n_base} points on ONE GPU "
                       f"({TOTAL_SHARDS * args.n_base} x {args.dim} f32), {args.n_query} queries",
           "queries_per_s": args.n_query / el, "ms_per_step": el * 1e3,
           "recall_at_10": recall_at_k(ids.contiguous(), gt), "graph_build_s": build_s}
    del eng, base
    torch.cuda.empty_cache()
    return out


def run_single(args, device, ggnn):
    base = synthetic(args.dataset, args.n_base, args.dim, 1234, device)
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    eng = ggnn.GGNN()
    eng.set_base(base)
    eng.set_return_results_on_gpu(True)
    t0 = time.perf_counter()
    eng.build(args.k_build, args.tau_build, args.refine)
    torch.cuda.synchronize()
    build_wall_s = time.perf_counter() - t0
    build_kernel_s = eng.last_timing_ms()["build_ms"] / 1000.0

    def step():
        return eng.query(query, args.k, args.tau_query, args.max_iters)

    # ground truth by exact brute force on the same data (untimed)
    gt, _ = eng.bf_query(query, args.k)
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
    held = synthetic(args.dataset, args.n_query, args.dim, 8642, device)
    held_gt, _ = eng.bf_query(held, args.k)
    held_ids, _ = eng.query(held, args.k, args.tau_query, args.max_iters)
    recall_heldout = recall_at_k(held_ids, held_gt)
    del held, held_gt, held_ids

    # work counters of one pass (untimed extra run) for the roofline figure
    eng.set_collect_counters(True)
    eng.query(query, args.k, args.tau_query, args.max_iters)
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
        big = synthetic(args.dataset, 10 * args.n_query, args.dim, 9876, device)
        for _ in range(2):
            eng.query(big, args.k, args.tau_query, args.max_iters)
        sat_ms = []
        for _ in range(3):
            eng.query(big, args.k, args.tau_query, args.max_iters)
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
            eng.query_async(query, args.k, args.tau_query, args.max_iters, slot=slot)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [eng.query_async(query, args.k, args.tau_query, args.max_iters, slot=i % 2)
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
    ms_per_step = elapsed / args.steps * 1000.0
    value = nq / (elapsed / args.steps)
    # bytes_q = D*s + n_dist*D*s + n_pop*KBuild*4 + S*4 + 8 + K*8 is what the
    # reference's algorithm moves.  With the exact pre-screen a distance evaluation
    # reads a D-byte code row and only the candidates that pass it read their 4D-byte float row:
    # the algorithmic bytes of THIS kernel are counted from its own row counters.
    fixed = nq * d * 4 + cnt["n_pop"] * args.k_build * 4 + nq * (32 * 4 + 8 + k * 8)
    ref_bytes = fixed + cnt["n_dist"] * d * 4
    code_dim = (d + 15) // 16 * 16
    prescreened = rows["code_rows"] > 0
    alg_bytes = fixed + rows["float_rows"] * d * 4 + rows["code_rows"] * code_dim
    if prescreened:
        alg_bytes += nq * (code_dim + 8) * 4  # per-dimension offsets + header, per query
    avg_kernel_ms = float(np.mean(kernel_ms))
    t_kernel = avg_kernel_ms * 1e-3
    alg_gbs = alg_bytes / t_kernel / 1e9
    traffic = pmc_traffic(args, prescreened)
    sq, sq_file = pmc_sq(args, prescreened)
    kernel_name = ("query_kernel<float,16,2,1,L2,Prescreen<8,1>,HB=1>" if prescreened
                   else "query_kernel<float,16,2,1,L2,NoPrescreen>")
    hbm_block = {
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "own_algorithmic": {"bytes_per_launch": alg_bytes, "achieved": alg_gbs,
                            "frac": alg_gbs / HBM_PEAK_GBS},
        "measured_fabric_traffic": (None if traffic is None else {
            "bytes_per_launch": traffic, "achieved": traffic / t_kernel / 1e9,
            "frac": traffic / t_kernel / 1e9 / HBM_PEAK_GBS,
            "note": "FETCH_SIZE x2 + WRITE_SIZE of the committed PMC passes; Infinity-Cache "
                    "hits are included (see profiles/*_pmc_l2.json for the L2 hit rate)"}),
        "reference_algorithm_equivalent": {
            "bytes_per_launch": ref_bytes, "achieved": ref_bytes / t_kernel / 1e9,
            "frac": ref_bytes / t_kernel / 1e9 / HBM_PEAK_GBS,
            "note": "bytes AVOIDED, not moved: SURVEY 8(d)'s n_dist x 4D formula over this "
                    "kernel's time; > 1 is possible because the exact pre-screen reads D-byte "
                    "code rows for most evaluations"},
    }
    if sq and "SQ_ACTIVE_INST_VALU" in sq:
        peak = N_SIMD * CLOCK_HZ / 4.0
        achieved = sq["SQ_ACTIVE_INST_VALU"] / t_kernel
        roofline = {"bound": "valu", "achieved": achieved / 1e9, "peak": peak / 1e9,
                    "unit": "G VALU-busy SIMD quad-cycles/s", "frac": achieved / peak,
                    "traffic": traffic, "kernel": kernel_name, "counters_from": sq_file,
                    "valu_insts_per_pop": sq.get("SQ_INSTS_VALU", 0.0) / max(1, cnt["n_pop"]),
                    "note": "the kernel is VALU-issue bound: SQ_ACTIVE_INST_VALU per launch "
                            "(committed PMC pass of this same workload) / (live HIP-event kernel "
                            "time x 1024 SIMDs x 2.4 GHz / 4); byte rates in `hbm`",
                    "hbm": hbm_block}
    else:
        roofline = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": alg_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
                    "note": "no committed SQ counters for this workload: byte rate of the "
                            "kernel's own algorithmic bytes (the default workload reports the "
                            "VALU-issue fraction that actually binds)", "hbm": hbm_block}
    roofline["without_prescreen"] = (None if plain_ms is None else {
        "kernel": "query_kernel<float,16,2,1,L2,NoPrescreen>", "query_kernel_ms": plain_ms,
        "queries_per_s": nq / (plain_ms * 1e-3), "bound": "hbm + infinity cache",
        "achieved": ref_bytes / (plain_ms * 1e-3) / 1e9,
        "frac": ref_bytes / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc_traffic(args, False),
        "results": "bit-identical to the pre-screened run"})

    out = {
        "metric": "queries/sec @ recall@10 (SIFT1M-shaped, k=10)",
        "value": value, "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_string(args), "parallelism": "single GPU"},
        "recall_at_10": recall, "recall_at_10_heldout_queries": recall_heldout, "c_at_1": c1,
        "graph_build_s": build_kernel_s, "graph_build_wall_s": build_wall_s,
        "bf_query_ms": bf_ms,
        "bf_query": {"ms": bf_ms,
                     "kernel": "bf_mfma_kernel (v_mfma_f32_32x32x2_f32) + certified exact re-rank",
                     "tflops": 2.0 * nq * args.n_base * d / (bf_ms * 1e-3) / 1e12,
                     "mfma_frac_of_f32_peak": 2.0 * nq * args.n_base * d / (bf_ms * 1e-3) / F32_MFMA_PEAK,
                     "queries_rescanned_by_the_exact_scan": bf_rescanned},
        "query_kernel_ms": avg_kernel_ms,
        "n_dist_per_query": cnt["n_dist"] / nq, "n_pop_per_query": cnt["n_pop"] / nq,
        "saturated_batch": saturated,
        "pipelined_batches": pipelined,
        "float_rows_per_query": rows["float_rows"] / nq,
        "code_rows_per_query": rows["code_rows"] / nq,
        "roofline": roofline,
    }
    if not args.no_build_roofline:
        out["build"] = {"graph_build_s": build_kernel_s, "merge_kernel": build_roofline(args, eng, base)}
    if not args.no_datasets:
        ds = dataset_sweep(args, device, ggnn)
        ds[args.dataset] = {"query_kernel_ms": avg_kernel_ms, "queries_per_s": nq / t_kernel,
                            "recall_at_10": recall, "graph_build_s": build_kernel_s}
        out["datasets"] = {"operating_point": f"tau_query={args.tau_query}, "
                                              f"max_iterations={args.max_iters}, k={args.k}",
                           "note": "same engine settings on other synthetic bases (latent "
                                   "dimension 24 / 32, i.i.d.): harder bases need more effort; "
                                   "the second point of each is tau 1.0 / 400 iterations",
                           "results": ds}
    if not args.no_datasets:
        # the reference's own four SIFT1M settings (sift1m_fvecs.py:19-30 / ggnn_benchmark.cpp:
        # 196-200: tau 0.34 / 0.41 / 0.51 at 200 iterations, 0.64 at 400) and two higher-effort
        # points on this synthetic base: recall against the exact ground truth, kernel rate
        pts = {}
        for tau, iters in ((0.34, 200), (0.41, 200), (0.51, 200), (0.64, 400), (0.9, 175), (1.0, 400)):
            r = measure_point(eng, query, gt, args, 5, tau, iters)
            ids_p, _ = eng.query(query, args.k, tau, iters)
            r["c_at_1"] = (ids_p[:, 0] == gt[:, 0]).float().mean().item()
            pts[f"tau={tau},iters={iters}"] = r
        out["operating_points"] = pts
    if not args.no_scaling_reference:
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


def run_sharded(args, device, ggnn, world, rank):
    """strong scaling on a fixed base: TOTAL_SHARDS shards spread over the ranks"""
    from ggnn_amd.distributed import ShardedGGNN
    if TOTAL_SHARDS % world:
        raise SystemExit(f"--gpus must divide {TOTAL_SHARDS}")
    spg = TOTAL_SHARDS // world

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    base = torch.cat([synthetic(args.dataset, args.n_base, args.dim, 1234 + rank * spg + s, device)
                      for s in range(spg)])
    query = synthetic(args.dataset, args.n_query, args.dim, 4321, device)
    sharded = ShardedGGNN()
    sharded.set_base(base, is_local_slice=True)
    sharded.set_shard_size(args.n_base)
    eng = sharded.engine
    t0 = time.perf_counter()
    sharded.build(args.k_build, args.tau_build, args.refine)
    torch.cuda.synchronize()
    build_wall_s = time.perf_counter() - t0
    build_kernel_s = eng.last_timing_ms()["build_ms"] / 1000.0

    def step():
        return sharded.query(query, args.k, args.tau_query, args.max_iters)

    gt, _ = sharded.bf_query(query, args.k)
    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, dists = step()
        kernel_ms.append(eng.last_timing_ms()["query_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64,
                     device=device if args.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    recall = recall_at_k(ids, gt)

    # informational: two batches in flight per rank (local search of batch i+1 enqueued before
    # batch i is exchanged and merged); not part of `value`
    pipelined = None
    if not args.no_pipelined:
        pip, same, err = -1.0, False, None
        try:  # informational only: never lose the main line over it
            sharded.finish(sharded.query_async(query, args.k, args.tau_query, args.max_iters, slot=0))
            torch.cuda.synchronize()
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
            err = repr(e)
        # every rank takes part in this reduction whether or not its attempt worked
        tp = torch.tensor([pip, 0.0 if err else 1.0], dtype=torch.float64,
                          device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        worst = float(tp[0].item())
        if err is None and worst > 0:
            pipelined = {"batches_in_flight": 2, "ms_per_batch": worst * 1e3,
                         "queries_per_s": args.n_query / worst, "results_equal_blocking": same}
        else:
            pipelined = {"error": err or "failed on another rank"}
    del sharded, eng, base
    torch.cuda.empty_cache()

    # the one-GPU point of the series on the SAME base, measured by rank 0 while the others wait
    one = None
    if rank == 0 and not args.no_scaling_reference:
        try:
            one = scaling_reference(args, device, ggnn, max(5, args.steps // 2))
        except Exception as e:  # e.g. not enough free memory next to another job: keep the line
            one = {"error": repr(e)}
    barrier()

    if rank == 0:
        nq = args.n_query
        value = nq / (elapsed / args.steps)
        out = {
            "metric": "queries/sec @ recall@10 (SIFT1M-shaped shards, k=10)",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1000.0,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"fixed base of {TOTAL_SHARDS} shards x {args.n_base} points "
                                   f"({TOTAL_SHARDS * args.n_base} x {args.dim} f32, "
                                   f"{args.dataset}), {nq} queries, k={args.k}, "
                                   f"k_build={args.k_build}, tau_build={args.tau_build}, "
                                   f"refine={args.refine}, tau_query={args.tau_query}, "
                                   f"max_iterations={args.max_iters}",
                       "parallelism": f"base partitioned over {world} ranks ({spg} resident "
                                      f"shard(s) per GPU), every rank searches all queries in its "
                                      f"shards, one RCCL all-gather of the sorted candidates + "
                                      f"device k-way merge; value = Nq / T"},
            "recall_at_10": recall,
            "graph_build_s_per_gpu": build_kernel_s, "graph_build_wall_s": build_wall_s,
            "query_kernel_ms_sum_over_local_shards": float(np.mean(kernel_ms)),
            "one_gpu_same_base": one,
            "speedup_vs_one_gpu_same_base": (None if not one or "queries_per_s" not in one
                                             else value / one["queries_per_s"]),
            "pipelined_batches": pipelined,
            "pipelined_speedup_vs_one_gpu_same_base": (
                None if not one or "queries_per_s" not in one or not pipelined
                or "queries_per_s" not in pipelined
                else pipelined["queries_per_s"] / one["queries_per_s"]),
            "roofline": None, "cpu_baseline": None,
            "note": "N=1 of this command is the BASELINE single-shard configuration; the "
                    "multi-GPU series keeps the BASE fixed (8 shards) instead, so compare with "
                    "one_gpu_same_base (also in the N=1 line as strong_scaling_one_gpu), not "
                    "with the N=1 `value`",
        }
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-base", type=int, default=1_000_000, help="points per shard")
    ap.add_argument("--n-query", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--k-build", type=int, default=24)
    ap.add_argument("--tau-build", type=float, default=0.5)
    ap.add_argument("--refine", type=int, default=2)
    ap.add_argument("--tau-query", type=float, default=0.9)
    ap.add_argument("--max-iters", type=int, default=175)
    ap.add_argument("--dataset", default="lowrank16")
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
    ap.add_argument("--single-device", action="store_true",
                    help="testing only: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_datasets = True
        args.no_build_roofline = args.no_scaling_reference = args.no_pipelined = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import ggnn_amd as ggnn
    if world > 1:
        run_sharded(args, device, ggnn, world, rank)
    else:
        run_single(args, device, ggnn)


if __name__ == "__main__":
    main()
