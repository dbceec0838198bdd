"""Condense rocprofv3 CSV output (gpurun_out/<tag>/) into the tracked summaries under profiles/."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/{tag}"
extra = sys.argv[3] if len(sys.argv) > 3 else ""
out_dir = "profiles"
os.makedirs(out_dir, exist_ok=True)
CMD = "python bench.py --lean" + (" " + extra if extra else "")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha, BUILD_SOURCES  # noqa: E402  (fingerprints of the profiled kernel sources)
SHA = kernel_source_sha()
BUILD_SHA = kernel_source_sha(BUILD_SOURCES)


def counters(name):
    path = f"{src}/pmc_{name}/bench_counter_collection.csv"
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(open(path)))


workload = None
try:
    bench = json.load(open(f"{src}/bench_prof.json"))
    workload = bench["config"]["workload"]
    json.dump(bench, open(f"{out_dir}/{tag}_bench_n1_profiled.json", "w"), indent=1)
except Exception as e:
    print("no bench line:", e)

stats = list(csv.DictReader(open(f"{src}/prof/bench_kernel_stats.csv")))
with open(f"{out_dir}/{tag}_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {CMD} --steps 20 "
            "--warmup 3   (MI355X, 1 GPU)\n")
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in stats:
        if float(r["Percentage"]) >= 0.001:
            w.writerow(r)

# ---- HBM-side traffic -----------------------------------------------------------------------
pmc = {}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in counters(name):
        if r["Counter_Name"] == counter and "ggnn_amd" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter] = {"launches": len(v), "avg_kb": sum(v) / len(v),
                                          "last_kb": v[-1], "max_kb": max(v)}
json.dump({"workload": workload, "kernel_source_sha": SHA, "build_source_sha": BUILD_SHA,
           "command": f"rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- {CMD} "
                      "--steps 3 --warmup 1 (one pass per counter)",
           "units": "rocprofv3 FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE counts 64 B per "
                    "128 B request for 16 B/lane loads -> multiply by 2 (MI355X_MICROARCH.md, HBM)",
           "kernels": pmc}, open(f"{out_dir}/{tag}_pmc_hbm.json", "w"), indent=1)


def per_kernel(names, want):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for n in names:
        for r in counters(n):
            k = r["Kernel_Name"]
            if want(k):
                agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {cn: sum(v) / len(v) for cn, v in c.items()} for k, c in agg.items()}


is_query = lambda k: "query_kernel" in k and "bf_" not in k
is_trav = lambda k: any(t in k for t in ("query_kernel", "merge_kernel", "sym_kernel")) and "bf_" not in k

# ---- SQ counters of the query kernels (instruction mix, VALU utilisation) -----------------------
sq = per_kernel(("sq1", "sq2"), is_query)
if sq:
    json.dump({"workload": workload, "kernel_source_sha": SHA,
               "command": f"rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- {CMD} "
                          "--steps 3 --warmup 1 (two passes)",
               "units": "per launch, summed over the device; SQ cycle counters tick once per 4 "
                        "clocks, so VALU utilisation = SQ_ACTIVE_INST_VALU / (kernel time x "
                        "clock / 4 x 1024 SIMDs)",
               "kernels": sq}, open(f"{out_dir}/{tag}_pmc_sq.json", "w"), indent=1)

# ---- SQ counters of the CONSTRUCTION kernels, summed over the launches of the profiled build ------
# (bench.py build_roofline: which roof binds merge / sym -- fabric bytes vs HBM, own bytes vs the L2s,
# VALU issue; the SQ cycle counters tick once per 4 clocks)
build_sq = {}
for n in ("sq1", "sq2"):
    for r in counters(n):
        k = r["Kernel_Name"]
        for needle in ("merge_kernel", "sym_kernel"):
            if "::" + needle + "<" in k:
                e = build_sq.setdefault(needle, {"kernels": set(), "launches": {}, "dur_ns": {}})
                e["kernels"].add(k.split("(")[0])
                c = r["Counter_Name"]
                e[c] = e.get(c, 0.0) + float(r["Counter_Value"])
                e["launches"][c] = e["launches"].get(c, 0) + 1
                e["dur_ns"][c] = e["dur_ns"].get(c, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if build_sq:
    for needle, e in build_sq.items():
        e["kernels"] = sorted(e["kernels"])
        ln = set(e["launches"].values())
        e["launches"] = ln.pop() if len(ln) == 1 else e["launches"]
        # duration under the profiler of the pass that counted SQ_ACTIVE_INST_VALU
        e["duration_s_sq2_pass"] = e["dur_ns"].get("SQ_ACTIVE_INST_VALU", 0) * 1e-9
        del e["dur_ns"]
    json.dump({"workload": workload, "build_source_sha": BUILD_SHA,
               "command": f"rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- {CMD} "
                          "--steps 3 --warmup 1 (two passes; the command builds ONE graph)",
               "units": "summed over every launch of the kernel family in one build; cycle counters "
                        "tick once per 4 clocks: VALU issue fraction = SQ_ACTIVE_INST_VALU x 4 / "
                        "(duration_s_sq2_pass x clock x 1024 SIMDs)",
               "kernels": build_sq}, open(f"{out_dir}/{tag}_pmc_build_sq.json", "w"), indent=1)

# ---- recomputable roofline fraction of this shape ------------------------------------------------
try:
    rl = bench["roofline"]
    qk = [r for r in stats if "query_kernel" in r["Name"] and "bf_" not in r["Name"]]
    doc = {"workload": workload, "kernel_source_sha": SHA,
           "bench_line": {k: bench.get(k) for k in ("value", "ms_per_step", "recall_at_10",
                                                    "query_kernel_ms", "n_dist_per_query",
                                                    "n_pop_per_query", "float_rows_per_query",
                                                    "code_rows_per_query")},
           "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac",
                                               "bytes_per_launch", "kernel_ms", "kernel")},
           "rocprofv3_kernel_stats": [{"name": r["Name"], "calls": int(r["Calls"]),
                                       "avg_ns": float(r["AverageNs"])} for r in qk],
           "pmc_hbm_bytes_per_launch_corrected": {
               k: (2 * v.get("FETCH_SIZE", {"avg_kb": 0})["avg_kb"] * 1024
                   + v.get("WRITE_SIZE", {"avg_kb": 0})["avg_kb"] * 1024)
               for k, v in pmc.items() if is_query(k)}}
    json.dump(doc, open(f"{out_dir}/{tag}_roofline.json", "w"), indent=1)
except Exception as e:
    print("no roofline summary:", e)

# ---- L2 (TCC) hit rate and requests that left the L2 ------------------------------------------
l2 = per_kernel(("l2a", "l2b"), is_trav)
for k, c in l2.items():
    if c.get("TCC_REQ_sum"):
        c["l2_hit_rate"] = c.get("TCC_HIT_sum", 0.0) / (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 1.0))
if l2:
    json.dump({"workload": workload, "kernel_source_sha": SHA,
               "command": f"rocprofv3 --pmc TCC_* --kernel-trace --output-format csv -- {CMD} "
                          "--steps 3 --warmup 1 (two passes)",
               "units": "per launch; TCC = the 16 L2 channels per XCD; TCC_EA0_RDREQ are read "
                        "requests the L2 sent on to the fabric (Infinity Cache / HBM; the two are "
                        "not distinguishable from the GPU side), _DRAM the ones addressed to DRAM",
               "kernels": l2}, open(f"{out_dir}/{tag}_pmc_l2.json", "w"), indent=1)

# ---- MFMA counters of the brute-force kernel ---------------------------------------------------
c = collections.defaultdict(list)
dur = []
for r in counters("mfma"):
    if "bf_mfma_kernel" in r["Kernel_Name"]:
        c[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
if c:
    cnt = {k: sum(v) / len(v) for k, v in c.items()}
    d = sum(dur) / len(dur)
    busy = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    json.dump({"command": "rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
                          f"SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -- {CMD} "
                          "--steps 3 --warmup 1",
               "kernel": "bf_mfma_kernel<float, L2, T=1, NU=16> (10 000 x 1 000 000 x 128 f32, k=10)",
               "counters": cnt, "kernel_duration_s_under_profiling": d,
               "expected_mfma_busy_cycles": "2*Nq_padded*N*D / 4096 flop per "
                   "v_mfma_f32_32x32x2_f32 * 64 cycles = 10112*1e6*128*2/4096*64 = 4.045e10",
               "mfma_utilisation_lower_bound_at_2.4GHz": busy / (1024 * 2.4e9 * d),
               "note": "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; the effective "
                       "clock under load is below 2.4 GHz (DVFS), so the true busy fraction "
                       "is higher"},
              open(f"{out_dir}/{tag}_pmc_mfma_bf_query.json", "w"), indent=1)

# other bf shapes (scripts/pmc_bf.sh summaries), if collected in this round (headline tag only)
for shape in (() if extra else ("d960", "d256", "u8")):
    p = f"gpurun_out/pmc_bf_{shape}/summary.json"
    if os.path.exists(p):
        doc = json.load(open(p))
        for k, v in doc.items():
            d = v.get("kernel_duration_s_under_profiling")
            if d and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                v["mfma_busy_fraction_at_2.4GHz"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4e9 * d)
                if "SQ_INSTS_MFMA" in v and v["SQ_INSTS_MFMA"]:
                    v["valu_per_mfma"] = v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"]
                    v["salu_per_mfma"] = v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_MFMA"]
        json.dump({"command": f"scripts/pmc_bf.sh {shape} ... (rocprofv3 --pmc, three passes over "
                              "scripts/bf_time*.py: 10 000 queries x 1 000 000 rows, k=10)",
                   "kernels": doc}, open(f"{out_dir}/{tag}_pmc_bf_{shape}.json", "w"), indent=1)

print(open(f"{out_dir}/{tag}_kernel_stats.csv").read()[:1800])
for k, v in pmc.items():
    if is_query(k) and "FETCH_SIZE" in v:
        f = v["FETCH_SIZE"]["avg_kb"]
        wv = v.get("WRITE_SIZE", {"avg_kb": 0})["avg_kb"]
        print(k, "HBM bytes/launch (corrected):", 2 * f * 1024 + wv * 1024)
for k, v in l2.items():
    print(k[:80], {a: round(b, 3) for a, b in v.items()})
