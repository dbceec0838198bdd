"""Condense rocprofv3 CSV output (gpurun_out/) into the tracked summaries under profiles/."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
out_dir = "profiles"
os.makedirs(out_dir, exist_ok=True)

stats = list(csv.DictReader(open(f"{src}/prof_{tag}/bench_kernel_stats.csv")))
with open(f"{out_dir}/{tag}_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 "
            "--warmup 3 --no-cpu-baseline   (MI355X, 1 GPU)\n")
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in stats:
        if float(r["Percentage"]) >= 0.001:
            w.writerow(r)

pmc = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    path = f"{src}/{name}_{tag}/bench_counter_collection.csv"
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "ggnn_amd" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter] = {"launches": len(v), "avg_kb": sum(v) / len(v),
                                          "last_kb": v[-1], "max_kb": max(v)}
workload = None
try:
    workload = json.load(open(f"{src}/bench_prof.json"))["config"]["workload"]
except Exception:
    pass
json.dump({"workload": workload, "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- "
                      "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (one pass per counter)",
           "units": "rocprofv3 FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE counts 64 B per "
                    "128 B request for 16 B/lane loads -> multiply by 2 (MI355X_MICROARCH.md, HBM)",
           "kernels": pmc}, open(f"{out_dir}/{tag}_pmc_hbm.json", "w"), indent=1)

# SQ counters of the query kernels (instruction mix, VALU utilisation)
sq = {}
for name in ("pmc_sq1", "pmc_sq2"):
    path = f"{src}/{name}_{tag}/bench_counter_collection.csv"
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "query_kernel" in k and "bf_" not in k:
            agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        for cn, v in c.items():
            sq.setdefault(k, {})[cn] = sum(v) / len(v)
if sq:
    json.dump({"workload": workload,
               "command": "rocprofv3 --pmc <SQ counters> --output-format csv -- python bench.py "
                          "--steps 3 --warmup 1 --no-cpu-baseline (two passes)",
               "units": "per launch, summed over the device; SQ cycle counters tick once per 4 "
                        "clocks, so VALU utilisation = SQ_ACTIVE_INST_VALU / (kernel time x "
                        "clock / 4 x 1024 SIMDs)",
               "kernels": sq}, open(f"{out_dir}/{tag}_pmc_sq.json", "w"), indent=1)

# MFMA counters of the brute-force kernel
path = f"{src}/pmc_mfma_{tag}/bench_counter_collection.csv"
if os.path.exists(path):
    c = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(path)):
        if "bf_mfma_kernel" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    if c:
        cnt = {k: sum(v) / len(v) for k, v in c.items()}
        d = sum(dur) / len(dur)
        busy = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        json.dump({"command": "rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
                              "SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -- "
                              "python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                   "kernel": "bf_mfma_kernel<float, L2, T=1, NU=16> (10 000 x 1 000 000 x 128 f32, k=10)",
                   "counters": cnt, "kernel_duration_s_under_profiling": d,
                   "expected_mfma_busy_cycles": "2*Nq_padded*N*D / 4096 flop per "
                       "v_mfma_f32_32x32x2_f32 * 64 cycles = 10112*1e6*128*2/4096*64 = 4.045e10",
                   "mfma_utilisation_lower_bound_at_2.4GHz": busy / (1024 * 2.4e9 * d),
                   "note": "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; the effective "
                           "clock under load is below 2.4 GHz (DVFS), so the true busy fraction "
                           "is higher"},
                  open(f"{out_dir}/{tag}_pmc_mfma_bf_query.json", "w"), indent=1)
print(open(f"{out_dir}/{tag}_kernel_stats.csv").read()[:1500])
for k, v in pmc.items():
    if "query_kernel" in k and "bf_" not in k and "FETCH_SIZE" in v:
        f = v["FETCH_SIZE"]["avg_kb"]
        wv = v.get("WRITE_SIZE", {"avg_kb": 0})["avg_kb"]
        print(k, "HBM bytes/launch (corrected):", 2 * f * 1024 + wv * 1024)
