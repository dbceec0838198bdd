"""Register / LDS / occupancy table of every kernel of one .hip file (cross-compiles, no GPU):
    python scripts/kernel_resources.py ggnn_amd/csrc/bf_mfma.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off",
                      "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
                     + sys.argv[3:], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("ggnn_amd::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name[:100]:100s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} "
          f"spill v{r.get('VGPRs Spill','?')} s{r.get('SGPRs Spill','?')} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
