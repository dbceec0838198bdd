"""Static instruction mix of one kernel by source line / basic block (no GPU):
    hipcc ... --cuda-device-only -gline-tables-only -S file.hip -o /tmp/x.s
    python scripts/isa_profile.py /tmp/x.s <mangled kernel name> [blocks]
Prints, per source line (file:line of the innermost .loc), the number of VALU / SALU / LDS / VMEM
instructions, and with 'blocks' the per-basic-block totals in program order."""
import collections, re, sys
path, kern = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "lines"
files = {}
inside = False
loc = ("?", 0)
by_line = collections.defaultdict(lambda: collections.Counter())
blocks = []
cur = None
def kind(op):
    if op.startswith(("v_mfma", "v_smfma")): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
for line in open(path):
    if not inside:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        if line.startswith(kern + ":"):
            inside = True
            cur = {"label": "entry", "c": collections.Counter(), "lines": collections.Counter()}
            blocks.append(cur)
        continue
    if line.startswith(".Lfunc_end") or line.startswith("\t.section") and "rodata" in line:
        break
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"(\.LBB[0-9_]+):", line)
    if m:
        cur = {"label": m.group(1), "c": collections.Counter(), "lines": collections.Counter()}
        blocks.append(cur)
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", line)
    if m and not line.strip().startswith((".", ";")):
        k = kind(m.group(1))
        by_line[loc][k] += 1
        cur["c"][k] += 1
        cur["lines"][loc] += 1
if mode == "blocks":
    for b in blocks:
        c = b["c"]
        if sum(c.values()) == 0: continue
        top = ", ".join(f"{f}:{l}x{n}" for (f, l), n in b["lines"].most_common(3))
        print(f"{b['label']:14s} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} wait {c['wait']:3d}  | {top}")
else:
    tot = collections.Counter()
    for (f, l), c in sorted(by_line.items()):
        print(f"{f}:{l:5d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} smem {c['smem']:3d} wait {c['wait']:3d}")
        tot.update(c)
    print("total", dict(tot))
