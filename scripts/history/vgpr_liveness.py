"""Approximate VGPR liveness of one kernel from its assembly (no GPU): where the register pressure
peaks and what is live there.
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off --cuda-device-only -gline-tables-only \
        -S file.hip -o /tmp/x.s
    python scripts/vgpr_liveness.py /tmp/x.s <mangled kernel name> [top]
Backward data flow over the basic blocks (branch targets from s_cbranch / s_branch); a destination
written under a partial EXEC mask is treated as a full definition, so the numbers are a lower bound
of what the allocator sees.  Prints the peak, the registers live there with the source line of their
last definition, and the pressure at every block entry."""
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 1
files, insts, labels = {}, [], {}
inside, loc = False, "?"
for line in open(path):
    if not inside:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
        if line.startswith(kern + ":"):
            inside = True
        continue
    s = line.split(";")[0].strip()
    if s.startswith(".loc"):
        p = s.split()
        loc = f"{files.get(p[1], p[1])}:{p[2]}"
        continue
    if s.startswith("s_endpgm"):
        insts.append(("s_endpgm", "", loc))
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"(\.LBB\d+_\d+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not s or s.startswith("."):
        continue
    op, _, rest = s.partition(" ")
    insts.append((op, rest.strip(), loc))


def regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def def_use(op, rest):
    ops = [o.strip() for o in rest.split(",")] if rest else []
    # rejoin v[a:b] pieces split by the comma inside s[..] etc. (none contain commas) -- fine
    if not ops:
        return [], []
    stores = op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "flat_store",
                            "ds_add", "ds_or", "ds_min", "ds_max", "global_atomic"))
    nodef = stores or op.startswith(("v_cmp", "v_cmpx", "s_", "v_readlane", "v_readfirstlane",
                                     "buffer_wbl2", "buffer_inv", "ds_nop"))
    if op.startswith(("v_readlane", "v_readfirstlane", "v_cmp")) or (op.startswith("s_")):
        return [], [r for o in ops for r in regs(o)]
    if nodef:
        return [], [r for o in ops for r in regs(o)]
    d = regs(ops[0])
    u = [r for o in ops[1:] for r in regs(o)]
    if op.startswith(("v_swap",)):
        d = regs(ops[0]) + regs(ops[1])
        u = d[:]
    if op.startswith(("v_fmac", "v_mac", "v_dot", "v_writelane", "v_accvgpr")) or "_dpp" in op or "dpp" in rest \
            or op.startswith("v_permlane") or op.startswith("v_mov_b32_dpp"):
        u = u + d  # destination is also read (accumulate / partial write / old value)
    return d, u


n = len(insts)
# basic blocks
leaders = {0} | set(labels.values())
for i, (op, rest, _) in enumerate(insts):
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")) and i + 1 < n:
        leaders.add(i + 1)
leaders = sorted(leaders)
block_of = {}
blocks = []
for bi, st in enumerate(leaders):
    en = leaders[bi + 1] if bi + 1 < len(leaders) else n
    blocks.append((st, en))
    for i in range(st, en):
        block_of[i] = bi
succ = []
for bi, (st, en) in enumerate(blocks):
    op, rest, _ = insts[en - 1]
    s = []
    if op.startswith("s_branch"):
        s.append(block_of[labels[rest.strip()]])
    elif op.startswith("s_cbranch"):
        s.append(block_of[labels[rest.strip()]])
        if en < n:
            s.append(block_of[en])
    elif op.startswith("s_endpgm"):
        pass
    elif en < n:
        s.append(block_of[en])
    succ.append(s)
du = [def_use(op, rest) for op, rest, _ in insts]
live_in = [set() for _ in blocks]
live_out = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for bi in range(len(blocks) - 1, -1, -1):
        out = set()
        for s in succ[bi]:
            out |= live_in[s]
        live = set(out)
        st, en = blocks[bi]
        for i in range(en - 1, st - 1, -1):
            d, u = du[i]
            live -= set(d)
            live |= set(u)
        if out != live_out[bi] or live != live_in[bi]:
            live_out[bi], live_in[bi] = out, live
            changed = True
# per-instruction pressure
pressure = [0] * n
live_at = {}
for bi, (st, en) in enumerate(blocks):
    live = set(live_out[bi])
    for i in range(en - 1, st - 1, -1):
        d, u = du[i]
        pressure[i] = len(live | set(d))
        live_at[i] = set(live | set(d))
        live -= set(d)
        live |= set(u)
order = sorted(range(n), key=lambda i: -pressure[i])
print(f"{n} instructions, {len(blocks)} blocks, peak live VGPRs {pressure[order[0]]}")
shown = 0
last = -100
for i in order:
    if abs(i - last) < 50:
        continue
    last = i
    shown += 1
    print(f"\n== pressure {pressure[i]} at instruction {i}: {insts[i][0]} {insts[i][1]}   [{insts[i][2]}]")
    # last definition site of each live register (scan backwards in program order)
    sites = {}
    for r in sorted(live_at[i]):
        for j in range(i, -1, -1):
            if r in du[j][0]:
                sites[r] = insts[j][2] + " " + insts[j][0]
                break
        else:
            sites[r] = "(entry)"
    by_site = {}
    for r, s_ in sites.items():
        by_site.setdefault(s_, []).append(r)
    for s_, rs in sorted(by_site.items(), key=lambda kv: -len(kv[1])):
        print(f"   {len(rs):3d}  {s_:50s} v{rs}")
    if shown >= top:
        break
