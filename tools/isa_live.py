"""Rough live-register picture of a kernel region: python tools/isa_live.py file.s kernel_substring mfma_from mfma_to
Backward liveness over the straight-line region between the mfma_from-th and mfma_to-th MFMA (no branches assumed inside): prints
max / mean live VGPR count and the live count at every 48th MFMA."""
import re, sys
src, key, m0, m1 = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
ins = []
mf = 0
for i in range(start, end):
    l = lines[i].split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."): continue
    if l.startswith("v_mfma"): mf += 1
    if mf < m0 or mf > m1: continue
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if not ops or not rest: continue
    stores = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "v_cmp", "v_readfirstlane", "v_readlane"))
    if stores: d, u = [], regs(rest)
    else: d, u = regs(ops[0]), regs(",".join(ops[1:]))
    if op.startswith(("v_fmac", "v_mac")) or "dpp" in op and op.startswith("v_fmac"): u += d
    ins.append((op, d, u, mf))
live = set()
for sweep in range(2):       # second sweep: what the region's own start uses is live at its end (a loop body)
    counts = []
    for op, d, u, m in reversed(ins):
        for r in d: live.discard(r)
        for r in u: live.add(r)
        counts.append((m, len(live)))
    counts.reverse()
print("instructions", len(ins), "max live", max(c for _, c in counts), "mean", sum(c for _, c in counts) / len(counts))
seen = set()
for m, c in counts:
    if m % 24 == 0 and m not in seen:
        seen.add(m); print(f"  mfma {m}: live {c}")
