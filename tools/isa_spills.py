"""Where a kernel spills: python tools/isa_spills.py file.s kernel_substring
Prints every scratch_* instruction of the kernel with the number of MFMAs before it (position inside the tile loop) and the
per-kernel resource lines.  Input: hipcc -S --cuda-device-only output."""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
mf = 0
for i in range(start, end):
    l = lines[i]
    if "v_mfma" in l: mf += 1
    if "scratch_" in l or "s_waitcnt vmcnt(0)" in l and "--waits" in sys.argv:
        print(f"line {i - start:5d}  mfma# {mf:4d}  {l.strip()}")
print("mfma total", mf)
for i in range(end, len(lines)):
    if ".name:" in lines[i] and key in lines[i]:
        for j in range(i - 12, i + 12):
            if any(k in lines[j] for k in ("vgpr_count", "spill", "private_segment", "sgpr_count", "agpr")): print(lines[j].strip())
        break
