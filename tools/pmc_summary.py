"""Per-kernel averages of the rocprofv3 PMC passes collected by profiles/run_profiles.sh (one directory per pass under
gpurun_out/prof_<tag>/pmc_*), plus the derived values quoted in DESIGN.md / bench.py:
    python tools/pmc_summary.py gpurun_out/prof_r01 > profiles/r01/pmc_summary_bench_1m.json
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; gfx950 correction of MI355X_MICROARCH.md's HBM
section: FETCH_SIZE reports half the bytes of wide coalesced reads -- an upper bound where rows are gathered);
MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8: the counter is summed over the 8 XCDs);
L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(root, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        e = acc[r["Kernel_Name"]][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
out = {}
for k, ctrs in acc.items():
    if "mgn::k_" not in k or "randn" in k:
        continue
    d = {c: {"avg_per_launch": v[1] / v[0], "launches": v[0]} for c, v in ctrs.items()}
    g = lambda c: d[c]["avg_per_launch"] if c in d else None
    der = {}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        fb, wb = g("FETCH_SIZE") * 1024.0, g("WRITE_SIZE") * 1024.0
        der.update(fetch_bytes_as_reported=fb, fetch_bytes_x2_corrected=2 * fb, write_bytes=wb, hbm_bytes_per_launch_corrected=2 * fb + wb,
                   note="FETCH_SIZE on gfx950 reads 1/2 of the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM section); x2 applied "
                        "as prescribed; the row-gather part is uncalibrated, so this is an upper bound")
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        cyc = g("GRBM_GUI_ACTIVE") / 8.0                      # summed over the 8 XCDs
        der.update(mfma_pipe_busy_frac=g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), cycles_per_launch=cyc)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        der["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_INSTS_VALU") is not None and g("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None:
        der["valu_instructions_per_launch"] = g("SQ_INSTS_VALU")
    d["derived"] = der
    out[k] = d
json.dump(out, sys.stdout, indent=1)
