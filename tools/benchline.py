"""Pretty-print a bench.py JSON line: python tools/benchline.py FILE   (never reads stdin: a forgotten pipe must not hang a GPU box)"""
import json, sys
if len(sys.argv) < 2:
    raise SystemExit("usage: benchline.py FILE")
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line); r = d["roofline"]
    print("ms/pstep %.3f (median %.3f)  %.3e edges/s | edge %.3f ms (%.1f %s, frac %.3f) | node side %.3f ms | step frac %.3f" % (
        d["ms_per_processor_step"], d.get("median", {}).get("ms_per_step", 0.0) / 15.0, d["value"], r["avg_launch_ms"], r["achieved"], r["unit"], r["frac"],
        r["node_side"]["avg_launch_ms"], r["processor_step"]["frac"]))
    for k in ("fp32_mfma_path", "bf16"):
        if k in d:
            print("  %-15s %.3f ms/pstep, edge %.3f ms, node side %.3f ms" % (k, d[k]["ms_per_processor_step"], d[k]["edge_kernel_ms"], d[k]["node_side_ms"]))
