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
    if "scattered_labels" in d:
        sc = d["scattered_labels"]
        for k in ("renumbered_f32", "kept_f32", "renumbered_bf16", "kept_bf16"):
            print("  scattered %-16s %.3f ms/pstep, edge %.3f ms, node side %.3f ms, set_graph %.2f s" % (
                k, sc[k]["ms_per_processor_step"], sc[k]["edge_kernel_ms"], sc[k]["node_side_ms"], sc[k]["graph_setup_s"]))
        print("  scattered vs coherent: f32 x%.3f, bf16 x%.3f" % (sc["vs_coherent_f32"], sc["vs_coherent_bf16"]))
    sec = d.get("secondary", {})
    if sec:
        print("  M-cyl %.1f us/pstep | rollout Euler %.1f ms, Tsit5 %.1f ms | train step %.2f ms (%.2f with new parameters every step)" % (
            sec["us_per_processor_step"], sec["rollout_100_saves"]["Euler"]["ms_per_rollout"], sec["rollout_100_saves"]["Tsit5"]["ms_per_rollout"],
            sec.get("train_step", {}).get("ms_per_step", 0.0), sec.get("train_step", {}).get("ms_per_iteration_with_new_params", 0.0)))
