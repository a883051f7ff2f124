import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); r=d["roofline"]
    print("ms/pstep %.3f edge %.3f ms (%.1f TF, frac %.3f) node %.3f ms (%.1f TF) alg_frac %.3f"%(d["ms_per_processor_step"], r["avg_launch_ms"], r["achieved"], r["frac"], r["node_kernel"]["avg_launch_ms"], r["node_kernel"]["achieved"], r["processor_step_algorithmic"]["frac"]))
