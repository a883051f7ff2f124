"""us per processor step on small jittered grids with the 16-row cooperative kernels on / off (MGN_COOP16), fp32, L = 128."""
import os, subprocess, sys, json
code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import torch, mgn_amd, bench
ps = bench.glorot_params()
out = {}
for nx in (24, 32, 45, 64, 90, 128):
    pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
    eng.set_params(ps); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
    for _ in range(5): eng.processor_steps_dev(15)
    eng.synchronize(); best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40): eng.processor_steps_dev(15)
        eng.synchronize(); best = min(best, (time.perf_counter() - t0) / 600)
    out[pos.shape[0]] = round(best * 1e6, 1)
    eng.close()
print(out)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = {"MGN_C16_EDGE_TILES_PER_CU": "99", "MGN_C16_NODE_TILES_PER_CU": "99"}
for name, env in (("32-row cooperative (MGN_COOP16=0)", {"MGN_COOP16": "0"}), ("default thresholds", {}), ("16-row at every size", ALL),
                  ("16-row at every size, RT=1 through the RT kernel", dict(ALL, MGN_C16_M1="1"))) * 2:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    print("%-50s" % name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
