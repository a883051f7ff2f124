"""Same-box A/B of library variants on the cylinder_flow-sized mesh (hipGraph replay of 15 processor steps):
python tools/ab_small.py name1 name2 ... ; each variant is lib/variants/<name>.so ("default" = the in-tree library)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import mgn_amd, bench
pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
for _ in range(5): eng.processor_steps_dev(15)
eng.synchronize(); t = time.perf_counter()
for _ in range(100): eng.processor_steps_dev(15)
eng.synchronize(); print("US_PER_STEP", (time.perf_counter() - t) / 1500 * 1e6)
''' % ROOT
names = sys.argv[1:]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        env = dict(os.environ)
        lib, *sets = n.split("+")                 # "name+VAR=value+...": environment knobs on top of a library variant
        for kv in sets:
            k, v = kv.split("=", 1)
            env[k] = v
        if lib != "default":
            env["MGN_LIB_PATH"] = os.path.join(ROOT, "meshgraphnets.jl_amd", "lib", "variants", lib + ".so")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        l = [x for x in out.stdout.splitlines() if x.startswith("US_PER_STEP")]
        if l: res[n].append(float(l[0].split()[1]))
        else: print(n, "FAILED", out.stderr[-300:])
for n in names:
    if res[n]: print(f"{n:20s} us/step min {min(res[n]):.1f} med {sorted(res[n])[len(res[n])//2]:.1f}")
