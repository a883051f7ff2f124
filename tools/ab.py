"""Same-box A/B of library variants: python tools/ab.py name1 name2 ... [--rounds R] [--nx NX]
Each variant is meshgraphnets.jl_amd/lib/variants/<name>.so (build with build.build_variant).  Runs the
1M-mesh processor bench in interleaved rounds in ONE process per variant-round and prints edge/node ms."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 3
nx = 1000
dtype = "f32"
for i, a in enumerate(sys.argv):
    if a == "--rounds": rounds = int(sys.argv[i + 1]); names.remove(sys.argv[i + 1])
    if a == "--nx": nx = int(sys.argv[i + 1]); names.remove(sys.argv[i + 1])
    if a == "--dtype": dtype = sys.argv[i + 1]; names.remove(sys.argv[i + 1])
res = {n: [] for n in names}
for r in range(rounds):
    for n in names:
        env = dict(os.environ)
        lib, *sets = n.split("+")                 # "name+VAR=value+...": environment knobs on top of a library variant
        for kv in sets:
            k, v = kv.split("=", 1)
            env[k] = v
        if lib != "default":
            env["MGN_LIB_PATH"] = os.path.join(ROOT, "meshgraphnets.jl_amd", "lib", "variants", lib + ".so")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--nx", str(nx),
                              "--no-cpu-baseline", "--no-secondary", "--dtype", dtype], env=env, capture_output=True, text=True)
        err, out = out.stderr, out.stdout
        line = [l for l in out.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out[-300:], err[-600:]); continue
        d = json.loads(line[-1]); rf = d["roofline"]
        res[n].append((rf["avg_launch_ms"], (rf.get("node_side") or rf.get("node_kernel"))["avg_launch_ms"], d["ms_per_processor_step"]))
for n in names:
    if res[n]:
        e = sorted(x[0] for x in res[n]); nd = sorted(x[1] for x in res[n]); st = sorted(x[2] for x in res[n])
        print(f"{n:28s} edge ms min {e[0]:.3f} med {e[len(e)//2]:.3f} | node ms min {nd[0]:.3f} | step ms min {st[0]:.3f} med {st[len(st)//2]:.3f}")
