"""The cylinder-sized mesh (BASELINE.json configs[1]: N = 2000 Delaunay, E = 11 954, L = 128, 15 steps, fp32) in a loop, as a
profiling target:   rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_mesh_loop.py [passes]
With --gaps <kernel_trace.csv> it reads a trace of itself instead and prints kernel durations and the idle time between
consecutive kernels of the replayed hipGraph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gaps(path):
    import csv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"].split("(")[0][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    ks = ks[len(ks) // 2:]                       # steady state: the replayed graph
    dur, gap = {}, {}
    for i, (n, s, e) in enumerate(ks):
        dur.setdefault(n, []).append(e - s)
        if i:
            gap.setdefault(ks[i - 1][0] + " -> " + n, []).append(s - ks[i - 1][2])
    med = lambda v: sorted(v)[len(v) // 2]
    for n, v in dur.items():
        print("kernel %-42s n=%5d median %.2f us" % (n, len(v), med(v) / 1e3))
    for n, v in gap.items():
        print("gap    %-84s n=%5d median %.2f us" % (n, len(v), med(v) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--gaps":
        gaps(sys.argv[2])
        sys.exit(0)
    import torch  # noqa: F401
    import mgn_amd
    import bench
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
    eng.set_params(bench.glorot_params())
    eng.set_graph(s, r, pos.shape[0])
    eng.latents_randn(1)
    for _ in range(passes):
        eng.processor_steps_dev(15)
    eng.synchronize()
