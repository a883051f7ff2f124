#!/bin/bash
# same-box A/B of library variants on the cylinder-sized mesh (N = 2000 Delaunay, fp32):  bash tools/ab_small.sh <variant.so|main> ...
for rep in 1; do
for v in "$@"; do
  if [ "$v" = main ]; then unset MGN_LIB_PATH; else export MGN_LIB_PATH=$v; fi
  python - <<PY
import sys, time
sys.path.insert(0, "$PWD")
import torch, mgn_amd, bench
pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15); eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
for _ in range(20): eng.processor_steps_dev(15)
eng.synchronize(); best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(50): eng.processor_steps_dev(15)
    eng.synchronize(); best = min(best, (time.perf_counter() - t0) / 750)
print("$v", "M-cyl %.2f us per processor step" % (best * 1e6))
PY
done; done
