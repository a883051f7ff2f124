import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, mgn_amd
import mgn_oracle as orc
from util import *
cfg = cfg_dict(mps=2)
pos, cells, node_type, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
ps = make_params(cfg, jitter=0.05)
rng = np.random.default_rng(1)
v = rng.standard_normal((N, 128)).astype(np.float32); e = rng.standard_normal((E, 128)).astype(np.float32)
rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 2)
for f16 in (0, 1):
    set_split_f16(f16)
    for bits in (0, 1, 2, 3, 7):
        set_c16_split(bits)
        eng = engine_for(cfg); eng.set_params(ps); eng.set_graph(s, r, N)
        v1, e1 = eng.processor_steps(v, e, 2)
        print("f16", f16, "c16_split bits", bits, "err v %.2e e %.2e" % (rel_max(v1, rv), rel_max(e1, re)), flush=True)
        eng.close()
