"""Error against the float64 oracle of the fp32-MFMA edge kernel and of the split path (csrc/split.hip; MGN_FP32_SPLIT 1 / 2) on the same
inputs: max |x - ref| / max |ref| over node and edge latents after 1 and 15 processor steps (22 500-node mesh, L = 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch  # noqa: F401
import mgn_amd
import mgn_oracle as orc
from util import cfg_dict, engine_for, make_params, rel_max, set_fp32_split

cfg = cfg_dict(mps=15)
pos, s, r = mgn_amd.synth.mesh_1m(1234, 150, 150)
N, E = pos.shape[0], s.size
for seed in (5, 6):
    ps = make_params(cfg, jitter=0.05, seed=seed) if "seed" in make_params.__code__.co_varnames else make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    for nsteps in (1, 15):
        rv, re = orc.processor_steps(ps, cfg, v, e, s, r, nsteps)
        row = []
        for split in (0, 1):
            old = set_fp32_split(split)
            eng = engine_for(cfg); eng.set_params(ps); eng.set_graph(s, r, N)
            v1, e1 = eng.processor_steps(v, e, nsteps)
            set_fp32_split(old)
            row.append((rel_max(v1, rv), rel_max(e1, re)))
        print("seed %d, %2d steps:  fp32 MFMA kernels  node %.2e edge %.2e   |   bf16-split edge kernel  node %.2e edge %.2e" % (
            seed, nsteps, row[0][0], row[0][1], row[1][0], row[1][1]))
