"""mgn_set_graph on a cylinder-sized mesh, and what the first calls behind it cost (a dataset loop installs a new mesh per trajectory):
python tools/set_graph_time.py"""
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np, mgn_amd, bench
ps = bench.glorot_params()
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps)
rng = np.random.default_rng(0)
rows = []
for it in range(6):
    pos, cells, node_type, _ = mgn_amd.synth.mesh_cyl(10 + it, 1900 + 20 * it)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
    tgt = rng.standard_normal((N, 2)).astype(np.float32); mask = np.arange(0, N, 2, dtype=np.int32)
    t = time.perf_counter(); eng.set_graph(s, r, N); t_sg = time.perf_counter() - t
    t = time.perf_counter(); eng.step(nf, ef, tgt, mask); t_s1 = time.perf_counter() - t
    t = time.perf_counter(); eng.step(nf, ef, tgt, mask); t_s2 = time.perf_counter() - t
    t = time.perf_counter(); eng.step(nf, ef, tgt, mask); t_s3 = time.perf_counter() - t
    t = time.perf_counter(); eng.step(nf, ef, tgt, mask); t_s4 = time.perf_counter() - t
    t = time.perf_counter(); eng.forward(nf, ef); t_f1 = time.perf_counter() - t
    t = time.perf_counter(); eng.forward(nf, ef); t_f2 = time.perf_counter() - t
    rows.append((t_sg, t_s1, t_s2, t_s3, t_s4, t_f1, t_f2))
for rw in rows[2:]:
    print("set_graph %.2f ms | step! 1st %.2f, 2nd %.2f, 3rd %.2f, 4th %.2f | forward 1st %.2f, 2nd %.2f" % tuple(1e3 * x for x in rw))
