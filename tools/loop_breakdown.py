"""Where an iteration of the reference's training loop (new parameters before every step!, src/MeshGraphNets.jl:375-377) spends its time:
the caller's update of the vector, mgn_set_params, mgn_step (which re-packs the training layouts on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch, mgn_amd
import mgn_oracle as orc

pos, cells, node_type, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
ps = orc.init_params(9, 3, 2, 128, 2, 15, 1234, 0.05)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
tgt = rng.standard_normal((N, 2)).astype(np.float32)
mask = np.nonzero(np.isin(node_type, [0, 5]))[0].astype(np.int32)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps); eng.set_graph(s, r, N)
d = lambda a: torch.from_numpy(a).cuda()
nf_d, ef_d, tgt_d, gs_d = d(nf), d(ef), d(tgt), torch.zeros(eng.param_count, device="cuda")
for _ in range(4):
    eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d)
K = 30
t_up, t_set, t_step = [], [], []
p = ps.copy()
for _ in range(K):
    t0 = time.perf_counter(); p *= np.float32(1.00001); t1 = time.perf_counter()
    eng.set_params(p); t2 = time.perf_counter()
    eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d); t3 = time.perf_counter()
    t_up.append(t1 - t0); t_set.append(t2 - t1); t_step.append(t3 - t2)
med = lambda x: float(np.median(x)) * 1e3
print("per iteration: caller's update %.2f ms, mgn_set_params %.2f ms, mgn_step (device arrays; re-packs) %.2f ms" % (med(t_up), med(t_set), med(t_step)))
t = []
for _ in range(K):
    t0 = time.perf_counter(); eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d); t.append(time.perf_counter() - t0)
print("mgn_step alone (parameters unchanged): %.2f ms" % med(t))
t_set, t_step = [], []
for _ in range(K):
    p *= np.float32(1.00001)
    t1 = time.perf_counter(); eng.set_params(p); t2 = time.perf_counter()
    eng.step(nf, ef, tgt, mask); t3 = time.perf_counter()
    t_set.append(t2 - t1); t_step.append(t3 - t2)
print("host arrays, fresh gradient vector per call: mgn_set_params %.2f ms, mgn_step %.2f ms" % (med(t_set), med(t_step)))
buf = np.zeros(eng.param_count, np.float32)
t_step = []
for _ in range(K):
    p *= np.float32(1.00001)
    eng.set_params(p); t2 = time.perf_counter()
    eng.step(nf, ef, tgt, mask, out=buf); t3 = time.perf_counter()
    t_step.append(t3 - t2)
print("host arrays, caller-owned gradient vector: mgn_step %.2f ms" % med(t_step))
if hasattr(eng, "set_params_dev"):
    pd = d(ps)
    t_set, t_step = [], []
    for _ in range(K):
        pd.mul_(1.00001); torch.cuda.synchronize()
        t1 = time.perf_counter(); eng.set_params_dev(pd); t2 = time.perf_counter()
        eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d); t3 = time.perf_counter()
        t_set.append(t2 - t1); t_step.append(t3 - t2)
    print("device-resident parameters: mgn_set_params %.2f ms, mgn_step %.2f ms" % (med(t_set), med(t_step)))
