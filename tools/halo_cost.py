"""What the pack and the unpack launch of a halo exchange cost on the per-GPU share of M-1M at 8 GPUs (125 k owned nodes, strips: ~1 000
halo rows): rank 3 of 8 partitions of the 1000 x 1000 mesh, one process, no communicator -- mgn_halo_pack / mgn_halo_unpack on device
buffers, timed over 200 calls each (launch overhead included: in the staged schedule they are launches on the compute stream).
    python tools/halo_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, mgn_amd, bench
pos, s, r = mgn_amd.synth.mesh_1m(1234, 1000, 1000)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, rank=3, nranks=8)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0], mesh_pos=pos)
eng.latents_randn(1)
nsend = int(eng.halo_send_index().size); nhalo = int(eng.n_halo); rowf = eng.halo_row_floats
send = torch.zeros(max(nsend, 1) * rowf, device="cuda"); recv = torch.zeros(max(nhalo, 1) * rowf, device="cuda")
for name, fn, ptr in (("pack", eng.halo_pack, send.data_ptr()), ("unpack", eng.halo_unpack, recv.data_ptr())):
    for _ in range(20): fn(ptr)
    eng.synchronize(); t = time.perf_counter()
    for _ in range(200): fn(ptr)
    eng.synchronize(); dt = (time.perf_counter() - t) / 200
    print("%s: %.1f us per call (n_own %d, send rows %d, halo rows %d, %d floats per row)" % (name, dt * 1e6, eng.n_own, nsend, nhalo, rowf))
eng.processor_steps_dev  # (the step itself: bench.py mid_size_meshes, N = 125 316: ~0.50 ms)
