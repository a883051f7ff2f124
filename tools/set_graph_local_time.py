"""What the two ingests of a partitioned handle cost on M-1M at 8 ranks (host side, once per trajectory): mgn_set_graph on the global lists
against mgn_partition_nodes + mgn_set_graph_local on the rank's own edges (the numpy filter that stands in for "the rank only holds its
part" is timed separately).  python tools/set_graph_local_time.py [nranks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd
from mgn_amd import MGN_DEVICE_NONE, Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pos, s, r = mgn_amd.synth.mesh_1m(1234, 1000, 1000)
N = pos.shape[0]
Engine.partition_nodes(8, 2)                     # (loads the library)
t = time.perf_counter(); owner = Engine.partition_nodes(N, P, mesh_pos=pos); t_part = time.perf_counter() - t
rk = P // 2
a = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
t = time.perf_counter(); a.set_graph(s, r, N, mesh_pos=pos); t_glob = time.perf_counter() - t
t = time.perf_counter()
touch = np.nonzero((owner[s] == rk) | (owner[r] == rk))[0].astype(np.int64)
st, rt = np.ascontiguousarray(s[touch]), np.ascontiguousarray(r[touch])
t_filter = time.perf_counter() - t
b = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
import ctypes as C
from mgn_amd._capi import i32, i64
t = time.perf_counter()
b._chk(b.lib.mgn_set_graph_local(b.h, N, i32(owner), s.size, touch.size, i32(st), i32(rt), i64(touch), 0))
t_loc = time.perf_counter() - t
print(f"rank {rk} of {P}: mgn_set_graph (global lists, {s.size} edges) {t_glob*1e3:.1f} ms | mgn_partition_nodes {t_part*1e3:.1f} ms + "
      f"mgn_set_graph_local ({touch.size} edges) {t_loc*1e3:.1f} ms (numpy filter of the global list, not part of either: {t_filter*1e3:.1f} ms)")
