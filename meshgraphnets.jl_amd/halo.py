"""Halo-row exchange between mesh partitions: after every node update each partition
sends the freshly projected P rows (v * W1_sender of the next step) of its boundary nodes to the
partitions that list them as halo senders -- a sparse all-to-all-v.  One process per GPU; the wire is
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

There is no precedent in the reference (single device, src/MeshGraphNets.jl:255-263).
"""
from __future__ import annotations

import numpy as np
import torch


class DistExchange:
    """One partition per process.  `engine` needs halo_counts(), halo_pack(ptr), halo_unpack(ptr),
    halo_row_floats.  Buffers live on `device` (cuda for RCCL, cpu for gloo)."""

    def __init__(self, engine, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = group
        s, r = engine.halo_counts()
        self.send_rows = [int(x) for x in s]
        self.recv_rows = [int(x) for x in r]
        L = engine.halo_row_floats
        self.send = torch.empty((max(1, sum(self.send_rows)), L), dtype=torch.float32, device=device)
        self.recv = torch.empty((max(1, sum(self.recv_rows)), L), dtype=torch.float32, device=device)
        self.n_send, self.n_recv = sum(self.send_rows), sum(self.recv_rows)
        self._work = None

    def start(self):
        """pack + launch the all-to-all-v asynchronously (RCCL runs it on its own stream, ordered after the pack
        kernel on the current stream)."""
        tensor_api = hasattr(self.engine, "halo_pack_tensor")   # NumPy stand-in engines of the CPU tests
        if tensor_api:
            self.engine.halo_pack_tensor(self.send)
        else:
            self.engine.halo_pack(self.send.data_ptr())
        self._work = self.dist.all_to_all_single(self.recv[: self.n_recv], self.send[: self.n_send],
                                                 output_split_sizes=self.recv_rows, input_split_sizes=self.send_rows,
                                                 group=self.group, async_op=True)

    def finish(self):
        """make the current stream wait for the collective, then copy the rows into the halo block of P."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if hasattr(self.engine, "halo_pack_tensor"):
            self.engine.halo_unpack_tensor(self.recv)
        else:
            self.engine.halo_unpack(self.recv.data_ptr())

    def __call__(self):
        self.start()
        self.finish()


class LoopbackExchange:
    """All partitions in ONE process (one GPU, or numpy stand-ins): the all-to-all-v is done by
    device copies.  Used to validate the partitioned path on a single MI355X (KAT-7)."""

    def __init__(self, engines, device):
        self.engines = engines
        P = len(engines)
        L = engines[0].halo_row_floats
        self.counts = [e.halo_counts() for e in engines]
        self.send = [torch.empty((max(1, int(c[0].sum())), L), dtype=torch.float32, device=device) for c in self.counts]
        self.recv = [torch.empty((max(1, int(c[1].sum())), L), dtype=torch.float32, device=device) for c in self.counts]
        self.soff = [np.concatenate([[0], np.cumsum(c[0])]).astype(np.int64) for c in self.counts]
        self.roff = [np.concatenate([[0], np.cumsum(c[1])]).astype(np.int64) for c in self.counts]
        for p in range(P):
            for q in range(P):
                assert self.counts[p][0][q] == self.counts[q][1][p], "send/recv halo counts disagree"

    def start(self):
        tensor_api = hasattr(self.engines[0], "halo_pack_tensor")
        for p, e in enumerate(self.engines):
            e.halo_pack_tensor(self.send[p]) if tensor_api else e.halo_pack(self.send[p].data_ptr())

    def __call__(self):
        self.start()
        self.finish()

    def finish(self):
        P = len(self.engines)
        tensor_api = hasattr(self.engines[0], "halo_pack_tensor")
        for p in range(P):          # receiver
            for q in range(P):      # sender
                n = int(self.counts[p][1][q])
                if n:
                    self.recv[p][self.roff[p][q]: self.roff[p][q] + n].copy_(
                        self.send[q][self.soff[q][p]: self.soff[q][p] + n])
        for p, e in enumerate(self.engines):
            e.halo_unpack_tensor(self.recv[p]) if tensor_api else e.halo_unpack(self.recv[p].data_ptr())


class HostStagedExchange:
    """The same all-to-all-v over a CPU backend (gloo): rows are packed on the device, staged through pinned host buffers
    and exchanged by the process group.  Not the production wire (that is DistExchange over RCCL / xGMI) -- it exists so that
    the REAL engine can be driven by several processes on a box with a single GPU (tests), and as a fallback transport."""

    def __init__(self, engine, device, group=None):
        import torch.distributed as dist
        self.dist, self.engine, self.group = dist, engine, group
        s, r = engine.halo_counts()
        self.send_rows, self.recv_rows = [int(x) for x in s], [int(x) for x in r]
        self.n_send, self.n_recv = sum(self.send_rows), sum(self.recv_rows)
        L = engine.halo_row_floats
        self.send_dev = torch.empty((max(1, self.n_send), L), dtype=torch.float32, device=device)
        self.recv_dev = torch.empty((max(1, self.n_recv), L), dtype=torch.float32, device=device)
        self.send_host = torch.empty((max(1, self.n_send), L), dtype=torch.float32).pin_memory()
        self.recv_host = torch.empty((max(1, self.n_recv), L), dtype=torch.float32).pin_memory()
        self._work = None

    def start(self):
        self.engine.halo_pack(self.send_dev.data_ptr())
        self.send_host.copy_(self.send_dev)                    # synchronous: the rows are on the host when it returns
        self._work = self.dist.all_to_all_single(self.recv_host[: self.n_recv], self.send_host[: self.n_send],
                                                 output_split_sizes=self.recv_rows, input_split_sizes=self.send_rows,
                                                 group=self.group, async_op=True)

    def finish(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        self.recv_dev.copy_(self.recv_host)                    # synchronous: recv_host is free for the next exchange
        self.engine.halo_unpack(self.recv_dev.data_ptr())

    def __call__(self):
        self.start()
        self.finish()
