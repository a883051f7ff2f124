// Device-side graph prologue: what create_base_graph does per trajectory (reference src/graph.jl:25-55) as kernels.
// Integer results are bit-identical to the host versions in graph_prologue.cpp (same keys, stable sorts, same visiting order).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine_internal.h"

namespace mgn {

// cells [C][3] (device) -> unique undirected edges (hi, lo) in first-occurrence order, written two-way:
// senders = [hi; lo], receivers = [lo; hi] (device buffers of `capacity` entries each).  *m = number of undirected edges.
// Returns hipErrorInvalidValue with *m set when 2 * m > capacity.  Synchronises the stream (the count sizes the output).
hipError_t dev_triangles_to_edges(const int32_t* cells, int64_t C, DevBuf& work, int32_t* senders, int32_t* receivers, int64_t capacity,
                                  int64_t* m, hipStream_t s);

// out[i][d] = (node_type[gid ? gid[i] : i] - type_min == d) for i < n   (GraphNetCore one_hot(vec, depth, offset = 1 - type_min))
hipError_t launch_one_hot(const int32_t* node_type, const int32_t* gid, int32_t n, int32_t type_min, int32_t depth, float* out, hipStream_t s);

// ef[j] = [pos[gs] - pos[gr] ; || . ||] for local edge j, gs / gr = global ids of its local sender / receiver
// (loc2glob == null: local ids are global ids); reference src/graph.jl:35-36, 49-52
hipError_t launch_edge_features_local(const float* pos, int dim, const int32_t* snd, const int32_t* rcv, const int32_t* loc2glob, int64_t E,
                                      float* ef, hipStream_t s);

// Radius graph in world space minus self loops and mesh-edge pairs, receiver-major with ascending senders (the order of
// mgn_world_edges), built on the device: uniform grid (cell >= radius, O(N) cells), nodes radix-sorted by cell, count / scan / fill.
// mesh_rowptr / mesh_snd: the mesh set's CSR by receiver (local == global ids: one partition).  Outputs (device, resized here):
// snd, rcv [E], rowptr [N + 1].  *E_out on the host.  Synchronises the stream.
hipError_t dev_world_edges(const float* pos, int dim, int32_t N, float radius, const int32_t* mesh_rowptr, const int32_t* mesh_snd, DevBuf& work,
                           DevBuf& snd, DevBuf& rcv, DevBuf& rowptr, int64_t* E_out, hipStream_t s);

hipError_t launch_iota64(int64_t* dst, int64_t n, hipStream_t s);

}  // namespace mgn
