// Device-side graph prologue.  See graph_dev.h.  Sorting / compaction / scans are rocPRIM device primitives
// (stable LSD radix sort, unique_by_key, exclusive_scan); key construction, emission, one-hot, edge features and the uniform-grid
// radius search are kernels here.  Everything is HBM-bound integer / byte work: coalesced streams, no MFMA.
#include "graph_dev.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

namespace mgn {
namespace {

constexpr int TPB = 256;
inline unsigned nblocks(int64_t n) { return (unsigned)((n + TPB - 1) / TPB); }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- triangles_to_edges -------------------------------------------------------------------------------------------------
// key = (max << 32) | min of edge e of cell c; position = e * C + c: the reference's visiting order (all (0,1) edges, then (1,2),
// then (2,0); GraphNetCore.triangles_to_edges as restated in oracle/mgn_oracle.py and csrc/graph_prologue.cpp)
__global__ void k_cell_keys(const int32_t* __restrict__ cells, int64_t C, uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * C) return;
    const int64_t e = i / C, c = i - e * C;
    const int32_t a = cells[3 * c + e], b = cells[3 * c + (e + 1) % 3];
    const uint32_t hi = (uint32_t)(a > b ? a : b), lo = (uint32_t)(a > b ? b : a);
    keys[i] = ((uint64_t)hi << 32) | lo;
    pos[i] = (uint32_t)i;
}
__global__ void k_emit_two_way(const uint64_t* __restrict__ keys, int64_t m, int32_t* __restrict__ senders, int32_t* __restrict__ receivers) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int32_t hi = (int32_t)(keys[i] >> 32), lo = (int32_t)(keys[i] & 0xFFFFFFFFu);
    senders[i] = hi;
    receivers[i] = lo;
    senders[m + i] = lo;
    receivers[m + i] = hi;
}

// ---- features -----------------------------------------------------------------------------------------------------------
__global__ void k_one_hot(const int32_t* __restrict__ node_type, const int32_t* __restrict__ gid, int32_t n, int32_t type_min, int32_t depth,
                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * depth) return;
    const int32_t row = (int32_t)(i / depth), d = (int32_t)(i - (int64_t)row * depth);
    out[i] = (node_type[gid ? gid[row] : row] - type_min == d) ? 1.0f : 0.0f;
}
__global__ void k_edge_features(const float* __restrict__ pos, int dim, const int32_t* __restrict__ snd, const int32_t* __restrict__ rcv,
                                const int32_t* __restrict__ loc2glob, int64_t E, float* __restrict__ ef) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    const int64_t gs = loc2glob ? loc2glob[snd[j]] : snd[j], gr = loc2glob ? loc2glob[rcv[j]] : rcv[j];
    float n2 = 0.f;
    float* o = ef + j * (dim + 1);
    for (int d = 0; d < dim; ++d) {
        const float v = pos[gs * dim + d] - pos[gr * dim + d];
        o[d] = v;
        n2 += v * v;
    }
    o[dim] = sqrtf(n2);
}
__global__ void k_iota64(int64_t* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i;
}

// ---- world edges: uniform grid ------------------------------------------------------------------------------------------
struct Grid {
    double lo[3], cell;
    int32_t nc[3];
    int dim;
};
__device__ __forceinline__ void cell_of(const Grid& g, const float* __restrict__ p, int32_t cc[3]) {
    cc[0] = cc[1] = cc[2] = 0;
    for (int d = 0; d < g.dim; ++d) {
        int64_t v = (int64_t)(((double)p[d] - g.lo[d]) / g.cell);
        v = v < 0 ? 0 : (v > g.nc[d] - 1 ? g.nc[d] - 1 : v);
        cc[d] = (int32_t)v;
    }
}
// ordered-int encoding of a float for atomicMin / atomicMax
__device__ __forceinline__ int fenc(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__global__ void k_bbox(const float* __restrict__ pos, int dim, int64_t N, int* __restrict__ mn, int* __restrict__ mx, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * dim) return;
    const float v = pos[i];
    if (!isfinite(v)) { *bad = 1; return; }
    const int d = (int)(i % dim);
    atomicMin(mn + d, fenc(v));
    atomicMax(mx + d, fenc(v));
}
__global__ void k_cell_ids(const float* __restrict__ pos, Grid g, int32_t N, uint32_t* __restrict__ cid, uint32_t* __restrict__ nid) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int32_t cc[3];
    cell_of(g, pos + (int64_t)i * g.dim, cc);
    cid[i] = (uint32_t)((cc[2] * g.nc[1] + cc[1]) * g.nc[0] + cc[0]);
    nid[i] = (uint32_t)i;
}
// start[c] = first position in the cell-sorted node list whose cell id is >= c   (c = 0 .. ncell)
__global__ void k_cell_start(const uint32_t* __restrict__ cid_sorted, int32_t N, int32_t ncell, int32_t* __restrict__ start) {
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell) return;
    int32_t lo = 0, hi = N;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (cid_sorted[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
    }
    start[c] = lo;
}
// FILL = false: cnt[r] = number of world edges ending at r.  FILL = true: senders of r written at rowptr[r].., ascending
// (the cells are visited in the host version's order, the candidates of a receiver then sorted: here by insertion, lists are short).
// Distances in double (exact differences and products of floats): no dependence on fma contraction -> the host's decisions.
template <bool FILL>
__global__ void k_world_search(const float* __restrict__ pos, Grid g, int32_t N, double r2, const int32_t* __restrict__ start,
                               const uint32_t* __restrict__ order, const int32_t* __restrict__ mesh_rowptr, const int32_t* __restrict__ mesh_snd,
                               int32_t* __restrict__ cnt, const int32_t* __restrict__ rowptr, int32_t* __restrict__ snd, int32_t* __restrict__ rcv) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    int32_t cc[3];
    const float* pr = pos + (int64_t)r * g.dim;
    cell_of(g, pr, cc);
    int32_t n = 0;
    const int32_t base = FILL ? rowptr[r] : 0;
    for (int dz = (g.dim > 2 ? -1 : 0); dz <= (g.dim > 2 ? 1 : 0); ++dz)
        for (int dy = (g.dim > 1 ? -1 : 0); dy <= (g.dim > 1 ? 1 : 0); ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = cc[0] + dx, y = cc[1] + dy, z = cc[2] + dz;
                if (x < 0 || x >= g.nc[0] || y < 0 || y >= g.nc[1] || z < 0 || z >= g.nc[2]) continue;
                const int32_t q = (z * g.nc[1] + y) * g.nc[0] + x;
                for (int32_t p = start[q]; p < start[q + 1]; ++p) {
                    const int32_t s_ = (int32_t)order[p];
                    if (s_ == r) continue;
                    double d2 = 0.0;
                    for (int d = 0; d < g.dim; ++d) {
                        const double dd = (double)pos[(int64_t)s_ * g.dim + d] - (double)pr[d];
                        d2 += dd * dd;
                    }
                    if (!(d2 < r2)) continue;
                    bool mesh = false;
                    for (int32_t m = mesh_rowptr[r]; m < mesh_rowptr[r + 1] && !mesh; ++m) mesh = mesh_snd[m] == s_;
                    if (mesh) continue;
                    if (FILL) {
                        int32_t k = n;                     // insertion into the ascending list
                        while (k > 0 && snd[base + k - 1] > s_) { snd[base + k] = snd[base + k - 1]; --k; }
                        snd[base + k] = s_;
                        rcv[base + n] = r;
                    }
                    ++n;
                }
            }
    if (!FILL) cnt[r] = n;
}

}  // namespace

hipError_t launch_iota64(int64_t* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_iota64, dim3(nblocks(n)), dim3(TPB), 0, s, dst, n);
    return hipGetLastError();
}

hipError_t launch_one_hot(const int32_t* node_type, const int32_t* gid, int32_t n, int32_t type_min, int32_t depth, float* out, hipStream_t s) {
    if (n <= 0 || depth <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_one_hot, dim3(nblocks((int64_t)n * depth)), dim3(TPB), 0, s, node_type, gid, n, type_min, depth, out);
    return hipGetLastError();
}

hipError_t launch_edge_features_local(const float* pos, int dim, const int32_t* snd, const int32_t* rcv, const int32_t* loc2glob, int64_t E,
                                      float* ef, hipStream_t s) {
    if (E <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_edge_features, dim3(nblocks(E)), dim3(TPB), 0, s, pos, dim, snd, rcv, loc2glob, E, ef);
    return hipGetLastError();
}

#define RP(expr)                           \
    do {                                   \
        hipError_t _e = (expr);            \
        if (_e != hipSuccess) return _e;   \
    } while (0)

hipError_t dev_triangles_to_edges(const int32_t* cells, int64_t C, DevBuf& work, int32_t* senders, int32_t* receivers, int64_t capacity,
                                  int64_t* m_out, hipStream_t s) {
    *m_out = 0;
    if (C <= 0) return hipSuccess;
    const int64_t n = 3 * C;
    if (n >= (int64_t)1 << 32) return hipErrorInvalidValue;      // positions are 32-bit
    // temporary storage of the three primitives (the largest is reused)
    size_t t_sort = 0, t_uniq = 0, t_sort2 = 0;
    RP(rocprim::radix_sort_pairs(nullptr, t_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 64, s));
    RP(rocprim::unique_by_key(nullptr, t_uniq, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                              (size_t*)nullptr, (size_t)n, rocprim::equal_to<uint64_t>(), s));
    RP(rocprim::radix_sort_pairs(nullptr, t_sort2, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0, 32, s));
    const size_t t_max = al256(t_sort > t_uniq ? (t_sort > t_sort2 ? t_sort : t_sort2) : (t_uniq > t_sort2 ? t_uniq : t_sort2));
    const size_t kb = al256((size_t)n * 8), vb = al256((size_t)n * 4);
    RP(work.ensure(2 * kb + 2 * vb + 256 + t_max));
    char* base = work.as<char>();
    uint64_t* k0 = reinterpret_cast<uint64_t*>(base);
    uint64_t* k1 = reinterpret_cast<uint64_t*>(base + kb);
    uint32_t* v0 = reinterpret_cast<uint32_t*>(base + 2 * kb);
    uint32_t* v1 = reinterpret_cast<uint32_t*>(base + 2 * kb + vb);
    size_t* d_cnt = reinterpret_cast<size_t*>(base + 2 * kb + 2 * vb);
    void* tmp = base + 2 * kb + 2 * vb + 256;
    hipLaunchKernelGGL(k_cell_keys, dim3(nblocks(n)), dim3(TPB), 0, s, cells, C, k0, v0);
    RP(hipGetLastError());
    size_t t = t_max;
    RP(rocprim::radix_sort_pairs(tmp, t, k0, k1, v0, v1, (size_t)n, 0, 64, s));             // stable: equal keys keep their visiting order
    t = t_max;
    RP(rocprim::unique_by_key(tmp, t, k1, v1, k0, v0, d_cnt, (size_t)n, rocprim::equal_to<uint64_t>(), s));   // first of every run
    size_t m = 0;
    RP(hipMemcpyAsync(&m, d_cnt, sizeof m, hipMemcpyDeviceToHost, s));
    RP(hipStreamSynchronize(s));
    *m_out = (int64_t)m;
    if ((int64_t)(2 * m) > capacity) return hipErrorInvalidValue;
    t = t_max;
    RP(rocprim::radix_sort_pairs(tmp, t, v0, v1, k0, k1, m, 0, 32, s));                      // back to first-occurrence order
    hipLaunchKernelGGL(k_emit_two_way, dim3(nblocks((int64_t)m)), dim3(TPB), 0, s, k1, (int64_t)m, senders, receivers);
    return hipGetLastError();
}

hipError_t dev_world_edges(const float* pos, int dim, int32_t N, float radius, const int32_t* mesh_rowptr, const int32_t* mesh_snd, DevBuf& work,
                           DevBuf& snd, DevBuf& rcv, DevBuf& rowptr, int64_t* E_out, hipStream_t s) {
    *E_out = 0;
    RP(rowptr.ensure((size_t)(N + 1) * 4));
    if (N <= 0) return hipMemsetAsync(rowptr.p, 0, 4, s);
    // bounding box (device reduction; the positions may live on the device)
    size_t t_sort = 0, t_scan = 0;
    RP(rocprim::radix_sort_pairs(nullptr, t_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)N, 0, 32, s));
    RP(rocprim::exclusive_scan(nullptr, t_scan, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(N + 1), rocprim::plus<int32_t>(), s));
    const size_t t_max = al256(t_sort > t_scan ? t_sort : t_scan);
    const size_t nb = al256((size_t)N * 4), cap_cells = (size_t)4 * (N > 16 ? N : 16) + 64;
    RP(work.ensure(256 + 5 * nb + al256((cap_cells + 2) * 4) + t_max + 256));
    char* base = work.as<char>();
    int* d_box = reinterpret_cast<int*>(base);                  // mn[3], mx[3], bad
    uint32_t* cid0 = reinterpret_cast<uint32_t*>(base + 256);
    uint32_t* cid1 = reinterpret_cast<uint32_t*>(base + 256 + nb);
    uint32_t* nid0 = reinterpret_cast<uint32_t*>(base + 256 + 2 * nb);
    uint32_t* nid1 = reinterpret_cast<uint32_t*>(base + 256 + 3 * nb);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + 256 + 4 * nb);              // [N + 1] (the scan reads one past: padded below)
    int32_t* start = reinterpret_cast<int32_t*>(base + 256 + 5 * nb + 256);
    void* tmp = base + 256 + 5 * nb + 256 + al256((cap_cells + 2) * 4);
    (void)tmp;
    const int hbox[7] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0};
    RP(hipMemcpyAsync(d_box, hbox, sizeof hbox, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bbox, dim3(nblocks((int64_t)N * dim)), dim3(TPB), 0, s, pos, dim, (int64_t)N, d_box, d_box + 3, d_box + 6);
    RP(hipGetLastError());
    int box[7];
    RP(hipMemcpyAsync(box, d_box, sizeof box, hipMemcpyDeviceToHost, s));
    RP(hipStreamSynchronize(s));
    if (box[6]) return hipErrorInvalidValue;                                      // NaN / Inf position
    auto fdec = [](int i) { const int j = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; memcpy(&f, &j, 4); return f; };
    Grid g{};
    g.dim = dim;
    g.cell = radius;
    double hi[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) { g.lo[d] = fdec(box[d]); hi[d] = fdec(box[3 + d]); }
    g.nc[0] = g.nc[1] = g.nc[2] = 1;
    // same cell-size rule as the host version: >= radius, enlarged until the grid has O(N) cells
    const double cap = 4.0 * (double)(N > 16 ? N : 16);
    for (int it = 0; it < 64; ++it) {
        double prod = 1.0;
        for (int d = 0; d < dim; ++d) {
            const int64_t c = (int64_t)((hi[d] - g.lo[d]) / g.cell) + 1;
            g.nc[d] = (int32_t)(c < 1 ? 1 : c);
            prod *= (double)g.nc[d];
        }
        if (prod <= cap) break;
        const double f = pow(prod / cap, 1.0 / dim);
        g.cell *= f > 1.05 ? f : 1.05;
    }
    const int32_t ncell = g.nc[0] * g.nc[1] * g.nc[2];
    hipLaunchKernelGGL(k_cell_ids, dim3(nblocks(N)), dim3(TPB), 0, s, pos, g, N, cid0, nid0);
    RP(hipGetLastError());
    size_t t = t_max;
    void* tmp2 = base + 256 + 5 * nb + 256 + al256((cap_cells + 2) * 4);
    RP(rocprim::radix_sort_pairs(tmp2, t, cid0, cid1, nid0, nid1, (size_t)N, 0, 32, s));   // stable: ascending node id inside a cell
    hipLaunchKernelGGL(k_cell_start, dim3(nblocks((int64_t)ncell + 1)), dim3(TPB), 0, s, cid1, N, ncell, start);
    RP(hipGetLastError());
    const double r2 = (double)radius * (double)radius;
    hipLaunchKernelGGL((k_world_search<false>), dim3(nblocks(N)), dim3(TPB), 0, s, pos, g, N, r2, start, nid1, mesh_rowptr, mesh_snd, cnt,
                       (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    RP(hipGetLastError());
    RP(hipMemsetAsync(cnt + N, 0, 4, s));
    t = t_max;
    RP(rocprim::exclusive_scan(tmp2, t, cnt, rowptr.as<int32_t>(), 0, (size_t)(N + 1), rocprim::plus<int32_t>(), s));
    int32_t E = 0;
    RP(hipMemcpyAsync(&E, rowptr.as<int32_t>() + N, 4, hipMemcpyDeviceToHost, s));
    RP(hipStreamSynchronize(s));
    *E_out = E;
    RP(snd.ensure((size_t)(E > 0 ? E : 1) * 4));
    RP(rcv.ensure((size_t)(E > 0 ? E : 1) * 4));
    if (E == 0) return hipSuccess;
    hipLaunchKernelGGL((k_world_search<true>), dim3(nblocks(N)), dim3(TPB), 0, s, pos, g, N, r2, start, nid1, mesh_rowptr, mesh_snd, (int32_t*)nullptr,
                       rowptr.as<int32_t>(), snd.as<int32_t>(), rcv.as<int32_t>());
    return hipGetLastError();
}

}  // namespace mgn
