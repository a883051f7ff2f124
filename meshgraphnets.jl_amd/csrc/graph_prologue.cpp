// Graph prologue at scale: host replacements for the reference's per-edge loops in create_base_graph
// (reference src/graph.jl:25-55), which cannot build a 6 M-edge graph, plus the world-edge search of cloth-like meshes.
// Pure host code: no handle, no GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <new>
#include <vector>

#include "../../include/mgn_hip.h"

// no C++ exception crosses the C ABI: allocation failures become MGN_E_OOM
#define MGN_NOEXCEPT_BEGIN try {
#define MGN_NOEXCEPT_END                              \
    }                                                 \
    catch (const std::bad_alloc&) { return MGN_E_OOM; } \
    catch (...) { return MGN_E_ARG; }

extern "C" {

int mgn_triangles_to_edges(const int32_t* cells, int64_t n_cells, int32_t* senders, int32_t* receivers, int64_t* n_directed) {
    if (!cells || n_cells < 0 || !n_directed) return MGN_E_ARG;
    MGN_NOEXCEPT_BEGIN
    // packed (max, min) key with the first-occurrence position, sort, unique, restore first-occurrence order
    struct K { uint64_t key; int64_t first; };
    std::vector<K> v((size_t)3 * n_cells);
    for (int64_t cidx = 0; cidx < n_cells; ++cidx) {
        const int32_t* t = cells + 3 * cidx;
        const int32_t pa[3] = {t[0], t[1], t[2]}, pb[3] = {t[1], t[2], t[0]};
        for (int e = 0; e < 3; ++e) {     // reference order: all (0,1) edges, then (1,2), then (2,0)
            const uint32_t hi = (uint32_t)(pa[e] > pb[e] ? pa[e] : pb[e]), lo = (uint32_t)(pa[e] > pb[e] ? pb[e] : pa[e]);
            v[(size_t)e * n_cells + cidx] = {((uint64_t)hi << 32) | lo, (int64_t)e * n_cells + cidx};
        }
    }
    std::sort(v.begin(), v.end(), [](const K& x, const K& y) { return x.key < y.key || (x.key == y.key && x.first < y.first); });
    size_t m = 0;
    for (size_t i = 0; i < v.size(); ++i)
        if (i == 0 || v[i].key != v[i - 1].key) v[m++] = v[i];
    v.resize(m);
    *n_directed = (int64_t)(2 * m);
    if (!senders || !receivers) return MGN_OK;
    std::sort(v.begin(), v.end(), [](const K& x, const K& y) { return x.first < y.first; });
    for (size_t i = 0; i < m; ++i) {
        const int32_t hi = (int32_t)(v[i].key >> 32), lo = (int32_t)(v[i].key & 0xFFFFFFFFu);
        senders[i] = hi; receivers[i] = lo;
        senders[m + i] = lo; receivers[m + i] = hi;
    }
    return MGN_OK;
    MGN_NOEXCEPT_END
}

// World edges of a cloth-like mesh: all ordered pairs (s, r), s != r, with |world_pos[s] - world_pos[r]| < radius that are
// not already joined by a mesh edge (DeepMind flag / cloth models; MGN-spec second edge set).  Uniform grid of cell size
// `radius`: each node looks at the 3^dim neighbouring cells.  Output order: receiver-major, senders ascending.
int mgn_world_edges(const float* world_pos, int32_t dim, int32_t N, float radius, const int32_t* mesh_senders, const int32_t* mesh_receivers,
                    int64_t n_mesh, int32_t index_base, int32_t* senders, int32_t* receivers, int64_t* n_edges) {
    if (!world_pos || dim < 1 || dim > 3 || N < 0 || !(radius > 0.f) || !n_edges || n_mesh < 0 || (n_mesh > 0 && (!mesh_senders || !mesh_receivers)))
        return MGN_E_ARG;
    MGN_NOEXCEPT_BEGIN
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) { lo[d] = 1e30f; hi[d] = -1e30f; }
    for (int64_t i = 0; i < N; ++i)
        for (int d = 0; d < dim; ++d) {
            const float v = world_pos[i * dim + d];
            if (!std::isfinite(v)) return MGN_E_ARG;       // a NaN / Inf position has no cell
            lo[d] = std::min(lo[d], v);
            hi[d] = std::max(hi[d], v);
        }
    // cell size >= radius (the 3^dim neighbourhood then covers the search radius), enlarged until the grid has O(N) cells: a
    // small radius in a large box must not allocate (extent / radius)^dim cells
    int64_t nc[3] = {1, 1, 1};
    double cell = radius;
    const double cap = 4.0 * (double)(N > 16 ? N : 16);
    for (int it = 0; it < 64; ++it) {
        double prod = 1.0;
        for (int d = 0; d < dim; ++d) {
            nc[d] = N > 0 ? std::max<int64_t>((int64_t)(((double)hi[d] - (double)lo[d]) / cell) + 1, 1) : 1;
            prod *= (double)nc[d];
        }
        if (prod <= cap) break;
        cell *= std::max(1.05, std::pow(prod / cap, 1.0 / dim));
    }
    auto cell_of = [&](int64_t i, int64_t cc[3]) {
        for (int d = 0; d < 3; ++d) cc[d] = 0;
        for (int d = 0; d < dim; ++d)
            cc[d] = std::min<int64_t>(std::max<int64_t>((int64_t)(((double)world_pos[i * dim + d] - (double)lo[d]) / cell), 0), nc[d] - 1);
    };
    const int64_t ncell = nc[0] * nc[1] * nc[2];
    std::vector<int32_t> start((size_t)ncell + 1, 0), order((size_t)N);
    for (int64_t i = 0; i < N; ++i) { int64_t cc[3]; cell_of(i, cc); ++start[(size_t)((cc[2] * nc[1] + cc[1]) * nc[0] + cc[0]) + 1]; }
    for (int64_t q = 0; q < ncell; ++q) start[q + 1] += start[q];
    {
        std::vector<int32_t> cur(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < N; ++i) { int64_t cc[3]; cell_of(i, cc); order[cur[(size_t)((cc[2] * nc[1] + cc[1]) * nc[0] + cc[0])]++] = (int32_t)i; }
    }
    // mesh neighbours by receiver (CSR) to exclude pairs that already share a mesh edge
    std::vector<int32_t> mrp((size_t)N + 1, 0), msnd((size_t)n_mesh);
    for (int64_t e = 0; e < n_mesh; ++e) {
        const int64_t s_ = (int64_t)mesh_senders[e] - index_base, r_ = (int64_t)mesh_receivers[e] - index_base;
        if (s_ < 0 || s_ >= N || r_ < 0 || r_ >= N) return MGN_E_ARG;
        ++mrp[(size_t)r_ + 1];
    }
    for (int64_t i = 0; i < N; ++i) mrp[i + 1] += mrp[i];
    {
        std::vector<int32_t> cur(mrp.begin(), mrp.end() - 1);
        for (int64_t e = 0; e < n_mesh; ++e) msnd[cur[mesh_receivers[e] - index_base]++] = mesh_senders[e] - index_base;
    }
    // distances in double: differences and products of floats are exact there, so the decision d2 < r2 does not depend on how a
    // compiler contracts the sum (the device version, csrc/graph_dev.hip, takes the same decisions bit for bit)
    const double r2 = (double)radius * (double)radius;
    int64_t count = 0;
    std::vector<int32_t> cand;
    for (int pass = 0; pass < ((senders && receivers) ? 2 : 1); ++pass) {
        count = 0;
        for (int64_t r_ = 0; r_ < N; ++r_) {
            int64_t cc[3];
            cell_of(r_, cc);
            cand.clear();
            for (int64_t dz = (dim > 2 ? -1 : 0); dz <= (dim > 2 ? 1 : 0); ++dz)
                for (int64_t dy = (dim > 1 ? -1 : 0); dy <= (dim > 1 ? 1 : 0); ++dy)
                    for (int64_t dx = -1; dx <= 1; ++dx) {
                        const int64_t x = cc[0] + dx, y = cc[1] + dy, z = cc[2] + dz;
                        if (x < 0 || x >= nc[0] || y < 0 || y >= nc[1] || z < 0 || z >= nc[2]) continue;
                        const int64_t q = (z * nc[1] + y) * nc[0] + x;
                        for (int32_t p = start[q]; p < start[q + 1]; ++p) {
                            const int32_t s_ = order[p];
                            if (s_ == r_) continue;
                            double d2 = 0.0;
                            for (int d = 0; d < dim; ++d) {
                                const double dd = (double)world_pos[(int64_t)s_ * dim + d] - (double)world_pos[r_ * dim + d];
                                d2 += dd * dd;
                            }
                            if (d2 < r2) cand.push_back(s_);
                        }
                    }
            std::sort(cand.begin(), cand.end());
            for (int32_t s_ : cand) {
                bool mesh = false;
                for (int32_t p = mrp[r_]; p < mrp[r_ + 1] && !mesh; ++p) mesh = msnd[p] == s_;
                if (mesh) continue;
                if (pass == 1) { senders[count] = s_ + index_base; receivers[count] = (int32_t)r_ + index_base; }
                ++count;
            }
        }
        if (pass == 0) {
            if (senders && receivers && count > *n_edges) { *n_edges = count; return MGN_E_ARG; }   // caller's buffers are too small
        }
    }
    *n_edges = count;
    return MGN_OK;
    MGN_NOEXCEPT_END
}

int mgn_edge_features(const float* pos, int32_t dim, const int32_t* senders, const int32_t* receivers, int64_t E,
                      int32_t index_base, float* ef) {
    if (!pos || dim < 1 || dim > 8 || (E > 0 && (!senders || !receivers || !ef)) || E < 0) return MGN_E_ARG;
    for (int64_t i = 0; i < E; ++i) {
        const float* ps = pos + (size_t)(senders[i] - index_base) * dim;
        const float* pr = pos + (size_t)(receivers[i] - index_base) * dim;
        float* o = ef + (size_t)i * (dim + 1);
        float n2 = 0.f;
        for (int d = 0; d < dim; ++d) {
            o[d] = ps[d] - pr[d];
            n2 += o[d] * o[d];
        }
        o[dim] = std::sqrt(n2);
    }
    return MGN_OK;
}

// ---- diagnostics (not part of the public header; meaningful only with -DMGN_DIAG_STAMPS) -----------------
}  // extern "C"
