// Host-side graph preparation (see graph_host.h).  Amortised once per trajectory, exactly where the
// reference builds its static graph: create_base_graph, reference src/graph.jl:25-55, called at
// src/MeshGraphNets.jl:360,418,596.  The reference keeps edges in triangles_to_edges order and
// scatters with atomics (NNlib); this engine re-orders edges by receiver so that the scatter-add
// becomes a segmented sum, and (nranks > 1) cuts the mesh by node ownership.
#include "graph_host.h"

#include <algorithm>
#include <numeric>

namespace mgn {

namespace {

void rcb_rec(std::vector<int32_t>& idx, int64_t lo, int64_t hi, int32_t part0, int32_t parts, const float* pos,
             int32_t dim, std::vector<int32_t>& owner) {
    if (parts <= 1 || hi - lo <= 0) {
        for (int64_t i = lo; i < hi; ++i) owner[idx[i]] = part0;
        return;
    }
    int axis = 0;
    float best = -1.f;
    for (int d = 0; d < dim; ++d) {
        float mn = pos[(int64_t)idx[lo] * dim + d], mx = mn;
        for (int64_t i = lo + 1; i < hi; ++i) {
            const float v = pos[(int64_t)idx[i] * dim + d];
            mn = std::min(mn, v);
            mx = std::max(mx, v);
        }
        if (mx - mn > best) {
            best = mx - mn;
            axis = d;
        }
    }
    const int32_t pl = parts / 2;
    const int64_t nleft = (hi - lo) * pl / parts;
    auto cmp = [&](int32_t a, int32_t b) {
        const float va = pos[(int64_t)a * dim + axis], vb = pos[(int64_t)b * dim + axis];
        return va < vb || (va == vb && a < b);
    };
    std::nth_element(idx.begin() + lo, idx.begin() + lo + nleft, idx.begin() + hi, cmp);
    rcb_rec(idx, lo, lo + nleft, part0, pl, pos, dim, owner);
    rcb_rec(idx, lo + nleft, hi, part0 + pl, parts - pl, pos, dim, owner);
}

}  // namespace

void rcb_partition(int32_t N, const float* pos, int32_t pos_dim, int32_t parts, std::vector<int32_t>& owner) {
    owner.assign(N, 0);
    if (parts <= 1) return;
    if (!pos || pos_dim <= 0) {
        for (int32_t i = 0; i < N; ++i) owner[i] = (int32_t)(((int64_t)i * parts) / N);
        return;
    }
    std::vector<int32_t> idx(N);
    std::iota(idx.begin(), idx.end(), 0);
    rcb_rec(idx, 0, N, 0, parts, pos, pos_dim, owner);
}

double locality_cost(const EdgeList& es, const std::vector<int32_t>& order_pos) {
    double sum = 0.0;
    int64_t n = 0;
    for (int64_t i = 0; i < es.E; ++i) {
        const int32_t a = order_pos[es.senders[i] - es.index_base], b = order_pos[es.receivers[i] - es.index_base];
        if (a < 0 || b < 0) continue;
        sum += (double)(a > b ? a - b : b - a);
        ++n;
    }
    return n > 0 ? sum / (double)n : 0.0;
}

namespace {

// Breadth-first order of the nodes `member` marks, over the edges of `es` (both directions), started from the lowest global id of
// every connected component, neighbours in the order the edge list names them.  Deterministic: every rank derives the same order
// from the same lists.  On a mesh the fronts are curves, so the ends of an edge land within one front width of each other -- the
// locality a generator's own numbering has -- whatever labels the nodes arrived with (DeepMind's trajectories carry arbitrary ones
// and create_base_graph passes them through, reference src/graph.jl:30-36).
std::vector<int32_t> bfs_order(int32_t N, const EdgeList& es, const std::vector<uint8_t>& member) {
    std::vector<int64_t> ptr((size_t)N + 1, 0);
    auto in = [&](int64_t i, int32_t& a, int32_t& b) {
        a = es.senders[i] - es.index_base;
        b = es.receivers[i] - es.index_base;
        return a != b && member[a] && member[b];
    };
    int32_t a, b;
    for (int64_t i = 0; i < es.E; ++i)
        if (in(i, a, b)) { ++ptr[(size_t)a + 1]; ++ptr[(size_t)b + 1]; }
    for (int32_t i = 0; i < N; ++i) ptr[(size_t)i + 1] += ptr[i];
    std::vector<int32_t> adj((size_t)ptr[N]);
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < es.E; ++i)
        if (in(i, a, b)) { adj[(size_t)cur[a]++] = b; adj[(size_t)cur[b]++] = a; }
    std::vector<int32_t> order;
    std::vector<uint8_t> seen((size_t)N, 0);
    for (int32_t root = 0; root < N; ++root) {
        if (!member[root] || seen[root]) continue;
        size_t head = order.size();
        order.push_back(root);
        seen[root] = 1;
        while (head < order.size()) {
            const int32_t u = order[head++];
            for (int64_t j = ptr[u]; j < ptr[(size_t)u + 1]; ++j) {
                const int32_t v = adj[(size_t)j];
                if (!seen[v]) { seen[v] = 1; order.push_back(v); }
            }
        }
    }
    return order;
}

}  // namespace

std::string build_local_graph(int32_t N, int nsets, const EdgeList* sets, const float* pos, int32_t pos_dim,
                              const int32_t* owner_in, int32_t rank, int32_t nranks, LocalGraph& g, int renumber) {
    if (N < 0) return "negative N";
    if (nsets < 1 || nsets > MAX_EDGE_SETS) return "bad number of edge sets";
    if (nranks < 1 || rank < 0 || rank >= nranks) return "bad rank/nranks";
    for (int k = 0; k < nsets; ++k) {
        const EdgeList& es = sets[k];
        if (es.E < 0) return "negative E";
        if (es.E > 0 && (!es.senders || !es.receivers)) return "null senders/receivers";
        if (es.E >= ((int64_t)1 << 31) || es.E_global >= ((int64_t)1 << 31)) return "E >= 2^31 not supported";
        if (es.gid) {
            if (es.E_global < es.E) return "rank-local edge list longer than the global list it is a part of";
            for (int64_t i = 0; i < es.E; ++i)
                if (es.gid[i] < 0 || es.gid[i] >= es.E_global || (i > 0 && es.gid[i] <= es.gid[i - 1]))
                    return "rank-local edge list: global positions must be ascending and inside [0, E_global)";
        }
    }
    std::vector<int32_t> owner_keep;
    if (owner_in) owner_keep.assign(owner_in, owner_in + N);   // owner_in may alias g.owner
    g = LocalGraph();
    g.N = N;
    g.rank = rank;
    g.nranks = nranks;
    g.nsets = nsets;
    if (owner_in) g.owner.swap(owner_keep);
    else rcb_partition(N, pos, pos_dim, nranks, g.owner);
    // One pass over the GLOBAL lists: range check, and -- on a partition -- the edges this rank has anything to do with (an end it owns:
    // 1 / nranks of them plus the cut), in list order.  Every later pass walks those (a rank of eight read all 6 M edges of M-1M six times).
    std::vector<int64_t> rel[MAX_EDGE_SETS];
    for (int k = 0; k < nsets; ++k) {
        const EdgeList& es = sets[k];
        if (nranks > 1) rel[k].reserve((size_t)(es.E / nranks + es.E / 64 + 16));
        for (int64_t i = 0; i < es.E; ++i) {
            const int64_t s = (int64_t)es.senders[i] - es.index_base, r = (int64_t)es.receivers[i] - es.index_base;
            if (s < 0 || s >= N || r < 0 || r >= N)
                return "edge index out of range at edge " + std::to_string(i) + " of set " + std::to_string(k);
            if (nranks > 1 && (g.owner[s] == rank || g.owner[r] == rank)) rel[k].push_back(i);
        }
    }
    // edges of set k that touch this rank, in list order: fn(edge position, sender, receiver)
    auto for_edges = [&](int k, auto&& fn) {
        const EdgeList& es = sets[k];
        if (nranks > 1) {
            for (const int64_t i : rel[k]) fn(i, (int32_t)(es.senders[i] - es.index_base), (int32_t)(es.receivers[i] - es.index_base));
        } else {
            for (int64_t i = 0; i < es.E; ++i) fn(i, (int32_t)(es.senders[i] - es.index_base), (int32_t)(es.receivers[i] - es.index_base));
        }
    };

    // owned nodes: boundary nodes (senders of an edge received on another rank) first, then interior, each in
    // ascending global id.  Boundary-first lets the driver project the boundary tiles, start the halo exchange and
    // overlap it with the projection of the interior tiles.
    std::vector<int32_t> g2l(N, -1);
    {
        std::vector<uint8_t> is_bnd(N, 0);
        if (nranks > 1)
            for (int k = 0; k < nsets; ++k)
                for_edges(k, [&](int64_t, int32_t s, int32_t r) {
                    if (g.owner[s] == rank && g.owner[r] != rank) is_bnd[s] = 1;
                });
        // the order of the owned nodes inside the two groups: ascending global id, or breadth-first over the mesh (see header)
        std::vector<int32_t> order;
        if (renumber != 0 && sets[0].E > 0) {
            std::vector<uint8_t> mine((size_t)N);
            int32_t n_mine = 0;
            for (int32_t i = 0; i < N; ++i) n_mine += (mine[i] = g.owner[i] == rank);
            order = bfs_order(N, sets[0], mine);
            bool use = renumber == 2;
            if (!use) {
                std::vector<int32_t> p_asc((size_t)N, -1), p_bfs((size_t)N, -1);
                int32_t k = 0;
                for (int32_t i = 0; i < N; ++i)
                    if (mine[i]) p_asc[i] = k++;
                for (size_t j = 0; j < order.size(); ++j) p_bfs[order[j]] = (int32_t)j;
                use = locality_cost(sets[0], p_asc) > RENUMBER_GAIN * locality_cost(sets[0], p_bfs);
            }
            if (!use || (int32_t)order.size() != n_mine) order.clear();
        }
        g.renumbered = !order.empty();
        if (order.empty())
            for (int32_t i = 0; i < N; ++i)
                if (g.owner[i] == rank) order.push_back(i);
        for (int pass = 0; pass < 2; ++pass) {
            for (int32_t i : order)
                if ((is_bnd[i] != 0) == (pass == 0)) {
                    g2l[i] = (int32_t)g.own_gid.size();
                    g.own_gid.push_back(i);
                }
            if (pass == 0) g.n_boundary = (int32_t)g.own_gid.size();
        }
    }
    g.n_own = (int32_t)g.own_gid.size();

    // halo nodes: remote senders of local edges (any set), grouped by owner rank then ascending gid
    {
        std::vector<uint8_t> is_halo(N, 0);
        for (int k = 0; k < nsets; ++k)
            for_edges(k, [&](int64_t, int32_t s, int32_t r) {
                if (g.owner[r] == rank && g.owner[s] != rank) is_halo[s] = 1;
            });
        g.recv_rows.assign(nranks, 0);
        for (int32_t i = 0; i < N; ++i)
            if (is_halo[i]) ++g.recv_rows[g.owner[i]];
        std::vector<int32_t> off(nranks + 1, 0);
        for (int32_t q = 0; q < nranks; ++q) off[q + 1] = off[q] + g.recv_rows[q];
        g.n_halo = off[nranks];
        g.halo_gid.assign(g.n_halo, 0);
        std::vector<int32_t> cur(off.begin(), off.end() - 1);
        for (int32_t i = 0; i < N; ++i)  // ascending gid inside each owner group
            if (is_halo[i]) {
                const int32_t q = g.owner[i];
                g2l[i] = g.n_own + cur[q];
                g.halo_gid[cur[q]++] = i;
            }
    }

    // per set: local edges = edges whose receiver is owned, in receiver-sorted, input-stable order (counting sort)
    for (int k = 0; k < nsets; ++k) {
        const EdgeList& es = sets[k];
        EdgeTopo& t = g.set[k];
        t.E = es.gid ? es.E_global : es.E;
        t.rowptr.assign((size_t)g.n_own + 1, 0);
        int64_t el = 0;
        for_edges(k, [&](int64_t, int32_t, int32_t r) {
            if (g.owner[r] == rank) {
                ++t.rowptr[(size_t)g2l[r] + 1];
                ++el;
            }
        });
        t.e_local = el;
        for (int32_t i = 0; i < g.n_own; ++i) t.rowptr[(size_t)i + 1] += t.rowptr[i];
        t.snd.assign(el, 0);
        t.rcv.assign(el, 0);
        t.edge_gid.assign(el, 0);
        std::vector<int32_t> cur(t.rowptr.begin(), t.rowptr.end() - 1);
        for_edges(k, [&](int64_t i, int32_t s, int32_t r) {
            if (g.owner[r] != rank) return;
            const int32_t lr = g2l[r];
            const int32_t p = cur[lr]++;
            t.snd[p] = g2l[s];
            t.rcv[p] = lr;
            t.edge_gid[p] = es.gid ? es.gid[i] : i;
        });
        t.halo_span = 0;
        for (int64_t p = el - 1; p >= 0; --p)
            if (t.snd[p] >= g.n_own) { t.halo_span = p + 1; break; }
    }

    // send lists: my owned nodes that are senders of edges received on peer q (unique, ascending gid)
    g.send_rows.assign(nranks, 0);
    g.send_idx.clear();
    if (nranks > 1) {
        std::vector<std::vector<int32_t>> lists(nranks);
        for (int k = 0; k < nsets; ++k)
            for_edges(k, [&](int64_t, int32_t s, int32_t r) {
                const int32_t q = g.owner[r];
                if (g.owner[s] == rank && q != rank) lists[q].push_back(s);
            });
        for (int32_t q = 0; q < nranks; ++q) {
            auto& l = lists[q];
            std::sort(l.begin(), l.end());
            l.erase(std::unique(l.begin(), l.end()), l.end());
            g.send_rows[q] = (int32_t)l.size();
            for (int32_t gid : l) g.send_idx.push_back(g2l[gid]);
        }
    }
    return std::string();
}

}  // namespace mgn
