// Host-side graph preparation: receiver sort, CSR, edge-cut partition (RCB), halo lists.
// Pure C++ (no HIP): exercised on CPU by tests through the mgn_* introspection entry points.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mgn {

constexpr int MAX_EDGE_SETS = 2;  // mesh edges (+ world edges, MGN-spec "per edge set")

// One edge set of this rank: the edges whose receiver is owned, receiver-sorted (input-stable).
struct EdgeTopo {
    int64_t E = 0;                   // global edges of the set
    int64_t e_local = 0;
    std::vector<int64_t> edge_gid;   // [e_local] global edge id, engine (receiver-sorted, stable) order
    std::vector<int32_t> snd, rcv;   // [e_local] local indices; snd may be >= n_own (halo)
    std::vector<int32_t> rowptr;     // [n_own+1] CSR by receiver
    int64_t halo_span = 0;           // 1 + position of the last edge whose sender is a halo node (0: none).  Mesh edges are
                                     // two-way, so such edges end at boundary nodes, which are numbered first: the span is short
};

struct LocalGraph {
    int32_t N = 0;              // global nodes
    int32_t rank = 0, nranks = 1;
    int32_t nsets = 1;
    int32_t n_own = 0, n_halo = 0;
    int32_t n_boundary = 0;          // owned nodes that some peer lists as halo; they are numbered FIRST
    EdgeTopo set[MAX_EDGE_SETS];
    std::vector<int32_t> own_gid;    // [n_own] global id of owned node i: boundary nodes, then interior; inside each group ascending, or
                                     // in breadth-first order of the mesh when the numbering it arrived with is scattered (`renumbered`)
    bool renumbered = false;
    std::vector<int32_t> halo_gid;   // [n_halo] grouped by owner rank, ascending gid inside a group (union over the edge sets)
    std::vector<int32_t> send_rows;  // [nranks] rows this rank sends to each peer per exchange
    std::vector<int32_t> recv_rows;  // [nranks] rows received from each peer (== halo group sizes)
    std::vector<int32_t> send_idx;   // [sum(send_rows)] local (owned) row of each sent row, peer-major
    std::vector<int32_t> owner;      // [N] owner rank of every global node (kept for tests/introspection)
};

// Deterministic recursive coordinate bisection of N points into `parts` parts of near-equal count
// (split the longer bounding-box axis at the count-proportional median).  pos may be null: then
// contiguous index blocks.  Every rank computes the same answer from the same inputs.
void rcb_partition(int32_t N, const float* pos, int32_t pos_dim, int32_t parts, std::vector<int32_t>& owner);

// Global description of one edge set as the caller hands it over.
// gid != null (rank-local ingest, mgn_set_graph_local): the arrays hold only the E edges this rank has an end of, in ascending global
// position gid[i] of a global list of E_global edges that the rank never sees.
struct EdgeList {
    int64_t E = 0;
    const int32_t* senders = nullptr;
    const int32_t* receivers = nullptr;
    int32_t index_base = 0;
    const int64_t* gid = nullptr;
    int64_t E_global = -1;
};

// Build rank `rank`'s local graph from `nsets` edge sets over the same nodes.  Node ownership comes from
// `owner` when given ([N], e.g. kept from an earlier call), else from rcb_partition(pos).  Halo / boundary /
// send lists are the union over the sets.  Returns empty string on success, else an error message.
// renumber: 0 = owned nodes in ascending global id (inside the boundary / interior groups), 2 = in breadth-first order of the mesh
// (edge set 0), 1 = whichever of the two keeps the ends of an edge closer together (locality_cost; breadth-first only when it is
// at least RENUMBER_GAIN times better: a mesh that arrives coherently numbered keeps its numbering).
constexpr double RENUMBER_GAIN = 4.0;
std::string build_local_graph(int32_t N, int nsets, const EdgeList* sets, const float* pos, int32_t pos_dim,
                              const int32_t* owner, int32_t rank, int32_t nranks, LocalGraph& g, int renumber = 1);

// Mean |position(sender) - position(receiver)| over the edges of `es` whose two ends are both listed in `order_pos` (>= 0): what a
// numbering costs the kernels that gather sender rows for receiver-sorted edge tiles (rows far apart share no cache line, no L2 set
// of the XCD that sweeps the tile range, no TLB entry).
double locality_cost(const EdgeList& es, const std::vector<int32_t>& order_pos);

}  // namespace mgn
