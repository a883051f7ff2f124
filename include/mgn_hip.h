/*
 * mgn_hip.h -- C ABI of the MI355X-native MeshGraphNets Encode-Process-Decode engine.
 *
 * This is the drop-in boundary for the ONE hot path of una-auxme/MeshGraphNets.jl: everything the
 * reference obtains from `using GraphNetCore` (reference src/MeshGraphNets.jl:8) that runs per
 * ODE right-hand side / per training datapoint.  A thin Julia `ccall` shim (julia/MGNHip.jl) or the
 * Python ctypes host (meshgraphnets.jl_amd/engine.py) binds exactly these symbols; see
 * INTEGRATION.md.  No C++ types, no torch types: plain pointers and sizes.
 *
 * Conventions
 *   - All matrices are C row-major [count][feat] == the bytes of Julia's column-major (feat x count).
 *   - Every function returns 0 (MGN_OK) or a negative mgn_status; text via mgn_last_error().
 *   - A handle is NOT thread-safe; one in-flight call per handle (the reference calls the model
 *     strictly sequentially from one task: ODE RHS inside `solve`, src/solve.jl:53-61).
 *   - Host-pointer entry points copy in/out and return after the D2H copy; `_dev` entry points take
 *     device pointers, enqueue on the handle's stream (mgn_set_stream) and do not synchronise.
 *   - The engine owns all device memory it allocates; caller owns every buffer it passes in.
 *   - There is no CPU fallback: without a HIP device every compute entry point fails with MGN_E_HIP.
 */
#ifndef MGN_HIP_H
#define MGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mgn_status {
    MGN_OK = 0,
    MGN_E_ARG = -1,   /* bad argument (reference: ArgumentError / DimensionMismatch)            */
    MGN_E_HIP = -2,   /* HIP runtime error / no device                                          */
    MGN_E_STATE = -3, /* call order violated (e.g. forward before set_params / set_graph)       */
    MGN_E_OOM = -4,   /* device or host allocation failed                                       */
    MGN_E_UNSUPPORTED = -5,
    MGN_E_RCCL = -6   /* communicator: RCCL error, peer time-out, exchange before mgn_comm_init              */
} mgn_status;

typedef enum mgn_dtype { MGN_F32 = 0, MGN_BF16 = 1 } mgn_dtype;

/* Mirrors the arguments of GraphNetCore.load(quantities, dims, e_norms, n_norms, o_norms, outputs,
 * mps, layer_size, hidden_layers, opt, device, path) at reference src/MeshGraphNets.jl:282-285:
 *   Fn = quantities, Fe = dims+1 (src/graph.jl:49-52), O = outputs, L = layer_size.            */
typedef struct mgn_config {
    int32_t Fn;            /* node input width  (cylinder_flow: 9 = velocity 2 + one-hot 7)       */
    int32_t Fe;            /* edge input width  (2-D mesh: 3 = rel pos 2 + norm)                  */
    int32_t O;             /* output width      (cylinder_flow: 2)                                */
    int32_t L;             /* latent width `layer_size`; HIP path supports 32, 64, 128            */
    int32_t hidden_layers; /* hidden layers per MLP (h + 1 Dense): 1 .. 4.  2 (the example's value) runs the tuned kernel */
                           /* families; other counts run the general instantiations (mgn_step / mgn_ode_vjp /      */
                           /* mgn_forward_vjp follow it too; MGN_BF16 is specialised for 2 and refuses others)     */
    int32_t mps;           /* message passing steps                                               */
    int32_t dtype;         /* mgn_dtype: MGN_F32, or MGN_BF16 (L = 128: bf16 storage + bf16 MFMA in the processor, */
                           /* fp32 accumulate / LayerNorm / residual / aggregation; encoder, decoder in fp32)      */
    int32_t rank;          /* this process's partition, 0 <= rank < nranks                        */
    int32_t nranks;        /* number of edge-cut partitions (1 = whole mesh on this GPU)          */
    int32_t device;        /* HIP device ordinal, -1 = current device, MGN_DEVICE_NONE = host-only handle */
    int32_t n_edge_sets;   /* 0 or 1: the reference's single edge set (FeatureGraph, src/graph.jl:87-96);          */
                           /* 2: mesh edges + world edges (MGN-spec "per edge set"; flag_simple-shaped, BASELINE cfg-3) */
    int32_t Fe2;           /* edge input width of the second set (used when n_edge_sets == 2)                      */
    int32_t ln_mode;       /* mgn_ln_mode: which LayerNorm GraphNetCore / Lux compute (the sources are not vendored:     */
                           /* julia/spec_probe.jl tells).  MGN_LN_VAR_EPS (0, MGN-spec v1): (x - mean) / sqrt(var + eps);  */
                           /* MGN_LN_STD_EPS: (x - mean) / (sqrt(var) + eps).  eps = 1e-5, biased variance, per row.       */
                           /* Every entry point follows it, the reverse passes (mgn_step, mgn_forward_vjp, mgn_ode_vjp) included */
    int32_t ln_dims;       /* mgn_ln_dims: MGN_LN_ROWS (0, MGN-spec v1): LayerNorm statistics per node / edge over its L features.   */
                           /* MGN_LN_ALL: over the WHOLE (L x rows) array of an MLP's output -- what Lux 0.5's LayerNorm(shape) computes  */
                           /* when it is left at dims = Colon() (julia/spec_probe.jl tells which one the installed GraphNetCore / Lux    */
                           /* run).  A different network, not a numerical variant: every LayerNorm then couples all rows, so the fused    */
                           /* tile kernels cannot be used; the mode runs unfused (MLP kernel with its LayerNorm off, grid-wide statistics    */
                           /* in double, apply pass) behind mgn_forward, mgn_processor_steps, mgn_set_static + mgn_ode_step (both forms),   */
                           /* mgn_rollout, mgn_processor_steps_dev (the resident latents pass through the same driver as rows), and in     */
                           /* both directions behind mgn_step, mgn_forward_vjp, mgn_ode_vjp: fp32, one partition, one edge set.  The staged */
                           /* per-step calls of the fused kernels (mgn_proc_*, mgn_fwd_*) answer MGN_E_UNSUPPORTED                          */
} mgn_config;

typedef enum { MGN_LN_VAR_EPS = 0, MGN_LN_STD_EPS = 1 } mgn_ln_mode;
typedef enum { MGN_LN_ROWS = 0, MGN_LN_ALL = 1 } mgn_ln_dims;

#define MGN_MAX_EDGE_SETS 2

/* A handle created with device == MGN_DEVICE_NONE owns no GPU state: only mgn_set_graph and the partition
 * introspection calls work on it (host logic: receiver sort, CSR, edge-cut partition, halo lists); every
 * compute entry point returns MGN_E_HIP.  It exists so that the partitioner can be used and tested on
 * hosts without a GPU; it is not a CPU compute path. */
#define MGN_DEVICE_NONE (-2)

typedef struct mgn_engine mgn_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* Version of THIS header's structs and prototypes; bumped whenever mgn_config / mgn_rollout_desc grow or a prototype changes.  A
 * binding compiled against another version must not call in (a shorter mgn_config would be read past its end): the Python and
 * Julia bindings compare mgn_abi_version() with the constant they were written for before the first mgn_create. */
#define MGN_ABI_VERSION 4
int mgn_abi_version(void);
int mgn_create(const mgn_config* cfg, mgn_handle** out);
void mgn_destroy(mgn_handle* h);
const char* mgn_last_error(const mgn_handle* h); /* h may be NULL: error of the last failed mgn_create */
/* Stream all work of the handle is enqueued on.  hip_stream is a hipStream_t; NULL is HIP's default (null)
 * stream -- what torch.cuda.current_stream().cuda_stream returns for torch's default stream -- and
 * MGN_STREAM_OWN selects the engine's private non-blocking stream (the state after mgn_create).  Callers that
 * mix engine calls with other GPU work (RCCL collectives, torch copies) must pass THEIR stream here. */
#define MGN_STREAM_OWN ((void*)(intptr_t)-1)
int mgn_set_stream(mgn_handle* h, void* hip_stream);
int mgn_synchronize(mgn_handle* h);

/* ---- parameters: mgn.ps (reference src/MeshGraphNets.jl:288,376-377) ------------------------- */
/* Packed order (MGN-spec v1): enc-node, enc-edge, step1-edge, step1-node, ..., decoder;
 * within an MLP W1,b1,W2,b2,W3,b3,[ln_scale,ln_bias]; each W row-major [in][out].
 * mgn_set_params copies the vector (0.1 ms for the 15-step model) and is meant to be called whenever the caller's
 * parameters MAY have changed -- the reference's loop updates them before every step! (src/MeshGraphNets.jl:375-377):
 * values equal to the ones held invalidate nothing; new ones are turned into the kernels' layouts on the device by the
 * next call that computes with them (1.5 ms for the inference kernels, 0.4 ms inside mgn_step).               */
size_t mgn_param_count(const mgn_config* cfg);
int mgn_set_params(mgn_handle* h, const float* packed, size_t n);
int mgn_get_params(mgn_handle* h, float* packed, size_t n);

/* ---- normalisers: mgn.n_norm / e_norm / o_norm frozen to per-feature affine maps --------------
 * forward:  y = x*scale + shift  (covers NormaliserOfflineMinMax, OfflineMeanStd and a frozen
 * NormaliserOnline; reference src/MeshGraphNets.jl:79-203, applied src/graph.jl:80-93);
 * inverse_data(o_norm, y) = y*out_scale + out_shift (src/solve.jl:205-210).
 * Any pointer may be NULL = identity.  Used by mgn_ode_step only.                                */
int mgn_set_norms(mgn_handle* h, const float* node_scale, const float* node_shift, /* [Fn] */
                  const float* edge_scale, const float* edge_shift,               /* [Fe] */
                  const float* out_scale, const float* out_shift);                /* [O]  */

/* ---- graph: once per trajectory, like create_base_graph (reference src/graph.jl:25-55, called at
 * src/MeshGraphNets.jl:360,418,596).  Sort-by-receiver, CSR, tiling and (nranks>1) the edge-cut
 * partition + halo lists are built here.  index_base is 1 at the Julia boundary (src/graph.jl:31-34).
 * mesh_pos [N][pos_dim] is optional; it drives the geometric partition when nranks > 1.         */
int mgn_set_graph(mgn_handle* h, int32_t N, int64_t E, const int32_t* senders, const int32_t* receivers,
                  int32_t index_base, const float* mesh_pos, int32_t pos_dim);

/* Rank-local ingest of the same graph (nranks > 1 at scale: mgn_set_graph walks the GLOBAL edge lists on every rank).
 * mgn_partition_nodes: the node partition mgn_set_graph would derive -- recursive coordinate bisection of mesh_pos [N][pos_dim], or
 * contiguous index blocks when mesh_pos is NULL -- as owner[N]; deterministic, no handle needed, every rank gets the same map.
 * mgn_set_graph_local: rank h->cfg.rank hands over only the edges it has an END of (sender or receiver owned; the others concern
 * neither its tiles nor its send lists), in the order of the global list, with their positions edge_gid[E_touch] (ascending) in that
 * list of E_global edges.  The handle ends up in exactly the state mgn_set_graph(N, E_global, ...) with the same partition leaves it
 * in (tests/test_abi_and_host.py); arrays indexed by edge at the boundary ([E][Fe] inputs, latents) stay GLOBAL arrays, of which the
 * rank touches the rows of its local edges.  One edge set; index_base as in mgn_set_graph.                                          */
int mgn_partition_nodes(int32_t N, const float* mesh_pos, int32_t pos_dim, int32_t nranks, int32_t* owner);
int mgn_set_graph_local(mgn_handle* h, int32_t N, const int32_t* owner, int64_t E_global, int64_t E_touch, const int32_t* senders,
                        const int32_t* receivers, const int64_t* edge_gid, int32_t index_base);

/* Second edge set (n_edge_sets == 2): topology of set `set` (>= 1) over the same N nodes, after mgn_set_graph and
 * as often as it changes (world edges are re-searched every step of a cloth rollout; the mesh set, the node
 * partition and the node order stay).  Edges live with their receiver's owner like mesh edges; with nranks > 1
 * the halo / send lists become the union over the sets, so query mgn_halo_* again afterwards.  E may be 0.
 * The reference has one edge set; this is the MGN-spec extension SURVEY.md 8c defines (GOLD-C).                */
int mgn_set_edge_set(mgn_handle* h, int32_t set, int64_t E, const int32_t* senders, const int32_t* receivers,
                     int32_t index_base);
/* Raw features [E_set][Fe2] of set >= 1 for the next mgn_forward / mgn_fwd_upload (which carry set 0's `ef`).
 * No normaliser is applied to them (normalise on the host, or fold the affine map into the first layer).       */
int mgn_set_edge_features(mgn_handle* h, int32_t set, const float* ef);
int mgn_edge_set_info(const mgn_handle* h, int32_t set, int64_t* E, int64_t* e_local);
/* Latents of edge set `set` (0 included), host GLOBAL arrays [E_set][L]; mgn_latents_import/export cover v and set 0. */
int mgn_edge_latents_import(mgn_handle* h, int32_t set, const float* e);
int mgn_edge_latents_export(mgn_handle* h, int32_t set, float* e);

/* Partition introspection (nranks == 1: n_own = N, n_halo = 0, e_local = E). */
int mgn_partition_info(const mgn_handle* h, int32_t* n_own, int32_t* n_halo, int64_t* e_local);
int mgn_owned_nodes(const mgn_handle* h, int32_t* global_ids /* [n_own], 0-based */);
int mgn_local_edges(const mgn_handle* h, int64_t* global_edge_ids /* [e_local], engine order */);
int mgn_halo_counts(const mgn_handle* h, int32_t* send_rows /* [nranks] */, int32_t* recv_rows /* [nranks] */);
int mgn_halo_nodes(const mgn_handle* h, int32_t* global_ids /* [n_halo], grouped by owner rank */);
int mgn_halo_send_index(const mgn_handle* h, int32_t* local_rows /* [sum(send_rows)] owned local rows, peer-major */);
/* local (receiver-sorted) edge list: snd may index halo rows (>= n_own); rowptr is CSR by receiver */
int mgn_local_graph(const mgn_handle* h, int32_t* snd /* [e_local] */, int32_t* rcv /* [e_local] */, int32_t* rowptr /* [n_own+1] */);
int mgn_node_owner(const mgn_handle* h, int32_t* owner /* [N] rank owning each global node */);
/* owned nodes are numbered boundary-first: local rows [0, n_boundary) are the nodes some peer lists as halo */
int mgn_boundary_count(const mgn_handle* h, int32_t* n_boundary);

/* ---- the model: mgn.model(graph, ps, st) -> output (reference src/solve.jl:200) ---------------
 * nf [N][Fn], ef [E][Fe] are the FeatureGraph fields (already normalised, src/graph.jl:87-96);
 * out [N][O].  With nranks > 1 every rank passes the GLOBAL arrays (each uploads only the rows it owns) and, after
 * mgn_comm_init, receives the complete GLOBAL output (owned rows are gathered over the communicator).           */
int mgn_forward(mgn_handle* h, const float* nf, const float* ef, float* out);

/* ---- the fused RHS: ode_step (reference src/solve.jl:188-219) incl. build_graph (src/graph.jl:75-97)
 * x [N][O] state, node_type_onehot [N][Fn-O] (raw), ef_raw [E][Fe] (raw edge_features),
 * val_mask [N] (0/1; reference repeats it over the O rows, src/MeshGraphNets.jl:588-591), may be NULL.
 * dxdt [N][O] = inverse_data(o_norm, model(graph)) .* val_mask.                                 */
int mgn_ode_step(mgn_handle* h, const float* x, const float* node_type_onehot, const float* ef_raw,
                 const float* val_mask, float* dxdt);

/* Static per-trajectory inputs of the RHS, set ONCE (like create_base_graph's outputs, reference src/graph.jl:54):
 * node_type_onehot [N][Fn-O], ef_raw [E][Fe], val_mask [N] or NULL.  They are uploaded once, the edge encoder runs
 * once and its output is cached; afterwards mgn_ode_step may be called with node_type_onehot = ef_raw = val_mask =
 * NULL and only moves the O x N state per call (on small meshes it replays a hipGraph of the launch sequence).
 * Invalidated by mgn_set_params / mgn_set_norms / mgn_set_graph. */
int mgn_set_static(mgn_handle* h, const float* node_type_onehot, const float* ef_raw, const float* val_mask);

/* ---- graph prologue helpers (GraphNetCore utilities used at reference src/graph.jl:26-52) -------
 * Scalable replacements for the reference's per-edge host loops (which cannot build a 6 M-edge graph).
 * No handle, no GPU needed.
 * mgn_triangles_to_edges: cells [C][3] (any index base) -> unique undirected edges, returned two-way
 *   (senders = [a;b], receivers = [b;a], a = max, b = min) in first-occurrence order like the reference.
 *   Call with senders = receivers = NULL to get the count; *n_directed receives 2 x (unique undirected edges).
 * mgn_edge_features: ef[e] = [pos[s]-pos[r] ; ||pos[s]-pos[r]||]  (src/graph.jl:35-36,49-52), ef is [E][dim+1]. */
int mgn_triangles_to_edges(const int32_t* cells, int64_t n_cells, int32_t* senders, int32_t* receivers, int64_t* n_directed);
/* World edges of a cloth-like mesh (second edge set of MGN-spec; no reference symbol, the reference has one edge set):
 * every ordered pair (s, r), s != r, closer than `radius` in world space and not already joined by a mesh edge; uniform
 * grid search, receiver-major output with ascending senders.  Call with senders = receivers = NULL to get the count in
 * *n_edges; with buffers, *n_edges holds their capacity on entry and the number written on return.               */
int mgn_world_edges(const float* world_pos, int32_t dim, int32_t N, float radius, const int32_t* mesh_senders,
                    const int32_t* mesh_receivers, int64_t n_mesh, int32_t index_base, int32_t* senders, int32_t* receivers,
                    int64_t* n_edges);
int mgn_edge_features(const float* mesh_pos, int32_t pos_dim, const int32_t* senders, const int32_t* receivers,
                      int64_t E, int32_t index_base, float* ef);

/* The same prologue ON THE DEVICE: the reference's per-edge host loops cannot build a 6 M-edge graph.
 * Integer results are bit-identical to the host versions above (same packed (max, min) keys, stable radix sort, first-occurrence
 * order restored by a second sort).  Pointers may be host or device memory (hipMemcpyDefault).
 * mgn_triangles_to_edges_dev: as mgn_triangles_to_edges; `capacity` = entries of each output buffer (6 x n_cells always suffices);
 *   MGN_E_ARG with *n_directed set when they are too small.
 * mgn_set_static_mesh: create_base_graph's FEATURE half (src/graph.jl:26-27,35-36,49-52) into the engine's resident static inputs:
 *   one_hot(node_type, depth = type_max - type_min + 1, offset = 1 - type_min) and edge_features = [mesh_pos[s] - mesh_pos[r]; norm]
 *   are computed by kernels in engine order -- no [E][Fe] host array, no PCIe transfer of it -- then the edge encoder runs once, as
 *   in mgn_set_static; afterwards mgn_ode_step(x, NULL, NULL, NULL) / the resident fast path apply.  Needs Fn - O == depth, Fe == pos_dim + 1.
 * mgn_world_edges_dev: the world-edge set `set` (>= 1) of a cloth-like mesh searched on the device (uniform grid, radix sort by cell)
 *   and installed WITHOUT a host round trip (what mgn_world_edges + mgn_set_edge_set do on the host; cloth rollouts re-search every
 *   step); with Fe2 == dim + 1 its features [rel world pos; norm] are written too.  One partition.  The same edges in the same
 *   order as mgn_world_edges (receiver-major, ascending senders).  mgn_edge_set_export hands the installed lists back (0-based). */
int mgn_triangles_to_edges_dev(mgn_handle* h, const int32_t* cells, int64_t n_cells, int32_t* senders, int32_t* receivers, int64_t capacity,
                               int64_t* n_directed);
int mgn_set_static_mesh(mgn_handle* h, const int32_t* node_type /* [N] */, int32_t type_min, int32_t type_max, const float* mesh_pos /* [N][pos_dim] */,
                        int32_t pos_dim, const float* val_mask /* [N] or NULL */);
int mgn_world_edges_dev(mgn_handle* h, int32_t set, const float* world_pos /* [N][dim] */, int32_t dim, float radius, int64_t* n_edges);
int mgn_edge_set_export(mgn_handle* h, int32_t set, int32_t* senders, int32_t* receivers);

/* Online-normaliser accumulation as a device reduction: what GraphNetCore's NormaliserOnline adds up per call
 * (normalisers built at reference src/MeshGraphNets.jl:92,193-199, applied in build_graph src/graph.jl:75-97):
 * sum[f] = sum_r x[r][f], sum_squares[f] = sum_r x[r][f]^2 over x [rows][dim] (host or device pointer), accumulated in double
 * in a fixed order (bitwise repeatable).  The caller keeps the running totals, count and max_accumulations logic.          */
int mgn_feature_stats(mgn_handle* h, const float* x, int64_t rows, int32_t dim, double* sum, double* sum_squares);

/* ---- native rollout driver: the whole `rollout` of reference src/solve.jl:42-68 on the device:
 * ODEProblem(ode_func_eval, x0, (t0, t1), ...) solved with fixed-step Euler (`adaptive = false, dt = dt`) or an
 * adaptive Tsit5 (own tableau + PI step controller, tstops = saveat = t0 + i*saves_dt), the right-hand side being
 * ode_func_eval (inflow overwrite from `inflow_data[floor(t / saves_dt)]`, src/solve.jl:151-152 -- see inflow_rule --, applied IN PLACE to
 * the array the RHS is evaluated on, like the reference) -> ode_step (mgn_ode_step semantics).  No host round trip
 * per RHS; one small D2H (error norm) per adaptive step.  Normalisers must be set with mgn_set_norms.
 * With nranks > 1 (after mgn_comm_init) every rank passes the GLOBAL arrays and integrates the rows it owns; the error norm of
 * the step controller is reduced over the ranks (the same bits everywhere: the same accept / reject decisions), the halo exchange
 * runs inside every right-hand side, and every rank returns the complete solution.  mgn_ode_step and mgn_set_static likewise
 * (global arrays in, complete dx/dt out on every rank). */
typedef struct mgn_rollout_desc {
    int32_t solver;          /* 0 = Euler fixed step, 1 = Tsit5 adaptive                                     */
    float t0, t1;            /* integration interval                                                         */
    float dt;                /* Euler: step; Tsit5: initial step (0 = automatic)                             */
    float saves_dt;          /* spacing of the save points AND of the inflow_data frames                     */
    int32_t n_saves;         /* solution is stored at t0 + i*saves_dt, i = 0..n_saves-1                      */
    float abstol, reltol;    /* Tsit5 only (OrdinaryDiffEq defaults: 1e-6, 1e-3)                             */
    const float* x0;               /* [N][O]                                                                 */
    const float* node_type_onehot; /* [N][Fn-O]                                                              */
    const float* ef_raw;           /* [E][Fe]                                                                */
    const float* val_mask;         /* [N] or NULL                                                            */
    const uint8_t* inflow_mask;    /* [N] 0/1 or NULL: rows overwritten from inflow_data                      */
    const float* inflow_data;      /* [n_frames][N][O] or NULL                                               */
    int32_t n_frames;
    float* out;                    /* [n_saves][N][O]                                                        */
    int32_t n_accept, n_reject, n_rhs; /* filled on return                                                   */
    /* Which inflow frame a right-hand side at time t reads.  MGN_INFLOW_REFERENCE (0, the default of a zeroed descriptor): the
     * reference's own expression `floor(Int, t / saves_dt) + 1` (src/solve.jl:151) evaluated in the solver's time type with no
     * tolerance -- so a t an ulp below a frame boundary reads the previous frame, exactly as the reference does (in Float64,
     * 0.29 / 0.01 floors to 28) -- and a frame outside inflow_data is MGN_E_ARG (reference: BoundsError).  MGN_INFLOW_TOLERANT:
     * floor(t / saves_dt + 1e-3) (step k of a fixed-step solve reads frame k whatever its time type), clamped to the frames given.                                                              */
    int32_t inflow_rule;
    /* The solver's time type.  0: Float32, the type of the example's `0.0f0:0.01f0:5.99f0` (examples/cylinder_flow/cylinder_flow.jl:
     * 79-93): t0, t1, dt, saves_dt above are the times, and every time operation is rounded to float.  1: Float64: the four
     * *_f64 fields are the times (a Float64 0.01 is not a float), the float ones are ignored.  The integrator's time advances as
     * OrdinaryDiffEq's does: t <- t + dt per accepted step, the stop's own value when a step ends on a stop.                */
    int32_t time_f64;
    double t0_f64, t1_f64, dt_f64, saves_dt_f64;
} mgn_rollout_desc;
typedef enum mgn_inflow_rule { MGN_INFLOW_REFERENCE = 0, MGN_INFLOW_TOLERANT = 1 } mgn_inflow_rule;
int mgn_rollout(mgn_handle* h, mgn_rollout_desc* d);

/* ---- training step -------------------------------------------------------
 * GraphNetCore.step!(mgn, graph, target, mask, mse_reduce) as called at reference src/strategies.jl:418-422 and
 * consumed at src/MeshGraphNets.jl:370-378:  out = model(graph);  loss = mean(mse_reduce(target, out)[mask]) with
 * mse_reduce = sum of squared differences over the O rows per node;  gs = d loss / d ps.
 *   nf [N][Fn], ef [E][Fe]: the FeatureGraph (already normalised, as build_graph hands it over, src/graph.jl:75-97)
 *   target [N][O];  mask [nmask]: node indices (Int32, 1-based at the Julia boundary: mask_index_base = 1,
 *   src/MeshGraphNets.jl:352);  grads [n_grads = mgn_param_count]: packed order of mgn_set_params, so that
 *   Optimisers.update(opt_state, ps, gs) keeps working on the Julia side;  *loss: the scalar.
 * fp32, one partition; both ln_mode values and both ln_dims values; hidden_layers 1 .. 4; with two edge sets the second set's
 * features are the ones installed by mgn_set_edge_features (as in mgn_forward).  Memory: a large mesh stores the activations of as
 * many processor steps as the device's free memory (hipMemGetInfo) minus MGN_TRAIN_RESERVE_GB (16) holds and recomputes the others
 * in the reverse pass -- the first handle on a device takes the stored steps, a later one recomputes; a refused allocation is retried
 * with fewer stored steps, MGN_E_OOM only when the arena without any does not fit (M-1M: 61 GB + 10.7 GB per stored step).  Deterministic: gradients are reduced in a fixed order; the only atomic
 * (the scalar loss / per-node seed when `mask` lists a node twice) adds identical terms, so the order does not matter.
 * nf, ef, target and grads may be HOST or DEVICE pointers (copied with hipMemcpyDefault on the handle's stream): with
 * device arrays -- the reference keeps graph, ps and gs on the GPU, src/MeshGraphNets.jl:255-263 -- the optimiser update
 * runs where the gradients are and no 4 x param_count bytes cross PCIe per step.  mask is read on the host.        */
int mgn_step(mgn_handle* h, const float* nf, const float* ef, const float* target, const int32_t* mask, int64_t nmask,
             int32_t mask_index_base, float* grads, size_t n_grads, float* loss);

/* Vector-Jacobian product of the right-hand side f = mgn_ode_step (reference ode_step, src/solve.jl:188-219) for the
 * solver-based training strategies, where the adjoint of `solve` needs lambda^T df/dx and lambda^T df/dps per RHS
 * evaluation (ZygoteVJP inside the sensitivity algorithm, src/strategies.jl:175-196).  Inputs as mgn_ode_step (raw edge
 * features, frozen normalisers of mgn_set_norms); lambda [N][O]; outputs xbar [N][O], grads [n_grads] (packed order) and,
 * when dxdt != NULL, f(x) itself.  Every array argument may be a host or a device pointer (hipMemcpyDefault).       */
int mgn_ode_vjp(mgn_handle* h, const float* x, const float* node_type_onehot, const float* ef_raw, const float* val_mask,
                const float* lambda, float* dxdt, float* xbar, float* grads, size_t n_grads);

/* Pullback of mgn_forward, i.e. of the model call `output, st = mgn.model(graph, ps, mgn.st)` at reference src/solve.jl:200 -- what
 * Zygote needs from the shim when it differentiates ode_func_train / train_loss as written (src/strategies.jl:183-195: the
 * normalisers, build_graph, inverse_data and `.* val_mask` around the model stay Julia code and are differentiated there).
 * nf [N][Fn], ef [E][Fe]: the FeatureGraph as given to mgn_forward; ybar [N][O]: cotangent of the output; out [N][O] (may be NULL):
 * the output itself; nfbar [N][Fn] = ybar^T d out / d nf; grads [n_grads] = ybar^T d out / d ps (packed order).  The edge
 * features are constants of a trajectory (create_base_graph, src/graph.jl:25-55): no cotangent is formed for them.
 * Restrictions and pointer kinds as mgn_step.                                                                              */
int mgn_forward_vjp(mgn_handle* h, const float* nf, const float* ef, const float* ybar, float* out, float* nfbar, float* grads,
                    size_t n_grads);

/* ---- multi-GPU: the halo exchange lives INSIDE the library (SURVEY.md 8b "the engine owns ... RCCL communicators", 8e) ----
 * One process (or thread) per partition, one handle each (mgn_config.rank / nranks).  After mgn_comm_init every compute
 * entry point that documents it runs at nranks > 1 with no host involvement per step: mgn_processor_steps_dev drives, per
 * processor step, boundary projection -> pack -> sparse all-to-all-v (one grouped ncclSend / ncclRecv pair per neighbour, each
 * pair over its own xGMI link, on the communicator's stream) -> interior projection and interior edge tiles while the rows
 * are on the wire -> wait -> boundary edge tiles.  There is no precedent in the reference (single device,
 * src/MeshGraphNets.jl:255-263).
 *   transport MGN_COMM_RCCL: RCCL (needs one GPU per rank).
 *   transport MGN_COMM_HOST: POSIX shared memory on one node, rows staged through the host.  Lets several ranks share one GPU
 *     (tests), serves host-only handles, and is a fallback where RCCL cannot initialise; not the production wire.
 * Bootstrap: ONE rank calls mgn_comm_unique_id and the host distributes the MGN_COMM_ID_BYTES bytes (MPI.bcast, a
 * torch.distributed store, a file), then every rank calls mgn_comm_init with them (collective: returns when all ranks have
 * joined).  mgn_comm_init_file does the distribution through a file on a shared filesystem: rank 0 writes it, the others wait
 * for it.  Errors of this group: MGN_E_RCCL.                                                                          */
#define MGN_COMM_ID_BYTES 128
typedef enum mgn_comm_transport { MGN_COMM_RCCL = 0, MGN_COMM_HOST = 1 } mgn_comm_transport;
int mgn_comm_unique_id(void* id /* [MGN_COMM_ID_BYTES] */, int32_t transport);
int mgn_comm_init(mgn_handle* h, const void* id, size_t id_bytes, int32_t transport);
int mgn_comm_init_file(mgn_handle* h, const char* path, int32_t transport);
int mgn_comm_destroy(mgn_handle* h);            /* also done by mgn_destroy */
int mgn_comm_barrier(mgn_handle* h);            /* drains the handle's stream, then meets the peers */
/* x[n] (host) reduced over the ranks in place, op 0 = sum, 1 = max; the same bits on every rank (timing brackets, checksums) */
int mgn_comm_allreduce(mgn_handle* h, double* x, int32_t n, int32_t op);
/* One blocking halo exchange of the CURRENT P rows (what the staged driver does between steps); for hosts that drive the
 * fine-grained mgn_fwd_* / mgn_proc_* stages themselves. */
int mgn_halo_exchange(mgn_handle* h);
/* Any per-node rows on the HOST: own_rows [n_own][width] -> halo_rows [n_halo][width] (order of mgn_halo_nodes) through the
 * communicator; works on host-only handles (MGN_COMM_HOST transport).  E.g. positions of halo nodes. */
int mgn_halo_exchange_host(mgn_handle* h, const float* own_rows, float* halo_rows, int32_t width);

/* ---- the benchmarked unit: nsteps processor steps on given latents -------------
 * v [N][L], e [E][L] in caller order, updated in place (host buffers).                          */
int mgn_processor_steps(mgn_handle* h, float* v, float* e, int32_t nsteps);

/* Device-resident variant: latents live in the engine (import/randn them first); nothing crosses
 * PCIe.  Enqueues 2*nsteps kernels (+1 projection) on the stream; with nranks > 1 (after mgn_comm_init) every rank calls
 * it and the halo exchange between the steps runs inside (see the multi-GPU section).          */
int mgn_latents_import(mgn_handle* h, const float* v, const float* e);   /* host, GLOBAL arrays   */
int mgn_latents_export(mgn_handle* h, float* v, float* e);               /* host, owned rows only */
int mgn_latents_randn(mgn_handle* h, uint64_t seed); /* N(0,1) keyed by GLOBAL node/edge id      */
int mgn_latents_checksum(mgn_handle* h, double* sum_v, double* sum_e, double* sumsq_v, double* sumsq_e);
int mgn_processor_steps_dev(mgn_handle* h, int32_t nsteps);

/* ---- fine-grained stages (multi-partition driver; each enqueues on the stream) ----------------
 * Order per forward:   enc -> [halo] -> for k: edge(k) -> node(k) -> [halo] -> ... -> dec
 * Order per processor: begin -> [halo] -> for k: edge(k) -> node(k) -> [halo]
 * [halo] = mgn_halo_pack -> exchange rows between ranks (RCCL all-to-all-v) -> mgn_halo_unpack.    */
int mgn_fwd_upload(mgn_handle* h, const float* nf, const float* ef); /* host GLOBAL arrays -> device local */
int mgn_fwd_encode(mgn_handle* h);
int mgn_proc_begin(mgn_handle* h);                      /* project P,Q of step 0 from current v     */
int mgn_proc_edge(mgn_handle* h, int32_t k);
int mgn_proc_node(mgn_handle* h, int32_t k, int32_t project_next); /* project_next: also emit P,Q of step k+1 */
/* Split form for overlapping the halo exchange (owned nodes are numbered boundary-first):
 *   phase 1: node MLP of step k on all nodes + projection (P,Q of step k+1; k = -1: of step 0) of the BOUNDARY tiles
 *   -> mgn_halo_pack + start the exchange ->
 *   phase 2: projection of the interior tiles     -> finish the exchange, mgn_halo_unpack.
 * phase 1 followed by phase 2 equals mgn_proc_node(h, k, 1) (k >= 0) or mgn_proc_begin (k = -1). */
int mgn_proc_node_phase(mgn_handle* h, int32_t k, int32_t phase);
/* The edge step split the same way (SURVEY.md 8e "interior edges while the halo is in flight, boundary edges after"):
 * edges whose sender is a halo node all lie in the first `boundary` 32-edge tiles of the receiver-sorted list (mesh edges
 * are two-way, so they end at boundary nodes, which are numbered first).  phase 1: tiles [boundary, total) -- needs no
 * exchanged row, may run before mgn_halo_unpack; phase 2: tiles [0, boundary).  phase 1 + phase 2 == mgn_proc_edge.   */
int mgn_proc_edge_phase(mgn_handle* h, int32_t k, int32_t phase);
int mgn_edge_boundary_tiles(const mgn_handle* h, int32_t set, int32_t* boundary, int32_t* total);
int mgn_fwd_decode(mgn_handle* h);
int mgn_fwd_download(mgn_handle* h, float* out);        /* host GLOBAL [N][O]; owned rows written   */
int mgn_halo_bytes_per_row(const mgn_handle* h);
int mgn_halo_pack(mgn_handle* h, void* send_dev);       /* device buffer, sum(send_rows) rows       */
int mgn_halo_unpack(mgn_handle* h, const void* recv_dev); /* device buffer, sum(recv_rows) rows     */

/* ---- measurement hooks ------------------------------------------------------------------------ */
/* Average device time (HIP events on the launch stream) of each kernel family since the last reset:
 * ms[0]=edge step, ms[1]=node step, ms[2]=encode, ms[3]=decode, ms[4]=halo pack/unpack; counts alike. */
int mgn_profile_enable(mgn_handle* h, int32_t on);
int mgn_profile_read(mgn_handle* h, double ms_avg[8], int64_t counts[8]);   /* slots: edge, node, encode, decode, halo, boundary edge tiles */

/* ---- data formats: host code, no handle -------------------------------------------
 * TFRecord framing + tf.train.Example decoding of the DeepMind MeshGraphNets datasets, as the reference reads them
 * through TFRecord.jl (read(path; channel_size) at src/dataset.jl:107-112; one Example per trajectory, consumed by
 * parse_data src/dataset.jl:61-75).  mgn_tfrecord_next: 1 = a record is loaded, 0 = end of file, < 0 = error
 * (CRC mismatch, truncation, malformed protobuf; text from mgn_tfrecord_error).  mgn_tfrecord_feature hands out the
 * payload of feature `key` of the current record: kind 1 = bytes_list (first value, what parse_data reinterprets by
 * meta.json's dtype), 2 = float_list (float32 array), 3 = int64_list (int64 array); the pointer stays valid until the
 * next mgn_tfrecord_next / close.  Unknown key -> MGN_E_ARG (KeyError on the Julia side).                        */
typedef struct mgn_tfrecord mgn_tfrecord;
int mgn_tfrecord_open(const char* path, int32_t verify_crc, mgn_tfrecord** out);
int mgn_tfrecord_next(mgn_tfrecord* r);
int mgn_tfrecord_feature_count(const mgn_tfrecord* r);
const char* mgn_tfrecord_feature_name(const mgn_tfrecord* r, int32_t i);
int mgn_tfrecord_feature(const mgn_tfrecord* r, const char* key, int32_t* kind, const void** data, int64_t* nbytes);
const char* mgn_tfrecord_error(const mgn_tfrecord* r);
void mgn_tfrecord_close(mgn_tfrecord* r);
uint32_t mgn_crc32c(const void* data, size_t n);   /* CRC-32C (Castagnoli) as used by the record framing */

#ifdef __cplusplus
}
#endif
#endif /* MGN_HIP_H */
