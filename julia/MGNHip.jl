# MGNHip.jl -- thin `ccall` shim that puts libmgn_hip.so behind the GraphNetCore surface
# una-auxme/MeshGraphNets.jl uses, so that src/graph.jl and src/solve.jl keep working unchanged.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: julia is not available in the build environment.  The shim binds
# exactly the symbols of include/mgn_hip.h and mirrors meshgraphnets.jl_amd/engine.py call for call; the
# Python ctypes host is the tested twin.  See INTEGRATION.md.
#
# Usage inside MeshGraphNets.jl:   replace `using GraphNetCore` (src/MeshGraphNets.jl:8) by
#     include("MGNHip.jl"); using .MGNHip
# Normalisers, one_hot, triangles_to_edges, parse_edges, mse_reduce, save!/load stay GraphNetCore's (host
# side, cheap); only GraphNetwork.model / FeatureGraph / step! are replaced.
module MGNHip

export FeatureGraph, GraphNetwork, set_trajectory_graph!, pack_params, set_static!, ode_step_resident, ode_step_fused, step!, feature_stats
export comm_unique_id, comm_init!, comm_init_file!, comm_barrier, processor_steps_dev!

const LIB = get(ENV, "MGN_HIP_LIB", joinpath(@__DIR__, "..", "meshgraphnets.jl_amd", "lib", "libmgn_hip.so"))

struct MgnConfig            # mirrors `mgn_config` (include/mgn_hip.h)
    Fn::Int32; Fe::Int32; O::Int32; L::Int32; hidden_layers::Int32; mps::Int32
    dtype::Int32; rank::Int32; nranks::Int32; device::Int32
    n_edge_sets::Int32; Fe2::Int32      # 1, 0: the reference's single edge set (src/graph.jl:87-96)
    ln_mode::Int32                      # 0: (x - mean) / sqrt(var + eps); 1: (x - mean) / (sqrt(var) + eps) -- see julia/spec_probe.jl
end

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:mgn_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc == -1 ? throw(ArgumentError(msg)) : error("mgn_hip ($rc): $msg")
end

"FeatureGraph(nf, ef, senders, receivers) -- same fields as GraphNetCore's (reference src/graph.jl:87-96)."
struct FeatureGraph{A <: AbstractMatrix{Float32}, I <: AbstractVector{<:Integer}}
    nf::A          # (Fn x N)  == row-major [N][Fn]
    ef::A          # (Fe x E)
    senders::I     # 1-based
    receivers::I
end

"""
Mutable holder with the fields the reference reads and writes: `model`, `ps`, `st`, `e_norm`, `n_norm`, `o_norm`
(src/solve.jl:54,200-208; src/graph.jl:80-93; src/MeshGraphNets.jl:288,376-377).  `ps` stays a Julia-owned
array tree so `Optimisers.update(opt_state, mgn.ps, gs)` keeps working; it is flattened (pack_params) and
uploaded only when it changed.
"""
mutable struct GraphNetwork
    handle::Ptr{Cvoid}
    cfg::MgnConfig
    model::Function
    ps
    st
    e_norm
    n_norm
    o_norm
    ps_hash::UInt
    graph_key::Tuple{Int, Int, UInt, UInt}     # (E, N, hash(senders), hash(receivers)) of the graph the engine holds
end

# `rank` / `nranks`: this process's partition of an edge-cut mesh (one process per GPU; see comm_init!).  The reference
# itself is single-device (src/MeshGraphNets.jl:255-263).
# `ln_mode`: what julia/spec_probe.jl reports for the installed GraphNetCore / Lux (0 unless it says otherwise).
function GraphNetwork(quantities, dims, e_norm, n_norm, o_norm, outputs, mps, layer_size, hidden_layers, ps; device = -1,
        rank = 0, nranks = 1, ln_mode = 0)
    cfg = MgnConfig(quantities, dims + 1, outputs, layer_size, hidden_layers, mps, 0, rank, nranks, device, 1, 0, ln_mode)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mgn_create, LIB), Cint, (Ref{MgnConfig}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || error(unsafe_string(ccall((:mgn_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    mgn = GraphNetwork(h[], cfg, identity, ps, NamedTuple(), e_norm, n_norm, o_norm, UInt(0), (-1, -1, UInt(0), UInt(0)))
    mgn.model = (graph, ps_, st) -> (forward(mgn, graph, ps_), st)      # mgn.model(graph, ps, st) -> (output, st)
    finalizer(m -> ccall((:mgn_destroy, LIB), Cvoid, (Ptr{Cvoid},), m.handle), mgn)
    return mgn
end

"""
Flatten a Lux parameter tree into MGN-spec packed order (include/mgn_hip.h): enc-node, enc-edge, step1-edge,
step1-node, ..., decoder; per MLP W1,b1,W2,b2,W3,b3,[ln_scale,ln_bias].  A Lux `Dense` weight is (out x in)
column-major, i.e. the row-major [in][out] block the engine expects: `vec(W)` is already the right bytes.
`leaves` must enumerate the (weight, bias, ...) arrays in that order for the concrete GraphNetCore model.
"""
pack_params(leaves) = reduce(vcat, (vec(Float32.(Array(x))) for x in leaves))

"Once per trajectory, where the reference calls create_base_graph (src/MeshGraphNets.jl:360,418,596)."
# Content key of a topology.  The key is taken from the CALLER's arrays (whatever their element type) and by content: an
# `objectid` would name the Int32 copy made below -- a cache that never hits and re-runs mgn_set_graph (receiver sort, CSR,
# buffer reallocation) for every ODE right-hand side -- and, being address based, can match a different array after GC.  The
# reference also mutates `senders` in place while it builds the graph (`senders .+= 1`, src/graph.jl:32): a content hash sees that.
graph_key(senders, receivers, N) = (length(senders), Int(N), hash(senders), hash(receivers))

function set_trajectory_graph!(mgn::GraphNetwork, senders::AbstractVector{<:Integer}, receivers::AbstractVector{<:Integer}, N::Integer;
        mesh_pos::Union{Nothing, Matrix{Float32}} = nothing)
    key = graph_key(senders, receivers, N)
    senders = senders isa Vector{Int32} ? senders : Vector{Int32}(senders)        # convert only when needed
    receivers = receivers isa Vector{Int32} ? receivers : Vector{Int32}(receivers)
    pos = mesh_pos === nothing ? C_NULL : pointer(mesh_pos)
    pd = mesh_pos === nothing ? 0 : size(mesh_pos, 1)
    GC.@preserve senders receivers mesh_pos check(mgn.handle,
        ccall((:mgn_set_graph, LIB), Cint,
            (Ptr{Cvoid}, Int32, Int64, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Float32}, Int32),
            mgn.handle, N, length(senders), senders, receivers, 1 #= Julia indices, src/graph.jl:31-34 =#, pos, pd))
    mgn.graph_key = key
    return mgn
end

sync_graph!(mgn::GraphNetwork, graph, N) =
    graph_key(graph.senders, graph.receivers, N) == mgn.graph_key || set_trajectory_graph!(mgn, graph.senders, graph.receivers, N)

function sync_params!(mgn::GraphNetwork, packed::Vector{Float32})
    hsh = hash(packed)
    hsh == mgn.ps_hash && return
    check(mgn.handle, ccall((:mgn_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), mgn.handle, packed, length(packed)))
    mgn.ps_hash = hsh
end

"`mgn.model(graph, ps, st)` at src/solve.jl:200.  `ps` is the packed Vector{Float32} (see pack_params)."
function forward(mgn::GraphNetwork, graph::FeatureGraph, ps::Vector{Float32})
    sync_params!(mgn, ps)
    N = size(graph.nf, 2)
    sync_graph!(mgn, graph, N)
    out = Matrix{Float32}(undef, mgn.cfg.O, N)
    nf = Array(graph.nf); ef = Array(graph.ef)             # host arrays: the engine copies in (H2D) itself
    GC.@preserve nf ef out check(mgn.handle,
        ccall((:mgn_forward, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), mgn.handle, nf, ef, out))
    return out
end

"""
`step!(mgn, graph, target, mask, loss_function)` as called at src/strategies.jl:418-422: returns `(gs, loss)` with
`loss = mean(mse_reduce(target, output)[mask])`.  `gs` is ONE packed Vector{Float32} in the order of `pack_params`, so
the caller's loop `for i in eachindex(gs); opt_state, ps = Optimisers.update(opt_state, ps, gs[i]); end`
(src/MeshGraphNets.jl:375-377) becomes a single `Optimisers.update(opt_state, mgn.ps, gs)` on the packed vector.
`mask` are the Int32 node indices built at src/MeshGraphNets.jl:352 (1-based).  Only `mse_reduce` runs on the device.
`mgn_step` copies with hipMemcpyDefault: with AMDGPU.jl arrays pass `pointer(graph.nf)` etc. of the ROCArrays and a
`ROCVector{Float32}` for `gs` instead of the host copies made below, and the gradients never cross PCIe (the
reference keeps graph, ps and gs on the GPU, src/MeshGraphNets.jl:255-263); `mask` stays a host vector.
"""
function step!(mgn::GraphNetwork, graph::FeatureGraph, target::Matrix{Float32}, mask::Vector{Int32}, loss_function = nothing)
    ps = mgn.ps::Vector{Float32}
    sync_params!(mgn, ps)
    N = size(graph.nf, 2)
    sync_graph!(mgn, graph, N)
    gs = Vector{Float32}(undef, length(ps))
    loss = Ref{Float32}(0)
    nf = Array(graph.nf); ef = Array(graph.ef)
    GC.@preserve nf ef target mask gs check(mgn.handle,
        ccall((:mgn_step, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Int64, Int32, Ptr{Float32}, Csize_t, Ref{Float32}),
            mgn.handle, nf, ef, target, mask, length(mask), 1, gs, length(gs), loss))
    return gs, loss[]
end

"Per-feature (sum, sum of squares) in Float64 of `x` (dim x rows): one accumulation step of a `NormaliserOnline`, on the device."
function feature_stats(mgn::GraphNetwork, x::Matrix{Float32})
    dim, rows = size(x)
    s = zeros(Float64, dim); q = zeros(Float64, dim)
    GC.@preserve x s q check(mgn.handle,
        ccall((:mgn_feature_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Ptr{Float64}, Ptr{Float64}),
            mgn.handle, x, rows, dim, s, q))
    return s, q
end

"""
Once per trajectory (where `create_base_graph` returns, src/MeshGraphNets.jl:360,418,596): make the static RHS
inputs device-resident and run the edge encoder once.  Afterwards `ode_step_resident(mgn, x)` moves only the state.
"""
function set_static!(mgn::GraphNetwork, node_type_onehot::Matrix{Float32}, edge_features::Matrix{Float32},
        val_mask_row::Union{Nothing, Vector{Float32}} = nothing)
    vm = val_mask_row === nothing ? C_NULL : pointer(val_mask_row)
    GC.@preserve node_type_onehot edge_features val_mask_row check(mgn.handle,
        ccall((:mgn_set_static, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            mgn.handle, node_type_onehot, edge_features, vm))
    return mgn
end

function ode_step_resident(mgn::GraphNetwork, x::Matrix{Float32})
    out = similar(x)
    GC.@preserve x out check(mgn.handle,
        ccall((:mgn_ode_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            mgn.handle, x, C_NULL, C_NULL, C_NULL, out))
    return out
end

"""
Fused right-hand side: everything `ode_step` does after the state split (src/solve.jl:198-218) in one call --
build_graph normalisation, model, inverse_data, `.* val_mask`.  Normalisers must have been frozen into affine
maps with `mgn_set_norms` (a NormaliserOnline past `max_acc`, or any offline normaliser).
"""
function ode_step_fused(mgn::GraphNetwork, x::Matrix{Float32}, node_type_onehot::Matrix{Float32},
        edge_features::Matrix{Float32}, val_mask_row::Vector{Float32})
    out = similar(x)
    GC.@preserve x node_type_onehot edge_features val_mask_row out check(mgn.handle,
        ccall((:mgn_ode_step, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            mgn.handle, x, node_type_onehot, edge_features, val_mask_row, out))
    return out
end

# ---- multi-GPU: one Julia process per GPU (e.g. under MPI.jl or Distributed), one handle each; the halo exchange (RCCL grouped
# send / recv over xGMI) and the overlap schedule run inside the library.  With `nranks > 1` and a communicator,
# `mgn.model(graph, ps, st)` takes the GLOBAL FeatureGraph on every rank and returns the complete output on every rank.
const COMM_ID_BYTES = 128
const COMM_RCCL = Int32(0)
const COMM_HOST = Int32(1)      # shared memory on one node (several ranks on one GPU; tests)

"Made on ONE rank; distribute the bytes to the others (`MPI.Bcast!(id, 0, comm)`), then every rank calls `comm_init!`."
function comm_unique_id(transport::Int32 = COMM_RCCL)
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    rc = ccall((:mgn_comm_unique_id, LIB), Cint, (Ptr{UInt8}, Int32), id, transport)
    rc == 0 || error(unsafe_string(ccall((:mgn_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return id
end

function comm_init!(mgn::GraphNetwork, id::Vector{UInt8}, transport::Int32 = COMM_RCCL)
    GC.@preserve id check(mgn.handle,
        ccall((:mgn_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t, Int32), mgn.handle, id, length(id), transport))
    return mgn
end

"Bootstrap without MPI: rank 0 writes the id to `path` (a filesystem every rank sees), the others wait for it."
comm_init_file!(mgn::GraphNetwork, path::AbstractString, transport::Int32 = COMM_RCCL) =
    (check(mgn.handle, ccall((:mgn_comm_init_file, LIB), Cint, (Ptr{Cvoid}, Cstring, Int32), mgn.handle, path, transport)); mgn)

comm_barrier(mgn::GraphNetwork) = check(mgn.handle, ccall((:mgn_comm_barrier, LIB), Cint, (Ptr{Cvoid},), mgn.handle))

"`nsteps` processor steps on the engine-resident latents (the benchmarked unit); every rank calls it."
processor_steps_dev!(mgn::GraphNetwork, nsteps::Integer) =
    check(mgn.handle, ccall((:mgn_processor_steps_dev, LIB), Cint, (Ptr{Cvoid}, Int32), mgn.handle, nsteps))

end # module
