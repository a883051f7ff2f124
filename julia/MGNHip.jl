# MGNHip.jl -- thin `ccall` shim that puts libmgn_hip.so behind the GraphNetCore surface
# una-auxme/MeshGraphNets.jl uses, so that src/graph.jl, src/solve.jl, src/strategies.jl and the update loop of
# src/MeshGraphNets.jl:375-377 keep working unchanged.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: julia is not available in the build environment.  What CAN be checked without
# Julia is checked: tests/test_julia_shim.py parses every `ccall` below and the two struct mirrors and compares symbol, argument
# count, C types and field order with include/mgn_hip.h.  The Python ctypes host (meshgraphnets.jl_amd/engine.py) mirrors this
# file call for call and is the tested twin.  See INTEGRATION.md for the diff against src/MeshGraphNets.jl.
#
# Usage inside MeshGraphNets.jl:   replace `using GraphNetCore` (src/MeshGraphNets.jl:8) by
#     include("MGNHip.jl"); using .MGNHip
# and nothing else in src/graph.jl / src/solve.jl / src/strategies.jl.  The names GraphNetCore keeps providing (normalisers,
# one_hot, triangles_to_edges, parse_edges, mse_reduce, inverse_data) are imported HERE, selectively, and re-exported: with both
# `using GraphNetCore` and `using .MGNHip` in the host module, `GraphNetwork`, `FeatureGraph`, `step!`, `load` and `save!` would be
# exported twice and `build_graph(mgn::GraphNetwork, ...)` (src/graph.jl:75) could not resolve its own signature.
module MGNHip

import GraphNetCore
import GraphNetCore: one_hot, triangles_to_edges, parse_edges, mse_reduce, inverse_data,
                     NormaliserOffline, NormaliserOfflineMinMax, NormaliserOfflineMeanStd, NormaliserOnline
import ChainRulesCore
import Serialization      # stdlib: normalisers + optimiser state of a checkpoint (load / save!)
import ChainRulesCore: NoTangent, ZeroTangent, Tangent

# what `using GraphNetCore` gave the reference, minus the five names replaced below
export one_hot, triangles_to_edges, parse_edges, mse_reduce, inverse_data
export NormaliserOffline, NormaliserOfflineMinMax, NormaliserOfflineMeanStd, NormaliserOnline
# the replaced surface
export FeatureGraph, GraphNetwork, step!, load, save!
# engine extras (optional fast paths; none is needed for the drop-in)
export set_trajectory_graph!, pack_params, init_params, set_norms!, freeze_norms!, set_static!, ode_step_resident, ode_step_fused,
       native_rollout, ode_vjp, forward_vjp, feature_stats
export comm_unique_id, comm_init!, comm_init_file!, comm_barrier, processor_steps_dev!

const LIB = get(ENV, "MGN_HIP_LIB", joinpath(@__DIR__, "..", "meshgraphnets.jl_amd", "lib", "libmgn_hip.so"))
const ABI_VERSION = 4       # MGN_ABI_VERSION of the include/mgn_hip.h the two struct mirrors below were written against

struct MgnConfig            # mirrors `mgn_config` (include/mgn_hip.h), field for field
    Fn::Int32
    Fe::Int32
    O::Int32
    L::Int32
    hidden_layers::Int32
    mps::Int32
    dtype::Int32
    rank::Int32
    nranks::Int32
    device::Int32
    n_edge_sets::Int32      # 1: the reference's single edge set (src/graph.jl:87-96)
    Fe2::Int32
    ln_mode::Int32          # 0: (x - mean) / sqrt(var + eps); 1: (x - mean) / (sqrt(var) + eps) -- see julia/spec_probe.jl
    ln_dims::Int32          # 0: LayerNorm statistics per node / edge; 1: over the whole array (Lux LayerNorm(shape) at dims = Colon())
end

mutable struct MgnRolloutDesc   # mirrors `mgn_rollout_desc` (include/mgn_hip.h), field for field
    solver::Int32
    t0::Float32
    t1::Float32
    dt::Float32
    saves_dt::Float32
    n_saves::Int32
    abstol::Float32
    reltol::Float32
    x0::Ptr{Float32}
    node_type_onehot::Ptr{Float32}
    ef_raw::Ptr{Float32}
    val_mask::Ptr{Float32}
    inflow_mask::Ptr{UInt8}
    inflow_data::Ptr{Float32}
    n_frames::Int32
    out::Ptr{Float32}
    n_accept::Int32
    n_reject::Int32
    n_rhs::Int32
    inflow_rule::Int32
    time_f64::Int32
    t0_f64::Float64
    t1_f64::Float64
    dt_f64::Float64
    saves_dt_f64::Float64
end

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:mgn_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc == -1 ? throw(ArgumentError(msg)) : error("mgn_hip ($rc): $msg")
end

"FeatureGraph(nf, ef, senders, receivers) -- same fields as GraphNetCore's (reference src/graph.jl:87-96)."
struct FeatureGraph{A <: AbstractMatrix{Float32}, B <: AbstractMatrix{Float32}, I <: AbstractVector{<:Integer}}
    nf::A          # (Fn x N)  == row-major [N][Fn]
    ef::B          # (Fe x E)
    senders::I     # 1-based
    receivers::I
end

"""
Mutable holder with the fields the reference reads and writes: `model`, `ps`, `st`, `e_norm`, `n_norm`, `o_norm`
(src/solve.jl:54,200-208; src/graph.jl:80-93; src/MeshGraphNets.jl:288,376-377).  `ps` is ONE packed Vector{Float32} in
MGN-spec order (pack_params): `Optimisers.setup(opt, mgn.ps)` / `Optimisers.update(opt_state, mgn.ps, gs[i])` work on it as on any
array; it is handed to the engine before every call (mgn_set_params recognises unchanged values: sync_params!).
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
    graph_key::Tuple{Int, Int, UInt, UInt}     # (E, N, hash(senders), hash(receivers)) of the graph the engine holds
end

# `rank` / `nranks`: this process's partition of an edge-cut mesh (one process per GPU; see comm_init!).  The reference
# itself is single-device (src/MeshGraphNets.jl:255-263).
# `ln_mode`, `ln_dims`: what julia/spec_probe.jl reports for the installed GraphNetCore / Lux (0 unless it says otherwise).
function GraphNetwork(quantities, dims, e_norm, n_norm, o_norm, outputs, mps, layer_size, hidden_layers, ps; device = -1,
        rank = 0, nranks = 1, ln_mode = 0, ln_dims = 0)
    v = ccall((:mgn_abi_version, LIB), Cint, ())
    v == ABI_VERSION || error("libmgn_hip.so has ABI version $v, this shim is written for $ABI_VERSION")
    cfg = MgnConfig(quantities, dims + 1, outputs, layer_size, hidden_layers, mps, 0, rank, nranks, device, 1, 0, ln_mode, ln_dims)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mgn_create, LIB), Cint, (Ref{MgnConfig}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || error(unsafe_string(ccall((:mgn_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    mgn = GraphNetwork(h[], cfg, identity, ps, NamedTuple(), e_norm, n_norm, o_norm, (-1, -1, UInt(0), UInt(0)))
    mgn.model = (graph, ps_, st) -> (forward(mgn, graph, ps_), st)      # mgn.model(graph, ps, st) -> (output, st)
    finalizer(m -> ccall((:mgn_destroy, LIB), Cvoid, (Ptr{Cvoid},), m.handle), mgn)
    return mgn
end

param_count(cfg::MgnConfig) = Int(ccall((:mgn_param_count, LIB), Csize_t, (Ref{MgnConfig},), cfg))

"""
Flatten a Lux parameter tree into MGN-spec packed order (include/mgn_hip.h): enc-node, enc-edge, step1-edge,
step1-node, ..., decoder; per MLP W1,b1,W2,b2,W3,b3,[ln_scale,ln_bias].  A Lux `Dense` weight is (out x in)
column-major, i.e. the row-major [in][out] block the engine expects: `vec(W)` is already the right bytes.
`leaves` must enumerate the (weight, bias, ...) arrays in that order for the concrete GraphNetCore model.
"""
pack_params(leaves) = reduce(vcat, (vec(Float32.(Array(x))) for x in leaves))

"""
Fresh parameters in packed order: Glorot-uniform weights, zero biases, LayerNorm scale 1 / bias 0 (Lux's defaults for `Dense` and
`LayerNorm`, which GraphNetCore's `build_mlp` uses [GNC-unverified]; the draws are Julia's, not GraphNetCore's).
"""
function init_params(cfg::MgnConfig; rng = nothing)
    r = rng === nothing ? () : (rng,)
    L, h = Int(cfg.L), Int(cfg.hidden_layers)
    ps = Float32[]
    function mlp!(fin, fout, ln)
        dims = vcat(fin, fill(L, h), fout)
        for i in 1:(length(dims) - 1)
            a = sqrt(6.0f0 / (dims[i] + dims[i + 1]))
            append!(ps, (rand(r..., Float32, dims[i] * dims[i + 1]) .* 2.0f0 .- 1.0f0) .* a)
            append!(ps, zeros(Float32, dims[i + 1]))
        end
        ln && (append!(ps, ones(Float32, fout)); append!(ps, zeros(Float32, fout)))
    end
    mlp!(Int(cfg.Fn), L, true)
    mlp!(Int(cfg.Fe), L, true)
    for _ in 1:cfg.mps
        mlp!(3L, L, true)
        mlp!(2L, L, true)
    end
    mlp!(L, Int(cfg.O), false)
    @assert length(ps) == param_count(cfg)
    return ps
end

# ---- load / save!: the checkpoint surface of src/MeshGraphNets.jl:282-285,460-471,537-540 ----------------------------------------
# What the unchanged call sites assume of GraphNetCore's pair (and what therefore round-trips here):
#   * `eval_network` builds FRESH normalisers with calc_norms and relies on `load` to bring back the trained ones (:529-540): a
#     cylinder_flow model normalises velocity (in and out) and the edge features with `NormaliserOnline` (:92,193-199), whose
#     statistics exist nowhere but in the training run that accumulated them;
#   * `train_network` resumes with `step = last(df_train.step)` (:324-325), already past `norm_steps`, so online normalisers never
#     re-accumulate, and it keeps the optimiser state `load` returns (:287-289: `Optimisers.setup` only when it is `nothing`).
# Four files in `path`: packed parameters (raw Float32), the loss log (CSV), the normalisers and the optimiser state (Julia's
# Serialization stdlib).  GraphNetCore's own JLD2 checkpoints are NOT read (different parameter tree, and JLD2 is not a dependency of
# this shim): a run started under GraphNetCore is continued by exporting its leaves once with `pack_params`.
"Loss log with the two columns the reference reads (`df_train.step`, `df_valid.loss`, src/MeshGraphNets.jl:324-330,383)."
mutable struct LossLog
    step::Vector{Int}
    loss::Vector{Float32}
end
LossLog() = LossLog(Int[], Float32[])

const CKPT_PARAMS = "mgn_hip_params.f32"      # packed parameters, raw little-endian Float32
const CKPT_LOG = "mgn_hip_log.csv"            # kind,step,loss
const CKPT_NORMS = "mgn_hip_norms.jls"        # snapshot((e_norm, n_norm, o_norm)): plain data, no device arrays, no closures
const CKPT_OPT = "mgn_hip_opt_state.jls"      # the Optimisers.jl state tree of the packed vector (host arrays), as it is
const CKPT_MANIFEST = "mgn_hip_manifest.txt"   # name,size,checksum of the four files above, written last (checkpoint.py's twin)

"""
`snapshot(x)`: a normaliser (or a Dict / tuple of them) as plain data -- arrays as host `Array`s, numbers and strings as they are,
structs as `(type = name, fields = Dict(field => snapshot))`, functions (the Lux device function a `NormaliserOnline` may keep)
dropped.  Written without naming any field of GraphNetCore's types [GNC-unverified]: whatever they hold is what is stored.
"""
snapshot(x::AbstractArray{<:Number}) = Array(x)
snapshot(x::AbstractArray) = map(snapshot, Array(x))
snapshot(x::Union{Number, AbstractString, Symbol, Nothing}) = x
snapshot(x::Function) = nothing
snapshot(x::AbstractDict) = Dict(k => snapshot(v) for (k, v) in x)
snapshot(x::Tuple) = map(snapshot, x)
snapshot(x::NamedTuple) = map(snapshot, x)
function snapshot(x)
    isstructtype(typeof(x)) || return x
    return (type = String(nameof(typeof(x))), fields = Dict(String(f) => snapshot(getfield(x, f)) for f in fieldnames(typeof(x))))
end

"""
`restore(template, snap)`: the stored statistics put over a freshly built normaliser of the same kind (the ones `calc_norms` hands
to `load`); returns the object to use.  Arrays are copied INTO the template's arrays (they stay on whatever device the template put
them: `copyto!` crosses), other fields are set on mutable structs; an immutable struct whose scalars differ is rebuilt through its
default constructor.  A kind mismatch (an online normaliser stored, an offline one passed) is an error, not a silent pick.
"""
restore(t::AbstractArray{<:Number}, s::AbstractArray{<:Number}) =
    size(t) == size(s) ? copyto!(t, s) : copyto!(similar(t, eltype(t), size(s)), s)
restore(t::Function, s) = t
restore(t::AbstractDict, s::AbstractDict) = (for (k, v) in s; t[k] = haskey(t, k) ? restore(t[k], v) : v; end; t)
restore(t::Tuple, s::Tuple) = map(restore, t, s)
function restore(t, s)
    (s isa NamedTuple && haskey(s, :type) && haskey(s, :fields)) || return s          # plain value: the stored one
    String(nameof(typeof(t))) == s.type ||
        throw(ArgumentError("checkpoint holds a $(s.type), load was handed a $(nameof(typeof(t))): build the normalisers as the training run did"))
    names = fieldnames(typeof(t))
    vals = [haskey(s.fields, String(f)) ? restore(getfield(t, f), s.fields[String(f)]) : getfield(t, f) for f in names]
    if ismutable(t)
        for (f, v) in zip(names, vals)
            v === getfield(t, f) || setfield!(t, f, convert(fieldtype(typeof(t), f), v))
        end
        return t
    end
    all(v === getfield(t, f) || v == getfield(t, f) for (f, v) in zip(names, vals)) && return t     # arrays were filled in place
    return typeof(t)(vals...)
end

"""
    load(quantities, dims, e_norms, n_norms, o_norms, outputs, mps, layer_size, hidden_layers, opt, device, path)
        -> (mgn, opt_state, df_train, df_valid)

Same call shape and return shape as GraphNetCore.load at src/MeshGraphNets.jl:282-285 and :537-540, so both call sites stay as they
are.  `device` (the reference's Lux device function) is accepted and ignored: the engine owns its GPU memory.  With a checkpoint
written by `save!` below in `path`, everything `save!` was given comes back: the parameters, the loss log, the NORMALISERS (their
stored statistics restored over `e_norms` / `n_norms` / `o_norms`, which `calc_norms` has just built empty: `eval_network` relies
on this, :529-540) and `opt_state` (so a resumed run keeps its Adam moments; `nothing` only when there is no checkpoint or `opt` is
`nothing`, which :287-289 turns into `Optimisers.setup(opt, mgn.ps)`).  A checkpoint that has parameters but no normaliser file (one
written before this was stored) is refused when any passed normaliser is a `NormaliserOnline`: it would evaluate with empty statistics.
"""
function load(quantities, dims, e_norms, n_norms, o_norms, outputs, mps, layer_size, hidden_layers, opt, device, path;
        hip_device = -1, ln_mode = 0, ln_dims = 0)
    df_train, df_valid = LossLog(), LossLog()
    opt_state = nothing
    pfile = joinpath(path, CKPT_PARAMS)
    have = isfile(pfile)
    if have
        mfile = joinpath(path, CKPT_MANIFEST)
        if isfile(mfile)               # (absent: a checkpoint written before the manifest existed -- taken as it is)
            for line in eachline(mfile)
                name, size, chk = split(line, ',')
                f = joinpath(path, name)
                (isfile(f) && filesize(f) == parse(Int, size) && file_checksum(f) == parse(UInt64, chk)) ||
                    error("checkpoint in $path is torn: $name is not the file its manifest lists (a run was killed inside save!; the directory mixes two saves)")
            end
        end
        nfile = joinpath(path, CKPT_NORMS)
        if isfile(nfile)
            stored = Serialization.deserialize(nfile)
            e_norms = restore(e_norms, stored.e_norm)
            n_norms = restore(n_norms, stored.n_norm)
            o_norms = restore(o_norms, stored.o_norm)
        else
            online(n) = n isa NormaliserOnline || (n isa AbstractDict && any(online, values(n)))
            (online(e_norms) || online(n_norms) || online(o_norms)) &&
                error("checkpoint in $path has no $CKPT_NORMS: its online normalisers' statistics were not saved and cannot be rebuilt")
        end
        ofile = joinpath(path, CKPT_OPT)
        (opt !== nothing && isfile(ofile)) && (opt_state = Serialization.deserialize(ofile))
        lfile = joinpath(path, CKPT_LOG)
        if isfile(lfile)
            for line in eachline(lfile)
                kind, step, loss = split(line, ',')
                log = kind == "train" ? df_train : df_valid
                push!(log.step, parse(Int, step)); push!(log.loss, parse(Float32, loss))
            end
        end
    end
    mgn = GraphNetwork(quantities, dims, e_norms, n_norms, o_norms, outputs, mps, layer_size, hidden_layers, nothing;
        device = hip_device, ln_mode = ln_mode, ln_dims = ln_dims)
    if have
        ps = Vector{Float32}(undef, param_count(mgn.cfg))
        filesize(pfile) == sizeof(ps) || error("checkpoint $pfile holds $(filesize(pfile)) bytes, this model has $(sizeof(ps))")
        read!(pfile, ps)
    else
        ps = init_params(mgn.cfg)
    end
    mgn.ps = ps
    return mgn, opt_state, df_train, df_valid
end

"""
`save!(mgn, opt_state, df_train, df_valid, step, loss, path; is_training = true)` as called at src/MeshGraphNets.jl:460-471: appends
(step, loss) to the training or the validation log and writes the whole state -- `mgn.ps`, `mgn.e_norm`, `mgn.n_norm`, `mgn.o_norm`
(with whatever an online normaliser has accumulated so far) and `opt_state`.  Every file is written beside its target and renamed, the manifest
(sizes and checksums of the four files) last: a run killed inside `save!` leaves either the previous checkpoint whole or a directory whose
manifest does not match, which `load` refuses instead of mixing the files of two saves.
"""
function save!(mgn::GraphNetwork, opt_state, df_train, df_valid, step, loss, path; is_training = true)
    mkpath(path)
    log = is_training ? df_train : df_valid
    push!(log.step, Int(step)); push!(log.loss, Float32(loss))
    function atomically(f, name)
        tmp = joinpath(path, name * ".tmp")
        f(tmp)
        mv(tmp, joinpath(path, name); force = true)
    end
    atomically(CKPT_NORMS) do tmp
        Serialization.serialize(tmp, (e_norm = snapshot(mgn.e_norm), n_norm = snapshot(mgn.n_norm), o_norm = snapshot(mgn.o_norm)))
    end
    atomically(CKPT_OPT) do tmp
        Serialization.serialize(tmp, opt_state)
    end
    atomically(CKPT_LOG) do tmp
        open(tmp, "w") do io
            for (kind, l) in (("train", df_train), ("valid", df_valid)), i in eachindex(l.step)
                println(io, kind, ',', l.step[i], ',', l.loss[i])
            end
        end
    end
    atomically(CKPT_PARAMS) do tmp
        write(tmp, mgn.ps::Vector{Float32})
    end
    atomically(CKPT_MANIFEST) do tmp            # last: sizes and checksums of the four -- `load` refuses a directory that mixes two saves
        open(tmp, "w") do io
            for name in (CKPT_NORMS, CKPT_OPT, CKPT_LOG, CKPT_PARAMS)
                f = joinpath(path, name)
                println(io, name, ',', filesize(f), ',', file_checksum(f))
            end
        end
    end
    return nothing
end

"sum over the file's bytes b_i (i from 1) of i * b_i, modulo 2^64 (checkpoint.py: file_checksum)"
function file_checksum(fname)
    s = UInt64(0)
    for (i, b) in enumerate(read(fname))
        s += UInt64(i) * UInt64(b)
    end
    return s
end

# ---- graph -----------------------------------------------------------------------------------------------------------------------
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
    pos = mesh_pos === nothing ? Ptr{Float32}(C_NULL) : pointer(mesh_pos)
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

# Called before every forward / step!: the optimiser updates `mgn.ps` in place (src/MeshGraphNets.jl:375-377), so the shim cannot know
# whether it changed.  mgn_set_params compares with what it holds and returns at once when nothing did (0.3 ms for 9 MB); when something
# did it only stores the vector (0.1 ms) -- the kernels' weight layouts are written on the device by the call that needs them.
function sync_params!(mgn::GraphNetwork, packed::Vector{Float32})
    check(mgn.handle, ccall((:mgn_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), mgn.handle, packed, length(packed)))
    return
end

# ---- the model: mgn.model(graph, ps, st) at src/solve.jl:200 ----------------------------------------------------------------------
"`ps` is the packed Vector{Float32} (see pack_params)."
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
Pullback of the model call: `(out, nfbar, gs) = forward_vjp(mgn, graph, ps, ybar)` with nfbar = ybar' d out / d nf (Fn x N) and
gs = ybar' d out / d ps (packed).  One forward with kept activations + one reverse pass on the device (mgn_forward_vjp).
"""
function forward_vjp(mgn::GraphNetwork, graph::FeatureGraph, ps::Vector{Float32}, ybar::AbstractMatrix)
    sync_params!(mgn, ps)
    N = size(graph.nf, 2)
    sync_graph!(mgn, graph, N)
    out = Matrix{Float32}(undef, mgn.cfg.O, N)
    nfbar = Matrix{Float32}(undef, mgn.cfg.Fn, N)
    gs = Vector{Float32}(undef, length(ps))
    nf = Array(graph.nf); ef = Array(graph.ef); yb = Matrix{Float32}(ybar)
    GC.@preserve nf ef yb out nfbar gs check(mgn.handle,
        ccall((:mgn_forward_vjp, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Csize_t),
            mgn.handle, nf, ef, yb, out, nfbar, gs, length(gs)))
    return out, nfbar, gs
end

# What Zygote sees when it differentiates ode_func_train / train_loss as written (src/strategies.jl:175-196): the normalisers,
# build_graph, inverse_data and `.* val_mask` around the model are Julia code and are differentiated by Zygote; the model itself is
# this rule.  The forward pass of the rule is the plain inference path; the reverse pass re-runs the forward with kept activations
# (mgn_forward_vjp) -- the sensitivity algorithm calls the pullback once per right-hand side it unwinds.
# The edge features are constants of a trajectory (create_base_graph, src/graph.jl:25-55): their cotangent is zero.
function ChainRulesCore.rrule(::typeof(forward), mgn::GraphNetwork, graph::FeatureGraph, ps::Vector{Float32})
    out = forward(mgn, graph, ps)
    function forward_pullback(ybar)
        _, nfbar, gs = forward_vjp(mgn, graph, ps, ChainRulesCore.unthunk(ybar))
        gbar = Tangent{typeof(graph)}(; nf = nfbar, ef = ZeroTangent(), senders = NoTangent(), receivers = NoTangent())
        return NoTangent(), NoTangent(), gbar, gs
    end
    return out, forward_pullback
end

# ---- step!: src/strategies.jl:418-422, consumed at src/MeshGraphNets.jl:370-378 ----------------------------------------------------
"""
`step!(mgn, graph, target, mask, loss_function)`: returns `(gs, loss)` with `loss = mean(mse_reduce(target, output)[mask])`.
`gs` is a ONE-ELEMENT TUPLE holding the packed gradient (the order of `pack_params`), so the caller's loop
`for i in eachindex(gs); opt_state, ps = Optimisers.update(opt_state, mgn.ps, gs[i]); mgn.ps = ps; end`
(src/MeshGraphNets.jl:375-377) runs unchanged: one `Optimisers.update` on the packed vector.
`mask` are the Int32 node indices built at src/MeshGraphNets.jl:352 (1-based).  Only `mse_reduce` runs on the device.
`mgn_step` copies with hipMemcpyDefault: with AMDGPU.jl arrays pass `pointer(graph.nf)` etc. of the ROCArrays and a
`ROCVector{Float32}` for `gs` instead of the host copies made below, and the gradients never cross PCIe (the
reference keeps graph, ps and gs on the GPU, src/MeshGraphNets.jl:255-263); `mask` stays a host vector.
"""
function step!(mgn::GraphNetwork, graph::FeatureGraph, target::AbstractMatrix, mask::AbstractVector{<:Integer}, loss_function = nothing)
    ps = mgn.ps::Vector{Float32}
    sync_params!(mgn, ps)
    N = size(graph.nf, 2)
    sync_graph!(mgn, graph, N)
    gs = Vector{Float32}(undef, length(ps))
    loss = Ref{Float32}(0)
    nf = Array(graph.nf); ef = Array(graph.ef); tg = Matrix{Float32}(Array(target)); mk = Vector{Int32}(Array(mask))
    GC.@preserve nf ef tg mk gs check(mgn.handle,
        ccall((:mgn_step, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Int64, Int32, Ptr{Float32}, Csize_t, Ref{Float32}),
            mgn.handle, nf, ef, tg, mk, length(mk), 1, gs, length(gs), loss))
    return (gs,), loss[]
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

# ---- fused right-hand side and native rollout (optional fast paths) ---------------------------------------------------------------
opt_ptr(x::Nothing) = Ptr{Float32}(C_NULL)
opt_ptr(x::Array{Float32}) = pointer(x)

"""
Frozen normalisers as per-feature affine maps (mgn_set_norms): forward `y = x * scale + shift` for node and edge features,
`inverse_data(o_norm, y) = y * out_scale + out_shift` for the outputs.  `nothing` = identity.  Needed by ode_step_fused,
ode_step_resident, native_rollout and ode_vjp -- not by `mgn.model` / `step!`, whose FeatureGraph arrives normalised.
"""
function set_norms!(mgn::GraphNetwork; node_scale = nothing, node_shift = nothing, edge_scale = nothing, edge_shift = nothing,
        out_scale = nothing, out_shift = nothing)
    GC.@preserve node_scale node_shift edge_scale edge_shift out_scale out_shift check(mgn.handle,
        ccall((:mgn_set_norms, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            mgn.handle, opt_ptr(node_scale), opt_ptr(node_shift), opt_ptr(edge_scale), opt_ptr(edge_shift), opt_ptr(out_scale), opt_ptr(out_shift)))
    return mgn
end

"""
`freeze_norms!(mgn, fields, target_fields, dims)`: derive the affine maps from the normaliser OBJECTS the reference built
(src/MeshGraphNets.jl:79-203) by evaluating each on a column of zeros and a column of ones -- every GraphNetCore normaliser is
affine once frozen, so scale = n(1) - n(0), shift = n(0), whatever its fields are called -- and install them (set_norms!).
Call it after the online normalisers have stopped accumulating (`norm_steps` passed; evaluation runs: always), since calling an
accumulating NormaliserOnline feeds it the probe.  `fields`: node fields in build_graph's order (src/graph.jl:80-86; `node_type`
is appended last there and here); `dims[f]`: rows of field f.
"""
function freeze_norms!(mgn::GraphNetwork, fields, target_fields, dims::AbstractDict)
    probe(n, d) = (z = vec(Float32.(Array(n(zeros(Float32, d, 1))))); o = vec(Float32.(Array(n(ones(Float32, d, 1))))); (o .- z, z))
    ns, nsh = Float32[], Float32[]
    for f in vcat(fields, "node_type")
        s, sh = probe(mgn.n_norm[f], dims[f]); append!(ns, s); append!(nsh, sh)
    end
    es, esh = probe(mgn.e_norm, Int(mgn.cfg.Fe))
    os, osh = Float32[], Float32[]
    for f in target_fields      # inverse_data(o_norm, y) = y * scale + shift, probed the same way
        inv(y) = inverse_data(mgn.o_norm[f], y)
        s, sh = probe(inv, dims[f]); append!(os, s); append!(osh, sh)
    end
    return set_norms!(mgn; node_scale = ns, node_shift = nsh, edge_scale = es, edge_shift = esh, out_scale = os, out_shift = osh)
end

"""
Once per trajectory (where `create_base_graph` returns, src/MeshGraphNets.jl:360,418,596): make the static RHS
inputs device-resident and run the edge encoder once.  Afterwards `ode_step_resident(mgn, x)` moves only the state.
"""
function set_static!(mgn::GraphNetwork, node_type_onehot::Matrix{Float32}, edge_features::Matrix{Float32},
        val_mask_row::Union{Nothing, Vector{Float32}} = nothing)
    GC.@preserve node_type_onehot edge_features val_mask_row check(mgn.handle,
        ccall((:mgn_set_static, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            mgn.handle, node_type_onehot, edge_features, opt_ptr(val_mask_row)))
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
maps (set_norms! / freeze_norms!).
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

"""
`lambda' df/dx` and `lambda' df/dps` of the fused right-hand side f = ode_step_fused (mgn_ode_vjp): the hand-written form of what
the rrule above gives Zygote piecewise, for callers that write their own adjoint of `solve`.  Returns (xbar, gs, f(x)).
"""
function ode_vjp(mgn::GraphNetwork, x::Matrix{Float32}, node_type_onehot::Matrix{Float32}, edge_features::Matrix{Float32},
        val_mask_row::Union{Nothing, Vector{Float32}}, lambda::Matrix{Float32})
    ps = mgn.ps::Vector{Float32}
    sync_params!(mgn, ps)
    dxdt = similar(x); xbar = similar(x); gs = Vector{Float32}(undef, length(ps))
    GC.@preserve x node_type_onehot edge_features val_mask_row lambda dxdt xbar gs check(mgn.handle,
        ccall((:mgn_ode_vjp, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Csize_t),
            mgn.handle, x, node_type_onehot, edge_features, opt_ptr(val_mask_row), lambda, dxdt, xbar, gs, length(gs)))
    return xbar, gs, dxdt
end

"""
    native_rollout(solver, mgn, x0, node_type_onehot, edge_features, val_mask_row, inflow_mask_row, inflow_data, start, stop, dt, saves)
        -> (sol_u::Vector{Matrix{Float32}}, sol_t)

`rollout` of src/solve.jl:42-68 on the device (mgn_rollout): the ODEProblem over `ode_func_eval` solved with fixed-step Euler
(`dt` given: `solve(prob, solver; adaptive = false, dt = dt, saveat = saves)`) or adaptive Tsit5 (`dt === nothing`:
`solve(prob, solver; saveat = saves, tstops = saves)`), no host round trip per right-hand side.  `solver` is `:Euler`, `:Tsit5` or
an OrdinaryDiffEq algorithm object of those names.  The inflow frame of a right-hand side is the reference's own
`floor(Int, t / saves_dt) + 1` in the element type of `saves` (src/solve.jl:151; MGN_INFLOW_REFERENCE), `inflow_data` being
(O x N x frames).  Needs frozen normalisers (freeze_norms!) and the trajectory's graph (set_trajectory_graph!).
"""
function native_rollout(solver, mgn::GraphNetwork, x0::Matrix{Float32}, node_type_onehot::Matrix{Float32}, edge_features::Matrix{Float32},
        val_mask_row::Union{Nothing, Vector{Float32}}, inflow_mask_row::Union{Nothing, Vector{UInt8}},
        inflow_data::Union{Nothing, Array{Float32, 3}}, start, stop, dt, saves; abstol = 1.0f-6, reltol = 1.0f-3, tolerant_inflow = false)
    name = solver isa Symbol ? solver : nameof(typeof(solver))
    name in (:Euler, :Tsit5) || throw(ArgumentError("native_rollout drives Euler and Tsit5; got $name"))
    sync_params!(mgn, mgn.ps::Vector{Float32})
    O, N = size(x0)
    ns = length(saves)
    out = Array{Float32, 3}(undef, O, N, ns)
    sdt = ns > 1 ? saves[2] - saves[1] : one(eltype(saves))
    f64 = eltype(saves) == Float64
    d = MgnRolloutDesc(name == :Euler ? 0 : 1, start, stop, dt === nothing ? 0 : dt, sdt, ns, abstol, reltol,
        pointer(x0), pointer(node_type_onehot), pointer(edge_features), opt_ptr(val_mask_row),
        inflow_mask_row === nothing ? Ptr{UInt8}(C_NULL) : pointer(inflow_mask_row), inflow_data === nothing ? Ptr{Float32}(C_NULL) : pointer(inflow_data),
        inflow_data === nothing ? 0 : size(inflow_data, 3), pointer(out), 0, 0, 0, tolerant_inflow ? 1 : 0, f64 ? 1 : 0,
        start, stop, dt === nothing ? 0 : dt, sdt)
    GC.@preserve x0 node_type_onehot edge_features val_mask_row inflow_mask_row inflow_data out check(mgn.handle,
        ccall((:mgn_rollout, LIB), Cint, (Ptr{Cvoid}, Ref{MgnRolloutDesc}), mgn.handle, d))
    return [out[:, :, i] for i in 1:ns], collect(saves)
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
