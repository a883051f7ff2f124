# write_trajectories.jl -- turn the engine's raw rollout dump into the reference's evaluation output
#
#   julia julia/write_trajectories.jl <dump_dir> <out_dir> [solver_name]
#
# The reference writes `<out>/<solver>/trajectories.h5` with, per trajectory i (1-based) and name in
# ("mesh_pos", "gt", "prediction", "error", "timesteps", "cells"), a group `/<i>/<name>` holding `data` (the array flattened in
# Julia's column-major order) and `size` (its dimensions): src/MeshGraphNets.jl:638-669.  The engine side writes that file itself where
# libhdf5 is present (`dataset_h5.write_trajectories_h5`); on a host without it `reference_api.dump_rollout` writes, per
# trajectory, `<dump_dir>/<i>/<name>.bin` (little-endian, the bytes of the Julia array: feature-major, then node, then time) and
# `<dump_dir>/<i>/manifest.json` ({name: {"dtype": "Float32" | "Int32", "size": [..]}}) -- and this script, run where HDF5.jl exists,
# produces the file the reference's plotting / comparison tools read.  Nothing else of eval_network! is replaced.
using HDF5, JSON

function main(dump_dir, out_dir, solver = "derivative_training")
    eval_path = joinpath(out_dir, lowercase(solver))
    mkpath(eval_path)
    trajs = sort(parse.(Int, filter(d -> isdir(joinpath(dump_dir, d)) && all(isdigit, d), readdir(dump_dir))))
    h5open(joinpath(eval_path, "trajectories.h5"), "w") do f
        for i in trajs
            g = create_group(f, string(i))
            manifest = JSON.parsefile(joinpath(dump_dir, string(i), "manifest.json"))
            for (name, desc) in manifest
                T = desc["dtype"] == "Int32" ? Int32 : Float32
                dims = Tuple(Int.(desc["size"]))
                value = Array{T}(undef, dims...)
                read!(joinpath(dump_dir, string(i), name * ".bin"), value)
                sub_g = create_group(g, name)
                sub_g["data"] = reshape(value, length(value))      # as src/MeshGraphNets.jl:647-648
                sub_g["size"] = collect(size(value))
            end
        end
    end
    @info "wrote $(joinpath(eval_path, "trajectories.h5")) ($(length(trajs)) trajectories)"
end

length(ARGS) >= 2 || error("usage: julia write_trajectories.jl <dump_dir> <out_dir> [solver_name]")
main(ARGS...)
