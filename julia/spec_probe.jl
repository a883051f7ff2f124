# spec_probe.jl -- which MGN-spec variant do the INSTALLED GraphNetCore / Lux compute?   (no GPU, < 1 s)
#
#   julia --project=<the environment MeshGraphNets.jl runs in> julia/spec_probe.jl
#
# GraphNetCore 0.3 / Lux 0.5 are not vendored with the reference (Project.toml:11,15,36,40; Manifest git-ignored), so every
# choice DESIGN.md marks [GNC-unverified] is answered here, each with the engine / shim setting that matches:
#   1. Dense layers per MLP for hidden_layers = 2         -> mgn_config.hidden_layers
#   2. leaf order of the Lux Chain (weight, bias, scale, bias) -> pack_params(leaves) in julia/MGNHip.jl
#   3. LayerNorm: denominator form and the axes of its statistics -> mgn_config.ln_mode / mgn_config.ln_dims (kw ln_mode / ln_dims of MGNHip.load);
#      both are served by every entry point of the shim, training included (DESIGN.md section 2)
using GraphNetCore, Lux, Random, Statistics

rng = Random.Xoshiro(0)
L, hidden = 8, 2
mlp = GraphNetCore.build_mlp(3, L, hidden, L; layer_norm = true)      # the builder Encoder / Processor / Decoder use
ps, st = Lux.setup(rng, mlp)

dense = [l for l in Lux.Functors.fleaves(mlp) if l isa Lux.Dense]
println("1. Dense layers for hidden_layers = $hidden: ", length(dense),
        length(dense) == hidden + 1 ? "   -> mgn_config.hidden_layers = $hidden (MGN-spec v1)" :
                                      "   -> create the handle with hidden_layers = $(length(dense) - 1)")

println("2. parameter leaves in Chain order:")
for (k, v) in pairs(Lux.Functors.fmapstructure_with_path((p, x) -> size(x), ps))
    println("     ", k, " => ", v)
end
println("   -> enumerate them in THIS order into pack_params(leaves); expected: (weight, bias) per Dense, then (scale, bias)")

x = Float32[1 2 4; 0 1 9; -3 2 2; 5 5 7]                              # 4 features x 3 rows
ln = Lux.LayerNorm((4,)); pl, sl = Lux.setup(rng, ln)
y, _ = ln(x, pl, sl)
mu = mean(x; dims = 1); sd = std(x; dims = 1, corrected = false)
cand = Dict("per row, sqrt(var + eps)   -> ln_mode = 0 (MGN_LN_VAR_EPS), MGN-spec v1" => (x .- mu) ./ sqrt.(sd .^ 2 .+ 1f-5),
            "per row, sqrt(var) + eps   -> ln_mode = 1 (MGN_LN_STD_EPS)"              => (x .- mu) ./ (sd .+ 1f-5),
            "WHOLE ARRAY statistics, sqrt(var + eps) -> ln_dims = 1 (MGN_LN_ALL), ln_mode = 0" =>
                (x .- mean(x)) ./ sqrt(var(x; corrected = false) + 1f-5),
            "WHOLE ARRAY statistics, sqrt(var) + eps -> ln_dims = 1 (MGN_LN_ALL), ln_mode = 1   (LuxLib 0.5's layernorm at dims = Colon())" =>
                (x .- mean(x)) ./ (std(x; corrected = false) + 1f-5))
best = argmin(k -> maximum(abs.(cand[k] .- y)), collect(keys(cand)))
println("3. LayerNorm((4,)) on a 4 x 3 matrix matches: ", best, "   (max |diff| ", maximum(abs.(cand[best] .- y)), ")")
println("   eps of the layer: ", hasproperty(ln, :epsilon) ? ln.epsilon : "n/a", "  (engine: 1e-5)")
