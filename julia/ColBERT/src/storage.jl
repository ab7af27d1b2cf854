# The index directory, in the reference's layout (src/savers.jl, src/loaders.jl, src/indexing.jl:82-85,140-143):
# config.json, plan.json, <i>.metadata.json and one JLD2 file per array, written with JLD2.save_object.

_json_write(path::String, obj) = open(io -> JSON.print(io, obj, 4), path, "w")

"config.json: every field of the config (savers.jl:110-121)"
function save(config::ColBERTConfig)
    isdir(config.index_path) || mkpath(config.index_path)
    d = Dict{String, Any}(string(f) => getfield(config, f) for f in fieldnames(ColBERTConfig))
    _json_write(joinpath(config.index_path, "config.json"), d)
end

"config.json -> ColBERTConfig (loaders.jl:66-74), without evaluating any text: fields are converted one by one"
function load_config(index_path::String)
    raw = JSON.parsefile(joinpath(index_path, "config.json"))
    kw = Dict{Symbol, Any}()
    for (f, T) in zip(fieldnames(ColBERTConfig), fieldtypes(ColBERTConfig))
        haskey(raw, string(f)) || continue
        v = raw[string(f)]
        if f == :chunksize
            kw[f] = v === nothing ? missing : Int(v)
        elseif f == :collection
            kw[f] = v isa AbstractVector ? String.(v) : String(v)
        elseif T === Int
            kw[f] = Int(v)
        elseif T === Bool
            kw[f] = Bool(v)
        else
            kw[f] = String(v)
        end
    end
    ColBERTConfig(; kw...)
end

function save_codec(index_path::String, centroids::Matrix{Float32}, bucket_cutoffs::Vector{Float32},
        bucket_weights::Vector{Float32}, avg_residual::Float32)
    JLD2.save_object(joinpath(index_path, "centroids.jld2"), centroids)
    JLD2.save_object(joinpath(index_path, "avg_residual.jld2"), avg_residual)
    JLD2.save_object(joinpath(index_path, "bucket_cutoffs.jld2"), bucket_cutoffs)
    JLD2.save_object(joinpath(index_path, "bucket_weights.jld2"), bucket_weights)
end

function load_codec(index_path::String)
    centroids = JLD2.load_object(joinpath(index_path, "centroids.jld2"))
    avg_residual = JLD2.load_object(joinpath(index_path, "avg_residual.jld2"))
    bucket_cutoffs = JLD2.load_object(joinpath(index_path, "bucket_cutoffs.jld2"))
    bucket_weights = JLD2.load_object(joinpath(index_path, "bucket_weights.jld2"))
    centroids isa Matrix{Float32} || error("centroids.jld2 must hold a Matrix{Float32}")
    avg_residual isa Float32 || error("avg_residual.jld2 must hold a Float32")
    bucket_cutoffs isa Vector{Float32} || error("bucket_cutoffs.jld2 must hold a Vector{Float32}")
    bucket_weights isa Vector{Float32} || error("bucket_weights.jld2 must hold a Vector{Float32}")
    Dict("centroids" => centroids, "avg_residual" => avg_residual, "bucket_cutoffs" => bucket_cutoffs,
        "bucket_weights" => bucket_weights)
end

"one chunk: <i>.codes.jld2, <i>.residuals.jld2, doclens.<i>.jld2, <i>.metadata.json (savers.jl:52-84)"
function save_chunk(index_path::String, codes::Vector{UInt32}, residuals::Matrix{UInt8}, chunk_idx::Int,
        passage_offset::Int, doclens::Vector{Int})
    prefix = joinpath(index_path, string(chunk_idx))
    JLD2.save_object("$(prefix).codes.jld2", codes)
    JLD2.save_object("$(prefix).residuals.jld2", residuals)
    JLD2.save_object(joinpath(index_path, "doclens.$(chunk_idx).jld2"), doclens)
    _json_write("$(prefix).metadata.json", Dict("passage_offset" => passage_offset,
        "num_passages" => length(doclens), "num_embeddings" => length(codes)))
end

_plan(index_path::String) = JSON.parsefile(joinpath(index_path, "plan.json"))

function load_doclens(index_path::String)
    n = _plan(index_path)["num_chunks"]
    reduce(vcat, (JLD2.load_object(joinpath(index_path, "doclens.$(i).jld2"))::Vector{Int} for i in 1:n); init = Int[])
end

function load_codes(index_path::String)
    n = _plan(index_path)["num_chunks"]
    reduce(vcat, (JLD2.load_object(joinpath(index_path, "$(i).codes.jld2"))::Vector{UInt32} for i in 1:n);
        init = UInt32[])
end

"every chunk's codes and residuals, concatenated (loaders.jl:91-113)"
function load_compressed_embs(index_path::String)
    plan = _plan(index_path)
    config = load_config(index_path)
    n_emb = Int(plan["num_embeddings"])
    codes = Vector{UInt32}(undef, n_emb)
    residuals = Matrix{UInt8}(undef, div(config.dim, 8) * config.nbits, n_emb)
    off = 0
    for i in 1:Int(plan["num_chunks"])
        c = JLD2.load_object(joinpath(index_path, "$(i).codes.jld2"))::Vector{UInt32}
        r = JLD2.load_object(joinpath(index_path, "$(i).residuals.jld2"))::Matrix{UInt8}
        codes[(off + 1):(off + length(c))] = c
        residuals[:, (off + 1):(off + length(c))] = r
        off += length(c)
    end
    off == n_emb || error("plan.json lists $(n_emb) embeddings, the chunks hold $(off)")
    codes, residuals
end

function save_chunk_metadata_property(index_path::String, property::String, values::Vector)
    plan = _plan(index_path)
    plan["num_chunks"] == length(values) || error("one value per chunk expected")
    for i in 1:length(values)
        path = joinpath(index_path, "$(i).metadata.json")
        meta = JSON.parsefile(path)
        meta[property] = values[i]
        _json_write(path, meta)
    end
end

"every file index() must have produced (collection_indexer.jl:299-340)"
function _check_all_files_are_saved(index_path::String)
    isfile(joinpath(index_path, "plan.json")) || return false
    files = ["config.json", "centroids.jld2", "avg_residual.jld2", "bucket_cutoffs.jld2", "bucket_weights.jld2",
        "ivf.jld2", "ivf_lengths.jld2"]
    for i in 1:Int(_plan(index_path)["num_chunks"])
        append!(files, ["$(i).codes.jld2", "$(i).residuals.jld2", "doclens.$(i).jld2", "$(i).metadata.json"])
    end
    all(f -> isfile(joinpath(index_path, f)), files)
end
