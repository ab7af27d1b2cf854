# Loading a checkpoint and the batching loops around the device encoder (src/local_loading.jl:139-209,
# src/modelling/checkpoint.jl:159-189,271-301).
#
# `tools/export_checkpoint.py <hf_dir> <hf_dir>` writes encoder.f32 (the flat fp32 blob of include/colbert_hip.h) and
# encoder.json next to the HuggingFace files; the tokenizer reads the checkpoint's own vocab.txt.

"(tokenizer, checkpoint) for a local HuggingFace-format ColBERT checkpoint directory"
function load_hgf_pretrained_local(dir::String; device::Int = 0)
    isdir(dir) || error("checkpoint directory $(dir) not found (only local checkpoints are supported)")
    meta_path = joinpath(dir, "encoder.json")
    isfile(meta_path) || error("$(meta_path) not found: run `python tools/export_checkpoint.py $(dir) $(dir)` first")
    meta = JSON.parsefile(meta_path)
    bert = meta["bert"]
    weights = Vector{Float32}(undef, Int(meta["n_floats"]))
    read!(joinpath(dir, "encoder.f32"), weights)
    ckpt = Checkpoint(weights; vocab = Int(bert["vocab_size"]), hidden = Int(bert["hidden_size"]),
        layers = Int(bert["num_hidden_layers"]), heads = Int(bert["num_attention_heads"]),
        intermediate = Int(bert["intermediate_size"]), max_pos = Int(bert["max_position_embeddings"]),
        type_vocab = Int(get(bert, "type_vocab_size", 2)), dim = Int(meta["dim"]),
        ln_eps = Float32(get(bert, "layer_norm_eps", 1.0e-12)), device = device)
    WordPieceTokenizer(joinpath(dir, "vocab.txt")), ckpt
end

"encode_passages (checkpoint.jl:159-189): batches of `index_bsize` passages -> (embs (dim, sum(doclens)), doclens)"
function encode_passages(ckpt::Checkpoint, tokenizer::WordPieceTokenizer, passages::AbstractVector{<:AbstractString},
        dim::Int, index_bsize::Int, doc_token::String, skiplist::Vector{Int}, doc_maxlen::Int)
    isempty(passages) && return zeros(Float32, dim, 0), zeros(Int, 0)
    embs = Matrix{Float32}[]
    doclens = Vector{Int}[]
    for off in 1:index_bsize:length(passages)
        batch = passages[off:min(length(passages), off + index_bsize - 1)]
        ids, mask = tensorize_docs(doc_token, tokenizer, batch, doc_maxlen)
        D, dl = _doc_embeddings_and_doclens(ckpt, skiplist, ids, mask)
        push!(embs, D)
        push!(doclens, dl)
    end
    reduce(hcat, embs), reduce(vcat, doclens)
end

"encode_queries (checkpoint.jl:271-301) -> (dim, query_maxlen, length(queries))"
function encode_queries(ckpt::Checkpoint, tokenizer::WordPieceTokenizer, queries::AbstractVector{<:AbstractString},
        dim::Int, index_bsize::Int, query_token::String, attend_to_mask_tokens::Bool, skiplist::Vector{Int},
        query_maxlen::Int)
    isempty(queries) && return zeros(Float32, dim, query_maxlen, 0)
    out = Array{Float32, 3}[]
    for off in 1:index_bsize:length(queries)
        batch = queries[off:min(length(queries), off + index_bsize - 1)]
        ids, mask = tensorize_queries(query_token, attend_to_mask_tokens, tokenizer, batch, query_maxlen)
        push!(out, _query_embeddings(ckpt, skiplist, ids, mask))
    end
    cat(out...; dims = 3)
end
