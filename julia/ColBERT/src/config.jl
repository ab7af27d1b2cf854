# The reference's configuration struct: same field names, types and defaults (src/infra/config.jl:54-90), so that
# config.json files and user scripts are interchangeable.
Base.@kwdef struct ColBERTConfig
    # run settings
    use_gpu::Bool = false
    rank::Int = 0
    nranks::Int = 1

    # tokenization settings
    query_token_id::String = "[unused0]"
    doc_token_id::String = "[unused1]"
    query_token::String = "[Q]"
    doc_token::String = "[D]"

    # resource settings
    checkpoint::String = "colbert-ir/colbertv2.0"
    collection::Union{String, Vector{String}} = ""

    # doc settings
    dim::Int = 128
    doc_maxlen::Int = 300
    mask_punctuation::Bool = true

    # query settings
    query_maxlen::Int = 32
    attend_to_mask_tokens::Bool = false

    # indexing settings
    index_path::String = ""
    index_bsize::Int = 64
    chunksize::Union{Missing, Int} = 25000
    passages_batch_size::Int = 5000
    nbits::Int = 2
    kmeans_niters::Int = 20

    # search settings
    nprobe::Int = 2
    ncandidates::Int = 8192
end
