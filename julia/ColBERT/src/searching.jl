# Searcher / search (src/searching.jl).  The arrays the reference keeps in host memory (codes, residuals, ivf,
# emb2pid, ...) are uploaded once into HBM and live behind `handle`.

mutable struct Searcher
    config::ColBERTConfig
    checkpoint::Checkpoint
    tokenizer::WordPieceTokenizer
    skiplist::Vector{Int}
    handle::Ptr{Cvoid}
    num_documents::Int
    function Searcher(config, checkpoint, tokenizer, skiplist, handle, num_documents)
        s = new(config, checkpoint, tokenizer, skiplist, handle, num_documents)
        finalizer(s -> _searcher_destroy(s.handle), s)
    end
end

"""
    Searcher(index_path::String)

Open an index directory (src/searching.jl:18-80): config.json, the checkpoint it names, the codec, the IVF, every
chunk's doclens / codes / residuals.  `emb2pid` (:82-91) is built by the library on the device.
"""
function Searcher(index_path::String; device::Int = 0)
    isdir(index_path) || error("Index at $(index_path) does not exist! Please build the index first and try again.")
    config = load_config(index_path)
    tokenizer, ckpt = load_hgf_pretrained_local(config.checkpoint; device = device)
    codec = load_codec(index_path)
    ivf = JLD2.load_object(joinpath(index_path, "ivf.jld2"))::Vector{Int}
    ivf_lengths = JLD2.load_object(joinpath(index_path, "ivf_lengths.jld2"))::Vector{Int}
    doclens = load_doclens(index_path)
    codes, residuals = load_compressed_embs(index_path)
    handle = _searcher_create(config.nbits, codec["centroids"], codec["bucket_weights"], doclens, codes, residuals,
        ivf, ivf_lengths; device = device)
    skiplist = Int[lookup(tokenizer, "[PAD]")]         # only the pad symbol (searching.jl:62)
    Searcher(config, ckpt, tokenizer, skiplist, handle, length(doclens))
end

"""
    search(searcher, query::String, k::Int) -> (pids::Vector{Int}, scores::Vector{Float32})

Same contract as src/searching.jl:93-128: 1-based pids by descending score, ties by ascending pid; a BoundsError if
fewer than `k` passages are candidates.
"""
function search(searcher::Searcher, query::String, k::Int)
    c = searcher.config
    Q = encode_queries(searcher.checkpoint, searcher.tokenizer, [query], c.dim, c.index_bsize, c.query_token,
        c.attend_to_mask_tokens, searcher.skiplist, c.query_maxlen)
    @assert size(Q)[3]==1 "size(Q): $(size(Q))"
    @assert isequal(size(Q)[2], c.query_maxlen) "size(Q): $(size(Q)), query_maxlen: $(c.query_maxlen)"
    search(searcher, reshape(Q, size(Q, 1), size(Q, 2)), k)
end

"search from query embeddings (dim, query_maxlen)"
search(searcher::Searcher, Q::Matrix{Float32}, k::Int) = _search(searcher.handle, Q, searcher.config.nprobe, k)
