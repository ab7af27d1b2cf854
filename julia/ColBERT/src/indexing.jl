# Indexer / index(indexer): the reference's index build (src/indexing.jl:1-147,
# src/indexing/collection_indexer.jl) with every array stage on the device.

struct Indexer
    config::ColBERTConfig
    checkpoint::Checkpoint
    tokenizer::WordPieceTokenizer
    collection::Vector{String}
    skiplist::Vector{Int}
end

"Indexer(config) (src/indexing.jl:24-52): load the checkpoint, read the collection (one passage per line)"
function Indexer(config::ColBERTConfig; device::Int = 0)
    tokenizer, ckpt = load_hgf_pretrained_local(config.checkpoint; device = device)
    collection = config.collection isa String ? readlines(config.collection) : config.collection
    skiplist = doc_skiplist(tokenizer, config.mask_punctuation)
    @info "Loaded $(length(collection)) documents from $(config.collection)."
    Indexer(config, ckpt, tokenizer, collection, skiplist)
end

_encode(ix::Indexer, passages) = encode_passages(ix.checkpoint, ix.tokenizer, passages, ix.config.dim,
    ix.config.index_bsize, ix.config.doc_token_id, ix.skiplist, ix.config.doc_maxlen)

"how many passages are sampled for clustering: 16 sqrt(120 N), at most N (collection_indexer.jl:17-24)"
_num_sampled_pids(n::Int) = min(1 + floor(Int, 16 * sqrt(120 * n)), n)

"the plan: chunking and the number of centroids, 2^floor(log2(16 sqrt(N avg_doclen))) (collection_indexer.jl:115-139)"
function _plan_dict(config::ColBERTConfig, num_documents::Int, avg_doclen_est::Float32, num_sample_embs::Int)
    chunksize = ismissing(config.chunksize) ? min(25000, 1 + fld(num_documents, config.nranks)) : config.chunksize
    num_embeddings_est = Float32(num_documents) * avg_doclen_est
    num_partitions = floor(Int, 2^floor(log2(16 * sqrt(num_embeddings_est))))
    Dict{String, Any}("chunksize" => chunksize, "num_chunks" => cld(num_documents, chunksize),
        "num_partitions" => min(num_sample_embs, num_partitions), "num_documents" => num_documents,
        "num_embeddings_est" => num_embeddings_est, "avg_doclen_est" => avg_doclen_est)
end

"train (collection_indexer.jl:219-237): k-means on the sample, codec statistics on the held-out part"
function train(sample::Matrix{Float32}, heldout::Matrix{Float32}, num_partitions::Int, nbits::Int, kmeans_niters::Int;
        device::Int = 0)
    centroids = sample[:, randperm(size(sample, 2))[1:num_partitions]]
    kmeans_gpu_onehot!(sample, centroids, num_partitions; max_iters = kmeans_niters, device = device)
    codes = zeros(UInt32, size(heldout, 2))
    bucket_cutoffs, bucket_weights, avg_residual = _compute_avg_residuals!(nbits, centroids, heldout, codes;
        device = device)
    centroids, bucket_cutoffs, bucket_weights, avg_residual
end

"""
    index(indexer::Indexer)

Build the index at `indexer.config.index_path` (src/indexing.jl:63-147).  Nothing is done if the directory exists.

`device_resident = true` keeps every large array in HBM between the stages (`_index_device`: the route the Python twin takes
by default).  It is opt-in here until it has run once under a real `julia` (ADVICE r05): the default is the host route, which
moves every stage's arrays through the C ABI's host entry points.  When the packed device encode refuses a model (a head size
other than 64, a GEMM mode other than f16x3: `ArgumentError` / `ErrorException` from `clb_encode_docs_packed_device`) the
device route falls through to the host route instead of failing.
"""
function index(indexer::Indexer; device::Int = 0, device_resident::Bool = false)
    config = indexer.config
    path = config.index_path
    if isdir(path)
        @info "Index at $(path) already exists! Skipping indexing."
        return
    end
    if device_resident && _device_route_fits(indexer; device = device)
        try
            return _index_device(indexer; device = device)
        catch err
            # remove only what THIS call created (_index_device makes the directory after tokenising: nothing of anyone else's
            # can be inside a directory that did not exist a moment ago)
            isdir(path) && rm(path; recursive = true, force = true)
            if err isa ArgumentError || (err isa ErrorException && occursin("packed", err.msg))
                @warn "the device-resident route refused this model; indexing through the host route" exception = err
            else
                rethrow()
            end
        end
    end
    n_docs = length(indexer.collection)
    # sample -> embeddings; held-out split: 5 % of the sample, at most 50 000
    sampled = sort(collect(Set(rand(1:n_docs, _num_sampled_pids(n_docs)))))
    sample, sample_doclens = _encode(indexer, indexer.collection[sampled])
    avg_doclen_est = Float32(sum(sample_doclens) / max(length(sample_doclens), 1))
    sample = sample[:, shuffle(1:size(sample, 2))]
    n_heldout = max(1, floor(Int, min(50000.0f0, 0.05f0 * size(sample, 2))))
    heldout = sample[:, (end - n_heldout + 1):end]
    sample = sample[:, 1:(end - n_heldout)]
    mkpath(path)
    JLD2.save_object(joinpath(path, "sample.jld2"), sample)
    JLD2.save_object(joinpath(path, "sample_heldout.jld2"), heldout)
    plan = _plan_dict(config, n_docs, avg_doclen_est, size(sample, 2))
    _json_write(joinpath(path, "plan.json"), plan)
    save(config)
    # codec
    centroids, bucket_cutoffs, bucket_weights, avg_residual = train(sample, heldout, plan["num_partitions"],
        config.nbits, config.kmeans_niters; device = device)
    save_codec(path, centroids, bucket_cutoffs, bucket_weights, avg_residual)
    # chunks: encode, compress, save (collection_indexer.jl:271-297)
    chunksize = plan["chunksize"]
    counts = Int[]
    for (chunk_idx, start) in enumerate(1:chunksize:n_docs)
        stop = min(n_docs, start + chunksize - 1)
        embs, doclens = _encode(indexer, indexer.collection[start:stop])
        codes, residuals = compress(centroids, bucket_cutoffs, config.dim, config.nbits, embs; device = device)
        save_chunk(path, codes, residuals, chunk_idx, start, doclens)
        push!(counts, length(codes))
    end
    # embedding offsets (indexing.jl:119-132)
    offsets = isempty(counts) ? [0] : cumsum([1; counts[1:(end - 1)]])
    plan["num_embeddings"] = sum(counts)
    plan["embeddings_offsets"] = offsets
    _json_write(joinpath(path, "plan.json"), plan)
    save_chunk_metadata_property(path, "embedding_offset", offsets)
    # IVF (collection_indexer.jl:349-353)
    ivf, ivf_lengths = _build_ivf(load_codes(path), plan["num_partitions"]; device = device)
    JLD2.save_object(joinpath(path, "ivf.jld2"), ivf)
    JLD2.save_object(joinpath(path, "ivf_lengths.jld2"), ivf_lengths)
    _check_all_files_are_saved(path) || error("the index at $(path) is incomplete")
    nothing
end


# ---- the same build with the embeddings kept in HBM -----------------------------------------------------------------
# Through the host-buffer entry points above every embedding crosses PCIe twice (the encoder's output comes back, then goes up
# again for k-means / compress: 41 GB each way at 1 M passages).  Here only token ids go up and only the finished index comes
# back: packed passage batches (clb_encode_docs_packed_device) -> sample cut out and shuffled on the device
# (clb_gather_rows_device) -> device k-means (clb_kmeans_shard_*_device) -> per chunk: encode, compress with the resident codec
# (clb_codec_compress_device), download codes / residuals, save -> IVF over the device array of all codes.  The files written
# are the reference's (src/indexing.jl:84-147); Python's indexer._index_through_device is the tested twin of this function.

"[CLS] [D] w1 .. wn [SEP] of a passage, cut to doc_maxlen tokens: its column of tensorize_docs, attended rows only"
function _passage_tokens(ix::Indexer, text::AbstractString)
    ids = first(encode_text(ix.tokenizer, text), ix.config.doc_maxlen - 1)
    Int32[ids[1]; Int32(lookup(ix.tokenizer, ix.config.doc_token_id)); ids[2:end]]
end

"upper bound of the route's HBM footprint (codes + IVF + sort scratch of the whole collection, one chunk, the sample)"
function _device_route_fits(ix::Indexer; device::Int = 0)
    n_docs = length(ix.collection)
    maxlen = ix.config.doc_maxlen
    chunk = ismissing(ix.config.chunksize) ? min(25000, 1 + n_docs) : ix.config.chunksize
    need = n_docs * maxlen * 28 + min(chunk, n_docs) * maxlen * (4 * ix.config.dim + div(ix.config.dim, 8) * ix.config.nbits) +
           _num_sampled_pids(n_docs) * maxlen * 4 * ix.config.dim * 2
    free, _ = device_memory(device)
    need < 0.8 * free
end

"encode the passages `tokens` (packed batches of 4 x index_bsize) into a fresh device matrix (dim, sum(doclens))"
function _encode_device(ix::Indexer, d_skip::DeviceBuffer, tokens::Vector{Vector{Int32}}, doclens::Vector{Int}; device::Int = 0)
    dim = ix.config.dim
    out = DeviceBuffer(4 * dim * max(sum(doclens), 1); device = device)
    bs = 4 * ix.config.index_bsize
    fill = 0
    for off in 1:bs:length(tokens)
        stop = min(length(tokens), off + bs - 1)
        got = _doc_embeddings_packed_device!(ix.checkpoint, d_skip, length(ix.skiplist), tokens[off:stop], out, 4 * dim * fill; device = device)
        got == doclens[off:stop] || error("the device's doclens disagree with the tokenizer's")
        fill += sum(got)
    end
    out
end

function _index_device(indexer::Indexer; device::Int = 0)
    config = indexer.config
    path = config.index_path
    dim, nbits = config.dim, config.nbits
    n_docs = length(indexer.collection)
    # every passage is tokenised once; a passage keeps its attended tokens outside the skiplist (checkpoint.jl:37-43), so the
    # doclens follow from the tokens alone and every output offset is known before anything is encoded
    tokens = [_passage_tokens(indexer, p) for p in indexer.collection]
    skip = Set(Int32.(indexer.skiplist))
    doclens = Int[count(t -> !(t in skip), toks) for toks in tokens]
    d_skip = DeviceBuffer(8 * max(length(indexer.skiplist), 1); device = device)
    device_upload!(d_skip, indexer.skiplist)
    # sample (collection_indexer.jl:17-24, 56-91): encode only the sampled passages, shuffle, split off the held-out part
    sampled = sort(collect(Set(rand(1:n_docs, _num_sampled_pids(n_docs)))))
    n_sample = sum(doclens[sampled])
    raw = _encode_device(indexer, d_skip, tokens[sampled], doclens[sampled]; device = device)
    avg_doclen_est = Float32(n_sample / max(length(sampled), 1))
    shuffled = DeviceBuffer(4 * dim * n_sample; device = device)
    gather_columns_device!(shuffled, 0, raw, 0, n_sample, 4 * dim, shuffle(1:n_sample))
    raw = nothing
    n_heldout = max(1, floor(Int, min(50000.0f0, 0.05f0 * n_sample)))
    n_train = n_sample - n_heldout
    sample = Matrix{Float32}(undef, dim, n_train)
    heldout = Matrix{Float32}(undef, dim, n_heldout)
    device_download!(sample, shuffled)                                  # the reference saves both (indexing.jl:84-90)
    device_download!(heldout, shuffled, 4 * dim * n_train)
    mkpath(path)
    JLD2.save_object(joinpath(path, "sample.jld2"), sample)
    JLD2.save_object(joinpath(path, "sample_heldout.jld2"), heldout)
    sample = nothing
    plan = _plan_dict(config, n_docs, avg_doclen_est, n_train)
    _json_write(joinpath(path, "plan.json"), plan)
    save(config)
    # train (collection_indexer.jl:219-237): the initial centroids are sample columns, gathered on the device
    K = plan["num_partitions"]
    d_init = DeviceBuffer(4 * dim * K; device = device)
    gather_columns_device!(d_init, 0, shuffled, 0, n_train, 4 * dim, randperm(n_train)[1:K])
    d_centroids = _kmeans_device(shuffled, dim, n_train, d_init, K; max_iters = config.kmeans_niters)
    shuffled = nothing
    centroids = Matrix{Float32}(undef, dim, K)
    device_download!(centroids, d_centroids)
    bucket_cutoffs, bucket_weights, avg_residual = _compute_avg_residuals!(nbits, centroids, heldout,
        zeros(UInt32, n_heldout); device = device)
    save_codec(path, centroids, bucket_cutoffs, bucket_weights, avg_residual)
    # chunk loop (collection_indexer.jl:271-297): only the codes stay on the device (for the IVF)
    chunksize = plan["chunksize"]
    n_emb = sum(doclens)
    rows = div(dim, 8) * nbits
    d_codes = DeviceBuffer(4 * max(n_emb, 1); device = device)
    codec = _codec_create(dim, nbits, K, d_centroids, bucket_cutoffs)
    counts = Int[]
    fill = 0
    try
        for (chunk_idx, start) in enumerate(1:chunksize:n_docs)
            stop = min(n_docs, start + chunksize - 1)
            n = sum(doclens[start:stop])
            d_embs = _encode_device(indexer, d_skip, tokens[start:stop], doclens[start:stop]; device = device)
            d_res = DeviceBuffer(max(rows * n, 1); device = device)
            _codec_compress_device!(codec, d_embs, n, d_codes, 4 * fill, d_res)
            codes = Vector{UInt32}(undef, n)
            residuals = Matrix{UInt8}(undef, rows, n)
            device_download!(codes, d_codes, 4 * fill)
            device_download!(residuals, d_res)
            save_chunk(path, codes, residuals, chunk_idx, start, doclens[start:stop])
            push!(counts, n)
            fill += n
        end
    finally
        _codec_destroy(codec)
    end
    offsets = isempty(counts) ? [0] : cumsum([1; counts[1:(end - 1)]])
    plan["num_embeddings"] = sum(counts)
    plan["embeddings_offsets"] = offsets
    _json_write(joinpath(path, "plan.json"), plan)
    save_chunk_metadata_property(path, "embedding_offset", offsets)
    ivf, ivf_lengths = _build_ivf_device(d_codes, n_emb, K)
    JLD2.save_object(joinpath(path, "ivf.jld2"), ivf)
    JLD2.save_object(joinpath(path, "ivf_lengths.jld2"), ivf_lengths)
    _check_all_files_are_saved(path) || error("the index at $(path) is incomplete")
    nothing
end
