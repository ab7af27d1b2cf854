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
"""
function index(indexer::Indexer; device::Int = 0)
    config = indexer.config
    path = config.index_path
    if isdir(path)
        @info "Index at $(path) already exists! Skipping indexing."
        return
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
