# capi.jl -- one `ccall` per entry point of include/colbert_hip.h.  Pure marshalling: pointers, lengths, status
# codes.  Julia arrays are passed as Ptr{T} under GC.@preserve; they are column-major and 1-based exactly as the
# C ABI expects, so nothing is copied or re-indexed on the Julia side.

const libcolbert = get(ENV, "COLBERT_HIP_LIB", "libcolbert_hip.so")

# ---- errors: codes 1..4 map 1:1 onto the exceptions the reference throws -------------------------------
function _check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:clb_last_error, libcolbert), Cstring, ()))
    rc == 1 && throw(DimensionMismatch(msg))
    rc == 2 && throw(DomainError(msg))
    rc == 3 && throw(BoundsError(msg))
    rc == 4 && throw(ArgumentError(msg))
    error("libcolbert_hip error $rc: $msg")
end

# ---- searcher handle (struct Searcher's array fields live in HBM behind it; src/searching.jl:1-16,44-59) ------
function _searcher_create(nbits::Int, centroids::Matrix{Float32}, bucket_weights::Vector{Float32},
        doclens::Vector{Int}, codes::Vector{UInt32}, residuals::Matrix{UInt8}, ivf::Vector{Int},
        ivf_lengths::Vector{Int}; device::Int = 0, pid_offset::Int = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve centroids bucket_weights doclens codes residuals ivf ivf_lengths begin
        _check(ccall((:clb_searcher_create, libcolbert), Cint,
            (Cint, Int64, Cint, Int64, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Int64}, Int64, Ptr{UInt32},
                Ptr{UInt8}, Ptr{Int64}, Ptr{Int64}, Int64, Ref{Ptr{Cvoid}}),
            device, size(centroids, 1), nbits, size(centroids, 2), centroids, bucket_weights,
            length(doclens), doclens, length(codes), codes, residuals, ivf, ivf_lengths, pid_offset, h))
    end
    h[]
end
_searcher_destroy(h::Ptr{Cvoid}) = ccall((:clb_searcher_destroy, libcolbert), Cint, (Ptr{Cvoid},), h)
"selection step by one (0) or sixteen (1) work-groups per query; -1 (default): chosen by the candidate capacity"
_searcher_set_wide_select(h::Ptr{Cvoid}, on::Integer) =
    _check(ccall((:clb_searcher_set_wide_select, libcolbert), Cint, (Ptr{Cvoid}, Cint), h, on))

"search() after encode_queries (src/searching.jl:102-127): one library call."
function _search(handle::Ptr{Cvoid}, Q::Matrix{Float32}, nprobe::Int, k::Int)
    pids = Vector{Int}(undef, k)
    scores = Vector{Float32}(undef, k)
    ncand = Ref{Int64}(0)
    GC.@preserve Q pids scores begin
        _check(ccall((:clb_search, libcolbert), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float32}, Ref{Int64}),
            handle, Q, size(Q, 2), nprobe, k, pids, scores, ncand))
    end
    pids, scores
end

# ---- codec / index-build call sites (src/indexing/codecs/residual.jl, src/utils.jl, collection_indexer.jl)
function compress(centroids::Matrix{Float32}, bucket_cutoffs::Vector{Float32}, dim::Int, nbits::Int,
        embs::AbstractMatrix{Float32}; device::Int = 0)
    embs = Matrix{Float32}(embs)
    codes = zeros(UInt32, size(embs, 2))
    residuals = Matrix{UInt8}(undef, div(dim, 8) * nbits, size(embs, 2))
    GC.@preserve centroids bucket_cutoffs embs codes residuals begin
        _check(ccall((:clb_compress, libcolbert), Cint,
            (Cint, Ptr{Float32}, Int64, Ptr{Float32}, Int64, Int64, Cint, Ptr{Float32}, Int64, Ptr{UInt32}, Ptr{UInt8}),
            device, centroids, size(centroids, 2), bucket_cutoffs, length(bucket_cutoffs), dim, nbits, embs,
            size(embs, 2), codes, residuals))
    end
    codes, residuals
end

function decompress(dim::Int, nbits::Int, centroids::Matrix{Float32}, bucket_weights::Vector{Float32},
        codes::Vector{UInt32}, residuals::AbstractMatrix{UInt8}; device::Int = 0)
    residuals = Matrix{UInt8}(residuals)
    out = Matrix{Float32}(undef, dim, length(codes))
    GC.@preserve centroids bucket_weights codes residuals out begin
        _check(ccall((:clb_decompress, libcolbert), Cint,
            (Cint, Int64, Cint, Ptr{Float32}, Int64, Ptr{Float32}, Int64, Ptr{UInt32}, Int64, Ptr{UInt8}, Int64, Int64, Ptr{Float32}),
            device, dim, nbits, centroids, size(centroids, 2), bucket_weights, length(bucket_weights), codes,
            length(codes), residuals, size(residuals, 1), size(residuals, 2), out))
    end
    out
end

"kmeans_gpu_onehot! (src/utils.jl:253-318); `centroids` holds the initial centroids and is updated in place."
function kmeans_gpu_onehot!(data::Matrix{Float32}, centroids::Matrix{Float32}, k::Int; max_iters::Int = 10,
        tol::Float32 = 1.0f-4, point_bsize::Int = 1000, device::Int = 0)
    size(centroids, 2) == k || throw(DimensionMismatch("size(centroids, 2) must be k!"))
    assignments = Vector{Int32}(undef, size(data, 2))
    iters = Ref{Int64}(0)
    GC.@preserve data centroids assignments begin
        _check(ccall((:clb_kmeans, libcolbert), Cint,
            (Cint, Ptr{Float32}, Int64, Int64, Ptr{Float32}, Int64, Int64, Float32, Int64, Ptr{Int32}, Ref{Int64}),
            device, data, size(data, 1), size(data, 2), centroids, k, max_iters, tol, point_bsize, assignments, iters))
    end
    assignments
end

function _compute_avg_residuals!(nbits::Int, centroids::Matrix{Float32}, heldout::Matrix{Float32},
        codes::Vector{UInt32}; device::Int = 0)
    cutoffs = Vector{Float32}(undef, (1 << nbits) - 1)
    weights = Vector{Float32}(undef, 1 << nbits)
    avg = Ref{Float32}(0)
    GC.@preserve centroids heldout codes cutoffs weights begin
        _check(ccall((:clb_compute_avg_residuals, libcolbert), Cint,
            (Cint, Cint, Ptr{Float32}, Int64, Int64, Ptr{Float32}, Int64, Ptr{UInt32}, Int64, Ptr{Float32}, Ptr{Float32}, Ref{Float32}),
            device, nbits, centroids, size(centroids, 1), size(centroids, 2), heldout, size(heldout, 2), codes,
            length(codes), cutoffs, weights, avg))
    end
    cutoffs, weights, avg[]
end

function _build_ivf(codes::Vector{UInt32}, num_partitions::Int; device::Int = 0)
    ivf = Vector{Int}(undef, length(codes))
    ivf_lengths = Vector{Int}(undef, num_partitions)
    GC.@preserve codes ivf ivf_lengths begin
        _check(ccall((:clb_build_ivf, libcolbert), Cint, (Cint, Ptr{UInt32}, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
            device, codes, length(codes), num_partitions, ivf, ivf_lengths))
    end
    ivf, ivf_lengths
end

# ---- encoder: the BERT + Dense weights live on the device (reference: src/modelling/checkpoint.jl) ---------------
mutable struct Checkpoint
    handle::Ptr{Cvoid}
    dim::Int
    function Checkpoint(weights::Vector{Float32}; vocab::Int, hidden::Int, layers::Int, heads::Int,
            intermediate::Int, max_pos::Int, type_vocab::Int = 2, dim::Int = 128, ln_eps::Float32 = 1.0f-12,
            device::Int = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve weights _check(ccall((:clb_encoder_create, libcolbert), Cint,
            (Cint, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Float32, Ptr{Float32}, Int64, Ref{Ptr{Cvoid}}),
            device, vocab, hidden, layers, heads, intermediate, max_pos, type_vocab, dim, ln_eps, weights,
            length(weights), h))
        c = new(h[], dim)
        finalizer(c -> ccall((:clb_encoder_destroy, libcolbert), Cint, (Ptr{Cvoid},), c.handle), c)
    end
end

"doc(bert, linear, integer_ids, bitmask) (src/modelling/checkpoint.jl:21-25)"
function doc(ckpt::Checkpoint, integer_ids::Matrix{Int32}, bitmask::AbstractMatrix{Bool})
    L, N = size(integer_ids)
    mask = Matrix{UInt8}(bitmask)
    out = Array{Float32, 3}(undef, ckpt.dim, L, N)
    GC.@preserve integer_ids mask out _check(ccall((:clb_encode, libcolbert), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Ptr{UInt8}, Int64, Int64, Ptr{Float32}), ckpt.handle, integer_ids, mask, L, N, out))
    out
end

"_doc_embeddings_and_doclens (checkpoint.jl:27-52)"
function _doc_embeddings_and_doclens(ckpt::Checkpoint, skiplist::Vector{Int}, integer_ids::Matrix{Int32},
        bitmask::AbstractMatrix{Bool})
    L, N = size(integer_ids)
    mask = Matrix{UInt8}(bitmask)
    out = Matrix{Float32}(undef, ckpt.dim, L * N)
    doclens = Vector{Int}(undef, N)
    n_out = Ref{Int64}(0)
    GC.@preserve integer_ids mask skiplist out doclens _check(ccall((:clb_encode_docs, libcolbert), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Ptr{UInt8}, Int64, Int64, Ptr{Int64}, Int64, Ptr{Float32}, Ptr{Int64}, Ref{Int64}),
        ckpt.handle, integer_ids, mask, L, N, skiplist, length(skiplist), out, doclens, n_out))
    out[:, 1:n_out[]], doclens
end

"_query_embeddings (checkpoint.jl:54-71)"
function _query_embeddings(ckpt::Checkpoint, skiplist::Vector{Int}, integer_ids::Matrix{Int32},
        bitmask::AbstractMatrix{Bool})
    L, N = size(integer_ids)
    mask = Matrix{UInt8}(bitmask)
    out = Array{Float32, 3}(undef, ckpt.dim, L, N)
    GC.@preserve integer_ids mask skiplist out _check(ccall((:clb_encode_queries, libcolbert), Cint,
        (Ptr{Cvoid}, Ptr{Int32}, Ptr{UInt8}, Int64, Int64, Ptr{Int64}, Int64, Ptr{Float32}),
        ckpt.handle, integer_ids, mask, L, N, skiplist, length(skiplist), out))
    out
end


# ---- exchange step of a sharded search / index build on RCCL (clb_comm_*; SURVEY.md 8(e)) -------------------------
# One `Communicator` per Julia process and GPU.  Rank 0 calls `comm_unique_id()` and hands the bytes to the other
# ranks (a file, MPI.jl, a socket); every rank then constructs `Communicator(device, rank, n_ranks, id)`.
# The device pointers these calls take are the ones the sharded search entry points produce (clb_search_shard_phase1,
# clb_packed_topk_bytes / clb_merge_topk_packed_device): a host that keeps its device buffers in AMDGPU.jl arrays
# passes their pointers.
"the bytes rank 0 creates and every rank passes to `Communicator`"
function comm_unique_id()
    n = ccall((:clb_comm_unique_id_bytes, libcolbert), Int64, ())
    id = Vector{UInt8}(undef, n)
    GC.@preserve id _check(ccall((:clb_comm_unique_id, libcolbert), Cint, (Ptr{UInt8}, Int64), id, n))
    id
end

mutable struct Communicator
    handle::Ptr{Cvoid}
    rank::Int
    n_ranks::Int
    function Communicator(device::Integer, rank::Integer, n_ranks::Integer, id::Vector{UInt8})
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve id _check(ccall((:clb_comm_create, libcolbert), Cint,
            (Cint, Cint, Cint, Ptr{UInt8}, Int64, Ref{Ptr{Cvoid}}), device, rank, n_ranks, id, length(id), h))
        c = new(h[], rank, n_ranks)
        finalizer(c -> ccall((:clb_comm_destroy, libcolbert), Cint, (Ptr{Cvoid},), c.handle), c)
    end
end

"all-gather of `bytes_per_rank` bytes per rank (device pointers), enqueued on `stream`"
comm_all_gather(c::Communicator, d_send::Ptr{Cvoid}, d_recv::Ptr{Cvoid}, bytes_per_rank::Integer, stream::Ptr{Cvoid} = C_NULL) =
    _check(ccall((:clb_comm_all_gather, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        c.handle, d_send, d_recv, bytes_per_rank, stream))

"in-place element-wise maximum over the ranks of `n` Float32 at a device pointer (the bound constants of the two-phase search)"
comm_all_reduce_max!(c::Communicator, d_buf::Ptr{Cvoid}, n::Integer, stream::Ptr{Cvoid} = C_NULL) =
    _check(ccall((:clb_comm_all_reduce_max_f32, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        c.handle, d_buf, n, stream))

"""
    sync_bound_consts!(searcher_handle, c::Communicator)

The two-phase sharded search (clb_search_shard_phase1 / phase2) cuts every shard at one global threshold, which is only
sound when every shard uses the SAME (the largest) error bound: collective over `c`, once after the shards are loaded.
The library refuses phase 2 with more than one shard until this (or clb_searcher_set_bound_consts) has run.
"""
sync_bound_consts!(handle::Ptr{Cvoid}, c::Communicator) =
    _check(ccall((:clb_searcher_sync_bound_consts, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle, c.handle))
