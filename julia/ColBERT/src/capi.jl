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
"batches of 16+ queries: score rows as 64-byte fp16 rows (0) or 32-byte rows of 8-bit cells (1); -1 (default): 0 -- alike on every shard"
_searcher_set_score_rows(h::Ptr{Cvoid}, form::Integer) =
    _check(ccall((:clb_searcher_set_score_rows, libcolbert), Cint, (Ptr{Cvoid}, Cint), h, form))
"batches of 16+ queries: the fp16 score table from one fp16 product (1) or the three-product bf16 split (3); -1 (default): 1 on a shard of a group, 3 on one GPU -- alike on every shard"
_searcher_set_centroid_products(h::Ptr{Cvoid}, n::Integer) =
    _check(ccall((:clb_searcher_set_centroid_products, libcolbert), Cint, (Ptr{Cvoid}, Cint), h, n))

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


# ---- plain device arrays (clb_device_*): what the device-resident route of index() keeps in HBM ---------------------
# The shim has no GPU array package of its own; a DeviceBuffer is a raw allocation of the library's device, freed by its
# finalizer.  Everything that touches its contents is a library call (upload / download / gather / encode / compress ...).
mutable struct DeviceBuffer
    ptr::Ptr{Cvoid}
    bytes::Int
    device::Int
    function DeviceBuffer(bytes::Integer; device::Integer = 0)
        p = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall((:clb_device_malloc, libcolbert), Cint, (Cint, Int64, Ref{Ptr{Cvoid}}), device, bytes, p))
        b = new(p[], bytes, device)
        finalizer(b -> (b.ptr == C_NULL || ccall((:clb_device_free, libcolbert), Cint, (Cint, Ptr{Cvoid}), b.device, b.ptr); b.ptr = C_NULL), b)
    end
end
"pointer `offset` bytes into the buffer (a slice of a larger allocation: the chunk loop writes codes / residuals in place)"
_at(b::DeviceBuffer, offset::Integer = 0) = b.ptr + offset

function device_upload!(b::DeviceBuffer, a::Array, offset::Integer = 0)
    sizeof(a) + offset <= b.bytes || throw(BoundsError(b, offset + sizeof(a)))
    GC.@preserve a _check(ccall((:clb_device_upload, libcolbert), Cint, (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        b.device, _at(b, offset), a, sizeof(a)))
    b
end
function device_download!(a::Array, b::DeviceBuffer, offset::Integer = 0)
    sizeof(a) + offset <= b.bytes || throw(BoundsError(b, offset + sizeof(a)))
    GC.@preserve a _check(ccall((:clb_device_download, libcolbert), Cint, (Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        b.device, a, _at(b, offset), sizeof(a)))
    a
end
device_synchronize(device::Integer = 0) = _check(ccall((:clb_device_synchronize, libcolbert), Cint, (Cint,), device))
"(free, total) bytes of HBM"
function device_memory(device::Integer = 0)
    f = Ref{Int64}(0); t = Ref{Int64}(0)
    _check(ccall((:clb_device_memory, libcolbert), Cint, (Cint, Ref{Int64}, Ref{Int64}), device, f, t))
    f[], t[]
end

"dst[:, i] = src[:, rows[i]] for a (row_bytes / 4, n_src) Float32-like device matrix: `rows` 1-BASED Julia indices"
function gather_columns_device!(dst::DeviceBuffer, dst_offset::Integer, src::DeviceBuffer, src_offset::Integer, n_src::Integer,
        row_bytes::Integer, rows::Vector{Int})
    idx = DeviceBuffer(8 * max(length(rows), 1); device = src.device)
    device_upload!(idx, rows .- 1)                                     # the ABI takes 0-based indices
    _check(ccall((:clb_gather_rows_device, libcolbert), Cint,
        (Cint, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
        src.device, _at(src, src_offset), n_src, row_bytes, idx.ptr, length(rows), _at(dst, dst_offset), C_NULL))
    dst
end

"""
_doc_embeddings_and_doclens (checkpoint.jl:27-52) for a PACKED batch left on the device (clb_encode_docs_packed_device):
`tokens[i]` = the attended token ids of passage i ([CLS] [D] w1 .. wn [SEP], 1-based); the kept embeddings are written at
`out + out_offset` (capacity: at least the batch's kept tokens); returns the batch's doclens.
"""
function _doc_embeddings_packed_device!(ckpt::Checkpoint, d_skiplist::DeviceBuffer, n_skip::Int, tokens::Vector{Vector{Int32}},
        out::DeviceBuffer, out_offset::Integer; device::Int = 0)
    N = length(tokens)
    lens = Int32[length(t) for t in tokens]
    rows = Int(sum(lens))
    cu = Int32[0; cumsum(lens)]
    buf = Vector{Int32}(undef, 3 * rows + N + 1)
    buf[1:rows] = reduce(vcat, tokens)
    buf[(rows + 1):(2 * rows)] = reduce(vcat, [Int32.(0:(l - 1)) for l in lens])                 # position of every row
    buf[(2 * rows + 1):(3 * rows)] = reduce(vcat, [fill(Int32(i - 1), lens[i]) for i in 1:N])    # passage of every row
    buf[(3 * rows + 1):end] = cu
    d = DeviceBuffer(sizeof(buf); device = device)
    device_upload!(d, buf)
    d_lens = DeviceBuffer(8 * N + 8; device = device)
    doclens = Vector{Int}(undef, N + 1)
    # the launch is asynchronous and reads raw device pointers of d / d_lens / d_skiplist / out: no finalizer may free them before
    # the download below has waited for the batch
    GC.@preserve d d_lens d_skiplist out begin
        _check(ccall((:clb_encode_docs_packed_device, libcolbert), Cint,
            (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64, Int64, Int64, Ptr{Int64}, Int64, Ptr{Float32},
                Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
            ckpt.handle, _at(d, 0), _at(d, 4 * rows), _at(d, 8 * rows), _at(d, 12 * rows), N, Int(maximum(lens)), rows,
            d_skiplist.ptr, n_skip, _at(out, out_offset), _at(d_lens, 0), _at(d_lens, 8 * N), C_NULL))
        device_download!(doclens, d_lens)                              # also waits for the batch (null stream)
    end
    _check(ccall((:clb_encoder_check_last_ids, libcolbert), Cint, (Ptr{Cvoid},), ckpt.handle))
    doclens[1:N]
end

"kmeans_gpu_onehot! (src/utils.jl:253-318) on a sample that is already in HBM; returns the centroids' device buffer"
function _kmeans_device(sample::DeviceBuffer, dim::Int, n::Int, init::DeviceBuffer, k::Int; max_iters::Int = 10,
        tol::Float32 = 1.0f-4, point_bsize::Int = 1000)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:clb_kmeans_shard_create_device, libcolbert), Cint,
        (Cint, Ptr{Float32}, Int64, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), sample.device, sample.ptr, dim, n, k, point_bsize, h))
    try
        _check(ccall((:clb_kmeans_shard_set_centroids, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Float32}), h[], init.ptr))
        block = DeviceBuffer(ccall((:clb_kmeans_shard_block_bytes, libcolbert), Int64, (Ptr{Cvoid},), h[]); device = sample.device)
        delta = Ref{Float32}(0); converged = Ref{Cint}(0)
        for _ in 1:max_iters
            _check(ccall((:clb_kmeans_shard_pass_device, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h[], block.ptr, C_NULL))
            _check(ccall((:clb_kmeans_shard_update_device, libcolbert), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float32, Ref{Float32}, Ref{Cint}, Ptr{Cvoid}), h[], block.ptr, 1, tol, delta, converged, C_NULL))
            @info "max delta: $(delta[])"
            converged[] != 0 && break
        end
        out = DeviceBuffer(4 * dim * k; device = sample.device)
        _check(ccall((:clb_kmeans_shard_get_centroids, libcolbert), Cint, (Ptr{Cvoid}, Ptr{Float32}), h[], out.ptr))
        return out
    finally
        ccall((:clb_kmeans_shard_destroy, libcolbert), Cint, (Ptr{Cvoid},), h[])
    end
end

"the chunk loop's resident codec (clb_codec_*): compress (residual.jl:586-604) on device arrays"
function _codec_create(dim::Int, nbits::Int, k::Int, d_centroids::DeviceBuffer, bucket_cutoffs::Vector{Float32})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve bucket_cutoffs _check(ccall((:clb_codec_create, libcolbert), Cint,
        (Cint, Int64, Cint, Int64, Ptr{Float32}, Ptr{Float32}, Int64, Ref{Ptr{Cvoid}}),
        d_centroids.device, dim, nbits, k, d_centroids.ptr, bucket_cutoffs, length(bucket_cutoffs), h))
    h[]
end
_codec_destroy(h::Ptr{Cvoid}) = ccall((:clb_codec_destroy, libcolbert), Cint, (Ptr{Cvoid},), h)
_codec_compress_device!(h::Ptr{Cvoid}, embs::DeviceBuffer, n::Int, codes::DeviceBuffer, codes_offset::Integer,
    residuals::DeviceBuffer) = _check(ccall((:clb_codec_compress_device, libcolbert), Cint,
    (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{UInt32}, Ptr{UInt8}, Ptr{Cvoid}), h, embs.ptr, n, _at(codes, codes_offset), residuals.ptr, C_NULL))

"_build_ivf (collection_indexer.jl:349-353) over the device array of all codes"
function _build_ivf_device(codes::DeviceBuffer, n::Int, num_partitions::Int)
    d_ivf = DeviceBuffer(8 * max(n, 1); device = codes.device)
    d_len = DeviceBuffer(8 * max(num_partitions, 1); device = codes.device)
    _check(ccall((:clb_build_ivf_device, libcolbert), Cint, (Cint, Ptr{UInt32}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
        codes.device, codes.ptr, n, num_partitions, d_ivf.ptr, d_len.ptr, C_NULL))
    device_download!(Vector{Int}(undef, n), d_ivf), device_download!(Vector{Int}(undef, num_partitions), d_len)
end
