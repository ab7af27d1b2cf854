# ColBERT.jl over libcolbert_hip.so (include/colbert_hip.h) -- the MI355X-native hot path behind the
# reference's public surface: ColBERTConfig, Indexer, index, Searcher, search (JuliaGenAI/ColBERT.jl
# src/ColBERT.jl:21,35,40), so that examples/indexing.jl and examples/searching.jl run unchanged.
#
# What stays Julia: configuration, the index directory (JLD2 + JSON, the reference's file layout), text ->
# token ids, the orchestration of index().  What becomes one `ccall` each: the BERT + Dense forward and its
# epilogues, k-means, codec statistics, compress, the IVF build, and everything search() does after the encoder.
# Flux / Transformers.jl / CUDA.jl are not needed: the weights are read from the flat export that
# tools/export_checkpoint.py writes next to the HuggingFace checkpoint.
#
# This package cannot be executed in the build image (no `julia`); it is kept to marshalling and host glue, and
# every device code path it drives is exercised through the Python ctypes driver over the same ABI.
module ColBERT

using JLD2
using JSON
using Logging
using Random
using Unicode

export ColBERTConfig, Indexer, index, Searcher, search

include("config.jl")
include("capi.jl")
include("storage.jl")
include("tokenizer.jl")
include("checkpoint.jl")
include("indexing.jl")
include("searching.jl")

end # module
