"""colbert.jl_amd -- MI355X-native late-interaction retrieval hot path behind ColBERT.jl's API.

Mirrors the reference's public surface (src/ColBERT.jl:21,35,40): ColBERTConfig, Indexer, index,
Searcher, search.  All compute goes through the C-ABI library libcolbert_hip.so
(include/colbert_hip.h, built from colbert.jl_amd/csrc); there is no CPU fallback.
"""
from . import synthetic  # noqa: F401
