"""colbert.jl_amd -- MI355X-native late-interaction retrieval hot path behind ColBERT.jl's API.

Mirrors the reference's public surface (src/ColBERT.jl:21,35,40): ColBERTConfig, Indexer, index,
Searcher, search.  All compute goes through the C-ABI library libcolbert_hip.so
(include/colbert_hip.h, built from colbert.jl_amd/csrc); there is no CPU fallback.
"""
from . import codec, storage, synthetic, tokenization  # noqa: F401
from ._lib import (ArgumentError, BoundsError, ColBERTError, DimensionMismatch, DomainError, HipError,  # noqa: F401
                   Unsupported, build, declared_symbols, lib)
from .config import ColBERTConfig  # noqa: F401
from .encoder import BertEncoder  # noqa: F401
from .indexer import Indexer, PrecomputedEncoder, index, train  # noqa: F401
from .searcher import Searcher, TextSearch, search  # noqa: F401
