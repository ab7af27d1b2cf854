"""ColBERTConfig -- field names and defaults of the reference's `Base.@kwdef struct ColBERTConfig`
(src/infra/config.jl:54-90), persisted as config.json exactly like src/savers.jl:110-121."""
from __future__ import annotations

import dataclasses
import json
import os
from typing import List, Optional, Union


@dataclasses.dataclass(frozen=True)
class ColBERTConfig:
    # run settings
    use_gpu: bool = False
    rank: int = 0
    nranks: int = 1
    # tokenization settings
    query_token_id: str = "[unused0]"
    doc_token_id: str = "[unused1]"
    query_token: str = "[Q]"
    doc_token: str = "[D]"
    # resource settings
    checkpoint: str = "colbert-ir/colbertv2.0"
    collection: Union[str, List[str]] = ""
    # doc settings
    dim: int = 128
    doc_maxlen: int = 300
    mask_punctuation: bool = True
    # query settings
    query_maxlen: int = 32
    attend_to_mask_tokens: bool = False
    # indexing settings
    index_path: str = ""
    index_bsize: int = 64
    chunksize: Optional[int] = 25000          # `missing` in Julia <-> None
    passages_batch_size: int = 5000
    nbits: int = 2
    kmeans_niters: int = 20
    # search settings
    nprobe: int = 2
    ncandidates: int = 8192

    def save(self, index_path: Optional[str] = None) -> str:
        path = os.path.join(index_path or self.index_path, "config.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dataclasses.asdict(self), f, indent=4)
        return path

    @staticmethod
    def load(index_path: str) -> "ColBERTConfig":
        with open(os.path.join(index_path, "config.json")) as f:
            raw = json.load(f)
        names = {f.name for f in dataclasses.fields(ColBERTConfig)}
        return ColBERTConfig(**{k: v for k, v in raw.items() if k in names})
