"""Indexer / index -- the reference's src/indexing.jl orchestration over the C ABI.

The encoder is injected (`encoder.encode_passages(list[str]) -> (embs (dim, n), doclens)`); the stages
after it -- sample, held-out split, plan, k-means + codec statistics, compress, IVF -- follow
src/indexing.jl:63-147 and run on the device.  RNG-dependent draws (_sample_pids, _heldout_split,
k-means initialisation) use numpy's generator: they cannot match Julia's Xoshiro stream."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import codec, storage
from .config import ColBERTConfig


class PrecomputedEncoder:
    """Stands in for the BERT checkpoint: passages are indices into precomputed embeddings."""

    def __init__(self, embs: np.ndarray, doclens: np.ndarray):
        self.embs = np.asfortranarray(embs, dtype=np.float32)
        self.doclens = np.asarray(doclens, dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.doclens)])

    def encode_passages(self, passage_ids):
        ids = np.asarray(passage_ids, dtype=np.int64)
        cols = np.concatenate([np.arange(self.offsets[p], self.offsets[p + 1]) for p in ids]) if ids.size else np.zeros(0, np.int64)
        return np.asfortranarray(self.embs[:, cols]), self.doclens[ids].copy()


class Indexer:
    """struct Indexer (indexing.jl:1-8).  `collection` is a list of passages (strings for a text
    encoder, passage indices for PrecomputedEncoder)."""

    def __init__(self, config: ColBERTConfig, encoder=None, collection=None, device: int = 0, seed: int = 0):
        self.config = config
        self.encoder = encoder
        if collection is None:
            if isinstance(config.collection, str) and config.collection:
                with open(config.collection) as f:           # readlines (indexing.jl:27)
                    collection = [ln.rstrip("\n") for ln in f]
            else:
                collection = list(config.collection)
        self.collection = collection
        self.device = device
        self.rng = np.random.default_rng(seed)


def train(sample, heldout, num_partitions: int, nbits: int, kmeans_niters: int, rng, device: int = 0):
    """train (collection_indexer.jl:219-237) -> (centroids, bucket_cutoffs, bucket_weights, avg_residual)"""
    sample = np.asfortranarray(sample, dtype=np.float32)
    init = sample[:, rng.permutation(sample.shape[1])[:num_partitions]]
    centroids, _, _ = codec.kmeans(sample, init, max_iters=kmeans_niters, device=device)
    cut, w, avg, _ = codec.compute_avg_residuals(nbits, centroids, heldout, device=device)
    return centroids, cut, w, avg


def index(indexer: Indexer) -> Optional[str]:
    """index(indexer) (indexing.jl:63-147)."""
    cfg = indexer.config
    path = cfg.index_path
    if os.path.isdir(path):                                  # indexing.jl:64-67
        return None
    n_docs = len(indexer.collection)
    # sample -> embeddings (collection_indexer.jl:17-24, 56-79)
    n_s = codec.num_sampled_pids(n_docs)
    sampled = np.unique(indexer.rng.integers(0, n_docs, size=n_s))
    sample, sample_doclens = indexer.encoder.encode_passages([indexer.collection[i] for i in sampled])
    avg_doclen_est = float(np.float32(sample_doclens.sum() / max(len(sample_doclens), 1)))
    # held-out split (collection_indexer.jl:81-91)
    sample = np.asfortranarray(sample[:, indexer.rng.permutation(sample.shape[1])])
    h = codec.heldout_size(sample.shape[1])
    sample, heldout = sample[:, : sample.shape[1] - h], sample[:, sample.shape[1] - h:]
    os.makedirs(path)
    storage._save(os.path.join(path, "sample"), sample); storage._save(os.path.join(path, "sample_heldout"), heldout)
    plan = codec.setup(n_docs, avg_doclen_est, sample.shape[1], cfg.chunksize, cfg.nranks)
    storage.save_json(path, "plan.json", plan)
    cfg.save(path)
    centroids, cut, w, avg = train(sample, heldout, plan["num_partitions"], cfg.nbits, cfg.kmeans_niters,
                                   indexer.rng, indexer.device)
    storage.save_codec(path, centroids, cut, w, avg)
    # chunk loop (collection_indexer.jl:271-297)
    counts, all_codes = [], []
    for ci, start in enumerate(range(0, n_docs, plan["chunksize"]), start=1):
        end = min(n_docs, start + plan["chunksize"])
        embs, doclens = indexer.encoder.encode_passages(indexer.collection[start:end])
        codes, res = codec.compress(centroids, cut, cfg.dim, cfg.nbits, embs, device=indexer.device)
        storage.save_chunk(path, codes, res, ci, start + 1, doclens)
        counts.append(len(codes)); all_codes.append(codes)
    total, offsets = codec.collect_embedding_id_offset(counts)   # indexing.jl:119-132
    plan["num_embeddings"] = total
    plan["embeddings_offsets"] = [int(o) for o in offsets]
    storage.save_json(path, "plan.json", plan)
    for ci, off in enumerate(offsets[: len(counts)], start=1):
        meta = storage.load_json(path, f"{ci}.metadata.json")
        meta["embedding_offset"] = int(off)
        storage.save_json(path, f"{ci}.metadata.json", meta)
    ivf, ivf_lengths = codec.build_ivf(np.concatenate(all_codes), plan["num_partitions"], device=indexer.device)
    storage._save(os.path.join(path, "ivf"), ivf); storage._save(os.path.join(path, "ivf_lengths"), ivf_lengths)
    assert storage.check_all_files_are_saved(path)
    return path
