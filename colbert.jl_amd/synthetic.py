"""Deterministic synthetic corpora for parity tests and bench.py.

Search-only corpora are generated directly in compressed form, with the index layout of the
reference's Searcher (src/searching.jl:1-16): centroids (dim,K) fp32, bucket_weights fp32[2^nbits],
codes UInt32[n_emb] (1-based), residuals UInt8 (dim/8*nbits, n_emb), doclens Int64[n_docs],
ivf Int64[n_emb] (1-based embedding ids grouped by centroid, ascending inside a list -- what
`_build_ivf`'s stable sortperm produces, collection_indexer.jl:349-353) and ivf_lengths Int64[K].
Arrays are numpy, column-major where the reference's are matrices.
"""
from __future__ import annotations

import math

import numpy as np

# values logged by the reference's README run (README.md:100)
README_BUCKET_CUTOFFS = np.array([-0.021662371, -0.00015685707, 0.020033525], dtype=np.float32)
README_BUCKET_WEIGHTS = np.array([-0.041035336, -0.009812315, 0.008938393, 0.039779153], dtype=np.float32)


def num_partitions_for(n_docs: int, avg_doclen: float) -> int:
    """The reference's own sizing rule (collection_indexer.jl:122-126), uncapped."""
    est = np.float32(n_docs) * np.float32(avg_doclen)
    return int(2 ** math.floor(math.log2(float(np.float32(16.0) * np.sqrt(est)))))


def build_ivf(codes: np.ndarray, K: int):
    """_build_ivf (collection_indexer.jl:349-353) on the host: stable sort of the codes."""
    ivf = np.argsort(codes, kind="stable").astype(np.int64) + 1
    lens = np.bincount(codes, minlength=K + 1)[1:].astype(np.int64)
    return ivf, lens


def _centroids(seed: int, K: int, dim: int) -> np.ndarray:
    rng = np.random.default_rng([seed, 0])
    cent = rng.standard_normal((K, dim), dtype=np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cent *= rng.uniform(0.6, 1.0, size=(K, 1)).astype(np.float32)   # k-means means are sub-unit-norm
    return np.asfortranarray(cent.T)                                 # (dim, K) column-major


def _block(seed: int, block: int, n_docs: int, K: int, dim: int, nbits: int, doclen_mean: float,
           doclen_std: float, constant_doclen: bool, topical: bool, doclen_max: int = 220):
    """Passages of one generation block: (doclens, codes, residuals as (n_emb, rows))."""
    rng = np.random.default_rng([seed, 1000 + block])
    if constant_doclen:
        doclens = np.full(n_docs, int(doclen_mean), dtype=np.int64)
    else:
        doclens = np.clip(np.rint(doclen_mean + doclen_std * rng.standard_normal(n_docs)), 8, doclen_max).astype(np.int64)
    n_emb = int(doclens.sum())
    if topical:
        doc_of = np.repeat(np.arange(n_docs, dtype=np.int32), doclens)
        topics = rng.integers(0, K, size=(n_docs, 4), dtype=np.int32)
        pick = rng.integers(0, 4, size=n_emb, dtype=np.int32)
        near = (topics[doc_of, pick] + rng.integers(-8, 8, size=n_emb, dtype=np.int32)) % K
        uni = rng.integers(0, K, size=n_emb, dtype=np.int32)
        codes = (np.where(rng.random(n_emb, dtype=np.float32) < 0.8, near, uni) + 1).astype(np.uint32)
    else:
        codes = rng.integers(1, K + 1, size=n_emb, dtype=np.uint32)
    rows = dim // 8 * nbits
    residuals = rng.integers(0, 256, size=(n_emb, rows), dtype=np.uint8)
    return doclens, codes, residuals


def make_index(seed: int, n_docs: int, K: int | None = None, dim: int = 128, nbits: int = 2,
               doclen_mean: float = 80.0, doclen_std: float = 16.0, constant_doclen: bool = False,
               topical: bool = True, n_blocks: int = 1, blocks=None, doclen_max: int = 220,
               ivf_on_device: bool = False):
    """A compressed index of `n_docs` passages.  `topical`: each passage draws 80 % of its tokens
    from the 16 centroids nearest (by id, a cheap stand-in for similarity) to 4 per-passage topic
    centroids, the rest uniformly -- uniform codes are the worst case for candidate counts.

    The collection is generated in `n_blocks` equal blocks of passages, each from its own RNG stream,
    so a passage shard (`blocks` = the block ids it holds) can be generated without the rest and is
    identical to the same passages of the full index.  Returns the Searcher's fields plus
    `pid_offset` (passages before the first generated block)."""
    if K is None:
        K = num_partitions_for(n_docs, doclen_mean)
    per = -(-n_docs // n_blocks)
    blocks = list(range(n_blocks)) if blocks is None else list(blocks)
    parts = []
    for b in blocks:
        lo, hi = b * per, min(n_docs, (b + 1) * per)
        parts.append(_block(seed, b, hi - lo, K, dim, nbits, doclen_mean, doclen_std, constant_doclen, topical,
                            doclen_max))
    doclens = np.concatenate([p[0] for p in parts])
    codes = np.concatenate([p[1] for p in parts])
    residuals = np.asfortranarray(np.concatenate([p[2] for p in parts], axis=0).T)   # (rows, n_emb) col-major
    if nbits == 2:
        weights = README_BUCKET_WEIGHTS.copy()
        cutoffs = README_BUCKET_CUTOFFS.copy()
    else:
        wr = np.random.default_rng([seed, 1])
        weights = np.sort(wr.normal(0, 0.03, 1 << nbits).astype(np.float32))
        cutoffs = ((weights[1:] + weights[:-1]) / 2).astype(np.float32)
    if ivf_on_device:      # 10 M-passage corpora: numpy's stable argsort of 8e8 keys takes minutes, clb_build_ivf seconds
        from . import codec
        ivf, ivf_lengths = codec.build_ivf(codes, K)
    else:
        ivf, ivf_lengths = build_ivf(codes, K)
    return {"dim": dim, "nbits": nbits, "centroids": _centroids(seed, K, dim), "bucket_weights": weights,
            "bucket_cutoffs": cutoffs, "doclens": doclens, "codes": codes, "residuals": residuals,
            "ivf": ivf, "ivf_lengths": ivf_lengths, "pid_offset": blocks[0] * per if blocks else 0}


def make_topic_queries(centroids: np.ndarray, seed: int, n_queries: int, T: int = 32, noise: float = 0.3) -> np.ndarray:
    """(dim, T, n_queries) unit query vectors that need only the (replicated) centroid table: every
    query draws 4 topic centroids like a passage does, its tokens are centroids from their
    neighbourhood plus Gaussian noise, renormalised.  Identical on every rank for a given seed."""
    rng = np.random.default_rng([seed, 2])
    dim, K = centroids.shape
    topics = rng.integers(0, K, size=(n_queries, 4))
    pick = rng.integers(0, 4, size=(n_queries, T))
    cid = (np.take_along_axis(topics, pick, axis=1) + rng.integers(-8, 8, size=(n_queries, T))) % K
    q = centroids[:, cid.ravel()].astype(np.float64)
    q /= np.linalg.norm(q, axis=0, keepdims=True)
    q += noise * rng.standard_normal(q.shape) / math.sqrt(dim)
    q /= np.linalg.norm(q, axis=0, keepdims=True)
    return np.asfortranarray(q.reshape(dim, n_queries, T).transpose(0, 2, 1).astype(np.float32))


def decompress_numpy(index: dict, eids0: np.ndarray) -> np.ndarray:
    """fp64-accurate decompression of a few embeddings (0-based ids) -- used only to build queries."""
    dim, nbits = index["dim"], index["nbits"]
    res = index["residuals"][:, eids0]
    per = 8 // nbits
    idx = np.stack([(res >> (nbits * i)) & ((1 << nbits) - 1) for i in range(per)], axis=1).reshape(dim, -1)
    x = index["centroids"][:, index["codes"][eids0].astype(np.int64) - 1] + index["bucket_weights"][idx]
    return x / (np.linalg.norm(x, axis=0, keepdims=True) + np.finfo(np.float32).eps)


def make_queries(index: dict, seed: int, n_queries: int, T: int = 32, noise: float = 0.3) -> np.ndarray:
    """(dim, T, n_queries) unit vectors: decompressed tokens of a random passage plus Gaussian
    noise, renormalised (gives realistic score gaps)."""
    rng = np.random.default_rng(seed)
    dim = index["dim"]
    off = np.concatenate([[0], np.cumsum(index["doclens"])])
    out = np.empty((dim, T, n_queries), dtype=np.float32, order="F")
    docs = rng.integers(0, index["doclens"].size, size=n_queries)
    for j, p in enumerate(docs):
        eids = off[p] + rng.integers(0, index["doclens"][p], size=T)
        q = decompress_numpy(index, eids) + noise * rng.standard_normal((dim, T)) / math.sqrt(dim)
        q /= np.linalg.norm(q, axis=0, keepdims=True)
        out[:, :, j] = q.astype(np.float32)
    return out


def make_embeddings(seed: int, n_docs: int, dim: int = 128, doclen_mean: float = 80.0,
                    doclen_std: float = 16.0, n_components: int = 4096):
    """Config-2 style input for the index build: unit vectors from a Gaussian mixture
    (centre + 0.25 * Gaussian, normalised).  Returns (embs (dim, n_emb) col-major, doclens)."""
    rng = np.random.default_rng(seed)
    doclens = np.clip(np.rint(doclen_mean + doclen_std * rng.standard_normal(n_docs)), 8, 220).astype(np.int64)
    n_emb = int(doclens.sum())
    centres = rng.standard_normal((n_components, dim), dtype=np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    comp = rng.integers(0, n_components, size=n_emb)
    x = centres[comp] + 0.25 * rng.standard_normal((n_emb, dim), dtype=np.float32) / np.float32(math.sqrt(dim)) * np.float32(math.sqrt(dim) / 4)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.asfortranarray(x.T.astype(np.float32)), doclens


class DeviceMixtureSource:
    """`make_embeddings` for collections whose fp32 embeddings do not fit host memory: the same 4096-component mixture
    (centre + 0.25 * Gaussian, normalised), generated ON THE DEVICE in fixed blocks of `block` passages, each from its
    own counter-based stream -- any passage range can be regenerated bit for bit (indexer.DeviceEmbeddingSource).
    torch is used as the random-number generator of the synthetic INPUT only."""

    def __init__(self, seed: int, n_docs: int, device, dim: int = 128, doclen_mean: float = 80.0, doclen_std: float = 16.0,
                 n_components: int = 4096, block: int = 25000):
        import torch
        self.seed, self.n_docs, self.dim, self.block = seed, n_docs, dim, block
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        rng = np.random.default_rng(seed)
        self.doclens = np.clip(np.rint(doclen_mean + doclen_std * rng.standard_normal(n_docs)), 8, 220).astype(np.int64)
        self.off = np.concatenate([[0], np.cumsum(self.doclens)])
        centres = rng.standard_normal((n_components, dim), dtype=np.float32)
        centres /= np.linalg.norm(centres, axis=1, keepdims=True)
        self.centres = torch.from_numpy(centres).to(self.device)

    def _block(self, b: int):
        import torch
        lo, hi = b * self.block, min(self.n_docs, (b + 1) * self.block)
        n = int(self.off[hi] - self.off[lo])
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 1_000_003 + b)
        comp = torch.randint(0, self.centres.shape[0], (n,), generator=g, device=self.device)
        x = torch.randn((n, self.dim), generator=g, device=self.device, dtype=torch.float32)
        x.mul_(0.25).add_(self.centres[comp])
        x.div_(torch.linalg.vector_norm(x, dim=1, keepdim=True))
        return x

    def chunk(self, start: int, end: int):
        """(n_emb, dim) float32 CUDA tensor of passages start..end-1"""
        import torch
        parts = []
        for b in range(start // self.block, (max(end, start + 1) - 1) // self.block + 1):
            lo = b * self.block
            x = self._block(b)
            a = int(self.off[max(start, lo)] - self.off[lo])
            z = int(self.off[min(end, lo + self.block, self.n_docs)] - self.off[lo])
            parts.append(x[a:z])
        return parts[0].contiguous() if len(parts) == 1 else torch.cat(parts)
