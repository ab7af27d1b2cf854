"""Index directory I/O with the reference's layout: JLD2 (HDF5 subset) + JSON files (src/savers.jl,
src/loaders.jl, src/indexing.jl:82-85,140-143).  Arrays go through `jld2.save_object` / `jld2.load_object`, the
Python counterparts of `JLD2.save_object` / `JLD2.load_object` (see jld2.py for what is and is not verified).

Files: config.json, plan.json, centroids / bucket_cutoffs / bucket_weights /
avg_residual, <i>.codes, <i>.residuals, doclens.<i>, <i>.metadata.json, ivf, ivf_lengths (chunks 1-based)."""
from __future__ import annotations

import json
import os

import numpy as np

from . import jld2

EXT = ".jld2"


def _save(path: str, a) -> None:
    jld2.save_object(path + EXT, a)


def _load(path: str):
    """`<path>.jld2` (the reference's format); an index directory written by round 1 of this package holds `<path>.npy`
    instead and is still read."""
    if not os.path.isfile(path + EXT) and os.path.isfile(path + ".npy"):
        a = np.load(path + ".npy")
        return a if a.ndim == 0 else np.asfortranarray(a)
    return jld2.load_object(path + EXT)


def save_codec(index_path, centroids, bucket_cutoffs, bucket_weights, avg_residual) -> None:
    """save_codec (savers.jl:16-29)"""
    os.makedirs(index_path, exist_ok=True)
    _save(os.path.join(index_path, "centroids"), np.asfortranarray(centroids, dtype=np.float32))
    _save(os.path.join(index_path, "bucket_cutoffs"), np.asarray(bucket_cutoffs, dtype=np.float32))
    _save(os.path.join(index_path, "bucket_weights"), np.asarray(bucket_weights, dtype=np.float32))
    _save(os.path.join(index_path, "avg_residual"), np.float32(avg_residual))


def save_chunk(index_path, codes, residuals, chunk_idx: int, passage_offset: int, doclens) -> None:
    """save_chunk (savers.jl:52-84); chunk_idx and passage_offset are 1-based"""
    _save(os.path.join(index_path, f"{chunk_idx}.codes"), np.asarray(codes, dtype=np.uint32))
    _save(os.path.join(index_path, f"{chunk_idx}.residuals"), np.asfortranarray(residuals, dtype=np.uint8))
    _save(os.path.join(index_path, f"doclens.{chunk_idx}"), np.asarray(doclens, dtype=np.int64))
    with open(os.path.join(index_path, f"{chunk_idx}.metadata.json"), "w") as f:
        json.dump({"passage_offset": int(passage_offset), "num_passages": int(len(doclens)),
                   "num_embeddings": int(len(codes))}, f, indent=4)


def save_json(index_path, name, obj) -> None:
    with open(os.path.join(index_path, name), "w") as f:
        json.dump(obj, f, indent=4)


def load_json(index_path, name):
    with open(os.path.join(index_path, name)) as f:
        return json.load(f)


def check_all_files_are_saved(index_path: str) -> bool:
    """_check_all_files_are_saved (collection_indexer.jl:299-340)"""
    if not os.path.isfile(os.path.join(index_path, "plan.json")):
        return False
    plan = load_json(index_path, "plan.json")
    files = ["config.json"] + [s + EXT for s in ("centroids", "bucket_cutoffs", "bucket_weights", "avg_residual",
                                                  "ivf", "ivf_lengths")]
    for i in range(1, plan["num_chunks"] + 1):
        files += [f"{i}.codes{EXT}", f"{i}.residuals{EXT}", f"doclens.{i}{EXT}", f"{i}.metadata.json"]
    return all(os.path.isfile(os.path.join(index_path, f)) for f in files)


def load_index(index_path: str) -> dict:
    """load_codec / load_doclens / load_compressed_embs (loaders.jl:10-38, 76-113) + ivf files."""
    plan = load_json(index_path, "plan.json")
    cfg = load_json(index_path, "config.json")
    codes, res, dl = [], [], []
    for i in range(1, plan["num_chunks"] + 1):
        codes.append(_load(os.path.join(index_path, f"{i}.codes")))
        res.append(_load(os.path.join(index_path, f"{i}.residuals")))
        dl.append(_load(os.path.join(index_path, f"doclens.{i}")))
    return {"dim": cfg["dim"], "nbits": cfg["nbits"],
            "centroids": _load(os.path.join(index_path, "centroids")),
            "bucket_cutoffs": _load(os.path.join(index_path, "bucket_cutoffs")),
            "bucket_weights": _load(os.path.join(index_path, "bucket_weights")),
            "avg_residual": np.float32(_load(os.path.join(index_path, "avg_residual"))),
            "codes": np.concatenate(codes), "residuals": np.asfortranarray(np.concatenate(res, axis=1)),
            "doclens": np.concatenate(dl), "ivf": _load(os.path.join(index_path, "ivf")),
            "ivf_lengths": _load(os.path.join(index_path, "ivf_lengths"))}
