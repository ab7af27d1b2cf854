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


def index(indexer: Indexer, device_resident: Optional[bool] = None) -> Optional[str]:
    """index(indexer) (indexing.jl:63-147).  With an encoder that can leave its embeddings on the device
    (BertEncoder.doc_embeddings_device) the build runs through index_device -- no embedding crosses PCIe -- and writes the
    same files (`device_resident=False` forces the host-buffer route; both draw the same numbers from indexer.rng): byte for
    byte when the encoder pads its batches, up to the rounding of different tile plans when it packs them (head size 64 with
    the f16x3 Linear layers: the device route packs 256 passages per call, the host entry point one batch)."""
    cfg = indexer.config
    path = cfg.index_path
    if os.path.isdir(path):                                  # indexing.jl:64-67
        return None
    if device_resident is None:
        device_resident = hasattr(indexer.encoder, "doc_embeddings_device") and _device_route_fits(indexer)
    if device_resident:
        created = [False]               # set once THIS call has made the directory (os.makedirs raises if it already exists)
        try:
            return _index_through_device(indexer, created)
        except BaseException:
            # a build that dies half-way must not leave a directory behind: index() would take it for a finished index
            # (the isdir test above, indexing.jl:64-67) and return without building -- but only a directory this call made:
            # one that another process or rank created between the isdir test and makedirs is not ours to delete
            if created[0]:
                import shutil
                shutil.rmtree(path, ignore_errors=True)
            raise
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


def _device_route_fits(indexer: Indexer) -> bool:
    """Whether the device-resident route's footprint fits the free HBM: it keeps the codes of the WHOLE collection for the
    IVF (4 B per embedding + 8 B of Int64 ivf + ~16 B of sort scratch), one chunk's embeddings (512 B each) and residuals, and
    the training sample.  Upper bound from doc_maxlen tokens per passage; the host route streams chunk by chunk instead."""
    cfg = indexer.config
    n_docs = len(indexer.collection)
    maxlen = int(getattr(indexer.encoder.config, "doc_maxlen", cfg.doc_maxlen))
    chunk = int(cfg.chunksize or min(25000, 1 + n_docs))
    need = n_docs * maxlen * 28 + min(chunk, n_docs) * maxlen * (2 * 4 * cfg.dim + cfg.dim // 8 * cfg.nbits) \
        + codec.num_sampled_pids(n_docs) * maxlen * 4 * cfg.dim * 3
    try:
        free, _ = codec.device_memory(indexer.device)
    except Exception:
        return True
    return need < 0.8 * free


class _BackgroundWriter:
    """The index directory's files are written by ONE worker thread while the device goes on (k-means runs while the 7-GB sample
    is transposed and written, chunk i + 1 is encoded while chunk i's codes and residuals go to disk): a callback hands over
    device tensors that nobody modifies any more (each chunk's residuals are their own allocation, the codes a slice of the
    final array, both complete -- the caller has waited for the device), the worker copies them to the host and saves them.
    At most `depth` jobs wait (a job holds a chunk's tensors alive).  The first exception is re-raised by finish()."""

    def __init__(self, depth: int = 4):
        import queue
        import threading
        self.q = queue.Queue(maxsize=depth)
        self.error = None
        self.busy_s = 0.0
        self.t = threading.Thread(target=self._run, name="colbert-index-writer", daemon=True)
        self.t.start()

    def _run(self):
        import time
        while True:
            job = self.q.get()
            if job is None:
                return
            if self.error is None:
                t0 = time.time()
                try:
                    job()
                except BaseException as e:                   # noqa: BLE001 -- handed to the caller's thread by finish()
                    self.error = e
                self.busy_s += time.time() - t0

    def submit(self, job):
        if self.error is not None:
            self.finish()
        if not self.t.is_alive():
            raise RuntimeError("the index writer has been stopped")
        self.q.put(job)

    def finish(self):
        """Drain the queue, stop the worker, raise what it caught."""
        if self.t.is_alive():
            self.q.put(None)
            self.t.join()
        if self.error is not None:
            e, self.error = self.error, None
            raise e


def _index_through_device(indexer: Indexer, created=None) -> str:
    """index() over index_device: the encoder's output, the sample, the codes and the residuals stay in HBM; what is
    written is the reference's directory (sample, sample_heldout, plan.json, config.json, the codec, per chunk codes /
    residuals / doclens / metadata, ivf, ivf_lengths -- indexing.jl:84-147)."""
    import time
    cfg = indexer.config
    path = cfg.index_path
    t0 = time.time()
    source = EncoderSource(indexer.encoder, indexer.collection, indexer.device, lazy=True)    # the tokenizer overlaps the device
    t_tok = time.time() - t0
    os.makedirs(path)                   # FileExistsError if somebody else made it meanwhile: propagates, nothing is removed
    if created is not None:
        created[0] = True
    state = {"write_s": 0.0}
    writer = _BackgroundWriter()

    def on_sample(sample, heldout, plan):
        t1 = time.time()
        storage.save_json(path, "plan.json", plan)
        cfg.save(path)
        state["plan"] = plan

        def job():
            storage._save(os.path.join(path, "sample"), np.asfortranarray(sample.cpu().numpy().T))
            storage._save(os.path.join(path, "sample_heldout"), np.asfortranarray(heldout.cpu().numpy().T))
        writer.submit(job)
        state["write_s"] += time.time() - t1

    counts = []

    def on_codec(centroids, cut, w, avg):
        t1 = time.time()
        writer.submit(lambda: storage.save_codec(path, np.asfortranarray(centroids.cpu().numpy().T), cut, w, avg))
        state["write_s"] += time.time() - t1

    def on_chunk(ci, start, end, codes, residuals, doclens):
        # a chunk's files are written as soon as it is compressed (collection_indexer.jl:271-297): a failure later on leaves
        # nothing half-described, and only the codes (for the IVF) outlive the chunk on the device
        t1 = time.time()
        writer.submit(lambda: storage.save_chunk(path, codes.cpu().numpy().view(np.uint32), np.asfortranarray(residuals.cpu().numpy().T),
                                                 ci, start + 1, doclens))
        counts.append(int(codes.numel()))
        state["write_chunks_s"] = state.get("write_chunks_s", 0.0) + time.time() - t1

    try:
        ix, rec = index_device(source, nbits=cfg.nbits, kmeans_niters=cfg.kmeans_niters, chunksize=cfg.chunksize, rng=indexer.rng,
                               nranks=cfg.nranks, on_sample=on_sample, on_codec=on_codec, on_chunk=on_chunk, keep_residuals=False)
    finally:
        t_w = time.time()
        writer.finish()                  # every chunk's files exist (or the writer's exception surfaces) before the metadata pass
    plan = state["plan"]
    total, offsets = codec.collect_embedding_id_offset(counts)
    plan["num_embeddings"] = total
    plan["embeddings_offsets"] = [int(o) for o in offsets]
    storage.save_json(path, "plan.json", plan)
    for ci, o in enumerate(offsets[: len(counts)], start=1):
        meta = storage.load_json(path, f"{ci}.metadata.json")
        meta["embedding_offset"] = int(o)
        storage.save_json(path, f"{ci}.metadata.json", meta)
    storage._save(os.path.join(path, "ivf"), ix["ivf"].cpu().numpy())
    storage._save(os.path.join(path, "ivf_lengths"), np.asarray(ix["ivf_lengths"]))
    assert storage.check_all_files_are_saved(path)
    # seconds per stage of this build (tools/bench_index_with_encoder.py): sample_and_split_s and chunks_s are encoder time
    rec.update({"tokenize_s": round(t_tok, 3), "write_sample_s": round(state["write_s"], 3),
                "write_index_s": round(time.time() - t_w + state.get("write_chunks_s", 0.0), 3),
                "writer_thread_busy_s": round(writer.busy_s, 3),
                "sampled_passages_not_encoded_twice": int(getattr(source, "reused_passages", 0))})
    indexer.last_build_record = rec
    return path


# ---- the same build with every large array resident in HBM ----------------------------------------------------------
class DeviceEmbeddingSource:
    """Stands in for the BERT checkpoint when the collection's fp32 embeddings do not fit host memory (1 M passages:
    80 M x 128 fp32 = 41 GB): `doclens` (Int64[n_docs], host) and `chunk(start, end)` -> the (n_emb, dim) float32 CUDA
    tensor of passages start..end-1 (0-based, end exclusive), i.e. what `encode_passages` returns, left on the device.
    `chunk` must be reproducible: index_device() reads every passage twice (sample, then compress), as the reference
    encodes its sample and then every chunk (collection_indexer.jl:56-79, 271-297)."""

    dim = 128
    doclens: np.ndarray

    def chunk(self, start: int, end: int):
        raise NotImplementedError

    def sample(self, pids):
        """The embeddings of the passages `pids` (sorted, 0-based), (n, dim) on the device.  Default: cut them out of the
        chunks they fall in; a source that can encode a list of passages directly overrides this (encode only the sample,
        as the reference does: collection_indexer.jl:56-79)."""
        return _sample_from_chunks(self, pids)


def _sample_from_chunks(source, pids, step: Optional[int] = None):
    import torch
    pids = np.asarray(pids, dtype=np.int64)
    doclens = np.asarray(source.doclens, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(doclens)])
    n_docs = doclens.size
    step = int(step or min(25000, 1 + n_docs))
    out = torch.empty((int(doclens[pids].sum()), source.dim), dtype=torch.float32, device=source.device)
    fill = 0
    for start in range(0, n_docs, step):
        end = min(n_docs, start + step)
        mine = pids[(pids >= start) & (pids < end)]
        if mine.size == 0:
            continue
        x = source.chunk(start, end)
        rows = np.concatenate([np.arange(off[p] - off[start], off[p + 1] - off[start]) for p in mine])
        codec.gather_rows_device(x.contiguous(), rows, out=out[fill:fill + rows.size])
        fill += rows.size
        del x
    assert fill == out.shape[0]
    return out


def _chunk_source_rows(known, doclens, known_first_row, n_cached_rows):
    """Row of the buffer [cached sample rows | this chunk's newly encoded rows] behind every embedding of a chunk, in passage order:
    passage j (doclens[j] rows) comes from the cache at known_first_row[its rank among the known ones] if known[j], else from the
    tail (the new passages back to back, in order, from row n_cached_rows)."""
    known = np.asarray(known, dtype=bool)
    dl = np.asarray(doclens, dtype=np.int64)
    src = np.empty(known.size, dtype=np.int64)                          # first source row of every passage
    src[known] = np.asarray(known_first_row, dtype=np.int64)
    src[~known] = n_cached_rows + np.concatenate([[0], np.cumsum(dl[~known])[:-1]]) if (~known).any() else 0
    dst = np.concatenate([[0], np.cumsum(dl)[:-1]]) if dl.size else np.zeros(0, np.int64)
    return np.repeat(src - dst, dl) + np.arange(int(dl.sum()), dtype=np.int64)


class EncoderSource(DeviceEmbeddingSource):
    """The BERT checkpoint as a device-resident source.  The collection is tokenised ONCE on the host (tensorize_docs in
    batches of index_bsize, checkpoint.jl:159-189); the token ids of every passage are kept (a few hundred bytes per
    passage), so `doclens` -- a passage keeps the tokens that are attended to and not in the skiplist (checkpoint.jl:37-43),
    which tokenisation alone decides -- needs no forward pass, and encoding a batch is padding + one upload + the device
    chain forward -> mask -> normalise -> compact (BertEncoder.doc_embeddings_device = clb_encode_docs_device) without a
    read-back: the host prepares batch i + 1 while the device encodes batch i."""

    def __init__(self, encoder, collection, device: int = 0, packed: bool = True, pack_batches: int = 4, lazy: bool = False):
        import torch
        self.packed = packed          # batches without padding rows (clb_encode_docs_packed_device) where the encoder can
        self.pack_batches = int(os.environ.get("COLBERT_PACK_BATCHES", pack_batches))
        # packed calls are filled by ROWS, not by passages: 170 row tiles of 256 = 43 520 rows give the 768-wide outputs 510 tiles
        # of 256 x 256 -- two full rounds of the 256 CUs -- and the wide ones 5.98 / 7.97 x 2 rounds (a 256-passage call of
        # ~22 390 rows left 8 of 264 tiles for a second round: 0.79 us per row against 0.65)
        self.pack_rows = int(os.environ.get("COLBERT_PACK_ROWS", 170 * 256))
        self.encoder = encoder
        self.collection = collection
        self.n_docs = len(collection)
        self.dim = encoder.dim
        self.device = torch.device("cuda", device)
        cfg = encoder.config
        self.skiplist = np.asarray(encoder.tokenizer.doc_skiplist(cfg.mask_punctuation), dtype=np.int64)
        self._d_skip = torch.from_numpy(self.skiplist).to(self.device)
        self._pad = np.int32(encoder.tokenizer.pad_id)
        self._marker = np.int32(encoder.tokenizer.lookup(cfg.doc_token_id))
        self._maxlen = int(cfg.doc_maxlen)
        self._tokens = [None] * self.n_docs
        self._doclens = None
        self._worker = None
        self.lazy = bool(lazy)
        # lazy (index_device): only the sampled passages are tokenised before the device starts on the sample
        # (prepare_sample); the rest follows on a host thread while the device encodes the sample and runs k-means
        if not lazy:
            self._tokenize(range(self.n_docs))

    def _tokenize(self, pids):
        """Per passage its column of tensorize_docs (doc_tokenization.jl:143-156), attended rows only -- [CLS] [D] w1 .. wn
        [SEP], 1-based ids, truncated to doc_maxlen -- in large slices on the tokenizer's thread pool (the GIL is released)."""
        pids = list(pids)
        tok = self.encoder.tokenizer.tok
        for start in range(0, len(pids), 8192):
            part = pids[start:start + 8192]
            for p, e in zip(part, tok.encode_batch([self.collection[p] for p in part], add_special_tokens=True)):
                ids = np.asarray(e.ids[:self._maxlen - 1], dtype=np.int32) + 1
                self._tokens[p] = np.concatenate([ids[:1], [self._marker], ids[1:]]).astype(np.int32)

    def _tokenize_in_worker_process(self, pids):
        """The passages `pids` through colbert.jl_amd/_tok_worker.py (a child process: the conversion loop of _tokenize holds the
        interpreter lock, and on a thread it starved the thread that feeds the device).  This thread only moves bytes -- pipe
        reads and writes release the lock.  Falls back to _tokenize in this thread if the worker cannot be started or dies."""
        import struct
        import subprocess
        import sys
        tok = self.encoder.tokenizer
        vocab_file = getattr(tok, "vocab_file", None)
        done = 0
        proc = None
        import tempfile
        errlog = tempfile.TemporaryFile()       # the worker's stderr: read only if it fails (a pipe nobody drains could block it)
        try:
            if vocab_file is None or os.environ.get("COLBERT_TOKENIZER_PROCESS", "1") == "0":
                raise OSError("no vocabulary file to hand to a worker process")
            worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tok_worker.py")
            proc = subprocess.Popen([sys.executable, worker, vocab_file, "1" if getattr(tok, "lowercase", True) else "0",
                                     str(self._maxlen), str(int(self._marker))], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                    stderr=errlog)

            def read_exact(n):
                buf = proc.stdout.read(n)
                if len(buf) != n:
                    raise OSError("tokenizer worker closed its pipe")
                return buf
            for start in range(0, len(pids), 8192):
                part = pids[start:start + 8192]
                enc = [self.collection[p].encode("utf-8") for p in part]
                proc.stdin.write(struct.pack("<II", len(part), sum(len(b) for b in enc)))
                proc.stdin.write(np.array([len(b) for b in enc], dtype=np.uint32).tobytes())
                proc.stdin.write(b"".join(enc))
                proc.stdin.flush()
                n, ntok = struct.unpack("<II", read_exact(8))
                toklen = np.frombuffer(read_exact(4 * n), dtype=np.int32)
                flat = np.frombuffer(read_exact(4 * ntok), dtype=np.int32)
                assert n == len(part) and int(toklen.sum()) == ntok
                off = np.concatenate([[0], np.cumsum(toklen)])
                for j, p in enumerate(part):
                    self._tokens[p] = flat[off[j]:off[j + 1]]
                done = start + len(part)
            proc.stdin.close()
            proc.wait(timeout=30)
        except Exception as e:      # noqa: BLE001 -- whatever went wrong with the worker, tokenising here is always safe
            why = f"{type(e).__name__}: {e}"
            if proc is not None:
                proc.kill()
                try:
                    proc.wait(timeout=10)
                    errlog.seek(0)
                    err = errlog.read().decode(errors="replace").strip().splitlines()
                    if err:
                        why += f" (worker: {err[-1][:200]})"
                except Exception:   # noqa: BLE001
                    pass
            if os.environ.get("COLBERT_TOKENIZER_PROCESS", "1") != "0" and vocab_file is not None:
                print(f"[colbert] tokenizer worker unavailable, tokenising {len(pids) - done} passages in this process: {why}", file=sys.stderr)
            self._tokenize(pids[done:])
        finally:
            if proc is not None:        # no zombie, no open descriptors, whichever way the worker ended
                for pipe in (proc.stdin, proc.stdout):
                    try:
                        if pipe is not None:
                            pipe.close()
                    except Exception:   # noqa: BLE001
                        pass
                try:
                    proc.wait(timeout=10)
                except Exception:       # noqa: BLE001
                    pass
            errlog.close()

    def prepare_sample(self, pids):
        """Tokenise the passages `pids` now and everything else in the background (a worker process fed by a thread)."""
        import threading
        if not self.lazy or self._doclens is not None or self._worker is not None:
            return                       # everything is (being) tokenised already
        self._tokenize(int(p) for p in pids)
        done = set(int(p) for p in pids)
        self._worker = threading.Thread(target=self._tokenize_in_worker_process, args=([p for p in range(self.n_docs) if p not in done],),
                                        daemon=True)
        self._worker.start()

    def _doclens_of(self, pids):
        """Kept tokens (attended and not in the skiplist, checkpoint.jl:37-43) of the passages `pids` -- from their tokens alone."""
        toks = [self._tokens[int(p)] for p in pids]
        if not toks:
            return np.zeros(0, np.int64)
        starts = np.concatenate([[0], np.cumsum([t.size for t in toks])[:-1]])        # every passage holds [CLS] [D] [SEP]: no empty segment
        return np.add.reduceat((~np.isin(np.concatenate(toks), self.skiplist)).astype(np.int64), starts)

    @property
    def doclens(self):
        if self._doclens is None:
            if self._worker is not None:
                self._worker.join()
                self._worker = None
            missing = [p for p in range(self.n_docs) if self._tokens[p] is None]
            if missing:
                self._tokenize(missing)
            self._doclens = self._doclens_of(range(self.n_docs))
        return self._doclens

    def _tensorize(self, pids):
        """What tensorize_docs returns for the passages `pids` as one batch, transposed to the device layout: ids int32
        (N, L) padded with [PAD] to the longest passage of the batch, mask uint8 (N, L)."""
        toks = [self._tokens[int(p)] for p in pids]
        L = max(t.size for t in toks)
        ids = np.full((len(toks), L), self._pad, dtype=np.int32)
        mask = np.zeros((len(toks), L), dtype=np.uint8)
        for j, t in enumerate(toks):
            ids[j, :t.size] = t
            mask[j, :t.size] = 1
        return ids, mask

    def _encode_packed(self, batch, dst, n_out):
        """The batch with its passages back to back and no padding rows: tensorize_docs pads to the batch's longest passage
        with [PAD] tokens that the attention mask hides and the skiplist drops -- rows that never reach the output.  One
        int32 upload (ids | positions | passage of every row | row offsets).  Writes the n_out kept rows into `dst` and
        returns the device doclens; None if the encoder cannot (then the caller pads, and no further batch tries)."""
        import torch
        from ._lib import ArgumentError
        toks = [self._tokens[int(p)] for p in batch]
        lens = np.array([t.size for t in toks], dtype=np.int32)
        rows = int(lens.sum())
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        buf = np.empty(3 * rows + cu.size, dtype=np.int32)
        buf[:rows] = np.concatenate(toks)
        buf[rows:2 * rows] = np.concatenate([np.arange(n, dtype=np.int32) for n in lens])
        buf[2 * rows:3 * rows] = np.repeat(np.arange(lens.size, dtype=np.int32), lens)
        buf[3 * rows:] = cu
        d = torch.from_numpy(buf).to(self.device)
        try:
            return self.encoder.doc_embeddings_packed_device(d[:rows], d[rows:2 * rows], d[2 * rows:3 * rows], d[3 * rows:], int(lens.max()),
                                                             self._d_skip, n_out=n_out, out=dst)[1]
        except ArgumentError:
            self.packed = False
            return None

    def _expected(self, pids):
        return self._doclens[np.asarray(pids, dtype=np.int64)] if self._doclens is not None else self._doclens_of(pids)

    def encode_pids(self, pids, out=None):
        """encode_passages of the passages `pids` (batches of index_bsize, in this order) with the result left on the
        device -> (n, dim) float32 CUDA tensor (`out`, if given: a contiguous buffer of exactly that shape).  Only enqueues,
        then checks the device's doclens once."""
        import torch
        pids = np.asarray(pids, dtype=np.int64)
        if pids.size == 0:
            return torch.empty((0, self.dim), dtype=torch.float32, device=self.device) if out is None else out
        # packed batches need no common length: several index_bsize batches go into one call (rows, not passages, fill the chip)
        bs = self.encoder.config.index_bsize * (self.pack_batches if self.packed else 1)
        # ONE output buffer: every batch's kept rows are written by the encoder's epilogue straight into its slice (what a
        # batch keeps follows from its tokens alone), no concatenation afterwards
        expected = self._expected(pids)
        if out is None:
            out = torch.empty((int(expected.sum()), self.dim), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and tuple(out.shape) == (int(expected.sum()), self.dim)
        lens, fill, first = [], 0, 0
        # whether the encoder packs at all is learnt from the first call's first batch (a fixed number of passages): if it does not,
        # every batch is a padded batch of index_bsize passages counted from the first one -- the batches of the host route
        if self.packed and not getattr(self, "_pack_checked", False):
            n0 = min(bs, pids.size)
            n_b0 = int(expected[:n0].sum())
            dl = self._encode_packed(pids[:n0], out[:n_b0], n_b0)
            self._pack_checked = True
            if dl is not None:
                lens.append(dl); fill, first = n_b0, n0
        # batch boundaries: fixed passage counts when padding; a ROW budget when packing (see pack_rows)
        if self.packed and self.pack_rows > 0:
            ntok = np.fromiter((self._tokens[int(p)].size for p in pids), dtype=np.int64, count=pids.size)
            cum = np.cumsum(ntok)
            starts, s0 = [first], first
            while s0 < pids.size:
                nxt = int(np.searchsorted(cum, (cum[s0 - 1] if s0 else 0) + self.pack_rows, side="right"))
                nxt = max(nxt, s0 + 1)
                if nxt >= pids.size:
                    break
                starts.append(nxt); s0 = nxt
            bounds = [(a, b) for a, b in zip(starts, starts[1:] + [pids.size]) if a < b]
        else:
            step0 = bs if self.packed else self.encoder.config.index_bsize
            bounds = [(a, min(pids.size, a + step0)) for a in range(first, pids.size, step0)]
        for start, stop in bounds:
            batch = pids[start:stop]
            n_b = int(expected[start:stop].sum())
            dst = out[fill:fill + n_b]
            dl = None
            if self.packed:
                dl = self._encode_packed(batch, dst, n_b)
            if dl is not None:
                lens.append(dl); fill += n_b
                continue
            # padded batches of index_bsize passages (the encoder cannot pack; a row-budget batch is cut down to that size)
            step = self.encoder.config.index_bsize
            for a in range(start, stop, step):
                sub = pids[a:min(stop, a + step)]
                n_s = int(expected[a:min(stop, a + step)].sum())
                ids, mask = self._tensorize(sub)
                d_ids = torch.from_numpy(ids).to(self.device)
                d_mask = torch.from_numpy(mask).to(self.device)
                _, dl = self.encoder.doc_embeddings_device(d_ids, d_mask, self._d_skip, n_out=n_s, out=out[fill:fill + n_s])
                lens.append(dl); fill += n_s
        self.encoder.check_last_ids()
        assert np.array_equal(torch.cat(lens).cpu().numpy(), expected)
        return out

    def encode(self, passages=None):
        """The whole collection -> ((n, dim) CUDA tensor, doclens)."""
        return self.encode_pids(np.arange(self.n_docs)), self.doclens.copy()

    # The sampled passages are encoded for training (collection_indexer.jl:56-79) and, in the reference, once more with their
    # chunk.  Here their embeddings stay in HBM (1 M passages: 175 k of them, 7.4 GB) and a chunk encodes only its OTHER
    # passages -- into the tail of the same buffer -- and is cut out of that buffer in passage order with one
    # clb_gather_rows_device.  Packed batches only: their embeddings already depend on the batch a passage travels in (the tile
    # plan's rounding); the padded route stays batch for batch what encode_passages does, so that its index directory is the host
    # route's byte for byte.  COLBERT_REUSE_SAMPLE=0 encodes every chunk in full.
    _cache = None

    def _chunk_rows_bound(self):
        return min(25000, self.n_docs) * self._maxlen                  # setup(): chunksize <= 25 000 passages

    def sample(self, pids):
        import torch
        pids = np.asarray(pids, dtype=np.int64)
        sorted_unique = pids.size > 0 and bool(np.all(np.diff(pids) > 0))
        if not (self.packed and sorted_unique and os.environ.get("COLBERT_REUSE_SAMPLE", "1") != "0"):
            return self.encode_pids(pids)
        expected = self._expected(pids)
        n_rows = int(expected.sum())
        try:
            buf = torch.empty((n_rows + self._chunk_rows_bound(), self.dim), dtype=torch.float32, device=self.device)
        except RuntimeError:                                            # no room for the tail: the plain route
            return self.encode_pids(pids)
        self.encode_pids(pids, out=buf[:n_rows])
        if self.packed:                                                 # (the first call may have found that the encoder cannot pack)
            self._cache = {"pids": pids, "off": np.concatenate([[0], np.cumsum(expected)]).astype(np.int64), "buf": buf, "n": n_rows}
        return buf[:n_rows]

    def chunk(self, start: int, end: int):
        from . import codec
        c = self._cache
        pids = np.arange(start, end, dtype=np.int64)
        if c is None:
            return self.encode_pids(pids)
        if end >= self.n_docs:
            self._cache = None                                          # the last chunk: the buffer goes with it
        i0, i1 = (int(v) for v in np.searchsorted(c["pids"], [start, end]))
        if i0 == i1:
            return self.encode_pids(pids)
        known = np.zeros(end - start, dtype=bool)
        known[c["pids"][i0:i1] - start] = True
        dl = np.asarray(self._expected(pids), dtype=np.int64)
        assert np.array_equal(dl[known], np.diff(c["off"][i0:i1 + 1]))
        n_new = int(dl[~known].sum())
        if n_new > c["buf"].shape[0] - c["n"]:
            return self.encode_pids(pids)                               # (a chunk larger than setup() makes them)
        if n_new:
            self.encode_pids(pids[~known], out=c["buf"][c["n"]:c["n"] + n_new])
        rows = _chunk_source_rows(known, dl, c["off"][i0:i1], c["n"])
        self.reused_passages = getattr(self, "reused_passages", 0) + int(known.sum())
        return codec.gather_rows_device(c["buf"], rows)


def index_device(source: DeviceEmbeddingSource, nbits: int = 2, kmeans_niters: int = 20, chunksize=None, seed: int = 0,
                 num_partitions=None, log=None, rng=None, nranks: int = 1, on_sample=None, on_codec=None, on_chunk=None,
                 keep_residuals: bool = True):
    """The array stages of index() (src/indexing.jl:63-147) with the sample, the codes, the residuals and the IVF kept in
    HBM: sample pids -> gather their embeddings -> shuffle, held-out split -> setup -> k-means (the device-resident shard
    handle) -> codec statistics -> per chunk: compress (resident codec) -> _build_ivf.  Returns (index, record): `index`
    holds CUDA tensors (centroids (K, dim), codes int32 [n] 1-based, residuals uint8 (n, dim/8*nbits), ivf int64 [n]) and
    host arrays (doclens, ivf_lengths, bucket_cutoffs, bucket_weights) -- what Searcher(index=...) takes; `record` the
    seconds per stage.  RNG-dependent draws use numpy's generator (module docstring).  torch is used for ALLOCATION only:
    every row gather / shuffle goes through clb_gather_rows_device, the same calls a host without torch makes
    (julia/ColBERT/src/indexing.jl).  `on_codec(centroids, cutoffs, weights, avg)` / `on_chunk(ci, start, end, codes,
    residuals, doclens)` let index() write the codec and each chunk's files as they are produced; `keep_residuals=False`
    then keeps only the codes (for the IVF) on the device, one chunk of residuals at a time."""
    import time

    import torch
    # a source that derives its doclens from work still in flight (EncoderSource(lazy=True): the tokenizer runs on a host thread
    # while the device encodes the sample and trains) names its size; its doclens are read after k-means
    lazy = bool(getattr(source, "lazy", False)) and hasattr(source, "prepare_sample")
    n_docs = int(source.n_docs) if lazy else int(np.asarray(source.doclens).size)
    dim = source.dim
    dev = source.device
    rng = np.random.default_rng(seed) if rng is None else rng
    rec = {"passages": int(n_docs)}
    sync = lambda: torch.cuda.synchronize(dev)

    def tick(name, t0):
        sync()
        rec[name] = round(time.time() - t0, 3)
        if log:
            log(f"index_device: {name} {rec[name]} s")

    # sample (collection_indexer.jl:17-24, 56-79): the embeddings of the sampled passages, chunk by chunk
    t0 = time.time()
    n_s = codec.num_sampled_pids(n_docs)
    sampled = np.unique(rng.integers(0, n_docs, size=n_s))
    if lazy:
        source.prepare_sample(sampled)
    sample = source.sample(sampled) if hasattr(source, "sample") else _sample_from_chunks(source, sampled, chunksize)
    n_sample = int(sample.shape[0])                          # = doclens[sampled].sum()
    avg_doclen_est = float(np.float32(n_sample / max(sampled.size, 1)))
    # held-out split (collection_indexer.jl:81-91): shuffle the columns, the last heldout_size go to the held-out set
    sample = codec.gather_rows_device(sample.contiguous(), rng.permutation(n_sample))
    h = codec.heldout_size(n_sample)
    heldout = sample[n_sample - h:]
    sample = sample[:n_sample - h]
    plan = codec.setup(n_docs, avg_doclen_est, sample.shape[0], chunksize, nranks)
    K = int(num_partitions or plan["num_partitions"])
    if on_sample is not None:                                # index(): the files written before training (indexing.jl:84-96)
        on_sample(sample, heldout, plan)
    init = codec.gather_rows_device(sample, rng.permutation(sample.shape[0])[:K])
    rec.update({"sample_points": int(sample.shape[0]), "heldout": int(h), "K": K, "chunksize": plan["chunksize"]})
    tick("sample_and_split_s", t0)

    # train (collection_indexer.jl:219-237)
    t0 = time.time()
    sample = sample.contiguous()
    centroids, iters, shard = codec.kmeans_device(sample, init, max_iters=kmeans_niters)
    shard.close()
    tick("kmeans_s", t0)
    rec["kmeans_iters"] = int(iters)
    rec["kmeans_s_per_iter"] = round(rec["kmeans_s"] / max(iters, 1), 4)
    t0 = time.time()
    cent_host = np.asfortranarray(centroids.cpu().numpy().T)
    cut, w, avg, _ = codec.compute_avg_residuals(nbits, cent_host, np.asfortranarray(heldout.cpu().numpy().T), device=dev.index)
    tick("codec_stats_s", t0)
    del sample, heldout, init
    if on_codec is not None:
        on_codec(centroids, cut, w, avg)

    # chunk loop (collection_indexer.jl:271-297)
    t0 = time.time()
    doclens = np.ascontiguousarray(source.doclens, dtype=np.int64)
    assert doclens.size == n_docs and int(doclens[sampled].sum()) == n_sample
    off = np.concatenate([[0], np.cumsum(doclens)])
    n_emb = int(off[-1])
    rec["embeddings"] = n_emb
    rows = dim // 8 * nbits
    codes = torch.empty(n_emb, dtype=torch.int32, device=dev)
    residuals = torch.empty((n_emb, rows), dtype=torch.uint8, device=dev) if keep_residuals else None
    cdc = codec.Codec(centroids, cut, dim, nbits)
    t_src = t_cb = 0.0
    for ci, start in enumerate(range(0, n_docs, plan["chunksize"]), start=1):
        end = min(n_docs, start + plan["chunksize"])
        t1 = time.time()
        x = source.chunk(start, end)
        sync()
        t_src += time.time() - t1
        a, b = int(off[start]), int(off[end])
        assert x.shape[0] == b - a
        res = residuals[a:b] if keep_residuals else torch.empty((b - a, rows), dtype=torch.uint8, device=dev)
        cdc.compress_device(x, codes[a:b], res)
        sync()
        del x
        if on_chunk is not None:
            t1 = time.time()
            on_chunk(ci, start, end, codes[a:b], res, doclens[start:end])
            t_cb += time.time() - t1
        del res
    cdc.close()
    tick("chunks_s", t0)
    rec["generate_chunks_s"] = round(t_src, 3)
    rec["compress_s"] = round(rec["chunks_s"] - t_src - t_cb, 3)
    rec["compress_Membeddings_per_s"] = round(n_emb / max(rec["compress_s"], 1e-9) / 1e6, 2)
    t0 = time.time()
    ivf, ivf_lengths = codec.build_ivf_device(codes, K)
    tick("build_ivf_s", t0)
    rec["total_build_s"] = round(rec["sample_and_split_s"] + rec["kmeans_s"] + rec["codec_stats_s"] + rec["chunks_s"] + rec["build_ivf_s"], 2)
    index = {"dim": dim, "nbits": nbits, "centroids": centroids, "bucket_cutoffs": cut, "bucket_weights": w,
             "avg_residual": avg, "doclens": doclens, "codes": codes, "residuals": residuals, "ivf": ivf,
             "ivf_lengths": ivf_lengths.cpu().numpy(), "pid_offset": 0}
    return index, rec


def index_to_host(index: dict) -> dict:
    """A device-resident index (index_device) as the numpy arrays of the reference's layout (what the oracle and
    storage.save_* take): centroids (dim, K), codes UInt32, residuals (rows, n) column-major, ivf Int64."""
    out = dict(index)
    t = lambda a: a.cpu().numpy() if hasattr(a, "data_ptr") else np.asarray(a)
    out["centroids"] = np.asfortranarray(t(index["centroids"]).T)
    out["codes"] = t(index["codes"]).view(np.uint32)
    out["residuals"] = np.asfortranarray(t(index["residuals"]).T)
    out["ivf"] = t(index["ivf"])
    out["ivf_lengths"] = t(index["ivf_lengths"])
    return out
