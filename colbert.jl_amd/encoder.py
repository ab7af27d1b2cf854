"""Checkpoint encoder -- the reference's src/modelling/checkpoint.jl over the C ABI.

`BertEncoder` holds the BERT + Dense weights of a ColBERT checkpoint on the device and mirrors
`encode_passages` (checkpoint.jl:159-189) and `encode_queries` (:271-301): tokenise on the host
(tokenization.py), run the forward + epilogue on the device in batches of `index_bsize`, concatenate."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional

import numpy as np

from . import tokenization
from ._lib import check, colmajor, fptr, i64, lib
from .config import ColBERTConfig

# order of the flat weight blob (include/colbert_hip.h); names follow the HuggingFace BERT state dict
EMB_KEYS = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
            "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]


def layer_keys(l: int):
    p = f"encoder.layer.{l}."
    return ([p + f"attention.self.{n}.weight" for n in ("query", "key", "value")] +
            [p + f"attention.self.{n}.bias" for n in ("query", "key", "value")] +
            [p + "attention.output.dense.weight", p + "attention.output.dense.bias",
             p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
             p + "intermediate.dense.weight", p + "intermediate.dense.bias",
             p + "output.dense.weight", p + "output.dense.bias",
             p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"])


def pack_weights(state: dict, bert_cfg: dict, dim: int) -> np.ndarray:
    """state: name -> array with the `bert.` prefix stripped, plus `linear.weight` (dim, H) and optionally
    `linear.bias` (zeros when absent, as local_loading.jl:97-99 loads the Dense with bias)."""
    H = bert_cfg["hidden_size"]
    parts = [np.asarray(state[k], dtype=np.float32).ravel() for k in EMB_KEYS]
    for l in range(bert_cfg["num_hidden_layers"]):
        parts += [np.asarray(state[k], dtype=np.float32).ravel() for k in layer_keys(l)]
    parts.append(np.asarray(state["linear.weight"], dtype=np.float32).reshape(dim, H).ravel())
    parts.append(np.asarray(state.get("linear.bias", np.zeros(dim, np.float32)), dtype=np.float32).ravel())
    return np.ascontiguousarray(np.concatenate(parts))


BERT_BASE = {"vocab_size": 30522, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12,
             "intermediate_size": 3072, "max_position_embeddings": 512, "type_vocab_size": 2, "layer_norm_eps": 1e-12}


def random_weights(bert_cfg: dict, dim: int = 128, seed: int = 0) -> np.ndarray:
    """A random-init blob of the architecture `bert_cfg` (benchmarks: no checkpoint exists in the build image):
    N(0, 0.02) matrices, zero biases, unit LayerNorm gains -- the HuggingFace initialisation."""
    rng = np.random.default_rng(seed)
    H, I = bert_cfg["hidden_size"], bert_cfg["intermediate_size"]
    mat = lambda *shape: (0.02 * rng.standard_normal(shape, dtype=np.float32)).ravel()     # noqa: E731
    zeros = lambda n: np.zeros(n, np.float32)                                              # noqa: E731
    ones = lambda n: np.ones(n, np.float32)                                                # noqa: E731
    parts = [mat(bert_cfg["vocab_size"], H), mat(bert_cfg["max_position_embeddings"], H),
             mat(bert_cfg.get("type_vocab_size", 2), H), ones(H), zeros(H)]
    for _ in range(bert_cfg["num_hidden_layers"]):
        parts += [mat(3 * H, H), zeros(3 * H), mat(H, H), zeros(H), ones(H), zeros(H), mat(I, H), zeros(I), mat(H, I),
                  zeros(H), ones(H), zeros(H)]
    parts += [mat(dim, H), zeros(dim)]
    return np.concatenate(parts)


class BertEncoder:
    def __init__(self, weights: np.ndarray, bert_cfg: dict, dim: int = 128, device: int = 0,
                 tokenizer=None, config: Optional[ColBERTConfig] = None, gemm: Optional[str] = None,
                 attention: str = "fused", ln_fold: Optional[int] = None):
        """`gemm`: arithmetic of the Linear layers -- "f16x3" (default: every fp32 operand, scaled by a power of two, split
        into TWO fp16 planes -- round-to-nearest makes them hold all 24 significant bits -- and three exact fp16 MFMA products
        per fp32 product, fp32 accumulation: fp32-faithful like bf16x6 at half its products), "bf16x6" (three bf16 planes, six
        products: fp32-faithful over the whole fp32 exponent range), "bf16x3" (two bf16 planes, three products, ~16 significant
        bits) or "f32" (fp32 MFMA); COLBERT_ENCODER_GEMM sets the default."""
        self.cfg = dict(bert_cfg); self.dim = dim; self.device = device
        self.tokenizer = tokenizer
        self.config = config or ColBERTConfig(dim=dim)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._h = C.c_void_p()
        check(lib().clb_encoder_create(device, i64(bert_cfg["vocab_size"]), i64(bert_cfg["hidden_size"]),
                                       i64(bert_cfg["num_hidden_layers"]), i64(bert_cfg["num_attention_heads"]),
                                       i64(bert_cfg["intermediate_size"]), i64(bert_cfg["max_position_embeddings"]),
                                       i64(bert_cfg.get("type_vocab_size", 2)), i64(dim),
                                       C.c_float(bert_cfg.get("layer_norm_eps", 1e-12)), fptr(w), i64(w.size),
                                       C.byref(self._h)))
        self.gemm = gemm or os.environ.get("COLBERT_ENCODER_GEMM", "f16x3")
        modes = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}
        if self.gemm not in modes:
            raise ValueError(f"gemm must be one of {sorted(modes)}, not {self.gemm!r}")
        check(lib().clb_encoder_set_gemm_mode(self._h, modes[self.gemm]))
        amodes = {"fused": 0, "resident": 1, "unfused": 2, "fused_f32": 3, "fused_lds": 5}     # 1-5: comparison paths (clb_encoder_set_attention_mode)
        if attention not in amodes:
            raise ValueError(f"attention must be one of {sorted(amodes)}, not {attention!r}")
        try:
            check(lib().clb_encoder_set_attention_mode(self._h, amodes[attention]))
        except Exception:
            self.close()                 # (a mode the library refuses: no handle is left behind)
            raise
        # LayerNorm folded around the Linear layers (clb_encoder_set_ln_fold): -1 long batches only (default), 0 never, 1 always
        if ln_fold is None and "COLBERT_ENC_LNFOLD" in os.environ:
            ln_fold = int(os.environ["COLBERT_ENC_LNFOLD"])
        if ln_fold is not None:
            try:
                check(lib().clb_encoder_set_ln_fold(self._h, C.c_int(int(ln_fold))))
            except Exception:
                self.close()
                raise

    @classmethod
    def from_export(cls, path: str, **kw) -> "BertEncoder":
        """Load what tools/export_checkpoint.py wrote: <path>/encoder.json + <path>/encoder.f32."""
        meta = json.load(open(os.path.join(path, "encoder.json")))
        w = np.fromfile(os.path.join(path, "encoder.f32"), dtype=np.float32)
        return cls(w, meta["bert"], dim=meta["dim"], **kw)

    def close(self):
        if getattr(self, "_h", None):
            lib().clb_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- device calls -----------------------------------------------------------------------------
    def doc(self, integer_ids, bitmask):
        """doc(bert, linear, integer_ids, bitmask) (checkpoint.jl:21-25) -> (dim, L, N) Float32."""
        ids = colmajor(integer_ids, np.int32); m = colmajor(np.asarray(bitmask).astype(np.uint8), np.uint8)
        L, N = ids.shape
        out = np.zeros((self.dim, L, N), dtype=np.float32, order="F")
        check(lib().clb_encode(self._h, fptr(ids), fptr(m), i64(L), i64(N), fptr(out)))
        return out

    def doc_embeddings_and_doclens(self, skiplist, integer_ids, bitmask):
        """_doc_embeddings_and_doclens (checkpoint.jl:27-52)."""
        ids = colmajor(integer_ids, np.int32); m = colmajor(np.asarray(bitmask).astype(np.uint8), np.uint8)
        sk = np.ascontiguousarray(skiplist, dtype=np.int64)
        L, N = ids.shape
        out = np.zeros((self.dim, L * N), dtype=np.float32, order="F")
        doclens = np.zeros(N, dtype=np.int64)
        n_out = i64(0)
        check(lib().clb_encode_docs(self._h, fptr(ids), fptr(m), i64(L), i64(N), fptr(sk), i64(sk.size), fptr(out),
                                    fptr(doclens), C.byref(n_out)))
        return np.asfortranarray(out[:, : n_out.value]), doclens

    def query_embeddings(self, skiplist, integer_ids, bitmask):
        """_query_embeddings (checkpoint.jl:54-71)."""
        ids = colmajor(integer_ids, np.int32); m = colmajor(np.asarray(bitmask).astype(np.uint8), np.uint8)
        sk = np.ascontiguousarray(skiplist, dtype=np.int64)
        L, N = ids.shape
        out = np.zeros((self.dim, L, N), dtype=np.float32, order="F")
        check(lib().clb_encode_queries(self._h, fptr(ids), fptr(m), i64(L), i64(N), fptr(sk), i64(sk.size), fptr(out)))
        return out

    def query_embeddings_device(self, d_ids, d_mask, d_skiplist, d_out):
        """_query_embeddings with torch CUDA tensors: ids int32 (N, L) row-major (= Julia (L, N)), mask uint8 (N, L),
        skiplist int64, out float32 (N, L, dim) -- enqueued on torch's current stream, not waited for.  `out` is the
        (B, T, dim) query tensor DeviceSearch takes."""
        import torch
        N, L = d_ids.shape
        st = torch.cuda.current_stream(d_ids.device).cuda_stream
        check(lib().clb_encode_queries_device(self._h, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_mask.data_ptr()), i64(L),
                                              i64(N), C.c_void_p(d_skiplist.data_ptr()), i64(d_skiplist.numel()),
                                              C.c_void_p(d_out.data_ptr()), C.c_void_p(st)))
        return d_out

    def doc_embeddings_device(self, d_ids, d_mask, d_skiplist, n_out: Optional[int] = None, out=None):
        """_doc_embeddings_and_doclens with torch CUDA tensors (clb_encode_docs_device): ids int32 (N, L) row-major, mask
        uint8 (N, L), skiplist int64 -> (embs float32 (n_kept, dim) CUDA tensor -- the reference's (dim, n_kept) matrix --,
        doclens int64 (N,) CUDA tensor).  `n_out`: the number of kept tokens when the caller already knows it (it follows
        from ids, mask and skiplist alone) -- then nothing is read back and the call only enqueues; else one 8-byte
        read-back per call."""
        import torch
        N, L = d_ids.shape
        if out is None:      # `out` (with n_out known): a contiguous (>= n_out, dim) slice of the caller's buffer -- only kept rows are written
            out = torch.empty((N * L, self.dim), dtype=torch.float32, device=d_ids.device)
        else:
            assert n_out is not None and out.is_contiguous() and out.shape[0] >= n_out and out.shape[1] == self.dim
        doclens = torch.empty(N, dtype=torch.int64, device=d_ids.device)
        n_dev = torch.zeros(1, dtype=torch.int64, device=d_ids.device)
        st = torch.cuda.current_stream(d_ids.device).cuda_stream
        check(lib().clb_encode_docs_device(self._h, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_mask.data_ptr()), i64(L), i64(N),
                                           C.c_void_p(d_skiplist.data_ptr()), i64(d_skiplist.numel()), C.c_void_p(out.data_ptr()),
                                           C.c_void_p(doclens.data_ptr()), C.c_void_p(n_dev.data_ptr()), C.c_void_p(st)))
        return out[: int(n_dev.item()) if n_out is None else int(n_out)], doclens

    def doc_embeddings_packed_device(self, d_ids, d_pos, d_seq, d_cu, Lmax: int, d_skiplist, n_out: Optional[int] = None, out=None):
        """_doc_embeddings_and_doclens for a PACKED batch (clb_encode_docs_packed_device): the N passages follow one another
        without padding rows -- ids / pos / seq int32 (rows,), cu int32 (N + 1,) row offsets, Lmax the longest passage.
        Same outputs as doc_embeddings_device.  Raises ArgumentError when the encoder cannot run packed batches (head size
        other than 64, a GEMM mode other than f16x3): pad then."""
        import torch
        rows, N = int(d_ids.numel()), int(d_cu.numel()) - 1
        if out is None:
            out = torch.empty((rows, self.dim), dtype=torch.float32, device=d_ids.device)
        else:
            assert n_out is not None and out.is_contiguous() and out.shape[0] >= n_out and out.shape[1] == self.dim
        doclens = torch.empty(N, dtype=torch.int64, device=d_ids.device)
        n_dev = torch.zeros(1, dtype=torch.int64, device=d_ids.device)
        st = torch.cuda.current_stream(d_ids.device).cuda_stream
        check(lib().clb_encode_docs_packed_device(self._h, C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_pos.data_ptr()),
                                                  C.c_void_p(d_seq.data_ptr()), C.c_void_p(d_cu.data_ptr()), i64(N), i64(Lmax), i64(rows),
                                                  C.c_void_p(d_skiplist.data_ptr()), i64(d_skiplist.numel()), C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(doclens.data_ptr()), C.c_void_p(n_dev.data_ptr()), C.c_void_p(st)))
        return out[: int(n_dev.item()) if n_out is None else int(n_out)], doclens

    def capture_query_graph(self, d_ids, d_mask, d_skiplist, d_out):
        """The ~90 launches of one `query_embeddings_device` call over STATIC buffers as a HIP graph (torch.cuda.CUDAGraph):
        write the next batch's ids into `d_ids` (and `d_mask`), `graph.replay()`, read `d_out`.  The library enqueues only
        kernels on the stream it is handed, so the forward can be captured; a first, uncaptured call sizes the workspaces."""
        import torch
        dev = d_ids.device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.query_embeddings_device(d_ids, d_mask, d_skiplist, d_out)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.query_embeddings_device(d_ids, d_mask, d_skiplist, d_out)
        return graph

    def check_last_ids(self):
        """Raise BoundsError if the last (asynchronous, device-resident) encode saw a token id outside the vocabulary --
        query_embeddings_device clamps such ids because it cannot report them when it is enqueued."""
        check(lib().clb_encoder_check_last_ids(self._h))

    def error_flag_ptr(self) -> int:
        """Device address of the encoder's sticky int32 error flag (clb_encoder_error_flag_device): a serving loop copies it
        back with its results and calls check_last_ids() only when it is non-zero."""
        p = C.c_void_p(0)
        check(lib().clb_encoder_error_flag_device(self._h, C.byref(p)))
        return int(p.value)

    # -- profiling (bench.py) -----------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        """HIP events around every stage of the forward, on the launching stream (not for timed regions)."""
        check(lib().clb_encoder_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile_read(self) -> dict:
        cap = 16
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); cnt = (C.c_int64 * cap)()
        n = lib().clb_encoder_profile_read(self._h, names, ms, cnt, cap)
        if n < 0:
            check(10)
        return {names[i].decode(): {"ms": ms[i], "launches": cnt[i]} for i in range(n)}

    # -- the reference's batching loops -----------------------------------------------------------------
    def encode_passages(self, passages: List[str], skiplist=None, doc_token: Optional[str] = None):
        """encode_passages (checkpoint.jl:159-189) -> (embs (dim, sum(doclens)), doclens)."""
        cfg = self.config
        if len(passages) == 0:
            return np.zeros((self.dim, 0), np.float32, order="F"), np.zeros(0, np.int64)
        skiplist = self.tokenizer.doc_skiplist(cfg.mask_punctuation) if skiplist is None else skiplist
        embs, doclens = [], []
        for off in range(0, len(passages), cfg.index_bsize):
            ids, mask = tokenization.tensorize_docs(doc_token or cfg.doc_token_id, self.tokenizer,
                                                    passages[off:off + cfg.index_bsize], cfg.doc_maxlen)
            D, dl = self.doc_embeddings_and_doclens(skiplist, ids, mask)
            embs.append(D); doclens.append(dl)
        return np.asfortranarray(np.concatenate(embs, axis=1)), np.concatenate(doclens)

    def encode_queries(self, queries: List[str], skiplist=None, query_token: Optional[str] = None):
        """encode_queries (checkpoint.jl:271-301) -> (dim, query_maxlen, n)."""
        cfg = self.config
        if len(queries) == 0:
            return np.zeros((self.dim, 0), np.float32, order="F")
        skiplist = [self.tokenizer.pad_id] if skiplist is None else skiplist      # searching.jl:62
        out = []
        for off in range(0, len(queries), cfg.index_bsize):
            ids, mask = tokenization.tensorize_queries(query_token or cfg.query_token, cfg.attend_to_mask_tokens,
                                                       self.tokenizer, queries[off:off + cfg.index_bsize],
                                                       cfg.query_maxlen)
            out.append(self.query_embeddings(skiplist, ids, mask))
        return np.asfortranarray(np.concatenate(out, axis=2))
