"""Host-side tokenisation mirrors of src/modelling/tokenization/*.jl.  Ids are 1-based, as in the reference
(Julia): [PAD]=1, [unused0]=2, [unused1]=3, [UNK]=101, [CLS]=102, [SEP]=103, [MASK]=104 for bert-base-uncased.

`WordPieceTokenizer` wraps the HuggingFace `tokenizers` WordPiece model over a vocab.txt and reproduces the
reference's pipelines: documents go through TextEncodeBase.trunc_and_pad(doc_maxlen-1) (src/indexing.jl:37-46):
cut to doc_maxlen-1 tokens, then padded to the LONGEST sequence of the batch; queries through
trunc_or_pad(query_maxlen-1) (src/searching.jl:32-40): cut or padded to exactly query_maxlen-1.  The marker
token is then inserted as the second row (_add_marker_row, tokenizer_utils.jl:140-143).  Both are pinned to the
REPL outputs recorded in the reference's docstrings (tests/golden/tokenizer_kats.json)."""
from __future__ import annotations

import os
from typing import List

import numpy as np


def cpu_budget() -> int:
    """CPUs this process may really use: the cgroup quota (cpu.max) where there is one, else the affinity mask -- a GPU box shows
    all of the host's cores (128) to a container that is entitled to a fraction of them."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# The HuggingFace tokenizer's thread pool (rayon) sizes itself by the VISIBLE cores when it is first used.  index() runs the
# tokenizer on a host thread while the main thread prepares and launches the encoder's batches (indexer.EncoderSource): with
# 128 pool threads on a 16-CPU share the batch preparation starved and the sample phase of a 1 M-passage build took twice as
# long (24 against 12 s).  The pool gets the CPU budget minus two cores for the threads that feed the device, at most 16.
os.environ.setdefault("RAYON_NUM_THREADS", str(max(1, min(16, cpu_budget() - 2))))

PUNCTUATION = list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")      # src/indexing.jl:30-31


class WordPieceTokenizer:
    def __init__(self, vocab_file: str, lowercase: bool = True):
        from tokenizers import BertWordPieceTokenizer
        self.vocab_file, self.lowercase = vocab_file, lowercase       # (a worker process rebuilds the tokenizer from these)
        self.tok = BertWordPieceTokenizer(vocab_file, lowercase=lowercase)
        self.vocab = self.tok.get_vocab()

    # 1-based ids ------------------------------------------------------------------------------------
    def lookup(self, token: str) -> int:
        """lookup(tokenizer.vocab, token): unknown tokens map to [UNK] -- the reference passes "[Q]"
        (config.query_token) for queries, which is not in the vocabulary."""
        return self.vocab.get(token, self.vocab["[UNK]"]) + 1

    @property
    def pad_id(self) -> int:
        return self.lookup("[PAD]")

    def doc_skiplist(self, mask_punctuation: bool = True) -> List[int]:
        syms = PUNCTUATION + ["[PAD]"] if mask_punctuation else ["[PAD]"]
        return [self.lookup(s) for s in syms]

    def encode(self, batch_text: List[str], max_tokens: int, pad_to_max: bool):
        """TextEncoders.encode after the truncation/padding pipe: [CLS] w1 .. wn [SEP] truncated to
        `max_tokens` (:tail), padded with [PAD] to max_tokens (queries: trunc_or_pad) or to the longest
        sequence of the batch (documents: trunc_and_pad).  Returns (integer_ids Int32 (len, batch) 1-based,
        bitmask Bool (len, batch))."""
        # encode_batch: the same ids as one encode() per text, tokenised on the library's thread pool
        seqs = [e.ids[:max_tokens] for e in self.tok.encode_batch(list(batch_text), add_special_tokens=True)]
        width = max_tokens if pad_to_max else max(len(e) for e in seqs)
        ids = np.full((width, len(seqs)), self.pad_id, dtype=np.int32, order="F")
        mask = np.zeros((width, len(seqs)), dtype=bool, order="F")
        for j, e in enumerate(seqs):
            ids[: len(e), j] = np.asarray(e, dtype=np.int32) + 1
            mask[: len(e), j] = True
        return ids, mask


def _add_marker_row(data: np.ndarray, marker):
    """_add_marker_row (tokenizer_utils.jl:140-143): the marker becomes row 2."""
    head = data[: min(1, data.shape[0]), :]
    row = np.full((1, data.shape[1]), marker, dtype=data.dtype)
    return np.asfortranarray(np.concatenate([head, row, data[1:, :]], axis=0))


def tensorize_docs(doc_token: str, tokenizer, batch_text: List[str], doc_maxlen: int = 300):
    """tensorize_docs (doc_tokenization.jl:143-156)."""
    ids, mask = tokenizer.encode(batch_text, doc_maxlen - 1, pad_to_max=False)
    ids = _add_marker_row(ids, np.int32(tokenizer.lookup(doc_token)))
    mask = _add_marker_row(mask, True)
    return ids, mask


def tensorize_queries(query_token: str, attend_to_mask_tokens: bool, tokenizer, batch_text: List[str],
                      query_maxlen: int = 32):
    """tensorize_queries (query_tokenization.jl:174-197): [PAD] -> [MASK] augmentation."""
    ids, mask = tokenizer.encode(batch_text, query_maxlen - 1, pad_to_max=True)
    ids = _add_marker_row(ids, np.int32(tokenizer.lookup(query_token)))
    mask = _add_marker_row(mask, True)
    mask_id = tokenizer.lookup("[MASK]")
    ids[ids == tokenizer.pad_id] = mask_id
    if attend_to_mask_tokens:
        mask[ids == mask_id] = True
    return ids, mask
