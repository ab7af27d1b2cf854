"""Tokenizer worker PROCESS of index()'s device route (indexer.EncoderSource): the passages that are not in the clustering sample
are tokenised here while the parent encodes the sample and trains -- in a process of its own, because the per-passage conversion
of the tokenizer's output to arrays holds the interpreter lock and, run on a thread, took it away from the thread that prepares
and launches the encoder's batches (the sample phase of a 1 M-passage build: 21 s against 12 s alone).  Started by file path:
imports nothing of the package (no library load, no torch), never touches the GPU.

protocol (little endian), per chunk on stdin:  uint32 n, uint32 nbytes, uint32 len[n], utf-8 blob
                          per chunk on stdout: uint32 n, uint32 ntok,   int32 toklen[n], int32 ids[ntok]
ids: [CLS] [D] w1 .. wn [SEP], 1-based, cut to doc_maxlen tokens -- a passage's column of tensorize_docs
(doc_tokenization.jl:143-156), attended rows only.   argv: vocab_file lowercase(0/1) doc_maxlen marker_id"""
import struct
import sys

import numpy as np


def main():
    vocab, lowercase, maxlen, marker = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4])
    from tokenizers import BertWordPieceTokenizer
    tok = BertWordPieceTokenizer(vocab, lowercase=lowercase)
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        hdr = inp.read(8)
        if len(hdr) < 8:
            return
        n, nbytes = struct.unpack("<II", hdr)
        lens = np.frombuffer(inp.read(4 * n), dtype=np.uint32)
        blob = inp.read(nbytes)
        texts, o = [], 0
        for ln in lens:
            texts.append(blob[o:o + int(ln)].decode("utf-8"))
            o += int(ln)
        parts = []
        for e in tok.encode_batch(texts, add_special_tokens=True):
            ids = np.asarray(e.ids[:maxlen - 1], dtype=np.int32) + 1
            parts.append(np.concatenate([ids[:1], [marker], ids[1:]]).astype(np.int32))
        toklen = np.array([p.size for p in parts], dtype=np.int32)
        flat = np.concatenate(parts) if parts else np.zeros(0, np.int32)
        out.write(struct.pack("<II", n, flat.size))
        out.write(toklen.tobytes())
        out.write(flat.tobytes())
        out.flush()


if __name__ == "__main__":
    main()
