"""index(indexer) end to end with the BERT encoder in front (bert-base geometry, random weights, a synthetic vocabulary and
synthetic "text": the embeddings are meaningless, the work is that of examples/indexing.jl): tokenise -> encode the sample
-> k-means -> encode + compress every chunk -> IVF -> write the index directory, then one text query through Searcher.
Prints one JSON line with the seconds per stage and passages per second.
    python tools/bench_index_with_encoder.py [--docs 20000]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=20000)
    ap.add_argument("--words", type=int, default=4000, help="size of the synthetic vocabulary")
    ap.add_argument("--kmeans-iters", type=int, default=4)
    args = ap.parse_args()
    import torch

    import colbert_jl_amd as clb
    from colbert_jl_amd import indexer as ix
    from colbert_jl_amd.encoder import BERT_BASE, random_weights
    from colbert_jl_amd.tokenization import WordPieceTokenizer
    rng = np.random.default_rng(1)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    words = sorted({"".join(rng.choice(letters, size=rng.integers(3, 9))) for _ in range(args.words * 2)})[: args.words]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(".,!?;:") + words
    tmp = tempfile.mkdtemp(prefix="clb_index_")
    vf = os.path.join(tmp, "vocab.txt")
    with open(vf, "w") as f:
        f.write("\n".join(vocab) + "\n")
    tok = WordPieceTokenizer(vf)
    lens = np.clip(np.rint(80 + 30 * rng.standard_normal(args.docs)), 8, 280).astype(int)
    collection = [" ".join(rng.choice(words, size=n)) + "." for n in lens]
    cfg = dict(BERT_BASE, vocab_size=len(vocab))
    config = clb.ColBERTConfig(index_path=os.path.join(tmp, "index"), doc_maxlen=300, query_maxlen=32, index_bsize=64, nbits=2,
                               kmeans_niters=args.kmeans_iters)
    enc = clb.BertEncoder(random_weights(cfg, 128, seed=2), cfg, dim=128, tokenizer=tok, config=config)
    rec = {"passages": args.docs, "mean_words": float(lens.mean())}
    t0 = time.time()
    src = ix.EncoderSource(enc, collection, 0)
    rec["tokenize_s"] = round(time.time() - t0, 3)
    rec["mean_doclen"] = float(src.doclens.mean())
    t0 = time.time()
    x = src.chunk(0, min(args.docs, 6400))                       # encoder alone on the first 100 batches
    torch.cuda.synchronize()
    dt = time.time() - t0
    rec["encode_only_passages_per_s"] = round(min(args.docs, 6400) / dt, 1)
    t0 = time.time()
    enc.encode_passages(collection[: min(args.docs, 6400)])     # the host entry point (clb_encode_docs per batch of 64: what the Julia shim calls)
    rec["encode_only_host_route_passages_per_s"] = round(min(args.docs, 6400) / (time.time() - t0), 1)
    del x, src
    indexer = clb.Indexer(config, encoder=enc, collection=collection, seed=3)
    t0 = time.time()
    assert clb.index(indexer) == config.index_path
    torch.cuda.synchronize()
    rec["index_s"] = round(time.time() - t0, 2)
    rec["index_passages_per_s"] = round(args.docs / rec["index_s"], 1)
    rec["stages"] = {k: v for k, v in getattr(indexer, "last_build_record", {}).items() if k.endswith("_s")}
    rec["index_bytes"] = sum(os.path.getsize(os.path.join(config.index_path, f)) for f in os.listdir(config.index_path))
    searcher = clb.Searcher(config.index_path, encoder=enc)
    pids, scores = clb.search(searcher, collection[17], 10)
    rec["query_is_its_own_top_hit"] = bool(pids[0] == 18)
    ts = searcher.text_search(10, graph=False)                     # a few text queries through the serving session
    lat = []
    for i in range(30):
        t1 = time.perf_counter()
        ts(collection[(37 * i) % args.docs])
        lat.append(time.perf_counter() - t1)
    ts.close()
    rec["text_query_p50_ms"] = round(float(np.median(lat[5:])) * 1e3, 3)
    rec["text_query_max_ms"] = round(float(np.max(lat[5:])) * 1e3, 3)
    searcher.close(); enc.close()
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
