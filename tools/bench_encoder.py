#!/usr/bin/env python3
"""Times the Checkpoint encoder (E1-E4 of SURVEY.md 8a) at bert-base geometry with random weights: `doc()` on a batch
of passages (index_bsize x doc_maxlen) and on a batch of queries, host buffers included, plus the device-resident
query path.  Prints one JSON line.  flops = 2 * params_in_GEMMs * tokens + attention."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def flops(cfg, L, N):
    H, I, layers = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    per_tok = layers * 2 * (4 * H * H + 2 * H * I) + 2 * H * 128
    attn = layers * N * 2 * 2 * L * L * H
    return per_tok * L * N + attn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--gemm", default=None, help="bf16x6 (default), bf16x3 or f32")
    ap.add_argument("--attention", default="fused", help="fused (default), resident or unfused")
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd.encoder import BERT_BASE, random_weights
    cfg = dict(BERT_BASE)
    enc = clb.BertEncoder(random_weights(cfg, 128, seed=1), cfg, dim=128, gemm=args.gemm, attention=args.attention)
    out_gemm = enc.gemm
    rng = np.random.default_rng(2)
    out = {"gemm": out_gemm}
    for name, L, N in (("passages_64x300", 300, 64), ("queries_32x32", 32, 32), ("queries_64x32", 32, 64)):
        ids = rng.integers(1, cfg["vocab_size"] + 1, size=(L, N)).astype(np.int32)
        mask = np.ones((L, N), bool)
        enc.doc(ids, mask)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            enc.doc(ids, mask)
        dt = (time.perf_counter() - t0) / args.reps
        out[name] = {"ms": round(dt * 1e3, 3), "tflops": round(flops(cfg, L, N) / dt / 1e12, 1),
                     "sequences_per_s": round(N / dt, 1)}
    # device-resident query path (what bench.py's end-to-end line uses)
    dev = torch.device("cuda", 0)
    N, L = 32, 32
    d_ids = torch.from_numpy(rng.integers(1, cfg["vocab_size"] + 1, size=(N, L)).astype(np.int32)).to(dev)
    d_mask = torch.ones((N, L), dtype=torch.uint8, device=dev)
    d_skip = torch.tensor([1], dtype=torch.int64, device=dev)
    d_out = torch.empty((N, L, 128), dtype=torch.float32, device=dev)
    for _ in range(3):
        enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["queries_32x32_device"] = {"ms": round(dt * 1e3, 3), "tflops": round(flops(cfg, L, N) / dt / 1e12, 1),
                                   "queries_per_s": round(N / dt, 1)}
    # passages with the output left on the device (what index() uses: clb_encode_docs_device), per stage
    N, L = 64, 300
    p_ids = torch.from_numpy(rng.integers(1, cfg["vocab_size"] + 1, size=(N, L)).astype(np.int32)).to(dev)
    p_mask = torch.ones((N, L), dtype=torch.uint8, device=dev)
    p_skip = torch.tensor([1, 1013, 1014], dtype=torch.int64, device=dev)
    for _ in range(2):
        enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    out["passages_64x300_device"] = {"ms": round(dt * 1e3, 3), "tflops_fp32_equivalent": round(flops(cfg, L, N) / dt / 1e12, 1),
                                     "passages_per_s": round(N / dt, 1)}
    enc.profile_enable(True)
    for _ in range(args.reps):
        enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    out["passages_64x300_device"]["stages_ms"] = {k: round(v["ms"] / args.reps, 4) for k, v in enc.profile_read().items()}
    enc.profile_enable(False)
    # the host-buffer and the device-resident path agree bit for bit
    host = enc.query_embeddings([1], d_ids.cpu().numpy().T.copy(), np.ones((L, N), bool))     # (dim, L, N)
    same = np.array_equal(np.ascontiguousarray(host.transpose(2, 1, 0)).view(np.uint32), d_out.cpu().numpy().view(np.uint32))
    out["device_path_equals_host_path"] = bool(same)
    print(json.dumps(out))
    enc.close()


if __name__ == "__main__":
    main()
