#!/usr/bin/env python3
"""Round 5: the 256 x 256 GEMM tile as four waves of 128 x 128 (encoder_big.hip, accumulators in AGPRs; COLBERT_ENC_WIDE_WAVES=1)
against eight waves of 64 x 128 (the default) -- a 64 x 300 passage batch (per-stage HIP-event times) and a packed batch filled to the row
budget of index().  One JSON line; `--save / --compare FILE` checks that both forms give the same bits."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None)
    ap.add_argument("--compare", default=None)
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd.encoder import BERT_BASE, random_weights
    cfg = dict(BERT_BASE)
    w = random_weights(cfg, 128, seed=1)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    N, L = 64, 300
    p_ids = torch.from_numpy(rng.integers(1, cfg["vocab_size"] + 1, size=(N, L)).astype(np.int32)).to(dev)
    p_mask = torch.ones((N, L), dtype=torch.uint8, device=dev)
    p_skip = torch.tensor([1, 1013, 1014], dtype=torch.int64, device=dev)
    lens = []
    while sum(lens) < 170 * 256 - 300:
        lens.append(int(np.clip(np.rint(86 + 30 * rng.standard_normal()), 8, 299)))
    plens = np.array(lens, dtype=np.int32)
    prow = int(plens.sum())
    pbuf = np.concatenate([rng.integers(1000, cfg["vocab_size"], size=prow).astype(np.int32),
                           np.concatenate([np.arange(n, dtype=np.int32) for n in plens]),
                           np.repeat(np.arange(plens.size, dtype=np.int32), plens),
                           np.concatenate([[0], np.cumsum(plens)]).astype(np.int32)])
    d = torch.from_numpy(pbuf).to(dev)
    enc = clb.BertEncoder(w, cfg, dim=128)
    for _ in range(2):
        x, _ = enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    enc.profile_enable(True)
    for _ in range(3):
        enc.doc_embeddings_device(p_ids, p_mask, p_skip)
    torch.cuda.synchronize()
    stages = {k: round(v["ms"] / 3, 4) for k, v in enc.profile_read().items()}
    enc.profile_enable(False)
    pk = lambda: enc.doc_embeddings_packed_device(d[:prow], d[prow:2 * prow], d[2 * prow:3 * prow], d[3 * prow:], int(plens.max()), p_skip)
    for _ in range(2):
        y, _ = pk()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        pk()
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / 6
    rec = {"wide_waves": os.environ.get("COLBERT_ENC_WIDE_WAVES", "0") == "1", "passages_64x300_ms": round(dt * 1e3, 3),
           "packed": {"passages": int(plens.size), "rows": prow, "ms": round(dtp * 1e3, 3), "passages_per_s": round(plens.size / dtp, 1)},
           "stages_ms": stages}
    xh, yh = x.cpu().numpy(), y.cpu().numpy()
    if args.save:
        np.savez(args.save, x=xh, y=yh)
    if args.compare:
        ref = np.load(args.compare)
        rec["same_bits_as_reference"] = bool(np.array_equal(ref["x"].view(np.uint32), xh.view(np.uint32)) and
                                             np.array_equal(ref["y"].view(np.uint32), yh.view(np.uint32)))
        rec["max_abs_diff"] = float(max(np.abs(ref["x"] - xh).max(), np.abs(ref["y"] - yh).max()))
    print(json.dumps(rec))
    enc.close()


if __name__ == "__main__":
    main()
