#!/usr/bin/env python3
"""Round 5 A/B of the passage-side encoder changes on one box: LayerNorm folded around the Linear layers (ln_fold) and
the LDS-shared K / V tiles of the attention, each on / off, on a 64 x 300 passage batch (per-stage HIP-event times) and on
256 packed passages of ~86 tokens.  One JSON line per configuration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
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
    plens = np.clip(np.rint(86 + 30 * rng.standard_normal(256)), 8, 299).astype(np.int32)
    prow = int(plens.sum())
    pbuf = np.concatenate([rng.integers(1000, cfg["vocab_size"], size=prow).astype(np.int32),
                           np.concatenate([np.arange(n, dtype=np.int32) for n in plens]),
                           np.repeat(np.arange(plens.size, dtype=np.int32), plens),
                           np.concatenate([[0], np.cumsum(plens)]).astype(np.int32)])
    d = torch.from_numpy(pbuf).to(dev)
    base = None
    for ln_fold, att in ((0, "fused"), (0, "fused_lds"), (-1, "fused"), (-1, "fused_lds")):
        enc = clb.BertEncoder(w, cfg, dim=128, ln_fold=ln_fold, attention=att)
        for _ in range(2):
            x, _ = enc.doc_embeddings_device(p_ids, p_mask, p_skip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            enc.doc_embeddings_device(p_ids, p_mask, p_skip)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        enc.profile_enable(True)
        for _ in range(3):
            enc.doc_embeddings_device(p_ids, p_mask, p_skip)
        torch.cuda.synchronize()
        stages = {k: round(v["ms"] / 3, 4) for k, v in enc.profile_read().items()}
        enc.profile_enable(False)
        for _ in range(2):
            enc.doc_embeddings_packed_device(d[:prow], d[prow:2 * prow], d[2 * prow:3 * prow], d[3 * prow:], int(plens.max()), p_skip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            enc.doc_embeddings_packed_device(d[:prow], d[prow:2 * prow], d[2 * prow:3 * prow], d[3 * prow:], int(plens.max()), p_skip)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 5
        xh = x.cpu().numpy()
        if base is None:
            base = xh
        rec = {"ln_fold": ln_fold, "attention": att, "passages_64x300_ms": round(dt * 1e3, 3), "stages_ms": stages,
               "packed_256_ms": round(dtp * 1e3, 3), "packed_passages_per_s": round(256 / dtp, 1), "packed_rows": prow,
               "max_abs_diff_vs_first_config": float(np.abs(xh - base).max()), "finite": bool(np.isfinite(xh).all())}
        print(json.dumps(rec), flush=True)
        enc.close()


if __name__ == "__main__":
    main()
