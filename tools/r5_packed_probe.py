#!/usr/bin/env python3
"""Packed passage batches of real lengths (~86 tokens) through the encoder: ms per call and passages/s for a given number of
passages per call.  Environment knobs of the library (COLBERT_ENC_ATT_QB, COLBERT_ENC_LNFOLD ...) are read once per process:
run one process per setting.   python tools/r5_packed_probe.py [passages ...]"""
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
    enc = clb.BertEncoder(random_weights(cfg, 128, seed=1), cfg, dim=128)
    dev = torch.device("cuda", 0)
    skip = torch.tensor([1, 1013, 1014], dtype=torch.int64, device=dev)
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("COLBERT_ENC")}}
    for n in [int(a) for a in sys.argv[1:]] or [256]:
        rng = np.random.default_rng(6)
        plens = np.clip(np.rint(86 + 30 * rng.standard_normal(n)), 8, 299).astype(np.int32)
        rows = int(plens.sum())
        buf = np.concatenate([rng.integers(1000, cfg["vocab_size"], size=rows).astype(np.int32),
                              np.concatenate([np.arange(k, dtype=np.int32) for k in plens]),
                              np.repeat(np.arange(plens.size, dtype=np.int32), plens),
                              np.concatenate([[0], np.cumsum(plens)]).astype(np.int32)])
        d = torch.from_numpy(buf).to(dev)
        call = lambda: enc.doc_embeddings_packed_device(d[:rows], d[rows:2 * rows], d[2 * rows:3 * rows], d[3 * rows:], int(plens.max()), skip)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        enc.profile_enable(True)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        st = {k: round(v["ms"] / 3, 3) for k, v in enc.profile_read().items()}
        enc.profile_enable(False)
        out[str(n)] = {"rows": rows, "ms": round(dt * 1e3, 3), "passages_per_s": round(n / dt, 1), "us_per_row": round(dt * 1e6 / rows, 4), "stages_ms": st}
    print(json.dumps(out))
    enc.close()


if __name__ == "__main__":
    main()
