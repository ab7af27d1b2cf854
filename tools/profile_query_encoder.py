"""30 device-resident query encodes (32 x 32 tokens, bert-base geometry, random weights) for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/enc -- python3 tools/profile_query_encoder.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_jl_amd as clb  # noqa: E402
from colbert_jl_amd.encoder import BERT_BASE, random_weights  # noqa: E402

cfg = dict(BERT_BASE)
enc = clb.BertEncoder(random_weights(cfg, 128, seed=1), cfg, dim=128)
rng = np.random.default_rng(2)
dev = torch.device("cuda", 0)
N, L = int(os.environ.get("ENC_N", 32)), int(os.environ.get("ENC_L", 32))
d_ids = torch.from_numpy(rng.integers(1, cfg["vocab_size"] + 1, size=(N, L)).astype(np.int32)).to(dev)
d_mask = torch.ones((N, L), dtype=torch.uint8, device=dev)
d_skip = torch.tensor([1], dtype=torch.int64, device=dev)
d_out = torch.empty((N, L, 128), dtype=torch.float32, device=dev)
for _ in range(30):
    enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
torch.cuda.synchronize()
