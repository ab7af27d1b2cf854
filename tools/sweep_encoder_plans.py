"""Per-stage times of the device-resident query encode (32 x 32 tokens by default) under different tile plans of the
plane GEMMs (COLBERT_ENC_PLAN, read once per process: every plan runs in its own child process).
    python tools/sweep_encoder_plans.py [--n 32] [--l 32] [plan ...]     plan = "qkv=64x128x3x1,ffn_in=..." or "-" (defaults)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json, time, numpy as np, torch
sys.path.insert(0, %r)
import colbert_jl_amd as clb
from colbert_jl_amd.encoder import BERT_BASE, random_weights
N, L = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(BERT_BASE)
enc = clb.BertEncoder(random_weights(cfg, 128, seed=1), cfg, dim=128)
rng = np.random.default_rng(2)
dev = torch.device("cuda", 0)
d_ids = torch.from_numpy(rng.integers(1, cfg["vocab_size"] + 1, size=(N, L)).astype(np.int32)).to(dev)
d_mask = torch.ones((N, L), dtype=torch.uint8, device=dev)
d_skip = torch.tensor([1], dtype=torch.int64, device=dev)
d_out = torch.empty((N, L, 128), dtype=torch.float32, device=dev)
for _ in range(10):
    enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
enc.profile_enable(True)
for _ in range(10):
    enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
torch.cuda.synchronize()
prof = enc.profile_read()
print(json.dumps({"ms": round(ms, 4), "stages": {k: round(v["ms"] / 10, 4) for k, v in prof.items()},
                  "checksum": float(d_out.double().sum().item())}))
''' % ROOT


def main():
    args = sys.argv[1:]
    n, l = 32, 32
    while args and args[0] in ("--n", "--l"):
        if args[0] == "--n":
            n = int(args[1])
        else:
            l = int(args[1])
        args = args[2:]
    plans = args or ["-"]
    for plan in plans:
        env = dict(os.environ)
        if plan == "old":
            env["COLBERT_ENCODER_PLANES"] = "0"
        elif plan != "-":
            env["COLBERT_ENC_PLAN"] = plan
        r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(l)], env=env, capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:]
        print(plan, line, flush=True)


if __name__ == "__main__":
    main()
