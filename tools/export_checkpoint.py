#!/usr/bin/env python3
"""Exports a HuggingFace ColBERT checkpoint directory (config.json, pytorch_model.bin or model.safetensors,
artifact.metadata) to the flat fp32 blob libcolbert_hip's encoder loads -- the role of `_load_model` in
src/local_loading.jl:64-104.

    python tools/export_checkpoint.py <hf_dir> <out_dir> [--dim 128]

Writes <out_dir>/encoder.f32 and <out_dir>/encoder.json.  Weight names: `bert.*` for the encoder, `linear.weight`
(dim, hidden) for the projection; a missing `linear.bias` becomes zeros (the reference builds the Dense with a bias)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_state(hf_dir: str) -> dict:
    st = os.path.join(hf_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        return dict(load_file(st))
    import torch
    sd = torch.load(os.path.join(hf_dir, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return {k: v.float().numpy() for k, v in sd.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("hf_dir"); ap.add_argument("out_dir"); ap.add_argument("--dim", type=int, default=None)
    a = ap.parse_args()
    from colbert_jl_amd.encoder import pack_weights
    cfg = json.load(open(os.path.join(a.hf_dir, "config.json")))
    dim = a.dim
    meta_file = os.path.join(a.hf_dir, "artifact.metadata")
    if dim is None and os.path.exists(meta_file):
        dim = json.load(open(meta_file)).get("dim")
    dim = dim or 128
    raw = load_state(a.hf_dir)
    state = {}
    for k, v in raw.items():
        state[k[len("bert."):] if k.startswith("bert.") else k] = v
    blob = pack_weights(state, cfg, dim)
    os.makedirs(a.out_dir, exist_ok=True)
    blob.tofile(os.path.join(a.out_dir, "encoder.f32"))
    keep = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
            "max_position_embeddings", "type_vocab_size", "layer_norm_eps")
    json.dump({"dim": dim, "bert": {k: cfg[k] for k in keep if k in cfg}, "n_floats": int(blob.size)},
              open(os.path.join(a.out_dir, "encoder.json"), "w"), indent=1)
    print("wrote", a.out_dir, blob.size, "floats")


if __name__ == "__main__":
    main()
