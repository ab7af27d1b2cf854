#!/usr/bin/env python3
"""Transcribes the known-answer vectors (inputs + expected outputs) that ColBERT.jl's own test-suite
holds for the hot path into tests/golden/reference_kats.json.

Run in the build container, where the reference checkout is mounted at /root/reference:
    python tests/golden/make_reference_kats.py
Only DATA is extracted (numeric literals of the reference's @test statements); no reference source
text is stored.  The long `_unpackbits` vector is parsed out of the test file mechanically so that no
transcription error can creep in; the short vectors are listed below with the test file:line they
come from.  Matrices are stored as nested row lists (Julia `[a b; c d]` -> [[a,b],[c,d]]); 3-d Bool
arrays are stored flat in Julia's column-major order together with their shape.
"""
import json
import os
import re

REF = os.environ.get("COLBERT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def parse_unpackbits_vector():
    """test/indexing/codecs/residual.jl:274-816: 64 packed bytes and the expected 512 bits."""
    path = os.path.join(REF, "test/indexing/codecs/residual.jl")
    text = open(path).read()
    start = text.index('@testset "_unpackbits" begin')
    end = text.index("# Test 2: All zeros", start)
    block = text[start:end]
    packed_src = block[block.index("UInt8["):block.index("],", block.index("UInt8["))]
    packed = [int(tok, 2) for tok in re.findall(r"0b([01]{8})", packed_src)]
    bools_src = block[block.index("Bool["):]
    bools_src = bools_src[:bools_src.index("]")]
    bits = [int(tok) for tok in re.findall(r"\b([01])\b", bools_src[len("Bool["):])]
    assert len(packed) == 64 and len(bits) == 512, (len(packed), len(bits))
    return {"source": "test/indexing/codecs/residual.jl:274-816", "nbits": 1,
            "packed_shape": [1, 64], "packed_flat": packed,
            "expected_shape": [1, 8, 64], "expected_flat": bits}


KATS = {
    "_binarize": {
        "source": "test/indexing/codecs/residual.jl:59-101",
        "cases": [
            {"data": [[0, 1], [2, 3]], "nbits": 3, "shape": [3, 2, 2],
             "expected_flat": [0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 1, 0]},
            {"data": [[0, 1], [2, 3]], "nbits": 2, "shape": [2, 2, 2],
             "expected_flat": [0, 0, 0, 1, 1, 0, 1, 1]},
            {"data": [[7]], "nbits": 3, "shape": [3, 1, 1], "expected_flat": [1, 1, 1]},
            {"data": [[0, 1], [0, 1]], "nbits": 1, "shape": [1, 2, 2], "expected_flat": [0, 0, 1, 1]},
        ],
        "domain_error": {"data": [[0, 1], [4, 2]], "nbits": 2},
    },
    "_unbinarize": {
        "source": "test/indexing/codecs/residual.jl:128-140",
        "cases": [
            {"shape": [5, 1, 1], "bits_flat": [1, 0, 0, 1, 1], "expected": [[25]]},
            {"shape": [6, 2, 2],
             "bits_flat": [1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 1, 0],
             "expected": [[55, 20], [49, 24]]},
        ],
    },
    "_bucket_indices": {
        "source": "test/indexing/codecs/residual.jl:163-190",
        "cases": [
            {"data": [[1, 6], [3, 12]], "cutoffs": [0, 5, 10, 15], "expected": [[1, 2], [1, 3]]},
            {"data": [[5, 15]], "cutoffs": [], "expected": [[0, 0]]},
            {"data": [[1.1, 2.5, 7.8]], "cutoffs": [0.0, 2.0, 5.0, 10.0], "expected": [[1, 2, 3]]},
        ],
    },
    "_packbits": {
        "source": "test/indexing/codecs/residual.jl:219-229,248-253",
        "cases": [
            {"shape": [1, 64, 1],
             "bits_flat": [1, 1, 0, 1, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1,
                           0, 0, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0,
                           1, 1, 0, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 1, 0],
             "expected_bytes": [0b11011011, 0b01101010, 0b00111011, 0b11011010, 0b11100000,
                                0b01011001, 0b11111011, 0b01010001]},
        ],
        "alternating_byte": 0x55,
        "domain_error_shape": [3, 7, 5],
    },
    "_cids_to_eids!": {
        "source": "test/search/ranking.jl:5-11,54-68",
        "cases": [{"centroid_ids": [2, 1], "ivf": [1, 2, 3, 4, 5, 6], "ivf_lengths": [3, 2, 1],
                   "n_eids": 5, "expected": [4, 5, 1, 2, 3]}],
        "dimension_mismatch": [
            {"n_eids": 5, "centroid_ids": [1, 2, 3], "ivf": [1, 2, 3, 4, 5, 6], "ivf_lengths": [2, 2, 2]},
            {"n_eids": 6, "centroid_ids": [1, 2, 3], "ivf": [1, 2, 3, 4, 5], "ivf_lengths": [2, 2, 2]},
        ],
    },
    "retrieve": {
        "source": "test/search/ranking.jl:74-82",
        "ivf": [3, 1, 4, 5, 6, 2], "ivf_lengths": [2, 3, 1],
        "centroids": [[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]],
        "emb2pid": [10, 20, 30, 40, 50, 60], "nprobe": 2, "Q": [[0.5], [0.5]],
        "expected_pids": [10, 20, 30],
    },
    "_collect_compressed_embs_for_pids": {
        "source": "test/search/ranking.jl:87-121",
        "cases": [
            {"doclens": [3, 2, 4], "codes": [1, 2, 3, 4, 5, 6, 7, 8, 9],
             "residuals": [[0x11, 0x12, 0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19],
                           [0x21, 0x22, 0x23, 0x24, 0x25, 0x26, 0x27, 0x28, 0x29]],
             "pids": [1, 3], "expected_codes": [1, 2, 3, 6, 7, 8, 9],
             "expected_residuals": [[0x11, 0x12, 0x13, 0x16, 0x17, 0x18, 0x19],
                                    [0x21, 0x22, 0x23, 0x26, 0x27, 0x28, 0x29]]},
            {"doclens": [3, 2, 4], "codes": [1, 2, 3, 4, 5, 6, 7, 8, 9],
             "residuals": [[0x11, 0x12, 0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19],
                           [0x21, 0x22, 0x23, 0x24, 0x25, 0x26, 0x27, 0x28, 0x29]],
             "pids": [], "expected_codes": [], "expected_residuals": [[], []]},
            {"doclens": [3, 0, 4], "codes": [1, 2, 3, 6, 7, 8, 9],
             "residuals": [[0x11, 0x12, 0x13, 0x16, 0x17, 0x18, 0x19],
                           [0x21, 0x22, 0x23, 0x26, 0x27, 0x28, 0x29]],
             "pids": [1, 3], "expected_codes": [1, 2, 3, 6, 7, 8, 9],
             "expected_residuals": [[0x11, 0x12, 0x13, 0x16, 0x17, 0x18, 0x19],
                                    [0x21, 0x22, 0x23, 0x26, 0x27, 0x28, 0x29]]},
        ],
    },
    "maxsim": {
        "source": "test/search/ranking.jl:139-152",
        "Q": [[1.0, 0.5], [0.5, 1.0]], "D": [[0.8, 0.3, 0.1], [0.2, 0.7, 0.4]],
        "pids": [1, 2], "doclens": [1, 2], "expected_scores": [1.5, 1.5],
        "dimension_mismatch": {"Q": [[1.0, 0.5], [0.5, 1.0]], "D": [[0.8, 0.3]], "pids": [1, 2],
                               "doclens": [1, 2]},
    },
    "_build_emb2pid": {
        "source": "test/searching.jl:10-17",
        "cases": [{"doclens": [3, 2, 4], "expected": [1, 1, 1, 2, 2, 3, 3, 3, 3]},
                  {"doclens": [0, 2, 0, 3], "expected": [2, 2, 4, 4, 4]},
                  {"doclens": [], "expected": []}],
    },
    "_topk": {
        "source": "test/utils.jl:163-177",
        "data": [[3.0, 1.0, 4.0], [1.0, 5.0, 9.0], [2.0, 6.0, 5.0]], "k": 2,
        "expected_dims1": [[1, 3, 2], [3, 2, 3]], "expected_dims2": [[3, 1], [3, 2], [2, 3]],
    },
    "compute_distances_kernel!": {
        "source": "test/utils.jl:15-37",
        "single": {"batch_data": [[1.0], [2.0]], "centroids": [[2.0], [3.0]], "expected": [[2.0]]},
        "scaled_ones_rule": "data[:,i] = i, centroids[:,j] = j  =>  dist[j,i] == dim*(i-j)^2 exactly",
    },
    "onehot_encode!": {
        "source": "test/utils.jl:117-129",
        "assignments": [4, 2, 3, 1], "k": 4,
        "expected": [[0, 0, 0, 1], [0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0]],
    },
    "_bucket_cutoffs_and_weights": {
        "source": "test/indexing/collection_indexer.jl:86-93",
        "heldout_avg_residual": [[0.0, 0.2], [0.4, 0.6], [0.8, 1.0]], "nbits": 2,
        "expected_cutoffs": [0.25, 0.5, 0.75], "expected_weights": [0.125, 0.375, 0.625, 0.875],
        "comparison": "isapprox",
    },
    "_collect_embedding_id_offset": {
        "source": "test/indexing/collection_indexer.jl:262-271",
        "cases": [{"counts": [3, 5, 2], "total": 10, "offsets": [1, 4, 9]},
                  {"counts": [], "total": 0, "offsets": [0]}],
    },
    "_build_ivf": {
        "source": "test/indexing/collection_indexer.jl:288-292",
        "codes": [5, 3, 8, 2, 5, 5, 4, 2, 2, 1, 3], "num_partitions": 10,
        "expected_ivf": [10, 4, 8, 9, 2, 11, 7, 1, 5, 6, 3],
        "expected_ivf_lengths": [1, 3, 2, 1, 3, 0, 0, 1, 0, 0],
    },
    "setup_sizing": {
        "source": "README.md:81-86 ; examples/AIHelpMe/indexing_output:4-9 (logged runs, not tests)",
        "cases": [{"num_documents": 10, "avg_doclen_est": 178.28572, "num_partitions": 512,
                   "num_embeddings_est": 1782.8572},
                  {"num_documents": 141431, "avg_doclen_est": 62.15259, "num_partitions": 32768}],
    },
    "readme_codec_values": {
        "source": "README.md:100 (logged run)",
        "bucket_cutoffs": [-0.021662371, -0.00015685707, 0.020033525],
        "bucket_weights": [-0.041035336, -0.009812315, 0.008938393, 0.039779153],
    },
}


def main():
    kats = dict(KATS)
    kats["_unpackbits"] = parse_unpackbits_vector()
    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
