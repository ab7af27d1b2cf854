"""Small synthetic indexes shared by the tests."""
import colbert_jl_amd  # noqa: F401  (import shim)
from colbert_jl_amd import synthetic


def tiny_index(seed=0, n_docs=300, K=64, **kw):
    idx = synthetic.make_index(seed, n_docs, K=K, **kw)
    Q = synthetic.make_queries(idx, seed + 1000, 1)[:, :, 0]
    return idx, Q
