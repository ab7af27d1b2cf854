"""Encoder: host tokenisation mirrors (CPU) and the HIP BERT forward against an independent fp32 reference of
the same architecture (HuggingFace `BertModel`, randomly initialised -- no checkpoint exists on disk; the
reference's own arithmetic lives in un-vendored Transformers.jl, so encoder parity is *unpinned*)."""
import json
import os

import numpy as np
import pytest

import colbert_jl_amd as clb
from colbert_jl_amd import tokenization

VOCAB = (["[PAD]", "[unused0]", "[unused1]"] + [f"[unused{i}]" for i in range(2, 20)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] +
         list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") +
         ["hello", "world", "this", "is", "a", "test", "of", "the", "token", "##izer", "##s", "longer", "passage",
          "with", "many", "words", "query", "what", "col", "##bert"])


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    d = tmp_path_factory.mktemp("vocab")
    f = d / "vocab.txt"
    f.write_text("\n".join(VOCAB) + "\n")
    return tokenization.WordPieceTokenizer(str(f))


def test_add_marker_row():
    """test/modelling/tokenization/tokenizer_utils.jl:3-19"""
    x = np.arange(12, dtype=np.int32).reshape(3, 4)
    y = tokenization._add_marker_row(x, np.int32(99))
    assert y.shape == (4, 4) and np.array_equal(y[0], x[0]) and np.all(y[1] == 99) and np.array_equal(y[2:], x[1:])
    m = tokenization._add_marker_row(np.ones((5, 2), bool), True)
    assert m.shape == (6, 2) and m.all()
    e = tokenization._add_marker_row(np.zeros((0, 3), np.int32), np.int32(7))
    assert e.shape == (1, 3) and np.all(e == 7)


def test_tensorize_docs(tok):
    ids, mask = tokenization.tensorize_docs("[unused1]", tok, ["hello world!", "this is a longer passage with many words"], 12)
    assert ids.shape == mask.shape == (11, 2) and ids.dtype == np.int32           # padded to the longest of the batch
    one = lambda t: tok.lookup(t)
    assert ids[0, 0] == one("[CLS]") and ids[1, 0] == one("[unused1]") == 3          # marker is row 2, 1-based id 3
    assert list(ids[2:6, 0]) == [one("hello"), one("world"), one("!"), one("[SEP]")]
    assert np.all(ids[6:, 0] == one("[PAD]")) and one("[PAD]") == 1
    assert mask[:6, 0].all() and not mask[6:, 0].any()
    assert mask[:, 1].sum() == 10 + 1                                              # [CLS] + 8 words + [SEP] + marker
    # truncation to doc_maxlen - 1 tokens before the marker
    ids, mask = tokenization.tensorize_docs("[unused1]", tok, ["this is a longer passage with many words"], 6)
    assert ids.shape == (6, 1) and mask.all()
    sk = tok.doc_skiplist(True)
    assert len(sk) == 33 and sk[-1] == 1 and len(tok.doc_skiplist(False)) == 1    # embedding_utils.jl:37-72


def _docstring_tokenizer(tmp_path, kat, vocab_size):
    """vocab.txt with the entries the recorded example touches at their bert-base-uncased line numbers."""
    lines = ["[unused%d]" % i for i in range(vocab_size)]
    for k, v in kat["vocab_1based"].items():
        lines[int(k) - 1] = v
    lines[100] = "[UNK]"
    f = tmp_path / "vocab.txt"
    f.write_text("\n".join(lines) + "\n")
    return tokenization.WordPieceTokenizer(str(f))


def test_tensorize_docs_reference_docstring(tmp_path):
    """doc_tokenization.jl:61-141: the REPL session recorded against colbertv2.0's vocabulary."""
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_kats.json")))
    kat = kats["docs"]
    t = _docstring_tokenizer(tmp_path, kat, kats["vocab_size"])
    ids, mask = tokenization.tensorize_docs(kat["marker"], t, kat["texts"], kat["maxlen"])
    assert np.array_equal(ids, np.asarray(kat["integer_ids"], np.int32))
    assert np.array_equal(mask, np.asarray(kat["bitmask"], bool))


def test_tensorize_queries_reference_docstring(tmp_path):
    """query_tokenization.jl:55-171 (wordpiece splits "ras ##p ##berries", truncation of the long query at 32)."""
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_kats.json")))
    kat = kats["queries"]
    # words of the examples whose ids the docstring does not show must not resolve by accident: every other line
    # of the synthetic vocabulary is an [unusedN] filler
    t = _docstring_tokenizer(tmp_path, kat, kats["vocab_size"])
    ids, mask = tokenization.tensorize_queries(kat["marker"], False, t, kat["texts"], kat["maxlen"])
    assert np.array_equal(ids, np.asarray(kat["integer_ids"], np.int32))
    assert np.array_equal(mask, np.asarray(kat["bitmask"], bool))


def test_tensorize_queries(tok):
    ids, mask = tokenization.tensorize_queries("[Q]", False, tok, ["what is col bert", "hello"], 8)
    assert ids.shape == (8, 2)
    assert ids[1, 0] == tok.lookup("[UNK]")                                        # "[Q]" is not in the vocab -> [UNK]
    assert not np.any(ids == tok.pad_id)                                           # [PAD] -> [MASK] augmentation
    assert np.all(ids[4:, 1] == tok.lookup("[MASK]"))
    assert mask[:, 1].sum() == 4 and mask[:, 0].sum() == 7                         # real tokens (+ marker) only
    ids2, mask2 = tokenization.tensorize_queries("[Q]", True, tok, ["hello"], 8)
    assert mask2.all()


def test_tokenizer_worker_process_matches_in_process(tok):
    """index()'s device route tokenises the passages outside the clustering sample in a worker PROCESS
    (colbert.jl_amd/_tok_worker.py: the conversion loop holds the interpreter lock and starved the thread feeding the device
    when it ran on a thread): the same token arrays as the in-process tokenizer, and the in-process path when the worker is
    switched off.  CPU only -- the worker never touches the GPU."""
    from colbert_jl_amd import indexer
    rng = np.random.default_rng(0)
    words = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with"]
    coll = [" ".join(rng.choice(words, size=rng.integers(1, 30))) + rng.choice([".", "!", ""]) for _ in range(9000)]
    coll[17] = "h\u00e9llo w\u00f6rld \u2014 caf\u00e9"                     # multi-byte text: lengths are byte lengths on the pipe

    class Enc:
        pass
    src = indexer.EncoderSource.__new__(indexer.EncoderSource)
    src.encoder = Enc(); src.encoder.tokenizer = tok
    src.collection, src.n_docs, src._maxlen = coll, len(coll), 16
    src._marker = np.int32(tok.lookup("[unused1]"))
    src._tokens = [None] * len(coll)
    src._tokenize(range(len(coll)))
    want = [t.copy() for t in src._tokens]
    assert max(t.size for t in want) == 16 and want[0][1] == src._marker
    src._tokens = [None] * len(coll)
    src._tokenize_in_worker_process(list(range(len(coll))))
    assert all(np.array_equal(a, b) for a, b in zip(want, src._tokens))
    os.environ["COLBERT_TOKENIZER_PROCESS"] = "0"
    try:
        src._tokens = [None] * len(coll)
        src._tokenize_in_worker_process(list(range(len(coll))))
        assert all(np.array_equal(a, b) for a, b in zip(want, src._tokens))
    finally:
        del os.environ["COLBERT_TOKENIZER_PROCESS"]


# ---------------------------------------------------------------------------------------------------------
def _random_bert(hidden=64, layers=2, heads=4, inter=128, vocab=120, max_pos=48, dim=32, seed=0):
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(seed)
    cfg = transformers.BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                                  num_attention_heads=heads, intermediate_size=inter,
                                  max_position_embeddings=max_pos, type_vocab_size=2, hidden_act="gelu",
                                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bert = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    linear = torch.nn.Linear(hidden, dim, bias=True)
    with torch.no_grad():                              # default init is tiny (std 0.02): make the test sensitive
        for p in bert.parameters():
            p.mul_(4.0)
    return torch, cfg, bert, linear


def _state(bert, linear):
    st = {k: v.detach().float().numpy() for k, v in bert.state_dict().items()}
    st["linear.weight"] = linear.weight.detach().numpy(); st["linear.bias"] = linear.bias.detach().numpy()
    return st


# tolerance against the independent fp32 implementation: "f32" = fp32 MFMA GEMMs and "bf16x6" = six exact bf16 plane
# products per fp32 product (differences are summation order / < 2^-22 per product); "bf16x3" = three plane products,
# < 2^-15 relative error per product on top of that
# "f16x3" (round 4) = two fp16 planes per (power-of-two scaled) operand, three products: held to bf16x6's tolerances
TUNING_BUILD = b"tuning build" in clb.lib().clb_version()      # make ABLATIONS=1: the comparison kernels exist
GEMM_TOL = {"f32": 2e-4, "bf16x6": 2e-4, "f16x3": 2e-4, "bf16x3": 6e-4}
GEMM_TOL_BASE = {"f32": 1e-3, "bf16x6": 1e-3, "f16x3": 1e-3, "bf16x3": 6e-3}


@pytest.mark.gpu
def test_f16x3_split_is_as_accurate_as_bf16x6():
    """The default split (two scaled fp16 planes, three products) against a float64 evaluation of the same network: its
    error must be that of the fp32-faithful modes (bf16x6, fp32 MFMA), not that of bf16x3 -- and an activation outside the
    fp16 range is reported, not returned as numbers."""
    torch, cfg, bert, linear = _random_bert()
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    w = pack_weights(_state(bert, linear), bcfg, 32)
    rng = np.random.default_rng(7)
    L, N = 40, 6
    ids0 = rng.integers(0, cfg.vocab_size, size=(N, L))
    mask = np.ones((N, L), bool)
    with torch.no_grad():
        ref = linear.double()(bert.double()(input_ids=torch.from_numpy(ids0), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state).numpy()
    err = {}
    for gemm in ("f32", "bf16x6", "f16x3", "bf16x3"):
        enc = clb.BertEncoder(w, bcfg, dim=32, gemm=gemm)
        got = enc.doc((ids0.T + 1).astype(np.int32), mask.T).transpose(2, 1, 0).astype(np.float64)
        enc.close()
        err[gemm] = float(np.abs(got - ref).max())
    print("[encoder vs float64]", {k: f"{v:.3g}" for k, v in err.items()})
    faithful = max(err["f32"], err["bf16x6"])
    assert err["f16x3"] <= 1.5 * faithful, err
    assert err["bf16x3"] > 3 * err["f16x3"], err                  # the test can tell the two classes apart
    # out of range: weights x 1e4 push the hidden activations past 4 094 -> Inf in a high plane -> reported
    big = w.copy()
    big *= 1e4
    enc = clb.BertEncoder(big, bcfg, dim=32, gemm="f16x3")
    with pytest.raises(clb.DomainError):
        enc.query_embeddings([1], (ids0.T + 1).astype(np.int32), mask.T)
    enc.close()
    enc = clb.BertEncoder(big, bcfg, dim=32, gemm="bf16x6")        # the bf16 split covers the whole fp32 range
    assert np.isfinite(enc.query_embeddings([1], (ids0.T + 1).astype(np.int32), mask.T)).all()
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["f32", "bf16x6", "f16x3", "bf16x3"])
def test_bert_forward_matches_fp32_reference(gemm):
    torch, cfg, bert, linear = _random_bert()
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    enc = clb.BertEncoder(pack_weights(_state(bert, linear), bcfg, 32), bcfg, dim=32, gemm=gemm)
    rng = np.random.default_rng(1)
    L, N = 37, 5
    ids0 = rng.integers(0, cfg.vocab_size, size=(N, L))
    lens = [37, 20, 1, 33, 9]
    mask = np.zeros((N, L), bool)
    for n, l in enumerate(lens):
        mask[n, :l] = True
    with torch.no_grad():
        ref = linear(bert(input_ids=torch.from_numpy(ids0), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state)
    ref = ref.numpy()                                                            # (N, L, dim)
    got = enc.doc((ids0.T + 1).astype(np.int32), mask.T)                         # (dim, L, N), 1-based ids
    assert got.shape == (32, L, N)
    err = np.abs(got.transpose(2, 1, 0) - ref)[mask].max()                       # padded query rows are don't-care
    assert err < GEMM_TOL[gemm], err
    print(f"[encoder {gemm}] max |got - torch fp32| = {err:.3g}")
    # epilogues on top of the forward == the oracle's epilogue on the same forward output
    from oracle import oracle as orc
    skip = [5, 17, 33]
    D, dl = enc.doc_embeddings_and_doclens(skip, (ids0.T + 1).astype(np.int32), mask.T)
    rD, rdl = orc.doc_epilogue(got, (ids0.T + 1).astype(np.int32), skip)
    assert np.array_equal(dl, rdl) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    Q = enc.query_embeddings(skip, (ids0.T + 1).astype(np.int32), mask.T)
    assert np.array_equal(Q.view(np.uint32), orc.query_epilogue(got, (ids0.T + 1).astype(np.int32), skip).view(np.uint32))
    with pytest.raises(clb.BoundsError):
        enc.doc(np.full((4, 1), cfg.vocab_size + 1, np.int32), np.ones((4, 1), bool))
    # the device-resident query path (fused epilogue kernel, four lanes per token) == the host-buffer path, bit for bit
    d_ids0 = torch.from_numpy((ids0 + 1).astype(np.int32)).cuda()                # (N, L) row-major = Julia (L, N)
    d_mask0 = torch.from_numpy(mask.astype(np.uint8)).cuda()
    d_skip0 = torch.tensor(skip, dtype=torch.int64, device="cuda")
    d_q = torch.empty((N, L, 32), dtype=torch.float32, device="cuda")
    enc.query_embeddings_device(d_ids0, d_mask0, d_skip0, d_q)
    torch.cuda.synchronize()
    assert np.array_equal(np.ascontiguousarray(Q.transpose(2, 1, 0)).view(np.uint32), d_q.cpu().numpy().view(np.uint32))
    # the asynchronous device path clamps such an id when it is enqueued and reports it on request
    d_ids = torch.full((1, 4), cfg.vocab_size + 1, dtype=torch.int32, device="cuda")
    d_mask = torch.ones((1, 4), dtype=torch.uint8, device="cuda")
    d_skip = torch.tensor([1], dtype=torch.int64, device="cuda")
    d_out = torch.empty((1, 4, 32), dtype=torch.float32, device="cuda")
    enc.query_embeddings_device(d_ids, d_mask, d_skip, d_out)
    with pytest.raises(clb.BoundsError):
        enc.check_last_ids()
    enc.query_embeddings_device(torch.ones_like(d_ids), d_mask, d_skip, d_out)
    enc.check_last_ids()
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("L", [33, 100, 200, 300, 470])
def test_fused_attention_long_sequences(L):
    """Head size 64 takes the fused attention kernel (one wave per 32 queries, scores in registers); the key-tile count
    is a template parameter, so every instantiation gets a ragged batch: against torch and against the unfused path."""
    torch, cfg, bert, linear = _random_bert(hidden=128, layers=2, heads=2, inter=256, vocab=90, max_pos=512, dim=32, seed=L)
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    w = pack_weights(_state(bert, linear), bcfg, 32)
    rng = np.random.default_rng(L)
    N = 4
    ids0 = rng.integers(0, cfg.vocab_size, size=(N, L))
    lens = [L, max(1, L // 2), 1, L - 1]
    mask = np.zeros((N, L), bool)
    for n, l in enumerate(lens):
        mask[n, :l] = True
    with torch.no_grad():
        ref = linear(bert(input_ids=torch.from_numpy(ids0), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state).numpy()
    enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f32")
    got = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
    enc.close()
    err = np.abs(got.transpose(2, 1, 0) - ref)[mask].max()
    print(f"[fused attention L={L}] max |got - torch fp32| = {err:.3g}")
    assert err < 2e-4, err
    for other in ("unfused", "resident"):
        enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f32", attention=other)
        alt = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
        enc.close()
        assert np.abs(got - alt).transpose(2, 1, 0)[mask].max() < 1e-4, other
    # the default behind the f16x3 Linear layers: attention on the fp16 planes the Q/K/V projection writes (attention_f16_kernel,
    # three exact products per fp32 product) -- against torch, and against the fp32-MFMA kernel behind the same Linear layers
    enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3")
    g16 = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
    again = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
    enc.close()
    assert np.array_equal(g16.view(np.uint32), again.view(np.uint32))
    err16 = np.abs(g16.transpose(2, 1, 0) - ref)[mask].max()
    print(f"[f16-plane attention L={L}] max |got - torch fp32| = {err16:.3g}")
    assert err16 < 2e-4, err16
    enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", attention="fused_f32")
    alt = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
    enc.close()
    assert np.abs(g16 - alt).transpose(2, 1, 0)[mask].max() < 1e-4
    # round 5: the K / V tiles of a (sequence, head) staged once in LDS for all its query blocks (attention="fused_lds",
    # from 33 tokens on) against every wave loading its own -- the same products in the same order: identical bits
    # (a comparison kernel: in tuning builds of the library only -- the product library refuses the mode)
    if TUNING_BUILD:
        enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", attention="fused_lds")
        per_wave = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
        enc.close()
        assert np.array_equal(g16.view(np.uint32), per_wave.view(np.uint32))
    else:
        with pytest.raises(clb.Unsupported):
            clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", attention="fused_lds")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["two_layers", "wide_three_layers", "long_batch_auto", "hidden_1280"])
def test_layernorm_folded_around_the_linear_layers(shape):
    """Round 5: the LayerNorms as statistics carried between the Linear layers (clb_encoder_set_ln_fold; the producing Linear
    leaves raw rows + per-row partial (mean, M2), the consuming Linear multiplies gamma (.) W and applies
    rstd (a . W'^T - mean u) + c) against the independent fp32 reference at the tolerance of the unfolded path, against the
    unfolded path itself, and deterministic.  `long_batch_auto`: more than 4 096 rows take the folded path on their own
    (what a 64 x 300 passage batch and the packed batches of index() do); padded AND packed.
    `hidden_1280`: a row's statistics travel as hidden / 64 <= 16 partial pairs, so a model wider than 1 024 must
    keep its LayerNorm passes -- the long batch takes the unfolded path bit for bit, and forcing the fold is refused."""
    hidden, layers, heads, inter, L, N = {"two_layers": (64, 2, 1, 128, 37, 5), "wide_three_layers": (192, 3, 3, 320, 50, 7),
                                          "long_batch_auto": (128, 2, 2, 256, 130, 36), "hidden_1280": (1280, 1, 20, 1280, 130, 36)}[shape]
    torch, cfg, bert, linear = _random_bert(hidden=hidden, layers=layers, heads=heads, inter=inter, vocab=150, max_pos=160, dim=32, seed=11)
    with torch.no_grad():                       # LayerNorm parameters away from (1, 0): the folded vectors u, c must carry them
        for name, p_ in bert.named_parameters():
            if "LayerNorm.weight" in name:
                p_.copy_(1.0 + 0.3 * torch.randn_like(p_))
            if "LayerNorm.bias" in name:
                p_.copy_(0.2 * torch.randn_like(p_))
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    w = pack_weights(_state(bert, linear), bcfg, 32)
    rng = np.random.default_rng(4)
    ids0 = rng.integers(0, cfg.vocab_size, size=(N, L))
    lens = rng.integers(L // 2, L + 1, size=N); lens[0] = L
    mask = np.zeros((N, L), bool)
    for n, l in enumerate(lens):
        mask[n, :l] = True
    ids0[~mask] = 4                              # padding carries an id of the skiplist below (as [PAD] does): dropped by the epilogue
    with torch.no_grad():
        ref = linear(bert(input_ids=torch.from_numpy(ids0), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state).numpy()
    jl_ids, jl_mask = (ids0.T + 1).astype(np.int32), mask.T
    plain = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", ln_fold=0)
    base = plain.doc(jl_ids, jl_mask)
    plain.close()
    if shape == "hidden_1280":
        with pytest.raises(clb.Unsupported):
            clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", ln_fold=1)
        enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3")            # default: folds long batches -- where it can
        got = enc.doc(jl_ids, jl_mask)
        enc.close()
        assert N * L > 4096 and np.array_equal(got.view(np.uint32), base.view(np.uint32))
        err = np.abs(got.transpose(2, 1, 0) - ref)[mask].max()
        assert err < 2 * GEMM_TOL["f16x3"], err
        return
    enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3", ln_fold=-1 if shape == "long_batch_auto" else 1)
    got = enc.doc(jl_ids, jl_mask)
    again = enc.doc(jl_ids, jl_mask)
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    if shape == "long_batch_auto":
        assert N * L > 4096 and not np.array_equal(got.view(np.uint32), base.view(np.uint32))       # it did take the other path
    err = np.abs(got.transpose(2, 1, 0) - ref)[mask].max()
    err0 = np.abs(base.transpose(2, 1, 0) - ref)[mask].max()
    print(f"[LayerNorm fold, {shape}] max |got - torch fp32| = {err:.3g} (unfolded: {err0:.3g})")
    assert err < GEMM_TOL["f16x3"] and err < 3 * err0 + 2e-5, (err, err0)
    assert np.abs(got - base).transpose(2, 1, 0)[mask].max() < 1e-4
    # the document epilogue on top, and packed batches (every row attended, no padding rows) through the same folded Linears
    skip = [5, 17, 33]
    D, dl = enc.doc_embeddings_and_doclens(skip, jl_ids, jl_mask)
    keep = mask & ~np.isin(ids0 + 1, skip)
    assert np.array_equal(dl, keep.sum(axis=1))
    flat = ref[keep]
    flat = flat / (np.linalg.norm(flat, axis=1, keepdims=True) + np.finfo(np.float32).eps)
    assert np.abs(D.T - flat).max() < 2e-4
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["f32", "bf16x6", "f16x3", "bf16x3"])
def test_bert_base_shape_and_export_roundtrip(tmp_path, tok, gemm):
    """bert-base-uncased geometry (12 x 768, 12 heads, FFN 3072), one short batch; weights through the export tool."""
    torch, cfg, bert, linear = _random_bert(hidden=768, layers=12, heads=12, inter=3072, vocab=len(VOCAB), max_pos=64, dim=128, seed=3)
    hf = tmp_path / "hf"; hf.mkdir()
    sd = {"bert." + k: v for k, v in bert.state_dict().items()}
    sd["linear.weight"] = linear.weight.detach()
    torch.save(sd, hf / "pytorch_model.bin")
    (hf / "config.json").write_text(json.dumps(cfg.to_dict()))
    (hf / "artifact.metadata").write_text(json.dumps({"dim": 128}))
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "export_checkpoint.py"), str(hf), str(tmp_path / "exp")], check=True)
    enc = clb.BertEncoder.from_export(str(tmp_path / "exp"), tokenizer=tok, gemm=gemm,
                                      config=clb.ColBERTConfig(doc_maxlen=24, query_maxlen=12, index_bsize=2))
    passages = ["hello world!", "this is a longer passage with many words.", "a test of the tokenizer"]
    embs, doclens = enc.encode_passages(passages)
    assert embs.shape == (128, doclens.sum()) and np.allclose(np.linalg.norm(embs, axis=0), 1, atol=1e-5)
    ids, mask = tokenization.tensorize_docs("[unused1]", tok, passages, 24)
    with torch.no_grad():
        h = bert(input_ids=torch.from_numpy((ids.T - 1).astype(np.int64)), attention_mask=torch.from_numpy(mask.T.astype(np.int64))).last_hidden_state
        ref = (h @ linear.weight.T).numpy()                                       # exported without linear.bias -> zeros
    keep = ~np.isin(ids, tok.doc_skiplist(True))
    assert np.array_equal(doclens, keep.sum(axis=0))
    flat = ref.transpose(0, 1, 2).reshape(-1, 128)[keep.T.ravel()]
    flat = flat / (np.linalg.norm(flat, axis=1, keepdims=True) + np.finfo(np.float32).eps)
    err = np.abs(embs.T - flat).max()
    print(f"[encoder {gemm}, bert-base geometry] max |got - torch fp32| = {err:.3g}")
    assert err < GEMM_TOL_BASE[gemm], err     # 12 layers of fp32 rounding in two different summation orders (+ the split)
    Q = enc.encode_queries(["what is col bert", "hello"])
    assert Q.shape == (128, 12, 2) and np.allclose(np.linalg.norm(Q, axis=0), 1, atol=1e-5)
    enc.close()


# the four passages of the reference's examples/sample_collection.tsv (BASELINE config 1), as data
SAMPLE_COLLECTION = ["hello world", "thank yo!", "a", "this is some longer text, so length should be longer"]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["synthetic", "sample_collection"])
def test_text_to_search_end_to_end(tmp_path, tok, which, monkeypatch):
    """BASELINE config 1 (a handful of passages through Indexer + Searcher with a text query): the API plumbing of
    examples/indexing.jl + examples/searching.jl with a randomly initialised encoder (no checkpoint is on disk) -- on
    synthetic sentences and on the text of the reference's examples/sample_collection.tsv."""
    torch, cfg, bert, linear = _random_bert(hidden=64, layers=2, heads=4, inter=128, vocab=len(VOCAB), max_pos=64, dim=128, seed=5)
    from colbert_jl_amd.encoder import pack_weights
    from oracle import oracle as orc
    bcfg = cfg.to_dict()
    config = clb.ColBERTConfig(index_path=str(tmp_path / "short_index"), doc_maxlen=24, query_maxlen=12, index_bsize=4,
                               chunksize=6, kmeans_niters=3, nbits=2)
    enc = clb.BertEncoder(pack_weights(_state(bert, linear), bcfg, 128), bcfg, dim=128, tokenizer=tok, config=config)
    words = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with", "many", "words", "query", "colbert"]
    rng = np.random.default_rng(7)
    collection = [" ".join(rng.choice(words, size=rng.integers(4, 12))) + "." for _ in range(10)]
    if which == "sample_collection":
        collection = list(SAMPLE_COLLECTION)
    n = len(collection)
    indexer = clb.Indexer(config, encoder=enc, collection=collection, seed=1)
    assert clb.index(indexer) == config.index_path
    searcher = clb.Searcher(config.index_path, encoder=enc)
    query = collection[3]
    pids, scores = clb.search(searcher, query, 3)
    assert pids.shape == (3,) and np.all(np.diff(scores) <= 0) and np.all((pids >= 1) & (pids <= n))
    assert pids[0] == 4            # the passage the query was copied from
    # same result as the oracle on the same query embeddings and index arrays
    from colbert_jl_amd import storage
    idx = storage.load_index(config.index_path)
    Q = enc.encode_queries([query])[:, :, 0]
    rp, rs, _ = orc.search(idx, Q, config.nprobe, 3)
    assert np.array_equal(pids, rp) and np.array_equal(scores.view(np.uint32), rs.view(np.uint32))
    with pytest.raises(clb.BoundsError):
        clb.search(searcher, query, n + 1)
    # the serving session (tokenizer on the host, encode -> search on the device without a host round trip), as stream
    # launches and as ONE captured HIP graph: the same pids and the same score bits for every query
    for graph in (False, True):
        ts = searcher.text_search(3, graph=graph)
        for q in (query, collection[0], "hello world", collection[-1]):
            want = clb.search(searcher, q, 3)
            got = ts(q)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), (graph, q)
        ts.close()
    # the throughput shape of the same session (TextSearch.search_many: encode in groups, search in batches, the next encode
    # beside the later batches of a group): every query's result is the single-query one, bit for bit, in query order
    ts = searcher.text_search(3, graph=False)
    many_q = [collection[i % n] for i in range(21)] + ["hello world", query]
    for group, batch in ((8, 4), (16, 16)):
        got_many = ts.search_many(many_q, group=group, batch=batch)
        assert len(got_many) == len(many_q)
        for q, (gp, gs) in zip(many_q, got_many):
            want = clb.search(searcher, q, 3)
            assert np.array_equal(gp, want[0]) and np.array_equal(gs.view(np.uint32), want[1].view(np.uint32)), (group, batch, q)
    ts.close()
    with pytest.raises(clb.BoundsError):
        searcher.text_search(n + 1, graph=False)(query)
    # the session copies the encoder's sticky error flag back with every result (clb_encoder_error_flag_device): what the
    # asynchronous encode can only clamp -- here an id outside the vocabulary -- is reported by that very call, and cleared
    import colbert_jl_amd.tokenization as tkz
    real = tkz.tensorize_queries

    def bad(*a, **k):
        ids, mask = real(*a, **k)
        ids = ids.copy(); ids[2, 0] = 10 ** 6
        return ids, mask
    ts = searcher.text_search(3, graph=True)
    monkeypatch.setattr(tkz, "tensorize_queries", bad)
    with pytest.raises(clb.BoundsError):
        ts(query)
    monkeypatch.setattr(tkz, "tensorize_queries", real)
    got = ts(query)
    assert np.array_equal(got[0], pids) and np.array_equal(got[1].view(np.uint32), scores.view(np.uint32))
    ts.close()
    searcher.close(); enc.close()


@pytest.mark.gpu
def test_index_device_resident_route_writes_the_same_files(tmp_path, tok):
    """index() with the encoder's output left in HBM (clb_encode_docs_device -> index_device) against the host-buffer
    route: the same embeddings and doclens per batch, and byte-identical index directories."""
    import filecmp
    torch, cfg, bert, linear = _random_bert(hidden=64, layers=2, heads=4, inter=128, vocab=len(VOCAB), max_pos=64, dim=128, seed=9)
    from colbert_jl_amd.encoder import pack_weights
    from colbert_jl_amd.indexer import EncoderSource
    bcfg = cfg.to_dict()
    words = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with", "many", "words", "query", "colbert"]
    rng = np.random.default_rng(11)
    collection = [" ".join(rng.choice(words, size=rng.integers(1, 14))) + rng.choice([".", "!", "", " , ok"]) for _ in range(23)]
    paths = {}
    for route in ("host", "device"):
        config = clb.ColBERTConfig(index_path=str(tmp_path / route), doc_maxlen=24, query_maxlen=12, index_bsize=5,
                                   chunksize=7, kmeans_niters=3, nbits=2)
        enc = clb.BertEncoder(pack_weights(_state(bert, linear), bcfg, 128), bcfg, dim=128, tokenizer=tok, config=config)
        if route == "device":
            src = EncoderSource(enc, collection, 0)
            want, want_dl = enc.encode_passages(collection)
            assert np.array_equal(src.doclens, want_dl)                 # doclens from tokenisation alone
            got, got_dl = src.encode()
            assert np.array_equal(got_dl, want_dl)
            assert np.array_equal(got.cpu().numpy().view(np.uint32), np.ascontiguousarray(want.T).view(np.uint32))
            x = src.sample(np.array([2, 3, 17]))
            ref = enc.encode_passages([collection[i] for i in (2, 3, 17)])[0]
            assert np.array_equal(x.cpu().numpy().view(np.uint32), np.ascontiguousarray(ref.T).view(np.uint32))
        indexer = clb.Indexer(config, encoder=enc, collection=collection, seed=3)
        assert clb.index(indexer, device_resident=(route == "device")) == config.index_path
        paths[route] = config.index_path
        enc.close()
    names = sorted(os.listdir(paths["host"]))
    assert names == sorted(os.listdir(paths["device"])) and len(names) >= 12
    for n in names:
        if n == "config.json":
            continue                                                    # holds its own index_path
        assert filecmp.cmp(os.path.join(paths["host"], n), os.path.join(paths["device"], n), shallow=False), n


@pytest.mark.gpu
def test_fp16_plane_attention_survives_batch_shape_changes():
    """attention_f16_kernel reads V from a key-blocked buffer whose slots past a sequence's end are never written: one
    encoder object run over batches of changing shape (as index() does: every batch is padded to its own longest passage)
    returns, for every batch, the bits a fresh encoder returns."""
    torch, cfg, bert, linear = _random_bert(hidden=128, layers=2, heads=2, inter=256, vocab=90, max_pos=512, dim=32, seed=21)
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    w = pack_weights(_state(bert, linear), bcfg, 32)
    rng = np.random.default_rng(22)
    shapes = [(70, 5), (45, 3), (300, 2), (20, 3), (33, 7), (70, 5), (64, 1), (129, 1)]     # 20 x 3, 64 x 1: the one-query (split-K) plans
    batches = []
    for L, N in shapes:
        ids = (rng.integers(0, cfg.vocab_size, size=(L, N)) + 1).astype(np.int32)
        mask = np.zeros((L, N), bool)
        for n in range(N):
            mask[: rng.integers(1, L + 1), n] = True
        mask[:, 0] = True
        batches.append((ids, mask))
    enc = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3")
    got = [enc.doc(ids, mask) for ids, mask in batches]
    enc.close()
    for (ids, mask), g in zip(batches, got):
        fresh = clb.BertEncoder(w, bcfg, dim=32, gemm="f16x3")
        want = fresh.doc(ids, mask)
        fresh.close()
        assert np.array_equal(g.view(np.uint32)[:, mask], want.view(np.uint32)[:, mask]), ids.shape
        assert np.isfinite(g).all()


@pytest.mark.gpu
def test_passage_batch_plans_with_an_odd_number_of_heads():
    """A batch long enough for the passage-batch tile plans (128 x 256 / 256 x 256 work-group tiles) on a model with three
    heads: 2H = 384 is not a multiple of the 256-wide tile, so one tile of the Q/K/V projection holds K columns and V
    columns side by side (EPI_QKV_ATT switches layout inside it), and N = 576 / 192 end in partial tiles."""
    torch, cfg, bert, linear = _random_bert(hidden=192, layers=1, heads=3, inter=320, vocab=90, max_pos=512, dim=64, seed=31)
    from colbert_jl_amd.encoder import pack_weights
    bcfg = cfg.to_dict()
    w = pack_weights(_state(bert, linear), bcfg, 64)
    rng = np.random.default_rng(32)
    N, L = 56, 300
    ids0 = rng.integers(0, cfg.vocab_size, size=(N, L))
    mask = np.zeros((N, L), bool)
    for n in range(N):
        mask[n, : rng.integers(1, L + 1)] = True
    mask[0, :] = True
    with torch.no_grad():
        ref = linear(bert(input_ids=torch.from_numpy(ids0), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state).numpy()
    outs = {}
    for attention in ("fused", "fused_f32"):
        enc = clb.BertEncoder(w, bcfg, dim=64, gemm="f16x3", attention=attention)
        outs[attention] = enc.doc((ids0.T + 1).astype(np.int32), mask.T)
        enc.close()
        err = np.abs(outs[attention].transpose(2, 1, 0) - ref)[mask].max()
        print(f"[passage-batch plans, 3 heads, {attention}] max |got - torch fp32| = {err:.3g}")
        assert err < 1e-4, (attention, err)
    assert np.abs(outs["fused"] - outs["fused_f32"]).transpose(2, 1, 0)[mask].max() < 5e-5


@pytest.mark.gpu
def test_packed_passage_batches_match_padded_ones(tok):
    """EncoderSource encodes a batch without its padding rows (clb_encode_docs_packed_device): the same doclens and, per kept
    token, the embedding of the padded batch up to the rounding of a different tile plan; an encoder that cannot run packed
    batches (here: the bf16x6 Linear layers) falls back to padding, bit for bit."""
    torch, cfg, bert, linear = _random_bert(hidden=128, layers=2, heads=2, inter=256, vocab=len(VOCAB), max_pos=64, dim=64, seed=41)
    from colbert_jl_amd.encoder import pack_weights
    from colbert_jl_amd.indexer import EncoderSource
    bcfg = cfg.to_dict()
    words = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with", "many", "words", "query", "colbert"]
    rng = np.random.default_rng(42)
    collection = [" ".join(rng.choice(words, size=rng.integers(1, 30))) + rng.choice([".", "!", "", " , ok"]) for _ in range(37)]
    config = clb.ColBERTConfig(index_path="unused", doc_maxlen=40, query_maxlen=12, index_bsize=8, nbits=2)
    w = pack_weights(_state(bert, linear), bcfg, 64)
    enc = clb.BertEncoder(w, bcfg, dim=64, tokenizer=tok, config=config)
    packed = EncoderSource(enc, collection, 0, packed=True)
    padded = EncoderSource(enc, collection, 0, packed=False)
    order = np.arange(len(collection))
    a = packed.encode_pids(order)
    assert packed.packed is True                                     # head size 64, f16x3: the packed path ran
    b = padded.encode_pids(order)
    assert a.shape == b.shape == (int(packed.doclens.sum()), 64)
    assert float((a - b).abs().max()) < 5e-5
    assert torch.allclose(a.norm(dim=1), torch.ones(a.shape[0], device=a.device), atol=1e-5)
    some = np.array([3, 4, 20, 36])
    smp = packed.sample(some)
    assert float((smp - padded.sample(some)).abs().max()) < 5e-5
    # round 5: the sampled passages' embeddings are kept and a chunk encodes only its other passages (packed route only):
    # the chunk in passage order, the sampled passages' rows the sample's bit for bit, the others as any packed batch
    assert packed._cache is not None and padded._cache is None
    off = np.concatenate([[0], np.cumsum(packed.doclens)])
    soff = np.concatenate([[0], np.cumsum(packed.doclens[some])])
    for lo, hi in ((0, 20), (20, 37)):
        x = packed.chunk(lo, hi)
        assert x.shape[0] == off[hi] - off[lo]
        assert float((x - a[off[lo]:off[hi]]).abs().max()) < 5e-5
        for j, pid in enumerate(some):
            if lo <= pid < hi:
                assert torch.equal(x[off[pid] - off[lo]:off[pid + 1] - off[lo]], smp[soff[j]:soff[j + 1]])
    assert packed.reused_passages == 4 and packed._cache is None      # released with the last chunk
    # packed sequences of different lengths through the LDS-shared K / V tiles == every wave loading its own, bit for bit
    if TUNING_BUILD:
        enc_pw = clb.BertEncoder(w, bcfg, dim=64, tokenizer=tok, config=config, attention="fused_lds")
        a_pw = EncoderSource(enc_pw, collection, 0, packed=True).encode_pids(order)
        assert torch.equal(a.view(torch.int32), a_pw.view(torch.int32))
        enc_pw.close()
    # the host entry point (clb_encode_docs: what the Julia shim calls) packs by itself -- also a mask with holes, whose
    # attended tokens keep their positions; reference: the same call on an encoder that cannot pack (fp32-MFMA attention)
    ref_enc = clb.BertEncoder(w, bcfg, dim=64, tokenizer=tok, config=config, attention="fused_f32")
    hd, hl = enc.encode_passages(collection)
    rd, rl = ref_enc.encode_passages(collection)
    assert np.array_equal(hl, rl) and np.array_equal(hl, packed.doclens) and hd.shape == rd.shape
    assert np.abs(hd - rd).max() < 5e-5
    ids, mask = tokenization.tensorize_docs(config.doc_token_id, tok, collection[:8], config.doc_maxlen)
    ids = np.array(ids); mask = np.array(mask, dtype=bool)
    holes = (rng.random(ids.shape) < 0.2) & mask
    holes[:2, :] = False
    ids[holes] = tok.pad_id; mask[holes] = False
    skip = tok.doc_skiplist(True)
    g_e, g_l = enc.doc_embeddings_and_doclens(skip, ids, mask)
    r_e, r_l = ref_enc.doc_embeddings_and_doclens(skip, ids, mask)
    assert np.array_equal(g_l, r_l) and g_e.shape == r_e.shape and np.abs(g_e - r_e).max() < 5e-5
    # an unattended token the skiplist does NOT drop must be computed: the call then takes the padded path
    ids2 = ids.copy(); ids2[holes] = tok.lookup("hello")
    g_e, g_l = enc.doc_embeddings_and_doclens(skip, ids2, mask)
    r_e, r_l = ref_enc.doc_embeddings_and_doclens(skip, ids2, mask)
    assert np.array_equal(g_l, r_l) and np.abs(g_e - r_e).max() < 5e-5
    ref_enc.close()
    enc.close()
    enc6 = clb.BertEncoder(w, bcfg, dim=64, tokenizer=tok, config=config, gemm="bf16x6")
    fallback = EncoderSource(enc6, collection, 0, packed=True)
    c = fallback.encode_pids(order)
    assert fallback.packed is False
    d = EncoderSource(enc6, collection, 0, packed=False).encode_pids(order)
    assert torch.equal(c, d)
    enc6.close()
