"""tools/run_config3.py (BASELINE config 3: HuggingFace ColBERT checkpoint + LoTTE-style collection / queries -> export ->
index() -> Searcher -> search, Success@k) on a FABRICATED HF-layout checkpoint and a 50-line TSV, so that the tool cannot
rot while the real assets (colbert-ir/colbertv2.0, LoTTE) are not in the image.  CPU: everything before the first device
call (export, parsing, metrics, the README parser) and that the run then fails loudly without a GPU.  GPU: the whole run."""
import importlib.util
import json
import os

import numpy as np
import pytest

import colbert_jl_amd as clb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with", "many", "words",
         "query", "colbert", "rabbit", "garden", "spots", "trick", "puppy", "tail", "urban", "fear"]
VOCAB = (["[PAD]", "[unused0]", "[unused1]"] + [f"[unused{i}]" for i in range(2, 99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] +
         list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + [str(d) for d in range(10)] + WORDS)


def _tool():
    spec = importlib.util.spec_from_file_location("run_config3", os.path.join(ROOT, "tools", "run_config3.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fabricate(tmp_path, n_passages=50, seed=3, word_scale=8.0, pos_scale=0.1):
    """An HF-layout checkpoint directory with colbertv2.0's key names (bert.* + linear.weight), a "pid<TAB>text" collection,
    LoTTE-style questions / qas files (every query is the text of one passage, whose pid is its answer)."""
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    from safetensors.numpy import save_file
    torch.manual_seed(seed)
    hf = tmp_path / "colbert_ckpt"
    hf.mkdir()
    cfg = transformers.BertConfig(vocab_size=len(VOCAB), hidden_size=64, num_hidden_layers=2, num_attention_heads=1,
                                  intermediate_size=128, max_position_embeddings=64, type_vocab_size=2, hidden_act="gelu")
    bert = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in bert.parameters():
            p.mul_(4.0)
        # a randomly initialised encoder has to be LEXICAL for the retrieval check to mean anything: token identity must
        # dominate position and marker (a trained ColBERT gets that from its training)
        bert.embeddings.word_embeddings.weight.mul_(word_scale)
        bert.embeddings.position_embeddings.weight.mul_(pos_scale)
    state = {"bert." + k: v.detach().float().numpy().copy() for k, v in bert.state_dict().items()}
    state["linear.weight"] = torch.nn.Linear(64, 128, bias=False).weight.detach().numpy().copy()      # colbertv2.0 has no linear.bias
    save_file(state, str(hf / "model.safetensors"))
    (hf / "config.json").write_text(json.dumps(cfg.to_dict()))
    (hf / "artifact.metadata").write_text(json.dumps({"dim": 128, "query_maxlen": 32, "doc_maxlen": 300}))
    (hf / "vocab.txt").write_text("\n".join(VOCAB) + "\n")
    rng = np.random.default_rng(seed)
    passages = [" ".join(rng.choice(WORDS, size=rng.integers(6, 14))) + "." for _ in range(n_passages)]
    coll = tmp_path / "collection.tsv"
    coll.write_text("".join(f"{i}\t{p}\n" for i, p in enumerate(passages)))                     # LoTTE: 0-based pid column
    picks = rng.choice(n_passages, size=12, replace=False)
    ques = tmp_path / "questions.search.tsv"
    ques.write_text("".join(f"{q}\t{passages[p]}\n" for q, p in enumerate(picks)))
    qas = tmp_path / "qas.search.jsonl"
    qas.write_text("".join(json.dumps({"qid": q, "query": passages[p], "answer_pids": [int(p)]}) + "\n" for q, p in enumerate(picks)))
    return str(hf), str(coll), str(ques), str(qas), passages, picks


def test_parsers_and_metrics(tmp_path):
    t = _tool()
    hf, coll, ques, qas, passages, picks = _fabricate(tmp_path)
    c = t.read_collection(coll)
    assert len(c) == 50 and c[7] == f"7\t{passages[7]}"                                 # whole lines, as readlines() gives them
    q = t.read_queries(ques)
    assert [x[0] for x in q] == list(range(12)) and q[3][1] == passages[picks[3]]
    plain = tmp_path / "plain.txt"
    plain.write_text("what are white spots on raspberries?\nare rabbits easy to housebreak?\n")   # examples/searching.jl's queries
    assert t.read_queries(str(plain)) == [(0, "what are white spots on raspberries?"), (1, "are rabbits easy to housebreak?")]
    answers = t.read_qas(qas)
    assert answers[5] == {int(picks[5])}
    ids = t.lotte_pid_of_line(c)
    assert ids == list(range(50)) and t.lotte_pid_of_line(["no id column"]) is None
    # results are 1-based line numbers; line i holds LoTTE pid i - 1
    res = {qi: [int(picks[qi]) + 1, 1, 2] for qi in range(12)}
    res[0] = [50, 49, int(picks[0]) + 1]                                                  # the answer only at rank 3
    res[1] = [1 if picks[1] != 0 else 2] * 3                                              # a miss
    m = t.success_at(res, answers, ids, ks=(1, 3))
    assert m["judged_queries"] == 12 and m["success@1"] == round(10 / 12, 4) and m["success@3"] == round(11 / 12, 4)


def test_readme_parser(tmp_path):
    t = _tool()
    md = tmp_path / "README.md"
    md.write_text('intro\n```julia\njulia>  document_passages = [\n    "first \\"quoted\\" passage, costs \\$5 [not a bracket]",\n'
                  '    "second passage",\n    "third"\n]\n```\ntext\n```julia\njulia>  query = "what was Cesar Milan\'s trick?";\n```\n')
    passages, query = t.parse_readme_example(str(md))
    assert passages == ['first "quoted" passage, costs $5 [not a bracket]', "second passage", "third"]
    assert query == "what was Cesar Milan's trick?"
    ref = "/root/reference/README.md"                  # the real file, when the reference checkout is present (not on the GPU box)
    if os.path.exists(ref):
        passages, query = t.parse_readme_example(ref)
        assert len(passages) == 10 and query == t.README_EXPECTED["query"] and "Cesar Milan" in passages[9]


def test_export_then_fails_loudly_without_a_gpu(tmp_path):
    """The fabricated checkpoint exports to the encoder blob (colbertv2.0 key layout: bert.* prefix, linear.weight without a
    bias); without a GPU the run stops at the first device call with HipError -- there is no CPU path to fall into."""
    t = _tool()
    hf, coll, ques, qas, _, _ = _fabricate(tmp_path)
    out = t.export_checkpoint(hf, str(tmp_path / "export"))
    meta = json.load(open(os.path.join(out, "encoder.json")))
    assert meta["dim"] == 128 and meta["bert"]["hidden_size"] == 64
    assert os.path.getsize(os.path.join(out, "encoder.f32")) == 4 * meta["n_floats"]
    if clb.lib().clb_device_count() == 0:
        with pytest.raises(clb.HipError):
            t.main([hf, coll, ques, "--qas", qas, "--workdir", str(tmp_path / "run"), "--k", "5"])
        assert not os.path.isdir(tmp_path / "run" / "index")                                # nothing half-built left behind


@pytest.mark.gpu
def test_config3_tool_end_to_end(tmp_path):
    t = _tool()
    hf, coll, ques, qas, passages, picks = _fabricate(tmp_path)
    rec = t.main([hf, coll, ques, "--qas", qas, "--workdir", str(tmp_path / "run"), "--k", "5", "--doc-maxlen", "40",
                  "--out", str(tmp_path / "rec.json")])
    assert rec["index_built"] and rec["index_s"] > 0 and rec["search"]["queries"] == 12
    assert rec["lotte"]["judged_queries"] == 12
    assert rec["lotte"]["success@5"] >= 0.9 and rec["lotte"]["success@1"] >= 0.75            # every query IS a passage of the collection
    assert json.load(open(tmp_path / "rec.json"))["lotte"] == rec["lotte"]
    # a second run finds the index directory and does not rebuild (indexing.jl:64-67)
    rec2 = t.main([hf, coll, ques, "--qas", qas, "--workdir", str(tmp_path / "run"), "--k", "5", "--doc-maxlen", "40"])
    assert rec2["index_built"] is False and rec2["lotte"] == rec["lotte"]
