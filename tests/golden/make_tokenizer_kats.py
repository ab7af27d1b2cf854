#!/usr/bin/env python3
"""Transcribes the recorded tokenizer outputs that ColBERT.jl's docstrings hold (REPL sessions run by the
reference's authors against the colbertv2.0 / bert-base-uncased vocabulary) into
tests/golden/tokenizer_kats.json:

  src/modelling/tokenization/doc_tokenization.jl:61-141    tensorize_docs,    doc_maxlen 20, "[unused1]"
  src/modelling/tokenization/query_tokenization.jl:55-171  tensorize_queries, query_maxlen 32, "[unused0]"

Run in the build container, where the reference checkout is mounted at /root/reference:
    python tests/golden/make_tokenizer_kats.py
Only DATA is extracted: the input strings, the printed integer_ids / bitmask matrices and the printed
decode() matrix.  The decode() matrix gives the vocabulary entry of every id that occurs, which is what lets
the test rebuild the slice of bert-base-uncased's vocab.txt these examples touch (no vocabulary file exists
on the build or the GPU machines).  Ids are Julia's 1-based lookups; vocab.txt line = id - 1.
"""
import json
import os
import re

REF = os.environ.get("COLBERT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tokenizer_kats.json")


def _block_after(text, prompt, start=0):
    """Lines of REPL output following the line that starts with `prompt` up to the next blank line."""
    i = text.index(prompt, start)
    j = text.index("\n", i) + 1
    k = text.index("\n\n", j)
    return text[j:k].split("\n"), k


def _matrix(lines, conv):
    rows = []
    for ln in lines[1:]:                                        # lines[0] is the "R×C Matrix{T}:" header
        rows.append([conv(tok) for tok in ln.split()])
    hdr = re.match(r"\s*(\d+)×(\d+) Matrix", lines[0])
    assert hdr and len(rows) == int(hdr.group(1)) and all(len(r) == int(hdr.group(2)) for r in rows), lines[0]
    return rows


def _strings(lines):
    rows = [re.findall(r'"((?:[^"\\]|\\.)*)"', ln) for ln in lines[1:] if ln.strip() != "```"]
    hdr = re.match(r"\s*(\d+)×(\d+) Matrix", lines[0])
    assert hdr and len(rows) == int(hdr.group(1)) and all(len(r) == int(hdr.group(2)) for r in rows), lines[0]
    return rows


def _texts(text):
    i = text.index("julia> batch_text = [")
    j = text.index("];", i)
    src = text[i:j]
    # Julia string concatenation with `*` across lines: join the pieces of one element
    elems, cur = [], None
    for ln in src.split("\n")[1:]:
        parts = re.findall(r'"((?:[^"\\]|\\.)*)"', ln)
        if not parts:
            continue
        piece = "".join(parts)
        cur = piece if cur is None else cur + piece
        if not ln.rstrip().endswith("*"):
            elems.append(cur)
            cur = None
    assert cur is None
    return elems


def parse(relpath, marker, maxlen, kind):
    text = open(os.path.join(REF, relpath)).read()
    # make sure a terminating blank line exists after the final matrix inside the docstring
    text = text.replace('```\n"""', '\n\n"""')
    ids_lines, pos = _block_after(text, "julia> integer_ids\n")
    mask_lines, pos = _block_after(text, "julia> bitmask", pos)
    dec_lines, pos = _block_after(text, "julia> TextEncoders.decode(tokenizer, integer_ids)", pos)
    ids = _matrix(ids_lines, int)
    mask = _matrix(mask_lines, int)
    dec = _strings(dec_lines)
    assert len(ids) == len(mask) == len(dec) == maxlen
    vocab = {}
    for r_i, r_s in zip(ids, dec):
        for i, s in zip(r_i, r_s):
            assert vocab.setdefault(i, s) == s, (i, s, vocab[i])
    return {"source": relpath, "kind": kind, "marker": marker, "maxlen": maxlen, "texts": _texts(text),
            "integer_ids": ids, "bitmask": mask,
            "vocab_1based": {str(k): v for k, v in sorted(vocab.items())}}


if __name__ == "__main__":
    out = {
        "note": "ids are 1-based (Julia); vocab.txt line number (0-based) = id - 1; bert-base-uncased has 30522 lines",
        "vocab_size": 30522,
        "docs": parse("src/modelling/tokenization/doc_tokenization.jl", "[unused1]", 20, "docs"),
        "queries": parse("src/modelling/tokenization/query_tokenization.jl", "[unused0]", 32, "queries"),
    }
    json.dump(out, open(OUT, "w"), indent=1, ensure_ascii=False)
    print("wrote", OUT, {k: (len(v["texts"]), len(v["vocab_1based"])) for k, v in out.items() if isinstance(v, dict)})
