# Text -> token ids on the host (src/modelling/tokenization/*.jl): BERT's uncased basic tokenizer + WordPiece over
# the checkpoint's vocab.txt, then the reference's tensorize rules.  Ids are 1-based line numbers of vocab.txt,
# as in the reference ([PAD] = 1, [unused0] = 2, [unused1] = 3, [UNK] = 101, [CLS] = 102, [SEP] = 103, [MASK] = 104
# for bert-base-uncased).

struct WordPieceTokenizer
    vocab::Dict{String, Int32}
    lowercase::Bool
end

function WordPieceTokenizer(vocab_file::String; lowercase::Bool = true)
    vocab = Dict{String, Int32}()
    for (i, line) in enumerate(eachline(vocab_file))
        vocab[String(rstrip(line, ['\r', '\n']))] = Int32(i)
    end
    WordPieceTokenizer(vocab, lowercase)
end

"lookup(vocab, token): a token that is not in the vocabulary maps to [UNK] -- this is what happens to the query marker \"[Q]\" the reference passes (src/searching.jl:96)"
lookup(t::WordPieceTokenizer, token::AbstractString) = get(t.vocab, String(token), t.vocab["[UNK]"])

_is_cjk(c::Char) = (0x4E00 <= UInt32(c) <= 0x9FFF) || (0x3400 <= UInt32(c) <= 0x4DBF) ||
                   (0x20000 <= UInt32(c) <= 0x2A6DF) || (0x2A700 <= UInt32(c) <= 0x2B73F) ||
                   (0x2B740 <= UInt32(c) <= 0x2B81F) || (0x2B820 <= UInt32(c) <= 0x2CEAF) ||
                   (0xF900 <= UInt32(c) <= 0xFAFF) || (0x2F800 <= UInt32(c) <= 0x2FA1F)
function _is_punct(c::Char)
    u = UInt32(c)
    ((33 <= u <= 47) || (58 <= u <= 64) || (91 <= u <= 96) || (123 <= u <= 126)) && return true
    Base.Unicode.category_code(c) in (Base.Unicode.UTF8PROC_CATEGORY_PC, Base.Unicode.UTF8PROC_CATEGORY_PD, Base.Unicode.UTF8PROC_CATEGORY_PS,
        Base.Unicode.UTF8PROC_CATEGORY_PE, Base.Unicode.UTF8PROC_CATEGORY_PI, Base.Unicode.UTF8PROC_CATEGORY_PF,
        Base.Unicode.UTF8PROC_CATEGORY_PO)
end
_is_control(c::Char) = !(c in ('\t', '\n', '\r')) &&
                       Base.Unicode.category_code(c) in (Base.Unicode.UTF8PROC_CATEGORY_CC, Base.Unicode.UTF8PROC_CATEGORY_CF)

"BERT's BasicTokenizer: clean, isolate CJK, split on whitespace, lowercase + strip accents, split punctuation"
function _basic_tokens(t::WordPieceTokenizer, text::AbstractString)
    buf = IOBuffer()
    for c in text
        (UInt32(c) == 0 || UInt32(c) == 0xFFFD || _is_control(c)) && continue
        if isspace(c)
            print(buf, ' ')
        elseif _is_cjk(c)
            print(buf, ' ', c, ' ')
        else
            print(buf, c)
        end
    end
    out = String[]
    for word in split(String(take!(buf)))
        w = String(word)
        if t.lowercase
            w = lowercase(w)
            w = filter(c -> Base.Unicode.category_code(c) != Base.Unicode.UTF8PROC_CATEGORY_MN, Unicode.normalize(w, :NFD))
        end
        cur = IOBuffer()
        for c in w
            if _is_punct(c)
                position(cur) > 0 && push!(out, String(take!(cur)))
                push!(out, string(c))
            else
                print(cur, c)
            end
        end
        position(cur) > 0 && push!(out, String(take!(cur)))
    end
    out
end

"greedy longest-match-first WordPiece; words longer than 100 characters or without a match become [UNK]"
function _wordpiece!(ids::Vector{Int32}, t::WordPieceTokenizer, word::String)
    chars = collect(word)
    unk = t.vocab["[UNK]"]
    if length(chars) > 100
        push!(ids, unk)
        return ids
    end
    pieces = Int32[]
    start = 1
    while start <= length(chars)
        stop = length(chars)
        found = Int32(0)
        while stop >= start
            piece = String(chars[start:stop])
            start > 1 && (piece = "##" * piece)
            id = get(t.vocab, piece, Int32(0))
            if id != 0
                found = id
                break
            end
            stop -= 1
        end
        if found == 0
            push!(ids, unk)
            return ids
        end
        push!(pieces, found)
        start = stop + 1
    end
    append!(ids, pieces)
end

"[CLS] w1 .. wn [SEP] as 1-based ids"
function encode_text(t::WordPieceTokenizer, text::AbstractString)
    ids = Int32[t.vocab["[CLS]"]]
    for w in _basic_tokens(t, text)
        _wordpiece!(ids, t, w)
    end
    push!(ids, t.vocab["[SEP]"])
    ids
end

"""
Token ids and mask of a batch: every sequence is cut to `max_tokens` (keeping its head) and padded with [PAD] to
`max_tokens` (`pad_to_max`, queries: trunc_or_pad, src/searching.jl:32-40) or to the longest sequence of the batch
(documents: trunc_and_pad, src/indexing.jl:37-46).  Returns (integer_ids::Matrix{Int32} (len, batch), bitmask::Matrix{Bool}).
"""
function _integer_ids_and_mask(t::WordPieceTokenizer, batch_text::AbstractVector{<:AbstractString}, max_tokens::Int,
        pad_to_max::Bool)
    seqs = [first(encode_text(t, s), max_tokens) for s in batch_text]
    width = pad_to_max ? max_tokens : maximum(length, seqs; init = 0)
    ids = fill(t.vocab["[PAD]"], width, length(seqs))
    mask = falses(width, length(seqs))
    for (j, s) in enumerate(seqs)
        ids[1:length(s), j] = s
        mask[1:length(s), j] .= true
    end
    ids, Matrix{Bool}(mask)
end

"the marker becomes row 2 (tokenizer_utils.jl:140-143)"
_add_marker_row(data::AbstractMatrix{T}, marker::T) where {T} =
    [data[1:min(1, size(data, 1)), :]; fill(marker, 1, size(data, 2)); data[2:end, :]]

"tensorize_docs (doc_tokenization.jl:143-156)"
function tensorize_docs(doc_token::String, t::WordPieceTokenizer, batch_text::AbstractVector{<:AbstractString},
        doc_maxlen::Int)
    ids, mask = _integer_ids_and_mask(t, batch_text, doc_maxlen - 1, false)
    _add_marker_row(ids, lookup(t, doc_token)), _add_marker_row(mask, true)
end

"tensorize_queries (query_tokenization.jl:174-197): pads are rewritten to [MASK] (query augmentation)"
function tensorize_queries(query_token::String, attend_to_mask_tokens::Bool, t::WordPieceTokenizer,
        batch_text::AbstractVector{<:AbstractString}, query_maxlen::Int)
    ids, mask = _integer_ids_and_mask(t, batch_text, query_maxlen - 1, true)
    ids = _add_marker_row(ids, lookup(t, query_token))
    mask = _add_marker_row(mask, true)
    mask_id = t.vocab["[MASK]"]
    ids[ids .== t.vocab["[PAD]"]] .= mask_id
    attend_to_mask_tokens && (mask[ids .== mask_id] .= true)
    ids, mask
end

const _PUNCTUATION = string.(collect("!\"#\$%&'()*+,-./:;<=>?@[\\]^_`{|}~"))

"ids whose embeddings are dropped from passages: punctuation + [PAD] (src/indexing.jl:30-34)"
doc_skiplist(t::WordPieceTokenizer, mask_punctuation::Bool) =
    Int[lookup(t, s) for s in (mask_punctuation ? [_PUNCTUATION; "[PAD]"] : ["[PAD]"])]
