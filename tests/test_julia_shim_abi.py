"""The Julia shim cannot be executed here (no `julia` in the image), so its boundary is checked mechanically instead:
every `ccall` in julia/ColBERT/src/*.jl must name a function include/colbert_hip.h declares, with the same number of
arguments, the same width class for every argument (32-bit int / 64-bit int / float / double / pointer) and the same
return type.  The reference has no FFI of its own; what the ccalls replace is listed per prototype in the header."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "colbert_hip.h")
JULIA_DIR = os.path.join(ROOT, "julia", "ColBERT", "src")


def c_class(t: str) -> str:
    t = t.strip()
    if "*" in t:
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    return {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "float": "f32",
            "double": "f64", "void": "void", "size_t": "i64"}[t]


def header_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(clb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name: the type is everything up to the last identifier (pointers keep their '*')
                mm = re.match(r"(.*?)(\b[A-Za-z_]\w*)?$", a, flags=re.S)
                typ = mm.group(1).strip() if mm.group(2) and mm.group(1).strip() else a
                arglist.append(c_class(typ))
        protos[name] = (c_class(ret), arglist)
    return protos


JL = {"Cint": "i32", "Int32": "i32", "UInt32": "i32", "Int64": "i64", "UInt64": "i64", "Int": "i64", "Float32": "f32",
      "Float64": "f64", "Cdouble": "f64", "Cfloat": "f32", "Cstring": "ptr", "Cvoid": "void", "Nothing": "void"}


def jl_class(t: str) -> str:
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return JL[t]


def split_top(s: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def julia_ccalls():
    calls = []
    for fn in sorted(os.listdir(JULIA_DIR)):
        if not fn.endswith(".jl"):
            continue
        text = open(os.path.join(JULIA_DIR, fn)).read()
        for m in re.finditer(r"ccall\(\(:(clb_[a-z0-9_]+),\s*libcolbert\)\s*,", text):
            i = m.end()
            # the rest of the call up to its closing parenthesis
            depth, j = 1, i
            while depth:
                depth += {"(": 1, ")": -1}.get(text[j], 0)
                j += 1
            parts = split_top(text[i:j - 1])
            ret, argt = parts[0], parts[1]
            assert argt.startswith("(") and argt.endswith(")"), (fn, m.group(1), argt)
            inner = argt[1:-1].strip()
            types = [x for x in split_top(inner) if x] if inner else []
            calls.append((fn, m.group(1), jl_class(ret), [jl_class(t) for t in types], len(parts) - 2))
    return calls


def test_every_ccall_matches_a_header_prototype():
    protos = header_prototypes()
    assert len(protos) >= 60, len(protos)
    calls = julia_ccalls()
    assert len(calls) >= 20, len(calls)
    for fn, name, ret, types, n_values in calls:
        assert name in protos, f"{fn}: ccall of {name}, which include/colbert_hip.h does not declare"
        pret, pargs = protos[name]
        assert ret == pret, f"{fn}: {name} returns {pret} in the header, {ret} in the ccall"
        assert len(types) == len(pargs), f"{fn}: {name} takes {len(pargs)} arguments in the header, the ccall declares {len(types)}"
        assert n_values == len(types), f"{fn}: {name}: {len(types)} argument types but {n_values} values"
        for i, (a, b) in enumerate(zip(types, pargs)):
            assert a == b, f"{fn}: {name} argument {i + 1}: header {b}, ccall {a}"


def test_the_shim_binds_the_search_and_build_entry_points():
    bound = {c[1] for c in julia_ccalls()}
    for name in ("clb_searcher_create", "clb_searcher_destroy", "clb_search", "clb_compress", "clb_decompress", "clb_kmeans",
                 "clb_compute_avg_residuals", "clb_build_ivf", "clb_encoder_create", "clb_encode", "clb_encode_docs",
                 "clb_encode_queries", "clb_comm_create", "clb_comm_all_gather", "clb_searcher_sync_bound_consts",
                 "clb_last_error",
                 # round 5: the device-resident route of index() (julia/ColBERT/src/indexing.jl: _index_device)
                 "clb_device_malloc", "clb_device_free", "clb_device_upload", "clb_device_download", "clb_device_memory",
                 "clb_gather_rows_device", "clb_encode_docs_packed_device", "clb_kmeans_shard_create_device",
                 "clb_kmeans_shard_pass_device", "clb_kmeans_shard_update_device", "clb_kmeans_shard_get_centroids",
                 "clb_codec_create", "clb_codec_compress_device", "clb_build_ivf_device", "clb_encoder_check_last_ids"):
        assert name in bound, name


def test_the_julia_device_route_uses_only_bound_helpers():
    """_index_device (indexing.jl) may only call functions the shim defines: a helper renamed in capi.jl would otherwise be
    found at run time only -- and Julia never runs here."""
    src = {fn: open(os.path.join(JULIA_DIR, fn)).read() for fn in os.listdir(JULIA_DIR) if fn.endswith(".jl")}
    text = src["indexing.jl"]
    body = text[text.index("function _index_device"):]
    defined = set()
    for t in src.values():
        defined |= set(re.findall(r"^(?:function\s+)?([A-Za-z_][\w!]*)\(", t, flags=re.M))
        defined |= set(re.findall(r"^(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", t, flags=re.M))
    used = set(re.findall(r"\b(_[a-z][\w!]*|device_[a-z_!]+|gather_columns_device!|save_[a-z_]+|DeviceBuffer)\(", body))
    assert len(used) >= 12, used
    for name in used - {"save_object"}:                # JLD2.save_object: the package's, not the shim's
        assert name in defined, f"_index_device calls {name}, which no file of the shim defines"


def test_header_parser_sees_every_declared_symbol():
    import colbert_jl_amd as clb
    assert set(header_prototypes()) == set(clb.declared_symbols())
