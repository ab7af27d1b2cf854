"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, refuses to compute without a GPU (no CPU fallback), and the host-side mirrors of the
reference's planning helpers agree with the oracle.  No compute calls here."""
import os

import numpy as np
import pytest

import colbert_jl_amd as clb


def test_library_exports_every_declared_symbol():
    l = clb.lib()
    syms = clb.declared_symbols()
    assert len(syms) >= 20
    assert [s for s in syms if not hasattr(l, s)] == []
    assert b"gfx950" in l.clb_version()


def test_no_cpu_fallback():
    if clb.lib().clb_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(clb.HipError):
        clb.codec._normalize_array(np.ones((8, 2), np.float32))
    idx = clb.synthetic.make_index(0, 20, K=8)
    with pytest.raises(clb.HipError):
        clb.Searcher(index=idx)
    # the library's RCCL communicator: argument checks first, then the device (no GPU: HipError, nothing opened)
    from colbert_jl_amd.distributed import LibraryComm
    assert int(clb.lib().clb_comm_unique_id_bytes()) == 128
    with pytest.raises(clb.ArgumentError):
        LibraryComm(0, 3, 2, bytes(128))          # rank outside 0..n_ranks-1
    with pytest.raises(clb.ArgumentError):
        LibraryComm(0, 0, 1, bytes(16))           # not a unique id
    with pytest.raises(clb.HipError):
        LibraryComm(0, 0, 1, bytes(128))


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.abspath(clb.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("from oracle", "import oracle", "libcolbert_oracle", '#include "colbert_oracle',
                               "#include <colbert_oracle", "orc_"):
                    assert needle not in text, (f, needle)


def test_argument_contracts_without_gpu():
    """Checks that run before any device work mirror the reference's exceptions."""
    idx = clb.synthetic.make_index(0, 20, K=8)
    bad = dict(idx); bad["ivf"] = idx["ivf"][:-1]; bad["ivf_lengths"] = idx["ivf_lengths"].copy()
    bad["ivf_lengths"][0] += 0
    with pytest.raises((clb.DimensionMismatch, clb.ColBERTError)):
        # length(ivf) must equal sum(ivf_lengths)  (ranking.jl:11-12)
        bad2 = dict(idx); bad2["ivf_lengths"] = idx["ivf_lengths"].copy(); bad2["ivf_lengths"][0] += 1
        clb.Searcher(index=bad2)
    with pytest.raises(clb.DomainError):                      # residual.jl:763-765
        clb.codec.decompress(128, 2, idx["centroids"], idx["bucket_weights"], idx["codes"][:-1], idx["residuals"])
    with pytest.raises(clb.DomainError):                      # residual.jl:766-768
        c = idx["codes"].copy(); c[0] = 9
        clb.codec.decompress(128, 2, idx["centroids"], idx["bucket_weights"], c, idx["residuals"])
    with pytest.raises(clb.DomainError):                      # residual.jl:705-706
        clb.codec.decompress(128, 2, idx["centroids"], idx["bucket_weights"][:3], idx["codes"], idx["residuals"])
    with pytest.raises(clb.DimensionMismatch):                # residual.jl:72-74
        clb.codec.compress_into_codes(idx["centroids"], np.zeros((128, 5), np.float32), n_codes=4)
    with pytest.raises(clb.DimensionMismatch):                # ranking.jl:71-74
        clb.codec.maxsim(np.eye(2, dtype=np.float32), np.array([[0.8, 0.3], [0.2, 0.7]], np.float32)[:, :1], [1, 2], [1, 2])
    with pytest.raises(clb.DomainError):                      # residual.jl:520
        clb.codec.compress(np.zeros((7, 2), np.float32), np.zeros(1, np.float32), 7, 1, np.zeros((7, 3), np.float32))


def test_round3_entry_points_check_their_arguments_first():
    """The entry points added in round 3 refuse null handles / bad values before touching a device (error 4 =
    ArgumentError), like every other export."""
    import ctypes as C
    l = clb.lib()
    null = C.c_void_p()
    assert l.clb_searcher_set_pass1_gather(null, 0) == 4
    assert l.clb_searcher_get_pass1_gather(null, None) == -1
    assert l.clb_debug_sort(0, 16, None, None, C.c_int64(4), 32, None, None) == 4          # bad key width / null arrays
    assert l.clb_debug_exclusive_scan(0, None, C.c_int64(-1), None) == 4
    assert l.clb_measure_copy_rate(0, C.c_int64(16), 1, None) == 4
    assert l.clb_measure_read_rate(0, C.c_int64(16), 1, None) == 4
    assert l.clb_searcher_set_score_rows(null, 1) == 4
    assert l.clb_searcher_get_score_rows(null) == -1
    assert l.clb_searcher_set_centroid_products(null, 1) == 4
    assert l.clb_searcher_get_centroid_products(null, None) == -1
    assert l.clb_searcher_sync_bound_consts(null, null) == 4
    assert l.clb_kmeans_shard_block_bytes(null) == 0
    assert l.clb_kmeans_shard_set_centroids(null, None) == 4
    assert l.clb_kmeans_shard_get_centroids(null, None) == 4
    assert l.clb_kmeans_shard_pass_device(null, None, None) == 4
    assert l.clb_kmeans_shard_update_device(null, None, C.c_int64(1), C.c_float(1e-4), None, None, None) == 4
    assert l.clb_encoder_check_last_ids(null) == 4
    assert l.clb_encoder_profile_enable(null, 1) == 4
    assert l.clb_encoder_profile_read(null, None, None, None, 0) == -1
    assert b"null" in l.clb_last_error()


def test_round4_entry_points_check_their_arguments_first():
    """The device-resident index-build entry points of round 4: argument contracts before any device work (the codec
    handle mirrors compress's DomainErrors, residual.jl:523-525), no CPU fallback behind them."""
    import ctypes as C
    l = clb.lib()
    null = C.c_void_p()
    out = C.c_void_p()
    cent = np.zeros((128, 4), np.float32, order="F")
    cut = np.zeros(3, np.float32)
    i64 = C.c_int64
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert l.clb_codec_create(0, i64(128), 2, i64(4), p(cent), p(cut), i64(3), None) == 4                 # out is null
    assert l.clb_codec_create(0, i64(7), 2, i64(4), p(cent), p(cut), i64(3), C.byref(out)) == 2           # dims % 8
    assert l.clb_codec_create(0, i64(128), 2, i64(4), p(cent), p(cut), i64(2), C.byref(out)) == 2         # 2^nbits - 1 cutoffs
    assert l.clb_codec_create(0, i64(128), 2, i64(0), p(cent), p(cut), i64(3), C.byref(out)) == 4         # no centroids
    assert l.clb_codec_compress_device(null, None, i64(5), None, None, None) == 4
    assert l.clb_codec_destroy(null) == 0
    assert l.clb_build_ivf_device(0, None, i64(-1), i64(4), None, None, None) == 4
    assert l.clb_build_ivf_device(0, None, i64(5), i64(4), None, None, None) == 4                          # null arrays
    assert l.clb_kmeans_shard_get_assignments(null, None) == 4
    assert l.clb_encode_docs_device(null, None, None, i64(8), i64(2), None, i64(0), None, None, None, None) == 4
    assert l.clb_encode_docs_packed_device(null, None, None, None, None, i64(2), i64(8), i64(12), None, i64(0), None, None, None, None) == 4
    assert l.clb_kmeans_shard_create_device(0, None, i64(128), i64(5), i64(0), i64(1000), C.byref(out)) == 1   # K < 1
    if l.clb_device_count() == 0:      # valid arguments reach the device check: HIP error, nothing computed on the host
        assert l.clb_codec_create(0, i64(128), 2, i64(4), p(cent), p(cut), i64(3), C.byref(out)) == 10
        assert l.clb_kmeans_shard_create_device(0, p(cent), i64(128), i64(4), i64(2), i64(1000), C.byref(out)) == 10
        idx = clb.synthetic.make_index(0, 20, K=8)
        dl = np.ascontiguousarray(idx["doclens"], np.int64); il = np.ascontiguousarray(idx["ivf_lengths"], np.int64)
        w = np.ascontiguousarray(idx["bucket_weights"], np.float32)
        assert l.clb_searcher_create_device(0, i64(128), 2, i64(8), None, p(w), i64(dl.size), p(dl), i64(int(dl.sum())), None, None,
                                            None, p(il), i64(0), C.byref(out)) == 10


def test_round5_entry_points_check_their_arguments_first():
    """Round 5: plain device arrays + the row gather for hosts without their own (the Julia shim's device-resident index()),
    and the encoder's sticky error flag: argument contracts first, then the device check -- no CPU fallback."""
    import ctypes as C
    l = clb.lib()
    i64 = C.c_int64
    out = C.c_void_p()
    f, t = i64(0), i64(0)
    assert l.clb_device_malloc(0, i64(16), None) == 4
    assert l.clb_device_malloc(0, i64(-1), C.byref(out)) == 4
    assert l.clb_device_free(0, None) == 0                                        # freeing nothing is fine
    assert l.clb_device_upload(0, None, None, i64(8)) == 4
    assert l.clb_device_download(0, None, None, i64(8)) == 4
    assert l.clb_device_memory(0, None, None) == 4
    assert l.clb_gather_rows_device(0, None, i64(4), i64(6), None, i64(2), None, None) == 4      # row_bytes % 4
    assert l.clb_gather_rows_device(0, None, i64(4), i64(512), None, i64(2), None, None) == 4    # null arrays
    assert l.clb_gather_rows_device(0, None, i64(4), i64(512), None, i64(0), None, None) == 0    # nothing to do
    assert l.clb_encoder_error_flag_device(None, C.byref(out)) == 4
    if l.clb_device_count() == 0:
        assert l.clb_device_malloc(0, i64(16), C.byref(out)) == 10
        assert l.clb_device_memory(0, C.byref(f), C.byref(t)) == 10
        assert l.clb_device_synchronize(0) == 10
        x = np.zeros(4, np.float32)
        assert l.clb_device_upload(0, C.c_void_p(16), x.ctypes.data_as(C.c_void_p), i64(16)) == 10


def test_plane_layout_and_xcd_tile_mapping():
    """Host restatement of two pieces of the encoder's plane GEMM (csrc/encoder_kernels.hpp): the K-blocked plane index is
    a bijection onto [0, rows * K), and the XCD-aware work-group -> tile map covers every (K slice, n tile, m tile) exactly
    once with every XCD owning a contiguous, equally long range of the (slice, n tile, m tile) order."""
    def plane_index(row, k, rows):
        return (k >> 5) * rows * 32 + row * 32 + (k & 31)
    for rows, K in ((5, 64), (64, 96), (33, 768)):
        seen = {plane_index(r, k, rows) for r in range(rows) for k in range(K)}
        assert seen == set(range(rows * K))
        assert all(plane_index(r, 32 * b + 1, rows) - plane_index(r, 32 * b, rows) == 1 for r in range(rows) for b in range(K // 32))

    def grid(M, N, bm, bn, ks):
        T = -(-M // bm) * -(-N // bn) * ks
        return 8 * -(-T // 8)

    def tile_of(b, M, N, bm, bn, ks):
        TM, TN = -(-M // bm), -(-N // bn)
        T = TM * TN * ks
        x, j = b & 7, b >> 3
        t = x * T // 8 + j
        if t >= (x + 1) * T // 8:
            return None
        c = t // TM
        return (c // TN, c % TN, t % TM, x)
    for M, N, bm, bn, ks in ((1024, 2304, 64, 64, 1), (1024, 768, 64, 64, 4), (32, 3072, 64, 64, 8), (19200, 768, 128, 128, 1),
                             (19200, 2304, 128, 128, 1), (100, 130, 64, 64, 2)):
        tiles = [tile_of(b, M, N, bm, bn, ks) for b in range(grid(M, N, bm, bn, ks))]
        live = [t for t in tiles if t is not None]
        TM, TN = -(-M // bm), -(-N // bn)
        assert sorted((z, n, m) for z, n, m, _ in live) == sorted((z, n, m) for z in range(ks) for n in range(TN) for m in range(TM))
        per_xcd = [sum(1 for t in live if t[3] == x) for x in range(8)]
        assert max(per_xcd) - min(per_xcd) <= 1              # every XCD gets its eighth of the tiles
        for x in range(8):                                   # ... one contiguous range of the (c, m) order: ~TN / 8 weight tiles
            ts = sorted((z * TN + n) * TM + m for z, n, m, xx in live if xx == x)
            assert ts == list(range(ts[0], ts[0] + len(ts))) if ts else True


def test_host_planning_helpers_match_oracle(oracle):
    for n in (1, 10, 1000, 141431, 10 ** 6):
        assert clb.codec.num_sampled_pids(n) == oracle.num_sampled_pids(n)
    for n in (1, 3, 19, 1000, 10 ** 6, 10 ** 7):
        assert clb.codec.heldout_size(n) == oracle.heldout_size(n)
    for args in ((10, 178.28572, 10 ** 9, 25000, 1), (141431, 62.15259, 10 ** 9, 25000, 1), (100, 50.0, 37, None, 2),
                 (10 ** 6, 80.0, 10 ** 9, None, 8)):
        assert clb.codec.setup(*args) == oracle.setup(*args)
    assert clb.codec.collect_embedding_id_offset([3, 5, 2])[0] == 10
    assert np.array_equal(clb.codec.collect_embedding_id_offset([3, 5, 2])[1], [1, 4, 9])
    assert clb.codec.collect_embedding_id_offset([])[0] == 0
    assert clb.synthetic.num_partitions_for(10 ** 5, 80.0) == 32768
    assert clb.synthetic.num_partitions_for(10 ** 6, 80.0) == 131072


def test_config_defaults_and_roundtrip(tmp_path):
    cfg = clb.ColBERTConfig()
    assert (cfg.dim, cfg.doc_maxlen, cfg.query_maxlen, cfg.index_bsize, cfg.chunksize, cfg.nbits, cfg.kmeans_niters,
            cfg.nprobe, cfg.ncandidates) == (128, 300, 32, 64, 25000, 2, 20, 2, 8192)
    assert (cfg.query_token_id, cfg.doc_token_id, cfg.query_token, cfg.doc_token) == ("[unused0]", "[unused1]", "[Q]", "[D]")
    assert len(cfg.__dataclass_fields__) == 22                 # src/infra/config.jl:54-90
    cfg2 = clb.ColBERTConfig(index_path=str(tmp_path / "idx"), nbits=4, chunksize=None)
    cfg2.save()
    assert clb.ColBERTConfig.load(cfg2.index_path) == cfg2


def test_synthetic_index_is_consistent(oracle):
    idx = clb.synthetic.make_index(3, 500, K=128)
    assert idx["codes"].min() >= 1 and idx["codes"].max() <= 128
    ivf, lens = oracle.build_ivf(idx["codes"], 128)
    assert np.array_equal(ivf, idx["ivf"]) and np.array_equal(lens, idx["ivf_lengths"])
    assert idx["residuals"].shape == (32, idx["doclens"].sum()) and idx["residuals"].flags.f_contiguous
    Q = clb.synthetic.make_queries(idx, 4, 3)
    assert Q.shape == (128, 32, 3) and np.allclose(np.linalg.norm(Q, axis=0), 1, atol=1e-5)


def test_gelu_erf_polynomial_is_accurate_to_one_ulp_of_one():
    """The encoder's GELU evaluates erf(t) = 1 - exp(q(t)), q a degree-9 fit of log erfc on [0, 4] (tools/fit_gelu_erf.py):
    the coefficients in csrc/encoder_kernels.hpp, evaluated here in fp32 the way the kernel does (Horner with fused
    multiply-adds, exp2 of q log2 e), stay within 1.5e-7 of erf everywhere -- and they are what the fit script produces."""
    import math
    import re
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ROOT, "colbert.jl_amd", "csrc", "encoder_kernels.hpp")).read()
    body = re.search(r"kErfQ\[10\]\s*=\s*\{([^}]*)\}", src).group(1)
    co = [np.float32(v.strip().rstrip("f")) for v in body.split(",")]
    assert len(co) == 10
    x = np.linspace(-6.0, 6.0, 240001).astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(4.0))
    q = np.full_like(t, co[9])
    for a in co[8::-1]:
        q = (q.astype(np.float64) * t + np.float64(a)).astype(np.float32)            # one rounding per fma
    e = (np.float32(1) - np.exp2((q * np.float32(1.4426950408889634)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    got = np.copysign(e, x).astype(np.float64)
    want = np.array([math.erf(float(v)) for v in x])
    assert np.abs(got - want).max() < 1.5e-7
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fit_gelu_erf
    assert np.allclose(np.array(fit_gelu_erf.fit(), dtype=np.float32), np.array(co), rtol=2e-5, atol=1e-9)


def test_key_blocked_v_layout_matches_the_accumulator_order():
    """Host restatement of vt_index (csrc/encoder_kernels.hpp): inside a tile of 32 keys, key t sits at the position at
    which attention_f16_kernel's lanes expect it -- lane half h, MFMA step u, slot j holds key 16u + 8(j>>2) + 4h + (j&3)
    (the rows of the 32 x 32 score accumulator a lane owns) and reads position 16h + 8u + j; the map is a bijection."""
    def pos_of(t):
        u, w = (t >> 4) & 1, t & 15
        return 16 * ((w >> 2) & 1) + 8 * u + 4 * (w >> 3) + (w & 3)
    assert sorted(pos_of(t) for t in range(32)) == list(range(32))
    for h in range(2):
        for u in range(2):
            for j in range(8):
                key = 16 * u + 8 * (j >> 2) + 4 * h + (j & 3)
                assert pos_of(key) == 16 * h + 8 * u + j
                # ... and that key is the accumulator row of register r = 8u + j in lane half h
                r = 8 * u + j
                assert key == (r & 3) + 8 * (r >> 2) + 4 * h


def test_hand_issued_lds_dma_is_the_only_user_of_m0():
    """The inline-asm LDS-DMA blocks (global_load_lds_dwordx4 behind `s_mov_b32 m0`) write M0 without declaring it (hipcc
    rejects M0 as a clobber): tools/check_m0.py disassembles the shipped library and fails if anything else in those
    kernels reads or writes M0, or if a DMA is not fed by its own block's M0 write -- a later edit that pulls in a builtin
    using M0 (indirect register indexing, the compiler's own global_load_lds lowering) cannot miscompile silently."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_m0", os.path.join(root, "tools", "check_m0.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    problems, kernels, dmas = mod.check(clb._lib.LIB_PATH, verbose=False)
    assert not problems, problems[:5]
    assert kernels >= 3 and dmas >= 8          # pass 1's GL = 1 gather and the plane GEMMs are in the library


def test_index_file_writer_thread_keeps_order_and_hands_back_errors(tmp_path):
    """indexer._BackgroundWriter (round 5: the index directory is written while the device goes on): jobs run in submission
    order on one thread, finish() drains them, the first exception surfaces in the caller's thread (from a later submit or
    from finish) and nothing after it runs."""
    import threading
    import time
    from colbert_jl_amd.indexer import _BackgroundWriter
    w = _BackgroundWriter(depth=2)
    seen, main = [], threading.get_ident()

    def job(i):
        def run():
            time.sleep(0.01)
            seen.append((i, threading.get_ident() != main))
        return run
    for i in range(6):                       # more jobs than the queue holds: submit blocks instead of piling tensors up
        w.submit(job(i))
    w.finish()
    assert seen == [(i, True) for i in range(6)] and w.busy_s > 0.05
    w.finish()                               # idempotent

    w = _BackgroundWriter()
    ran = []
    w.submit(lambda: ran.append(1))
    w.submit(lambda: (_ for _ in ()).throw(OSError("disk full")))
    with pytest.raises(OSError, match="disk full"):
        for _ in range(50):                  # queued behind the failure: skipped (the directory is already known to be incomplete);
            w.submit(lambda: ran.append(2))  # once the worker has met it, submit itself raises
            time.sleep(0.005)
        w.finish()
    assert ran == [1]
    with pytest.raises(RuntimeError):
        w.submit(lambda: None)


def test_chunk_rows_from_cached_sample_and_new_passages():
    """indexer._chunk_source_rows (round 5: a chunk of index() is cut out of [the sample's embeddings | its newly encoded
    passages] instead of being encoded in full): against a plain loop, with empty passages, all-known and none-known chunks."""
    from colbert_jl_amd.indexer import _chunk_source_rows
    rng = np.random.default_rng(5)
    for trial in range(30):
        n = int(rng.integers(1, 40))
        dl = rng.integers(0, 6, size=n)
        known = rng.random(n) < (0.0 if trial == 0 else 1.0 if trial == 1 else 0.4)
        n_cached = int(rng.integers(0, 500))
        first = np.sort(rng.choice(max(n_cached, 1) + 50, size=int(known.sum()), replace=False)) if known.any() else np.zeros(0, np.int64)
        want, tail = [], n_cached
        ki = 0
        for j in range(n):
            if known[j]:
                want.extend(range(int(first[ki]), int(first[ki]) + int(dl[j]))); ki += 1
            else:
                want.extend(range(tail, tail + int(dl[j]))); tail += int(dl[j])
        got = _chunk_source_rows(known, dl, first, n_cached)
        assert got.dtype == np.int64 and np.array_equal(got, np.asarray(want, dtype=np.int64)), trial
