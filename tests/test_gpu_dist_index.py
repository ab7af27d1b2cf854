"""Multi-GPU index build on its PRODUCT backend (distributed_index.HipBackend over libcolbert_hip): BASELINE config 5's
build half.  What it distributes: kmeans_gpu_onehot! (src/utils.jl:271-315) over sharded points, the chunk loop of
src/indexing/collection_indexer.jl:271-297 per shard.  Three angles, all against the oracle's sharded restatement:
  * the device-resident exchange (clb_kmeans_shard_pass_device / update_device) with two shards held by one process --
    the gathered buffer is what an all-gather of the two ranks' blocks delivers;
  * build_index_sharded(HipBackend) over nccl (= RCCL) with world size 1, through torch.distributed and through the
    library's own communicator (clb_comm_all_gather);
  * build_index_sharded(HipBackend) with two real rank processes sharing this box's one GPU (exchange over gloo, the
    host-buffer entry points)."""
import os
import tempfile

import numpy as np
import pytest

import colbert_jl_amd as clb  # noqa: F401
from colbert_jl_amd import codec, synthetic
from colbert_jl_amd.distributed_index import HipBackend, build_index_sharded, kmeans_sharded

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _problem(dim=128, K=40):
    embs, doclens = synthetic.make_embeddings(seed=81, n_docs=160, dim=dim, doclen_mean=24, doclen_std=5, n_components=16)
    rng = np.random.default_rng(82)
    cut = int(np.cumsum(doclens)[79])                           # passages 1..80 on rank 0, 81..160 on rank 1
    sample_cols = [np.sort(rng.choice(cut, 900, replace=False)),
                   np.sort(cut + rng.choice(embs.shape[1] - cut, 1100, replace=False))]
    held = np.asfortranarray(embs[:, rng.choice(embs.shape[1], 300, replace=False)])
    init = np.asfortranarray(embs[:, rng.choice(embs.shape[1], K, replace=False)])
    return embs, doclens, cut, sample_cols, held, init


def _oracle_sharded_kmeans(oracle, shards, init, iters):
    c = init.copy(order="F")
    done = 0
    for done in range(1, iters + 1):
        parts = [oracle.kmeans_shard_pass(x, c)[:2] for x in shards]
        c, _d, conv = oracle.kmeans_reduce_update(c, [p[0] for p in parts], [p[1] for p in parts])
        if conv:
            break
    return c, done


@pytest.mark.parametrize("dim,K", [(128, 40), (16, 9)])
def test_device_exchange_two_shards_one_process(oracle, dim, K):
    torch = pytest.importorskip("torch")
    embs, _dl, _cut, sample_cols, _held, init = _problem(dim, K)
    shards = [np.asfortranarray(embs[:, cols]) for cols in sample_cols]
    hs = [codec.KMeansShard(x, K, 1000) for x in shards]
    nb = hs[0].block_bytes
    assert nb == hs[1].block_bytes and nb >= 4 * dim * K + 8 * K
    dev = torch.device("cuda", 0)
    gathered = torch.empty(2 * nb, dtype=torch.uint8, device=dev)
    for h in hs:
        h.set_centroids(init)
    ref = init.copy(order="F")
    for _ in range(5):
        for r, h in enumerate(hs):
            h.pass_device(gathered[r * nb:(r + 1) * nb])        # rank r's block lands where the all-gather would put it
        res = [h.update_device(gathered, 2) for h in hs]
        parts = [oracle.kmeans_shard_pass(x, ref)[:2] for x in shards]
        ref, rd, rconv = oracle.kmeans_reduce_update(ref, [p[0] for p in parts], [p[1] for p in parts])
        for (d, conv), h in zip(res, hs):
            assert conv == rconv and bits(np.float32(d)) == bits(np.float32(rd))
            assert np.array_equal(bits(h.get_centroids()), bits(ref))
    # an empty shard (a rank without sample points) contributes zeros and does not read its host pointer
    empty = codec.KMeansShard(np.zeros((dim, 0), np.float32, order="F"), K, 1000)
    empty.set_centroids(ref)
    blk = torch.empty(nb, dtype=torch.uint8, device=dev)
    empty.pass_device(blk)
    torch.cuda.synchronize()
    assert int(blk.cpu().numpy().view(np.uint8).sum()) == 0
    for h in hs + [empty]:
        h.close()


@pytest.mark.parametrize("exchange", ["torch.distributed", "library communicator"])
def test_hip_backend_build_over_rccl_world1(oracle, exchange):
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    embs, doclens, _cut, sample_cols, held, init = _problem()
    sample = np.asfortranarray(embs[:, np.concatenate(sample_cols)])
    store = tempfile.NamedTemporaryFile(prefix="clb_pg_", delete=False); store.close(); os.unlink(store.name)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="file://" + store.name, rank=0, world_size=1, device_id=dev)
    try:
        if exchange == "library communicator":
            from colbert_jl_amd.distributed import LibraryComm
            comm = LibraryComm(0, 0, 1, LibraryComm.unique_id())
            c, it = kmeans_sharded(sample, init, HipBackend(0), max_iters=6, comm_device=dev, all_gather=comm.all_gather)
            comm.close()
            rc, rit = _oracle_sharded_kmeans(oracle, [sample], init, 6)
            assert it == rit and np.array_equal(bits(c), bits(rc))
            return
        out = build_index_sharded(embs, doclens, sample, held, init, HipBackend(0), nbits=2, kmeans_niters=6, comm_device=dev)
    finally:
        dist.destroy_process_group()
    rc, rit = _oracle_sharded_kmeans(oracle, [sample], init, 6)
    assert out["kmeans_iters"] == rit and np.array_equal(bits(out["centroids"]), bits(rc))
    # one shard == the single-device loop of the reference
    ref1, _, _ = oracle.kmeans(sample, init, max_iters=6)
    assert np.array_equal(bits(out["centroids"]), bits(ref1))
    rcut, rw, _ravg, _ = oracle.compute_avg_residuals(2, rc, held)
    assert np.array_equal(bits(out["bucket_cutoffs"]), bits(rcut)) and np.array_equal(bits(out["bucket_weights"]), bits(rw))
    codes, res = oracle.compress(rc, rcut, 128, 2, embs)
    assert np.array_equal(out["codes"], codes) and np.array_equal(out["residuals"], res)
    ivf, lens = oracle.build_ivf(codes, init.shape[1])
    assert np.array_equal(out["ivf"], ivf) and np.array_equal(out["ivf_lengths"], lens)


def _worker(rank, world, store, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
    embs, doclens, cut, sample_cols, held, init = _problem()
    lo, hi = (0, cut) if rank == 0 else (cut, embs.shape[1])
    dl = doclens[:80] if rank == 0 else doclens[80:]
    out = build_index_sharded(np.asfortranarray(embs[:, lo:hi]), dl, np.asfortranarray(embs[:, sample_cols[rank]]),
                              held, init, HipBackend(0), nbits=2, kmeans_niters=6)
    q.put((rank, out))
    dist.destroy_process_group()


def test_hip_backend_two_rank_processes_one_gpu(oracle):
    """Two rank processes, both on this box's GPU, exchange over gloo: every compute call of build_index_sharded goes
    through libcolbert_hip; the result must be the oracle's sharded restatement bit for bit, on both ranks."""
    import torch.multiprocessing as mp
    store = tempfile.NamedTemporaryFile(prefix="clb_pg_", delete=False); store.close(); os.unlink(store.name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, store.name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    embs, doclens, cut, sample_cols, held, init = _problem()
    shards = [np.asfortranarray(embs[:, cols]) for cols in sample_cols]
    rc, rit = _oracle_sharded_kmeans(oracle, shards, init, 6)
    for r in (0, 1):
        assert res[r]["kmeans_iters"] == rit and np.array_equal(bits(res[r]["centroids"]), bits(rc))
    rcut, rw, _ravg, _ = oracle.compute_avg_residuals(2, rc, held)
    for r, (lo, hi) in enumerate([(0, cut), (cut, embs.shape[1])]):
        assert np.array_equal(bits(res[r]["bucket_cutoffs"]), bits(rcut)) and np.array_equal(bits(res[r]["bucket_weights"]), bits(rw))
        codes, rr = oracle.compress(rc, rcut, 128, 2, np.asfortranarray(embs[:, lo:hi]))
        assert np.array_equal(res[r]["codes"], codes) and np.array_equal(res[r]["residuals"], rr)
        ivf, lens = oracle.build_ivf(codes, init.shape[1])
        assert np.array_equal(res[r]["ivf"], ivf) and np.array_equal(res[r]["ivf_lengths"], lens)


# ---- the same with every large array resident in HBM (index_device_sharded): the path bench.py --gpus N times ----------
def _device_worker(rank, world, store, q):
    import torch
    import torch.distributed as dist
    from colbert_jl_amd.distributed_index import index_device_sharded
    from colbert_jl_amd.indexer import index_to_host
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    n_local = 1200
    n_total = n_local * world
    src = synthetic.DeviceMixtureSource(seed=90 + rank, n_docs=n_local, device=dev, block=500, n_components=64)
    keep = {}
    index, rec = index_device_sharded(src, rank * n_local, n_total, HipBackend(0), nbits=2, kmeans_niters=3, seed=91,
                                      chunksize=700, keep=keep)
    host = index_to_host(index)
    out = {k: host[k] for k in ("centroids", "bucket_cutoffs", "bucket_weights", "codes", "residuals", "ivf", "ivf_lengths",
                                "kmeans_iters", "pid_offset")}
    out["sample"] = np.asfortranarray(keep["sample"].cpu().numpy().T)
    out["init"] = keep["init"]
    out["heldout"] = None if keep["heldout"] is None else np.asfortranarray(keep["heldout"].cpu().numpy().T)
    out["embs"] = np.asfortranarray(src.chunk(0, n_local).cpu().numpy().T)
    out["K"] = rec["K"]
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_index_device_sharded_rank_processes_one_gpu(oracle, world):
    """Two / four rank processes on this box's GPU, exchanges staged over gloo: sample rule, init all-gather, device k-means
    shards, statistics broadcast, resident codec, device IVF.  Against the oracle's sharded restatement, bit for bit."""
    import torch.multiprocessing as mp
    store = tempfile.NamedTemporaryFile(prefix="clb_pg_", delete=False); store.close(); os.unlink(store.name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_device_worker, args=(r, world, store.name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    K = res[0]["K"]
    assert all(res[r]["K"] == K for r in range(world)) and K == (4096 if world == 2 else K)
    per = -(-K // world)
    for r in range(world):
        assert np.array_equal(res[0]["init"], res[r]["init"])                   # identical on every rank
        hi = min(K, (r + 1) * per)
        assert np.array_equal(res[0]["init"][:, r * per:hi], res[r]["sample"][:, :hi - r * per])   # rank r's share, rank order
    rc, rit = _oracle_sharded_kmeans(oracle, [res[r]["sample"] for r in range(world)], res[0]["init"], 3)
    rcut, rw, _ravg, _ = oracle.compute_avg_residuals(2, rc, res[0]["heldout"])
    for r in range(world):
        assert res[r]["pid_offset"] == r * 1200
        assert res[r]["kmeans_iters"] == rit and np.array_equal(bits(res[r]["centroids"]), bits(rc))
        assert np.array_equal(bits(res[r]["bucket_cutoffs"]), bits(rcut)) and np.array_equal(bits(res[r]["bucket_weights"]), bits(rw))
        codes, rr = oracle.compress(rc, rcut, 128, 2, res[r]["embs"])
        assert np.array_equal(res[r]["codes"], codes) and np.array_equal(res[r]["residuals"], rr)
        ivf, lens = oracle.build_ivf(codes, K)
        assert np.array_equal(res[r]["ivf"], ivf) and np.array_equal(res[r]["ivf_lengths"], lens)


def test_index_device_sharded_with_the_encoder_as_source(oracle, tmp_path):
    """index_device_sharded over RCCL (world 1) with indexer.EncoderSource -- the BERT encoder's packed batches -- as the
    embedding source: only the sample is encoded for training, the chunks are encoded once; k-means, statistics, codes,
    residuals and IVF equal the oracle's on the embeddings the source yields."""
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    import torch.distributed as dist
    from colbert_jl_amd.distributed_index import index_device_sharded
    from colbert_jl_amd.encoder import pack_weights
    from colbert_jl_amd.indexer import EncoderSource, index_to_host
    from colbert_jl_amd.tokenization import WordPieceTokenizer
    words = ["hello", "world", "this", "is", "a", "test", "of", "the", "tokenizer", "longer", "passage", "with", "many", "words", "query", "colbert"]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(20)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", ".", ",", "!"] + words
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab) + "\n")
    tok = WordPieceTokenizer(str(vf))
    torch.manual_seed(51)
    cfg = transformers.BertConfig(vocab_size=len(vocab), hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                                  max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bert = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    linear = torch.nn.Linear(128, 128, bias=True).eval()
    with torch.no_grad():
        for p in bert.parameters():
            p.mul_(4.0)
    state = {k: v.detach().float().numpy() for k, v in bert.state_dict().items()}
    state["linear.weight"] = linear.weight.detach().numpy(); state["linear.bias"] = linear.bias.detach().numpy()
    rng = np.random.default_rng(52)
    collection = [" ".join(rng.choice(words, size=rng.integers(2, 25))) + "." for _ in range(300)]
    config = clb.ColBERTConfig(index_path="unused", doc_maxlen=40, query_maxlen=12, index_bsize=16, nbits=2)
    enc = clb.BertEncoder(pack_weights(state, cfg.to_dict(), 128), cfg.to_dict(), dim=128, tokenizer=tok, config=config)
    src = EncoderSource(enc, collection, 0)
    store = tempfile.NamedTemporaryFile(prefix="clb_pg_", delete=False); store.close(); os.unlink(store.name)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="file://" + store.name, rank=0, world_size=1, device_id=dev)
    try:
        keep = {}
        index, rec = index_device_sharded(src, 0, len(collection), HipBackend(0), nbits=2, kmeans_niters=3, seed=53, chunksize=120, keep=keep)
    finally:
        dist.destroy_process_group()
    assert src.packed is True
    host = index_to_host(index)
    sample = np.asfortranarray(keep["sample"].cpu().numpy().T)
    rc, rit = _oracle_sharded_kmeans(oracle, [sample], keep["init"], 3)
    assert host["kmeans_iters"] == rit and np.array_equal(bits(host["centroids"]), bits(rc))
    embs = np.asfortranarray(torch.cat([src.chunk(s, min(s + 120, 300)) for s in range(0, 300, 120)]).cpu().numpy().T)
    assert embs.shape[1] == int(src.doclens.sum()) == host["codes"].size
    codes, res = oracle.compress(rc, host["bucket_cutoffs"], 128, 2, embs)
    assert np.array_equal(host["codes"], codes) and np.array_equal(host["residuals"], res)
    ivf, lens = oracle.build_ivf(codes, rec["K"])
    assert np.array_equal(host["ivf"], ivf) and np.array_equal(host["ivf_lengths"], lens)
    enc.close()
