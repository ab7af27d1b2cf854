"""The JLD2 (HDF5-subset) files of the index directory: writer/reader round trips, the checksums HDF5 requires,
and the structure a JLD2.jl / HDF5 reader walks (superblock -> root group link -> dataset messages)."""
import os
import struct

import numpy as np
import pytest

import colbert_jl_amd as clb  # noqa: F401
from colbert_jl_amd import jld2, storage


def test_lookup3_known_answers():
    # the self-test values of Bob Jenkins' lookup3.c (driver5)
    assert jld2.lookup3(b"") == 0xDEADBEEF
    assert jld2.lookup3(b"Four score and seven years ago") == 0x17770551
    assert jld2.lookup3(b"Four score and seven years ago", 1) == 0xCD628161


@pytest.mark.parametrize("arr", [
    np.float32(0.0123),                                                    # avg_residual :: Float32
    np.arange(3, dtype=np.float32) / 7,                                    # bucket_cutoffs :: Vector{Float32}
    np.asfortranarray(np.random.default_rng(0).standard_normal((128, 70)).astype(np.float32)),   # centroids
    np.random.default_rng(1).integers(1, 2 ** 32, size=5000, dtype=np.uint32),                   # codes (contiguous)
    np.asfortranarray(np.random.default_rng(2).integers(0, 256, size=(32, 999), dtype=np.uint8)),  # residuals
    np.random.default_rng(3).integers(0, 300, size=1500).astype(np.int64),                       # doclens / ivf
    np.zeros(0, dtype=np.int64),                                           # empty vector
    np.asfortranarray(np.zeros((32, 0), dtype=np.uint8)),                  # empty matrix
    np.float64(2.5), np.arange(10, dtype=np.int32), np.arange(6, dtype=np.uint16).reshape(2, 3),
])
def test_round_trip(tmp_path, arr):
    path = str(tmp_path / "x.jld2")
    jld2.save_object(path, arr)
    back = jld2.load_object(path)
    a = np.asarray(arr)
    assert np.asarray(back).dtype == a.dtype and np.asarray(back).shape == a.shape
    assert np.array_equal(np.asarray(back), a)
    if a.ndim == 2:
        assert back.flags.f_contiguous                                      # Julia memory order


def test_file_structure(tmp_path):
    """Walk the file the way an HDF5 reader does, independently of jld2._File."""
    path = str(tmp_path / "c.jld2")
    C = np.asfortranarray(np.arange(128 * 100, dtype=np.float32).reshape(128, 100, order="F"))
    jld2.save_object(path, C)
    buf = open(path, "rb").read()
    assert buf.startswith(b"HDF5-based Julia Data Format, version 0.1.1\x00")
    sb = 512
    assert buf[sb:sb + 8] == b"\x89HDF\r\n\x1a\n" and buf[sb + 8] == 2 and buf[sb + 9:sb + 11] == b"\x08\x08"
    base, ext, eof, root = struct.unpack_from("<QQQQ", buf, sb + 12)
    assert base == 512 and ext == 2 ** 64 - 1 and base + eof == len(buf)
    assert struct.unpack_from("<I", buf, sb + 44)[0] == jld2.lookup3(buf[sb:sb + 44])
    r = base + root
    assert buf[r:r + 4] == b"OHDR" and buf[r + 4] == 2
    assert b"single_stored_object" in buf[r:r + 128]
    # the matrix is stored column-major with reversed dims (100, 128) in the dataspace message
    ds = base + 48
    assert buf[ds:ds + 4] == b"OHDR"
    k = buf.index(struct.pack("<BBBB", 2, 2, 0, 1), ds)
    assert struct.unpack_from("<QQ", buf, k + 4) == (100, 128)
    raw = C.tobytes(order="F")
    assert raw in buf and (buf.index(raw) - base) % 8 == 0
    # a flipped bit in the object header is detected
    bad = bytearray(buf); bad[ds + 12] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(jld2.JLD2Error):
        jld2.load_object(path)


def test_reader_accepts_v1_headers_and_other_link_names(tmp_path):
    """A minimal file as an HDF5 1.8 library lays it out at base address 0: version-2 dataspace replaced by version
    1, datatype version 1, layout version 3 -- the reader is not tied to what the writer emits."""
    a = np.arange(12, dtype=np.int64)
    dataspace = struct.pack("<BBBBI", 1, 1, 0, 0, 0) + struct.pack("<Q", 12)
    datatype = struct.pack("<BBBBIHH", (1 << 4) | 0, 0x08, 0, 0, 8, 0, 64)
    hdr = lambda msgs: jld2._object_header(msgs)          # noqa: E731
    ds_addr = 48
    probe = hdr([jld2._header_message(1, dataspace), jld2._header_message(3, datatype),
                 jld2._header_message(8, struct.pack("<BBQQ", 3, 1, 0, 96))])
    data_addr = (ds_addr + len(probe) + 7) // 8 * 8
    dset = hdr([jld2._header_message(1, dataspace), jld2._header_message(3, datatype),
                jld2._header_message(8, struct.pack("<BBQQ", 3, 1, data_addr, 96))])
    body = dset + b"\0" * (data_addr - ds_addr - len(dset)) + a.tobytes()
    root_addr = (ds_addr + len(body) + 7) // 8 * 8
    name = b"data"
    link = struct.pack("<BBBBB", 1, 0x18, 0, 0, len(name)) + name + struct.pack("<Q", ds_addr)   # type + charset fields
    root = hdr([jld2._header_message(2, struct.pack("<BBQQ", 0, 0, jld2.UNDEFINED, jld2.UNDEFINED)),
                jld2._header_message(6, link)])
    sb = jld2.H5_SIGNATURE + struct.pack("<BBBB", 2, 8, 8, 0) + struct.pack("<QQQQ", 0, jld2.UNDEFINED,
                                                                                 root_addr + len(root), root_addr)
    sb += struct.pack("<I", jld2.lookup3(sb))
    blob = sb + body
    blob += b"\0" * (root_addr - len(blob)) + root
    path = str(tmp_path / "h5.jld2")
    open(path, "wb").write(blob)
    assert np.array_equal(jld2.load_object(path), a)


def test_index_directory_uses_the_reference_file_names(tmp_path):
    """save_codec / save_chunk write exactly the files ColBERT.jl's loaders open (src/loaders.jl:10-38,76-140)."""
    d = str(tmp_path / "idx"); os.makedirs(d)
    rng = np.random.default_rng(5)
    C = np.asfortranarray(rng.standard_normal((128, 16)).astype(np.float32))
    storage.save_codec(d, C, np.float32([-.1, 0, .1]), np.float32([-.2, -.05, .05, .2]), np.float32(0.03))
    storage.save_chunk(d, rng.integers(1, 17, 40).astype(np.uint32), rng.integers(0, 256, (32, 40)).astype(np.uint8), 1, 1,
                       np.array([10, 30]))
    for f in ("centroids.jld2", "avg_residual.jld2", "bucket_cutoffs.jld2", "bucket_weights.jld2", "1.codes.jld2",
              "1.residuals.jld2", "doclens.1.jld2", "1.metadata.json"):
        assert os.path.isfile(os.path.join(d, f)), f
    assert np.array_equal(jld2.load_object(os.path.join(d, "centroids.jld2")), C)
    assert jld2.load_object(os.path.join(d, "avg_residual.jld2")) == np.float32(0.03)
    assert jld2.load_object(os.path.join(d, "doclens.1.jld2")).dtype == np.int64
