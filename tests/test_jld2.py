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


# ---------------------------------------------------------------------------------------------------------------------
# Field-by-field conformance of what the writer emits with the HDF5 file-format specification (version 3.0, sections
# II.A superblock, IV.A.1.b version-2 object header, IV.A.2 header messages) -- parsed HERE from the specification,
# independently of jld2._File, plus the constants JLD2.jl itself uses for plain numbers (its h5fieldtype methods:
# class | version 3 << 4, bit field 0x20 / 0x1f for IEEE floats, 0x08 for signed integers).  What JLD2.jl would have to
# accept when it opens one of these files is exactly these structures (src/loaders.jl:10-38 -> JLD2.load_object).
# ---------------------------------------------------------------------------------------------------------------------
def _walk_v2_header(buf, p):
    """[(type, flags, body)] of the version-2 object header at absolute offset p; asserts the prefix fields."""
    assert buf[p:p + 4] == b"OHDR" and buf[p + 4] == 2
    flags = buf[p + 5]
    assert flags & 0xC0 == 0                                   # reserved bits
    assert flags & 0x20 == 0 and flags & 0x10 == 0            # no times, no attribute phase-change fields
    nsz = 1 << (flags & 3)
    size = int.from_bytes(buf[p + 6:p + 6 + nsz], "little")
    q = p + 6 + nsz
    assert struct.unpack_from("<I", buf, q + size)[0] == jld2.lookup3(buf[p:q + size])
    out, lo, hi = [], q, q + size
    while lo < hi:
        mtype, msize, mflags = buf[lo], struct.unpack_from("<H", buf, lo + 1)[0], buf[lo + 3]
        assert flags & 0x04 == 0                               # creation order not tracked: 4-byte message headers
        out.append((mtype, mflags, buf[lo + 4:lo + 4 + msize]))
        lo += 4 + msize
    assert lo == hi, "messages must fill the chunk exactly (no gap smaller than a message header)"
    return out, q + size + 4


@pytest.mark.parametrize("arr,julia_type", [
    (np.float32(0.5), "Float32"),
    (np.asfortranarray(np.arange(128 * 40, dtype=np.float32).reshape(128, 40, order="F")), "Matrix{Float32}"),
    (np.arange(5, dtype=np.float32), "Vector{Float32}"),
    (np.arange(7000, dtype=np.uint32), "Vector{UInt32}"),
    (np.asfortranarray(np.arange(32 * 500, dtype=np.uint8).reshape(32, 500, order="F")), "Matrix{UInt8}"),
    (np.arange(3000, dtype=np.int64), "Vector{Int64}"),
])
def test_written_structures_follow_the_hdf5_specification(tmp_path, arr, julia_type):
    path = str(tmp_path / "s.jld2")
    jld2.save_object(path, arr)
    buf = open(path, "rb").read()
    a = np.asarray(arr)
    # --- JLD2 text header: 512 bytes, version string, NUL-terminated
    assert len(buf) > 512 and buf[:37] == b"HDF5-based Julia Data Format, version " [:37]
    assert buf[38:43] == b"0.1.1" and buf[43] == 0
    # --- superblock version 2 (spec II.A): signature, version, size of offsets / lengths, consistency flags, four addresses
    sb = 512
    assert buf[sb:sb + 8] == b"\x89HDF\r\n\x1a\n"
    ver, so, sl, cflags = buf[sb + 8:sb + 12]
    assert (ver, so, sl, cflags) == (2, 8, 8, 0)
    base, ext, eof, root = struct.unpack_from("<QQQQ", buf, sb + 12)
    assert base == 512 and ext == 0xFFFFFFFFFFFFFFFF          # base address = where the superblock sits (JLD2's offset)
    assert base + eof == len(buf) and root % 8 == 0
    assert struct.unpack_from("<I", buf, sb + 44)[0] == jld2.lookup3(buf[sb:sb + 44])
    # --- root group: link info (v0, no creation order, no dense storage), group info (v0), ONE hard link
    msgs, _ = _walk_v2_header(buf, base + root)
    types = [m[0] for m in msgs]
    assert types == [2, 10, 6]
    li = msgs[0][2]
    assert li[0] == 0 and li[1] == 0 and struct.unpack_from("<QQ", li, 2) == (2 ** 64 - 1, 2 ** 64 - 1) and len(li) == 18
    assert msgs[1][2] == b"\x00\x00"
    lk = msgs[2][2]
    assert lk[0] == 1                                          # link message version 1
    assert lk[1] == 0x00                                       # hard link, 1-byte name length, no creation order / charset
    assert lk[2] == 20 and lk[3:23] == b"single_stored_object"
    ds_rel = struct.unpack_from("<Q", lk, 23)[0]
    assert len(lk) == 31 and ds_rel == 48                      # the dataset header follows the 48-byte superblock
    # --- dataset: dataspace, datatype (shared-message flag bit 0 set as constant), layout
    msgs, end = _walk_v2_header(buf, base + ds_rel)
    assert [m[0] for m in msgs] == [1, 3, 8]
    sp = msgs[0][2]
    assert sp[0] == 2 and sp[2] == 0                           # dataspace version 2, no maximum dimensions
    if a.ndim == 0:
        assert sp[1] == 0 and sp[3] == 0 and len(sp) == 4      # scalar
    else:
        assert sp[1] == a.ndim and sp[3] == 1 and len(sp) == 4 + 8 * a.ndim
        dims = struct.unpack_from("<" + "Q" * a.ndim, sp, 4)
        assert dims == tuple(reversed(a.shape))                # Julia (column-major) dims reversed: fastest LAST
    dt = msgs[1][2]
    cls, version = dt[0] & 0x0F, dt[0] >> 4
    assert version == 3                                        # what JLD2.jl writes; HDF5 accepts 1..3
    assert struct.unpack_from("<I", dt, 4)[0] == a.dtype.itemsize
    if a.dtype.kind == "f":
        # IEEE little-endian: byte order 0, padding 0, mantissa normalisation 2 (implied MSB), sign location 31
        assert cls == 1 and dt[1] == 0x20 and dt[2] == 31 and dt[3] == 0
        assert struct.unpack_from("<HHBBBBI", dt, 8) == (0, 32, 23, 8, 0, 23, 127) and len(dt) == 20
    else:
        assert cls == 0 and dt[1] == (0x08 if a.dtype.kind == "i" else 0x00) and dt[2] == 0 and dt[3] == 0
        assert struct.unpack_from("<HH", dt, 8) == (0, 8 * a.dtype.itemsize) and len(dt) == 12
    lay = msgs[2][2]
    raw = np.asfortranarray(a).tobytes(order="F")
    assert lay[0] == 3                                         # data layout message version 3
    if lay[1] == 0:                                            # compact: size (2 bytes) + the data inside the header
        assert struct.unpack_from("<H", lay, 2)[0] == len(raw) and lay[4:] == raw and len(raw) < 8192
    else:                                                      # contiguous: address (relative to base) + size
        assert lay[1] == 1 and len(lay) == 18
        addr, size = struct.unpack_from("<QQ", lay, 2)
        assert size == len(raw) and addr % 8 == 0 and base + addr >= end
        assert buf[base + addr:base + addr + size] == raw
        assert base + addr + size <= base + root               # data lies between the dataset header and the root group


def test_npy_index_directories_of_round_one_still_load(tmp_path):
    a = np.arange(12, dtype=np.int64)
    np.save(str(tmp_path / "ivf.npy"), a)
    assert np.array_equal(storage._load(str(tmp_path / "ivf")), a)
