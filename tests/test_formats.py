"""TFRecord / tf.train.Example / mark.pkl readers (easydgl_amd/formats.py) — CPU only.  The byte-level expectations
below come from the published formats (TFRecord framing with masked CRC32C; protobuf wire format of tf.train.Example),
not from TensorFlow: the records are produced by the module's own writer and by hand-assembled bytes."""
import os
import pickle
import struct

import numpy as np
import pytest

from easydgl_amd import formats as F


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors for CRC32C
    assert F.crc32c(b"") == 0
    assert F.crc32c(b"123456789") == 0xE3069283
    assert F.crc32c(bytes(32)) == 0x8A9136AA
    assert F.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43


def test_example_wire_format_by_hand():
    # Example{features{feature{key:"a" value{int64_list{value:[1, 300, -1]}}}}} assembled byte by byte
    packed = bytes([0x01]) + bytes([0xAC, 0x02]) + bytes([0xFF] * 9 + [0x01])
    int64_list = bytes([0x0A, len(packed)]) + packed
    feature = bytes([0x1A, len(int64_list)]) + int64_list
    entry = bytes([0x0A, 0x01]) + b"a" + bytes([0x12, len(feature)]) + feature
    features = bytes([0x0A, len(entry)]) + entry
    example = bytes([0x0A, len(features)]) + features
    got = F.parse_example(example)
    np.testing.assert_array_equal(got["a"], np.array([1, 300, -1], dtype=np.int64))
    # the writer produces the same bytes
    assert F.encode_example({"a": np.array([1, 300, -1], dtype=np.int64)}) == example
    # non-packed encoding of the same list is accepted as well
    unpacked = bytes([0x08, 0x01, 0x08, 0xAC, 0x02])
    feature2 = bytes([0x1A, len(unpacked)]) + unpacked
    entry2 = bytes([0x0A, 0x01]) + b"a" + bytes([0x12, len(feature2)]) + feature2
    ex2 = bytes([0x0A, len(entry2) + 2, 0x0A, len(entry2)]) + entry2
    np.testing.assert_array_equal(F.parse_example(ex2)["a"], [1, 300])


def test_tfrecord_round_trip_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    T = 21
    ids = rng.integers(0, 5000, size=(7, T)).astype(np.int64)
    ts = (9.5e8 + rng.random((7, T)) * 1e6).astype(np.float32)
    recs = [F.encode_example({"seqs_i": ids[i], "seqs_t": ts[i], "seqs_hour": np.arange(T)}) for i in range(7)]
    p1, p2 = str(tmp_path / "train000.tfrec"), str(tmp_path / "train001.tfrec")
    F.write_tfrecord(p1, recs[:4])
    F.write_tfrecord(p2, recs[4:])
    # framing: 8-byte little-endian length first
    raw = open(p1, "rb").read()
    assert struct.unpack("<Q", raw[:8])[0] == len(recs[0])
    a, b = F.load_sequences(str(tmp_path / "train*.tfrec"), seqslen=T - 1, verify=True)
    np.testing.assert_array_equal(a, ids)
    np.testing.assert_array_equal(b, ts)
    assert a.dtype == np.int64 and b.dtype == np.float32
    with pytest.raises(ValueError):
        F.load_sequences(p1, seqslen=T, verify=True)      # wrong FixedLenFeature length
    bad = bytearray(raw)
    bad[20] ^= 0x40
    open(p1, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        list(F.read_tfrecord(p1, verify=True))
    # npz conversion round trip
    F.write_tfrecord(p1, recs[:4])
    out = str(tmp_path / "all.npz")
    assert tuple(F.convert(str(tmp_path / "train*.tfrec"), T - 1, out)) == (7, T)
    a2, b2 = F.load_sequences(out, seqslen=T - 1)
    np.testing.assert_array_equal(a2, ids)
    np.testing.assert_array_equal(b2, ts)


def test_mark_table_from_pickled_csr(tmp_path):
    sp = pytest.importorskip("scipy.sparse")
    dense = np.zeros((12, 5), dtype=np.int64)
    dense[np.arange(1, 12), np.arange(1, 12) % 5] = 1
    dense[3, 4] = 1
    p = str(tmp_path / "mark.pkl")
    with open(p, "wb") as f:
        pickle.dump(sp.csr_matrix(dense), f)
    tab = F.load_mark_table(p, num_items=10)
    assert tab.dtype == np.uint8 and tab.shape == (12, 5)
    np.testing.assert_array_equal(tab, dense)
    with pytest.raises(ValueError):
        F.load_mark_table(p, num_items=13)
