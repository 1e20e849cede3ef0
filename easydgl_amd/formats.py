"""On-disk formats of the reference, read without TensorFlow.

* TFRecord files of ``tf.train.Example{seqs_i, seqs_t, seqs_month, seqs_day, seqs_weekday, seqs_hour}`` written by
  ``data/linkpred.py:25-38,172-191`` and decoded by ``src/dataloader.py:11-33`` (``TfExampleDecoder``: FixedLenFeature
  ``seqs_i`` int64 ``[seqslen+1]``, ``seqs_t`` float32 ``[seqslen+1]``).
* ``mark.pkl``: a pickled ``scipy.sparse.csr_matrix`` ``[>= num_items, E]`` that ``EasyDGL.py:45-46`` turns into a dense
  0/1 table with ``pickle.load(...).toarray()``.

TFRecord framing (public format): ``uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)``,
little endian, ``masked(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32``.  ``tf.train.Example`` wire format (protobuf):
``Example{1: Features{1: map<string, Feature>}}``, ``Feature{1: BytesList | 2: FloatList | 3: Int64List}``, each list
``{1: repeated value}`` (packed or not).  Only what the reference's records need is implemented; everything is plain
numpy/Python so that a one-time conversion to ``.npz`` (``convert``) feeds the training driver.
"""
from __future__ import annotations

import glob
import os
import pickle
import struct
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli), table driven
# ---------------------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def _crc_table() -> np.ndarray:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab[i] = c
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------
# TFRecord framing
# ---------------------------------------------------------------------------------------------------------------
def read_tfrecord(path: str, verify: bool = False) -> Iterator[bytes]:
    """Yields the payload of every record of one file.  ``verify`` checks both CRCs (slow: pure Python)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and masked_crc32c(head[:8]) != lcrc:
                raise ValueError(f"{path}: corrupt record length")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError(f"{path}: truncated record")
            if verify and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise ValueError(f"{path}: corrupt record payload")
            yield data


def write_tfrecord(path: str, records: Iterable[bytes]) -> None:
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head)
            f.write(struct.pack("<I", masked_crc32c(head)))
            f.write(data)
            f.write(struct.pack("<I", masked_crc32c(data)))


# ---------------------------------------------------------------------------------------------------------------
# tf.train.Example wire format
# ---------------------------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) for every field of a message; length-delimited values as bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def _int64_list(buf: bytes) -> np.ndarray:
    vals: List[int] = []
    for num, wt, val in _fields(buf):
        if num != 1:
            continue
        if wt == 2:  # packed
            pos = 0
            while pos < len(val):
                v, pos = _varint(val, pos)
                vals.append(v)
        else:
            vals.append(val)
    arr = np.array(vals, dtype=np.uint64)
    return arr.astype(np.int64)  # two's complement: negative int64 are 10-byte varints


def _float_list(buf: bytes) -> np.ndarray:
    chunks: List[np.ndarray] = []
    for num, wt, val in _fields(buf):
        if num != 1:
            continue
        chunks.append(np.frombuffer(val, dtype="<f4"))
    return np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.float32)


def parse_example(buf: bytes) -> Dict[str, np.ndarray]:
    """``tf.io.parse_single_example`` for int64 / float / bytes lists: feature name -> 1-D array (bytes: list)."""
    out: Dict[str, np.ndarray] = {}
    for num, _, features in _fields(buf):
        if num != 1:
            continue
        for fnum, _, entry in _fields(features):
            if fnum != 1:
                continue
            name, feat = None, b""
            for enum, _, eval_ in _fields(entry):
                if enum == 1:
                    name = eval_.decode("utf-8")
                elif enum == 2:
                    feat = eval_
            for knum, _, lst in _fields(feat):
                if knum == 3:
                    out[name] = _int64_list(lst)
                elif knum == 2:
                    out[name] = _float_list(lst)
                elif knum == 1:
                    out[name] = [v for n_, _, v in _fields(lst) if n_ == 1]
    return out


def _enc_varint(v: int) -> bytes:
    v &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_example(features: Dict[str, np.ndarray]) -> bytes:
    """Serialises ``{name: int64 array | float32 array}`` as a ``tf.train.Example`` (packed lists, as TensorFlow writes)."""
    entries = b""
    for name, arr in features.items():
        arr = np.asarray(arr)
        if arr.dtype.kind in "iu":
            lst = _enc_ld(1, b"".join(_enc_varint(int(v)) for v in arr.tolist()))
            feat = _enc_ld(3, lst)
        else:
            feat = _enc_ld(2, _enc_ld(1, arr.astype("<f4").tobytes()))
        entries += _enc_ld(1, _enc_ld(1, name.encode("utf-8")) + _enc_ld(2, feat))
    return _enc_ld(1, entries)


# ---------------------------------------------------------------------------------------------------------------
# datasets
# ---------------------------------------------------------------------------------------------------------------
def expand(patterns) -> List[str]:
    """File patterns as the reference passes them (``--train "path/train*.tfrec"``; comma separated lists allowed)."""
    if isinstance(patterns, str):
        patterns = patterns.split(",")
    files: List[str] = []
    for p in patterns:
        hit = sorted(glob.glob(p))
        files += hit if hit else ([p] if os.path.exists(p) else [])
    if not files:
        raise FileNotFoundError(f"no input files match {patterns}")
    return files


def load_sequences(patterns, seqslen: int, verify: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """All records of the matching files as ``seqs_i`` int64 ``[N, seqslen+1]``, ``seqs_t`` float32 ``[N, seqslen+1]``
    (``TfExampleDecoder(seqslen + 1)``, util.py:121-124).  ``.npz`` files written by ``convert`` are read directly."""
    T = seqslen + 1
    ids, ts = [], []
    for path in expand(patterns):
        if path.endswith(".npz"):
            z = np.load(path)
            a, b = z["seqs_i"].astype(np.int64), z["seqs_t"].astype(np.float32)
            if a.shape[1] != T:
                raise ValueError(f"{path}: sequences have {a.shape[1]} positions, the model expects seqslen+1 = {T}")
            ids.append(a)
            ts.append(b)
            continue
        for rec in read_tfrecord(path, verify=verify):
            ex = parse_example(rec)
            a, b = ex["seqs_i"], ex["seqs_t"]
            if a.shape[0] != T or b.shape[0] != T:   # FixedLenFeature([seqslen]) would raise as well
                raise ValueError(f"{path}: record with {a.shape[0]} tokens, expected {T}")
            ids.append(a[None].astype(np.int64))
            ts.append(b[None].astype(np.float32))
    return np.concatenate(ids, 0), np.concatenate(ts, 0)


def load_mark_table(path: str, num_items: int = None) -> np.ndarray:
    """``pickle.load(open(FLAGS.mark, 'rb')).toarray()`` (EasyDGL.py:45-46) as uint8 ``[rows, E]``; ``num_items`` only
    checks that every item id has a row.  An ``.npy`` file holding the dense table is accepted too."""
    if path.endswith(".npy"):
        tab = np.load(path)
    else:
        with open(path, "rb") as f:
            obj = pickle.load(f)
        tab = obj.toarray() if hasattr(obj, "toarray") else np.asarray(obj)
    if tab.ndim != 2:
        raise ValueError(f"{path}: mark table must be 2-D, got shape {tab.shape}")
    if num_items is not None and tab.shape[0] < num_items:
        raise ValueError(f"{path}: mark table has {tab.shape[0]} rows, need >= num_items = {num_items}")
    if tab.min() < 0 or tab.max() > 255:
        raise ValueError(f"{path}: mark values outside [0, 255]")
    return np.ascontiguousarray(tab.astype(np.uint8))


def convert(patterns, seqslen: int, out_npz: str, verify: bool = True) -> Tuple[int, int]:
    """One-time TFRecord -> ``.npz`` conversion (``seqs_i`` int64, ``seqs_t`` float32)."""
    ids, ts = load_sequences(patterns, seqslen, verify=verify)
    np.savez_compressed(out_npz, seqs_i=ids, seqs_t=ts)
    return ids.shape


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="TFRecord -> npz conversion of the reference's sequence files")
    ap.add_argument("--input", required=True, help="file pattern(s), comma separated")
    ap.add_argument("--seqslen", type=int, required=True)
    ap.add_argument("--output", required=True)
    a = ap.parse_args()
    print("converted", convert(a.input, a.seqslen, a.output))
