"""Kaldi vector ark/scp reader + writer for float embeddings (kaldiio / kaldi_io are not
installed here).  On-disk compatibility with the reference's tools:

  writer  <-> kaldiio.WriteHelper('ark,scp:...') at bin/extract.py:110-111,137-139 and
              cli/speaker.py:375-382
  reader  <-> kaldiio.load_scp_sequential in utils/plda/plda_utils.py:20-29

Binary vector record:  `key` ' ' '\\0' 'B' ('FV ' | 'DV ') '\\x04' <int32 dim> <dim x f32|f64 LE>;
scp line: `key path:offset` with offset pointing at the '\\0B' marker.
"""
import os
import struct
from collections import OrderedDict

import numpy as np


class VectorWriter:
    """with VectorWriter(ark_path, scp_path) as w: w(key, vec)"""

    def __init__(self, ark_path, scp_path=None):
        self.ark_path = os.path.abspath(ark_path)
        self._ark = open(ark_path, "wb")
        self._scp = open(scp_path, "w") if scp_path else None

    def __call__(self, key, vec):
        vec = np.ascontiguousarray(vec)
        if vec.dtype == np.float64:
            tag, data = b"DV ", vec.astype("<f8").tobytes()
        else:
            tag, data = b"FV ", vec.astype("<f4").tobytes()
        self._ark.write(key.encode() + b" ")
        offset = self._ark.tell()
        self._ark.write(b"\0B" + tag + b"\x04" + struct.pack("<i", vec.shape[0]) + data)
        if self._scp:
            self._scp.write("%s %s:%d\n" % (key, self.ark_path, offset))

    def close(self):
        self._ark.close()
        if self._scp:
            self._scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_vectors(keys, mat, ark_path, scp_path=None):
    """All rows of `mat` (n, D) under `keys` as one ark (+ scp): the bytes VectorWriter writes record by record, built
    in one pass (a 4 096-utterance list: 2 ms instead of 15 -- a fifth of the time the MI355X needs to embed it)."""
    mat = np.ascontiguousarray(mat)
    if mat.dtype == np.float64:
        tag, data = b"DV ", mat.astype("<f8", copy=False)
    else:
        tag, data = b"FV ", mat.astype("<f4", copy=False)
    n, dim = data.shape
    assert len(keys) == n
    head = b"\0B" + tag + b"\x04" + struct.pack("<i", dim)       # ('\4' = the size of the dimension field)
    row_bytes = data.dtype.itemsize * dim
    kb = [k.encode() + b" " for k in keys]
    # every record into one preallocated byte array: the row payloads by ONE strided numpy copy when the keys have
    # the same length (the usual case), by slices otherwise
    klen = np.fromiter((len(k) for k in kb), dtype=np.int64, count=n)
    rec = klen + len(head) + row_bytes
    starts = np.concatenate(([0], np.cumsum(rec)[:-1]))
    buf = np.empty(int(rec.sum()), dtype=np.uint8)
    rows = data.view(np.uint8).reshape(n, row_bytes)
    if n and (klen == klen[0]).all():
        r0 = int(rec[0])
        table = buf.reshape(n, r0)
        pre = np.frombuffer(b"".join(kb), dtype=np.uint8).reshape(n, int(klen[0]))
        table[:, :klen[0]] = pre
        table[:, klen[0]:klen[0] + len(head)] = np.frombuffer(head, dtype=np.uint8)
        table[:, klen[0] + len(head):] = rows
    else:
        hb = np.frombuffer(head, dtype=np.uint8)
        for i in range(n):
            o = int(starts[i])
            buf[o:o + klen[i]] = np.frombuffer(kb[i], dtype=np.uint8)
            buf[o + klen[i]:o + klen[i] + len(head)] = hb
            buf[o + klen[i] + len(head):o + rec[i]] = rows[i]
    with open(ark_path, "wb") as f:
        f.write(buf.data)
    if scp_path:
        offs = starts + klen
        ark_abs = os.path.abspath(ark_path)
        with open(scp_path, "w") as f:
            f.write("".join(["%s %s:%d\n" % (k, ark_abs, o) for k, o in zip(keys, offs.tolist())]))


def _read_vector_at(f):
    marker = f.read(2)
    if marker != b"\0B":
        # text form: "[ 1 2 3 ]"
        rest = marker + f.readline()
        toks = rest.decode().replace("[", " ").replace("]", " ").split()
        return np.array([float(t) for t in toks], dtype=np.float32)
    tag = f.read(3)
    if tag == b"FV ":
        dt, size = "<f4", 4
    elif tag == b"DV ":
        dt, size = "<f8", 8
    else:
        raise ValueError("not a Kaldi vector record (tag %r)" % tag)
    if f.read(1) != b"\x04":
        raise ValueError("bad int-size byte in Kaldi vector record")
    dim = struct.unpack("<i", f.read(4))[0]
    return np.frombuffer(f.read(dim * size), dtype=dt).copy()


def read_vec_scp(scp_path):
    """scp -> OrderedDict key -> vector (order of the scp preserved)."""
    out = OrderedDict()
    handles = {}
    try:
        with open(scp_path, "r") as scp:
            for line in scp:
                line = line.strip()
                if not line:
                    continue
                key, loc = line.split(None, 1)
                path, _, off = loc.rpartition(":")
                if not path:
                    path, off = loc, "0"
                f = handles.get(path)
                if f is None:
                    f = handles[path] = open(path, "rb")
                f.seek(int(off))
                out[key] = _read_vector_at(f)
    finally:
        for f in handles.values():
            f.close()
    return out


def read_vec_ark(ark_path):
    """sequential binary ark -> OrderedDict."""
    out = OrderedDict()
    with open(ark_path, "rb") as f:
        while True:
            key = b""
            while True:
                c = f.read(1)
                if not c:
                    return out
                if c == b" ":
                    break
                key += c
            out[key.decode()] = _read_vector_at(f)
