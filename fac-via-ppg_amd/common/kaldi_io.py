"""Readers for the Kaldi binary objects the PPG front-end ships with (data/feats/final.mat, reduce_dim.mat): what the
reference gets from ``kaldi.util.io.read_matrix`` (compute_ppg.py:25) and ``feat.read_sparse_mat`` (feat.py:159-170).

Binary layout (Kaldi matrix/kaldi-matrix.cc, sparse-matrix.cc): ``\\0B`` then a token --
  ``FM `` float matrix: <4><int32 rows> <4><int32 cols> rows*cols float32, row-major
  ``SM `` sparse float matrix: <4><int32 rows>, then per row ``SV `` <4><int32 dim> <4><int32 n> n x (<4><int32 index> <4><float32 value>)
(the ``<4>`` bytes are Kaldi's size markers of the basic types)."""
import struct

import numpy as np


class KaldiFormatError(ValueError):
    pass


def _header(buf, path):
    if buf[:2] != b"\x00B":
        raise KaldiFormatError("%s: not a Kaldi binary object (text-mode objects are not supported)" % path)
    end = buf.index(b" ", 2)
    return buf[2:end].decode(), end + 1


def _int(buf, off, path):
    if buf[off] != 4:
        raise KaldiFormatError("%s: expected a 4-byte integer marker at offset %d" % (path, off))
    return struct.unpack_from("<i", buf, off + 1)[0], off + 5


def read_matrix(path):
    """Kaldi float matrix -> float32 ndarray [rows, cols]."""
    with open(path, "rb") as f:
        buf = f.read()
    token, off = _header(buf, path)
    if token != "FM":
        raise KaldiFormatError("%s: token %r, expected a float matrix (FM)" % (path, token))
    rows, off = _int(buf, off, path)
    cols, off = _int(buf, off, path)
    if len(buf) - off != rows * cols * 4:
        raise KaldiFormatError("%s: %d x %d matrix needs %d bytes, file has %d" % (path, rows, cols, rows * cols * 4, len(buf) - off))
    return np.frombuffer(buf, dtype="<f4", count=rows * cols, offset=off).reshape(rows, cols).copy()


def read_sparse_matrix(path):
    """Kaldi sparse float matrix -> dense float32 ndarray [rows, dim] (the reference densifies it before use,
    compute_ppg.py:86-89)."""
    with open(path, "rb") as f:
        buf = f.read()
    token, off = _header(buf, path)
    if token != "SM":
        raise KaldiFormatError("%s: token %r, expected a sparse matrix (SM)" % (path, token))
    rows, off = _int(buf, off, path)
    out = None
    for r in range(rows):
        if buf[off:off + 3] != b"SV ":
            raise KaldiFormatError("%s: row %d does not start with a sparse vector" % (path, r))
        off += 3
        dim, off = _int(buf, off, path)
        n, off = _int(buf, off, path)
        if out is None:
            out = np.zeros((rows, dim), dtype=np.float32)
        elif dim != out.shape[1]:
            raise KaldiFormatError("%s: ragged sparse matrix" % path)
        rec = np.frombuffer(buf, dtype=np.dtype([("m1", "u1"), ("i", "<i4"), ("m2", "u1"), ("v", "<f4")]), count=n, offset=off)
        if n and (np.any(rec["m1"] != 4) or np.any(rec["m2"] != 4)):
            raise KaldiFormatError("%s: bad element markers in row %d" % (path, r))
        out[r, rec["i"]] = rec["v"]
        off += n * 10
    if off != len(buf):
        raise KaldiFormatError("%s: %d trailing bytes" % (path, len(buf) - off))
    return out
