"""Data formats on either side of the search path: what the reference's callers hold when they reach
`vsr_corpus_load` / `vsr_search`.  Host-side only; nothing here touches the GPU.

  * pgvector text form   `[a,b,...]`   vector_in / vector_out   (pgvector/src/vector.c:165-270, 278-315)
  * pgvector binary form  int16 dim, int16 unused (= 0), float4[dim] big-endian   vector_recv / vector_send  (:363-411)
  * shared_vectors.bin (+ .meta) of the C++ benches   SharedVectorTable::save_vectors / load_vectors
    (logical_partition_benchmark/benchmark/src/shared_vector_table.cpp:169-201): int32 dim, int64 count, float32[count*dim];
    .meta = int32 dim, int64 count, (int32 document_id, int32 block_id)[count]

Errors are ValueError with pgvector's message texts (the shim maps them to ereport).
"""
import ctypes
import ctypes.util
import math
import struct

import numpy as np

VECTOR_MAX_DIM = 16000                                    # vector.h:4
_SPACE = " \t\n\r\v\f"                                    # vector_isspace, vector.c:144-158

_libc = ctypes.CDLL(ctypes.util.find_library("c") or None, use_errno=True)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]
_ERANGE = 34


def _check_element(v):
    if math.isnan(v):
        raise ValueError("NaN not allowed in vector")                         # vector.c:101-113
    if math.isinf(v):
        raise ValueError("infinite value not allowed in vector")


def _check_dim(dim):
    if dim < 1:
        raise ValueError("vector must have at least 1 dimension")             # vector.c:85-96
    if dim > VECTOR_MAX_DIM:
        raise ValueError(f"vector cannot have more than {VECTOR_MAX_DIM} dimensions")


def _check_expected(expected_dim, dim):
    if expected_dim is not None and expected_dim != -1 and dim != expected_dim:
        raise ValueError(f"expected {expected_dim} dimensions, not {dim}")    # CheckExpectedDim, vector.c:72-80


def vector_from_text(lit, expected_dim=None):
    """vector_in: same grammar, same number parser (libc strtof: no double rounding), same error texts."""
    if isinstance(lit, bytes):
        lit = lit.decode()
    raw = lit.encode()
    bad = f'invalid input syntax for type vector: "{lit}"'
    n = len(raw)
    pos = 0

    def skip(p):
        while p < n and chr(raw[p]) in _SPACE:
            p += 1
        return p

    pos = skip(pos)
    if pos >= n or raw[pos:pos + 1] != b"[":
        raise ValueError(bad + '\nDETAIL:  Vector contents must start with "[".')
    pos = skip(pos + 1)
    if raw[pos:pos + 1] == b"]":
        raise ValueError("vector must have at least 1 dimension")
    out = []
    buf = ctypes.create_string_buffer(raw + b"\0")
    base = ctypes.addressof(buf)
    while True:
        if len(out) == VECTOR_MAX_DIM:
            raise ValueError(f"vector cannot have more than {VECTOR_MAX_DIM} dimensions")
        pos = skip(pos)
        if pos >= n:
            raise ValueError(bad)
        end = ctypes.c_char_p()
        ctypes.set_errno(0)
        val = _libc.strtof(ctypes.c_char_p(base + pos), ctypes.byref(end))
        stop = ctypes.cast(end, ctypes.c_void_p).value - base
        if stop == pos:
            raise ValueError(bad)
        if ctypes.get_errno() == _ERANGE and math.isinf(val):
            raise ValueError(f'"{raw[pos:stop].decode()}" is out of range for type vector')
        _check_element(val)
        out.append(val)
        pos = skip(stop)
        c = raw[pos:pos + 1]
        if c == b",":
            pos += 1
        elif c == b"]":
            pos += 1
            break
        else:
            raise ValueError(bad)
    pos = skip(pos)
    if pos != n:
        raise ValueError(bad + "\nDETAIL:  Junk after closing right brace.")
    _check_dim(len(out))
    _check_expected(expected_dim, len(out))
    return np.asarray(out, dtype=np.float32)


def _float4_shortest(v):
    """float_to_shortest_decimal_bufn for float4 (PostgreSQL's Ryu f2s): shortest round-trip digits, fixed notation for
    decimal exponents in [-4, 6), else d.ddde+XX with at least two exponent digits."""
    v = np.float32(v)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    sign = "-" if np.signbit(v) else ""
    a = abs(v)
    if a == 0:
        return sign + "0"
    sci = np.format_float_scientific(a, unique=True, trim="-", exp_digits=1)   # e.g. '1.5e+38', '1.e+00' -> trimmed
    mant, exp = sci.split("e")
    exp = int(exp)
    digits = mant.replace(".", "")
    if -4 <= exp < 6:
        if exp >= 0:
            whole = digits[:exp + 1].ljust(exp + 1, "0")
            frac = digits[exp + 1:]
        else:
            whole = "0"
            frac = "0" * (-exp - 1) + digits
        return sign + whole + ("." + frac if frac else "")
    body = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
    return f"{sign}{body}e{'+' if exp >= 0 else '-'}{abs(exp):02d}"


def vector_to_text(v):
    """vector_out."""
    v = np.asarray(v, dtype=np.float32).ravel()
    return "[" + ",".join(_float4_shortest(x) for x in v) + "]"


def vector_from_binary(b, expected_dim=None):
    """vector_recv."""
    if len(b) < 4:
        raise ValueError("insufficient data left in message")
    dim, unused = struct.unpack(">hh", b[:4])
    _check_dim(dim)
    _check_expected(expected_dim, dim)
    if unused != 0:
        raise ValueError(f"expected unused to be 0, not {unused}")
    if len(b) != 4 + 4 * dim:
        raise ValueError("insufficient data left in message" if len(b) < 4 + 4 * dim else "incorrect binary data format")
    x = np.frombuffer(b, dtype=">f4", count=dim, offset=4).astype(np.float32)
    for e in x:
        _check_element(float(e))
    return x


def vector_to_binary(v):
    """vector_send."""
    v = np.asarray(v, dtype=np.float32).ravel()
    _check_dim(v.size)
    return struct.pack(">hh", v.size, 0) + v.astype(">f4").tobytes()


def write_shared_vectors(path, rows, doc_ids, block_ids):
    """SharedVectorTable::save_vectors (shared_vector_table.cpp:169-201); used to make test inputs."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n, dim = rows.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iq", dim, n))
        f.write(rows.tobytes())
    ids = np.empty((n, 2), dtype="<i4")
    ids[:, 0] = doc_ids
    ids[:, 1] = block_ids
    with open(path + ".meta", "wb") as f:
        f.write(struct.pack("<iq", dim, n))
        f.write(ids.tobytes())


def read_shared_vectors(path, mmap=True):
    """SharedVectorTable::load_vectors: (rows float32 [n, dim], document_ids int32 [n], block_ids int32 [n]).
    Rows are memory-mapped by default (a 10M x 128 table is 5 GB); `vsr_corpus_load` copies them once."""
    with open(path, "rb") as f:
        head = f.read(12)
    if len(head) != 12:
        raise ValueError(f"Malformed shared vector file: {path}")
    dim, n = struct.unpack("<iq", head)
    if dim < 1 or n < 0:
        raise ValueError(f"Malformed shared vector file: {path}")
    if mmap and n:
        rows = np.memmap(path, dtype="<f4", mode="r", offset=12, shape=(n, dim))
    else:
        rows = np.fromfile(path, dtype="<f4", offset=12, count=n * dim).reshape(n, dim)
    if rows.shape != (n, dim):
        raise ValueError(f"Malformed shared vector file: {path}")
    with open(path + ".meta", "rb") as f:
        mdim, mn = struct.unpack("<iq", f.read(12))
        if (mdim, mn) != (dim, n):
            raise ValueError(f"Metadata mismatch for {path}")
        ids = np.frombuffer(f.read(8 * n), dtype="<i4")
    if ids.size != 2 * n:
        raise ValueError(f"Malformed shared vector metadata: {path}.meta")
    ids = ids.reshape(n, 2)
    return rows, np.ascontiguousarray(ids[:, 0]), np.ascontiguousarray(ids[:, 1])
