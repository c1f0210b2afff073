"""Host-side data formats (vsrbac/formats.py) against the reference's own expectations:
pgvector's text I/O known answers (tests/golden/pgvector_vector_io.json, transcribed from
pgvector/test/expected/vector_type.out:1-160), the binary layout of vector_send / vector_recv (vector.c:363-411) and the
shared_vectors.bin layout of the C++ benches (shared_vector_table.cpp:169-240)."""
import json
import os
import struct

import numpy as np
import pytest

from vsrbac import formats


@pytest.fixture(scope="module")
def io_cases(golden_dir):
    with open(os.path.join(golden_dir, "pgvector_vector_io.json")) as f:
        return json.load(f)


def test_text_known_answers(io_cases):
    for lit, out in io_cases["text_ok"]:
        assert formats.vector_to_text(formats.vector_from_text(lit)) == out, lit
    for lit, msg in io_cases["text_error"]:
        with pytest.raises(ValueError) as e:
            formats.vector_from_text(lit)
        assert str(e.value).split("\n")[0] == msg, lit
        if lit in io_cases["details"]:
            assert str(e.value).split("\n")[1] == "DETAIL:  " + io_cases["details"][lit]
    for lit, dim, out in io_cases["typmod"]:
        if out.startswith("["):
            assert formats.vector_to_text(formats.vector_from_text(lit, dim)) == out
        else:
            with pytest.raises(ValueError) as e:
                formats.vector_from_text(lit, dim)
            assert str(e.value) == out


def test_text_output_is_shortest_round_trip_float4():
    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.normal(size=200).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200),
                           np.asarray([1e6, 999999.0, 1e-4, 9.9e-5, 123456.7, 1e5, 3.4028235e38, 1e-45, 0.1, 100.0],
                                      dtype=np.float32)]).astype(np.float32)
    vals = vals[np.isfinite(vals)]
    text = formats.vector_to_text(vals)
    back = formats.vector_from_text(text)
    np.testing.assert_array_equal(back, vals)                 # round trip is exact
    # printf-like thresholds of PostgreSQL's float4 output: fixed for 1e-4 <= |x| < 1e6, exponent form outside
    assert formats.vector_to_text([1e6, 999999, 1e-4, 9.9e-5, 100]) == "[1e+06,999999,0.0001,9.9e-05,100]"
    with pytest.raises(ValueError, match="cannot have more than 16000"):
        formats.vector_from_text("[" + ",".join(["1"] * 16001) + "]")


def test_binary_layout_and_checks():
    v = np.asarray([1.5, -2.0, 3.25], dtype=np.float32)
    b = formats.vector_to_binary(v)
    assert b == struct.pack(">hh", 3, 0) + struct.pack(">fff", 1.5, -2.0, 3.25)      # int16 dim, int16 0, float4 BE
    np.testing.assert_array_equal(formats.vector_from_binary(b), v)
    with pytest.raises(ValueError, match="expected unused to be 0, not 7"):
        formats.vector_from_binary(struct.pack(">hh", 3, 7) + b[4:])
    with pytest.raises(ValueError, match="NaN not allowed in vector"):
        formats.vector_from_binary(struct.pack(">hh", 1, 0) + struct.pack(">f", float("nan")))
    with pytest.raises(ValueError, match="infinite value not allowed in vector"):
        formats.vector_from_binary(struct.pack(">hh", 1, 0) + struct.pack(">f", float("inf")))
    with pytest.raises(ValueError, match="at least 1 dimension"):
        formats.vector_from_binary(struct.pack(">hh", 0, 0))
    with pytest.raises(ValueError, match="expected 2 dimensions, not 3"):
        formats.vector_from_binary(b, expected_dim=2)


def test_shared_vectors_file_round_trip(tmp_path):
    rng = np.random.default_rng(4)
    rows = rng.normal(size=(37, 12)).astype(np.float32)
    doc = rng.integers(1, 9, 37).astype(np.int32)
    blk = np.arange(37, dtype=np.int32) + 100
    path = str(tmp_path / "shared_vectors.bin")
    formats.write_shared_vectors(path, rows, doc, blk)
    raw = open(path, "rb").read()
    assert struct.unpack("<iq", raw[:12]) == (12, 37) and len(raw) == 12 + 37 * 12 * 4       # int32 dim, int64 count, floats
    meta = open(path + ".meta", "rb").read()
    assert struct.unpack("<iq", meta[:12]) == (12, 37) and struct.unpack("<ii", meta[12:20]) == (int(doc[0]), 100)
    for mm in (True, False):
        r, d, b = formats.read_shared_vectors(path, mmap=mm)
        np.testing.assert_array_equal(np.asarray(r), rows)
        np.testing.assert_array_equal(d, doc)
        np.testing.assert_array_equal(b, blk)
    with open(path + ".meta", "r+b") as f:
        f.write(struct.pack("<iq", 11, 37))
    with pytest.raises(ValueError, match="mismatch"):
        formats.read_shared_vectors(path)
