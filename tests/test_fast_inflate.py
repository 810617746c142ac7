"""The host-side zlib-stream decoder of the NetCDF-4 read path (csrc/fast_inflate.cpp, reached here through
ecckd_inflate_host) against Python's zlib: every block type, every compression level and strategy, long codes, runs, random
bytes, the shuffled-FLOAT data it is meant for - and streams that are damaged, cut short or of another length, which it must
refuse (the caller then asks zlib) without reading or writing outside its buffers."""
import ctypes as C
import zlib

import numpy as np
import pytest

from ecckd_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return _lib.load_library()


def _inflate(lib, stream, out_len, guard=64):
    """(ok, bytes): the output buffer sits between two guard zones that must stay untouched."""
    src = np.frombuffer(stream, dtype=np.uint8).copy() if len(stream) else np.zeros(0, np.uint8)
    buf = np.full(out_len + 2 * guard, 0xA5, dtype=np.uint8)
    ok = C.c_int(-1)
    rc = lib.ecckd_inflate_host(src.ctypes.data_as(C.c_void_p), src.size, (buf.ctypes.data + guard), out_len, C.byref(ok))
    assert rc == 0 and ok.value in (0, 1)
    assert (buf[:guard] == 0xA5).all() and (buf[out_len + guard:] == 0xA5).all(), "wrote outside the output buffer"
    return bool(ok.value), buf[guard:guard + out_len].tobytes()


def _payloads():
    rs = np.random.RandomState(7)
    od = np.exp(np.cumsum(rs.normal(0, 0.05, 60_000))).astype("<f4")
    shuffled = od.view(np.uint8).reshape(-1, 4).T.copy().tobytes()          # HDF5's shuffle filter: byte planes
    text = (b"the quick brown fox jumps over the lazy dog " * 400)
    return {
        "empty": b"",
        "one byte": b"x",
        "zeros": bytes(100_000),
        "text": text,
        "random": rs.bytes(70_000),
        "shuffled floats": shuffled,
        "floats": od.tobytes(),
        "period 3": bytes([1, 2, 3]) * 20_000,
        "period 7 then noise": bytes(range(7)) * 3000 + rs.bytes(5000),
        "skewed alphabet": rs.choice(256, 90_000, p=np.r_[[0.5], np.full(255, 0.5 / 255)]).astype(np.uint8).tobytes(),
        "two symbols": rs.choice([0, 255], 50_000).astype(np.uint8).tobytes(),
    }


@pytest.mark.parametrize("name", list(_payloads()))
def test_round_trip_every_level_and_strategy(lib, name):
    data = _payloads()[name]
    for level in (0, 1, 2, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
            for wbits in (15, 9):
                c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9 if level else 1, strategy)
                stream = c.compress(data) + c.flush()
                ok, out = _inflate(lib, stream, len(data))
                assert ok, (name, level, strategy, wbits)
                assert out == data, (name, level, strategy, wbits)


def test_many_blocks_and_sync_flushes(lib):
    rs = np.random.RandomState(3)
    c = zlib.compressobj(6)
    parts, data = [], b""
    for k in range(40):
        piece = (rs.bytes(rs.randint(1, 3000)) if k % 3 else bytes(rs.randint(1, 5000)))
        data += piece
        parts.append(c.compress(piece))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if k % 2 else zlib.Z_FULL_FLUSH))   # empty stored blocks between the others
    stream = b"".join(parts) + c.flush()
    ok, out = _inflate(lib, stream, len(data))
    assert ok and out == data


def test_refuses_what_is_not_a_stream_of_that_length(lib):
    data = _payloads()["shuffled floats"]
    stream = zlib.compress(data, 2)
    assert _inflate(lib, stream, len(data))[0]
    assert not _inflate(lib, stream, len(data) - 1)[0]            # more output than room
    assert not _inflate(lib, stream, len(data) + 1)[0]            # less output than asked for
    assert not _inflate(lib, stream[:-1], len(data))[0]           # cut short
    assert not _inflate(lib, stream[: len(stream) // 2], len(data))[0]
    assert not _inflate(lib, stream + b"\0", len(data))[0]        # bytes behind the checksum
    assert not _inflate(lib, b"", 0)[0] and not _inflate(lib, b"\x78", 0)[0]
    bad_sum = bytearray(stream); bad_sum[-1] ^= 1
    assert not _inflate(lib, bytes(bad_sum), len(data))[0]
    bad_head = bytearray(stream); bad_head[0] = 0x79
    assert not _inflate(lib, bytes(bad_head), len(data))[0]
    raw = zlib.compressobj(2, zlib.DEFLATED, -15)
    assert not _inflate(lib, raw.compress(data) + raw.flush(), len(data))[0]   # a raw deflate stream has no zlib header
    with_dict = zlib.compressobj(2, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, b"dictionary")
    assert not _inflate(lib, with_dict.compress(data) + with_dict.flush(), len(data))[0]


def test_damaged_streams_never_decode_to_something_else(lib):
    """One byte of the stream flipped, at every 97th position: refused (then zlib would refuse it too), or - where the flip
    happens to leave a valid stream of the same length and checksum, which cannot happen for a single byte - the same data."""
    rs = np.random.RandomState(11)
    for name in ("shuffled floats", "text", "random"):
        data = _payloads()[name]
        stream = zlib.compress(data, 2)
        for pos in range(0, len(stream), 97):
            damaged = bytearray(stream)
            damaged[pos] ^= 1 << rs.randint(8)
            ok, out = _inflate(lib, bytes(damaged), len(data))
            if ok:
                assert out == data
            try:
                ref = zlib.decompress(bytes(damaged))
            except zlib.error:
                ref = None
            assert ok == (ref == data)


def test_long_codes_and_far_matches(lib):
    """An alphabet whose frequencies fall off geometrically gives 15-bit codes (second-level tables); matches at the far end of
    the window and of the maximum length."""
    rs = np.random.RandomState(5)
    p = 0.5 ** np.arange(1, 41); p /= p.sum()
    skew = rs.choice(40, 400_000, p=p).astype(np.uint8).tobytes()
    block = rs.bytes(300)
    far = block + rs.bytes(32_768 - 300) + block + block * 10
    for data in (skew, far):
        for level in (1, 9):
            stream = zlib.compress(data, level)
            ok, out = _inflate(lib, stream, len(data))
            assert ok and out == data


def test_incomplete_codes_are_refused_like_zlib(lib):
    """A dynamic block whose literal/length code leaves part of the code space unassigned: zlib's inflate_table refuses it even
    if the missing codes never occur ("incomplete literal/length tree"); so does this decoder.  Built by hand: code lengths
    {'a': 1, end-of-block: 2} (one 2-bit code unassigned), one distance code of length 1 (allowed to stand alone)."""
    bits = []
    def put(value, n):
        for i in range(n):
            bits.append((value >> i) & 1)
    put(1, 1); put(2, 2)                      # final block, dynamic
    put(0, 5); put(0, 5); put(14, 4)          # HLIT = 257, HDIST = 1, HCLEN = 18 code-length codes
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = {0: 2, 1: 2, 2: 2, 18: 2}            # a complete code-length code: four 2-bit codes for 0, 1, 2, 18
    for sym in order[:18]:
        put(cl.get(sym, 0), 3)
    code = {0: 0b00, 1: 0b01, 2: 0b10, 18: 0b11}                 # canonical codes (MSB first) of those four
    def sym(s_):
        c = code[s_]
        put((c >> 1) & 1, 1); put(c & 1, 1)
    # lengths: 97 zeros, 'a' = 1, 158 zeros, end-of-block = 2, then the one distance code = 1
    sym(18); put(97 - 11, 7)
    sym(1)
    sym(18); put(138 - 11, 7)
    sym(18); put(20 - 11, 7)
    sym(2)
    sym(1)
    put(0, 1)                                  # 'a'
    put(0b01, 2)                               # end of block = 10 sent MSB first: bits 1, 0
    while len(bits) % 8:
        bits.append(0)
    body = bytes(sum(b << i for i, b in enumerate(bits[k:k + 8])) for k in range(0, len(bits), 8))
    stream = b"\x78\x9c" + body + zlib.adler32(b"a").to_bytes(4, "big")
    with pytest.raises(zlib.error):
        zlib.decompress(stream)
    assert not _inflate(lib, stream, 1)[0]
